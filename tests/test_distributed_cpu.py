"""CPU, world_size 2, gloo: the data-parallel contract of the training path (fruit_pipeline.py:116-118):
identical initial weights on every rank, rank-specific rays, ONE all-reduce over the flat gradient arena whose
mean is applied by the optimiser.  (The kernels themselves need a GPU; here the arena is filled by hand.)"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fruitnerf_amd.data import synthetic_apple as sa
    from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
    from fruitnerf_amd.data.semantics import apple_metadata
    from fruitnerf_amd.params import ParamArena
    from fruitnerf_amd.training import start_gradient_sync, sync_gradients
    from tests import util
    cfg = FruitNerfModelConfig(log2_hashmap_size=6)
    cfg.proposal_net_args_list = util.small_config(prop_log2=5).proposal_net_args_list
    torch.manual_seed(0)  # same seed on every rank == DDP's rank-0 broadcast
    m = FruitModel(cfg, apple_metadata(), num_train_data=4, device="cpu")
    arena = ParamArena([("proposal_networks", list(m.proposal_networks.parameters())),
                        ("fields", list(m.field.parameters()))], "cpu")
    # identical parameters everywhere
    ref = arena.params.clone()
    dist.broadcast(ref, src=0)
    same_init = bool(torch.equal(ref, arena.params))
    # rank-specific gradients: rank r contributes (r + 1) * pattern
    pattern = torch.arange(arena.numel, dtype=torch.float32) % 7
    arena.grads.copy_((rank + 1) * pattern)
    scale = sync_gradients(arena, world)
    mean_ok = bool(torch.allclose(arena.grads * scale, pattern * (sum(range(1, world + 1)) / world)))
    # the bucketed, asynchronous exchange the fused training step uses: same sums, bucket by bucket
    arena.grads.copy_((rank + 1) * pattern)
    pending = []
    for name in ("fields", "proposal_networks"):
        pending += start_gradient_sync(arena, arena.group_ranges[name], world, bucket_elems=1000)
    covered = 0
    for a, b, work in pending:
        work.wait()
        covered += b - a
    bucketed_ok = bool(covered == arena.numel and len(pending) > 2 and
                       torch.allclose(arena.grads * scale, pattern * (sum(range(1, world + 1)) / world)))
    # every parameter's .grad is a view of the all-reduced arena at its recorded offset (what Adam consumes)
    views_ok = all(p.grad.data_ptr() == arena.grads.data_ptr() + 4 * off and
                   torch.equal(p.grad.reshape(-1), arena.grads[off:off + n]) for _, p, off, n in arena.entries)
    # rank-specific rays (seed + rank), same dataset
    scene = sa.make_scene(seed=0)
    c2w = sa.make_cameras(4, seed=0)
    data = sa.render_dataset(scene, c2w, H=16, W=16, fx=22.0, fy=22.0)
    from oracle import pixel_sampler as ops          # (CPU ranks: the oracle's sampler; the product samples on the device)
    u = torch.rand(32, 3, generator=torch.Generator().manual_seed(1234 + rank))
    o, d, cam, batch = ops.sample_pixels(data, torch.arange(4), u)
    gathered = [torch.zeros_like(d) for _ in range(world)]
    dist.all_gather(gathered, d)
    different_rays = not torch.equal(gathered[0], gathered[1])
    q.put((rank, same_init, mean_ok and bucketed_ok, views_ok, different_rays, scale))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_contract_world2_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same_init, mean_ok, views_ok, different_rays, scale in results:
        assert same_init, f"rank {rank}: initial weights differ across ranks"
        assert mean_ok, f"rank {rank}: all-reduced gradient mean is wrong"
        assert views_ok, f"rank {rank}: parameter .grad tensors are not views of the gradient arena"
        assert different_rays, "ranks drew identical rays"
        assert scale == 0.5


def _torch_adam(params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, grad_scale=1.0, zero_grad=True,
                weight_decay=0.0):
    """Stand-in for the HIP Adam kernel (elementwise, deterministic): what matters here is WHICH elements take a step
    with WHICH gradient, not the kernel's arithmetic (tests/test_gpu_training_parity.py::test_adam_matches_torch)."""
    g = grads * grad_scale
    exp_avg.mul_(beta1).add_(g, alpha=1 - beta1)
    exp_avg_sq.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    params.addcdiv_(exp_avg / bc1, (exp_avg_sq / bc2).sqrt() + eps, value=-lr)
    if zero_grad:
        grads.zero_()


def _sharded_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fruitnerf_amd import _kernels as K
    from fruitnerf_amd import training as T
    from fruitnerf_amd.params import ParamArena
    K.adam_step = _torch_adam
    torch.manual_seed(0)
    # a "proposal_networks" group and a "fields" group whose length is NOT a multiple of world * SHARD_ALIGN (a tail)
    p_prop = [torch.nn.Parameter(torch.randn(300))]
    p_field = [torch.nn.Parameter(torch.randn(5000)), torch.nn.Parameter(torch.randn(3 * T.SHARD_ALIGN + 77))]
    arena = ParamArena([("proposal_networks", p_prop), ("fields", p_field)], "cpu")

    class Model:
        def arena(self):
            return arena

    span = tuple(arena.group_ranges["fields"])
    pattern = (torch.arange(arena.numel, dtype=torch.float32) % 13) - 6.0
    start = arena.params.clone()
    results = {}
    for sharded in (False, True):
        arena.params.copy_(start)
        arena.grads.copy_((rank + 1) * pattern)
        opt = T.FusedAdam(Model())
        for step in range(3):          # three steps: moments of the own shard must carry over
            lrs = opt.begin_step(skip=("proposal_networks",))
            pend = (T.start_sharded_gradient_sync(arena, span, world) if sharded
                    else T.start_gradient_sync(arena, span, world, bucket_elems=2048))
            for e in pend:
                T.finish_exchange_entry(opt, e, lrs["fields"], 1.0 / world, "fields")
            grads_clean = bool((arena.grads[span[0]:span[1]] == 0).all())
            arena.grads.copy_((rank + 1) * pattern * (step + 2))
        results[sharded] = (arena.params.clone(), opt.exp_avg.clone(), grads_clean, pend)
    pa, ma, clean_a, _ = results[False]
    pb, mb, clean_b, pend_b = results[True]
    sh = pend_b[0]
    gathered = [torch.zeros_like(pb) for _ in range(world)]
    dist.all_gather(gathered, pb)
    q.put((rank,
           bool(torch.equal(pa, pb)),                                   # same parameters as the all-reduce path, bit for bit
           bool(all(torch.equal(g, pb) for g in gathered)),             # identical on every rank
           clean_a and clean_b,                                         # the whole span's gradient is left zeroed
           bool(torch.equal(ma[sh.my_a:sh.my_b], mb[sh.my_a:sh.my_b])   # the own shard's (and the tail's) moments are current
                and torch.equal(ma[sh.main_b:sh.b], mb[sh.main_b:sh.b])),
           (sh.my_b - sh.my_a) % T.SHARD_ALIGN == 0 and sh.main_b < sh.b and sh.my_a == sh.a + rank * (sh.my_b - sh.my_a),
           bool(torch.equal(pb[:span[0]], start[:span[0]]))))           # the other group is untouched
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_field_optimizer_world2_gloo():
    """training.SHARDED_FIELD_OPTIMIZER's pieces on two gloo ranks: reduce-scatter in place, the optimiser step of the own
    shard, zeroing of the rest, all-gather of the parameters in place — against the all-reduce path on the same inputs."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same_as_allreduce, same_everywhere, grads_zeroed, moments_ok, layout_ok, other_group_ok in results:
        assert same_as_allreduce, f"rank {rank}: sharded step differs from the all-reduce step"
        assert same_everywhere, f"rank {rank}: parameters differ between the ranks after the all-gather"
        assert grads_zeroed, f"rank {rank}: gradient span not left zeroed"
        assert moments_ok, f"rank {rank}: moments of the own shard differ from the all-reduce path's"
        assert layout_ok, f"rank {rank}: shard layout"
        assert other_group_ok, f"rank {rank}: the proposal networks' parameters moved"
