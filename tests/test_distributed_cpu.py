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
    b = sa.PixelBatcher(data, torch.arange(4), seed=1234 + rank)
    o, d, cam, batch = b.sample(32)
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
