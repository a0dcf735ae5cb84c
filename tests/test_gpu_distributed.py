"""The multi-rank training path on the one GPU a test box has: a ONE-rank RCCL process group with the gradient exchange
forced on (training.EXCHANGE_MIN_WORLD = 1).  The all-reduces move no bytes, but everything else is the N > 1 code:
bucketed asynchronous collectives on RCCL's stream issued per scatter level group, per-bucket waits, the separate
optimiser launches, the pose-gradient exchange.  (Two ranks: tests/test_distributed_cpu.py over gloo.)"""
import json
import os
import subprocess
import sys

import pytest
import torch

from . import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def test_bench_runs_over_a_one_rank_rccl_group(dev):
    env = dict(os.environ, FNR_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29631",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2",
                          "--no-cpu-baseline", "--no-quality"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["value"] > 0
    assert d["config"]["table_optimizer"].startswith("separate")
    assert d["config"]["rccl_ranks"] == {"world_size": 1, "backend": "nccl", "devices": 1}   # what the group itself reports


def test_bench_gpus_2_starts_two_ranks(dev):
    """`python bench.py --gpus 2` with no launcher around it must start the two ranks itself (torch.distributed.run on
    127.0.0.1) and report them.  On the one-GPU test box the ranks share device 0 over gloo (RCCL refuses two ranks on
    one device): FNR_BENCH_BACKEND=gloo FNR_BENCH_ONE_DEVICE=1 — the control flow, exchange path and timing protocol
    are the N > 1 code."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(FNR_BENCH_BACKEND="gloo", FNR_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                          "--no-cpu-baseline", "--no-quality"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0)"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["steps"] == 4 and d["value"] > 0
    assert d["config"]["table_optimizer"].startswith("separate")
    assert d["config"]["rccl_ranks"] == {"world_size": 2, "backend": "gloo", "devices": 1}
    # WHICH entry point a 4-step window with two ranks interleaving on one device measures as the slowest is timing, not
    # function: only the shape of the roofline object is asserted here
    sys.path.insert(0, ROOT)
    import bench
    assert d["roofline"]["bound"] in ("hbm", "mfma") and d["roofline"]["kernel"] in bench.ROOFLINE_OPS


def test_two_ranks_hold_identical_parameters_after_fused_steps(dev):
    """DDP's contract on the exchange path with REAL kernels and rank-specific rays: two gloo ranks sharing device 0 (RCCL
    refuses two ranks on one device) take 12 fused training steps — field / proposal / pose gradients averaged by the
    bucketed exchange, the sharded optimiser step (default) — and must end with bit-identical parameters and poses
    (tools/microbench/ddp_consistency.py is the worker; it asserts on rank 0)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(FNR_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", GPU_MAX_HW_QUEUES="8")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tools", "microbench", "ddp_consistency.py")],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    assert "ranks hold identical parameters: True | finite: True" in out.stdout, out.stdout[-2000:]


def test_exchange_path_step_is_the_single_process_step(dev):
    """Training steps (camera optimiser included) from identical states: the exchange path — scatter in level
    groups with an all-reduce per group, separate optimiser launches per bucket, pose gradient all-reduced — against
    the single-process path (optimiser steps fused into the scatter / pose-gradient kernels).  After the first step the
    hash table, its moments and the poses are bit-identical (fixed-point scatter sums, same optimiser arithmetic); the
    MLP weights meet in float atomics and agree to 1e-6.  A second step stays together on average."""
    import torch.distributed as dist
    import fruitnerf_amd.training as T
    from fruitnerf_amd import _kernels as K
    from fruitnerf_amd.cameras.camera_optimizers import CameraAdam, CameraOptimizerConfig
    from fruitnerf_amd.data import synthetic_apple as sa
    from fruitnerf_amd.rays import RayBundle
    n_cam, HW, focal, R = 8, 64, 90.0, 256
    cfg = util.small_config(log2=15, prop_log2=13)
    om = util.make_oracle(cfg, num_images=n_cam, seed=21)
    scene = sa.make_scene(seed=0)
    c2w = sa.make_cameras(n_cam, seed=0)
    data = {k: (v.to(dev) if torch.is_tensor(v) else v)
            for k, v in sa.render_dataset(scene, c2w, H=HW, W=HW, fx=focal, fy=focal).items()}
    g = torch.Generator().manual_seed(1)
    us = [torch.rand(R, 3, generator=g).to(dev) for _ in range(2)]
    jits = [[torch.rand(R, 1, generator=g).to(dev) for _ in range(3)] for _ in range(2)]

    def run(world_arg):
        hm = util.make_hip_like(om, dev)
        hm.train()
        opt = T.FusedAdam(hm)
        cam = CameraOptimizerConfig(mode="SO3xR3").setup(n_cam, dev)
        cadam = CameraAdam(cam)
        batcher = sa.PixelBatcher(data, torch.arange(n_cam, device=dev), seed=0)
        batcher._set = K.ImageSetArg(data["images"], data["masks"], data["c2w"], focal, focal, HW / 2.0, HW / 2.0)
        snaps = []
        for step in range(2):
            c2w_adj = cam.adjusted_cameras(batcher._set, batcher.image_ids)
            o, d, ci, image, mask = K.sample_pixels(batcher._set, batcher.image_ids, us[step], c2w_adj)
            batcher.last_draw = {"u": us[step], "cam": ci, "c2w_adjusted": c2w_adj}
            T.fused_train_iteration(hm, opt, RayBundle(o, d, None, ci[:, None]), {"image": image, "fruit_mask": mask[:, None]},
                                    step, world_size=world_arg, jitter=jits[step], camera=(cam, cadam, batcher))
            if step == 1:
                hm.field.flush_deferred_update()     # a pending deferred field update (step 0's is applied by step 1's forward)
            torch.cuda.synchronize()
            snaps.append((hm.arena().params.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(),
                          cam.pose_adjustment.data.clone()))
        table = hm.field.mlp_base_grid.hash_table
        a, n = [(off, k) for _, p, off, k in hm.arena().entries if p is table][0]
        return snaps, (a, a + n)

    single = run(1)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29641", rank=0, world_size=1, device_id=dev)
    old = T.EXCHANGE_MIN_WORLD
    T.EXCHANGE_MIN_WORLD = 1
    saved_sharded, T.SHARDED_FIELD_OPTIMIZER = T.SHARDED_FIELD_OPTIMIZER, False     # (the all-reduce path first)
    assert saved_sharded, "the sharded optimiser step is the default (round 6)"
    try:
        exch = run(1)
        # the same steps with the field's wait + optimiser step deferred to the next step's encode (and flushed by the
        # parameter read at the end), and with the scatter / collectives in four level groups: bit-identical states
        T.DEFER_FIELD_UPDATE = True
        deferred = run(1)
        T.DEFER_FIELD_UPDATE = False
        old_groups, T.EXCHANGE_LEVEL_GROUPS = T.EXCHANGE_LEVEL_GROUPS, 4
        grouped = run(1)
        T.EXCHANGE_LEVEL_GROUPS = old_groups
        # the sharded optimiser step (reduce-scatter, own shard's step, all-gather of the parameters; the default since
        # round 6): on one rank the shard is the whole span but for its unaligned tail
        T.SHARDED_FIELD_OPTIMIZER = True
        sharded = run(1)
    finally:
        T.SHARDED_FIELD_OPTIMIZER = saved_sharded
        T.DEFER_FIELD_UPDATE = False
        T.EXCHANGE_MIN_WORLD = old
        if created:
            dist.destroy_process_group()
    for x, y in zip(exch[0][1], deferred[0][1]):            # after the second step (the first one's field update was
        assert torch.equal(x, y)                            # applied inside the second step's forward pass)
    for snap_a, snap_b in zip(exch[0], grouped[0]):
        for x, y in zip(snap_a, snap_b):
            assert torch.equal(x, y)
    for snap_a, snap_b in zip(exch[0], sharded[0]):
        for x, y in zip(snap_a, snap_b):
            assert torch.equal(x, y), "sharded optimiser step differs from the all-reduce path on one rank"
    (s1, s2), (a, b) = single
    (e1, e2), _ = exch
    for x, y in zip(s1[:3], e1[:3]):                     # after the first step
        assert torch.equal(x[a:b], y[a:b])
        assert float((x - y).abs().max()) <= 1e-6 * max(1.0, float(x.abs().max()))
    assert torch.equal(s1[3], e1[3])
    assert int((s1[0][a:b] != util.make_hip_like(om, dev).arena().params[a:b]).sum()) > 0   # the table did move
    # second step: its inputs (the MLP weights) already differ in the last bits between ANY two runs
    for x, y in zip(s2, e2):
        assert float((x - y).abs().mean()) <= 1e-4 * float(x.abs().mean()) + 1e-9


def test_exchange_path_keeps_the_two_stream_schedule(dev):
    """Round 4: with a gradient exchange the second stream's segment carries what the single process has there — the
    ray-gradient reduction, the pose gradient and its all-reduce, the proposal networks' all-reduce, their waits and
    optimiser steps, and the look-ahead — and is enqueued AHEAD of the table scatter so that the small collectives precede
    the field's 67 MB on the communicator's stream.  TrainingSteps over a one-rank RCCL group against the single process:
    same states after 8 steps (update and non-update steps), every step but the first ran on a look-ahead, and the order
    of the all-reduce calls is small groups first."""
    import torch.distributed as dist
    import fruitnerf_amd.training as T
    from fruitnerf_amd.cameras.camera_optimizers import CameraAdam, CameraOptimizerConfig
    from fruitnerf_amd.data import synthetic_apple as sa
    n_cam, HW, focal, R, steps = 8, 64, 90.0, 512, 14
    cfg = util.small_config(log2=15, prop_log2=13)
    om = util.make_oracle(cfg, num_images=n_cam, seed=21)
    scene = sa.make_scene(seed=0)
    c2w = sa.make_cameras(n_cam, seed=0)
    data = {k: (v.to(dev) if torch.is_tensor(v) else v)
            for k, v in sa.render_dataset(scene, c2w, H=HW, W=HW, fx=focal, fy=focal).items()}

    def run(world_arg, log=None):
        hm = util.make_hip_like(om, dev)
        hm.train()
        opt = T.FusedAdam(hm)
        cam = CameraOptimizerConfig(mode="SO3xR3").setup(n_cam, dev)
        batcher = sa.PixelBatcher(data, torch.arange(n_cam, device=dev), seed=3)
        loop = T.TrainingSteps(hm, opt, batcher, R, camera=(cam, CameraAdam(cam)), world_size=world_arg)
        for step in range(steps):
            if log is not None:
                log.append(("step", step, bool(hm.proposal_sampler.updated_now())))
            loop.step()
        hm.field.flush_deferred_update()
        torch.cuda.synchronize()
        table = hm.field.mlp_base_grid.hash_table
        a, n = [(off, k) for _, p, off, k in hm.arena().entries if p is table][0]
        return (hm.arena().params.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), cam.pose_adjustment.data.clone(),
                hm.__dict__.get("_ahead_used", 0), (a, a + n))

    single = run(1)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29643", rank=0, world_size=1, device_id=dev)
    old, old_defer, old_sharded = T.EXCHANGE_MIN_WORLD, T.DEFER_FIELD_UPDATE, T.SHARDED_FIELD_OPTIMIZER
    T.EXCHANGE_MIN_WORLD, T.DEFER_FIELD_UPDATE = 1, True
    T.SHARDED_FIELD_OPTIMIZER = False    # the ORDER of the all-reduces is what is logged here (sharded: a reduce-scatter in the field's place)
    log = []
    real = dist.all_reduce

    def spy(tensor, *a, **k):
        log.append(("all_reduce", int(tensor.numel())))
        return real(tensor, *a, **k)

    dist.all_reduce = spy
    try:
        exch = run(1, log)
    finally:
        dist.all_reduce = real
        T.EXCHANGE_MIN_WORLD, T.DEFER_FIELD_UPDATE = old, old_defer
        if created:
            dist.destroy_process_group()
    assert single[4] == exch[4] == steps - 1, "every step but the first runs on what the previous one sampled ahead"
    a, b = single[5]
    for name, x, y in zip(("parameters", "exp_avg", "exp_avg_sq"), single[:3], exch[:3]):
        assert torch.equal(x[a:b], y[a:b]), f"hash table {name}"
        assert float((x - y).abs().max()) <= 1e-6 * max(1.0, float(x.abs().max())), name
    assert torch.equal(single[3], exch[3]), "camera poses"
    # per step: [proposal networks (update steps only)], poses (n_cam x 6), then the field group — the largest — last
    i = 0
    seen_update = seen_plain = False
    while i < len(log):
        assert log[i][0] == "step"
        updated = log[i][2]
        calls = []
        i += 1
        while i < len(log) and log[i][0] == "all_reduce":
            calls.append(log[i][1])
            i += 1
        assert len(calls) == (3 if updated else 2), (updated, calls)
        assert calls[-1] == max(calls) and calls[-2] == n_cam * 6, calls
        seen_update, seen_plain = seen_update or updated, seen_plain or not updated
    assert seen_update and seen_plain
