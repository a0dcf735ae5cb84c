"""RCCL on a 1-GPU box: a one-rank "nccl" process group runs the complete multi-GPU exchange code path of
fused_train_iteration (bucketed async all-reduce on RCCL's stream, per-bucket wait + Adam, camera pose all-reduce).
An all-reduce over one rank is the identity and 1/world = 1, so K steps must leave the parameters where the same K
steps without the exchange leave them, up to the run-to-run noise of the float atomics in the dW reductions (measured
here by running the local path twice); the timing difference is the exchange's fixed cost (launches + handshakes).
  python tools/microbench/rccl_single_rank.py"""
import os, sys, time, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fruitnerf_amd.training as T
from fruitnerf_amd.data import synthetic_apple as sa
from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
from fruitnerf_amd.data.semantics import apple_metadata
from fruitnerf_amd.rays import RayBundle
from fruitnerf_amd.cameras.camera_optimizers import CameraAdam, CameraOptimizerConfig

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
HW, n_train, K, KT, R = 128, 20, 8, 40, 4096
focal = 1111.0 * HW / 800
scene = sa.make_scene(seed=0, device=dev); c2w = sa.make_cameras(n_train, seed=0, device=dev)
data = sa.render_dataset(scene, c2w, H=HW, W=HW, fx=focal, fy=focal)


def run(min_world):
    T.EXCHANGE_MIN_WORLD = min_world
    batcher = sa.PixelBatcher(data, torch.arange(n_train, device=dev), seed=100)
    torch.manual_seed(0)
    model = FruitModel(FruitNerfModelConfig(), apple_metadata(), num_train_data=n_train, device=dev); model.train()
    opt = T.FusedAdam(model)
    co = CameraOptimizerConfig(mode="SO3xR3").setup(n_train, dev); camera = (co, CameraAdam(co), batcher)
    snap = None
    for step in range(K + KT):
        if step == K:
            torch.cuda.synchronize()
            snap = torch.cat([model.arena().params, co.pose_adjustment.data.view(-1)]).clone()
            t0 = time.perf_counter()
        o, d, cam, batch = batcher.sample(R, co)
        T.fused_train_iteration(model, opt, RayBundle(o, d, None, cam), batch, step, world_size=1, camera=camera)
    host_ms = (time.perf_counter() - t0) / KT * 1e3      # enqueue-only time: < wall means the GPU is the bound
    torch.cuda.synchronize()
    print(f"  [min_world {min_world}] host enqueue {host_ms:.3f} ms/step", flush=True)
    return snap, (time.perf_counter() - t0) / KT * 1e3


if len(sys.argv) > 1:      # "local" | "exchange": one mode only (for rocprofv3 --kernel-trace)
    _, ms = run(2 if sys.argv[1] == "local" else 1)
    print(f"{sys.argv[1]}: {ms:.3f} ms/step", flush=True)
    dist.destroy_process_group()
    sys.exit(0)
p_a, ms_a = run(2)
p_b, ms_b = run(2)
p_c, ms_c = run(1)
noise = float((p_a - p_b).abs().max())
dev_c = float((p_a - p_c).abs().max())
print(f"one-rank RCCL exchange path after {K} steps: max|local - local'| = {noise:.3e} (atomic-order noise), "
      f"max|local - exchange| = {dev_c:.3e} | finite {bool(torch.isfinite(p_c).all())} | ms/step local {ms_a:.3f} / "
      f"{ms_b:.3f} vs with exchange {ms_c:.3f}", flush=True)
# (Adam turns sign flips of near-zero gradients into lr-sized differences, so this is only a coarse check)
assert dev_c <= max(10 * noise, 1e-6), "the exchange path changes the update"
dist.destroy_process_group()
