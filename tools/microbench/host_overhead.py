"""Host enqueue time vs GPU time of the fused training step (is the Python launch loop the bottleneck?)."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
import numpy as np
from fruitnerf_amd.data import synthetic_apple as sa
from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
from fruitnerf_amd.data.semantics import apple_metadata
from fruitnerf_amd.rays import RayBundle
from fruitnerf_amd.training import FusedAdam, fused_train_iteration
from fruitnerf_amd.cameras.camera_optimizers import CameraAdam, CameraOptimizerConfig
dev = torch.device('cuda:0')
HW = 200
RAYS = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
scene = sa.make_scene(seed=0, device=dev); c2w = sa.make_cameras(100, seed=0, device=dev)
data = sa.render_dataset(scene, c2w, H=HW, W=HW, fx=1111.0 * HW / 800, fy=1111.0 * HW / 800)
ids = torch.arange(90, device=dev)
batcher = sa.PixelBatcher(data, ids, seed=1)
torch.manual_seed(0)
model = FruitModel(FruitNerfModelConfig(), apple_metadata(), num_train_data=90, device=dev); model.train()
opt = FusedAdam(model)
for cam_mode in ("off", "SO3xR3"):
    camera = None
    if cam_mode != "off":
        co = CameraOptimizerConfig(mode=cam_mode).setup(90, dev); camera = (co, CameraAdam(co), batcher)
    step = [0]
    def one():
        o, d, cam, batch = batcher.sample(RAYS, camera[0] if camera else None)
        fused_train_iteration(model, opt, RayBundle(o, d, None, cam), batch, step[0], camera=camera); step[0] += 1
    for _ in range(30): one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): one()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"camera {cam_mode}: host enqueue {t_enq / 200 * 1e3:.3f} ms/step, total {t_all / 200 * 1e3:.3f} ms/step")
