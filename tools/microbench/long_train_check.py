"""Train N steps on the synthetic scene and report parameter / output health (non-finite values, ranges)."""
import sys, torch
sys.path.insert(0, '/root/repo')
import numpy as np
from fruitnerf_amd.data import synthetic_apple as sa
from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
from fruitnerf_amd.data.semantics import apple_metadata
from fruitnerf_amd.rays import RayBundle
from fruitnerf_amd.training import FusedAdam, fused_train_iteration
from fruitnerf_amd.cameras.camera_optimizers import CameraAdam, CameraOptimizerConfig
dev = torch.device('cuda:0')
HW = 200; focal = 1111.0 * HW / 800
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
scene = sa.make_scene(seed=0, device=dev); c2w = sa.make_cameras(100, seed=0, device=dev)
data = sa.render_dataset(scene, c2w, H=HW, W=HW, fx=focal, fy=focal)
batcher = sa.PixelBatcher(data, torch.arange(90, device=dev), seed=1)
torch.manual_seed(0)
model = FruitModel(FruitNerfModelConfig(), apple_metadata(), num_train_data=90, device=dev); model.train()
opt = FusedAdam(model)
CAM = len(sys.argv) > 2 and sys.argv[2] == "camera"
co = CameraOptimizerConfig(mode="SO3xR3").setup(90, dev) if CAM else None
camera = (co, CameraAdam(co), batcher) if CAM else None
def report(tag):
    torch.cuda.synchronize()
    a = model.arena()
    bad = int((~torch.isfinite(a.params)).sum())
    if co is not None:
        pa_ = co.pose_adjustment.data
        print(f"   camera: |t| max {float(pa_[:, :3].norm(dim=1).max()):.4f} |w| max {float(pa_[:, 3:].norm(dim=1).max()):.4f} finite {bool(torch.isfinite(pa_).all())}")
    print(f"[{tag}] non-finite params {bad}, max|p| {float(a.params.abs().max()):.3e}, max|m| {float(opt.exp_avg.abs().max()):.3e}, max v {float(opt.exp_avg_sq.max()):.3e}")
    for name, p in model.named_parameters():
        if not torch.isfinite(p).all() or float(p.abs().max()) > 1e3:
            print("   suspicious:", name, float(p.abs().max()), int((~torch.isfinite(p)).sum()))
    model.eval()
    with torch.no_grad():
        n = 4096
        g = torch.Generator(device=dev).manual_seed(3)
        y = torch.randint(0, HW, (n,), device=dev, generator=g); x = torch.randint(0, HW, (n,), device=dev, generator=g)
        ci = torch.full((n,), 95, device=dev)
        o, d = sa.pixel_rays(c2w, ci, y, x, focal, focal, HW / 2, HW / 2)
        out = model(RayBundle(o, d, None, None))
        tgt = data["images"][ci, y, x].float() / 255
        mse = torch.mean((out["rgb"] - tgt) ** 2)
        print(f"   eval: psnr {float(-10 * torch.log10(mse)):.2f} finite rgb {bool(torch.isfinite(out['rgb']).all())} acc mean {float(out['accumulation'].mean()):.3f} sem range [{float(out['semantics'].min()):.2f}, {float(out['semantics'].max()):.2f}]")
    model.train()
for step in range(STEPS):
    o, d, cam, batch = batcher.sample(4096, co)
    ld, md = fused_train_iteration(model, opt, RayBundle(o, d, None, cam), batch, step, camera=camera)
    if step in (200, 1000, 4000, 8000, 12000, 16000, 20000, 25000, STEPS - 1):
        print(step, {k: round(float(v), 5) for k, v in ld.items()}, {k: round(float(v), 4) for k, v in md.items()})
        report(step)
