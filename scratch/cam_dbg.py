import sys, torch
sys.path.insert(0, '/root/repo')
from oracle import camera_opt as oc
from fruitnerf_amd import _kernels as K
from fruitnerf_amd.cameras.camera_optimizers import CameraOptimizerConfig
from fruitnerf_amd.data import synthetic_apple as sa
dev = torch.device('cuda:0')
n = 8
g = torch.Generator().manual_seed(3)
pose0 = torch.cat([torch.randn(n, 3, generator=g) * 0.02, torch.randn(n, 3, generator=g) * 0.03], dim=1)
c2w = sa.make_cameras(n, seed=0)
hcam = CameraOptimizerConfig(mode="SO3xR3").setup(n, dev)
with torch.no_grad(): hcam.pose_adjustment.copy_(pose0.to(dev))
delta = hcam(torch.arange(n, device=dev)).cpu()
ref = oc.exp_map_SO3xR3(pose0)
print("delta R err", (delta[:, :, :3] - ref[:, :, :3]).abs().max().item(), "t err", (delta[:, :, 3] - ref[:, :, 3]).abs().max().item())
print(delta[1], ref[1])
data = sa.render_dataset(sa.make_scene(seed=0), c2w, H=16, W=16, fx=20.0, fy=20.0)
ddev = {kk: (v.to(dev) if torch.is_tensor(v) else v) for kk, v in data.items()}
iset = K.ImageSetArg(ddev["images"], ddev["masks"], ddev["c2w"], 20.0, 20.0, 8.0, 8.0)
adj = hcam.adjusted_cameras(iset, torch.arange(n, device=dev)).cpu()
ref2 = oc.multiply(c2w, ref)
print("adj err", (adj - ref2).abs().max().item())
print((adj - ref2)[1])
print("c2w[1]", c2w[1]); print("R1^T R1", c2w[1][:, :3].T @ c2w[1][:, :3])
