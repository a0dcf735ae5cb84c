"""cProfile of the host side of the fused training step (top cumulative entries)."""
import cProfile, pstats, sys, torch
sys.path.insert(0, '/root/repo')
from fruitnerf_amd.data import synthetic_apple as sa
from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
from fruitnerf_amd.data.semantics import apple_metadata
from fruitnerf_amd.rays import RayBundle
from fruitnerf_amd.training import FusedAdam, fused_train_iteration
from fruitnerf_amd.cameras.camera_optimizers import CameraAdam, CameraOptimizerConfig
dev = torch.device('cuda:0')
HW = 200
scene = sa.make_scene(seed=0, device=dev); c2w = sa.make_cameras(100, seed=0, device=dev)
data = sa.render_dataset(scene, c2w, H=HW, W=HW, fx=1111.0 * HW / 800, fy=1111.0 * HW / 800)
batcher = sa.PixelBatcher(data, torch.arange(90, device=dev), seed=1)
torch.manual_seed(0)
model = FruitModel(FruitNerfModelConfig(), apple_metadata(), num_train_data=90, device=dev); model.train()
opt = FusedAdam(model)
co = CameraOptimizerConfig(mode="SO3xR3").setup(90, dev); camera = (co, CameraAdam(co), batcher)
step = [0]
def one():
    o, d, cam, batch = batcher.sample(4096, camera[0])
    fused_train_iteration(model, opt, RayBundle(o, d, None, cam), batch, step[0], camera=camera); step[0] += 1
for _ in range(30): one()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): one()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
