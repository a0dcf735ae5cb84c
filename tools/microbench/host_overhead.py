import os, sys, time, torch
sys.path.insert(0, '/root/repo')
from fruitnerf_amd.data import synthetic_apple as sa
from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
from fruitnerf_amd.rays import RayBundle
from fruitnerf_amd.training import FusedAdam, train_iteration
import cProfile, pstats
dev = torch.device('cuda:0')
scene = sa.make_scene(seed=0, device=dev); c2w = sa.make_cameras(20, seed=0, device=dev)
data = sa.render_dataset(scene, c2w, H=200, W=200, fx=277., fy=277.)
b = sa.PixelBatcher(data, torch.arange(18, device=dev), seed=1)
torch.manual_seed(0)
m = FruitModel(FruitNerfModelConfig(), num_train_data=18, device=dev); m.train(); opt = FusedAdam(m)
step = [0]
def one():
    o, d, cam, batch = b.sample(4096)
    train_iteration(m, opt, RayBundle(o, d, None, cam), batch, step[0]); step[0] += 1
for _ in range(20): one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): one()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/50:.3f} ms/step, total {1e3*(t2-t0)/50:.3f} ms/step")
t0 = time.perf_counter()
for _ in range(50): b.sample(4096)
torch.cuda.synchronize(); print(f"batcher only {1e3*(time.perf_counter()-t0)/50:.3f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(30): one()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
