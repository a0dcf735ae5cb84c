"""Where do the slow export passes spend their host time?  cProfile per pass, top entries of every slow pass."""
import cProfile, io, pstats, sys, time, torch
sys.path.insert(0, '/root/repo')
from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
from fruitnerf_amd.data.semantics import apple_metadata
from fruitnerf_amd.data.fruit_datamanager import ExportDataManager
from fruitnerf_amd.export.exporter_utils import sample_volume
dev = torch.device('cuda:0')
torch.manual_seed(0)
N = 256
m = FruitModel(FruitNerfModelConfig(), apple_metadata(), num_train_data=90, device=dev, test_mode="export"); m.eval()
class P: pass
pipe = P(); pipe.model = m; pipe.datamanager = ExportDataManager(dev, eval_num_rays_per_batch=32768)
m.setup_inference(True, N, deterministic=True)
for it in range(10):
    n_rays = pipe.datamanager.setup_inference(aabb=((-1., -1., -1.), (1., 1., 1.)), num_points=N)
    pr = cProfile.Profile()
    torch.cuda.synchronize(); t = time.perf_counter()
    pr.enable()
    sets = sample_volume(pipe, n_rays, transform_json={"scale": 1.0})
    torch.cuda.synchronize()
    pr.disable()
    dt = time.perf_counter() - t
    print(f"pass {it}: {dt*1e3:.1f} ms")
    if dt > 0.03:
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(8)
        print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:2500])
