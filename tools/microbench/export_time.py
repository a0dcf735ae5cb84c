"""Volume export (256^3 lattice) timing with the per-entry-point breakdown."""
import sys, time, torch, collections
sys.path.insert(0, '/root/repo')
from fruitnerf_amd import _lib as L
from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
from fruitnerf_amd.data.fruit_datamanager import ExportDataManager
from fruitnerf_amd.export.exporter_utils import sample_volume
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = FruitModel(FruitNerfModelConfig(), num_train_data=90, device=dev, test_mode="export"); m.eval()
N = 256
class P: pass
pipe = P(); pipe.model = m; pipe.datamanager = ExportDataManager(dev, eval_num_rays_per_batch=32768)
m.setup_inference(True, N)
for it in range(3):
    n_rays = pipe.datamanager.setup_inference(aabb=((-1., -1., -1.), (1., 1., 1.)), num_points=N)
    if it == 2: L.profile_enable(True)
    torch.cuda.synchronize(); t = time.perf_counter()
    sets = sample_volume(pipe, n_rays, transform_json={"scale": 1.0})
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"pass {it}: {dt*1e3:.1f} ms = {n_rays*N/dt/1e6:.0f} M samples/s")
recs = L.profile_collect(); L.profile_enable(False)
agg = collections.defaultdict(float)
for op, u, ms in recs: agg[op] += ms
print({k: round(v, 2) for k, v in agg.items()})
