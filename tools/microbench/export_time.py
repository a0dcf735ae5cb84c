"""Volume export (N^3 lattice) timing: per pass wall time, kernel time per entry point (HIP events), allocator events
(reserved-bytes delta = hipMalloc/hipFree inside the pass) and Python GC runs.  Usage: export_time.py [N] [passes]"""
import collections, gc, sys, time, torch
sys.path.insert(0, '/root/repo')
from fruitnerf_amd import _lib as L
from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
from fruitnerf_amd.data.semantics import apple_metadata
from fruitnerf_amd.data.fruit_datamanager import ExportDataManager
from fruitnerf_amd.export.exporter_utils import sample_volume
from fruitnerf_amd.hostinfo import usable_cpus
torch.set_num_threads(usable_cpus())   # the container's CPU quota (see fruitnerf_amd/hostinfo.py)
dev = torch.device('cuda:0')
torch.manual_seed(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 8
m = FruitModel(FruitNerfModelConfig(), apple_metadata(), num_train_data=90, device=dev, test_mode="export"); m.eval()
class P: pass
pipe = P(); pipe.model = m; pipe.datamanager = ExportDataManager(dev, eval_num_rays_per_batch=32768)
m.setup_inference(True, N, deterministic=True)
gc_runs = []
gc.callbacks.append(lambda phase, info: gc_runs.append((phase, info["generation"])) if phase == "stop" else None)
for it in range(passes):
    n_rays = pipe.datamanager.setup_inference(aabb=((-1., -1., -1.), (1., 1., 1.)), num_points=N)
    L.profile_enable(True)
    st0 = torch.cuda.memory_stats(dev)
    gc_runs.clear()
    torch.cuda.synchronize(); t = time.perf_counter()
    sets = sample_volume(pipe, n_rays, transform_json={"scale": 1.0})
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    st1 = torch.cuda.memory_stats(dev)
    recs = L.profile_collect(); L.profile_enable(False)
    agg = collections.defaultdict(float)
    for op, u, ms in recs: agg[op] += ms
    print(f"pass {it}: {dt*1e3:.1f} ms = {n_rays*N/dt/1e6:.0f} M samples/s | kernels {sum(agg.values()):.2f} ms "
          f"{ {k: round(v, 2) for k, v in agg.items()} } | segments alloc'd {st1['num_device_alloc']-st0['num_device_alloc']} "
          f"freed {st1['num_device_free']-st0['num_device_free']} reserved {st1['reserved_bytes.all.current']/2**30:.2f} GiB "
          f"| gc {gc_runs}")
