"""Where a slow 256^3 export pass spends its extra host time: wall-clock marks inside sample_volume's lattice path
(enqueue of the batches | the one counts wait | the six device-to-host copies | host post-processing)."""
import ctypes, os, sys, time
if os.environ.get("FNR_SPIN") == "1":    # hipDeviceScheduleSpin before the context exists: host waits busy-poll
    print("hipSetDeviceFlags(spin) ->", ctypes.CDLL("libamdhip64.so").hipSetDeviceFlags(1))
import torch
sys.path.insert(0, '/root/repo')
from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
from fruitnerf_amd.data.semantics import apple_metadata
from fruitnerf_amd.data.fruit_datamanager import ExportDataManager
from fruitnerf_amd.export import exporter_utils as E
from fruitnerf_amd.hostinfo import usable_cpus
if os.environ.get('FNR_THREADS_BY_QUOTA', '1') == '1':
    torch.set_num_threads(usable_cpus())
dev = torch.device('cuda:0')
torch.manual_seed(0)
N = 256
m = FruitModel(FruitNerfModelConfig(), apple_metadata(), num_train_data=90, device=dev, test_mode="export"); m.eval()
class P: pass
pipe = P(); pipe.model = m; pipe.datamanager = ExportDataManager(dev, eval_num_rays_per_batch=32768)
m.setup_inference(True, N, deterministic=True)
marks = []
side = torch.cuda.Stream()
orig_read, orig_compact = E._read_counts, E.K.export_compact
def read(counts, state):
    marks.append(("enqueued", time.perf_counter()))
    r = orig_read(counts, state)
    marks.append(("counts", time.perf_counter()))
    return r
E._read_counts = read
for it in range(16):
    n_rays = pipe.datamanager.setup_inference(aabb=((-1., -1., -1.), (1., 1., 1.)), num_points=N)
    torch.cuda.synchronize()
    if os.environ.get("FNR_KEEP_BUSY") == "1":   # a spinning kernel on a side stream for the length of the pass
        with torch.cuda.stream(side):
            torch.cuda._sleep(int(2.0e9 * 0.012))
    marks.clear()
    t0 = time.perf_counter()
    sets = E.sample_volume(pipe, n_rays, transform_json={"scale": 1.0})
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    d = dict(marks)
    print(f"pass {it}: total {1e3*(t1-t0):6.1f} ms | enqueue {1e3*(d['enqueued']-t0):6.1f} | counts wait {1e3*(d['counts']-d['enqueued']):6.1f} | "
          f"copies + host {1e3*(t1-d['counts']):6.1f} | trailing sync {1e3*(t2-t1):5.1f}")
