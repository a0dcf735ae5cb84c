"""Where does the host's time per training step go?  cProfile over N steps of bench.py's loop (TrainingSteps), without
waiting for the GPU inside the window (the queue is drained before and after): top functions by own time and by
cumulative time, and the wall time per step of the enqueue loop.   usage: python tools/host_profile.py [steps] [exchange]"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fruitnerf_amd.data import synthetic_apple as sa  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
exchange = len(sys.argv) > 2 and sys.argv[2] == "exchange"
dev = torch.device("cuda", 0)
if exchange:
    import torch.distributed as dist
    import fruitnerf_amd.training as T
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29655", rank=0, world_size=1, device_id=dev)
    T.EXCHANGE_MIN_WORLD, T.DEFER_FIELD_UPDATE = 1, True
scene = sa.make_scene(seed=0, device=dev)
c2w = sa.make_cameras(bench.N_CAMERAS, seed=0, device=dev)
data = sa.render_dataset(scene, c2w, H=800, W=800, fx=1111.0, fy=1111.0)
i_train, _ = bench.split_indices(bench.N_CAMERAS, bench.TRAIN_SPLIT)
run = bench.MethodRun("fruit_nerf", "bf16x3", "SO3xR3", dev, 0, 1, data, torch.as_tensor(i_train, device=dev), len(i_train))
for _ in range(40):
    run.one_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    run.one_step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"{'exchange path' if exchange else 'single process'}: host enqueue {t_enq / steps * 1e3:.3f} ms/step, "
      f"with the GPU {t_all / steps * 1e3:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    run.one_step()
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats(key).print_stats(28)
    print("\n".join(l for l in out.getvalue().splitlines() if l.strip())[:6000])
