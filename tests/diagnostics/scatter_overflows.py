"""Does the binned scatter ever overflow a bin's queue during training (fnr_debug_scatter_overflows)?  Overflowed records
reach the gradient table through global float atomics: their summation order, hence the last bits, depend on timing.
usage: scatter_overflows.py [method] [steps] [report every]"""
import sys, torch
sys.path.insert(0, ".")
import bench
from fruitnerf_amd import _lib as L
from fruitnerf_amd.data import synthetic_apple as sa

dev = torch.device("cuda", 0)
scene = sa.make_scene(seed=0, device=dev)
c2w = sa.make_cameras(bench.N_CAMERAS, seed=0, device=dev)
data = sa.render_dataset(scene, c2w, H=800, W=800, fx=1111.0, fy=1111.0)
i_train, _ = bench.split_indices(bench.N_CAMERAS, bench.TRAIN_SPLIT)
method = sys.argv[1] if len(sys.argv) > 1 else "fruit_nerf_big"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
every = int(sys.argv[3]) if len(sys.argv) > 3 else 250
r = bench.MethodRun(method, "bf16x3", "SO3xR3", dev, 0, 1, data, torch.as_tensor(i_train, device=dev), len(i_train))
L.scatter_overflows(reset=True)
steps_with = 0
for i in range(steps):
    r.one_step(want_metrics=False)
    if (i + 1) % every == 0:
        print(f"{method} steps {i + 1 - every:5d}..{i + 1:5d}: {L.scatter_overflows(reset=True)} overflowed records", flush=True)
