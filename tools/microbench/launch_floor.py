"""What a tiny dependent kernel costs inside a busy stream: a 64 MB Adam launch (the 'big' kernel) followed by k tiny
Adam launches (540 parameters), k = 0, 8, 32; the slope is the GPU-side cost of one more small launch in the step."""
import sys
import time

import torch

sys.path.insert(0, '/root/repo')
from fruitnerf_amd import _kernels as K   # noqa: E402

dev = torch.device('cuda:0')
big = [torch.zeros(4 << 20, device=dev) for _ in range(4)]
small = [torch.zeros(540, device=dev) for _ in range(4)]


def run(k, reps=200):
    for _ in range(20):
        K.adam_step(*big, 1e-3, 0.9, 0.999, 1e-8, 1)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c0 = time.perf_counter()
    t0.record()
    for _ in range(reps):
        K.adam_step(*big, 1e-3, 0.9, 0.999, 1e-8, 1)
        for _ in range(k):
            K.adam_step(*small, 1e-3, 0.9, 0.999, 1e-8, 1)
    t1.record()
    c1 = time.perf_counter()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps * 1e3, (c1 - c0) / reps * 1e6


base = None
for k in (0, 8, 32):
    gpu, cpu = run(k)
    if base is None:
        base = gpu
    print(f"{k:3d} tiny launches after a big one: {gpu:8.1f} us GPU per iteration, {cpu:8.1f} us CPU enqueue"
          + (f"  -> {(gpu - base) / k:5.2f} us per tiny launch" if k else ""))

# same thing replayed from a HIP graph (no CPU in the loop)
for k in (0, 8, 32):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(10):
                K.adam_step(*big, 1e-3, 0.9, 0.999, 1e-8, 1)
                for _ in range(k):
                    K.adam_step(*small, 1e-3, 0.9, 0.999, 1e-8, 1)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        g.replay()
    t1.record()
    torch.cuda.synchronize()
    gpu = t0.elapsed_time(t1) / 200 * 1e3
    if k == 0:
        gbase = gpu
    print(f"graph, {k:3d} tiny launches: {gpu:8.1f} us per iteration" + (f"  -> {(gpu - gbase) / k:5.2f} us per tiny launch" if k else ""))
