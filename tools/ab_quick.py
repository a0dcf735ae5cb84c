"""Same-process A/B of the training step on one box: ONE scene, ONE model, alternating windows of the headline loop
(bench.MethodRun + TrainingSteps: pixel sampling, fwd + bwd + optimiser, camera optimiser, two streams) with a knob set to
A or B — module variables of fruitnerf_amd.training or environment variables the library reads per call.  The update
schedule of the proposal networks drifts slowly (every 2nd step until step 1000), so consecutive windows see the same mix of
step shapes; window pairs are repeated and the per-arm MEDIAN is reported, plus the host enqueue time per step.
Nothing here checks results: every knob is bit-identical by construction and covered by tests/test_gpu_determinism.py.

usage: python tools/ab_quick.py [--method fruit_nerf] [--steps 200] [--pairs 3] knob=A,B [knob=A,B ...]
  knob = T.<NAME>        a module variable of fruitnerf_amd.training (values: 0 / 1 / int)
  knob = env.<NAME>      an environment variable (value `-` = unset)
  several knobs switch together (arm A = every first value, arm B = every second)
  no knob: `pairs` windows of the default configuration (a library variant is selected with FNR_LIB_PATH outside)."""
import argparse
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import fruitnerf_amd.training as T  # noqa: E402
from fruitnerf_amd.data import synthetic_apple as sa  # noqa: E402


def set_knob(name, value):
    if name.startswith("T."):
        setattr(T, name[2:], type(getattr(T, name[2:]))(int(value)))
    elif name.startswith("env."):
        if value == "-":
            os.environ.pop(name[4:], None)
        else:
            os.environ[name[4:]] = value
    else:
        raise SystemExit(f"unknown knob {name}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--method", default="fruit_nerf")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--pairs", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("knobs", nargs="*")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    HW, focal = 800, 1111.0
    scene = sa.make_scene(seed=0, device=dev)
    c2w = sa.make_cameras(bench.N_CAMERAS, seed=0, device=dev)
    data = sa.render_dataset(scene, c2w, H=HW, W=HW, fx=focal, fy=focal)
    i_train, _ = bench.split_indices(bench.N_CAMERAS, bench.TRAIN_SPLIT)
    run = bench.MethodRun(args.method, "bf16x3", "SO3xR3", dev, 0, 1, data, torch.as_tensor(i_train, device=dev), len(i_train))
    knobs = [(k.split("=")[0], k.split("=")[1].split(",")) for k in args.knobs]
    arms = ("A", "B") if knobs else ("A",)
    for name, vals in knobs:
        set_knob(name, vals[0])
    for _ in range(args.warmup):
        run.one_step()
    torch.cuda.synchronize()
    res = {a: [] for a in arms}
    host = {a: [] for a in arms}
    for pair in range(args.pairs):
        for ai, arm in enumerate(arms):
            for name, vals in knobs:
                set_knob(name, vals[ai])
            for _ in range(4):                      # let the knob's buffers / streams settle outside the window
                run.one_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                run.one_step()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            res[arm].append((t2 - t0) / args.steps * 1e3)
            host[arm].append((t1 - t0) / args.steps * 1e3)
    label = " ".join(args.knobs) or f"default (lib {os.environ.get('FNR_LIB_PATH', 'in-tree')})"
    for arm in arms:
        ms = statistics.median(res[arm])
        print(f"{label} | arm {arm}: median {ms:.4f} ms/step = {run.rays / ms * 1e3 / 1e6:.3f} M rays/s  "
              f"windows {[round(x, 4) for x in res[arm]]}  host enqueue {statistics.median(host[arm]):.3f} ms/step "
              f"(steps {run.step_idx})")
    if len(arms) == 2:
        a, b = statistics.median(res["A"]), statistics.median(res["B"])
        print(f"{label} | B vs A: {(a / b - 1) * 100:+.2f} % rays/s")


if __name__ == "__main__":
    main()
