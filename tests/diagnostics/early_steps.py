"""Per-step host enqueue time and, every 20 steps, the time of the 20-step window (no synchronize inside a window) of the
bench loop from step 0: host stalls (allocator growth, first-use paths) and the proposal-update schedule show up here.
argument `events`: per-entry-point GPU time of the 20-step windows instead."""
import sys, time, torch
sys.path.insert(0, ".")
import bench
from fruitnerf_amd import _lib as L
from fruitnerf_amd.data import synthetic_apple as sa

dev = torch.device("cuda", 0)
scene = sa.make_scene(seed=0, device=dev)
c2w = sa.make_cameras(bench.N_CAMERAS, seed=0, device=dev)
data = sa.render_dataset(scene, c2w, H=800, W=800, fx=1111.0, fy=1111.0)
i_train, _ = bench.split_indices(bench.N_CAMERAS, bench.TRAIN_SPLIT)
run = bench.MethodRun("fruit_nerf", "bf16x3", "SO3xR3", dev, 0, 1, data, torch.as_tensor(i_train, device=dev), len(i_train))
mode = sys.argv[1] if len(sys.argv) > 1 else "steps"
if mode == "steps":
    n_win = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    for w in range(n_win):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        host = []
        for i in range(20):
            h0 = time.perf_counter()
            run.one_step()
            host.append(1e3 * (time.perf_counter() - h0))
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        stats = torch.cuda.memory_stats(dev)
        print(f"steps {20 * w:4d}..{20 * w + 20:4d}: {dt / 20 * 1e3:.4f} ms/step, host enqueue {t_enq / 20 * 1e3:.3f} ms/step "
              f"(max {max(host):.2f} at +{host.index(max(host))}), hipMalloc calls so far {stats['num_device_alloc']}, "
              f"reserved {stats['reserved_bytes.all.current'] / 2**20:.0f} MiB", flush=True)
    sys.exit(0)
for _ in range(5):
    run.one_step()
torch.cuda.synchronize()
base = None
for w in range(12):
    L.profile_enable(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        run.one_step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
    line = f"steps {5 + 20 * w:4d}..{25 + 20 * w:4d}  {dt:.4f} ms/step"
    agg = {}
    for op, units, ms in L.profile_collect():
        agg[f"{op}[{units}]"] = agg.get(f"{op}[{units}]", 0.0) + ms / 20 * 1e3
    L.profile_enable(False)
    if base is None:
        base = agg
    top = sorted(agg, key=lambda k: -abs(agg[k] - base[k]))[:6]
    line += "  us vs first window: " + ", ".join(f"{k} {agg[k]:.1f} ({agg[k] - base[k]:+.1f})" for k in top)
    print(line, flush=True)
