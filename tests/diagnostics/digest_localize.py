"""Localise the rare divergence of long `fruit_nerf_big` runs (round 4: 3 of 55 bench-flow runs of 3 000 steps left the
others, tests/diagnostics/bench_flow_digest.py): the same flow, with a digest of EVERY parameter tensor (+ both Adam
moments of the hash tables, + the camera poses) every `every` steps.  Run 0 is the reference; a later run that differs
prints the first checkpoint at which it does and WHICH tensors differ there (and at the next checkpoint) — the tensors
that move first name the kernel — plus the scatter's overflow counter (records that went through float atomics).
usage: digest_localize.py [method] [runs] [steps] [every]      env: FNR_OVERLAP_PROPOSAL_BACKWARD / FNR_SAMPLE_AHEAD /
FNR_STREAM_SAFE bisect the cause"""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
import fruitnerf_amd.training as T  # noqa: E402
from fruitnerf_amd import _lib as L  # noqa: E402
from fruitnerf_amd.data import synthetic_apple as sa  # noqa: E402
from fruitnerf_amd.rays import RayBundle  # noqa: E402

dev = torch.device("cuda", 0)
method = sys.argv[1] if len(sys.argv) > 1 else "fruit_nerf_big"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
every = int(sys.argv[4]) if len(sys.argv) > 4 else 100
eval_at = {500, 1000, 2000, 2500, 3000}
HW, focal = 800, 1111.0
scene = sa.make_scene(seed=0, device=dev)
c2w = sa.make_cameras(bench.N_CAMERAS, seed=0, device=dev)
data = sa.render_dataset(scene, c2w, H=HW, W=HW, fx=focal, fy=focal)
i_train, i_eval = bench.split_indices(bench.N_CAMERAS, bench.TRAIN_SPLIT)


def digests(run):
    """[(name, int)] for every parameter, the tables' moments and the poses: two integer sums per tensor, one host read."""
    model, opt = run.model, run.opt
    arena = model.arena()
    offs = {id(p): (off, n) for _, p, off, n in arena.entries}
    items = []
    for name, p in model.named_parameters():
        off, n = offs[id(p)]
        items.append((name, arena.params[off:off + n]))
        if name.endswith("hash_table"):
            items.append((name + ".exp_avg", opt.exp_avg[off:off + n]))
            items.append((name + ".exp_avg_sq", opt.exp_avg_sq[off:off + n]))
    if run.camera is not None:
        items.append(("camera.pose_adjustment", run.camera[0].pose_adjustment.data.reshape(-1)))
    vals = []
    for _, t in items:
        v = t.detach().contiguous().view(torch.int32).to(torch.int64)
        w = (torch.arange(v.numel(), device=v.device, dtype=torch.int64) % 1000003) + 1
        vals.append(torch.stack([v.sum(), (v * w).sum()]))
    host = torch.stack(vals).cpu().tolist()
    return [(name, a * 1000003 + b) for (name, _), (a, b) in zip(items, host)]


def eval_pass(model):
    model.eval()
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    with torch.no_grad():
        for img in i_eval[:5]:
            n = 65536
            y = torch.randint(0, HW, (n,), device=dev, generator=g)
            x = torch.randint(0, HW, (n,), device=dev, generator=g)
            ci = torch.full((n,), int(img), device=dev)
            o, d = sa.pixel_rays(c2w, ci, y, x, focal, focal, HW / 2.0, HW / 2.0)
            for s in range(0, n, 32768):
                float(model(RayBundle(o[s:s + 32768], d[s:s + 32768], None, None))["rgb"].sum())
    model.train()


def one_run():
    run = bench.MethodRun(method, "bf16x3", "SO3xR3", dev, 0, 1, data, torch.as_tensor(i_train, device=dev), len(i_train))
    for _ in range(20):
        run.one_step()
    torch.cuda.synchronize()
    bench.timed_window(run, 200, lambda: None, False, dev)
    L.profile_enable(True)
    T.SERIALIZE_STREAMS = True
    for _ in range(12):
        run.one_step()
    torch.cuda.synchronize()
    T.SERIALIZE_STREAMS = False
    L.profile_collect()
    L.profile_enable(False)
    out = {}
    while run.step_idx < steps:
        run.one_step(want_metrics=False)
        if run.step_idx % every == 0:
            out[run.step_idx] = digests(run)
        if run.step_idx in eval_at:
            eval_pass(run.model)
    return out


print(f"{method}: {runs} runs x {steps} steps, digests every {every}; overlap {T.OVERLAP_PROPOSAL_BACKWARD} ahead {T.SAMPLE_AHEAD} "
      f"stream_safe {T.STREAM_SAFE} sparse_touch {T.SPARSE_TOUCH_SKIPPING}", flush=True)
ref = None
t0 = time.time()
for k in range(runs):
    L.scatter_overflows(reset=True)
    got = one_run()
    ovf = L.scatter_overflows()
    if ref is None:
        ref = got
        print(f"run 0: reference, {len(got)} checkpoints x {len(next(iter(got.values())))} tensors; scatter overflows {ovf}; "
              f"{time.time() - t0:.0f} s", flush=True)
        continue
    marks = sorted(ref)
    bad = [m for m in marks if got[m] != ref[m]]
    if not bad:
        print(f"run {k}: identical at all {len(marks)} checkpoints; scatter overflows {ovf}; {time.time() - t0:.0f} s", flush=True)
        continue
    first = bad[0]
    nxt = marks[marks.index(first) + 1] if marks.index(first) + 1 < len(marks) else None
    diff0 = [n for (n, a), (_, b) in zip(got[first], ref[first]) if a != b]
    print(f"run {k}: DIFFERS first at step {first} ({len(bad)} of {len(marks)} checkpoints); scatter overflows {ovf}", flush=True)
    print(f"   tensors that differ at step {first}: {diff0}", flush=True)
    if nxt is not None:
        print(f"   ... and at step {nxt}: {[n for (n, a), (_, b) in zip(got[nxt], ref[nxt]) if a != b]}", flush=True)
