"""Where do two long runs of a method part ways?  Run A: one stream, sampling at the start of each step; run B: second
stream + sampling ahead (the defaults).  A checksum of the parameters every `every` steps; an eval pass (as bench.py's
quality gate) after the steps listed.   usage: long_divergence.py [method] [steps] [every] [eval steps, comma separated]"""
import hashlib, sys, torch
sys.path.insert(0, ".")
import bench
import fruitnerf_amd.training as T
from fruitnerf_amd.data import synthetic_apple as sa
from fruitnerf_amd.rays import RayBundle

dev = torch.device("cuda", 0)
scene = sa.make_scene(seed=0, device=dev)
c2w = sa.make_cameras(bench.N_CAMERAS, seed=0, device=dev)
data = sa.render_dataset(scene, c2w, H=800, W=800, fx=1111.0, fy=1111.0)
i_train, _ = bench.split_indices(bench.N_CAMERAS, bench.TRAIN_SPLIT)
method = sys.argv[1] if len(sys.argv) > 1 else "fruit_nerf_big"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2300
every = int(sys.argv[3]) if len(sys.argv) > 3 else 50
eval_at = {int(a) for a in sys.argv[4].split(",")} if len(sys.argv) > 4 else {2000}


def digest(r):
    return hashlib.sha1(r.model.arena().params.cpu().numpy().tobytes()).hexdigest()[:12]


def run(overlap, ahead):
    T.OVERLAP_PROPOSAL_BACKWARD, T.SAMPLE_AHEAD = overlap, ahead
    r = bench.MethodRun(method, "bf16x3", "SO3xR3", dev, 0, 1, data, torch.as_tensor(i_train, device=dev), len(i_train))
    out = {}
    for i in range(steps):
        r.one_step(want_metrics=False)
        if (i + 1) % every == 0:
            out[i + 1] = digest(r)
        if i + 1 in eval_at:
            r.model.eval()
            with torch.no_grad():
                g = torch.Generator(device=dev); g.manual_seed(7)
                for img in (3, 7):
                    n = 65536
                    y = torch.randint(0, 800, (n,), device=dev, generator=g); x = torch.randint(0, 800, (n,), device=dev, generator=g)
                    o, d = sa.pixel_rays(c2w, torch.full((n,), img, device=dev), y, x, 1111.0, 1111.0, 400.0, 400.0)
                    for s in range(0, n, 32768):
                        float(r.model(RayBundle(o[s:s + 32768], d[s:s + 32768], None, None))["rgb"].sum())
            r.model.train()
    return out


if len(sys.argv) > 5 and sys.argv[5].startswith("repeat"):   # repeat<N>: N runs in the default mode, digests only
    for k in range(int(sys.argv[5][6:])):
        b = run(True, True)
        print("two streams + ahead, run", k, {s: b[s] for s in sorted(b) if s % (every * 8) == 0 or s == max(b)}, flush=True)
    sys.exit(0)
a = run(False, False)
print("run A digests:", {k: a[k] for k in sorted(a) if k % (every * 8) == 0 or k == max(a)}, flush=True)   # compare across processes
if len(sys.argv) > 5 and sys.argv[5] == "only-a":
    sys.exit(0)
for name, (o, h) in {"one stream again": (False, False), "second stream + ahead": (True, True)}.items():
    b = run(o, h)
    bad = [k for k in sorted(a) if a[k] != b[k]]
    print(name, "first difference at step", bad[0] if bad else None, f"({len(bad)} of {len(a)} checkpoints differ)", flush=True)
