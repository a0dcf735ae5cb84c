"""The hunt for the one `fruit_nerf_big` bench run that ended elsewhere (DESIGN 4 "Measured (round 3)", 7): bench.py's own
flow for a method — warm-up, the timed window with its event-timed (stream-serialised) steps, the 12-step breakdown pass,
then training on to `marks` with an eval pass at each, as the quality gate does — repeated `runs` times in ONE process and
(run this script several times) across processes; a digest of the parameters at every mark.  All digests of a mark must be
equal; the first odd one out names the interval in which a run left the others, and FNR_OVERLAP_PROPOSAL_BACKWARD=0 /
FNR_SAMPLE_AHEAD=0 / FNR_BENCH_PROFILE_EVERY=1000 bisect the cause.
usage: bench_flow_digest.py [method] [runs] [marks, comma separated, default 500,1000,2000,2500,3000]"""
import hashlib
import sys
import torch
sys.path.insert(0, ".")
import bench
import fruitnerf_amd.training as T
from fruitnerf_amd import _lib as L
from fruitnerf_amd.data import synthetic_apple as sa
from fruitnerf_amd.rays import RayBundle

dev = torch.device("cuda", 0)
method = sys.argv[1] if len(sys.argv) > 1 else "fruit_nerf_big"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
marks = [int(m) for m in (sys.argv[3] if len(sys.argv) > 3 else "500,1000,2000,2500,3000").split(",")]
HW, focal = 800, 1111.0
scene = sa.make_scene(seed=0, device=dev)
c2w = sa.make_cameras(bench.N_CAMERAS, seed=0, device=dev)
data = sa.render_dataset(scene, c2w, H=HW, W=HW, fx=focal, fy=focal)
i_train, i_eval = bench.split_indices(bench.N_CAMERAS, bench.TRAIN_SPLIT)


def digest(t):
    # on the device: a .cpu() copy of a 300 MB arena per mark is what made an earlier stress run cost 9 GPU-minutes
    v = t.detach().contiguous().view(-1).view(torch.int32).to(torch.int64)
    w = torch.arange(1, v.numel() + 1, device=v.device, dtype=torch.int64) % 1000003
    return hashlib.sha1(f"{int(v.sum())}:{int((v * w).sum())}".encode()).hexdigest()[:10]


def eval_pass(model):   # bench.py's heldout_quality(), numbers dropped
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
    for _ in range(20):                       # bench.py's default warm-up
        run.one_step()
    torch.cuda.synchronize()
    bench.timed_window(run, 200, lambda: None, False, dev)          # events on every PROFILE_EVERY-th step, serialised
    L.profile_enable(True)
    T.SERIALIZE_STREAMS = True
    for _ in range(12):                       # the per-entry-point breakdown pass
        run.one_step()
    torch.cuda.synchronize()
    T.SERIALIZE_STREAMS = False
    L.profile_collect()
    L.profile_enable(False)
    out = {}
    for mark in marks:
        while run.step_idx < mark:
            run.one_step(want_metrics=False)
        out[mark] = digest(run.model.arena().params)
        eval_pass(run.model)
    return out


ref = None
for k in range(runs):
    got = one_run()
    ref = ref or got
    odd = [m for m in marks if got[m] != ref[m]]
    print(f"{method} run {k}: {got}" + (f"  <-- differs from run 0 from mark {odd[0]} on" if odd else ""), flush=True)
