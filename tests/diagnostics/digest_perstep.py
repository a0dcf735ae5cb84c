"""Which step, and which parameters FIRST?  The rare divergence of long `fruit_nerf_big` runs (digest_localize.py: by the
next 50-step checkpoint every tensor differs) with a record per STEP kept on the device: the bits of the step's three losses
and an integer checksum of eight spans of the parameter arena (proposal network 0 / 1: table, MLP; field: embedding, table,
MLPs; camera poses).  Run 0 is the reference; a run that leaves it prints the first steps at which anything differs and
which columns.  Reading: losses differ first -> that step's forward saw different inputs (rays / look-ahead) or is itself
nondeterministic; a single span first -> the kernels that write it.
usage: digest_perstep.py [method] [runs] [steps]"""
import os
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
eval_at = {500, 1000, 2000, 2500, 3000}
HW, focal = 800, 1111.0
scene = sa.make_scene(seed=0, device=dev)
c2w = sa.make_cameras(bench.N_CAMERAS, seed=0, device=dev)
data = sa.render_dataset(scene, c2w, H=HW, W=HW, fx=focal, fy=focal)
i_train, i_eval = bench.split_indices(bench.N_CAMERAS, bench.TRAIN_SPLIT)
COLS = ["rgb_loss", "semantics_loss", "interlevel_loss"]
# FNR_DIGEST_WS=1: also a checksum per step of what the proposal networks' MLP backward left for the scatter — the d_feats
# region at the head of each level's persistent workspace (unchanged on steps that do not train the networks).  At an event
# step it tells "the scatter's INPUT already differed" (k_prop_bwd's output, or memory corrupted after it) from "emit /
# accumulate went wrong on equal inputs".
WS_DIGEST = os.environ.get("FNR_DIGEST_WS") == "1"


def workspace_views(run):
    from fruitnerf_amd import _kernels as K
    out = []
    samples = run.model.config.num_proposal_samples_per_ray
    for q, net in enumerate(run.model.proposal_networks):
        # the workspace of THIS run: keyed by its model's second stream (round-4 call 14 took the first run's buffer
        # for every run — each model has its own stream out of torch's pool — and compared frozen bytes)
        side = run.model.__dict__.get("_side_stream")
        cands = [t for key, t in K._SCATTER_WS.items()
                 if key[4] == f"prop{q}" and side is not None and key[2] == side.cuda_stream]
        if not cands:
            return None
        ws = max(cands, key=lambda t: t.numel())
        nbytes = int(net.prop_struct().grid.n_levels) * run.rays * int(samples[q]) * 8
        out.append((f"prop{q}.d_feats", ws[:nbytes]))
    return out


def spans_of(run):
    """[(name, tensor view)] — contiguous pieces of the arena + the poses."""
    model = run.model
    arena = model.arena()
    offs = {id(p): (off, n) for _, p, off, n in arena.entries}
    out = []
    for i, net in enumerate(model.proposal_networks):
        off, n = offs[id(net.encoding.hash_table)]
        out.append((f"prop{i}.table", arena.params[off:off + n]))
        ps = [offs[id(p)] for p in net.mlp_base[1].parameters()]
        a, b = min(o for o, _ in ps), max(o + k for o, k in ps)
        out.append((f"prop{i}.mlp", arena.params[a:b]))
    f = model.field
    off, n = offs[id(f.embedding_appearance.embedding.weight)]
    out.append(("field.embedding", arena.params[off:off + n]))
    off, n = offs[id(f.mlp_base_grid.hash_table)]
    out.append(("field.table", arena.params[off:off + n]))
    rest = [offs[id(p)] for name, p in f.named_parameters() if "hash_table" not in name and "embedding" not in name]
    a, b = min(o for o, _ in rest), max(o + k for o, k in rest)
    out.append(("field.mlps", arena.params[a:b]))
    out.append(("camera.poses", run.camera[0].pose_adjustment.data.reshape(-1)))
    return out


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
    sp = spans_of(run)
    if WS_DIGEST:
        wv = workspace_views(run)
        assert wv is not None, "the paired proposal backward has not run yet"
        sp = sp + wv
    names = COLS + [n for n, _ in sp]
    rec = torch.zeros(steps, len(names), dtype=torch.int64, device=dev)
    while run.step_idx < steps:
        i = run.step_idx
        ld, _ = run.one_step(want_metrics=False)
        rec[i, 0:3] = torch.stack([ld[k] for k in COLS]).view(torch.int32).to(torch.int64)
        for c, (_, t) in enumerate(sp):
            rec[i, 3 + c] = t.view(torch.int32).sum(dtype=torch.int64)
        if run.step_idx in eval_at:
            eval_pass(run.model)
    return names, rec.cpu()


print(f"{method}: {runs} runs x {steps} steps, a record per step; overlap {T.OVERLAP_PROPOSAL_BACKWARD} ahead {T.SAMPLE_AHEAD} "
      f"stream_safe {T.STREAM_SAFE} sparse_touch {T.SPARSE_TOUCH_SKIPPING}", flush=True)
ref = None
t0 = time.time()
for k in range(runs):
    names, rec = one_run()
    if ref is None:
        ref = rec
        print(f"run 0: reference ({rec.shape[0]} steps x {names}); {time.time() - t0:.0f} s", flush=True)
        continue
    diff = rec != ref
    rows = diff.any(dim=1).nonzero().flatten().tolist()
    if not rows:
        print(f"run {k}: identical at every step; {time.time() - t0:.0f} s", flush=True)
        continue
    print(f"run {k}: DIFFERS from step {rows[0]} on ({len(rows)} steps differ)", flush=True)
    for r in rows[:4]:
        print(f"   step {r}: {[n for n, d in zip(names, diff[r].tolist()) if d]}", flush=True)
        if r == rows[0]:
            for n, a, b in zip(names, rec[r].tolist(), ref[r].tolist()):
                if a != b:
                    print(f"      {n}: {a} vs {b} (delta {a - b})", flush=True)
