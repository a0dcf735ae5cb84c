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
# Per-BIN checksums of both proposal tables every step (always on: two reductions per step): at an event step they name the
# (level, accumulate bin) whose rows differ — one bin (a queue count), a whole level (the level's maximum, i.e. the
# fixed-point scale), or scattered rows (records).
# FNR_DIGEST_SEEN=1 — ONLY with a `seen` build of the library (-DFNR_SCATTER_DEBUG_SEEN), which left the tree in round 5 once
# the divergence was confirmed and fixed (the instrumented hash_scatter.hip is in the history at commit 6211740):
# what the scatter kernels of the step saw — the queue count and level maximum every accumulate workgroup of the two
# proposal levels READ, and the records the emit kernels PLACED per level — is copied out of the library every step.
SEEN_DIGEST = os.environ.get("FNR_DIGEST_SEEN") == "1"
PROP_BIN_ROWS = 4096        # rows per accumulate bin of a 2^17-row proposal table (hash_scatter.hip: scatter_plan)
SEEN_SLOTS, SEEN_BINS, MAX_LEVELS = 3, 4096, 16


def seen_views(buf):
    """The library's ScatterSeen block (hash_scatter.hip) as tensors: acc_n, acc_vmax [3, 4096] int32, emit_records [3, 16] int64."""
    words = SEEN_SLOTS * SEEN_BINS
    acc_n = buf[:4 * words].view(torch.int32).view(SEEN_SLOTS, SEEN_BINS)
    acc_v = buf[4 * words:8 * words].view(torch.int32).view(SEEN_SLOTS, SEEN_BINS)
    o = 8 * words
    emit = buf[o:o + 8 * SEEN_SLOTS * MAX_LEVELS].view(torch.int64).view(SEEN_SLOTS, MAX_LEVELS)
    o += 8 * SEEN_SLOTS * MAX_LEVELS + 8 * SEEN_SLOTS          # emit_records, emit_calls
    hq = 32 * 8                                               # SEEN_HQ: (level, bin) pairs of a proposal call
    emit_sum = buf[o:o + 8 * 2 * hq].view(torch.int64).view(2, hq)
    acc_sum = buf[o + 8 * 2 * hq:o + 16 * 2 * hq].view(torch.int64).view(2, hq)
    acc_grad = buf[o + 16 * 2 * hq:o + 24 * 2 * hq].view(torch.int64).view(2, hq)
    acc_pmv = buf[o + 24 * 2 * hq:o + 32 * 2 * hq].view(torch.int64).view(2, hq)
    o += 32 * 2 * hq
    # complements of the SMALLEST count / maximum any wave of the workgroup read (acc_n / acc_v hold the largest)
    n_minc = buf[o:o + 4 * words].view(torch.int32).view(SEEN_SLOTS, SEEN_BINS)
    v_minc = buf[o + 4 * words:o + 8 * words].view(torch.int32).view(SEEN_SLOTS, SEEN_BINS)
    return acc_n, acc_v, emit, emit_sum, acc_sum, acc_grad, acc_pmv, n_minc, v_minc


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
    serialize_was = T.SERIALIZE_STREAMS      # (FNR_SERIALIZE_STREAMS=1: the hunt's `serial` leg keeps it for the whole run)
    T.SERIALIZE_STREAMS = True
    for _ in range(12):
        run.one_step()
    torch.cuda.synchronize()
    T.SERIALIZE_STREAMS = serialize_was
    L.profile_collect()
    L.profile_enable(False)
    sp = spans_of(run)
    if WS_DIGEST:
        wv = workspace_views(run)
        assert wv is not None, "the paired proposal backward has not run yet"
        sp = sp + wv
    names = COLS + [n for n, _ in sp]
    rec = torch.zeros(steps, len(names), dtype=torch.int64, device=dev)
    tables = [t for n, t in sp if n.endswith(".table") and n.startswith("prop")]
    nbin = [t.numel() // (2 * PROP_BIN_ROWS) for t in tables]
    bins = torch.zeros(steps, len(tables), max(nbin), dtype=torch.int64, device=dev)
    seen = seen_buf = None
    if SEEN_DIGEST:
        import ctypes as C
        lib = L.load()
        if not hasattr(lib, "fnr_debug_scatter_seen_bytes"):
            raise SystemExit("FNR_DIGEST_SEEN=1 needs the instrumented library (-DFNR_SCATTER_DEBUG_SEEN: hash_scatter.hip of "
                             "commit 6211740); the shipped library does not export fnr_debug_scatter_seen_*")
        lib.fnr_debug_scatter_seen_bytes.restype = C.c_size_t
        lib.fnr_debug_scatter_seen_copy.restype = C.c_size_t
        lib.fnr_debug_scatter_seen_copy.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        nb = int(lib.fnr_debug_scatter_seen_bytes())
        seen_buf = torch.zeros((nb + 7) // 8 * 8, dtype=torch.uint8, device=dev)
        # per step, slots 1 and 2 (the two proposal levels): counts and maxima of their bins, records placed per level
        seen = {"n": torch.zeros(steps, 2, max(nbin), dtype=torch.int32, device=dev),
                "vmax": torch.zeros(steps, 2, max(nbin), dtype=torch.int32, device=dev),
                "emit": torch.zeros(steps, 2, MAX_LEVELS, dtype=torch.int64, device=dev),
                # order-independent checksums of every bin's records: as written by emit / as read back by accumulate
                "emit_sum": torch.zeros(steps, 2, 256, dtype=torch.int64, device=dev),
                "acc_sum": torch.zeros(steps, 2, 256, dtype=torch.int64, device=dev),
                # checksum of every bin's fixed-point gradient sums as the optimiser sweep found them in LDS
                "acc_grad": torch.zeros(steps, 2, 256, dtype=torch.int64, device=dev),
                # checksum of the parameters and moments the fused optimiser sweep read for every bin's rows
                "acc_pmv": torch.zeros(steps, 2, 256, dtype=torch.int64, device=dev),
                # smallest count / maximum any wave of the bin's workgroup read ("n" / "vmax" are the largest)
                "n_min": torch.zeros(steps, 2, max(nbin), dtype=torch.int32, device=dev),
                "vmax_min": torch.zeros(steps, 2, max(nbin), dtype=torch.int32, device=dev)}
        assert lib.fnr_debug_scatter_seen_copy(L.ptr(seen_buf), seen_buf.numel(), 1, L.stream_ptr(dev)) == nb   # reset
    while run.step_idx < steps:
        i = run.step_idx
        ld, _ = run.one_step(want_metrics=False)
        rec[i, 0:3] = torch.stack([ld[k] for k in COLS]).view(torch.int32).to(torch.int64)
        for c, (_, t) in enumerate(sp):
            rec[i, 3 + c] = t.view(torch.int32).sum(dtype=torch.int64)
        for q, t in enumerate(tables):
            bins[i, q, :nbin[q]] = t.view(torch.int32).view(nbin[q], -1).sum(dim=1, dtype=torch.int64)
        if seen is not None:
            lib.fnr_debug_scatter_seen_copy(L.ptr(seen_buf), seen_buf.numel(), 1, L.stream_ptr(dev))
            acc_n, acc_v, emit, emit_sum, acc_sum, acc_grad, acc_pmv, n_minc, v_minc = seen_views(seen_buf)
            seen["n_min"][i] = ~n_minc[1:3, :max(nbin)]
            seen["vmax_min"][i] = ~v_minc[1:3, :max(nbin)]
            seen["acc_grad"][i] = acc_grad
            seen["acc_pmv"][i] = acc_pmv
            seen["n"][i] = acc_n[1:3, :max(nbin)]
            seen["vmax"][i] = acc_v[1:3, :max(nbin)]
            seen["emit"][i] = emit[1:3]
            seen["emit_sum"][i] = emit_sum
            seen["acc_sum"][i] = acc_sum
        if run.step_idx in eval_at:
            eval_pass(run.model)
    extra = {"bins": bins.cpu(), "nbin": nbin}
    if seen is not None:
        extra["seen"] = {k: v.cpu() for k, v in seen.items()}
    return names, rec.cpu(), extra


def explain(step, extra, ref_extra):
    """What the finer records say about the first differing step."""
    b, rb = extra["bins"][step], ref_extra["bins"][step]
    for q in range(b.shape[0]):
        nb = extra["nbin"][q]
        bad = (b[q, :nb] != rb[q, :nb]).nonzero().flatten().tolist()
        if not bad:
            continue
        per_level = nb // 5 if nb % 5 == 0 else nb
        where = [(g // per_level, g % per_level) for g in bad]
        print(f"      prop{q}.table: {len(bad)} of {nb} bins differ; (level, bin): {where[:24]}{' ...' if len(where) > 24 else ''}",
              flush=True)
        print(f"         deltas: {[int(b[q, g] - rb[q, g]) for g in bad[:12]]}", flush=True)
    if "seen" not in extra:
        return
    sn, rn = extra["seen"], ref_extra["seen"]
    for q in range(2):
        nb = extra["nbin"][q]
        dn = (sn["n"][step, q, :nb] != rn["n"][step, q, :nb]).nonzero().flatten().tolist()
        dv = (sn["vmax"][step, q, :nb] != rn["vmax"][step, q, :nb]).nonzero().flatten().tolist()
        de = (sn["emit"][step, q] != rn["emit"][step, q]).nonzero().flatten().tolist()
        # the accumulate kernel indexes its bins as gbin = level * bins_per_level + bin, like the table's
        print(f"      prop{q} scatter: counts READ differ in bins {dn[:16]} ({[(int(sn['n'][step, q, g]), int(rn['n'][step, q, g])) for g in dn[:8]]}), "
              f"maxima READ differ in bins {dv[:16]}, records PLACED differ at levels {de}", flush=True)
        dw = (sn["emit_sum"][step, q] != rn["emit_sum"][step, q]).nonzero().flatten().tolist()
        dr = (sn["acc_sum"][step, q] != rn["acc_sum"][step, q]).nonzero().flatten().tolist()
        dg = (sn["acc_grad"][step, q] != rn["acc_grad"][step, q]).nonzero().flatten().tolist()
        print(f"         record checksums vs the reference run: WRITTEN differ at (level, bin) {[(w // 32, w % 32) for w in dw[:12]]}, "
              f"READ BACK differ at {[(w // 32, w % 32) for w in dr[:12]]}; gradient sums in LDS differ at "
              f"{[(w // 32, w % 32) for w in dg[:12]]}; parameters / moments READ by the sweep differ at "
              f"{[(w // 32, w % 32) for w in (sn['acc_pmv'][step, q] != rn['acc_pmv'][step, q]).nonzero().flatten().tolist()[:12]]}",
              flush=True)
        tot_n = [int(sn["n"][step, q, lv * (nb // 5):(lv + 1) * (nb // 5)].sum()) for lv in range(5)] if nb % 5 == 0 else []
        print(f"         records placed per level {sn['emit'][step, q, :5].tolist()} vs counts read per level {tot_n}", flush=True)


print(f"{method}: {runs} runs x {steps} steps, a record per step; overlap {T.OVERLAP_PROPOSAL_BACKWARD} ahead {T.SAMPLE_AHEAD} "
      f"stream_safe {T.STREAM_SAFE} sparse_touch {T.SPARSE_TOUCH_SKIPPING}", flush=True)
ref = None
t0 = time.time()
def self_check(k, extra):
    """Needs no reference run: within one step the counts the accumulate workgroups READ must add up, level by level, to the
    records the emit kernel PLACED (no queue overflowed: bench's overflow counter), and the bins of a level must all have
    read the same level maximum."""
    if "seen" not in extra:
        return
    sn = extra["seen"]
    for q in range(2):
        nb = extra["nbin"][q]
        if nb % 5:
            continue
        per = nb // 5
        n = sn["n"][:, q, :nb].to(torch.int64).view(-1, 5, per)
        placed = sn["emit"][:, q, :5]
        bad = (n.sum(dim=2) != placed).any(dim=1).nonzero().flatten().tolist()
        v = sn["vmax"][:, q, :nb].view(-1, 5, per)
        uneven = (v != v[:, :, :1]).any(dim=2).any(dim=1).nonzero().flatten().tolist()
        # (level, bin) pairs whose records came back from the queue with another checksum than they were written with
        torn = (sn["emit_sum"][:, q] != sn["acc_sum"][:, q]).any(dim=1).nonzero().flatten().tolist()
        if torn:
            st = torn[0]
            where = (sn["emit_sum"][st, q] != sn["acc_sum"][st, q]).nonzero().flatten().tolist()
            print(f"run {k}: prop{q} SELF-CHECK: records read back != records written at steps {torn[:8]}; "
                  f"step {st}: (level, bin) {[(w // 32, w % 32) for w in where[:16]]}", flush=True)
        # THE mechanism found at the end of round 4 (hash_scatter.hip, accumulate_bin): the waves of one accumulate workgroup
        # read DIFFERENT values of the bin's count or of the level's maximum — a reset store (this workgroup's thread 0, or
        # the level's last workgroup) overtook a wave's load that the workgroup barrier had not waited for
        ran = sn["n"][:, q, :nb] != 0           # (steps that trained the networks; an all-zero row has min = ~0 = -1)
        split_n = ((sn["n"][:, q, :nb] != sn["n_min"][:, q, :nb]) & ran)
        split_v = ((sn["vmax"][:, q, :nb] != sn["vmax_min"][:, q, :nb]) & ran)
        split = (split_n | split_v).any(dim=1).nonzero().flatten().tolist()
        if split:
            st = split[0]
            wn, wv = split_n[st].nonzero().flatten().tolist(), split_v[st].nonzero().flatten().tolist()
            print(f"run {k}: prop{q} SELF-CHECK: the WAVES of a workgroup read different counters at steps {split[:8]}; "
                  f"step {st}: count differs in bins {wn[:8]} "
                  f"({[(int(sn['n_min'][st, q, g]), int(sn['n'][st, q, g])) for g in wn[:4]]}), maximum (bits) in bins {wv[:8]} "
                  f"({[(int(sn['vmax_min'][st, q, g]), int(sn['vmax'][st, q, g])) for g in wv[:4]]})", flush=True)
        if bad or uneven:
            print(f"run {k}: prop{q} SELF-CHECK: counts read != records placed at steps {bad[:8]}; "
                  f"bins of one level read different maxima at steps {uneven[:8]}", flush=True)
            for st in bad[:2]:
                print(f"      step {st}: placed {placed[st].tolist()} read {n[st].sum(dim=1).tolist()}", flush=True)
            for st in uneven[:2]:
                lv = (v[st] != v[st, :, :1]).any(dim=1).nonzero().flatten().tolist()
                print(f"      step {st}: levels {lv}: maxima (bits) {[sorted(set(v[st, l].tolist())) for l in lv]}", flush=True)


ref_extra = None
for k in range(runs):
    names, rec, extra = one_run()
    self_check(k, extra)
    if ref is None:
        ref, ref_extra = rec, extra
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
            explain(r, extra, ref_extra)
