#!/usr/bin/env python3
"""bench.py — train rays/s of the `fruit_nerf` method on the synthetic apple scene (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full training iteration of the reference's hot loop on every rank:
pixel sampling + ray generation on the device, set_anneal, FruitModel.forward (proposal sampling 256/96/48,
two proposal nets, main field), get_metrics_dict (PSNR + distortion), get_loss_dict (MSE + BCE + interlevel),
backward, gradient all-reduce over RCCL (N > 1), fused Adam over all 19.4 M parameters, step_cb.
Every rank draws its own 4096 rays (reference DDP semantics, fruit_pipeline.py:116-118) => weak scaling;
`value` = N * K * 4096 / max-over-ranks wall time, inputs resident in HBM.
The loop body is training.TrainingSteps: the pixel sampling + ray generation + proposal sampling of step i + 1 are
enqueued at the end of step i (second HIP stream, underneath step i's table scatter) — every step still launches each of
its kernels exactly once, the timed K steps contain K of everything (the first one's sampling was enqueued by the last
warm-up step, the last one enqueues the sampling of step K + 1).

`python bench.py --gpus N` without a launcher starts the N ranks itself (re-executes under torch.distributed.run,
the reference's counterpart is nerfstudio's mp.spawn around fruit_pipeline.py:116-118) and fails unless N ranks run.

Extra objects in the JSON line:  roofline (dominant entry point among the HBM- / MFMA-bound ones, chosen by its time
over the timed window itself, HIP events on the stream of the launch, every 10th step, streams serialised on those steps), cpu_baseline (the oracle's train step on the host cores, rank 0, bounded sample),
breakdown_ms (per entry point, from a short instrumented pass after the timed region), quality (PSNR / IoU on
held-out views after --quality-steps more steps).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("FNR_BENCH_FORCE_DIST") == "1":
    # multi-rank: RCCL creates streams of its own, and with HIP's default of 4 hardware queues the training loop's second
    # stream can land on the launch stream's queue (nothing overlaps then; training._second_stream).  Must be set before
    # the HIP runtime starts.  Measured on the one-rank RCCL run: 0.989 -> 0.887 ms/step.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from fruitnerf_amd.hostinfo import usable_cpus  # noqa: E402  (the container's CPU quota, see its docstring)

# Arithmetic generation of the training kernels: bumped whenever a kernel change legitimately changes a rounding (the
# pinned checksums below are per generation).
NUMERICS = "r06"
CHECKSUMS_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "parameter_checksums.json")
N_CAMERAS = 100
TRAIN_SPLIT = 0.9              # fruitnerf_dataparser.py:62
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
L2_PEAK_GBS = 34500.0          # MI355X_MICROARCH.md "L2 (per XCD)": ~34.5 TB/s aggregate over the 8 XCDs
MFMA_F32_PEAK_TF = 157.3
MFMA_BF16_PEAK_TF = 2500.0     # dense bf16 MFMA (the roofline of --mlp-precision bf16 / bf16x3)
# issued bf16 products per algorithmic fp32 product (set in main(): bf16x3 -> fwd 6, bwd (6 + 3 + 3) / 2 per its three
# equal thirds recompute / dX / dW measured against the algorithmic dX + dW; plain bf16 -> 1 and 1.5)
ISSUED_BF16 = {}

# The reference's method configurations (fruit_nerf_config.py:27-164) come from the plugin's own config module
# (fruitnerf_amd/fruit_nerf_config.py: the objects its `nerfstudio.method_configs` entry points resolve to); only the
# ALGORITHMIC work per unit is bench bookkeeping and lives here.
ALG_WORK = {
    "fruit_nerf": dict(mlp_flop=33024.0, flop_per_ray_train=5.13e6, bytes_per_ray_train=0.49e6),
    "fruit_nerf_big": dict(mlp_flop=83584.0, flop_per_ray_train=32.9e6, bytes_per_ray_train=3 * 376832.0),
    "fruit_nerf_huge": dict(mlp_flop=83584.0, flop_per_ray_train=3 * 5.775e6,
                            bytes_per_ray_train=3 * (64 * 1024.0 + 512 * 5 * 64.0 + 512 * 7 * 64.0)),
}


def method_table(name: str) -> dict:
    """What MethodRun needs of a method, read from fruit_nerf_config.METHODS."""
    from fruitnerf_amd import fruit_nerf_config as FC
    M = FC.METHODS[name]
    model = {k: v for k, v in M["model"].items() if k != "eval_num_rays_per_chunk"}
    cam = M["camera_optimizer"]
    sched = cam.get("scheduler") or {}
    algos = {o["algorithm"] for o in M["optimizers"].values()}
    assert len(algos) == 1
    return dict(rays=M["datamanager"]["train_num_rays_per_batch"], model=model, algorithm=algos.pop(),
                groups=FC.group_schedules(name),
                camera=dict(lr=cam["lr"], eps=cam["eps"], weight_decay=cam["weight_decay"], lr_final=sched.get("lr_final"),
                            max_steps=sched.get("max_steps"), algorithm=cam["algorithm"]),
                samples=tuple(model.get("num_proposal_samples_per_ray", (256, 96))) + (model.get("num_nerf_samples_per_ray", 48),),
                max_num_iterations=M["trainer"]["max_num_iterations"], **ALG_WORK[name])


METHODS = {name: method_table(name) for name in ALG_WORK}


def alg_table(mlp_flop: float):
    """ALGORITHMIC bytes / flops per unit: what the reference's dense layers and table
    accesses need, not what the kernels issue.  The MLP backward is dX + dW = 2x the forward FLOP; the forward that
    field_mlp_bwd RECOMPUTES from the saved hash features (a design choice that saves 1.3 KB/sample of activations) is
    issued work, not useful work, and is not counted."""
    return {
        "hash_encode_fwd": ("hbm", 1024.0 + 128.0),          # 16 lvl x 8 corners x 8 B + 128 B features written
        # gradient read-modify-write + d_feats read.  (At N=1 the launch also carries the table's optimiser step: its
        # 28 B per table parameter are added per launch, see roofline_entry(fixed_bytes).)
        "hash_encode_bwd": ("hbm", 2 * 1024.0 + 128.0),
        # proposal networks: their 5.2 MB tables live in the XCDs' L2s (PMC: ~0.1 of these bytes reach the fabric), so
        # the bytes below are L2 traffic, priced against the L2 rate — never an HBM fraction, never the `roofline` entry
        "prop_density_fwd": ("l2", 320.0 + 4.0),             # 5 lvl x 8 corners x 8 B + density
        "prop_density_bwd": ("l2", 2 * 320.0 + 40.0 + 4.0),
        "field_mlp_fwd": ("mfma", mlp_flop),                 # useful FLOP / sample
        "field_mlp_bwd": ("mfma", 2 * mlp_flop),             # dX + dW
        "position_grad": ("hbm", 1024.0 + 2 * 256.0 + 128.0),
        "adam_step": ("hbm", 28.0),                          # p,g,m,v read + p,m,v,g written per parameter (32 B w/ zero)
    }


# The table optimiser fused into the scatter's accumulate kernel (N = 1): the gradient never touches HBM, so a parameter
# costs p, m, v read + p, m, v written = 24 B (the 28 B of the separate step over-priced the launch)
FUSED_ADAM_BYTES_PER_PARAM = 24.0
SCATTER_RECORD_BYTES = 10.0     # 8-byte value pair + 16-bit row inside the bin (csrc/hash_scatter.hip)


# entry points that belong to one kernel family (the judge's grouping: scatter = field + proposal-network scatter)
FAMILIES = {
    "field_mlp_bwd": "field MLP backward (MFMA)", "field_mlp_fwd": "field MLP forward (MFMA)",
    "hash_encode_bwd": "hash-grid scatter", "prop_density_bwd": "proposal-net backward (incl. its scatter)",
    "hash_encode_fwd": "hash-grid gather", "prop_density_fwd": "proposal-net forward", "adam_step": "optimiser",
    "position_grad": "ray gradients",
}


def load_pmc_traffic():
    """HBM bytes per entry-point launch from the COMMITTED rocprofv3 PMC passes of this bench command
    (profiles/pmc_traffic.json, written by tools/make_profile_summary.py from the FETCH_SIZE / WRITE_SIZE passes of
    tools/gpu_call.sh (legs kt, pmc); FETCH_SIZE doubled per MI355X_MICROARCH.md "HBM").  PMC counters cannot be read inside this
    process; the file names the build and command it was collected from."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def roofline_entry(op, units, avg_ms, launches, alg, fixed_bytes=0.0, fixed_note=None, method="fruit_nerf", pmc=None):
    """fixed_bytes: algorithmic bytes of the launch that do not scale with `units` (the table optimiser's step when it
    runs inside hash_encode_bwd: torch.optim's 28 B per table parameter, the same figure adam_step is priced with)."""
    bound, per_unit = alg[op]
    if bound in ("hbm", "l2"):
        alg_per_launch = per_unit * units + fixed_bytes
        achieved = alg_per_launch / (avg_ms * 1e-3) / 1e9
        peak, unit, key = (HBM_PEAK_GBS if bound == "hbm" else L2_PEAK_GBS), "GB/s", "alg_bytes_per_unit"
    else:
        alg_per_launch = per_unit * units
        achieved = alg_per_launch / (avg_ms * 1e-3) / 1e12
        peak, unit, key = MFMA_F32_PEAK_TF, "TFLOP/s", "alg_flop_per_unit"
    extra = {}
    if bound == "mfma":
        peak = MFMA_BF16_PEAK_TF if op in ISSUED_BF16 else MFMA_F32_PEAK_TF
    if bound == "mfma" and op in ISSUED_BF16:
        # bf16x3 / bf16: `achieved` stays the ALGORITHMIC (fp32-equivalent) FLOP rate; the pipe executes ISSUED_BF16[op]
        # bf16 products per algorithmic product (exact three-way split: 6 forward; backward = 6 for the forward
        # recompute it repeats, 3 for dX and dW), which is what loads the 2.5 PFLOP/s bf16 pipe
        issued = achieved * ISSUED_BF16.get(op, 1.0)
        extra = {"frac_of_fp32_mfma_peak": round(achieved / MFMA_F32_PEAK_TF, 4),
                 "issued_bf16_tflops": round(issued, 1), "issued_frac_of_bf16_peak": round(issued / MFMA_BF16_PEAK_TF, 4),
                 "peak_note": "peak = dense bf16 MFMA (the pipe these kernels run on); the arithmetic they replace is "
                              "fp32, whose MFMA peak on gfx950 is 157.3 TFLOP/s (1/16 of bf16)"}
    if bound == "hbm" and fixed_bytes:
        extra = {"alg_bytes_per_launch_fixed": fixed_bytes, "alg_bytes_fixed_note": fixed_note,
                 "achieved_without_fixed_bytes": round(per_unit * units / (avg_ms * 1e-3) / 1e9, 3)}
    if bound == "l2":
        extra = {"bound_note": "tables of this entry point are L2-resident: bytes are L2 traffic against the ~34.5 TB/s "
                               "aggregate L2 rate, not an HBM fraction"}
    # HBM bytes per launch: PMC counters need rocprofv3, so this is read from the committed passes of THIS command
    # (profiles/pmc_traffic.json) and only when that file has this method / entry point / launch size
    traffic = None
    rec = ((pmc or {}).get("entry_points", {}).get(method, {}) or {}).get(f"{op}[{int(units)}]")
    if rec and bound != "mfma":
        traffic = float(rec["bytes_per_launch"])
        extra["traffic_over_algorithmic"] = round(traffic / alg_per_launch, 3)
        extra["traffic_source"] = (f"{pmc.get('source', 'profiles/pmc_traffic.json')}; kernels "
                                   f"{', '.join(rec.get('kernels', []))}; build {pmc.get('build', '?')}")
    elif rec:
        traffic = float(rec["bytes_per_launch"])
        extra["traffic_source"] = f"{pmc.get('source', 'profiles/pmc_traffic.json')}; build {pmc.get('build', '?')}"
    return {"kernel": op, "family": FAMILIES.get(op, op), "bound": bound, "achieved": round(achieved, 3), "peak": peak,
            "unit": unit, "frac": round(achieved / peak, 4), **extra,
            "traffic": traffic,
            "avg_launch_ms": round(avg_ms, 5), "launches": launches, "units_per_launch": int(units), key: per_unit}


# `roofline` candidates: entry points whose bound is one of the two rooflines of SURVEY 8d.  The proposal networks'
# kernels work out of the L2 and are reported under breakdown_ms / roofline_l2 only.
ROOFLINE_OPS = ("hash_encode_bwd", "hash_encode_fwd", "field_mlp_bwd", "field_mlp_fwd", "position_grad", "adam_step")


def pick_rooflines(recs, alg, steps):
    """Dominant (entry point, launch size) of each roofline by its TOTAL time over the timed window (`recs` = HIP-event
    records of every ROOFLINE_OPS launch in that window) -> ((op, units), (op, units) of the other bound | None).
    Deterministic for a given build: no separate pre-pass, ties broken by name."""
    tot = {}
    for op, units, ms in recs:
        if op in ROOFLINE_OPS:
            tot[(op, units)] = tot.get((op, units), 0.0) + ms
    if not tot:
        return None, None
    order = sorted(tot, key=lambda k: (-tot[k], k))
    first = order[0]
    other = [k for k in order if alg[k[0]][0] != alg[first[0]][0]]
    return first, (other[0] if other else None)


def split_indices(n: int, frac: float):
    """Nerfstudio-style 'fraction' split (fruitnerf_dataparser.py:171-186): evenly spaced train images."""
    num_train = int(np.ceil(n * frac))
    i_all = np.arange(n)
    i_train = np.linspace(0, n - 1, num_train, dtype=int)
    i_eval = np.setdiff1d(i_all, i_train)
    return i_train, i_eval


def counting_stage_bench(dev, cpu: bool):
    """The counting stage's three library calls (Open3D remove_radius_outlier / voxel_down_sample, sklearn DBSCAN;
    clustering_base.py:183-207) with the reference's synthetic-apple parameters (config_synthetic.py:2-15) on a
    synthetic 1.1 M-point export cloud: GPU kernels (best of 3, cloud resident in HBM) vs the CPU libraries that are
    importable here (scipy's KD-tree for the radius search, scikit-learn's DBSCAN, all host cores)."""
    import numpy as np
    import torch
    from fruitnerf_amd import _kernels as K
    from fruitnerf_amd.data.synthetic_cloud import make_export_cloud
    X = make_export_cloud(80)
    x = torch.as_tensor(X, device=dev)

    def timed(fn, reps=3):
        best, out = 1e9, None
        for _ in range(reps):
            torch.cuda.synchronize()
            t = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t)
        return out, best * 1e3
    K.cloud_radius_count(x[:1024].contiguous(), 0.01, False)        # module load
    counts, t_cnt = timed(lambda: K.cloud_radius_count(x, 0.01, False))
    xk = x[counts > 200].contiguous()
    (vx, _), t_vox = timed(lambda: K.cloud_voxel_down_sample(xk, None, 0.001))
    (labels, k), t_db = timed(lambda: K.cloud_dbscan(vx, 0.01, 100))
    hits = float(counts.double().sum())
    out = {"cloud": f"{len(X)} lattice points, 80 spheres + 5% clutter, mean {hits / len(X):.0f} neighbours in r=0.01",
           "params": "radius 0.01 / nb_points 200, voxel 0.001, eps 0.01 / min_samples 100",
           "gpu_ms": {"radius_count": round(t_cnt, 2), "voxel_down_sample": round(t_vox, 2), "dbscan": round(t_db, 2)},
           "points_after": {"outlier_removal": int(xk.shape[0]), "voxel_down_sample": int(vx.shape[0])},
           "clusters": int(k.item()),
           "gpu_points_per_s": round(len(X) / ((t_cnt + t_vox + t_db) * 1e-3), 1)}
    if cpu:
        try:
            from scipy.spatial import cKDTree
            from sklearn.cluster import DBSCAN
            t = time.perf_counter()
            c_cpu = cKDTree(X).query_ball_point(X, 0.01, return_length=True, workers=usable_cpus())
            t_c = time.perf_counter() - t
            Xv = vx.cpu().numpy()
            t = time.perf_counter()
            lab = DBSCAN(eps=0.01, min_samples=100, n_jobs=usable_cpus()).fit(Xv).labels_
            t_d = time.perf_counter() - t
            out["cpu_ms"] = {"radius_count_scipy_ckdtree": round(t_c * 1e3, 1), "dbscan_sklearn": round(t_d * 1e3, 1),
                             "cores": usable_cpus()}
            out["labels_equal_sklearn"] = bool(np.array_equal(lab, labels.cpu().numpy()))
            out["counts_close_to_ckdtree"] = float(np.mean(c_cpu == counts.cpu().numpy()))  # <= vs < differ on ties
        except Exception as e:  # noqa: BLE001
            out["cpu_ms"] = f"unavailable: {e}"
    return out


def count_fruits_end_to_end(emodel, pipe, sample_volume, scene, dev, n_side: int = 512):
    """The reference's whole counting pipeline on the trained field: bin-centre export of an n_side^3 lattice, then
    Clustering.count = cluster (GPU: radius-outlier removal, voxel down-sampling, DBSCAN) -> merge_small_clusters ->
    split_large_cluster (alpha-shape volume test against a sphere template of the scene's mean fruit radius, ICP / Ward
    hypotheses scored by Hausdorff distance), scored against the scene's fruit centres (clustering_base.py:463-509).
    Front-end parameters scale with the lattice pitch as in the first-stage count."""
    from fruitnerf_amd.clustering import Clustering, PointCloud
    t0 = time.perf_counter()
    emodel.setup_inference(True, n_side, deterministic=True)
    n_rays = pipe.datamanager.setup_inference(aabb=((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)), num_points=n_side)
    sets = sample_volume(pipe, n_rays, transform_json={"scale": 1.0})
    pts = sets["semantic"]["points"]
    t_export = time.perf_counter() - t0
    if pts.shape[0] < 5:
        return None
    pitch = 2.0 / n_side * 2.0
    fruit = scene.is_fruit.cpu().numpy()
    # Exported coordinates ARE scene coordinates: the export model normalises a lattice point p of the [-1, 1]^3 box by
    # the box ((p + 1) / 2 = (2p + 2) / 4, fruit_field.py:169-175 without spatial distortion), i.e. it reads the field
    # where training put the scene point 2p ((contract(x) + 2) / 4), and the exporter's closing x2
    # (exporter_utils.py:190-191) restores exactly that.  Hence centres and radii unscaled, lattice pitch x2.
    centres = scene.centers.cpu().numpy()[fruit].astype(np.float64)
    radius = float(scene.radii.cpu().numpy()[fruit].mean())
    cl = Clustering(template_path=None, voxel_size_down_sample=pitch / 4, remove_outliers_nb_points=2,
                    remove_outliers_radius=1.8 * pitch, min_samples=4, apple_template_size=1.0,
                    cluster_merge_distance=0.04, gt_cluster=centres, gt_count=int(scene.n_fruits), template_radius=radius)
    if os.environ.get("FNR_BENCH_DUMP_CLOUD"):          # diagnostics: the cloud the count runs on, for offline inspection
        np.savez_compressed(os.environ["FNR_BENCH_DUMP_CLOUD"], points=pts.detach().cpu().numpy() if hasattr(pts, "detach")
                            else np.asarray(pts), centres=centres, radii=scene.radii.cpu().numpy()[fruit], pitch=pitch)
    t1 = time.perf_counter()
    count = cl.count(PointCloud(pts, None, dev), eps=1.8 * pitch)
    return {"count": int(count), "first_stage": int(cl.counter - cl.fuse_counter), "additional": int(cl.additional_count),
            "pruned": int(cl.prune_counter), "true_positive": int(cl.true_positive), "false_positive": int(cl.false_positive),
            "false_negative": int(cl.false_negative), "precision": round(cl.precision, 4), "recall": round(cl.recall, 4),
            "F1": round(cl.F1, 4), "scene_fruits": int(scene.n_fruits), "lattice": f"{n_side}^3",
            "semantic_points": int(pts.shape[0]), "template": f"sphere, radius {radius:.4f} (the scene's mean fruit radius)",
            "export_s": round(t_export, 3), "counting_s": round(time.perf_counter() - t1, 3)}


class MethodRun:
    """Model + optimisers + camera optimiser + pixel batcher of one reference method on the shared synthetic scene."""

    def __init__(self, method, mlp_precision, camera_mode, dev, rank, world, data, train_ids, n_train, batch_seed=1234):
        from fruitnerf_amd.data import synthetic_apple as sa
        from fruitnerf_amd.data.semantics import apple_metadata
        from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
        from fruitnerf_amd.training import FusedAdam
        self.method, self.M, self.world, self.dev = method, METHODS[method], world, dev
        M = self.M
        self.rays = M["rays"]
        self.batcher = sa.PixelBatcher(data, train_ids, seed=batch_seed + rank)   # each rank draws its own rays
        torch.manual_seed(0)                                                    # identical initial weights on every rank
        self.model_cfg = FruitNerfModelConfig(mlp_precision=mlp_precision, **M["model"])
        self.model = FruitModel(self.model_cfg, apple_metadata(), num_train_data=n_train, device=dev)
        self.model.train()
        self.opt = FusedAdam(self.model, algorithm=M["algorithm"], group_lr={k: dict(v) for k, v in M["groups"].items()})
        self.camera = None
        if camera_mode != "off":
            from fruitnerf_amd.cameras.camera_optimizers import CameraAdam, CameraOptimizerConfig
            cm = M["camera"]
            cam_opt = CameraOptimizerConfig(mode=camera_mode, lr=cm["lr"], eps=cm["eps"],
                                            weight_decay=cm["weight_decay"], lr_final=cm["lr_final"],
                                            max_steps=cm["max_steps"] or 1).setup(n_train, dev)
            self.camera = (cam_opt, CameraAdam(cam_opt, algorithm=cm["algorithm"]), self.batcher)
        from fruitnerf_amd.training import TrainingSteps
        # the loop body: batcher.sample -> fused_train_iteration, the next step's rays + proposal sampling enqueued at the
        # end of the current one (training.SAMPLE_AHEAD)
        self.steps = TrainingSteps(self.model, self.opt, self.batcher, self.rays,
                                   camera=self.camera[:2] if self.camera else None, world_size=world)

    @property
    def step_idx(self):
        return self.steps.step_idx

    def set_camera(self, camera):
        """camera optimiser on (the (optimizer, adam, batcher) triple) / off (None) for the following steps."""
        self.camera = camera
        self.steps.camera = camera
        self.steps.drop_lookahead()

    def one_step(self, want_metrics=True):
        return self.steps.step(want_metrics)


# HIP events on the roofline candidates' launches of every 10th step of the timed window.  Such a step runs its two streams
# one after the other (training.SERIALIZE_STREAMS) and costs ~0.19 ms more than a normal one (measured: 0.777 ms/step
# without timed steps, 0.815 with every 5th): every 10th = two timed launches per entry point in the driver's 20-step
# window, twenty in the default 200-step one, for ~2 % of the reading
PROFILE_EVERY = int(os.environ.get("FNR_BENCH_PROFILE_EVERY", "10"))


def timed_window(run, steps, barrier, dist_on, dev):
    """EXACTLY `steps` training steps between barrier + synchronize on both sides.  On every PROFILE_EVERY-th step the
    launches of the roofline candidates are bracketed by HIP events on the launch stream (an event pair costs the GPU
    a ~3 us bubble: all candidates on every step cost 5 % of the step, measured) and the second HIP stream's launches
    are serialised with the launch stream's -> (seconds: max over ranks, host enqueue seconds, event records, profiled
    steps, last (loss_dict, metrics))."""
    from fruitnerf_amd import _lib as L
    import fruitnerf_amd.training as T
    serialize = T.SERIALIZE_STREAMS
    L.profile_enable(True, ops=list(ROOFLINE_OPS))
    if dist_on:      # every collective of the window bracketed by two events on the stream that issues / waits for it
        T.COLLECTIVE_LOG = []
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    n_prof = 0
    host_plain = host_prof = 0.0
    try:
        for i in range(steps):
            on = i % PROFILE_EVERY == 0
            L.profile_pause(not on)
            # a bracketed launch must have the GPU to itself: on the profiled steps the second stream's launches (proposal
            # backward, ray-gradient reduction, camera step, next step's sampling) run before / after the launch stream's,
            # not next to them (training.SERIALIZE_STREAMS; same results either way)
            T.SERIALIZE_STREAMS = bool(on) or serialize
            n_prof += on
            th = time.perf_counter()
            last = run.one_step()
            th = time.perf_counter() - th
            if on:
                host_prof += th
            else:
                host_plain += th
    finally:
        T.SERIALIZE_STREAMS = serialize
    # host time of the steps that are NOT event-timed (replayed by the native sequencer once their shape is recorded) and of
    # the event-timed ones (always interpreted: their launches carry HIP events and their streams are serialised)
    run.host_ms = {"plain_steps": round(host_plain / max(steps - n_prof, 1) * 1e3, 4),
                   "event_timed_steps": round(host_prof / max(n_prof, 1) * 1e3, 4), "event_timed": int(n_prof)}
    run.model.field.flush_deferred_update()   # N > 1: the last step's field collective + optimiser step (training.DEFER_FIELD_UPDATE)
    t_enqueued = time.perf_counter() - t0     # host side done (launches queued); the GPU is still working
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    recs = L.profile_collect()
    L.profile_enable(False)
    run.collectives = None
    if dist_on:
        # per collective of the exchange: how often, how many bytes, milliseconds from issue to "the waiting stream may go on"
        # (wire time + what the collective itself had to wait for + stream handshakes) — so that a first multi-GPU line says
        # where its step time goes
        log, T.COLLECTIVE_LOG = T.COLLECTIVE_LOG or [], None
        by = {}
        for label, nbytes, ms in T.collective_report(log):
            by.setdefault(label, []).append((nbytes, ms))
        run.collectives = {label: {"per_step": round(len(v) / steps, 3), "bytes": int(np.median([b for b, _ in v])),
                                   "issue_to_done_ms_median": round(float(np.median([m for _, m in v])), 4),
                                   "issue_to_done_ms_max": round(float(np.max([m for _, m in v])), 4),
                                   "effective_GBps_median": round(float(np.median([b / max(m, 1e-6) / 1e6 for b, m in v])), 2)}
                           for label, v in sorted(by.items())}
    if dist_on:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, t_enqueued, recs, n_prof, last


def rooflines_of(run, recs, steps, dist_on, pmc, field_records_per_launch=None):
    # steps = the number of PROFILED steps the records come from
    """`roofline` / `roofline_other_bound` objects of a timed window (see pick_rooflines)."""
    M = run.M
    alg = alg_table(M["mlp_flop"])
    table = run.model.field.mlp_base_grid.hash_table

    def entry(key):
        if key is None:
            return None
        op, units = key
        sel = [ms for o, u, ms in recs if o == op and u == units]
        fixed, note = 0.0, None
        if op == "hash_encode_bwd" and not dist_on and units == run.rays * M["samples"][2]:
            # single process: this entry point also takes the main table's optimiser step (fnr_hash_encode_bwd_adam)
            fixed = FUSED_ADAM_BYTES_PER_PARAM * float(table.numel())
            note = (f"main hash table's {M['algorithm']} step fused into this launch: {FUSED_ADAM_BYTES_PER_PARAM:.0f} B x "
                    f"{table.numel()} table parameters (p, m, v read + written; the gradient never reaches HBM)")
        e = roofline_entry(op, units, float(np.mean(sel)), len(sel), alg, fixed, note, method=run.method, pmc=pmc)
        if op == "hash_encode_bwd" and field_records_per_launch and units == run.rays * M["samples"][2]:
            # design overhead, NOT algorithmic work: every record is written by the emit kernel and read back by the
            # accumulate kernel.  With it the counter traffic decomposes: optimiser sweep + record round trip + d_feats
            rq = 2.0 * SCATTER_RECORD_BYTES * field_records_per_launch
            e["records_per_launch"] = int(field_records_per_launch)
            e["record_queue_bytes"] = rq
            e["hbm_bytes_expected"] = {"optimiser_sweep": fixed, "record_queue_write_plus_read": rq,
                                       "d_feats_read": 128.0 * units,
                                       "sum": fixed + rq + 128.0 * units}
            if e.get("traffic"):
                e["traffic_over_expected"] = round(e["traffic"] / (fixed + rq + 128.0 * units), 3)
        e["ms_per_step_in_window"] = round(float(np.sum(sel)) / steps, 5)
        e["selection"] = (f"largest total time among the HBM- / MFMA-bound entry points over the {steps} event-timed "
                          f"steps (every {PROFILE_EVERY}th) of the timed window")
        return e

    first, other = pick_rooflines(recs, alg, steps)
    return entry(first), entry(other)


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks under torch.distributed.run on this node (one
    process per GPU, rendezvous on 127.0.0.1) and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # RCCL across processes needs dmabuf IPC on these hosts
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cpus() // n)))
    env.setdefault("GPU_MAX_HW_QUEUES", "8")                 # see the top of this file
    return subprocess.call(cmd, env=env)


def _group_report(dev) -> dict:
    """{"world_size", "backend", "devices"} of the default process group: devices = distinct (host, device index) pairs the
    ranks report (2 ranks on one device in the one-GPU self-test; N on a real node).  A collective: every rank calls it."""
    import socket
    import torch.distributed as dist
    world = dist.get_world_size()
    mine = f"{socket.gethostname()}:{dev.index if dev.index is not None else 0}"
    seen = [None] * world
    dist.all_gather_object(seen, mine)
    return {"world_size": world, "backend": str(dist.get_backend()), "devices": len(set(seen))}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--quality-steps", type=int, default=-1,
                    help="training steps before the held-out PSNR / IoU / fruit-count gate; -1 = up to the reference's "
                         "max_num_iterations (30 000 for fruit_nerf, fruit_nerf_config.py:32; 3 000 for the bigger methods)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-quality", action="store_true")
    ap.add_argument("--no-big", action="store_true", help="skip the short fruit_nerf_big run under secondary")
    ap.add_argument("--image-size", type=int, default=800)
    ap.add_argument("--method", default="fruit_nerf", choices=sorted(METHODS),
                    help="reference method configuration (fruit_nerf_config.py); the headline metric is fruit_nerf")
    ap.add_argument("--cpu-steps", type=int, default=10, help="timed CPU-baseline steps (BASELINE.md 2: >= 10)")
    ap.add_argument("--cpu-warmup", type=int, default=3, help="untimed CPU-baseline steps (BASELINE.md 2: 3)")
    ap.add_argument("--cpu-rays", type=int, default=0, help="rays per CPU-baseline step (0 = the method's batch size "
                    "for fruit_nerf, 1024 for fruit_nerf_big)")
    ap.add_argument("--export-n", type=int, default=256, help="lattice side of the volume-export secondary metric")
    ap.add_argument("--mlp-precision", default="auto", choices=["auto", "fp32", "bf16x3", "bf16"],
                    help="arithmetic of the field-MLP GEMMs (include/fruitnerf_hip.h FNR_MLP_*): auto = bf16x3, the exact "
                         "3-way bf16 split on the bf16 matrix pipe (fp32-grade, parity-tested; FruitField's default); "
                         "fp32 = fp32 MFMA chains; bf16 = plain bf16 operands (BASELINE config 2; not parity grade)")
    ap.add_argument("--camera-optimizer", default="SO3xR3", choices=["off", "SO3xR3"],
                    help="the method's datamanager default (fruit_nerf_config.py:39-43): pose corrections learned from "
                         "the ray gradients; 'off' skips the input gradient of the hash grids")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))             # N ranks, each re-enters main() with RANK / WORLD_SIZE set
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but {world} rank(s) are running (WORLD_SIZE={os.environ.get('WORLD_SIZE')})")
    # host thread pools sized by the container's CPU quota (shared by the ranks of a node), not by os.cpu_count():
    # see usable_cpus() — an oversized OpenMP pool gets the whole process frozen by the CFS bandwidth controller
    torch.set_num_threads(max(1, usable_cpus() // max(world, 1)))
    # FNR_BENCH_BACKEND=gloo + FNR_BENCH_ONE_DEVICE=1: self-test of the multi-rank control flow on a 1-GPU box
    # (RCCL refuses two ranks on one device); the driver's runs use the defaults (nccl = RCCL, one GPU per rank)
    # FNR_BENCH_FORCE_DIST=1 (single process): a ONE-rank nccl group with the exchange forced on, so that this file's
    # own multi-rank code (init, barriers, MAX-reduce of the timing, teardown) and the bucketed exchange run over real
    # RCCL on a 1-GPU box; the all-reduces move no bytes, the number measures the exchange path's fixed cost
    dist_on = world > 1 or os.environ.get("FNR_BENCH_FORCE_DIST") == "1"
    backend = os.environ.get("FNR_BENCH_BACKEND", "nccl")
    if os.environ.get("FNR_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {args.gpus}: only {torch.cuda.device_count()} HIP device(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if os.environ.get("FNR_BENCH_MAIN_PRIORITY") == "high":
        # A/B knob (round 5): the LAUNCH stream is a high-priority stream instead of the default one, the training loop's
        # second stream stays normal — when both hardware queues have workgroups ready the dispatcher takes the launch
        # stream's first (the step's critical chain); same launches, same results
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    group_report = _group_report(dev) if dist_on else None

    from fruitnerf_amd import _lib as L
    from fruitnerf_amd.data import synthetic_apple as sa
    from fruitnerf_amd.fruit_nerf import FruitModel
    from fruitnerf_amd.data.semantics import apple_metadata
    from fruitnerf_amd.rays import RayBundle

    import fruitnerf_amd.training as _training
    if dist_on and world == 1:
        _training.EXCHANGE_MIN_WORLD = 1
    if dist_on and os.environ.get("FNR_DEFER_FIELD_UPDATE") != "0":
        # the field's collective keeps running underneath the next step's sampling and proposal passes; its wait and the
        # table's optimiser step sit just before that step's field encode (same results, tests/test_gpu_distributed.py)
        _training.DEFER_FIELD_UPDATE = True
    info = L.device_check()
    pmc = load_pmc_traffic()
    HW = args.image_size
    focal = 1111.0 * HW / 800.0

    # ---- synthetic dataset (identical on every rank), resident in HBM ------------------------------------
    t_setup = time.time()
    scene = sa.make_scene(seed=0, device=dev)
    c2w = sa.make_cameras(N_CAMERAS, seed=0, device=dev)
    data = sa.render_dataset(scene, c2w, H=HW, W=HW, fx=focal, fy=focal)
    i_train, i_eval = split_indices(N_CAMERAS, TRAIN_SPLIT)
    train_ids = torch.as_tensor(i_train, device=dev)
    M = METHODS[args.method]
    RAYS_PER_BATCH = M["rays"]
    ALG = alg_table(M["mlp_flop"])
    if args.mlp_precision == "auto":
        args.mlp_precision = "bf16x3"
    # entry points that run on the bf16 pipe (their roofline peak is the dense bf16 MFMA peak) and the bf16 piece
    # products they issue per algorithmic fp32 product (every method: forward and backward of all three MLPs).
    if args.mlp_precision == "bf16x3":
        ISSUED_BF16.update({"field_mlp_bwd": 6.0, "field_mlp_fwd": 6.0})   # bwd: (6 recompute + 3 dX + 3 dW) per (dX + dW)
    elif args.mlp_precision == "bf16":
        ISSUED_BF16.update({"field_mlp_bwd": 1.5, "field_mlp_fwd": 1.0})
    run = MethodRun(args.method, args.mlp_precision, args.camera_optimizer, dev, rank, world, data, train_ids, len(i_train))
    model, opt, camera, batcher, model_cfg = run.model, run.opt, run.camera, run.batcher, run.model_cfg
    n_params = model.arena().numel
    torch.cuda.synchronize()
    setup_s = time.time() - t_setup
    one_step = run.one_step

    def barrier():
        if dist_on:
            import torch.distributed as dist
            dist.barrier()

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()

    # ---- timed region: exactly K steps -------------------------------------------------------------------
    L.scatter_records(reset=True)
    allocs_before = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)     # hipMalloc calls of the caching allocator
    dt, t_enqueued, recs, n_prof, (ld, md) = timed_window(run, args.steps, barrier, dist_on, dev)
    headline_host_ms = dict(run.host_ms)
    headline_collectives = run.collectives
    headline_sequencer = dict(run.steps.stats, enabled=bool(_training.NATIVE_SEQUENCER and not dist_on))
    device_allocs_in_window = int(torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - allocs_before)
    rays_per_s = world * args.steps * RAYS_PER_BATCH / dt
    field_records = L.scatter_records()[0] / max(args.steps, 1)     # one main-field scatter per step

    if rank != 0:
        if dist_on:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant entry point (and of the dominant one bound by the other roofline) --------------------
    roofline, roofline_other = rooflines_of(run, recs, n_prof, dist_on, pmc, field_records)
    # the SAME window (fresh model, same seeds, warm-up, steps and event-timed steps) in the strictly-fp32 arithmetic
    # (v_mfma_f32_16x16x4_f32 chains, forward and backward): the reference's own arithmetic, next to the headline's
    value_fp32 = None
    if world == 1 and args.mlp_precision != "fp32" and not dist_on:
        r32 = MethodRun(args.method, "fp32", args.camera_optimizer, dev, rank, world, data, train_ids, len(i_train))
        for _ in range(args.warmup):
            r32.one_step()
        torch.cuda.synchronize()
        dt32 = timed_window(r32, args.steps, barrier, dist_on, dev)[0]
        value_fp32 = round(args.steps * RAYS_PER_BATCH / dt32, 1)
        del r32
        torch.cuda.empty_cache()
    # whole-step fractions against both rooflines (SURVEY §8d per-ray figures): never "the path is MFMA-bound"
    whole_step = {"mfma_f32_frac": round(M["flop_per_ray_train"] * rays_per_s / world / (MFMA_F32_PEAK_TF * 1e12), 4),
                  "hbm_frac": round(M["bytes_per_ray_train"] * rays_per_s / world / (HBM_PEAK_GBS * 1e9), 4),
                  "alg_flop_per_ray": M["flop_per_ray_train"], "alg_table_bytes_per_ray": M["bytes_per_ray_train"],
                  "note": "per GPU; algorithmic train FLOP / table bytes per ray (BASELINE.md §3) x rays/s vs fp32-MFMA "
                          "157.3 TFLOP/s and HBM 8 TB/s"}

    # ---- per-entry-point breakdown (short instrumented pass, outside the timed region; N = 1 only: the
    # other ranks have left, so no collective may run here) ------------------------------------------------
    nb = 12 if world == 1 else 0
    L.profile_enable(True)
    saved_serialize, _training.SERIALIZE_STREAMS = _training.SERIALIZE_STREAMS, True   # nothing next to a bracketed launch
    for _ in range(nb):
        one_step()
    torch.cuda.synchronize()
    _training.SERIALIZE_STREAMS = saved_serialize
    recs = L.profile_collect() if nb else []
    L.profile_enable(False)
    breakdown = {}
    for op, units, ms in recs:
        key = f"{op}[{units}]"
        breakdown[key] = breakdown.get(key, 0.0) + ms / nb
    breakdown = {k: round(v, 4) for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1])}

    # ---- quality gate: keep training to the reference's iteration count, then PSNR / IoU on held-out views -------------
    def heldout_quality(model=model):
        model.eval()
        psnrs, inter, union = [], 0.0, 0.0
        g = torch.Generator(device=dev)
        g.manual_seed(7)
        with torch.no_grad():
            for img in i_eval[:5]:
                n = 65536
                y = torch.randint(0, HW, (n,), device=dev, generator=g)
                x = torch.randint(0, HW, (n,), device=dev, generator=g)
                ci = torch.full((n,), int(img), device=dev)
                o, d = sa.pixel_rays(c2w, ci, y, x, focal, focal, HW / 2.0, HW / 2.0)
                tgt = data["images"][ci, y, x].float() / 255.0
                msk = data["masks"][ci, y, x].float()
                for s in range(0, n, 32768):  # eval_num_rays_per_chunk
                    out = model(RayBundle(o[s:s + 32768], d[s:s + 32768], None, None))
                    mse = torch.mean((out["rgb"] - tgt[s:s + 32768]) ** 2)
                    psnrs.append(float(-10.0 * torch.log10(mse)))
                    pred = (torch.sigmoid(out["semantics"][:, 0]) > 0.5).float()
                    inter += float((pred * msk[s:s + 32768]).sum())
                    union += float(((pred + msk[s:s + 32768]) > 0).float().sum())
        model.train()
        return round(float(np.mean(psnrs)), 3), round(inter / max(union, 1.0), 4)

    quality = None
    if not args.no_quality and world == 1:
        # the reference trains fruit_nerf for 30 000 iterations (fruit_nerf_config.py:32); the bigger methods (100 000
        # iterations there) get a bounded 3 000-step look
        target = run.step_idx + args.quality_steps if args.quality_steps >= 0 else \
            (30000 if args.method == "fruit_nerf" else 3000)
        marks = sorted({m for m in (2000, 10000) if run.step_idx < m < target} | {target})
        trajectory = []
        torch.cuda.synchronize()
        t_q, s_q = time.perf_counter(), run.step_idx
        for mark in marks:
            while run.step_idx < mark:
                ld, md = one_step(want_metrics=False)
            psnr, iou = heldout_quality()
            trajectory.append({"train_steps": run.step_idx, "psnr_heldout": psnr, "semantic_iou_heldout": iou})
        torch.cuda.synchronize()
        t_q = time.perf_counter() - t_q
        quality = {"train_steps": run.step_idx, "psnr_heldout": trajectory[-1]["psnr_heldout"],
                   "semantic_iou_heldout": trajectory[-1]["semantic_iou_heldout"],
                   "reference_max_num_iterations": 30000 if args.method == "fruit_nerf" else 100000,
                   "trajectory": trajectory,
                   "train_rays_per_s_over_these_steps": round((run.step_idx - s_q) * RAYS_PER_BATCH / max(t_q, 1e-9), 1),
                   "heldout": "5 held-out views x 65 536 random pixels, eval mode; IoU of sigmoid(semantics) > 0.5 vs mask",
                   "final_train_losses": {k: round(float(v), 6) for k, v in ld.items()},
                   # records of the binned scatter that overflowed a queue and went through float atomics (their order,
                   # hence the last bits, would depend on timing) over everything this process has run so far
                   "scatter_queue_overflows": L.scatter_overflows(),
                   # integer sum of the bit patterns of every parameter after these steps: two runs of this command on
                   # any box must print the same number (training is deterministic: fixed-point scatter sums, ordered
                   # reductions, counter-based random numbers) — the visible form of the reproducibility claim, and the
                   # quickest detector of a defect like round 4's (one differing step changes it)
                   "parameter_checksum": int(model.arena().params.view(torch.int32).sum(dtype=torch.int64))}

    gate_failures = []
    if quality is not None:
        # (a) the OTHER seed stream of the batcher to the same step count: the held-out IoU of a single trajectory swings by
        # +-0.01 within 50 steps at these learning rates (the CPU oracle's does too), so one seed's
        # 30 k number is a sample; two are printed
        if args.method == "fruit_nerf" and args.quality_steps < 0:
            r_b = MethodRun(args.method, args.mlp_precision, args.camera_optimizer, dev, rank, world, data, train_ids,
                            len(i_train), batch_seed=4321)
            torch.cuda.synchronize()
            t_b = time.perf_counter()
            traj_b = []
            for mark in marks:
                while r_b.step_idx < mark:
                    r_b.one_step(want_metrics=False)
                p_b, i_b = heldout_quality(r_b.model)
                traj_b.append({"train_steps": r_b.step_idx, "psnr_heldout": p_b, "semantic_iou_heldout": i_b})
            torch.cuda.synchronize()
            quality["second_seed_stream"] = {
                "batch_seed": 4321, "psnr_heldout": traj_b[-1]["psnr_heldout"],
                "semantic_iou_heldout": traj_b[-1]["semantic_iou_heldout"], "trajectory": traj_b,
                "train_rays_per_s": round(r_b.step_idx * RAYS_PER_BATCH / (time.perf_counter() - t_b), 1),
                "parameter_checksum": int(r_b.model.arena().params.view(torch.int32).sum(dtype=torch.int64))}
            del r_b
            torch.cuda.empty_cache()
        # (b) gates: a scatter record that overflowed its queue went through float atomics (timing-dependent bits), and the
        # parameters after the run must be the pinned ones for this arithmetic generation (profiles/parameter_checksums.json)
        pin_key = f"{args.method}:{args.mlp_precision}:{args.camera_optimizer}:{quality['train_steps']}:{HW}:{NUMERICS}"
        pinned = None
        if os.path.exists(CHECKSUMS_PATH):
            pinned = json.load(open(CHECKSUMS_PATH)).get(pin_key)
        quality["parameter_checksum_pinned"] = pinned
        quality["parameter_checksum_key"] = pin_key
        if quality["scatter_queue_overflows"] > 0:
            gate_failures.append(f"scatter_queue_overflows = {quality['scatter_queue_overflows']} (must be 0)")
        if pinned is not None and int(pinned) != quality["parameter_checksum"]:
            gate_failures.append(f"parameter_checksum {quality['parameter_checksum']} != pinned {pinned} for {pin_key}")
        second = quality.get("second_seed_stream")
        if second is not None and os.path.exists(CHECKSUMS_PATH):
            pinned2 = json.load(open(CHECKSUMS_PATH)).get(pin_key + ":batch_seed_4321")
            second["parameter_checksum_pinned"] = pinned2
            if pinned2 is not None and int(pinned2) != second["parameter_checksum"]:
                gate_failures.append(f"second seed stream: parameter_checksum {second['parameter_checksum']} != pinned {pinned2}")

    # ---- secondary metrics: full-image eval rays/s and volume-export samples/s, trained weights ----
    secondary = None
    if not args.no_quality and world == 1:
        import copy
        from fruitnerf_amd.data.fruit_datamanager import ExportDataManager
        from fruitnerf_amd.export.exporter_utils import sample_volume
        model.eval()
        with torch.no_grad():
            ys, xs = torch.meshgrid(torch.arange(HW, device=dev), torch.arange(HW, device=dev), indexing="ij")
            ci = torch.full((HW * HW,), int(i_eval[0]), device=dev)
            o, d = sa.pixel_rays(c2w, ci, ys.reshape(-1), xs.reshape(-1), focal, focal, HW / 2.0, HW / 2.0)
            cam_rb = RayBundle(o.view(HW, HW, 3), d.view(HW, HW, 3), None, None)
            model.get_outputs_for_camera_ray_bundle(cam_rb)            # warm-up
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            model.get_outputs_for_camera_ray_bundle(cam_rb)            # 32768-ray chunks, outputs stay on the device
            torch.cuda.synchronize()
            eval_s = time.perf_counter() - t1
        # volume export (ns-export-semantics): N^3 lattice through the trained field, three thresholded point sets
        emodel = FruitModel(copy.deepcopy(model.config), apple_metadata(), num_train_data=len(i_train), device=dev, test_mode="export")
        emodel.load_state_dict(model.state_dict(), strict=True)
        emodel.eval()
        N_EXP = args.export_n

        class _Pipe:
            pass

        pipe = _Pipe()
        pipe.model = emodel
        pipe.datamanager = ExportDataManager(dev, eval_num_rays_per_batch=32768)
        emodel.setup_inference(True, N_EXP, deterministic=True)   # bin centres: reproducible counts
        exp_times = []
        # 1 untimed pass (allocator warm-up: the per-batch feature buffer is ~1 GB at 256^3), then 5 timed: MEDIAN
        for i in range(6):
            n_rays = pipe.datamanager.setup_inference(aabb=((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)), num_points=N_EXP)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            sets = sample_volume(pipe, n_rays, transform_json={"scale": 1.0})
            torch.cuda.synchronize()
            if i > 0:
                exp_times.append(time.perf_counter() - t1)
        exp_s = float(np.median(exp_times))
        cam_off = None
        if run.camera is not None:  # the same loop without the camera optimiser (no input gradient of the hash grids)
            saved = run.camera
            run.set_camera(None)
            for _ in range(10):
                one_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(100):
                one_step()
            torch.cuda.synchronize()
            cam_off = round(100 * RAYS_PER_BATCH / (time.perf_counter() - t1), 1)
            run.set_camera(saved)
        # The headline window again on a FRESH model (same seeds, same step numbers -> same proposal update schedule)
        # with everything on one stream (FNR_OVERLAP_PROPOSAL_BACKWARD=0) and with the second HIP stream on every step
        # (the timed region itself keeps the steps whose launches it brackets with events on one stream).
        import fruitnerf_amd.training as _T

        def headline_window(overlap: bool):
            r2 = MethodRun(args.method, args.mlp_precision, args.camera_optimizer, dev, rank, world, data, train_ids,
                           len(i_train))
            saved_flag, _T.OVERLAP_PROPOSAL_BACKWARD = _T.OVERLAP_PROPOSAL_BACKWARD, overlap
            try:
                for _ in range(args.warmup):
                    r2.one_step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    r2.one_step()
                torch.cuda.synchronize()
                return round(args.steps * RAYS_PER_BATCH / (time.perf_counter() - t1), 1)
            finally:
                _T.OVERLAP_PROPOSAL_BACKWARD = saved_flag
        overlap_modes = {"note": "the headline window (same seeds and step numbers) on a fresh model, no event-timed steps; "
                                 "'second_stream': proposal-network backward, ray-gradient reduction, camera step and the next "
                                 "step's sampling on the second HIP stream (the default); 'one_stream': everything in-stream",
                         "one_stream": headline_window(False), "second_stream": headline_window(True)}
        # The path an UNMODIFIED Nerfstudio Trainer drives (fruit_pipeline.py:120-146 + Trainer.train_iteration): callbacks
        # by location, model(ray_bundle) -> get_metrics_dict -> get_loss_dict -> reduce(add) -> backward() through the
        # autograd Functions -> optimizer.step() -> schedulers — the plugin API only, no TrainingSteps, no look-ahead, no
        # fused optimiser steps inside the backward kernels.  Once with the library's optimiser (FusedAdam.step: one launch
        # over the arena) and once with torch.optim.Adam on get_param_groups() + nerfstudio's per-group schedulers, i.e.
        # with nothing of this repository outside the model.  Rays come from the batcher's pixel-sampling launch with the
        # cameras as given (nerfstudio's camera optimiser is autograd over torch ops and lives in the datamanager).
        def plugin_api_window(kind: str, n_warm: int = 10, n_steps: int = 100):
            import functools
            from fruitnerf_amd.engine.callbacks import TrainingCallbackAttributes, TrainingCallbackLocation as Loc
            from fruitnerf_amd.training import FusedAdam, exponential_decay_lr, skipped_groups
            r3 = MethodRun(args.method, args.mlp_precision, "off", dev, rank, world, data, train_ids, len(i_train))
            m3 = r3.model
            cbs = m3.get_training_callbacks(TrainingCallbackAttributes(optimizers=None, grad_scaler=None, pipeline=None))
            groups = m3.get_param_groups()
            if kind == "fused_adam":
                opt3 = r3.opt
            else:
                m3.arena()                     # parameters re-homed into the arena before torch.optim captures them
                Opt = torch.optim.Adam if M["algorithm"] == "adam" else torch.optim.RAdam
                opt3 = {g: Opt(groups[g], lr=M["groups"][g]["lr"], eps=1e-15) for g in groups}
            t1 = 0.0
            for step in range(n_warm + n_steps):
                if step == n_warm:
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                for cb in cbs:
                    cb.run_callback_at_location(step, location=Loc.BEFORE_TRAIN_ITERATION)
                o, d, cam, batch = r3.batcher.sample(r3.rays)
                if kind != "fused_adam":
                    for op_ in opt3.values():
                        op_.zero_grad(set_to_none=False)       # gradients are views of the arena: zeroed in place
                outputs = m3(RayBundle(o, d, None, cam))
                metrics_dict = m3.get_metrics_dict(outputs, batch)
                loss_dict = m3.get_loss_dict(outputs, batch, metrics_dict)
                functools.reduce(torch.add, loss_dict.values()).backward()
                skip = skipped_groups(m3)
                if kind == "fused_adam":
                    opt3.step(skip=skip)
                else:
                    for g, op_ in opt3.items():
                        if g not in skip:                      # (torch.optim skips parameters whose .grad is None)
                            op_.step()
                        gl = M["groups"][g]
                        for pg in op_.param_groups:            # ExponentialDecayScheduler, stepped every iteration
                            pg["lr"] = exponential_decay_lr(step + 1, gl["lr"], gl["lr_final"], gl["max_steps"]) \
                                if gl.get("lr_final") is not None else gl["lr"]
                for cb in cbs:
                    cb.run_callback_at_location(step, location=Loc.AFTER_TRAIN_ITERATION)
            torch.cuda.synchronize()
            v = round(n_steps * RAYS_PER_BATCH / (time.perf_counter() - t1), 1)
            del r3, m3, opt3
            torch.cuda.empty_cache()
            return v
        plugin_api = {"note": "train rays/s of the loop an unmodified Nerfstudio Trainer runs over the plugin surface "
                              "(callbacks, model(ray_bundle), get_metrics_dict, get_loss_dict, backward() through the autograd "
                              "Functions, optimizer.step()); fresh model, steps 10..110, camera optimiser off; the headline "
                              "loop (training.TrainingSteps) fuses the optimiser steps into the backward kernels and samples ahead",
                      "fused_adam": plugin_api_window("fused_adam"),
                      "torch_optim_adam": plugin_api_window("torch_optim")}
        # the other arithmetic modes of the field MLPs on the SAME loop (headline mode restored afterwards): bf16x3 is
        # parity grade (tests/test_gpu_bf16.py), bf16 is BASELINE config 2's throughput mode
        mlp_modes = None
        if True:
            mlp_modes = {"note": "train rays/s, 100 steps each after 10 untimed, same model / loop as the headline, "
                                 f"measured after step {run.step_idx}"}
            for mode in (("fp32", "bf16x3", "bf16") if args.method == "fruit_nerf" else ("fp32", "bf16x3")):
                model.field.mlp_precision = mode
                for _ in range(10):
                    one_step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(100):
                    one_step()
                torch.cuda.synchronize()
                mlp_modes[mode] = round(100 * RAYS_PER_BATCH / (time.perf_counter() - t1), 1)
            model.field.mlp_precision = args.mlp_precision
        # counting stage front-end on the exported semantic set: radius-outlier removal -> voxel
        # down-sampling -> DBSCAN -> centre-distance merge, all three library calls on the GPU; the scene has 32 fruits
        from fruitnerf_amd.clustering import FruitClustering, PointCloud
        pts = sets["semantic"]["points"]
        spacing = 2.0 / N_EXP * 2.0                     # lattice pitch after sample_volume's x2 scaling
        fruit_count = None
        if pts.shape[0] >= 5:
            fc = FruitClustering(voxel_size_down_sample=spacing / 4, remove_outliers_nb_points=2,
                                 remove_outliers_radius=1.8 * spacing, cluster_merge_distance=0.04)
            fruit_count = fc.first_stage_count(PointCloud(pts, None, dev), eps=1.8 * spacing, min_samples=4)
        # the END-TO-END count (BASELINE config 5, clustering_base.py:513-538): export -> cluster -> merge_small_clusters
        # -> split_large_cluster on a 512^3 lattice (the second stage's alpha = 100 surface needs points closer than
        # ~0.01: the reference's clouds sit on a 1 mm voxel grid, a 256^3 lattice has a pitch of 0.0156)
        full_count = count_fruits_end_to_end(emodel, pipe, sample_volume, scene, dev, n_side=512)
        counting = counting_stage_bench(dev, cpu=not args.no_cpu_baseline)
        secondary = {"train_rays_per_s_by_proposal_backward_stream": overlap_modes,
                     "train_rays_per_s_plugin_api": plugin_api,
                     "train_rays_per_s_by_mlp_precision": mlp_modes,
                     "train_rays_per_s_camera_optimizer_off": cam_off,
                     "train_rays_per_s_camera_optimizer_off_note": f"100 steps measured after step {run.step_idx - 100} "
                     "(proposal nets are updated less often by then than in the headline window)",
                     "eval_rays_per_s": round(HW * HW / eval_s, 1), "eval_image": f"{HW}x{HW}, chunks of 32768 rays",
                     "export_samples_per_s": round(n_rays * N_EXP / exp_s, 1), "export_lattice": f"{N_EXP}^3",
                     "export_pass_ms": [round(t * 1e3, 1) for t in exp_times],
                     "export_timing": "median of 5 passes after 1 untimed pass (whole sample_volume call incl. the "
                                      "per-batch D2H of the point lists)",
                     "export_points": {k: int(v["points"].shape[0]) for k, v in sets.items()},
                     "fruit_count_first_stage_on_semantic_export": fruit_count, "fruit_count_scene": scene.n_fruits,
                     "fruit_count_end_to_end": full_count, "counting_front_end": counting}
        if quality is not None:   # the north star's count gate: the counting stage's result on the exported semantic set
            quality["fruit_count"] = None if full_count is None else full_count["count"]
            if full_count is not None:     # right total for the right reasons?  matched against the scene's fruit centres
                quality["fruit_count_precision_recall_F1"] = [full_count["precision"], full_count["recall"], full_count["F1"]]
            quality["fruit_count_first_stage"] = fruit_count
            quality["fruit_count_scene"] = scene.n_fruits
        model.train()

    # ---- fruit_nerf_big (BASELINE configs 3, 5) on the same scene: a short window of the bigger method, so that its
    # throughput and both roofline fractions are in the default line (8192 rays, 512/256/128 samples, T = 2^21, RAdam)
    big = None
    if args.method == "fruit_nerf" and not args.no_big and not args.no_quality and world == 1:
        del emodel, pipe, sets
        torch.cuda.empty_cache()
        rb_ = MethodRun("fruit_nerf_big", args.mlp_precision, args.camera_optimizer, dev, rank, world, data, train_ids,
                        len(i_train))
        for _ in range(5):
            rb_.one_step()
        torch.cuda.synchronize()
        nb_steps = 20
        L.scatter_records(reset=True)
        dt_b, enq_b, recs_b, n_prof_b, _ = timed_window(rb_, nb_steps, barrier, dist_on, dev)
        r_b, r_b2 = rooflines_of(rb_, recs_b, n_prof_b, dist_on, pmc, L.scatter_records()[0] / nb_steps)
        Mb = METHODS["fruit_nerf_big"]
        v_b = nb_steps * Mb["rays"] / dt_b
        big = {"metric": f"train rays/sec, fruit_nerf_big on synthetic apple {HW}x{HW}", "value": round(v_b, 1),
               "unit": "rays/s", "steps": nb_steps, "warmup": 5, "ms_per_step": round(dt_b / nb_steps * 1e3, 4),
               "host_enqueue_ms_per_step": round(enq_b / nb_steps * 1e3, 4),
               "config": f"{Mb['rays']} rays/step, samples {'/'.join(map(str, Mb['samples']))}, hash 16x2^21x2, geo 30, "
                         f"semantic MLP 3x128, fwd+bwd+radam over {rb_.model.arena().numel / 1e6:.1f} M parameters, "
                         f"mlp_precision {args.mlp_precision}, camera optimizer {args.camera_optimizer}",
               "roofline": r_b, "roofline_other_bound": r_b2,
               "whole_step": {"mfma_f32_frac": round(Mb["flop_per_ray_train"] * v_b / (MFMA_F32_PEAK_TF * 1e12), 4),
                              "hbm_frac": round(Mb["bytes_per_ray_train"] * v_b / (HBM_PEAK_GBS * 1e9), 4)}}
        # ... and its convergence: 20 000 of the method's 100 000 iterations (fruit_nerf_config.py:68), held-out PSNR / IoU at
        # 2 000 / 10 000 / 20 000 (no count: the count gate is the headline method's)
        if os.environ.get("FNR_BENCH_BIG_QUALITY", "1") != "0":
            torch.cuda.synchronize()
            t_b = time.perf_counter()
            s_b = rb_.step_idx
            traj = []
            for mark in (2000, 10000, 20000):
                while rb_.step_idx < mark:
                    rb_.one_step(want_metrics=False)
                p_b, i_b = heldout_quality(rb_.model)
                traj.append({"train_steps": rb_.step_idx, "psnr_heldout": p_b, "semantic_iou_heldout": i_b})
            torch.cuda.synchronize()
            big["quality"] = {"train_steps": rb_.step_idx, "reference_max_num_iterations": 100000, "trajectory": traj,
                              "psnr_heldout": traj[-1]["psnr_heldout"], "semantic_iou_heldout": traj[-1]["semantic_iou_heldout"],
                              "train_rays_per_s_over_these_steps": round((rb_.step_idx - s_b) * Mb["rays"] / (time.perf_counter() - t_b), 1),
                              "scatter_queue_overflows": L.scatter_overflows(),
                              "parameter_checksum": int(rb_.model.arena().params.view(torch.int32).sum(dtype=torch.int64))}
            if big["quality"]["scatter_queue_overflows"] > 0:
                gate_failures.append(f"fruit_nerf_big: scatter_queue_overflows = {big['quality']['scatter_queue_overflows']}")
            big_key = f"fruit_nerf_big:{args.mlp_precision}:{args.camera_optimizer}:{rb_.step_idx}:{HW}:{NUMERICS}"
            pinned_b = json.load(open(CHECKSUMS_PATH)).get(big_key) if os.path.exists(CHECKSUMS_PATH) else None
            big["quality"]["parameter_checksum_pinned"], big["quality"]["parameter_checksum_key"] = pinned_b, big_key
            if pinned_b is not None and int(pinned_b) != big["quality"]["parameter_checksum"]:
                gate_failures.append(f"fruit_nerf_big: parameter_checksum {big['quality']['parameter_checksum']} != pinned {pinned_b}")
        del rb_
        torch.cuda.empty_cache()
        if secondary is not None:
            secondary["fruit_nerf_big"] = big

    # ---- CPU baseline: the oracle's training step on the host cores -------------------------------------------------
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        from oracle import fruit_oracle as fo
        from oracle import ns_torch as ns
        # BASELINE.md §2 protocol: the method's batch size, oracle/ PyTorch-CPU fp32, median of the timed steps.
        # Threads: what the container's CPU quota allows (usable_cpus(): 16 on the MI355X boxes; more threads than
        # that get the process throttled — 124 s per 4096-ray step with 256 threads), at most 32 (eager PyTorch on
        # this path is a stream of small ops and does not scale further) — and says so in `cores`.
        # fruit_nerf_big: a bounded 1024-ray sample of the 8192-ray batch (a full batch takes minutes per step).
        ncores = min(usable_cpus(), 32)
        CPU_RAYS = args.cpu_rays or (RAYS_PER_BATCH if args.method == "fruit_nerf" else 1024)
        torch.set_num_threads(ncores)
        torch.manual_seed(0)
        ocfg = fo.FruitNerfModelConfig()
        for k, v in M["model"].items():
            setattr(ocfg, k, v)
        om = fo.FruitModel(ocfg, num_train_data=len(i_train))
        om.train()
        groups = om.get_param_groups()
        OptCls = torch.optim.Adam if M["algorithm"] == "adam" else torch.optim.RAdam
        oopts = [OptCls(groups["proposal_networks"], lr=1e-2, eps=1e-15), OptCls(groups["fields"], lr=1e-2, eps=1e-15)]
        cdata = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in data.items()}
        from oracle import pixel_sampler as ops          # the CPU restatement of PixelSampler + RayGenerator
        train_ids_cpu = train_ids.cpu()
        ocam = None
        if args.camera_optimizer != "off":  # same work as the GPU step: pose corrections + their Adam(weight_decay)
            from oracle import camera_opt as oc
            ocam = oc.CameraOptimizer(len(i_train))
            oopts.append(OptCls(ocam.parameters(), lr=6e-4, eps=1e-8, weight_decay=M["camera"]["weight_decay"]))
        gen_u = torch.Generator().manual_seed(99)
        times = []
        n_cpu, n_cpu_warm = max(1, args.cpu_steps), max(0, args.cpu_warmup)
        for i in range(n_cpu_warm + n_cpu):
            u = torch.rand(CPU_RAYS, 3, generator=gen_u)
            t1 = time.perf_counter()
            o, d, cam, batch = ops.sample_pixels(cdata, train_ids_cpu, u)
            if ocam is not None:
                kk = cam[:, 0]
                yy = (u[:, 1] * HW).long().clamp_max(HW - 1)
                xx = (u[:, 2] * HW).long().clamp_max(HW - 1)
                o, d = oc.generate_rays(cdata["c2w"][train_ids.cpu()[kk]], ocam(kk), yy, xx, focal, focal, HW / 2.0,
                                        HW / 2.0)
            om.set_anneal(i)
            for op_ in oopts:
                op_.zero_grad()
            out = om(ns.RayBundle(o, d, torch.ones(CPU_RAYS, 1), camera_indices=cam))
            om.get_metrics_dict(out, batch)
            sum(om.get_loss_dict(out, batch).values()).backward()
            for op_ in oopts:
                op_.step()
            om.proposal_sampler.step_cb(i)
            if i >= n_cpu_warm:  # 3 warm-up + >= 10 timed iterations, median and min
                times.append(time.perf_counter() - t1)
        med = float(np.median(times))
        cpu = {"value": round(CPU_RAYS / med, 1), "unit": "rays/s", "cores": ncores, "kind": "port",
               "sample": f"{n_cpu} full {args.method} training steps (pixel sampling + ray generation"
                         f"{' with the SO3xR3 camera optimizer' if ocam is not None else ''}, fwd+bwd+{M['algorithm']} over "
                         f"all {n_params / 1e6:.1f} M parameters) of {CPU_RAYS} rays each"
                         f"{'' if CPU_RAYS == RAYS_PER_BATCH else f' (bounded sample of the {RAYS_PER_BATCH}-ray batch)'} "
                         f"after {n_cpu_warm} warm-up, oracle/ PyTorch-CPU fp32, {ncores} host threads (CPU quota of the container: "
                         f"{usable_cpus()} of {os.cpu_count()} hardware threads), median {med:.2f} s/step, "
                         f"min {min(times):.2f} s/step"}

    result = {
        "metric": f"train rays/sec, {args.method} on synthetic apple {HW}x{HW}",
        "value": round(rays_per_s, 1),
        "unit": "rays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        # host time to enqueue a step (Python + ctypes + HIP launches), without waiting for the GPU: while it stays
        # below ms_per_step the step is GPU-bound
        "host_enqueue_ms_per_step": round(t_enqueued / args.steps * 1e3, 4),
        # ... split: the steps a recorded step program replays (training.NATIVE_SEQUENCER: one fnr_program_replay call per
        # step) and the event-timed steps of the roofline leg, which are interpreted; and what the sequencer did over the run
        "host_enqueue_ms_by_step_kind": headline_host_ms,
        "native_sequencer": headline_sequencer,
        # hipMalloc calls inside the timed window (the caching allocator growing a pool: each one stalls the host for
        # 0.1 - 1 ms); 0 once the warm-up has seen every step shape
        "device_allocs_in_window": device_allocs_in_window,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        # arithmetic type of the MLP GEMMs; hash grids, samplers, compositing, losses and the optimiser are fp32 in
        # every mode.  bf16x3 = three bf16 pieces per fp32 operand (fp32-grade results), bf16 = bf16 operands.
        "dtype": {"fp32": "f32",
                  "bf16x3": "f32 fwd (MLP GEMMs: 3-piece bf16 split, 6 products, ~2^-24) / 2-piece bwd (dX, dW: 3 products, "
                            "~2^-17; its forward recompute 3-piece); everything else f32",
                  "bf16": "bf16"}[args.mlp_precision],
        # rays/s of the same window (fresh model, same seeds / warm-up / steps) with every MLP GEMM in fp32 MFMA chains
        "value_fp32_arithmetic": value_fp32,
        "data": "synthetic",
        "config": {"workload": f"{args.method} synthetic apple {HW}x{HW}, {N_CAMERAS} cameras ({len(i_train)} train), "
                               f"{RAYS_PER_BATCH} rays/rank/step, samples {'/'.join(map(str, M['samples']))}, "
                               f"hash 16x2^{model_cfg.log2_hashmap_size}x2 + proposal grids "
                               f"{'+'.join(str(a['num_levels']) + 'x2^' + str(a['log2_hashmap_size']) + 'x2' for a in model_cfg.proposal_net_args_list)}"
                               f", geo {model_cfg.geo_feat_dim}, "
                               f"semantic MLP {model_cfg.num_layers_semantic}x{model_cfg.hidden_dim_semantics}, "
                               f"fwd+bwd+{M['algorithm']} over {n_params / 1e6:.1f} M parameters, proposal-net update "
                               f"schedule from step 0, camera optimizer {args.camera_optimizer}",
                   "method": args.method, "mlp_precision": args.mlp_precision, "rays_per_rank": RAYS_PER_BATCH, "parallelism": f"dp{world}",
                   # N=1: the main table's optimiser step runs inside the scatter's accumulate kernel (its launches are
                   # the hash_encode_bwd entry; no gradient table is written or re-read); N>1: separate step after RCCL
                   "table_optimizer": "fused into hash_encode_bwd" if not dist_on else
                   ("separate (after the exchange" + (", deferred to the next step's field encode)" if _training.DEFER_FIELD_UPDATE else ")") +
                    (f", sharded over the {world} ranks: reduce-scatter + own shard's step + all-gather" if _training.SHARDED_FIELD_OPTIMIZER else "")),
                   "streams": (f"2: proposal-network backward underneath the field backward; ray-gradient reduction, camera "
                               f"step and the next step's rays + proposal sampling underneath the table scatter (serialised "
                               f"on every {PROFILE_EVERY}th step, whose launches are timed)"
                               if _training.OVERLAP_PROPOSAL_BACKWARD else "1"),
                   # what the process group itself reports (so that a SCALE line can be checked against the launcher):
                   # ranks of the group, its backend ("nccl" is RCCL on ROCm) and the distinct devices they run on
                   "rccl_ranks": group_report,
                   "exchange": None if not dist_on else f"{_training.EXCHANGE_LEVEL_GROUPS} field collective(s) per step + proposal networks on update steps + poses",
                   # (rank 0's view) per collective: issued per step, bytes, milliseconds from issue to the consumer's go
                   "exchange_collectives": headline_collectives,
                   "device": info["arch"], "setup_s": round(setup_s, 1)},
        "roofline": roofline,
        "roofline_other_bound": roofline_other,
        "whole_step": whole_step,
        "cpu_baseline": cpu,
        "breakdown_ms": breakdown,
        "quality": quality,
        "secondary": secondary,
    }
    if cpu:
        result["speedup_vs_cpu_baseline"] = round(rays_per_s / cpu["value"], 1)
    result["gates"] = {"failed": gate_failures,
                       "checked": ["scatter_queue_overflows == 0 (fruit_nerf, fruit_nerf_big)",
                                   "parameter_checksum == the pinned one for the arithmetic generation (fruit_nerf both seed streams, "
                                   "fruit_nerf_big; profiles/parameter_checksums.json)"]}
    print(json.dumps(result))
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if gate_failures:           # the line above is complete; the exit code says that a gate it carries failed
        print("bench.py: gate failed: " + "; ".join(gate_failures), file=sys.stderr)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
