"""sample_volume — MI355X-native mirror of /root/reference/fruit_nerf/export/exporter_utils.py:47-258.

Export loop: per batch of orthographic rays run the model in 'export' mode, threshold (density >= 70,
logit >= 3, sigmoid(logit) > 0.9), gather the three point sets, rescale.  The per-sample field queries and
the order-preserving three-stream compaction run in libfruitnerf_hip.so; Open3D object creation and PLY
writing are out of scope here — the function returns the point / colour arrays the
reference would hand to Open3D (float64, same order).
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Optional

import torch

from .. import _kernels as K

SET_NAMES = ("semantic_colormap", "semantic", "density")


def _ensure_buffers(state, dev):
    cap = state["cap"]
    if state.get("buf_cap") != cap:
        state["points"] = [torch.empty(cap, 3, device=dev) for _ in range(3)]
        state["colors"] = [torch.empty(cap, 4, device=dev) for _ in range(3)]
        state["buf_cap"] = cap


def _read_counts(counts, state):
    """The three running totals on the host (pinned destination + one stream synchronize instead of a pageable
    `counts.cpu()`)."""
    dev = counts.device
    if dev.type != "cuda":                      # CPU stand-in kernels of the gloo sharding test
        return counts.tolist()
    host = state.get("counts_host")
    if host is None:
        host = state["counts_host"] = torch.empty(3, dtype=torch.int64).pin_memory()
    host.copy_(counts, non_blocking=True)
    torch.cuda.current_stream(dev).synchronize()
    return host.tolist()


def _compact_batch(lat, ray_begin, n_rays, positions, density, rgb, logit, state):
    """One batch compacted into the front of the stream buffers; returns its three counts (one host wait, like the
    reference's per-batch .cpu(), exporter_utils.py:126-153).  Re-runs the batch with larger buffers when they
    overflow."""
    dev = density.device
    while True:
        _ensure_buffers(state, dev)
        counts = torch.zeros(3, dtype=torch.int64, device=dev)
        K.export_compact(lat, ray_begin, n_rays, positions, density, rgb, logit, state["points"], state["colors"],
                         counts)
        c = _read_counts(counts, state)
        if max(c) <= state["cap"]:
            return c
        while state["cap"] < max(c):
            state["cap"] *= 2


def sample_volume(pipeline, num_points: int, output_dir: Optional[Path] = None, config=None,
                  transform_json: Optional[dict] = None, rank: int = 0, world_size: int = 1) -> Dict[str, dict]:
    """`pipeline` needs `.model` (FruitModel in test_mode='export' after setup_inference) and
    `.datamanager` (setup_inference done; `next_sample_volume`).  `num_points` = number of rays, as in the
    reference (exporter_utils.py:94,172).

    world_size > 1: the ray batches (x-major ray order, data/fruit_datamanager.py:105-111) are split into
    contiguous runs, one per rank; each rank compacts its batches locally and the variable-length point lists are
    all-gathered in rank order — the single-process lists, point for point, on every rank."""
    model = pipeline.model
    dm = pipeline.datamanager
    pts = {k: [] for k in SET_NAMES}
    cols = {k: [] for k in SET_NAMES}
    state = {"cap": 1 << 20}
    lattice = getattr(dm, "export_lattice", None)
    lat = None
    smp = getattr(model, "proposal_sampler", None)
    jittered = bool(getattr(smp, "training", False) and getattr(smp, "train_stratified", False))
    if lattice is not None and lattice["n_samples"] == getattr(model, "num_inference_samples", None) and not jittered:
        # fused lattice path: sample positions are implicit (bin centres); a jittering sampler (the reference's as-run
        # export, see components/ray_samplers.py) takes the generic path below
        lat = K.LatticeArg(lattice["xs"], lattice["ys"], lattice["zs"])
    batch = dm.orthographic_ray_generator.ray_batch_size
    n_batches = -(-num_points // batch)
    lo, hi = 0, n_batches
    if world_size > 1:
        from ..sharding import shard_range
        lo, hi = shard_range(n_batches, rank, world_size)
        dm.train_count = lo                     # batch `count` is 1-based: the next one drawn is lo + 1
    if lat is not None:
        # fused lattice path: same batches as OrthographicRayGenerator (ray_generators.py:52-58), positions implicit.
        # The compaction's counts are RUNNING totals on the device (fnr_export_compact), so the batches append to the
        # stream buffers back to back and the host waits ONCE, at the end of the pass — the point lists are the
        # per-batch lists concatenated, as before.  The pass is deterministic: if the buffers were too small it is
        # simply repeated with larger ones.
        dev = model.device
        first = dm.train_count
        while True:
            _ensure_buffers(state, dev)
            counts = torch.zeros(3, dtype=torch.int64, device=dev)
            dm.train_count = first
            for _ in range(lo, hi):
                dm.train_count += 1
                start, end = dm.orthographic_ray_generator.batch_range(dm.train_count)
                density, rgb, logit = model.export_lattice_batch(lat, lattice["direction"], start, end - start)
                K.export_compact(lat, start, end - start, None, density, rgb, logit, state["points"], state["colors"],
                                 counts)
            c = _read_counts(counts, state)
            if max(c) <= state["cap"]:
                break
            while state["cap"] < max(c):
                state["cap"] *= 2
        for s, name in enumerate(SET_NAMES):
            pts[name].append(state["points"][s][:c[s]].to("cpu", copy=True))
            cols[name].append(state["colors"][s][:c[s]].to("cpu", copy=True))
        lo = hi                                  # nothing left for the generic loop
    for _ in range(lo, hi):                      # generic path (explicit positions, e.g. the jittering sampler)
        with torch.no_grad():
            ray_bundle, _ = dm.next_sample_volume(0)
            outputs = model(ray_bundle)
        c = _compact_batch(None, 0, 0, outputs["point_location"].reshape(-1, 3),
                           outputs["density"].reshape(-1).contiguous(),
                           outputs["rgb"].reshape(-1, 3).contiguous(),
                           outputs["semantics"].reshape(-1).contiguous(), state)
        for s, name in enumerate(SET_NAMES):
            # copy=True: the stream buffers are reused by the next batch (a no-op distinction on a GPU, where the
            # transfer already copies)
            pts[name].append(state["points"][s][:c[s]].to("cpu", copy=True))
            cols[name].append(state["colors"][s][:c[s]].to("cpu", copy=True))

    if world_size > 1:
        from ..sharding import gather_chunks
        dev = model.device
        for name in SET_NAMES:
            pts[name] = [gather_chunks([t.to(dev) for t in pts[name]], world_size, torch.empty(0, 3, device=dev)).cpu()]
            cols[name] = [gather_chunks([t.to(dev) for t in cols[name]], world_size,
                                        torch.empty(0, 4, device=dev)).cpu()]

    scale = 1.0 if transform_json is None else float(transform_json["scale"])
    pcd_list = {}
    for name in SET_NAMES:
        p = torch.cat(pts[name], dim=0)
        c = torch.cat(cols[name], dim=0)
        if name != "semantic_colormap" and c.shape[0] != 0:
            c = c / c.max()  # exporter_utils.py:202-203,227-228
        # pcd.scale(1/scale) then pcd.scale(2) about the origin (exporter_utils.py:190-191,216-217,241-242)
        p = p.double() * (1.0 / scale) * 2.0
        path = None
        if output_dir is not None and config is not None:
            path = str(Path(output_dir) / config.load_dir.parts[-3] / f"{name}.ply")
        pcd_list[name] = {"points": p.numpy(), "colors": c.double().numpy()[:, :3], "path": path}
    return pcd_list
