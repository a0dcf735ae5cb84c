"""Thin tensor-level wrappers over the C ABI (one Python function per entry point).

All tensors must live on a HIP device; every call is enqueued on PyTorch's current stream for that
device.  No arithmetic happens here.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _lib as L


def _f32c(t: Tensor) -> Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class StepArena:
    """A caller-owned slab for the per-step buffers of a training loop (training.TrainingSteps): bump allocation, reset
    every other step.  While one is active (`use_arena`) the wrappers below take their output / scratch buffers from it
    instead of torch's caching allocator, so a step's buffers sit at the SAME addresses whenever the same sequence of
    calls runs from the same reset point — what a recorded step program (fnr_program_*) replays against.  Requests
    that do not fit fall back to torch.empty and are counted (`overflow_bytes`): the owner grows the slab between steps."""

    ALIGN = 256

    def __init__(self, device, capacity: int):
        self.device = torch.device(device)
        self.capacity = int(capacity)
        self.buf = torch.empty(self.capacity, dtype=torch.uint8, device=self.device)
        self._typed = {torch.uint8: self.buf}
        self.offset = 0
        self.overflow_bytes = 0
        self.high_water = 0

    def reset(self) -> None:
        self.offset = 0

    def alloc(self, shape, dtype) -> Tensor:
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * _ITEMSIZE[dtype]
        start = self.offset
        end = start + nbytes
        if end > self.capacity:
            self.overflow_bytes += nbytes
            return torch.empty(*shape, dtype=dtype, device=self.device)
        self.offset = (end + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        if self.offset > self.high_water:
            self.high_water = self.offset
        typed = self._typed.get(dtype)
        if typed is None:
            typed = self._typed[dtype] = self.buf.view(dtype)
        isz = _ITEMSIZE[dtype]
        t = typed[start // isz:end // isz]           # (start is a multiple of ALIGN, hence of the item size)
        return t.view(*shape) if len(shape) != 1 else t


_ITEMSIZE = {torch.float32: 4, torch.uint8: 1, torch.int32: 4, torch.int64: 8, torch.float64: 8}
_ARENA: Optional[StepArena] = None


def use_arena(arena: Optional[StepArena]) -> Optional[StepArena]:
    """Make `arena` (or none) the source of the wrappers' per-call buffers -> the previous one."""
    global _ARENA
    prev, _ARENA = _ARENA, arena
    return prev


def _empty(*shape, dtype=torch.float32, device=None) -> Tensor:
    a = _ARENA
    if a is not None and a.device == device:
        return a.alloc(shape, dtype)
    return torch.empty(*shape, dtype=dtype, device=device)


class RaysArg:
    """fnr_rays view of RayBundle tensors (keeps the converted tensors alive)."""

    def __init__(self, origins: Tensor, directions: Tensor, nears: Optional[Tensor], fars: Optional[Tensor],
                 camera_indices: Optional[Tensor] = None):
        L.require_gpu_tensor(origins, "ray origins")
        self.device = origins.device
        self.origins = _f32c(origins.reshape(-1, 3))
        self.directions = _f32c(directions.reshape(-1, 3))
        self.n = self.origins.shape[0]
        self.nears = None if nears is None else _f32c(nears.reshape(-1))
        self.fars = None if fars is None else _f32c(fars.reshape(-1))
        self.cam = None
        if camera_indices is not None:
            self.cam = camera_indices.reshape(-1).to(torch.int32).contiguous()
        self.c = L.fnr_rays(self.n, L.ptr(self.origins), L.ptr(self.directions), L.ptr(self.nears), L.ptr(self.fars),
                            L.ptr(self.cam))

    @property
    def ref(self):
        return C.byref(self.c)


_warp_cache = {}


def make_warp(mode: int, aabb: Tensor) -> L.fnr_warp:
    # reading the aabb back is a device->host copy (a full stream sync): do it once per buffer version
    key = (mode, aabb.data_ptr(), aabb._version, str(aabb.device))
    cached = _warp_cache.get(key)
    if cached is not None:
        return cached
    w = _make_warp_uncached(mode, aabb)
    if len(_warp_cache) > 64:
        _warp_cache.clear()
    _warp_cache[key] = w
    return w


def _make_warp_uncached(mode: int, aabb: Tensor) -> L.fnr_warp:
    w = L.fnr_warp()
    w.mode = mode
    a = aabb.detach().to("cpu", torch.float32).reshape(-1).tolist()
    for i in range(6):
        w.aabb[i] = a[i]
    return w


def make_grid(table: Tensor, n_levels: int, log2_hashmap_size: int, scalings: Sequence[int]) -> L.fnr_grid:
    g = L.fnr_grid()
    g.n_levels = n_levels
    g.log2_hashmap_size = log2_hashmap_size
    for i, s in enumerate(scalings):
        g.scalings[i] = int(s)
    g.table = L.ptr(table)
    return g


def hash_scalings(num_levels: int, min_res: int, max_res: int) -> list:
    """floor(min_res * growth**levels) in float32, exactly as nerfstudio HashEncoding.__init__
    (reference call sites fruit_field.py:124-131, fruit_nerf.py:111-127)."""
    levels = torch.arange(num_levels)
    growth = np.exp((np.log(max_res) - np.log(min_res)) / (num_levels - 1)) if num_levels > 1 else 1
    return [int(v) for v in torch.floor(min_res * growth ** levels).tolist()]


_host_linspace_cache = {}


def host_linspace(start: float, end: float, steps: int, device) -> Tensor:
    """torch.linspace evaluated on the CPU (ATen's rounding), cached on the device."""
    key = (float(start), float(end), int(steps), str(device))
    t = _host_linspace_cache.get(key)
    if t is None:
        t = torch.linspace(start, end, steps, dtype=torch.float32).to(device)
        _host_linspace_cache[key] = t
    return t


# ---- stream dependencies (recordable by step programs: fnr_program_*) ---------------------------------


def _raw(stream) -> Optional[int]:
    """hipStream_t of a torch.cuda.Stream (or a raw handle / None = the default stream)."""
    return stream.cuda_stream if hasattr(stream, "cuda_stream") else stream


class Event:
    """A hipEvent_t (timing disabled) owned by the caller: what crosses between the launch stream and the second stream
    of a training step.  Unlike torch.cuda.Event its record / wait go through the C ABI, so a step program records them."""

    def __init__(self):
        h = C.c_void_p()
        L.check(L.load().fnr_event_create(C.byref(h)), "event_create")
        self.handle = h.value

    def record(self, stream) -> None:
        L.check(L.load().fnr_event_record(self.handle, _raw(stream)), "event_record")

    def __del__(self):
        try:
            if self.handle and L._lib is not None:
                L.load().fnr_event_destroy(self.handle)
        except Exception:   # interpreter shutdown
            pass


def stream_wait_event(stream, event: Event) -> None:
    L.check(L.load().fnr_stream_wait_event(_raw(stream), event.handle), "stream_wait_event")


def stream_wait_stream(waiting, signalling) -> None:
    """Work enqueued on `waiting` from now on runs after everything `signalling` holds now (the host does not block)."""
    L.check(L.load().fnr_stream_wait_stream(_raw(waiting), _raw(signalling)), "stream_wait_stream")


# ---- samplers ---------------------------------------------------------------------------------------


def sample_spaced(rays: RaysArg, spacing_kind: int, S: int, t_rand: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    """t_rand: None (bin edges = linspace), [R] / [R,1] (one jitter per ray) or [R, S+1] (one per bin edge)."""
    lib = L.load()
    dev = rays.device
    base = host_linspace(0.0, 1.0, S + 1, dev)
    spacing = _empty(rays.n, S + 1, device=dev)
    euclid = _empty(rays.n, S + 1, device=dev)
    tr = None if t_rand is None else _f32c(t_rand.reshape(-1))
    per_bin = 0
    if tr is not None and tr.numel() != rays.n:
        if tr.numel() != rays.n * (S + 1):
            raise ValueError(f"t_rand has {tr.numel()} entries; expected {rays.n} (per ray) or {rays.n * (S + 1)} (per bin)")
        per_bin = 1
    L.check(lib.fnr_sample_spaced(rays.ref, spacing_kind, S, L.ptr(base), L.ptr(tr), per_bin, L.ptr(spacing),
                                  L.ptr(euclid), L.stream_ptr(dev)), "sample_spaced")
    return spacing, euclid


def weights_pdf(rays: RaysArg, spacing_kind: int, S_prev: int, S_new: int, density: Tensor, spacing_prev: Tensor,
                euclid_prev: Tensor, anneal: float, rand: Optional[Tensor], want_depth: bool = True):
    lib = L.load()
    dev = rays.device
    weights = _empty(rays.n, S_prev, device=dev)
    depth = _empty(rays.n, device=dev) if want_depth else None
    spacing_new = euclid_new = u_base = None
    if S_new > 0:
        nb = S_new + 1
        u_base = host_linspace(0.0, 1.0 - (1.0 / nb), nb, dev)
        spacing_new = _empty(rays.n, nb, device=dev)
        euclid_new = _empty(rays.n, nb, device=dev)
    rd = None if rand is None else _f32c(rand.reshape(-1))
    L.check(lib.fnr_weights_pdf(rays.ref, spacing_kind, S_prev, S_new, L.ptr(density), L.ptr(spacing_prev),
                                L.ptr(euclid_prev), float(anneal), L.ptr(u_base), L.ptr(rd), L.ptr(weights),
                                L.ptr(depth), L.ptr(spacing_new), L.ptr(euclid_new), L.stream_ptr(dev)),
            "weights_pdf")
    return weights, depth, spacing_new, euclid_new


# ---- networks ----------------------------------------------------------------------------------------


def prop_density_fwd(net: L.fnr_prop_net, warp: L.fnr_warp, rays: RaysArg, euclid: Tensor, S: int,
                     save_feats: bool = False):
    lib = L.load()
    dev = rays.device
    density = _empty(rays.n, S, device=dev)
    feats = _empty(net.grid.n_levels, rays.n * S, 2, device=dev) if save_feats else None
    L.check(lib.fnr_prop_density_fwd(C.byref(net), C.byref(warp), rays.ref, L.ptr(euclid), S, L.ptr(density),
                                     L.ptr(feats), L.stream_ptr(dev)), "prop_density_fwd")
    return density, feats


def hash_encode_fwd(grid: L.fnr_grid, warp: L.fnr_warp, rays: RaysArg, euclid: Tensor, S: int,
                    want_jacobian: bool = False):
    """-> feats [L,N,2], selector [N] (+ the input Jacobian [L,3,N,2] for position_grad_from_jacobian)."""
    lib = L.load()
    dev = rays.device
    N = rays.n * S
    feats = _empty(grid.n_levels, N, 2, device=dev)
    selector = _empty(N, dtype=torch.uint8, device=dev)
    jac = _empty(grid.n_levels, 3, N, 2, device=dev) if want_jacobian else None
    L.check(lib.fnr_hash_encode_fwd(C.byref(grid), C.byref(warp), rays.ref, L.ptr(euclid), S, L.ptr(feats),
                                    L.ptr(selector), L.ptr(jac), L.stream_ptr(dev)), "hash_encode_fwd")
    return (feats, selector, jac) if want_jacobian else (feats, selector)


class LatticeArg:
    def __init__(self, xs: Tensor, ys: Tensor, zs: Tensor):
        self.xs, self.ys, self.zs = _f32c(xs), _f32c(ys), _f32c(zs)
        self.device = self.xs.device
        self.c = L.fnr_lattice(self.xs.numel(), self.ys.numel(), self.zs.numel(), L.ptr(self.xs), L.ptr(self.ys),
                               L.ptr(self.zs))

    @property
    def ref(self):
        return C.byref(self.c)


def hash_encode_lattice(grid: L.fnr_grid, warp: L.fnr_warp, lat: LatticeArg, ray_begin: int, n_rays: int):
    lib = L.load()
    dev = lat.device
    N = n_rays * lat.c.n_z
    feats = _empty(grid.n_levels, N, 2, device=dev)
    selector = _empty(N, dtype=torch.uint8, device=dev)
    L.check(lib.fnr_hash_encode_lattice(C.byref(grid), C.byref(warp), lat.ref, ray_begin, n_rays, L.ptr(feats),
                                        L.ptr(selector), L.stream_ptr(dev)), "hash_encode_lattice")
    return feats, selector


_MLP_FWD_WS = {}


def _mlp_fwd_workspace(dev, n_rays: int) -> Tensor:
    """Per-device scratch for the packed fragment image + per-ray colour bias (reused and grown on demand:
    calls are ordered on the device's stream)."""
    key = (dev.type, dev.index)
    need = L.load().fnr_field_mlp_fwd_workspace_bytes(n_rays)
    ws = _MLP_FWD_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        _MLP_FWD_WS[key] = ws
    return ws


def field_mlp_fwd(net: L.fnr_field_net, rays: RaysArg, S: int, feats: Tensor, selector: Optional[Tensor],
                  mean_embedding: Optional[Tensor], want_geo: bool = False, want_h: bool = False):
    lib = L.load()
    dev = rays.device
    N = rays.n * S
    density = _empty(N, device=dev)
    rgb = _empty(N, 3, device=dev)
    logit = _empty(N, device=dev)
    geo = _empty(N, net.geo_feat_dim, device=dev) if want_geo else None
    # base-MLP output [N, 16 | 32]: saved for field_mlp_bwd; the fruit_nerf_big shape always needs it (its two
    # launches hand h over through it); + the per-ray part of mlp_head's first layer [R,64]
    h_dim = lib.fnr_field_h_dim(C.byref(net))
    if h_dim < 0:
        raise RuntimeError("fruitnerf_hip field_mlp_fwd: " + L.last_error())
    h = _empty(N, h_dim, device=dev) if (want_h or h_dim > 16) else None
    ray_bias = _empty(rays.n, 64, device=dev) if want_h else None
    # training: a private workspace, so that its packed fragment image can be handed to field_mlp_bwd
    ws = (_empty(lib.fnr_field_mlp_fwd_workspace_bytes(0), dtype=torch.uint8, device=dev) if want_h
          else _mlp_fwd_workspace(dev, rays.n))
    L.check(lib.fnr_field_mlp_fwd(C.byref(net), rays.ref, S, L.ptr(feats), L.ptr(selector), L.ptr(mean_embedding),
                                  L.ptr(density), L.ptr(rgb), L.ptr(logit), L.ptr(geo), L.ptr(h), L.ptr(ray_bias),
                                  L.ptr(ws), ws.numel(), L.stream_ptr(dev)),
            "field_mlp_fwd")
    if want_h:
        # the workspace holds the fragment images of THIS call's weights in THIS call's arithmetic (fnr_field_net.mlp_mode)
        return density, rgb, logit, geo, (h, ray_bias, ws, int(net.mlp_mode))
    return density, rgb, logit, geo


def embedding_mean(embedding: Tensor) -> Tensor:
    lib = L.load()
    out = _empty(embedding.shape[1], device=embedding.device)
    L.check(lib.fnr_embedding_mean(L.ptr(embedding), embedding.shape[0], embedding.shape[1], L.ptr(out),
                                   L.stream_ptr(embedding.device)), "embedding_mean")
    return out


def composite_fwd(rays: RaysArg, S: int, euclid: Tensor, density: Tensor, rgb: Tensor, logit: Tensor, training: bool):
    lib = L.load()
    dev = rays.device
    weights = _empty(rays.n, S, device=dev)
    out_rgb = _empty(rays.n, 3, device=dev)
    acc = _empty(rays.n, device=dev)
    depth = _empty(rays.n, device=dev)
    sem = _empty(rays.n, device=dev)
    label = _empty(rays.n, dtype=torch.int64, device=dev)
    L.check(lib.fnr_composite_fwd(rays.ref, S, L.ptr(euclid), L.ptr(density), L.ptr(rgb), L.ptr(logit),
                                  1 if training else 0, L.ptr(weights), L.ptr(out_rgb), L.ptr(acc), L.ptr(depth),
                                  L.ptr(sem), L.ptr(label), L.stream_ptr(dev)), "composite_fwd")
    return weights, out_rgb, acc, depth, sem, label


def composite_fwd_bwd_targets(rays: RaysArg, S: int, euclid: Tensor, density: Tensor, rgb: Tensor, logit: Tensor, image: Tensor,
                              mask: Tensor, sem_weight: float):
    """composite_fwd (training) and composite_bwd_targets as one launch (fnr_composite_fwd_bwd_targets): the same outputs and
    gradients, bit for bit.  -> (weights, out_rgb, acc, depth, sem, label), (d_density, d_rgb, d_logit)."""
    lib = L.load()
    dev = rays.device
    N = rays.n * S
    weights = _empty(rays.n, S, device=dev)
    out_rgb = _empty(rays.n, 3, device=dev)
    acc = _empty(rays.n, device=dev)
    depth = _empty(rays.n, device=dev)
    sem = _empty(rays.n, device=dev)
    label = _empty(rays.n, dtype=torch.int64, device=dev)
    d_density = _empty(N, device=dev)
    d_rgb = _empty(N, 3, device=dev)
    d_logit = _empty(N, device=dev)
    L.check(lib.fnr_composite_fwd_bwd_targets(rays.ref, S, L.ptr(euclid), L.ptr(density), L.ptr(rgb), L.ptr(logit),
                                              L.ptr(_f32c(image)), L.ptr(_f32c(mask.reshape(-1))), float(sem_weight),
                                              L.ptr(weights), L.ptr(out_rgb), L.ptr(acc), L.ptr(depth), L.ptr(sem), L.ptr(label),
                                              L.ptr(d_density), L.ptr(d_rgb), L.ptr(d_logit), L.stream_ptr(dev)),
            "composite_fwd_bwd_targets")
    return (weights, out_rgb, acc, depth, sem, label), (d_density, d_rgb, d_logit)


# ---- export -------------------------------------------------------------------------------------------


def export_compact(lat: Optional[LatticeArg], ray_begin: int, n_rays: int, positions: Optional[Tensor],
                   density: Tensor, rgb: Tensor, logit: Tensor, points: Sequence[Tensor], colors: Sequence[Tensor],
                   counts: Tensor) -> None:
    """Appends this batch's selected samples to the three point/colour streams (order-preserving); `counts`
    (uint64 x3, stored as int64 on the device) holds the running totals.  Positions: lattice or explicit."""
    lib = L.load()
    dev = density.device
    N = density.numel()
    ws = _empty(lib.fnr_export_workspace_bytes(N), dtype=torch.uint8, device=dev)
    cap = points[0].shape[0]
    assert all(p.shape[0] == cap for p in points) and all(c.shape[0] == cap for c in colors)
    parr = (C.c_void_p * 3)(*[L.ptr(p) for p in points])
    carr = (C.c_void_p * 3)(*[L.ptr(c) for c in colors])
    pos = None if positions is None else _f32c(positions.reshape(-1, 3))
    L.check(lib.fnr_export_compact(None if lat is None else lat.ref, ray_begin, n_rays, L.ptr(pos), N,
                                   L.ptr(density), L.ptr(rgb), L.ptr(logit), parr, carr, cap, L.ptr(counts),
                                   L.ptr(ws), L.stream_ptr(dev)), "export_compact")


# ---- training ---------------------------------------------------------------------------------------


def losses_fwd(rgb: Tensor, image: Tensor, semantics: Tensor, fruit_mask: Tensor, semantic_loss_weight: float):
    lib = L.load()
    dev = rgb.device
    R = rgb.shape[0]
    losses = _empty(3, device=dev)  # rgb_loss, semantics_loss, psnr
    d_rgb = _empty(R, 3, device=dev)
    d_sem = _empty(R, device=dev)
    L.check(lib.fnr_losses_fwd(R, L.ptr(_f32c(rgb)), L.ptr(_f32c(image.reshape(R, 3))),
                               L.ptr(_f32c(semantics.reshape(R))), L.ptr(_f32c(fruit_mask.reshape(R))),
                               float(semantic_loss_weight), L.ptr(losses), L.ptr(d_rgb), L.ptr(d_sem),
                               L.stream_ptr(dev)), "losses_fwd")
    return losses, d_rgb, d_sem


def interlevel_fwd(S_f: int, spacing_f: Tensor, weights_f: Tensor, S_p: int, spacing_p: Tensor, weights_p: Tensor,
                   mult: float, loss_acc: Tensor) -> Tensor:
    lib = L.load()
    dev = spacing_f.device
    R = spacing_f.shape[0]
    d_wp = _empty(R, S_p, device=dev)
    L.check(lib.fnr_interlevel_fwd(R, S_f, L.ptr(spacing_f), L.ptr(weights_f), S_p, L.ptr(spacing_p), L.ptr(weights_p),
                                   float(mult), L.ptr(loss_acc), L.ptr(d_wp), L.stream_ptr(dev)), "interlevel_fwd")
    return d_wp


def distortion(S: int, spacing: Tensor, weights: Tensor, out: Optional[Tensor] = None) -> Optional[Tensor]:
    """out: zeroed [FNR_LOSS_SLOTS] accumulator supplied by the caller (who sums it later); None -> returns the sum."""
    lib = L.load()
    dev = spacing.device
    slots = torch.zeros(L.FNR_LOSS_SLOTS, device=dev) if out is None else out
    L.check(lib.fnr_distortion(spacing.shape[0], S, L.ptr(spacing), L.ptr(weights), L.ptr(slots), L.stream_ptr(dev)),
            "distortion")
    return slots.sum() if out is None else None


def train_losses(rgb: Tensor, image: Tensor, semantics: Tensor, fruit_mask: Tensor, semantic_loss_weight: float,
                 S_f: int, spacing_f: Tensor, weights_f: Tensor, levels, interlevel_mult: float, want_distortion: bool,
                 accum: Tensor, fuse_weights_bwd: bool = False, want_ray_grads: bool = True):
    """losses_fwd + interlevel_fwd per proposal level + distortion + the slot sums in one launch.
    levels: [(S_p, spacing_p, weights_p), ...]; accum: float buffer of FNR_TRAIN_LOSSES_ACCUM_FLOATS, zeroed before its
    first use (every completed call leaves it zeroed; it is re-zeroed here when the call fails).
    -> losses [5] (rgb_loss, semantics_loss, psnr, interlevel_loss, distortion), d_rgb [R,3], d_semantics [R],
       [d_weights_p per level].
    fuse_weights_bwd: levels are (S_p, spacing_p, weights_p, euclid_p, density_p) and the last list holds each level's
    d(loss)/d(density) [R,S_p] (= weights_bwd of the level, unit upstream) instead of d_weights_p.
    want_ray_grads False: d_rgb / d_semantics are neither allocated nor written (-> None, None): the caller's composite
    backward forms them itself (composite_bwd_targets)."""
    lib = L.load()
    dev = rgb.device
    R = rgb.shape[0]
    rgb, image = _f32c(rgb), _f32c(image.reshape(R, 3))
    semantics, fruit_mask = _f32c(semantics.reshape(R)), _f32c(fruit_mask.reshape(R))
    # (never from the step arena: callers keep a step's loss scalars beyond the step after next — a replayed step
    #  program writes them to the buffer fnr_step_scalars.losses names)
    losses = torch.empty(5, device=dev)
    d_rgb = _empty(R, 3, device=dev) if want_ray_grads else None
    d_sem = _empty(R, device=dev) if want_ray_grads else None
    n = len(levels)
    outs = [_empty(R, lv[0], device=dev) for lv in levels]
    sp = (C.c_int * max(n, 1))(*[int(lv[0]) for lv in levels])
    vps = lambda ts: (C.c_void_p * max(n, 1))(*[L.ptr(t) for t in ts])   # noqa: E731
    if fuse_weights_bwd:
        d_wp, eu, dn, d_dn = None, vps([lv[3] for lv in levels]), vps([lv[4] for lv in levels]), vps(outs)
    else:
        d_wp, eu, dn, d_dn = vps(outs), None, None, None
    rc = lib.fnr_train_losses(R, L.ptr(rgb), L.ptr(image), L.ptr(semantics), L.ptr(fruit_mask),
                              float(semantic_loss_weight), L.ptr(d_rgb), L.ptr(d_sem), S_f, L.ptr(spacing_f),
                              L.ptr(weights_f), n, sp, vps([lv[1] for lv in levels]), vps([lv[2] for lv in levels]),
                              d_wp, eu, dn, d_dn, float(interlevel_mult), 1 if want_distortion else 0, L.ptr(accum),
                              L.ptr(losses), L.stream_ptr(dev))
    if rc != 0:
        accum.zero_()     # a failed call may leave slots / completion counters dirty for every later step
    L.check(rc, "train_losses")
    return losses, d_rgb, d_sem, outs


def composite_bwd(rays: RaysArg, S: int, euclid: Tensor, density: Tensor, rgb: Tensor, weights: Tensor, g_rgb: Tensor,
                  g_sem: Tensor):
    lib = L.load()
    dev = rays.device
    N = rays.n * S
    d_density = _empty(N, device=dev)
    d_rgb = _empty(N, 3, device=dev)
    d_logit = _empty(N, device=dev)
    L.check(lib.fnr_composite_bwd(rays.ref, S, L.ptr(euclid), L.ptr(density), L.ptr(rgb), L.ptr(weights),
                                  L.ptr(_f32c(g_rgb)), L.ptr(_f32c(g_sem.reshape(-1))), L.ptr(d_density),
                                  L.ptr(d_rgb), L.ptr(d_logit), L.stream_ptr(dev)), "composite_bwd")
    return d_density, d_rgb, d_logit


def composite_bwd_targets(rays: RaysArg, S: int, euclid: Tensor, density: Tensor, rgb: Tensor, weights: Tensor,
                          out_rgb: Tensor, image: Tensor, out_sem: Tensor, mask: Tensor, sem_weight: float):
    """composite_bwd with the per-ray loss gradients formed inside the kernel from the composited outputs and the batch
    (fnr_composite_bwd_targets): no dependence on the losses launch, same bits."""
    lib = L.load()
    dev = rays.device
    N = rays.n * S
    d_density = _empty(N, device=dev)
    d_rgb = _empty(N, 3, device=dev)
    d_logit = _empty(N, device=dev)
    L.check(lib.fnr_composite_bwd_targets(rays.ref, S, L.ptr(euclid), L.ptr(density), L.ptr(rgb), L.ptr(weights),
                                          L.ptr(_f32c(out_rgb)), L.ptr(_f32c(image)), L.ptr(_f32c(out_sem.reshape(-1))),
                                          L.ptr(_f32c(mask.reshape(-1))), float(sem_weight), L.ptr(d_density),
                                          L.ptr(d_rgb), L.ptr(d_logit), L.stream_ptr(dev)), "composite_bwd_targets")
    return d_density, d_rgb, d_logit


def weights_bwd(S: int, euclid: Tensor, density: Tensor, weights: Tensor, d_weights: Tensor,
                upstream: Optional[Tensor]) -> Tensor:
    lib = L.load()
    dev = euclid.device
    R = euclid.shape[0]
    d_density = _empty(R, S, device=dev)
    L.check(lib.fnr_weights_bwd(R, S, L.ptr(euclid), L.ptr(density), L.ptr(weights), L.ptr(d_weights),
                                L.ptr(upstream), L.ptr(d_density), L.stream_ptr(dev)), "weights_bwd")
    return d_density


def field_mlp_bwd(net: L.fnr_field_net, grads: L.fnr_field_net, rays: RaysArg, S: int, feats: Tensor, h_saved,
                  selector: Tensor, d_density: Tensor, d_rgb: Tensor, d_logit: Tensor, jacobian: Optional[Tensor] = None,
                  weight_adam=None):
    """h_saved: what field_mlp_fwd(want_h=True) returned — (h [N,16], ray_bias [R,64], packed weights); a bare h tensor
    is accepted too (the per-ray bias and the fragment image are then recomputed).
    jacobian (hash_encode_fwd(want_jacobian=True)): -> (d_feats, d_position [N,4]): the hash grid's input gradient per
    sample rides along (fnr_field_mlp_bwd_rays); position_grad_reduce(..., d_position.view(1, N, 4), ...) finishes it.
    weight_adam = (fnr_table_adam, gradient arena): the optimiser step of the MLP weights + embedding is taken by the
    kernels that finish their gradients (fnr_field_mlp_bwd_adam, FusedAdam.weight_adam_args)."""
    lib = L.load()
    fwd_mode = h_saved[3] if isinstance(h_saved, tuple) and len(h_saved) > 3 else None
    h_saved, ray_bias, packed = (tuple(h_saved) + (None, None))[:3] if isinstance(h_saved, tuple) else (h_saved, None, None)
    if fwd_mode is not None and fwd_mode != int(net.mlp_mode):
        packed = None   # the forward ran in another arithmetic: its workspace lacks this mode's fragment images — repack
    dev = rays.device
    N = rays.n * S
    d_feats = _empty(*feats.shape, device=dev)
    nbytes = lib.fnr_field_mlp_bwd_workspace_bytes(rays.n, S)
    ws = _empty(nbytes, dtype=torch.uint8, device=dev)
    if weight_adam is not None:
        adam, grad_arena = weight_adam
        d_pos = _empty(N, 4, device=dev) if jacobian is not None else None
        L.check(lib.fnr_field_mlp_bwd_adam(C.byref(net), C.byref(grads), rays.ref, S, L.ptr(feats), L.ptr(h_saved),
                                           L.ptr(ray_bias), L.ptr(packed), L.ptr(selector), L.ptr(d_density), L.ptr(d_rgb),
                                           L.ptr(d_logit), L.ptr(d_feats), L.ptr(jacobian), L.ptr(d_pos), C.byref(adam),
                                           L.ptr(grad_arena), L.ptr(ws), nbytes, L.stream_ptr(dev)), "field_mlp_bwd_adam")
        return (d_feats, d_pos) if jacobian is not None else d_feats
    if jacobian is not None:
        d_pos = _empty(N, 4, device=dev)
        L.check(lib.fnr_field_mlp_bwd_rays(C.byref(net), C.byref(grads), rays.ref, S, L.ptr(feats), L.ptr(h_saved),
                                           L.ptr(ray_bias), L.ptr(packed), L.ptr(selector), L.ptr(d_density), L.ptr(d_rgb),
                                           L.ptr(d_logit), L.ptr(d_feats), L.ptr(jacobian), L.ptr(d_pos), L.ptr(ws), nbytes,
                                           L.stream_ptr(dev)), "field_mlp_bwd_rays")
        return d_feats, d_pos
    L.check(lib.fnr_field_mlp_bwd(C.byref(net), C.byref(grads), rays.ref, S, L.ptr(feats), L.ptr(h_saved),
                                  L.ptr(ray_bias), L.ptr(packed), L.ptr(selector), L.ptr(d_density), L.ptr(d_rgb), L.ptr(d_logit), L.ptr(d_feats), L.ptr(ws), nbytes,
                                  L.stream_ptr(dev)), "field_mlp_bwd")
    return d_feats


class _WorkspaceCache:
    """Persistent scatter workspaces, keyed by (device, stream, size, entry point): LRU, at most `per_tag` buffers per
    (device, entry point).  A training loop uses one key per entry point for its whole life; a process that builds many
    models in sequence (each with its own second stream out of torch's pool of 32) would otherwise keep up to 32 x the
    multi-GB queues alive.  Evicting is always safe: the buffer is only dropped from the cache (a caller that comes back
    for it allocates a new one and says workspace_clean = 0)."""

    def __init__(self, per_tag: int = 2):
        self.per_tag = per_tag
        self.entries = {}          # key -> buffer; dict order = recency (oldest first)
        self.fresh = 0             # buffers made so far (a call that got a fresh one was told workspace_clean = 0)

    def get(self, key, make):
        """-> (buffer, clean): clean = 1 when the buffer has been used by these kernels before (counters left zeroed)."""
        buf = self.entries.pop(key, None)
        if buf is not None:
            self.entries[key] = buf
            return buf, 1
        same = [k for k in self.entries if (k[0], k[1], k[4]) == (key[0], key[1], key[4])]
        for k in same[:max(0, len(same) - self.per_tag + 1)]:
            del self.entries[k]
        buf = self.entries[key] = make()
        self.fresh += 1
        return buf, 0

    # dict-like views for diagnostics
    def items(self):
        return self.entries.items()

    def __len__(self):
        return len(self.entries)

    def clear(self):
        self.entries.clear()


_SCATTER_WS = _WorkspaceCache()


# FNR_SCATTER_MEMSET=1 (hunt switch): never tell the library that the counters were left clean — every
# scatter call then zeroes its counter block with a hipMemsetAsync ahead of the emit kernel instead of relying on the
# previous call's accumulate kernel having put them back to zero (~3 us per call).
SCATTER_MEMSET_EVERY_CALL = os.environ.get("FNR_SCATTER_MEMSET") == "1"


def _scatter_workspace(dev, nbytes: int, tag: str):
    """Persistent scatter workspace per (device, stream, size, entry point) -> (buffer, clean flag).  The kernels leave
    the queue counters zeroed, so after its first use the buffer needs no memset launch (workspace_clean = 1)."""
    key = (dev.type, dev.index, L.stream_ptr(dev), nbytes, tag)
    buf, clean = _SCATTER_WS.get(key, lambda: torch.empty(nbytes, dtype=torch.uint8, device=dev))
    return buf, (0 if SCATTER_MEMSET_EVERY_CALL else clean)


def fresh_workspaces() -> int:
    """How many scatter workspaces have been created so far: a step during which this moved ran a scatter with
    workspace_clean = 0 (its memset launch is not part of the steady-state sequence a step program should hold)."""
    return _SCATTER_WS.fresh


def _forget_scatter_workspaces(dev) -> None:
    """A scatter entry point failed: its launches may have been enqueued in part (emit without accumulate), which leaves
    queue counters that the next call — told workspace_clean = 1 — would add to.  Drop every cached workspace of the
    device; the next calls allocate new ones and zero their counters."""
    for k in [k for k, _ in _SCATTER_WS.items() if (k[0], k[1]) == (dev.type, dev.index)]:
        del _SCATTER_WS.entries[k]


def _scatter_check(rc: int, what: str, dev) -> None:
    if rc != 0:
        _forget_scatter_workspaces(dev)
    L.check(rc, what)


def hash_encode_bwd(grid_grad: L.fnr_grid, warp: L.fnr_warp, rays: RaysArg, euclid: Tensor, S: int,
                    d_feats: Tensor, level_begin: int = 0, level_count: Optional[int] = None) -> None:
    """Scatter the feature gradients of levels [level_begin, level_begin + level_count) (default: all) into the
    gradient table."""
    lib = L.load()
    if level_count is None:
        level_count = grid_grad.n_levels - level_begin
    nbytes = lib.fnr_hash_scatter_workspace_bytes(rays.n * S, level_count, grid_grad.log2_hashmap_size)
    ws, clean = _scatter_workspace(rays.device, nbytes, "field")
    _scatter_check(lib.fnr_hash_encode_bwd(C.byref(grid_grad), C.byref(warp), rays.ref, L.ptr(euclid), S, L.ptr(d_feats),
                                           level_begin, level_count, L.ptr(ws), nbytes, clean, L.stream_ptr(rays.device)),
                   "hash_encode_bwd", rays.device)


def hash_encode_bwd_adam(grid_grad: L.fnr_grid, warp: L.fnr_warp, rays: RaysArg, euclid: Tensor, S: int,
                         d_feats: Tensor, adam: L.fnr_table_adam) -> None:
    """hash_encode_bwd over all levels with the table's optimiser step fused into the accumulate kernel (the gradient
    table is not written; `adam` points at the table's parameter / moment slices)."""
    lib = L.load()
    nbytes = lib.fnr_hash_scatter_workspace_bytes(rays.n * S, grid_grad.n_levels, grid_grad.log2_hashmap_size)
    ws, clean = _scatter_workspace(rays.device, nbytes, "field")
    _scatter_check(lib.fnr_hash_encode_bwd_adam(C.byref(grid_grad), C.byref(warp), rays.ref, L.ptr(euclid), S, L.ptr(d_feats),
                                                L.ptr(ws), nbytes, clean, C.byref(adam), L.stream_ptr(rays.device)),
                   "hash_encode_bwd_adam", rays.device)


def prop_density_bwd(net: L.fnr_prop_net, grads: L.fnr_prop_net, warp: L.fnr_warp, rays: RaysArg, euclid: Tensor,
                     S: int, feats: Tensor, d_density: Tensor, want_position_grad: bool = False,
                     adam=None) -> Optional[Tensor]:
    """want_position_grad: also return d(loss)/d(unit-cube position) [N,4] for position_grad_reduce(n_levels=1).
    adam = (table fnr_table_adam, weight fnr_table_adam, gradient arena): the network's optimiser step is taken by the
    kernels that finish its gradients (fnr_prop_density_bwd_adam)."""
    lib = L.load()
    nbytes = lib.fnr_prop_density_bwd_workspace_bytes(rays.n * S, net.grid.n_levels, net.grid.log2_hashmap_size)
    ws, clean = _scatter_workspace(rays.device, nbytes, "prop")
    d_pos = _empty(rays.n * S, 4, device=rays.device) if want_position_grad else None
    if adam is not None:
        t_adam, w_adam, grad_arena = adam
        _scatter_check(lib.fnr_prop_density_bwd_adam(C.byref(net), C.byref(grads), C.byref(warp), rays.ref, L.ptr(euclid), S,
                                                     L.ptr(feats), L.ptr(d_density), L.ptr(d_pos), C.byref(t_adam),
                                                     C.byref(w_adam), L.ptr(grad_arena), L.ptr(ws), nbytes, clean,
                                                     L.stream_ptr(rays.device)), "prop_density_bwd_adam", rays.device)
        return d_pos
    _scatter_check(lib.fnr_prop_density_bwd(C.byref(net), C.byref(grads), C.byref(warp), rays.ref, L.ptr(euclid), S,
                                            L.ptr(feats), L.ptr(d_density), L.ptr(d_pos), L.ptr(ws), nbytes, clean,
                                            L.stream_ptr(rays.device)),
                   "prop_density_bwd", rays.device)
    return d_pos


def prop_density_bwd_pair(nets, grads, warps, rays: RaysArg, euclids, S, feats, d_density, want_position_grad: bool = False,
                          adam=None, position_ready=None):
    """fnr_prop_density_bwd_pair: both proposal levels of a step (lists of two), their accumulate launches as one.
    adam = ([table fnr_table_adam x 2], weight fnr_table_adam, gradient arena) or None.  -> [d_position | None] x 2.
    position_ready (Event): fnr_prop_density_bwd_pair_split — both MLP backwards first, the event recorded on
    the current stream once the d_position tensors are final, then the scatter."""
    lib = L.load()
    dev = rays.device
    ws, nbytes, clean, d_pos = [], [], [], []
    for q in range(2):
        nb = lib.fnr_prop_density_bwd_workspace_bytes(rays.n * S[q], nets[q].grid.n_levels, nets[q].grid.log2_hashmap_size)
        w, c = _scatter_workspace(dev, nb, f"prop{q}")
        ws.append(w), nbytes.append(nb), clean.append(c)
        d_pos.append(_empty(rays.n * S[q], 4, device=dev) if want_position_grad else None)
    PN, PW, PT = C.POINTER(L.fnr_prop_net), C.POINTER(L.fnr_warp), C.POINTER(L.fnr_table_adam)
    vp = lambda ts: (C.c_void_p * 2)(*[L.ptr(t) for t in ts])   # noqa: E731
    t_adams = w_adam = grad_arena = None
    if adam is not None:
        t_list, w_adam_s, grad_arena = adam
        t_adams = (PT * 2)(*[C.pointer(t) for t in t_list])
        w_adam = C.byref(w_adam_s)
    args = ((PN * 2)(*[C.pointer(n) for n in nets]), (PN * 2)(*[C.pointer(g) for g in grads]),
            (PW * 2)(*[C.pointer(w) for w in warps]), rays.ref, vp(euclids), (C.c_int * 2)(*[int(x) for x in S]), vp(feats),
            vp(d_density), vp(d_pos), t_adams, w_adam, L.ptr(grad_arena), vp(ws), (C.c_size_t * 2)(*nbytes),
            (C.c_int * 2)(*clean), L.stream_ptr(dev))
    if position_ready is not None:
        _scatter_check(lib.fnr_prop_density_bwd_pair_split(*args, C.c_void_p(position_ready.handle)),
                       "prop_density_bwd_pair_split", dev)
    else:
        _scatter_check(lib.fnr_prop_density_bwd_pair(*args), "prop_density_bwd_pair", dev)
    return d_pos


def hash_encode_input_grad(grid: L.fnr_grid, warp: L.fnr_warp, rays: RaysArg, euclid: Tensor, S: int,
                           d_feats: Tensor) -> Tensor:
    """Per-level gradient w.r.t. the unit-cube sample positions: [L][N][4]."""
    lib = L.load()
    partial = _empty(grid.n_levels, rays.n * S, 4, device=rays.device)
    L.check(lib.fnr_hash_encode_input_grad(C.byref(grid), C.byref(warp), rays.ref, L.ptr(euclid), S, L.ptr(d_feats),
                                           L.ptr(partial), L.stream_ptr(rays.device)), "hash_encode_input_grad")
    return partial


def position_grad_from_jacobian(warp: L.fnr_warp, rays: RaysArg, euclid: Tensor, S: int, jacobian: Tensor,
                                d_feats: Tensor, d_origins: Tensor, d_directions: Tensor) -> None:
    """d_origins / d_directions [R,3] += ray gradient of d_feats [L,N,2] through the saved Jacobian [L,3,N,2]."""
    lib = L.load()
    L.check(lib.fnr_position_grad_from_jacobian(C.byref(warp), rays.ref, L.ptr(euclid), S, jacobian.shape[0],
                                                L.ptr(jacobian), L.ptr(d_feats), L.ptr(d_origins), L.ptr(d_directions),
                                                L.stream_ptr(rays.device)), "position_grad_from_jacobian")


def position_grad_reduce_multi(sources, rays: RaysArg, d_origins: Tensor, d_directions: Tensor,
                              accumulate: bool = False) -> None:
    """One launch for several ray-gradient sources [(warp, euclid [R,S+1], S, partial [n_levels,N,4] | [N,4]), ...]:
    d_origins / d_directions [R,3] = (or +=) the sum of their ray gradients, in source order."""
    lib = L.load()
    if len(sources) > L.FNR_MAX_POSITION_SOURCES:
        # more sources than one launch takes (> 3 proposal iterations + the field): in chunks, the later ones accumulating
        m = L.FNR_MAX_POSITION_SOURCES
        for a in range(0, len(sources), m):
            position_grad_reduce_multi(sources[a:a + m], rays, d_origins, d_directions, accumulate=accumulate or a > 0)
        return
    n = len(sources)
    warps = (C.POINTER(L.fnr_warp) * n)(*[C.pointer(w) for w, _, _, _ in sources])
    euclid = (C.c_void_p * n)(*[L.ptr(e) for _, e, _, _ in sources])
    S = (C.c_int * n)(*[int(s_) for _, _, s_, _ in sources])
    levels = (C.c_int * n)(*[(p.shape[0] if p.dim() == 3 else 1) for _, _, _, p in sources])
    partials = (C.c_void_p * n)(*[L.ptr(p) for _, _, _, p in sources])
    L.check(lib.fnr_position_grad_reduce_multi(n, warps, rays.ref, euclid, S, levels, partials, 1 if accumulate else 0,
                                               L.ptr(d_origins), L.ptr(d_directions), L.stream_ptr(rays.device)),
            "position_grad_reduce_multi")


def position_grad_reduce(warp: L.fnr_warp, rays: RaysArg, euclid: Tensor, S: int, partial: Tensor, d_origins: Tensor,
                         d_directions: Tensor) -> None:
    """d_origins / d_directions [R,3] += the ray gradient carried by `partial` ([n_levels][N][4] or [N][4])."""
    lib = L.load()
    n_levels = partial.shape[0] if partial.dim() == 3 else 1
    L.check(lib.fnr_position_grad_reduce(C.byref(warp), rays.ref, L.ptr(euclid), S, n_levels, L.ptr(partial),
                                         L.ptr(d_origins), L.ptr(d_directions), L.stream_ptr(rays.device)),
            "position_grad_reduce")


def adam_step(params: Tensor, grads: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, lr: float, beta1: float,
              beta2: float, eps: float, step: int, grad_scale: float = 1.0, zero_grad: bool = True,
              weight_decay: float = 0.0) -> None:
    lib = L.load()
    L.check(lib.fnr_adam_step(L.ptr(params), L.ptr(grads), L.ptr(exp_avg), L.ptr(exp_avg_sq), params.numel(),
                              float(lr), float(beta1), float(beta2), float(eps), int(step), float(grad_scale),
                              float(weight_decay), 1 if zero_grad else 0, L.stream_ptr(params.device)), "adam_step")


def radam_step(params: Tensor, grads: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, lr: float, beta1: float,
               beta2: float, eps: float, step: int, grad_scale: float = 1.0, zero_grad: bool = True,
               weight_decay: float = 0.0) -> None:
    """torch.optim.RAdam (the optimiser of the fruit_nerf_big / fruit_nerf_huge configs) over a flat buffer."""
    lib = L.load()
    L.check(lib.fnr_radam_step(L.ptr(params), L.ptr(grads), L.ptr(exp_avg), L.ptr(exp_avg_sq), params.numel(),
                               float(lr), float(beta1), float(beta2), float(eps), int(step), float(grad_scale),
                               float(weight_decay), 1 if zero_grad else 0, L.stream_ptr(params.device)), "radam_step")


def adam_step_spans(params: Tensor, grads: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, spans, algorithm: str,
                    beta1: float, beta2: float, eps: float, grad_scale: float = 1.0, zero_grad: bool = True,
                    weight_decay: float = 0.0) -> None:
    """One launch over several spans of one arena: spans = [(offset, count, lr, step), ...] (<= FNR_MAX_ADAM_SPANS),
    each updated exactly as adam_step / radam_step would with its own learning rate and step count."""
    lib = L.load()
    arr = (L.fnr_adam_span * len(spans))(*[L.fnr_adam_span(int(a), int(n), int(step), float(lr), 0)
                                           for a, n, lr, step in spans])
    L.check(lib.fnr_adam_step_spans(L.ptr(params), L.ptr(grads), L.ptr(exp_avg), L.ptr(exp_avg_sq), len(spans), arr,
                                    0 if algorithm == "adam" else 1, float(beta1), float(beta2), float(eps),
                                    float(grad_scale), float(weight_decay), 1 if zero_grad else 0,
                                    L.stream_ptr(params.device)), "adam_step_spans")


# ---- point-cloud front-end of the counting stage ------------------------------------------------------


def _cloud_args(xyz: Tensor):
    if xyz.dtype != torch.float64 or xyz.dim() != 2 or xyz.shape[1] != 3 or not xyz.is_cuda:
        raise ValueError("point clouds are [n,3] float64 device tensors (Open3D / PLY precision)")
    xyz = xyz.contiguous()
    n = xyz.shape[0]
    ws = torch.empty(L.load().fnr_cloud_workspace_bytes(n), dtype=torch.uint8, device=xyz.device)
    return xyz, n, ws


def _bounds(xyz: Tensor):
    """Host copies of the cloud's axis-aligned bounds (Open3D GetMinBound / GetMaxBound); one 48-byte D2H sync."""
    lo_hi = torch.empty(6, dtype=torch.float64, device=xyz.device)
    ws = torch.empty(64, dtype=torch.uint8, device=xyz.device)
    L.check(L.load().fnr_cloud_bounds(L.ptr(xyz), xyz.shape[0], L.ptr(lo_hi), L.ptr(ws), ws.numel(),
                                      L.stream_ptr(xyz.device)), "cloud_bounds")
    v = lo_hi.cpu().tolist()
    return (C.c_double * 3)(*v[:3]), (C.c_double * 3)(*v[3:])


def cloud_radius_count(xyz: Tensor, radius: float, inclusive: bool) -> Tensor:
    """-> int32 [n]: neighbours within radius (strict < unless inclusive), the point itself included."""
    xyz, n, ws = _cloud_args(xyz)
    counts = torch.empty(n, dtype=torch.int32, device=xyz.device)
    if n == 0:
        return counts
    lo, hi = _bounds(xyz)
    L.check(L.load().fnr_cloud_radius_count(L.ptr(xyz), n, lo, hi, float(radius), 1 if inclusive else 0, L.ptr(counts),
                                            L.ptr(ws), ws.numel(), L.stream_ptr(xyz.device)), "cloud_radius_count")
    return counts


def cloud_dbscan(xyz: Tensor, eps: float, min_samples: int):
    """-> (labels int32 [n], n_clusters int32 [1] device)."""
    xyz, n, ws = _cloud_args(xyz)
    labels = torch.empty(n, dtype=torch.int32, device=xyz.device)
    n_clusters = torch.zeros(1, dtype=torch.int32, device=xyz.device)
    if n == 0:
        return labels, n_clusters
    lo, hi = _bounds(xyz)
    L.check(L.load().fnr_cloud_dbscan(L.ptr(xyz), n, lo, hi, float(eps), int(min_samples), L.ptr(labels),
                                      L.ptr(n_clusters), L.ptr(ws), ws.numel(), L.stream_ptr(xyz.device)),
            "cloud_dbscan")
    return labels, n_clusters


def cloud_voxel_down_sample(xyz: Tensor, rgb: Optional[Tensor], voxel_size: float):
    """-> (xyz_out [m,3], rgb_out [m,3] | None), one point per occupied voxel in ascending (iz, iy, ix) order."""
    xyz, n, ws = _cloud_args(xyz)
    if rgb is not None:
        if rgb.shape != xyz.shape or rgb.dtype != torch.float64:
            raise ValueError("colours are [n,3] float64 like the points")
        rgb = rgb.contiguous()
    if n == 0:
        return xyz.clone(), (None if rgb is None else rgb.clone())
    xyz_out = torch.empty_like(xyz)
    rgb_out = None if rgb is None else torch.empty_like(rgb)
    n_out = torch.zeros(1, dtype=torch.int32, device=xyz.device)
    lo, hi = _bounds(xyz)
    L.check(L.load().fnr_cloud_voxel_down_sample(L.ptr(xyz), L.ptr(rgb), n, lo, hi, float(voxel_size), L.ptr(xyz_out),
                                                 L.ptr(rgb_out), L.ptr(n_out), L.ptr(ws), ws.numel(),
                                                 L.stream_ptr(xyz.device)), "cloud_voxel_down_sample")
    m = int(n_out.item())
    return xyz_out[:m], (None if rgb_out is None else rgb_out[:m])


# ---- eval-image metrics ------------------------------------------------------------------------------


_GAUSS11 = None


def ssim_window() -> "C.Array":
    """The 11 float32 weights of torchmetrics' default SSIM window (gaussian, sigma 1.5), evaluated with torch on the
    host exactly as torchmetrics.functional.image.helper._gaussian does (arange((1 - k) / 2, (1 + k) / 2), exp, / sum)."""
    global _GAUSS11
    if _GAUSS11 is None:
        dist = torch.arange((1 - 11) / 2, (1 + 11) / 2, 1, dtype=torch.float32)
        g = torch.exp(-torch.pow(dist / 1.5, 2) / 2)
        _GAUSS11 = (C.c_float * 11)(*(g / g.sum()).tolist())
    return _GAUSS11


def image_metrics(rgb: Tensor, image: Tensor, semantics: Optional[Tensor] = None, mask: Optional[Tensor] = None) -> Tensor:
    """fnr_image_metrics on [H,W,3] images (+ [H,W] logits / mask) -> 8 doubles on the device (see the header)."""
    lib = L.load()
    L.require_gpu_tensor(rgb, "rgb")
    H, W = int(rgb.shape[0]), int(rgb.shape[1])
    rgb, image = _f32c(rgb), _f32c(image)
    sem = None if semantics is None else _f32c(semantics.reshape(H, W))
    msk = None if mask is None else _f32c(mask.reshape(H, W))
    out = torch.empty(8, dtype=torch.float64, device=rgb.device)
    nbytes = lib.fnr_image_metrics_workspace_bytes(H, W)
    ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=rgb.device)
    L.check(lib.fnr_image_metrics(H, W, L.ptr(rgb), L.ptr(image), L.ptr(sem), L.ptr(msk), ssim_window(), L.ptr(out),
                                  L.ptr(ws), ws.numel(), L.stream_ptr(rgb.device)), "image_metrics")
    return out


# ---- caller side -------------------------------------------------------------------------------------


class ImageSetArg:
    """fnr_image_set over the on-device image batch (uint8 images [M,H,W,3], uint8 masks [M,H,W], c2w [M,3,4])."""

    def __init__(self, images: Tensor, masks: Tensor, c2w: Tensor, fx: float, fy: float, cx: float, cy: float):
        L.require_gpu_tensor(images, "images")
        assert images.dtype == torch.uint8 and masks.dtype == torch.uint8
        self.images, self.masks, self.c2w = images.contiguous(), masks.contiguous(), _f32c(c2w)
        M, H, W, _ = images.shape
        self.c = L.fnr_image_set(M, H, W, L.ptr(self.images), L.ptr(self.masks), L.ptr(self.c2w), float(fx), float(fy),
                                 float(cx), float(cy))


def camera_adjust(image_set: ImageSetArg, train_ids: Tensor, pose_adjustment: Tensor) -> Tensor:
    """c2w' [n_train,3,4] = multiply(c2w[train_ids], exp_map_SO3xR3(pose_adjustment))."""
    lib = L.load()
    ids = train_ids.to(torch.int64).contiguous()
    out = torch.empty(ids.numel(), 3, 4, device=pose_adjustment.device)
    L.check(lib.fnr_camera_adjust(L.ptr(image_set.c2w), L.ptr(ids), ids.numel(), L.ptr(_f32c(pose_adjustment)), L.ptr(out),
                                  L.stream_ptr(out.device)), "camera_adjust")
    return out


def camera_pose_grad(image_set: ImageSetArg, train_ids: Tensor, u: Tensor, cam: Tensor, pose_adjustment: Tensor,
                     c2w_adjusted: Tensor, d_origins: Tensor, d_directions: Tensor, pose_grad: Tensor) -> None:
    """pose_grad [n_train,6] += d(loss)/d(pose_adjustment) from the ray gradients of the rays drawn with `u`."""
    lib = L.load()
    ids = train_ids.to(torch.int64).contiguous()
    L.check(lib.fnr_camera_pose_grad(C.byref(image_set.c), L.ptr(ids), ids.numel(), u.shape[0], L.ptr(_f32c(u)),
                                     L.ptr(cam), L.ptr(_f32c(pose_adjustment)), L.ptr(c2w_adjusted), L.ptr(d_origins),
                                     L.ptr(d_directions), L.ptr(pose_grad), L.stream_ptr(u.device)), "camera_pose_grad")


def camera_pose_grad_adam(image_set: ImageSetArg, train_ids: Tensor, u: Tensor, cam: Tensor, c2w_adjusted: Tensor,
                          d_origins: Tensor, d_directions: Tensor, pose_grad: Tensor, adam: L.fnr_table_adam) -> None:
    """camera_pose_grad + the pose table's Adam / RAdam step (adam.params = pose_adjustment) in one launch; pose_grad
    is consumed and left zero."""
    lib = L.load()
    ids = train_ids.to(torch.int64).contiguous()
    L.check(lib.fnr_camera_pose_grad_adam(C.byref(image_set.c), L.ptr(ids), ids.numel(), u.shape[0], L.ptr(_f32c(u)),
                                          L.ptr(cam), L.ptr(c2w_adjusted), L.ptr(d_origins), L.ptr(d_directions),
                                          L.ptr(pose_grad), C.byref(adam), L.stream_ptr(u.device)),
            "camera_pose_grad_adam")


def train_prologue(image_set: ImageSetArg, train_ids: Tensor, n_rays: int, seed: int, offset: int,
                   pose_adjustment: Optional[Tensor], near: float, far: float, S0: int, n_jitter: int = 3,
                   spacing_kind: int = 1) -> dict:
    """fnr_train_prologue: random numbers + camera adjust + pixel sampling + level-0 spaced sampling in one launch."""
    lib = L.load()
    dev = image_set.images.device
    R = int(n_rays)
    ids = train_ids.to(torch.int64).contiguous()
    n_train = ids.numel()
    out = {
        "u": _empty(R, 3, device=dev), "jitter": _empty(n_jitter, R, device=dev),
        "origins": _empty(R, 3, device=dev), "directions": _empty(R, 3, device=dev),
        "cam": _empty(R, dtype=torch.int32, device=dev), "image": _empty(R, 3, device=dev),
        "mask": _empty(R, device=dev), "spacing": _empty(R, S0 + 1, device=dev),
        "euclid": _empty(R, S0 + 1, device=dev),
        "c2w_adjusted": _empty(n_train, 3, 4, device=dev) if pose_adjustment is not None else None,
        "S0": S0, "near": float(near), "far": float(far), "spacing_kind": spacing_kind,
    }
    base = host_linspace(0.0, 1.0, S0 + 1, dev)
    L.check(lib.fnr_train_prologue(C.byref(image_set.c), L.ptr(ids), n_train, R, int(seed) & (2 ** 64 - 1),
                                   int(offset) & (2 ** 64 - 1),
                                   L.ptr(_f32c(pose_adjustment)) if pose_adjustment is not None else None,
                                   L.ptr(out["c2w_adjusted"]), L.ptr(out["u"]), L.ptr(out["jitter"]), n_jitter,
                                   L.ptr(out["origins"]), L.ptr(out["directions"]), L.ptr(out["cam"]), L.ptr(out["image"]),
                                   L.ptr(out["mask"]), float(near), float(far), spacing_kind, S0, L.ptr(base),
                                   L.ptr(out["spacing"]), L.ptr(out["euclid"]), L.stream_ptr(dev)), "train_prologue")
    return out


def sample_pixels(image_set: ImageSetArg, train_ids: Tensor, u: Tensor, c2w_adjusted: Optional[Tensor] = None):
    lib = L.load()
    dev = u.device
    R = u.shape[0]
    ids = train_ids.to(torch.int64).contiguous()
    origins = _empty(R, 3, device=dev)
    directions = _empty(R, 3, device=dev)
    cam = _empty(R, dtype=torch.int32, device=dev)
    image = _empty(R, 3, device=dev)
    mask = _empty(R, device=dev)
    L.check(lib.fnr_sample_pixels(C.byref(image_set.c), L.ptr(ids), ids.numel(), R, L.ptr(_f32c(u)),
                                  L.ptr(c2w_adjusted), L.ptr(origins), L.ptr(directions), L.ptr(cam), L.ptr(image), L.ptr(mask), L.stream_ptr(dev)),
            "sample_pixels")
    return origins, directions, cam, image, mask
