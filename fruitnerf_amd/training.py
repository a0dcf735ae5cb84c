"""Differentiable training path of FruitModel on the HIP kernels.

Mirrors what Nerfstudio's Trainer does around the reference model:
    set_anneal(step) -> model(ray_bundle) -> get_loss_dict -> sum -> backward -> optimizer step -> step_cb
with the reference's semantics:
  * rgb loss reaches the field through rgb samples AND compositing weights (density);
  * the semantic loss only trains mlp_semantics + the head (detached geo features and detached weights,
    fruit_field.py:263-265, fruit_nerf.py:343-345);
  * proposal networks are trained by the interlevel loss only, and only on "updated" steps
    (ProposalNetworkSampler, fruit_nerf.py:131-158);
  * data parallelism = DistributedDataParallel's gradient all-reduce(mean) (fruit_pipeline.py:116-118):
    here ONE RCCL all-reduce over the flat gradient arena, then the fused Adam step scales by 1/world.

Gradients are written by the HIP backward kernels straight into the model's flat gradient arena
(`param.grad` are views of it); the autograd Functions below only carry the dependency structure.
"""
from __future__ import annotations

import contextlib
import functools
import os

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _kernels as K
from . import _lib as L


def _anchor(model) -> Tensor:
    a = getattr(model, "_grad_anchor", None)
    if a is None or a.device != model.device:
        a = torch.zeros(1, device=model.device, requires_grad=True)
        model._grad_anchor = a
    return a


def _field_ray_grads(model, rctx, d_feats: Tensor, d_origins: Tensor, d_directions: Tensor,
                     d_position: Optional[Tensor] = None) -> None:
    """d(loss)/d(ray origins, directions) through the main field's hash grid (position_grad.hip): the only path from
    the losses to the rays — PDFSampler detaches its bins, SHEncoding runs under no_grad.
    d_position [N,4]: the per-sample input gradient field_mlp_bwd(jacobian=...) already formed — only the warp /
    frustum chain and the sum over each ray are left."""
    fld = model.field
    lv = rctx.levels[-1]
    if d_position is not None:
        K.position_grad_reduce(fld.warp_struct(), rctx.rays, lv["euclid"], lv["S"], d_position.view(1, -1, 4), d_origins,
                               d_directions)
        return
    if rctx.field_jacobian is not None:   # saved by the forward encode: no table gathers here
        K.position_grad_from_jacobian(fld.warp_struct(), rctx.rays, lv["euclid"], lv["S"], rctx.field_jacobian, d_feats,
                                      d_origins, d_directions)
        return
    partial = K.hash_encode_input_grad(fld.net_struct().grid, fld.warp_struct(), rctx.rays, lv["euclid"], lv["S"], d_feats)
    K.position_grad_reduce(fld.warp_struct(), rctx.rays, lv["euclid"], lv["S"], partial, d_origins, d_directions)


class _RenderFn(torch.autograd.Function):
    """rays -> (rgb, semantics, accumulation, depth, prop depths); backward runs compositing-bwd,
    field-MLP-bwd (MFMA) and the hash-grid scatter."""

    @staticmethod
    def forward(ctx, anchor, origins, directions, model, ray_bundle, jitter):
        # origins / directions are ray_bundle's tensors, passed explicitly so that autograd can hand their gradient
        # to a camera-pose optimiser when they carry history (fruit_nerf_config.py:39-43)
        outputs, rctx = model._render(ray_bundle, jitter)
        ctx.model = model
        ctx.rctx = rctx
        n_prop = len(rctx.levels) - 1
        outs = [outputs["rgb"], outputs["semantics"], outputs["accumulation"], outputs["depth"]]
        outs += [outputs[f"prop_depth_{i}"] for i in range(n_prop)]
        ctx.mark_non_differentiable(*outs[2:])
        ctx.rctx_holder = rctx
        model._last_render_ctx = rctx
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_rgb, g_sem, *_):
        model, rctx = ctx.model, ctx.rctx
        arena = model.arena()
        arena.reattach_grads()
        rays = rctx.rays
        lv = rctx.levels[-1]
        S = lv["S"]
        dev = rays.device
        if g_rgb is None:
            g_rgb = torch.zeros(rays.n, 3, device=dev)
        if g_sem is None:
            g_sem = torch.zeros(rays.n, 1, device=dev)
        d_density, d_rgb, d_logit = K.composite_bwd(rays, S, lv["euclid"], rctx.sample_density, rctx.sample_rgb,
                                                    rctx.weights, g_rgb.contiguous(), g_sem.contiguous())
        fld = model.field
        net, gnet = fld.net_struct(), fld.net_struct(grads=True)
        want_rays = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        d_pos = None
        if want_rays and rctx.field_jacobian is not None:
            d_feats, d_pos = K.field_mlp_bwd(net, gnet, rays, S, rctx.field_feats, rctx.field_h, rctx.field_selector,
                                             d_density, d_rgb, d_logit, jacobian=rctx.field_jacobian)
        else:
            d_feats = K.field_mlp_bwd(net, gnet, rays, S, rctx.field_feats, rctx.field_h, rctx.field_selector, d_density,
                                      d_rgb, d_logit)
        d_o = d_d = None
        if want_rays:
            d_o = torch.zeros(rays.n, 3, device=dev)
            d_d = torch.zeros(rays.n, 3, device=dev)
            if d_pos is None:   # no saved Jacobian: the gather path reads the table, so it runs BEFORE any table update
                _field_ray_grads(model, rctx, d_feats, d_o, d_d)
        K.hash_encode_bwd(gnet.grid, fld.warp_struct(), rays, lv["euclid"], S, d_feats)
        if d_pos is not None:
            _field_ray_grads(model, rctx, d_feats, d_o, d_d, d_pos)
        return None, d_o, d_d, None, None, None


class _FieldFn(torch.autograd.Function):
    """FruitField.forward / get_density as one autograd node: (density, rgb, logit[, geo]) per sample; backward =
    field-MLP backward (MFMA) + hash-grid scatter, gradients accumulate into the field's arena (`param.grad`)."""

    @staticmethod
    def forward(ctx, anchor, field, rays, euclid, S, want_geo):
        net = field.net_struct()
        feats, selector = K.hash_encode_fwd(net.grid, field.warp_struct(), rays, euclid, S)
        density, rgb, logit, geo, saved = K.field_mlp_fwd(net, rays, S, feats, selector, None, want_geo=want_geo,
                                                          want_h=True)
        ctx.field, ctx.rays, ctx.euclid, ctx.S = field, rays, euclid, S
        ctx.saved = (feats, selector, saved)
        if geo is None:
            geo = density.new_empty(0)
        ctx.mark_non_differentiable(geo)
        return density, rgb, logit, geo

    @staticmethod
    def backward(ctx, g_density, g_rgb, g_logit, _):
        field, rays, S = ctx.field, ctx.rays, ctx.S
        feats, selector, saved = ctx.saved
        field._arena.reattach_grads()
        N = rays.n * S
        dev = rays.device
        z = lambda g, *shape: (torch.zeros(*shape, device=dev) if g is None else g.reshape(*shape).float().contiguous())  # noqa: E731
        net, gnet = field.net_struct(), field.net_struct(grads=True)
        d_feats = K.field_mlp_bwd(net, gnet, rays, S, feats, saved, selector, z(g_density, N), z(g_rgb, N, 3),
                                  z(g_logit, N))
        K.hash_encode_bwd(gnet.grid, field.warp_struct(), rays, ctx.euclid, S, d_feats)
        return None, None, None, None, None, None


def field_with_grad(field, rays, euclid, S: int, want_geo: bool):
    anchor = getattr(field, "_grad_anchor", None)
    if anchor is None or anchor.device != rays.device:
        anchor = field._grad_anchor = torch.zeros(1, device=rays.device, requires_grad=True)
    density, rgb, logit, geo = _FieldFn.apply(anchor, field, rays, euclid, S, want_geo)
    return density, rgb, logit, (geo if want_geo else None)


class _InterlevelFn(torch.autograd.Function):
    """interlevel_loss(weights_list, ray_samples_list) (fruit_nerf.py:368-370); backward trains the
    proposal networks (only when this step 'updated' them)."""

    @staticmethod
    def forward(ctx, anchor, origins, directions, model, rctx, mult):
        fin = rctx.levels[-1]
        dev = rctx.rays.device
        from . import _lib as L
        loss = torch.zeros(L.FNR_LOSS_SLOTS, device=dev)
        d_wps = []
        for lv in rctx.levels[:-1]:
            d_wps.append(K.interlevel_fwd(fin["S"], fin["spacing"], fin["weights"], lv["S"], lv["spacing"],
                                          lv["weights"], mult, loss))
        ctx.model, ctx.rctx, ctx.d_wps = model, rctx, d_wps
        return loss.sum()

    @staticmethod
    def backward(ctx, g):
        model, rctx = ctx.model, ctx.rctx
        if not (rctx.training and rctx.updated):
            return None, None, None, None, None, None  # proposal densities were computed under no_grad
        arena = model.arena()
        arena.reattach_grads()
        rays = rctx.rays
        up = g.reshape(1).float().contiguous()
        want_rays = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        d_o = torch.zeros(rays.n, 3, device=rays.device) if want_rays else None
        d_d = torch.zeros(rays.n, 3, device=rays.device) if want_rays else None
        _proposal_backward(model, rctx, ctx.d_wps, up, d_o, d_d)
        return None, d_o, d_d, None, None, None


def _proposal_backward(model, rctx, d_wps, upstream: Optional[Tensor], d_origins: Optional[Tensor],
                       d_directions: Optional[Tensor], level_streams: bool = False,
                       collect: Optional[list] = None, optimizer: Optional["FusedAdam"] = None,
                       position_ready=None) -> bool:
    """Backward of the interlevel loss into the proposal networks (and, when asked, into the rays).
    position_ready (torch.cuda.Event, with collect): recorded as soon as the collected d_position tensors are final,
    ahead of the levels' scatter (fnr_prop_density_bwd_pair_split) -> True when it was (the paired path), else False.
    upstream None: d_wps already holds d(loss)/d(density) per level (train_losses(fuse_weights_bwd=True)).
    level_streams: the levels' chains (MLP backward -> weight reduce -> scatter emit -> accumulate; they share nothing when
    every level has its own network) run side by side, level 0 on the current stream and the others on side streams;
    their ray-gradient sums (+= into the same [R,3] buffers) follow on the current stream after the join.
    collect: a list that receives the levels' ray-gradient sources (warp, euclid, S, d_position) INSTEAD of their
    reduction into d_origins / d_directions — the caller reduces all sources of the step in one launch.
    optimizer: the networks' optimiser steps are taken by the kernels that finish their gradients
    (fnr_prop_density_bwd_adam; single process, one network per level)."""
    cfg = model.config
    rays = rctx.rays
    dev = rays.device
    levels = list(zip(rctx.levels[:-1], d_wps))
    side_by_side = level_streams and len(levels) > 1 and not cfg.use_same_proposal_network
    main = torch.cuda.current_stream(dev) if side_by_side else None
    pool = model.__dict__.setdefault("_level_streams", []) if side_by_side else []
    while side_by_side and len(pool) < len(levels) - 1:
        pool.append(torch.cuda.Stream(device=dev))
    pending = []
    if PAIR_PROPOSAL_LEVELS and len(levels) == 2 and not cfg.use_same_proposal_network and not side_by_side \
            and upstream is None:
        # both levels through one entry point: their accumulate launches run as one (fnr_prop_density_bwd_pair)
        nets = [model.proposal_networks[0], model.proposal_networks[1]]
        adam = None
        if optimizer is not None:
            t_adams = [optimizer.table_adam_args(n.encoding.hash_table, "proposal_networks")[0] for n in nets]
            (w_adam, grad_arena), _ = optimizer.weight_adam_args("proposal_networks")
            adam = (t_adams, w_adam, grad_arena)
        d_pos = K.prop_density_bwd_pair([n.prop_struct() for n in nets], [n.prop_struct(grads=True) for n in nets],
                                        [n.warp_struct() for n in nets], rays, [lv["euclid"] for lv, _ in levels],
                                        [lv["S"] for lv, _ in levels], [lv["feats"] for lv, _ in levels],
                                        [d_wp for _, d_wp in levels], want_position_grad=d_origins is not None, adam=adam,
                                        position_ready=position_ready if collect is not None else None)
        for net, (lv, _), dp in zip(nets, levels, d_pos):
            if d_origins is None:
                continue
            if collect is not None:
                collect.append((net.warp_struct(), lv["euclid"], lv["S"], dp))
            else:
                K.position_grad_reduce(net.warp_struct(), rays, lv["euclid"], lv["S"], dp, d_origins, d_directions)
        return position_ready is not None and collect is not None
    for i, (lv, d_wp) in enumerate(levels):
        net = model.proposal_networks[0 if cfg.use_same_proposal_network else i]
        stream = pool[i - 1] if side_by_side and i > 0 else None
        if stream is not None:
            stream.wait_stream(main)
        with torch.cuda.stream(stream) if stream is not None else _null_context():
            d_density = d_wp if upstream is None else \
                K.weights_bwd(lv["S"], lv["euclid"], lv["density"], lv["weights"], d_wp, upstream)
            adam = None
            if optimizer is not None:
                t_adam, _ = optimizer.table_adam_args(net.encoding.hash_table, "proposal_networks")
                (w_adam, grad_arena), _ = optimizer.weight_adam_args("proposal_networks")
                adam = (t_adam, w_adam, grad_arena)
            d_pos = K.prop_density_bwd(net.prop_struct(), net.prop_struct(grads=True), net.warp_struct(), rays,
                                       lv["euclid"], lv["S"], lv["feats"], d_density,
                                       want_position_grad=d_origins is not None, adam=adam)
        if d_origins is None:
            continue
        if collect is not None:
            collect.append((net.warp_struct(), lv["euclid"], lv["S"], d_pos))
        elif side_by_side:
            pending.append((net, lv, d_pos, stream))
        else:
            K.position_grad_reduce(net.warp_struct(), rays, lv["euclid"], lv["S"], d_pos, d_origins, d_directions)
    for stream in pool[:len(levels) - 1] if side_by_side else ():
        main.wait_stream(stream)
    for net, lv, d_pos, stream in pending:
        if stream is not None:
            d_pos.record_stream(main)
        K.position_grad_reduce(net.warp_struct(), rays, lv["euclid"], lv["S"], d_pos, d_origins, d_directions)


class _null_context:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


class _LossFn(torch.autograd.Function):
    """rgb_loss + semantics_loss in one launch; unit gradients are produced in the forward."""

    @staticmethod
    def forward(ctx, rgb, semantics, image, fruit_mask, weight, cached):
        # get_metrics_dict runs just before get_loss_dict on the same outputs/batch (fruit_pipeline.py:131,144)
        # and needs the same MSE: it leaves its launch's results here instead of launching twice
        losses, d_rgb, d_sem = cached if cached is not None else K.losses_fwd(rgb, image, semantics, fruit_mask,
                                                                              weight)
        ctx.save_for_backward(d_rgb, d_sem)
        ctx.sem_shape = semantics.shape
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, g0, g1):
        d_rgb, d_sem = ctx.saved_tensors
        return d_rgb * g0, (d_sem * g1).view(ctx.sem_shape), None, None, None, None


def render_with_grad(model, ray_bundle, jitter=None):
    outs = _RenderFn.apply(_anchor(model), ray_bundle.origins, ray_bundle.directions, model, ray_bundle, jitter)
    rctx = model._last_render_ctx
    n_prop = len(rctx.levels) - 1
    outputs = {"rgb": outs[0], "semantics": outs[1], "accumulation": outs[2], "depth": outs[3]}
    for i in range(n_prop):
        outputs[f"prop_depth_{i}"] = outs[4 + i]
    return outputs, rctx


def fused_losses(model, outputs, batch) -> Dict[str, Tensor]:
    """get_loss_dict (fruit_nerf.py:359-372)."""
    dev = outputs["rgb"].device
    image = batch["image"].to(dev)
    mask = batch["fruit_mask"].to(dev)
    cached = outputs.pop("_loss_cache", None)
    if cached is not None and (cached[3] is not batch or cached[4] != model.config.semantic_loss_weight):
        cached = None
    rgb_loss, sem_loss = _LossFn.apply(outputs["rgb"], outputs["semantics"], image, mask,
                                       model.config.semantic_loss_weight, None if cached is None else cached[:3])
    loss_dict = {"rgb_loss": rgb_loss, "semantics_loss": sem_loss}
    if model.training:
        rctx = outputs["_ctx"]
        rb = rctx.ray_bundle
        loss_dict["interlevel_loss"] = _InterlevelFn.apply(_anchor(model), rb.origins, rb.directions, model, rctx,
                                                           model.config.interlevel_loss_mult)
    return loss_dict


def metrics(model, outputs, batch) -> Dict[str, Tensor]:
    """get_metrics_dict (fruit_nerf.py:396-401): PSNR(data_range=1) + distortion (metric only)."""
    dev = outputs["rgb"].device
    with torch.no_grad():
        image = batch["image"].to(dev)
        w = model.config.semantic_loss_weight
        losses, d_rgb, d_sem = K.losses_fwd(outputs["rgb"].detach(), image, outputs["semantics"].detach(),
                                            batch["fruit_mask"].to(dev), w)
        outputs["_loss_cache"] = (losses, d_rgb, d_sem, batch, w)
        psnr = losses[2]
        fin = outputs["_ctx"].levels[-1]
        dist = K.distortion(fin["S"], fin["spacing"], fin["weights"])
    return {"psnr": psnr, "distortion": dist}


# ---- optimiser ----------------------------------------------------------------------------------------------


@functools.lru_cache(maxsize=4096)     # (numpy scalar arithmetic: ~6 us per call, six calls per step)
def exponential_decay_lr(step: int, lr_init: float, lr_final: float, max_steps: int) -> float:
    """nerfstudio ExponentialDecayScheduler without warm-up (fruit_nerf_config.py:49,53)."""
    t = float(np.clip(step / max_steps, 0, 1))
    return float(np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t))


L_MAX_ADAM_SPANS = 8   # include/fruitnerf_hip.h: FNR_MAX_ADAM_SPANS
# fnr_table_adam.touched: the fused table steps skip pairs of rows that never received a gradient (their update is exactly
# zero).  FNR_SPARSE_TOUCH=0 sweeps every row (A/B; results are bit-identical either way, tests/test_gpu_training_parity.py)
SPARSE_TOUCH_SKIPPING = os.environ.get("FNR_SPARSE_TOUCH", "1") != "0"


class FusedAdam:
    """torch.optim.Adam semantics for both parameter groups of the `fruit_nerf` method
    (AdamOptimizerConfig(lr=1e-2, eps=1e-15) + ExponentialDecay to 1e-4 over 200k steps,
    fruit_nerf_config.py:47-56) as one launch per group over the flat arena, fused with the 1/world
    scaling of the all-reduced gradient and with zero_grad."""

    def __init__(self, model, lr: float = 1e-2, eps: float = 1e-15, betas=(0.9, 0.999), lr_final: float = 1e-4,
                 max_steps: int = 200000, group_lr: Optional[Dict[str, dict]] = None, algorithm: str = "adam",
                 weight_decay: float = 0.0, skip_groups_without_grad: bool = True):
        # algorithm="radam": torch.optim.RAdam, the optimiser of the fruit_nerf_big / fruit_nerf_huge method configs
        # (RAdamOptimizerConfig, fruit_nerf_config.py:97-106,148-160)
        if algorithm not in ("adam", "radam"):
            raise ValueError(f"unknown optimiser algorithm {algorithm!r}")
        self.algorithm, self.weight_decay = algorithm, weight_decay
        # True (torch >= 2.0, where Optimizer.zero_grad(set_to_none=True) is the default): a group that received no
        # gradient this iteration — the proposal networks on the steps that evaluate them under no_grad — is skipped by
        # torch.optim entirely.  False reproduces torch 1.13 (nerfstudio 0.3.2's other supported version): zero_grad leaves
        # ZERO tensors, so Adam still decays the moments, moves the parameters along them and advances its step count.
        self.skip_groups_without_grad = skip_groups_without_grad
        self.model = model
        self.arena = model.arena()
        self.betas, self.eps = betas, eps
        self.exp_avg = torch.zeros_like(self.arena.params)
        self.exp_avg_sq = torch.zeros_like(self.arena.params)
        self.groups = group_lr or {name: dict(lr=lr, lr_final=lr_final, max_steps=max_steps)
                                   for name in self.arena.group_ranges}
        self.step_count = 0                                   # scheduler steps (every iteration, every group)
        self.group_steps = {name: 0 for name in self.arena.group_ranges}   # optimiser steps a group actually took
        # sparse-touch skipping of the fused table steps (fnr_table_adam.touched): one bit per pair of table rows, "ever
        # received a gradient", per hash table; valid while it covers every pair with a non-zero moment (true from zero
        # moments on).  Keyed by the table's arena offset -> (length, bitmap).  Any optimiser step over a table's span that
        # does NOT go through the fused table kernels (step_span / step: the exchange path, fuse_table_optimizer=False,
        # scaler_step) makes moments non-zero without setting bits, so those paths drop the span's bitmap and the next
        # fused step rebuilds it from the moments; rebuild_touched() after loading or editing moments from outside.
        self._touched: Dict[int, Tuple[int, Tensor]] = {}

    def current_lr(self, name: str) -> float:
        g = self.groups[name]
        if g.get("lr_final") is None:
            return g["lr"]
        return exponential_decay_lr(self.step_count, g["lr"], g["lr_final"], g["max_steps"])

    def begin_step(self, skip=()) -> Dict[str, float]:
        """Advance the counters and return this update's learning rate per parameter group (scheduler.step() runs
        after optimizer.step(): update k uses lr(k-1)).

        skip: groups that received NO gradient this iteration.  The reference's proposal networks are evaluated under
        no_grad on 4 of 5 steps after warm-up (fruit_nerf.py:131-136 via ProposalNetworkSampler), zero_grad() leaves
        their .grad = None (torch >= 2.0: set_to_none), and torch.optim.Adam / RAdam skip such parameters entirely —
        no moment decay, no movement, and their per-parameter `step` (bias correction, RAdam's rho_t) does not advance.
        The learning-rate schedulers still tick every iteration."""
        self.step_count += 1
        for name in self.group_steps:
            if name not in skip:
                self.group_steps[name] += 1
        return {name: (exponential_decay_lr(self.step_count - 1, g["lr"], g["lr_final"], g["max_steps"])
                       if g.get("lr_final") is not None else g["lr"]) for name, g in self.groups.items()}

    def touched_bitmap(self, table: Tensor, a: int, n: int) -> Optional[Tensor]:
        """The persistent pair bitmap of `table` (arena span [a, a + n)); None when skipping does not apply."""
        if not SPARSE_TOUCH_SKIPPING or self.weight_decay != 0.0 or n % 128 != 0:
            return None
        hit = self._touched.get(a)
        bm = hit[1] if hit is not None and hit[0] == n else None
        if bm is None or bm.device != self.arena.params.device:
            # from the moments (all zero at construction): bit i of word i / 32 = "pair i has a non-zero moment"
            m, v = self.exp_avg[a:a + n].view(-1, 4), self.exp_avg_sq[a:a + n].view(-1, 4)
            ever = ((m != 0) | (v != 0)).any(dim=1).view(-1, 32)           # [n / 128 words, 32 pairs]
            words = (ever.to(torch.int64) << torch.arange(32, device=ever.device, dtype=torch.int64)).sum(dim=1)
            bm = torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32).contiguous()   # same 32 bits
            self._touched[a] = (n, bm)
        return bm

    def _invalidate_touched(self, a: int, b: int) -> None:
        """Arena elements [a, b) are about to take an optimiser step outside the fused table kernels."""
        if self._touched:
            for ta in [ta for ta, (tn, _) in self._touched.items() if ta < b and a < ta + tn]:
                del self._touched[ta]

    def rebuild_touched(self) -> None:
        """Forget the bitmaps: they are rebuilt from the moments at the next fused table step (after exp_avg / exp_avg_sq
        were loaded or edited from outside)."""
        self._touched.clear()

    def table_adam_args(self, table: Tensor, group: str = "fields", grad_scale: float = 1.0):
        """fnr_table_adam for the NEXT update of `table` (a parameter of `group` that lives in the arena): learning
        rate and step count as begin_step() will set them — the fused scatter (fnr_hash_encode_bwd_adam) runs before
        the optimiser's own step and takes this parameter's update with it.  -> (struct, (a, b) arena span)."""
        from . import _lib as L
        hit = [(off, n) for _, p, off, n in self.arena.entries if p is table]
        if len(hit) != 1:
            raise RuntimeError("parameter not found in the arena")
        a, n = hit[0]
        g = self.groups[group]
        lr = (exponential_decay_lr(self.step_count, g["lr"], g["lr_final"], g["max_steps"])
              if g.get("lr_final") is not None else g["lr"])
        args = L.table_adam(0 if self.algorithm == "adam" else 1, lr, self.betas[0], self.betas[1], self.eps,
                            self.group_steps[group] + 1, grad_scale, self.weight_decay,
                            L.ptr(self.arena.params[a:a + n]), L.ptr(self.exp_avg[a:a + n]),
                            L.ptr(self.exp_avg_sq[a:a + n]), L.ptr(self.touched_bitmap(table, a, n)),
                            slot=L.ADAM_SLOTS.get(group, 0))
        return args, (a, a + n)

    def weight_adam_args(self, group: str = "fields", grad_scale: float = 1.0):
        """fnr_table_adam for the NEXT update of `group`, addressed through the whole arenas (fnr_field_mlp_bwd_adam: the
        kernels that finish the gradients of the field's MLP weights and embedding take those parameters' step).
        -> ((struct, gradient arena), the group's arena span)."""
        from . import _lib as L
        g = self.groups[group]
        lr = (exponential_decay_lr(self.step_count, g["lr"], g["lr_final"], g["max_steps"])
              if g.get("lr_final") is not None else g["lr"])
        args = L.table_adam(0 if self.algorithm == "adam" else 1, lr, self.betas[0], self.betas[1], self.eps,
                            self.group_steps[group] + 1, grad_scale, self.weight_decay,
                            L.ptr(self.arena.params), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq), None,
                            slot=L.ADAM_SLOTS.get(group, 0))
        return (args, self.arena.grads), tuple(self.arena.group_ranges[group])

    def step_span(self, a: int, b: int, lr: float, grad_scale: float = 1.0, group: Optional[str] = None,
                  step: Optional[int] = None) -> None:
        """Adam update (+ zero_grad) of arena elements [a, b); begin_step() must have been called for this step.
        group: whose step counter feeds the bias corrections (default: the group that contains a); step: that counter's
        value when the update was planned (a deferred update runs after later bookkeeping)."""
        if b > a:
            self._invalidate_touched(a, b)
            if group is None:
                group = next(n for n, (ga, gb) in self.arena.group_ranges.items() if ga <= a < gb)
            fn = K.adam_step if self.algorithm == "adam" else K.radam_step
            fn(self.arena.params[a:b], self.arena.grads[a:b], self.exp_avg[a:b], self.exp_avg_sq[a:b], lr,
               self.betas[0], self.betas[1], self.eps, self.group_steps[group] if step is None else step, grad_scale, True,
               weight_decay=self.weight_decay)

    def plan_runs(self, lrs: Dict[str, float], skip=(), done=()) -> list:
        """[a, b, lr, group] arena runs of one optimiser step (after begin_step): groups in `skip` left out, adjacent
        groups merged when learning rate and step count coincide, the `done` spans cut out."""
        runs = []
        for name, (a, b) in self.arena.group_ranges.items():
            if name in skip:
                continue
            if runs and runs[-1][1] == a and runs[-1][2] == lrs[name] and \
                    self.group_steps[runs[-1][3]] == self.group_steps[name]:
                runs[-1][1] = b
            else:
                runs.append([a, b, lrs[name], name])
        for da, db in sorted(done):
            cut = []
            for a, b, lr, name in runs:
                if db <= a or da >= b:
                    cut.append([a, b, lr, name])
                else:
                    if a < da:
                        cut.append([a, da, lr, name])
                    if db < b:
                        cut.append([db, b, lr, name])
            runs = cut
        return [r for r in runs if r[1] > r[0]]

    def step(self, grad_scale: float = 1.0, skip=(), done=()) -> None:
        """One optimiser step over every group except `skip` (see begin_step).  Adjacent groups whose learning rate AND
        step count coincide share one launch.  done: arena spans whose update a fused kernel already applied for this
        step (table_adam_args) — they are cut out of the launches."""
        lrs = self.begin_step(skip)
        runs = self.plan_runs(lrs, skip, done)
        if len(runs) <= 1 or len(runs) > L_MAX_ADAM_SPANS:
            for a, b, lr, name in runs:
                self.step_span(a, b, lr, grad_scale, group=name)
        else:   # one launch for all of them (the groups' MLP weights are a few thousand floats each)
            for a, b, _, _ in runs:
                self._invalidate_touched(a, b)
            K.adam_step_spans(self.arena.params, self.arena.grads, self.exp_avg, self.exp_avg_sq,
                              [(a, b - a, lr, self.group_steps[name]) for a, b, lr, name in runs], self.algorithm,
                              self.betas[0], self.betas[1], self.eps, grad_scale, True, weight_decay=self.weight_decay)


def scaler_step(optimizer: "FusedAdam", grad_scaler, skip=(), done=()) -> bool:
    """`grad_scaler.step(optimizer)` for FusedAdam (Nerfstudio's Trainer, mixed_precision=True:
    `grad_scaler.scale(loss).backward()` -> `optimizers.optimizer_scaler_step_all(grad_scaler)` -> `grad_scaler.update()`,
    fruit_nerf_config.py:33, fruit_pipeline.py:109).  A scaled loss reaches the arena as scaled gradients (the autograd
    Functions multiply by their upstream gradient), so the step unscales inside the Adam kernel (grad_scale = 1 / scale)
    and, as torch.amp.GradScaler does, is skipped altogether when a gradient is not finite; the inf check is recorded
    with the scaler so that its own `update()` grows / backs off the scale.  -> whether the step was taken.
    The HIP kernels compute in fp32, so loss scaling protects nothing here — it is honoured, not needed; the fused loop
    (fused_train_iteration) never scales."""
    if grad_scaler is None or not grad_scaler.is_enabled():
        optimizer.step(skip=skip, done=done)
        return True
    grads = optimizer.arena.grads
    scale = float(grad_scaler.get_scale())                      # host sync, like GradScaler.step's found_inf.item()
    found_inf = (~torch.isfinite(grads).all()).to(torch.float32).reshape(1)
    try:   # what GradScaler.unscale_ / step record per optimiser, so that update() sees this inf check
        from torch.amp.grad_scaler import OptState
        state = grad_scaler._per_optimizer_states[id(optimizer)]
        state["found_inf_per_device"] = {grads.device: found_inf}
        state["stage"] = OptState.STEPPED
    except Exception as exc:   # pragma: no cover - a torch version with another GradScaler layout
        raise RuntimeError("FusedAdam cannot record its inf check with this torch.amp.GradScaler: " + repr(exc))
    if bool(found_inf.item()):
        optimizer.arena.zero_grad()                             # the step is skipped; scheduler ticks are the Trainer's
        return False
    optimizer.step(grad_scale=1.0 / scale, skip=skip, done=done)
    return True


def skipped_groups(model, optimizer: Optional["FusedAdam"] = None) -> tuple:
    """Parameter groups that got no gradient in the last training render (the proposal networks on steps that did not
    'update' them) — empty when the optimiser steps such groups anyway (FusedAdam.skip_groups_without_grad=False)."""
    if optimizer is not None and not optimizer.skip_groups_without_grad:
        return ()
    return () if bool(getattr(model, "_last_render_updated", True)) else ("proposal_networks",)


# 2 in production; the single-GPU RCCL self-test (tools/microbench/rccl_single_rank.py) lowers it to 1 so that a
# one-rank process group runs the full exchange code path (all-reduce over one rank is the identity)
EXCHANGE_MIN_WORLD = 2
# The field's gradient table is scattered and handed to the communicator in this many groups of levels.  1 (default since
# round 3): one scatter over all levels and ONE 67 MB collective — on the one-rank RCCL self-test four groups cost
# +14 us of kernels (four emit / accumulate pairs) and eight cross-stream handshakes (~12 us each on the GPU) per step;
# more groups start the first collective earlier (FNR_EXCHANGE_LEVEL_GROUPS for measurements on real peers).
EXCHANGE_LEVEL_GROUPS = max(1, int(os.environ.get("FNR_EXCHANGE_LEVEL_GROUPS", "1")))
# True: the wait for the field's collective and the table's optimiser step move from the end of step k to just before
# the field encode of step k + 1 (FruitField flushes them before any use of its parameters), so the collective also runs
# underneath the next step's pixel sampling and proposal passes.  Same arithmetic, same results.
DEFER_FIELD_UPDATE = os.environ.get("FNR_DEFER_FIELD_UPDATE") == "1"
GRAD_BUCKET_ELEMS = 4 << 20   # 16 MiB fp32 buckets: large enough for xGMI ring bandwidth, small enough to pipeline


# bench.py --gpus N (and whoever wants a self-explaining first multi-GPU run): while COLLECTIVE_LOG is a list, every
# collective of the gradient exchange is bracketed by two HIP events on the stream that issues / waits for it — recorded
# right before the call is issued and right after the waiting stream has passed its wait — and noted as
# (label, bytes, issue event, done event): elapsed_time(issue, done) is the time from issue to "the consumer may go on",
# i.e. wire time + whatever the collective had to wait for + the handshakes.  None (default): no events, no cost.
COLLECTIVE_LOG: Optional[list] = None
_pending_collectives: Dict[int, tuple] = {}


def _issued(work, label: str, nbytes: int):
    if COLLECTIVE_LOG is not None and work is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        _pending_collectives[id(work)] = (label, int(nbytes), ev)
    return work


def _wait(work) -> None:
    work.wait()
    tok = _pending_collectives.pop(id(work), None) if _pending_collectives else None
    if tok is not None and COLLECTIVE_LOG is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        COLLECTIVE_LOG.append(tok + (ev,))


def collective_report(log: list) -> list:
    """[(label, bytes, milliseconds issue -> done)] of a COLLECTIVE_LOG (synchronises the events)."""
    out = []
    for label, nbytes, a, b in log:
        b.synchronize()
        out.append((label, nbytes, float(a.elapsed_time(b))))
    return out


def start_gradient_sync(arena, span, world_size: int, bucket_elems: int = GRAD_BUCKET_ELEMS, label: str = "all_reduce"):
    """Launch the all-reduce(SUM) of arena.grads[span] as asynchronous buckets on the process group's communication
    stream (RCCL over xGMI on GPUs, gloo in the CPU tests) and return [(a, b, work)].  The caller keeps launching
    compute that does not touch that span (the proposal-network backward runs while the field gradient is in
    flight), then waits per bucket and applies Adam to it while the next bucket is still being reduced."""
    if world_size < EXCHANGE_MIN_WORLD:
        return []
    import torch.distributed as dist
    a0, b0 = span
    out = []
    for a in range(a0, b0, bucket_elems):
        b = min(a + bucket_elems, b0)
        out.append((a, b, _issued(dist.all_reduce(arena.grads[a:b], op=dist.ReduceOp.SUM, async_op=True), label, 4 * (b - a))))
    return out


def sync_gradients(arena, world_size: int) -> float:
    """DDP's gradient exchange (fruit_pipeline.py:116-118) as ONE all-reduce(SUM) over the flat gradient
    arena (RCCL over xGMI on GPUs; gloo in the CPU tests).  Returns the scale (1/world) the optimiser must
    apply — folding the mean into the Adam kernel saves a pass over the 78 MB buffer."""
    if world_size < EXCHANGE_MIN_WORLD:
        return 1.0
    import torch.distributed as dist
    dist.all_reduce(arena.grads, op=dist.ReduceOp.SUM)
    return 1.0 / world_size


# The field group's optimiser step SHARDED over the ranks ("reduce-scatter + sharded Adam + all-gather"): with
# N ranks every rank all-reduces the field's 67 MB gradient and then runs the SAME 16.8 M-parameter Adam sweep (40 B per
# parameter: 670 MB of HBM traffic, ~110 us) as every other rank.  Sharded, the gradient is REDUCE-SCATTERED (the first half
# of a ring all-reduce), each rank takes the optimiser step of its 1/N of the span only (sweep / N; its moments are the
# only ones that stay current: optimiser state / N as well), the rest of its local gradient is zeroed, and the updated
# parameters are ALL-GATHERED in place (the second half of the all-reduce): the same bytes on the links, (N - 1) / N of
# the sweep saved per rank, and the ranks' parameters are identical by construction instead of by determinism.  Same
# arithmetic per element as the all-reduce path.  The DEFAULT since round 6 (FNR_SHARDED_FIELD_OPTIMIZER=0: every rank runs
# the whole sweep behind an all-reduce): bit-identical to the all-reduce path on a one-rank RCCL group
# (tests/test_gpu_distributed.py) and on two gloo ranks (tests/test_distributed_cpu.py: gloo implements both collectives in
# place); with N ranks only 1 / N of the 88 us sweep stays on a rank's critical path.
SHARDED_FIELD_OPTIMIZER = os.environ.get("FNR_SHARDED_FIELD_OPTIMIZER", "1") != "0"
SHARD_ALIGN = 1024    # elements: every rank's shard starts on a 4 KiB boundary of the span


class _ShardedSpan:
    """One reduce-scattered span of the gradient arena: [a, main_b) is split into `world` equal shards, this rank owns
    [my_a, my_b); [main_b, b) (fewer than world * SHARD_ALIGN elements) is all-reduced like before (`tail`)."""

    def __init__(self, a, b, main_b, my_a, my_b, work, tail):
        self.a, self.b, self.main_b, self.my_a, self.my_b, self.work, self.tail = a, b, main_b, my_a, my_b, work, tail

    def finish(self, optimizer: "FusedAdam", lr: float, scale: float, group: str, step: Optional[int] = None) -> None:
        """Wait for the shard's sum, take its optimiser step, zero what this rank does not own, start the all-gather of
        the updated parameters and make the current stream wait for it (the host does not block)."""
        import torch.distributed as dist
        arena = optimizer.arena
        if self.work is not None:
            _wait(self.work)
            optimizer.step_span(self.my_a, self.my_b, lr, scale, group=group, step=step)   # (zeroes the gradient it consumes)
            if self.my_a > self.a:
                arena.grads[self.a:self.my_a].zero_()
            if self.main_b > self.my_b:
                arena.grads[self.my_b:self.main_b].zero_()
            gathered = _issued(dist.all_gather_into_tensor(arena.params[self.a:self.main_b], arena.params[self.my_a:self.my_b],
                                                           async_op=True), "all_gather(field parameters)", 4 * (self.main_b - self.a))
        else:
            gathered = None
        for ta, tb, twork in self.tail:
            _wait(twork)
            optimizer.step_span(ta, tb, lr, scale, group=group, step=step)
        if gathered is not None:
            _wait(gathered)


def start_sharded_gradient_sync(arena, span, world_size: int, rank: Optional[int] = None) -> list:
    """Reduce-scatter(SUM) of arena.grads[span], in place (a rank's output is its own slice of the input: RCCL's and
    gloo's in-place form) -> [_ShardedSpan]."""
    import torch.distributed as dist
    if rank is None:
        rank = dist.get_rank()
    a, b = span
    shard = ((b - a) // world_size) // SHARD_ALIGN * SHARD_ALIGN
    main_b = a + shard * world_size
    work = None
    my_a = my_b = a
    if shard > 0:
        my_a, my_b = a + rank * shard, a + (rank + 1) * shard
        work = _issued(dist.reduce_scatter_tensor(arena.grads[my_a:my_b], arena.grads[a:main_b], op=dist.ReduceOp.SUM,
                                                  async_op=True), "reduce_scatter(field gradient)", 4 * (main_b - a))
    tail = start_gradient_sync(arena, (main_b, b), world_size, label="all_reduce(field tail)") if b > main_b else []
    return [_ShardedSpan(a, b, main_b, my_a, my_b, work, tail)]


def finish_exchange_entry(optimizer: "FusedAdam", entry, lr: float, scale: float, group: str, step: Optional[int] = None) -> None:
    """Wait for one pending collective of the gradient exchange and apply what follows it: an all-reduced bucket
    (a, b, work) takes its optimiser step on every rank; a _ShardedSpan see there."""
    if isinstance(entry, _ShardedSpan):
        entry.finish(optimizer, lr, scale, group, step)
        return
    a, b, work = entry
    _wait(work)                                    # the stream this runs on waits for this bucket only
    optimizer.step_span(a, b, lr, scale, group=group, step=step)


def train_iteration(model, optimizer: FusedAdam, ray_bundle, batch, step: int, world_size: int = 1,
                    jitter: Optional[List[Tensor]] = None, want_metrics: bool = True, grad_scaler=None):
    """One Trainer.train_iteration for the hot path.  grad_scaler (default: the one the model was
    constructed with, fruit_pipeline.py:109): the loss is scaled before backward and the optimiser step goes through
    scaler_step(), as Nerfstudio's Trainer does under mixed_precision=True."""
    if grad_scaler is None:
        grad_scaler = getattr(model, "grad_scaler", None)
    scaling = grad_scaler is not None and grad_scaler.is_enabled()
    model.set_anneal(step)                                     # BEFORE_TRAIN_ITERATION callback
    outputs = model(ray_bundle, jitter=jitter)
    metrics_dict = model.get_metrics_dict(outputs, batch) if want_metrics else {}
    loss_dict = model.get_loss_dict(outputs, batch, metrics_dict)
    loss = sum(loss_dict.values())                             # functools.reduce(torch.add, loss_dict.values())
    (grad_scaler.scale(loss) if scaling else loss).backward()
    scale = sync_gradients(model.arena(), world_size)
    if scaling:
        if world_size >= EXCHANGE_MIN_WORLD:
            raise NotImplementedError("loss scaling with the gradient exchange: use the fused loop (it never scales)")
        scaler_step(optimizer, grad_scaler, skip=skipped_groups(model, optimizer))
        grad_scaler.update()
    else:
        optimizer.step(grad_scale=scale, skip=skipped_groups(model, optimizer))
    model.proposal_sampler.step_cb(step)                       # AFTER_TRAIN_ITERATION callback
    return loss_dict, metrics_dict


class _FieldGradientExchange:
    """Data-parallel overlap plan for the field's gradient (67 of the 78 MB arena): the hash-grid scatter runs per
    group of levels and each group's slice of the gradient table (contiguous: level l owns rows [l T, (l+1) T)) is
    handed to the communicator as soon as its scatter has been enqueued.  The rest of the group's parameters (MLP
    weights, embedding: final before the scatter starts) ride in the first / last level group's collective when they
    are adjacent to the table in the arena — one collective and two stream handshakes fewer per step."""

    def __init__(self, model, world_size: int, level_groups: Optional[int] = None):
        self.arena = model.arena()
        self.world = world_size
        self.level_groups = EXCHANGE_LEVEL_GROUPS if level_groups is None else level_groups
        self.pending = []
        table = model.field.mlp_base_grid.hash_table
        hit = [(off, n) for _, p, off, n in self.arena.entries if p is table]
        if len(hit) != 1:
            raise RuntimeError("hash table not found in the parameter arena")
        self.table_off, self.table_n = hit[0]
        self.n_levels = int(model.field.mlp_base_grid.num_levels)                   # python int (no device sync)
        self.per_level = self.table_n // self.n_levels
        self.f0, self.f1 = self.arena.group_ranges["fields"]
        self.head_sent = self.tail_sent = False
        self.hold = False          # True: levels_done / field_done only note their spans; issue() starts the collectives
        self.held = []

    def levels_done(self, level_begin: int, level_count: int) -> None:
        a = self.table_off + level_begin * self.per_level
        b = a + level_count * self.per_level
        if level_begin == 0 and not self.head_sent:
            a, self.head_sent = self.f0, True
        if level_begin + level_count == self.n_levels and not self.tail_sent:
            b, self.tail_sent = self.f1, True
        if self.hold:
            self.held.append(((a, b), max(GRAD_BUCKET_ELEMS, b - a)))
        else:
            self.pending += self._start((a, b), max(GRAD_BUCKET_ELEMS, b - a))

    def _start(self, span, bucket_elems: int) -> list:
        if SHARDED_FIELD_OPTIMIZER:
            return start_sharded_gradient_sync(self.arena, span, self.world)
        return start_gradient_sync(self.arena, span, self.world, bucket_elems=bucket_elems, label="all_reduce(field gradient)")

    def issue(self) -> None:
        """Start the collectives of the spans noted while `hold` was set (in order)."""
        for span, bucket in self.held:
            self.pending += self._start(span, bucket)
        self.held = []
        self.hold = False

    def field_done(self) -> None:
        """Whatever of the group no level collective carried (nothing when all levels went through levels_done)."""
        spans = ([] if self.head_sent else [(self.f0, self.table_off)]) + \
                ([] if self.tail_sent else [(self.table_off + self.table_n, self.f1)])
        for span in spans:
            if span[1] > span[0]:
                if self.hold:
                    self.held.append((span, GRAD_BUCKET_ELEMS))
                else:
                    self.pending += self._start(span, GRAD_BUCKET_ELEMS)


# HIP multiplexes its streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default), round robin in creation order.
# In a single process the second stream lands on its own queue; once RCCL has created its streams it may land on the
# LAUNCH stream's queue, and then nothing overlaps (round 4, rocprofv3 kernel trace of the one-rank RCCL run,
# profiles/r04_raw/kt_exchange_step.txt: every kernel of both streams on queue 0, the second stream's 150 us segment run
# ahead of the table scatter instead of next to it).  A stream of another priority gets a hardware queue of its own:
# "high" is the default when a process group exists and GPU_MAX_HW_QUEUES is below 8, "normal" otherwise.  Measured on
# the one-rank RCCL run (round 4, ms/step): colliding queues 0.989, high priority 0.930, GPU_MAX_HW_QUEUES=8 with normal
# priority 0.887 (bench.py exports that for multi-rank runs; a host application should too), single
# process 0.795.  FNR_SIDE_STREAM_PRIORITY = high | normal overrides.
SIDE_STREAM_PRIORITY = os.environ.get("FNR_SIDE_STREAM_PRIORITY", "auto")


def _second_stream(model, dev):
    side = model.__dict__.get("_side_stream")
    if side is None or side.device != dev:
        want = SIDE_STREAM_PRIORITY
        if want == "auto":
            import torch.distributed as dist
            queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4)      # (HIP reads it when the runtime starts)
            want = "high" if (dist.is_available() and dist.is_initialized() and queues < 8) else "normal"
        side = model.__dict__["_side_stream"] = torch.cuda.Stream(device=dev, priority=-1 if want == "high" else 0)
    return side


# FNR_STREAM_SAFE=1 (debug): every tensor that crosses between the launch stream and the second stream is ALSO registered
# with the consuming stream (Tensor.record_stream), so the caching allocator itself refuses to recycle its block before that
# stream is past it.  The default relies on the fork / join structure instead (see TrainingSteps.step and _ForkJoin below):
# measured +6 % step time for ~40 extra events a step.  Training must be bit-identical with the switch on and off
# (tests/test_gpu_determinism.py::test_stream_safe_mode_changes_nothing); a difference would mean a block was recycled
# under a kernel that still read it.
STREAM_SAFE = os.environ.get("FNR_STREAM_SAFE") == "1"


def _tensors_in(obj, seen, depth=0):
    if torch.is_tensor(obj):
        if obj.is_cuda and id(obj) not in seen:
            seen.add(id(obj))
            yield obj
        return
    if depth > 6 or obj is None or isinstance(obj, (str, bytes, int, float, bool)) or id(obj) in seen:
        return
    if isinstance(obj, dict):
        seen.add(id(obj))
        for v in obj.values():
            yield from _tensors_in(v, seen, depth + 1)
    elif isinstance(obj, (list, tuple)):
        seen.add(id(obj))
        for v in obj:
            yield from _tensors_in(v, seen, depth + 1)
    elif hasattr(obj, "__dict__") and not isinstance(obj, (torch.nn.Module, type)) and not callable(obj):
        seen.add(id(obj))
        for v in vars(obj).values():
            yield from _tensors_in(v, seen, depth + 1)


def crosses_to(stream, *objs) -> int:
    """STREAM_SAFE: record_stream(stream) on every device tensor reachable from objs (lists, tuples, dicts, RayBundle /
    RenderContext / RaysArg attributes) -> how many; a no-op (0) otherwise."""
    if not STREAM_SAFE or stream is None:
        return 0
    n = 0
    for t in _tensors_in(objs, set()):
        t.record_stream(stream)
        n += 1
    return n


class _ForkJoin:
    """The rule that makes two allocator pools safe without record_stream, asserted instead of assumed: every segment of
    work on the second stream opens with a wait on the launch stream (all of it, or an event recorded on it during THIS
    call) and the call may not return before the launch stream has waited for the second one — so a block of either pool
    is only ever reused by work enqueued behind every consumer of its previous contents."""

    def __init__(self, main):
        self.main, self.open, self.segments, self.joins = main, None, 0, 0

    def fork(self, side, event=None):
        # (through the C ABI, not torch.cuda.Stream.wait_*: a step program records these — fnr_stream_wait_*)
        if event is not None:
            K.stream_wait_event(side, event)
        else:
            K.stream_wait_stream(side, self.main)
        self.open = side
        self.segments += 1

    def join(self):
        if self.open is not None:
            K.stream_wait_stream(self.main, self.open)
            self.open = None
            self.joins += 1

    def check(self):
        if self.open is not None:
            raise RuntimeError("fused_forward_backward: returning with unjoined work on the second stream")


def fused_forward_backward(model, ray_bundle, batch, jitter: Optional[List[Tensor]] = None,
                           want_metrics: bool = True, exchange: Optional[_FieldGradientExchange] = None,
                           ray_grads: Optional[dict] = None, overlap_proposal_backward: bool = False,
                           table_adam=None, weight_adam=None, proposal_optimizer: Optional["FusedAdam"] = None,
                           after_ray_grads=None, serialize_streams: bool = False, ahead=None, exchange_finish=None):
    """model(ray_bundle) -> get_metrics_dict -> get_loss_dict -> sum -> backward without the autograd engine:
    the same kernels in the same order, called directly.

    The loss graph of this model is fixed (rgb + semantics through the renderer, interlevel through the proposal
    networks, unit upstream gradients), so forward, losses and backward are one straight-line sequence of HIP
    launches; per step that removes ~20 elementwise/fill/reduce launches and the engine's start-up gap that
    autograd put between them.  Gradients land in the model's arena exactly as with loss.backward().
    ray_grads: pass a dict to also receive d(loss)/d(origins) and d(loss)/d(directions) [R,3] under the keys
    "origins" / "directions" (what a camera-pose optimiser back-propagates further).
    table_adam: fnr_table_adam of the main hash table (FusedAdam.table_adam_args): its gradient is not materialised,
    the scatter's accumulate kernel applies the optimiser step to the table (single process only).
    after_ray_grads: called once the ray gradients are final (the camera optimiser's backward + step); with
    overlap_proposal_backward it runs, like the reduction of the ray gradients itself, next to the table scatter,
    which neither of them depends on.  ahead: called after it, once the proposal networks' steps are enqueued too
    (TrainingSteps' look-ahead), on the second stream.
    serialize_streams (with overlap_proposal_backward): the same launches on the same two streams (same allocator pools),
    but each stream waits for everything the other has enqueued — nothing runs concurrently (bench.py: the steps whose
    launches it brackets with HIP events)."""
    from . import _lib as L
    cfg = model.config
    dev = model.device
    with torch.no_grad():
        ray_bundle = model._collide(ray_bundle)
        image, mask = batch["image"].to(dev), batch["fruit_mask"].to(dev)
        # (single process only: on the exchange path — one-rank RCCL A/B, profiles/r05_raw/exchange_losses_ab.log — it is
        #  neutral to slightly negative, 0.860 - 0.870 against 0.857 ms/step: the second stream has the collectives' tail there)
        losses_on_side = bool(LOSSES_ON_SIDE and overlap_proposal_backward and not serialize_streams and exchange is None)
        # ... and then the compositing launch runs its own backward too (FUSE_COMPOSITE_BACKWARD): nothing between the field's
        # forward and backward pass depends on another launch
        fuse_composite = bool(FUSE_COMPOSITE_BACKWARD and losses_on_side)
        outputs, rctx = model._render(ray_bundle, jitter, save_input_jacobian=ray_grads is not None,
                                      loss_targets=(image, mask, cfg.semantic_loss_weight) if fuse_composite else None)
        rays, fin = rctx.rays, rctx.levels[-1]
        S = fin["S"]
        # one fill for everything this step accumulates into: the loss slots (+ completion counter) and, with a camera
        # optimiser, the ray gradients d(loss)/d(origins | directions); one launch for every loss and metric
        # the loss / metric accumulator is persistent: fnr_train_losses leaves it zeroed (its last workgroup cleans up),
        # so a step launches no fill; the ray gradients are WRITTEN by the one reduction over all their sources
        n_slots = L.FNR_TRAIN_LOSSES_ACCUM_FLOATS
        accum = model.__dict__.get("_loss_accum")
        if accum is None or accum.device != dev:
            accum = model.__dict__["_loss_accum"] = torch.zeros(n_slots, device=dev)
        # on steps that train the proposal networks their weights backward (unit upstream) rides along: the list
        # then holds d(loss)/d(density) per level instead of d(loss)/d(weights)
        prop_bwd = bool(rctx.training and rctx.updated)
        prop_levels = [(lv["S"], lv["spacing"], lv["weights"]) + ((lv["euclid"], lv["density"]) if prop_bwd else ())
                       for lv in rctx.levels[:-1]]
        # The loss VALUES (five scalars for the log) and the interlevel loss with the proposal levels' weights backward feed
        # the second stream's work only; what the field backward needs from the losses — the per-ray gradients of MSE and
        # BCE — the composite backward forms itself (fnr_composite_bwd_targets: same expressions, same bits).  So with two
        # streams the losses launch (~30 us, latency-bound) opens the SECOND stream's segment, next to the composite + MLP
        # backward instead of ahead of them (round 5; LOSSES_ON_SIDE).
        main = torch.cuda.current_stream(dev)
        fj = _ForkJoin(main)
        side = None
        if losses_on_side:
            side = _second_stream(model, dev)
            fj.fork(side)                       # behind the forward pass
            crosses_to(side, outputs, image, mask, rctx, accum)
        with torch.cuda.stream(side) if losses_on_side else contextlib.nullcontext():
            losses, d_rgb, d_sem, d_wps = K.train_losses(outputs["rgb"], image, outputs["semantics"], mask,
                                                         cfg.semantic_loss_weight, S, fin["spacing"], fin["weights"],
                                                         prop_levels, cfg.interlevel_loss_mult, want_metrics,
                                                         accum, fuse_weights_bwd=prop_bwd,
                                                         want_ray_grads=not losses_on_side)
        loss_dict = {"rgb_loss": losses[0], "semantics_loss": losses[1], "interlevel_loss": losses[3]}
        metrics_dict = {"psnr": losses[2], "distortion": losses[4]} if want_metrics else {}

        # ---- backward (what loss.backward() runs through _LossFn, _RenderFn, _InterlevelFn) ----
        arena = model.arena()
        arena.reattach_grads()
        d_o = d_d = None
        ray_sources = [] if ray_grads is not None else None     # (warp, euclid, S, partial) of every chain, in order
        if ray_grads is not None:
            both = K._empty(2, rays.n, 3, device=dev)
            d_o, d_d = ray_grads["origins"], ray_grads["directions"] = both[0], both[1]
        # The proposal-network backward (interlevel loss) and the field backward (rgb + semantic losses) share no
        # buffers.  overlap_proposal_backward=True runs the former on a second HIP stream so that its ~14 small/medium
        # launches fill the gaps and tails of the field kernels (measured: -3 % on the steps that have one); bench.py
        # serialises the two streams on the steps whose launches it brackets with HIP events (serialize_streams: a
        # duration measured while another stream's kernels share the CUs describes neither kernel).
        up = None   # d_wps is d(loss)/d(density) already (fuse_weights_bwd above)
        # The ray gradients' sources are complete once the MLP backward is (its d_pos) and the proposal chain's MLP
        # backwards have run; their reduction and the camera optimiser's ~30 us of small launches do not need the
        # scatters: single process + second stream -> they run next to one
        # ... and with a gradient exchange the same segment holds what follows the collectives of the small groups
        # (after_ray_grads = fused_train_iteration's exchange_tail: pose gradient + its all-reduce, the proposal networks'
        # all-reduce, their waits and optimiser steps), so that N > 1 keeps the schedule N = 1 has
        tail_on_side = bool(overlap_proposal_backward and ((ray_grads is not None and exchange is None) or
                                                           (exchange is not None and after_ray_grads is not None)))
        sources_early = False   # the proposal chain recorded `pos_ready` ahead of its scatter
        if prop_bwd:
            if overlap_proposal_backward:
                if not losses_on_side:          # (else: the segment is open already, the losses launch heads it)
                    side = _second_stream(model, dev)
                    fj.fork(side)
                crosses_to(side, rctx, d_wps, d_o, d_d)
                pos_ready = None
                if tail_on_side and not serialize_streams and exchange is None:
                    pos_ready = model.__dict__.get("_pos_ready_event")
                    if pos_ready is None:
                        pos_ready = model.__dict__["_pos_ready_event"] = K.Event()
                with torch.cuda.stream(side):
                    sources_early = bool(_proposal_backward(model, rctx, d_wps, up, d_o, d_d, collect=ray_sources,
                                                            optimizer=proposal_optimizer, position_ready=pos_ready))
                crosses_to(main, ray_sources)
                if serialize_streams:
                    K.stream_wait_stream(main, side)
        if getattr(rctx, "composite_grads", None) is not None:
            d_density, d_rgb_s, d_logit = rctx.composite_grads
        elif losses_on_side:
            d_density, d_rgb_s, d_logit = K.composite_bwd_targets(rays, S, fin["euclid"], rctx.sample_density,
                                                                  rctx.sample_rgb, rctx.weights, outputs["rgb"], image,
                                                                  outputs["semantics"], mask, cfg.semantic_loss_weight)
        else:
            d_density, d_rgb_s, d_logit = K.composite_bwd(rays, S, fin["euclid"], rctx.sample_density, rctx.sample_rgb,
                                                          rctx.weights, d_rgb, d_sem)
        fld = model.field
        net, gnet = fld.net_struct(), fld.net_struct(grads=True)
        d_pos = None
        if ray_grads is not None and rctx.field_jacobian is not None:
            d_feats, d_pos = K.field_mlp_bwd(net, gnet, rays, S, rctx.field_feats, rctx.field_h, rctx.field_selector,
                                             d_density, d_rgb_s, d_logit, jacobian=rctx.field_jacobian,
                                             weight_adam=weight_adam)
        else:
            d_feats = K.field_mlp_bwd(net, gnet, rays, S, rctx.field_feats, rctx.field_h, rctx.field_selector, d_density,
                                      d_rgb_s, d_logit, weight_adam=weight_adam)
        field_source = None
        if ray_grads is not None:
            if d_pos is not None:
                field_source = (fld.warp_struct(), fin["euclid"], S, d_pos)
            else:
                # no saved Jacobian (the model is not in training mode): the gather path reads the TABLE, which the
                # fused scatter below may update in place — gather now
                partial = K.hash_encode_input_grad(net.grid, fld.warp_struct(), rays, fin["euclid"], S, d_feats)
                field_source = (fld.warp_struct(), fin["euclid"], S, partial)
        if tail_on_side and sources_early:
            # a step that trains the proposal networks: the second stream is the longer chain (their backward, then the
            # look-ahead that needs their step AND the cameras'), so the reduction + camera step go to THIS stream, between
            # the MLP backward and the table scatter, and the look-ahead only waits for them and for the proposal scatter
            K.stream_wait_event(main, pos_ready)
            K.position_grad_reduce_multi(ray_sources + [field_source], rays, d_o, d_d, accumulate=False)
            if after_ray_grads is not None:
                after_ray_grads()
            tail_ready = model.__dict__.get("_tail_ready_event")
            if tail_ready is None:
                tail_ready = model.__dict__["_tail_ready_event"] = K.Event()
            tail_ready.record(main)                # the cameras have taken their step
        elif tail_on_side:
            tail_ready = model.__dict__.get("_tail_ready_event")
            if tail_ready is None:
                tail_ready = model.__dict__["_tail_ready_event"] = K.Event()
            tail_ready.record(main)                # the MLP backward (its d_pos) is enqueued
        def scatter():
            if exchange is None and table_adam is not None:
                K.hash_encode_bwd_adam(gnet.grid, fld.warp_struct(), rays, fin["euclid"], S, d_feats, table_adam)
            elif exchange is None:
                K.hash_encode_bwd(gnet.grid, fld.warp_struct(), rays, fin["euclid"], S, d_feats)
            else:
                n_lv = int(gnet.grid.n_levels)
                per = -(-n_lv // max(1, exchange.level_groups))
                for lb in range(0, n_lv, per):
                    cnt = min(per, n_lv - lb)
                    K.hash_encode_bwd(gnet.grid, fld.warp_struct(), rays, fin["euclid"], S, d_feats, lb, cnt)
                    exchange.levels_done(lb, cnt)      # this slice of the gradient table is final: all-reduce it now
                exchange.field_done()

        def tail(first=True, last=True):
            """The second stream's segment; with a gradient exchange in two parts: first = up to the issue of the small
            collectives (after_ray_grads), last = their waits + optimiser steps (exchange_finish) and the look-ahead."""
            if side is None:                       # a step without a proposal backward: the event is the whole fork
                side_ = _second_stream(model, dev)
            else:
                side_ = side
            if first:
                # behind the scatter (serialize_streams) instead of underneath it
                fj.fork(side_, None if serialize_streams else tail_ready)
                crosses_to(side_, ray_sources, field_source, d_o, d_d, rays)
            with torch.cuda.stream(side_):
                if first and not sources_early:
                    if ray_grads is not None:
                        K.position_grad_reduce_multi(ray_sources + [field_source], rays, d_o, d_d, accumulate=False)
                    if after_ray_grads is not None:
                        after_ray_grads()
                if last:
                    if exchange_finish is not None:
                        exchange_finish()
                    if ahead is not None:
                        ahead()

        if tail_on_side and exchange is not None and not serialize_streams and exchange.level_groups <= 1:
            # Gradient exchange, host order (the host is only ~0.2 ms ahead of the GPU on this path, so what it enqueues
            # first matters): (1) the scatter's launches — the critical chain; (2) the second stream's segment up to the
            # ISSUE of the small collectives (proposal networks 10.5 MB, poses 2 KB), so that they precede the field's
            # 67 MB on the communicator's stream, which runs collectives in issue order; (3) the field's collective — the
            # scatter is still running on the GPU; (4) the small groups' waits + optimiser steps and the look-ahead.
            # (Enqueueing the whole segment before the scatter left the launch stream idle for ~60 us per step, round 4
            # kernel trace profiles/r04_raw/kt_exchange_high_step.txt.)
            exchange.hold = True
            scatter()
            tail(first=True, last=False)
            exchange.issue()
            tail(first=False, last=True)
            fj.join()
            fj.check()
            return loss_dict, metrics_dict
        scatter()
        if tail_on_side:
            tail()
            fj.join()                              # every local of this call outlives the second stream's launches
            fj.check()
            return loss_dict, metrics_dict
        if side is not None:
            fj.join()                              # proposal gradients (and their ray-gradient sources) are final
            for src in ray_sources or ():
                src[3].record_stream(main)
        elif prop_bwd:
            _proposal_backward(model, rctx, d_wps, up, d_o, d_d, level_streams=PROPOSAL_LEVEL_STREAMS,
                               collect=ray_sources, optimizer=proposal_optimizer)
        if ray_grads is not None:
            # one launch: proposal levels first, the field last (the order the separate launches added them in)
            K.position_grad_reduce_multi(ray_sources + [field_source], rays, d_o, d_d, accumulate=False)
        if after_ray_grads is not None:
            after_ray_grads()
        if exchange_finish is not None:
            exchange_finish()
        if ahead is not None:
            ahead()
        fj.check()
    return loss_dict, metrics_dict


def camera_backward(camera_optimizer, batcher, ray_grads: dict, world_size: int = 1):
    """The datamanager side of loss.backward() for the camera-pose optimiser (fruit_nerf_config.py:39-43): ray
    gradients -> pose_adjustment.grad (camera_opt.hip).  With several ranks the 6 x num_cameras gradient is averaged
    like the model's (nerfstudio leaves the datamanager outside DDP, which lets the ranks' poses drift apart; one 2 KB
    all-reduce keeps them identical); it is issued asynchronously -> (work | None, gradient scale)."""
    d = batcher.last_draw
    pose = camera_optimizer.pose_adjustment
    K.camera_pose_grad(batcher._set, batcher.image_ids, d["u"], d["cam"], pose.data, d["c2w_adjusted"],
                       ray_grads["origins"], ray_grads["directions"], pose.grad)
    if world_size >= EXCHANGE_MIN_WORLD:
        import torch.distributed as dist
        return _issued(dist.all_reduce(pose.grad, op=dist.ReduceOp.SUM, async_op=True), "all_reduce(poses)",
                       4 * pose.grad.numel()), 1.0 / world_size
    return None, 1.0


def camera_backward_and_step(camera_optimizer, camera_adam, batcher, ray_grads: dict, world_size: int = 1) -> None:
    """camera_backward() + the camera optimiser's Adam step (one launch in a single process)."""
    # (the draw may come from the other stream's pool: the look-ahead draws on the second stream, this runs on either)
    crosses_to(torch.cuda.current_stream(ray_grads["origins"].device), batcher.last_draw, ray_grads)
    if world_size < EXCHANGE_MIN_WORLD and FUSE_CAMERA_OPTIMIZER:
        d = batcher.last_draw
        pose = camera_optimizer.pose_adjustment
        K.camera_pose_grad_adam(batcher._set, batcher.image_ids, d["u"], d["cam"], d["c2w_adjusted"],
                                ray_grads["origins"], ray_grads["directions"], pose.grad, camera_adam.fused_step_args())
        return
    work, scale = camera_backward(camera_optimizer, batcher, ray_grads, world_size)
    if work is not None:
        _wait(work)
    camera_adam.step(grad_scale=scale)


# proposal-network backward on a second HIP stream (see fused_forward_backward): same bits (tests/test_gpu_determinism.py,
# test_proposal_backward_on_a_second_stream_changes_nothing), -3 % step time on the steps that train the proposal networks.
# FNR_OVERLAP_PROPOSAL_BACKWARD=0 keeps every launch on one stream (per-kernel timings attributable to one kernel:
# what the profiles under profiles/ and the profiled steps of bench.py use)
OVERLAP_PROPOSAL_BACKWARD = os.environ.get("FNR_OVERLAP_PROPOSAL_BACKWARD", "1") != "0"
# with the second stream: same launches on the same streams, but nothing concurrent (fused_forward_backward's
# serialize_streams): a step like this gives per-kernel durations that describe one kernel, without switching allocator
# pools between steps (one-stream and two-stream steps alternating made the caching allocator grow both pools: hipMalloc
# calls inside bench.py's timed window)
SERIALIZE_STREAMS = os.environ.get("FNR_SERIALIZE_STREAMS") == "1"   # (profiling runs: tools/gpu_call.sh, legs kt / pmc)
# The proposal levels' backward chains next to each other (level 0 on the launch stream, level 1 on a side stream; they
# share no buffers).  OFF: measured on MI355X (round 3, A/B on one box) the step gets 4 % SLOWER (0.908 -> 0.943 ms) — a
# cross-stream fork + join costs ~12 us of GPU time per handshake on this stack and the two chains of latency-bound
# kernels slow each other in the XCDs' L2s (each network's 5 MB tables fit one L2, two do not).  Kept for measurements.
PROPOSAL_LEVEL_STREAMS = os.environ.get("FNR_PROPOSAL_LEVEL_STREAMS", "0") == "1"
PAIR_PROPOSAL_LEVELS = os.environ.get("FNR_PAIR_PROPOSAL_LEVELS", "1") != "0"   # see _proposal_backward
# The losses launch on the second stream, the composite backward forming its own per-ray loss gradients (see
# fused_forward_backward).  FNR_LOSSES_ON_SIDE=0: losses on the launch stream ahead of the backward, as before (A/B).
LOSSES_ON_SIDE = os.environ.get("FNR_LOSSES_ON_SIDE", "1") != "0"
# The compositing launch of a training step runs its own backward (fnr_composite_fwd_bwd_targets) when the losses launch is on
# the second stream.  FNR_FUSE_COMPOSITE_BACKWARD=0: two launches (A/B); same bits either way.
FUSE_COMPOSITE_BACKWARD = os.environ.get("FNR_FUSE_COMPOSITE_BACKWARD", "1") != "0"
FUSE_CAMERA_OPTIMIZER = True  # single process: the pose table's optimiser step runs inside the pose-gradient kernel
FUSE_WEIGHT_OPTIMIZER = True  # ... and the field's MLP weights + embedding step inside k_reduce_dw / k_embedding_grad
FUSE_TABLE_OPTIMIZER = True   # single process: the main hash table's Adam / RAdam step runs inside the scatter


def fused_train_iteration(model, optimizer: FusedAdam, ray_bundle, batch, step: int, world_size: int = 1,
                          jitter: Optional[List[Tensor]] = None, want_metrics: bool = True, camera=None,
                          fuse_table_optimizer: Optional[bool] = None, ahead=None):
    """train_iteration() on fused_forward_backward(); returns the same (loss_dict, metrics_dict) tensors.

    ahead: called with no arguments once the proposal networks' and the cameras' optimiser steps of this iteration are
    enqueued — the point from which the next iteration's rays and proposal sampling can be enqueued (TrainingSteps).

    world_size > 1 (DDP semantics, fruit_pipeline.py:116-118): the field's gradient (67 MB of the 78 MB arena) is
    all-reduced in 16 MiB buckets on the communication stream, each bucket = the table rows of 4 levels, issued as
    soon as those levels' scatter has been enqueued — underneath the remaining scatter groups and the whole
    proposal-network backward; Adam then consumes bucket k while bucket k+1 is still on the wire."""
    model.set_anneal(step)                                     # BEFORE_TRAIN_ITERATION callback
    arena = model.arena()
    spans = arena.group_ranges
    exchange = _FieldGradientExchange(model, world_size) if world_size >= EXCHANGE_MIN_WORLD else None
    ray_grads = {} if camera is not None else None   # camera = (CameraOptimizer, CameraAdam, PixelBatcher)
    # Without a gradient exchange the main table's optimiser step is fused into the scatter's accumulate kernel (its
    # 16.8 M parameters are 86 % of the arena: 40 -> 24 bytes of HBM traffic per parameter and step, bit-identical
    # results); the table's gradient stays zero.
    fuse = FUSE_TABLE_OPTIMIZER if fuse_table_optimizer is None else fuse_table_optimizer
    table_adam, weight_adam, done = None, None, ()
    if exchange is None and fuse:
        table_adam, span = optimizer.table_adam_args(model.field.mlp_base_grid.hash_table, "fields")
        done = (span,)
        if FUSE_WEIGHT_OPTIMIZER:
            # ... and the rest of the "fields" group (MLP weights, embedding) in the kernels that finish ITS gradients:
            # the whole group is done when the backward returns, and a step that does not train the proposal networks
            # ends without an optimiser launch
            weight_adam, span = optimizer.weight_adam_args("fields")
            done = (span,)
    # ... and the proposal networks' on the steps that train them (one network per level: a shared network needs both
    # levels' gradients summed before its step)
    prop_opt = None
    if exchange is None and fuse and FUSE_WEIGHT_OPTIMIZER and not model.config.use_same_proposal_network \
            and model.training and model.proposal_sampler.updated_now():
        prop_opt = optimizer
        done = done + (tuple(spans["proposal_networks"]),)
    # The look-ahead reads the proposal networks: it may only be enqueued from inside the backward when their step of THIS
    # iteration is (fused into their backward) or does not happen at all (no gradient -> the optimiser skips the group);
    # otherwise (unfused configurations, torch-1.13 stepping) it follows optimizer.step() below
    prop_step_pending = prop_opt is None and (bool(model.training and model.proposal_sampler.updated_now())
                                              or not optimizer.skip_groups_without_grad)
    ahead_early = ahead if (ahead is not None and not prop_step_pending) else None
    camera_step = None
    if exchange is None and camera is not None:
        def camera_step():   # the datamanager's backward + optimiser step, as soon as it can run
            with torch.no_grad():
                camera_backward_and_step(camera[0], camera[1], camera[2], ray_grads, world_size)
    lrs = exchange_finish = None
    if exchange is not None:
        # The exchange path's counterpart of the single-process tail (round 4): everything that follows the collectives of
        # the SMALL groups — the proposal networks' 10.5 MB on the steps that train them, the 2 KB pose gradient — runs in
        # the second stream's segment of fused_forward_backward (their all-reduces are issued there, ahead of the field's
        # 67 MB; their waits, 1 / world + optimiser steps and the look-ahead follow on that stream), next to the table
        # scatter.  The update schedule is a function of the step, identical on every rank: on steps that do not train the
        # proposal networks their gradients are zero everywhere, their exchange is skipped and — as in the reference
        # (grad = None -> torch.optim skips them) — so is their optimiser step.
        prop_updated = bool(model.training and model.proposal_sampler.updated_now())    # what _render is about to see
        prop_stepped = prop_updated or not optimizer.skip_groups_without_grad
        lrs = optimizer.begin_step(skip=() if prop_stepped else ("proposal_networks",))
        scale = 1.0 / world_size

        small_groups = {}

        def camera_step():   # noqa: F811  (the exchange path's first half of the tail: issue the small collectives)
            with torch.no_grad():
                small_groups["prop"] = start_gradient_sync(arena, spans["proposal_networks"], world_size,
                                                           label="all_reduce(proposal networks)") if prop_updated else []
                small_groups["cam"] = (camera_backward(camera[0], camera[2], ray_grads, world_size)
                                       if camera is not None else (None, 1.0))

        def exchange_finish():   # ... second half: their waits, 1 / world + optimiser steps
            with torch.no_grad():
                for a, b, work in small_groups["prop"]:
                    _wait(work)                                # the stream this runs on waits for this bucket only
                    optimizer.step_span(a, b, lrs["proposal_networks"], scale, group="proposal_networks")
                if prop_stepped and not prop_updated:   # torch < 2.0 semantics: zero gradients everywhere, still a step
                    pa, pb = spans["proposal_networks"]
                    optimizer.step_span(pa, pb, lrs["proposal_networks"], scale, group="proposal_networks")
                if camera is not None:
                    cam_work, cam_scale = small_groups["cam"]
                    if cam_work is not None:
                        _wait(cam_work)
                    camera[1].step(grad_scale=cam_scale)
    loss_dict, metrics_dict = fused_forward_backward(model, ray_bundle, batch, jitter, want_metrics, exchange,
                                                     ray_grads, overlap_proposal_backward=OVERLAP_PROPOSAL_BACKWARD,
                                                     table_adam=table_adam, weight_adam=weight_adam,
                                                     proposal_optimizer=prop_opt, after_ray_grads=camera_step,
                                                     serialize_streams=SERIALIZE_STREAMS,
                                                     ahead=ahead_early if exchange is None else ahead,
                                                     exchange_finish=exchange_finish if exchange is not None else None)
    with torch.no_grad():
        if exchange is None:
            optimizer.step(skip=skipped_groups(model, optimizer), done=done)
            if ahead is not None and ahead_early is None:
                ahead()
        else:
            assert prop_updated == bool(getattr(model, "_last_render_updated", True)), "proposal update schedule"
            pending = list(exchange.pending)                   # the field's buckets (issued behind the scatter)
            deferred = []
            for entry in pending:
                a = entry.a if isinstance(entry, _ShardedSpan) else entry[0]
                name = "fields" if a >= spans["fields"][0] else "proposal_networks"
                if DEFER_FIELD_UPDATE and name == "fields":
                    deferred.append(entry)
                    continue
                finish_exchange_entry(optimizer, entry, lrs[name], scale, name)   # the compute stream waits for this bucket only
            if deferred:
                lr_f, step_f = lrs["fields"], optimizer.group_steps["fields"]

                def finish(deferred=deferred, lr_f=lr_f, step_f=step_f, scale=scale):
                    with torch.no_grad():
                        for entry in deferred:
                            finish_exchange_entry(optimizer, entry, lr_f, scale, "fields", step=step_f)
                model.field.defer_update(finish)
    model.proposal_sampler.step_cb(step)                       # AFTER_TRAIN_ITERATION callback
    return loss_dict, metrics_dict


# The next iteration's rays and proposal sampling, enqueued at the end of the current one (TrainingSteps): -6 % step time
# with the second stream (they run underneath the main table's scatter), bit-identical training either way
# (tests/test_gpu_determinism.py).  FNR_SAMPLE_AHEAD=0: every iteration samples at its own start.
SAMPLE_AHEAD = os.environ.get("FNR_SAMPLE_AHEAD", "1") != "0"


# Native step sequencer (round 6): a step whose launch sequence has been seen before is REPLAYED by one call of the C ABI
# (fnr_program_replay) instead of being re-interpreted — ~30 entry-point calls, 48 buffer allocations, 150 pointer
# conversions and the stream bookkeeping of this file collapse into a dictionary lookup, a 64-byte struct of per-step
# scalars and one ctypes call (host enqueue 0.53 -> ~0.2 ms/step; what is left is the HIP runtime's launch cost).  Same
# entry points, same arguments, same streams, same order: bit-identical training (tests/test_gpu_sequencer.py).
# FNR_NATIVE_SEQUENCER=0: every step is interpreted (A/B, debugging).
NATIVE_SEQUENCER = os.environ.get("FNR_NATIVE_SEQUENCER", "1") != "0"
# a step shape is recorded on its n-th interpreted occurrence, never before the loop's third step: the first two create
# lazily-built state — second stream, events, touched bitmaps, collider planes, persistent workspaces — through launches a
# recording must not contain (a scatter workspace's first use is also caught per step: K.fresh_workspaces)
RECORD_ON_OCCURRENCE = 1
RECORD_FROM_STEP = 2


class _StepProgram:
    """One recorded step (fnr_program_*) + what the host has to restore around a replay."""

    def __init__(self):
        import ctypes as C
        h = C.c_void_p()
        L.check(L.load().fnr_program_create(C.byref(h)), "program_create")
        self.handle = h.value
        self.keep = []              # every tensor / struct the recorded calls point at
        self.next = None            # (RayBundle, batch) the recorded look-ahead produced (fixed addresses)
        self.last_draw = self.last_presample = None
        self.next_offset = 0        # bump offset of the other arena behind the look-ahead's buffers
        self.n_ops = 0

    def __del__(self):
        try:
            if self.handle and L._lib is not None:
                L.load().fnr_program_destroy(self.handle)
        except Exception:   # interpreter shutdown
            pass


class TrainingSteps:
    """The training loop's body for one model: batcher.sample -> fused_train_iteration, with the start of iteration
    i + 1 — fnr_train_prologue (pixels, corrected cameras, rays, level-0 bins, jitters) and the proposal sampler's levels
    (FruitModel.sample_ahead) — enqueued at the END of iteration i, as soon as the proposal networks' and the cameras'
    optimiser steps are: with training.OVERLAP_PROPOSAL_BACKWARD on the second HIP stream, underneath the main table's
    scatter + optimiser step, which they do not depend on.  Same launches on the same values as sampling at the start
    of iteration i + 1 (counter-based random numbers, schedule flags checked by the model), so the parameters after any
    number of steps are bit-identical with SAMPLE_AHEAD off.

    Single process on a HIP device: the per-step buffers come from two step arenas (K.StepArena; iteration i uses arena
    i & 1, whose first bytes hold what iteration i - 1 sampled ahead for it), so a step shape — (parity, does this step
    train the proposal networks, does the next one, metrics, camera optimiser, arithmetic) — always runs on the same
    addresses; its second interpreted occurrence is recorded (fnr_program_begin / _end) and every later one replayed by
    fnr_program_replay with the step's scalars (NATIVE_SEQUENCER).  The loss tensors a step returns are fresh
    allocations; everything else a step produced lives in its arena until the step after next.

    camera: (CameraOptimizer, CameraAdam) or None; the batcher is the third member fused_train_iteration wants."""

    def __init__(self, model, optimizer: FusedAdam, batcher, n_rays: int, camera=None, world_size: int = 1):
        self.model, self.optimizer, self.batcher, self.n_rays = model, optimizer, batcher, int(n_rays)
        self.camera = (camera[0], camera[1], batcher) if camera is not None else None
        self.world_size = world_size
        self.step_idx = 0
        self._next = None
        self._next_version = ()
        # step arenas + programs (single process, HIP device)
        self._arenas = None
        self._arena_version = 0
        self._next_layout = None        # (arena version, parity, offset behind the look-ahead, clean) of what _next points into
        self._programs = {}             # key -> _StepProgram
        self._unrecordable = {}         # key -> reason
        self._seen = {}                 # key -> interpreted occurrences
        self._scalars = L.fnr_step_scalars()
        self.stats = {"replayed": 0, "interpreted": 0, "recorded": 0, "record_failed": 0, "arena_grown": 0}

    # ---- look-ahead ----------------------------------------------------------------------------------------------
    def _draw(self, finishing_step: Optional[int]):
        from .rays import RayBundle
        level0 = self.model.level0_spec()
        n_train = getattr(getattr(self.batcher, "image_ids", None), "numel", lambda: 0)()
        if self.camera is not None and self.n_rays < n_train:
            level0 = None     # fnr_train_prologue adjusts one camera per ray slot: fewer rays than cameras -> separate launches
        o, d, cam, batch = self.batcher.sample(self.n_rays, self.camera[0] if self.camera else None, level0=level0)
        rb = RayBundle(o, d, None, cam, presampled=self.batcher.last_presample)
        if finishing_step is not None:
            self.model.sample_ahead(rb, finishing_step)
        return rb, batch

    def drop_lookahead(self) -> None:
        """Forget what was sampled ahead (the proposal networks or the cameras were changed from outside, or the
        batcher was used in between): the next step() samples at its start.  The draw itself is consumed."""
        self._next = None
        self._next_layout = None

    def drop_programs(self) -> None:
        """Forget every recorded step program (anything a program captured BY VALUE was changed from outside: loss
        weights, near / far planes, the batcher's image set ...; pointers and the scheduled scalars are tracked)."""
        self._programs.clear()
        self._unrecordable.clear()
        self._seen.clear()

    def _outside_version(self) -> tuple:
        """Version counters of what the prologue of a look-ahead read besides the proposal networks: the camera poses."""
        if self.camera is None:
            return ()
        pose = getattr(self.camera[0], "pose_adjustment", None)
        return () if pose is None else (pose.data_ptr(), pose._version)

    # ---- step arenas ---------------------------------------------------------------------------------------------
    def _arena_mode(self) -> bool:
        dev = getattr(self.model, "device", None)
        # (FNR_STREAM_SAFE is about the caching allocator's view of the cross-stream tensors: that mode keeps the allocator)
        return bool(NATIVE_SEQUENCER and not STREAM_SAFE and self.world_size < EXCHANGE_MIN_WORLD and dev is not None
                    and dev.type == "cuda" and hasattr(self.model, "proposal_sampler"))

    def _ensure_arenas(self) -> None:
        need = 0
        if self._arenas is not None:
            for a in self._arenas:
                if a.overflow_bytes:
                    need = max(need, int((a.high_water + a.overflow_bytes) * 1.25) + (32 << 20))
            if not need:
                return
            self.stats["arena_grown"] += 1
        else:
            cfg = self.model.config
            s_prop = sum(cfg.num_proposal_samples_per_ray)
            need = int(self.n_rays * (cfg.num_nerf_samples_per_ray * 900 + s_prop * 80) * 1.3) + (32 << 20)
        dev = self.model.device
        self._arenas = [K.StepArena(dev, need), K.StepArena(dev, need)]
        self._arena_version += 1
        self._programs.clear()          # they point into the old slabs (which live on while a program or _next holds them)
        self._seen.clear()
        self._next_layout = None

    def _program_key(self, parity: int, start_offset: int, updated: bool, updated_next: bool, want_metrics: bool) -> tuple:
        model = self.model
        fld = model.field
        cfg = model.config
        dev = model.device
        side = model.__dict__.get("_side_stream")
        return (self._arena_version, parity, start_offset, updated, updated_next, bool(want_metrics), self.camera is not None,
                getattr(fld, "mlp_precision", None), model.arena().params.data_ptr(), id(self.optimizer),
                L.stream_ptr(dev), None if side is None else side.cuda_stream,
                OVERLAP_PROPOSAL_BACKWARD, SERIALIZE_STREAMS, LOSSES_ON_SIDE, PAIR_PROPOSAL_LEVELS, FUSE_CAMERA_OPTIMIZER,
                FUSE_WEIGHT_OPTIMIZER, FUSE_TABLE_OPTIMIZER, SPARSE_TOUCH_SKIPPING, STREAM_SAFE, FUSE_COMPOSITE_BACKWARD,
                cfg.semantic_loss_weight, cfg.interlevel_loss_mult, cfg.near_plane, cfg.far_plane)

    # ---- one step ------------------------------------------------------------------------------------------------
    def step(self, want_metrics: bool = True):
        if self._next is not None and self._next[0] == self.step_idx and self._next_version != self._outside_version():
            # the cameras were edited between two steps: the rays drawn ahead used the old poses.  Draw again with the
            # SAME counter (counter-based random numbers: the same pixels through the new poses — what sampling at the
            # start of this step would have drawn)
            self._next = None
            self._next_layout = None
            if getattr(self.batcher, "_offset", 0) > 0:
                self.batcher._offset -= 1
        if not self._arena_mode():
            return self._step_interpreted(want_metrics, None)
        model = self.model
        sampler = model.proposal_sampler
        step = self.step_idx
        parity = step & 1
        self._ensure_arenas()
        arena = self._arenas[parity]
        have_next = self._next is not None and self._next[0] == step
        in_layout = have_next and self._next_layout is not None and self._next_layout[:2] == (self._arena_version, parity)
        if in_layout:
            arena.offset = self._next_layout[2]      # continue behind what the previous step sampled ahead into this arena
        else:
            arena.reset()
        key = None
        if in_layout and self._next_layout[3] and model.training and SAMPLE_AHEAD and not L.profile_recording():
            ahead = self._next[1].presampled.get("ahead") if self._next[1].presampled else None
            updated = sampler.updated_now()
            if ahead is not None and ahead["updated"] == updated and ahead["anneal"] == model.anneal_at(step) \
                    and ahead.get("param_version") == model.lookahead_version():
                # what the look-ahead of THIS step will see: sample_ahead() runs behind _render, which restarts the count
                # of steps since the last update on a step that updates (ProposalNetworkSampler.updated_after)
                since = 0 if updated else sampler._steps_since_update
                updated_next = bool(since + 1 > sampler.update_sched(step) or step < 10)
                key = self._program_key(parity, arena.offset, updated, updated_next, want_metrics)
        if key is not None:
            prog = self._programs.get(key)
            if prog is not None:
                return self._replay(prog, step, want_metrics, updated)
            n = self._seen[key] = self._seen.get(key, 0) + 1
            if n >= RECORD_ON_OCCURRENCE and step >= RECORD_FROM_STEP and key not in self._unrecordable:
                return self._step_interpreted(want_metrics, arena, record_key=key)
        return self._step_interpreted(want_metrics, arena)

    def _step_interpreted(self, want_metrics: bool, arena, record_key=None):
        if self._next is not None and self._next[0] == self.step_idx:
            _, rb, batch = self._next
        else:
            prev = K.use_arena(arena)
            try:
                rb, batch = self._draw(None)
            finally:
                K.use_arena(prev)
        self._next = None
        self._next_layout = None
        step = self.step_idx
        other = self._arenas[1 - (step & 1)] if arena is not None else None
        launch_stream = torch.cuda.current_stream(self.model.device) if STREAM_SAFE else None

        def ahead():
            # On the second stream these tensors are consumed on the launch stream by the next iteration.  No
            # Tensor.record_stream (one event per tensor and free: ~40 barrier packets a step, measured +6 % step time):
            # the buffers of iteration i + 1 are only written by work the second stream enqueues after waiting for the
            # launch stream, and the launch stream ends each iteration with a wait on the second — i.e. after every
            # consumer of their previous contents (_ForkJoin asserts that structure; FNR_STREAM_SAFE=1 registers the
            # tensors with the allocator as well).  With step arenas they go to the OTHER arena, from its start: its
            # previous contents belong to iteration i - 1, which the launch stream has joined.
            if other is not None:
                other.reset()
                K.use_arena(other)
            try:
                self._next = (step + 1,) + self._draw(step)
            finally:
                if other is not None:
                    K.use_arena(arena)
            self._next_version = self._outside_version()
            if other is not None:
                # (arena version, parity, where the next step's own buffers start, nothing overflowed into the allocator)
                self._next_layout = (self._arena_version, 1 - (step & 1), other.offset, not other.overflow_bytes)
            crosses_to(launch_stream, self._next, getattr(self.batcher, "last_draw", None),
                       getattr(self.batcher, "last_presample", None))

        prog = log = None
        fresh_ws = K.fresh_workspaces()
        if record_key is not None:
            prog = _StepProgram()
            log = L.begin_call_log(keep=prog.keep)
            rc = L.load().fnr_program_begin(prog.handle)
            if rc != 0:
                L.end_call_log(log)
                L.check(rc, "program_begin")
        prev = K.use_arena(arena)
        try:
            out = fused_train_iteration(self.model, self.optimizer, rb, batch, step, world_size=self.world_size,
                                        want_metrics=want_metrics, camera=self.camera,
                                        ahead=ahead if (SAMPLE_AHEAD and self.model.training) else None)
        except BaseException:
            if prog is not None:
                L.end_call_log(log)
                L.load().fnr_program_abort(prog.handle)
            raise
        finally:
            K.use_arena(prev)
        if prog is not None:
            self._finish_recording(prog, log, record_key, rb, batch, arena, other, fresh_ws)
        self.stats["interpreted"] += 1
        self.step_idx += 1
        return out

    def _finish_recording(self, prog, log, key, rb, batch, arena, other, fresh_ws_before) -> None:
        names = L.end_call_log(log)
        lib = L.load()
        rc = lib.fnr_program_end(prog.handle)
        reason = None
        if rc != 0:
            reason = "unrecordable entry point: " + L.last_error()
        else:
            asked = [n for n in names if n not in _PROGRAM_QUERIES]
            got = [lib.fnr_program_op_name(prog.handle, i).decode() for i in range(lib.fnr_program_size(prog.handle))]
            if asked != got:
                reason = f"python asked for {len(asked)} calls, the library recorded {len(got)}: " + \
                         repr([(a, g) for a, g in zip(asked, got) if a != g][:3])
        transient = arena.overflow_bytes or other.overflow_bytes or K.fresh_workspaces() != fresh_ws_before \
            or self._next is None or self._next_layout is None or not self._next_layout[3]
        if reason is not None:
            self._unrecordable[key] = reason
            self.stats["record_failed"] += 1
            lib.fnr_program_abort(prog.handle)
            return
        if transient:               # grown arenas / a first-use workspace: the same shape records on a later occurrence
            lib.fnr_program_abort(prog.handle)
            return
        prog.n_ops = int(lib.fnr_program_size(prog.handle))
        prog.next = (self._next[1], self._next[2])
        prog.next_offset = self._next_layout[2]
        prog.last_draw = getattr(self.batcher, "last_draw", None)
        prog.last_presample = getattr(self.batcher, "last_presample", None)
        prog.keep += [rb, batch, self._arenas]
        self._programs[key] = prog
        self.stats["recorded"] += 1

    def _replay(self, prog, step: int, want_metrics: bool, updated: bool):
        """The host side of one replayed step: exactly the bookkeeping the interpreted step does on Python objects (schedule
        counters, optimiser step counts, the batcher's random-number counter, the look-ahead's hand-over) around ONE call."""
        model, opt, sampler, batcher = self.model, self.optimizer, self.model.proposal_sampler, self.batcher
        sc = self._scalars
        sampler._anneal = model.anneal_at(step)                          # set_anneal(step)
        # optimiser scalars of THIS update (FusedAdam.table_adam_args / weight_adam_args: lr at the current scheduler step,
        # step count + 1), then what optimizer.step(skip, done) does to the counters once everything is fused
        groups, count = opt.groups, opt.step_count
        for name, slot in (("fields", 0), ("proposal_networks", 1)):
            g = groups[name]
            sc.adam[slot].lr = (exponential_decay_lr(count, g["lr"], g["lr_final"], g["max_steps"])
                                if g.get("lr_final") is not None else g["lr"])
            sc.adam[slot].step = opt.group_steps[name] + 1
        skip = () if (updated or not opt.skip_groups_without_grad) else ("proposal_networks",)
        opt.step_count = count + 1
        for name in opt.group_steps:
            if name not in skip:
                opt.group_steps[name] += 1
        if self.camera is not None:
            sc.adam[2].lr, sc.adam[2].step = self.camera[1].advance()
        # _render's bookkeeping
        model.__dict__["_ahead_used"] = model.__dict__.get("_ahead_used", 0) + 1
        if updated:
            sampler._steps_since_update = 0
        model._last_render_updated = bool(updated)
        # the look-ahead of iteration step + 1 (TrainingSteps._draw / FruitModel.sample_ahead)
        batcher._offset = getattr(batcher, "_offset", 0) + 1
        sc.prologue_offset = batcher._offset & 0xFFFFFFFFFFFFFFFF
        anneal_next = model.anneal_at(step + 1)
        sc.anneal = anneal_next
        losses = torch.empty(5, device=model.device)
        sc.losses = losses.data_ptr()
        rc = L._lib.fnr_program_replay(prog.handle, sc)
        if rc != 0:
            K._forget_scatter_workspaces(model.device)
            self.drop_programs()
            L.check(rc, "program_replay")
        rb, batch = prog.next
        ahead = rb.presampled["ahead"]
        ahead["anneal"] = anneal_next
        ahead["param_version"] = model.lookahead_version()
        batcher.last_draw, batcher.last_presample = prog.last_draw, prog.last_presample
        self._next = (step + 1, rb, batch)
        self._next_version = self._outside_version()
        self._next_layout = (self._arena_version, 1 - (step & 1), prog.next_offset, True)
        sampler.step_cb(step)                                              # AFTER_TRAIN_ITERATION callback
        self.stats["replayed"] += 1
        self.step_idx = step + 1
        l0, l1, l2, l3, l4 = losses.unbind(0)
        return ({"rgb_loss": l0, "semantics_loss": l1, "interlevel_loss": l3},
                {"psnr": l2, "distortion": l4} if want_metrics else {})


# entry points a recording does not contain: pure host queries, and the program API itself
_PROGRAM_QUERIES = frozenset(["fnr_field_mlp_fwd_workspace_bytes", "fnr_field_h_dim", "fnr_field_mlp_bwd_workspace_bytes",
                              "fnr_hash_scatter_workspace_bytes", "fnr_prop_density_bwd_workspace_bytes", "fnr_last_error",
                              "fnr_program_begin", "fnr_program_end", "fnr_program_abort", "fnr_program_size",
                              "fnr_program_op_name", "fnr_program_create", "fnr_program_destroy", "fnr_event_create",
                              "fnr_event_destroy", "fnr_abi_version"])
