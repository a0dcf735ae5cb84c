"""FruitModel — MI355X-native mirror of /root/reference/fruit_nerf/fruit_nerf.py:50-458.

Same config fields, constructor kwargs, method names, output keys and loss names as the reference
Nerfstudio model.  The hot path (collider -> proposal sampling -> proposal nets -> field -> weights ->
renderers -> losses, and the export field queries) runs in libfruitnerf_hip.so; this file only sequences
kernel launches on PyTorch's current HIP stream.  There is no PyTorch fallback for any of it.
"""
from __future__ import annotations

from collections import defaultdict
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple, Type, Union

import numpy as np
import torch
from torch import Tensor, nn

from . import _kernels as K
from . import _lib as L
from .components.ray_samplers import UniformSamplerWithNoise
from .fruit_field import FieldHeadNames, FruitField, SceneContraction
from .params import HashEncoding, MLP, ParamArena
from .rays import RayBundle


@dataclass
class FruitNerfModelConfig:
    """FruitNerfModelConfig(NerfactoModelConfig) — fruit_nerf.py:50-59 plus the inherited Nerfacto 0.3.2
    defaults the hot path reads."""
    _target: Type = field(default_factory=lambda: FruitModel)
    near_plane: float = 0.05
    far_plane: float = 1000.0
    background_color: str = "last_sample"
    hidden_dim: int = 64            # ignored by FruitField construction (fruit_nerf.py:88-103)
    hidden_dim_color: int = 64      # ignored
    appearance_embed_dim: int = 32  # ignored
    num_levels: int = 16
    base_res: int = 16              # ignored
    max_res: int = 2048
    log2_hashmap_size: int = 19
    features_per_level: int = 2     # ignored
    num_proposal_samples_per_ray: Tuple[int, ...] = (256, 96)
    num_nerf_samples_per_ray: int = 48
    proposal_update_every: int = 5
    proposal_warmup: int = 5000
    num_proposal_iterations: int = 2
    use_same_proposal_network: bool = False
    proposal_net_args_list: List[Dict] = field(default_factory=lambda: [
        {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 128, "use_linear": False},
        {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 256, "use_linear": False},
    ])
    proposal_initial_sampler: str = "piecewise"
    interlevel_loss_mult: float = 1.0
    distortion_loss_mult: float = 0.002
    use_proposal_weight_anneal: bool = True
    use_average_appearance_embedding: bool = True
    proposal_weights_anneal_slope: float = 10.0
    proposal_weights_anneal_max_num_iters: int = 1000
    use_single_jitter: bool = True
    disable_scene_contraction: bool = False
    use_gradient_scaling: bool = False
    eval_num_rays_per_chunk: int = 1 << 15
    mlp_precision: Optional[str] = None   # "auto" | "fp32" | "bf16x3" | "bf16" (FruitField.mlp_precision; None: env / auto)
    eval_outputs_on_cpu: bool = False   # True: full-image eval returns CPU tensors like the reference (fruit_nerf.py:245)
    # FruitNerfModelConfig proper
    semantic_loss_weight: float = 1.0
    pass_semantic_gradients: bool = False
    num_layers_semantic: int = 2
    hidden_dim_semantics: int = 64
    geo_feat_dim: int = 15

    def setup(self, **kwargs):
        return self._target(self, **kwargs)


class HashMLPDensityField(nn.Module):
    """Parameter holder for nerfstudio HashMLPDensityField (fruit_nerf.py:104-129); evaluated by
    fnr_prop_density_fwd."""

    def __init__(self, aabb: Tensor, num_layers: int = 2, hidden_dim: int = 64, spatial_distortion=None,
                 use_linear: bool = False, num_levels: int = 8, max_res: int = 1024, base_res: int = 16,
                 log2_hashmap_size: int = 18, features_per_level: int = 2, implementation: str = "hip"):
        super().__init__()
        if use_linear or num_layers != 2:
            raise NotImplementedError("proposal nets are built as hash -> Linear -> ReLU -> Linear (use_linear=False)")
        self.register_buffer("aabb", aabb.clone().float())
        self.spatial_distortion = spatial_distortion
        self.hidden_dim = hidden_dim
        self.encoding = HashEncoding(num_levels=num_levels, min_res=base_res, max_res=max_res,
                                     log2_hashmap_size=log2_hashmap_size, features_per_level=features_per_level)
        network = MLP(in_dim=self.encoding.get_out_dim(), num_layers=num_layers, layer_width=hidden_dim, out_dim=1)
        self.mlp_base = nn.Sequential(self.encoding, network)

    def prop_struct(self, grads: bool = False) -> L.fnr_prop_net:
        def P(p):
            return (p.grad if grads else p.data).data_ptr()

        e = self.encoding
        key = (grads, P(e.hash_table), P(self.mlp_base[1].layers[0].weight))
        cache = self.__dict__.setdefault("_struct_cache", {})
        hit = cache.get(grads)
        if hit is not None and hit[0] == key:
            return hit[1]
        net = L.fnr_prop_net()
        cache[grads] = (key, net)
        net.grid = K.make_grid(e.hash_table.grad if grads else e.hash_table.data, e.num_levels, e.log2_hashmap_size,
                               e.scalings)
        net.hidden_dim = self.hidden_dim
        lyr = self.mlp_base[1].layers
        net.w0, net.b0, net.w1, net.b1 = P(lyr[0].weight), P(lyr[0].bias), P(lyr[1].weight), P(lyr[1].bias)
        return net

    def warp_struct(self) -> L.fnr_warp:
        return K.make_warp(0 if self.spatial_distortion is not None else 1, self.aabb)

    @torch.no_grad()
    def density_fn(self, positions: Tensor) -> Tensor:
        """Field.density_fn (positions [...,3] -> density [...,1]); zero-length frustums."""
        shape = positions.shape[:-1]
        pos = positions.reshape(-1, 3).float()
        rays = K.RaysArg(pos, torch.ones_like(pos), None, None)
        euclid = torch.zeros(rays.n, 2, device=pos.device)
        density, _ = K.prop_density_fwd(self.prop_struct(), self.warp_struct(), rays, euclid, 1)
        return density.view(*shape, 1)


class ProposalNetworkSampler(nn.Module):
    """State of nerfstudio's ProposalNetworkSampler (fruit_nerf.py:151-158): anneal, update schedule."""

    def __init__(self, num_proposal_samples_per_ray=(64,), num_nerf_samples_per_ray=32,
                 num_proposal_network_iterations=2, single_jitter=False, update_sched: Callable = lambda x: 1,
                 initial_sampler=None):
        super().__init__()
        if initial_sampler is not None:
            raise NotImplementedError("only the default piecewise initial sampler is built (fruit_nerf.py:140,157)")
        if not single_jitter:
            raise NotImplementedError("use_single_jitter=False is not built (Nerfacto default is True)")
        self.num_proposal_samples_per_ray = tuple(num_proposal_samples_per_ray)
        self.num_nerf_samples_per_ray = num_nerf_samples_per_ray
        self.num_proposal_network_iterations = num_proposal_network_iterations
        self.update_sched = update_sched
        self._anneal = 1.0
        self._steps_since_update = 0
        self._step = 0

    def set_anneal(self, anneal: float) -> None:
        self._anneal = anneal

    def step_cb(self, step):
        self._step = step
        self._steps_since_update += 1

    def updated_now(self) -> bool:
        return bool(self._steps_since_update > self.update_sched(self._step) or self._step < 10)

    def updated_after(self, step: int) -> bool:
        """What updated_now() will answer once step_cb(step) has run (the next iteration's forward pass)."""
        return bool(self._steps_since_update + 1 > self.update_sched(step) or step < 10)


@dataclass
class RenderContext:
    """Everything one forward pass produced (kept for the backward pass in training)."""
    rays: K.RaysArg
    levels: List[dict]          # per sampling level: S, spacing, euclid, density, weights, feats
    updated: bool
    training: bool
    field_feats: Optional[Tensor] = None
    field_selector: Optional[Tensor] = None
    field_h: Optional[Tensor] = None
    labels: Optional[Tensor] = None
    sample_rgb: Optional[Tensor] = None
    sample_logit: Optional[Tensor] = None
    sample_density: Optional[Tensor] = None
    weights: Optional[Tensor] = None
    field_jacobian: Optional[Tensor] = None  # d feats / d(unit-cube position) [L,3,N,2] when the rays want gradients
    ray_bundle: Optional[object] = None   # the caller's RayBundle (its origins / directions may carry autograd history)
    composite_grads: Optional[tuple] = None   # (d_density, d_rgb, d_logit) when the compositing launch ran its own backward


class FruitModel(nn.Module):
    config: FruitNerfModelConfig

    def __init__(self, config: FruitNerfModelConfig, metadata: Dict, scene_box=None,
                 num_train_data: int = 1, device: Union[str, torch.device] = "cuda", grad_scaler=None,
                 test_mode: Optional[str] = None, render_rgb_inference: bool = True, **kwargs) -> None:
        """fruit_nerf.py:71-76 + nerfstudio Model.__init__; called by FruitPipeline with exactly
        (config, scene_box=, num_train_data=, metadata=, device=, grad_scaler=, test_mode=, render_rgb_inference=)
        (fruit_pipeline.py:104-112).  `metadata["semantics"]` is required, as in the reference (a nerfstudio
        `Semantics`; only `.colors` is read, so the check is duck-typed — a host passes nerfstudio's class)."""
        super().__init__()
        assert "semantics" in metadata.keys() and hasattr(metadata["semantics"], "colors"), \
            'FruitModel needs metadata["semantics"] (nerfstudio Semantics: filenames, classes, colors, mask_classes)'
        self.semantics = metadata["semantics"]
        # the Trainer's GradScaler (fruit_pipeline.py:109; enabled under mixed_precision=True, fruit_nerf_config.py:33).
        # The kernels compute in fp32 and the fused loop never scales its loss; a host that does scale
        # (`grad_scaler.scale(loss).backward()`) gets correctly scaled gradients from the autograd path and steps
        # FusedAdam through training.scaler_step(optimizer, grad_scaler) (unscale inside the Adam kernel, skip on inf).
        self.grad_scaler = grad_scaler
        self.test_mode = test_mode
        self.config = config
        self.num_train_data = num_train_data
        self.kwargs = kwargs
        if scene_box is None:
            aabb = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])  # fruitnerf_dataparser.py:218-223
        else:
            aabb = scene_box.aabb if hasattr(scene_box, "aabb") else torch.as_tensor(scene_box)
        self.scene_aabb = aabb.float()
        self._device = torch.device(device)
        self.colormap = torch.as_tensor(self.semantics.colors).clone().detach()   # fruit_nerf.py:76
        self._arena: Optional[ParamArena] = None
        self.populate_modules()
        # nerfstudio Model.__init__ registers this zero-length parameter after populate_modules() (base_model.py; every
        # checkpoint a Nerfstudio Trainer wrote carries the key `_model.device_indicator_param`, and
        # FruitPipeline.load_pipeline loads with strict=True, fruit_pipeline.py:239-240).  It belongs to no parameter group
        # (get_param_groups) and stays outside the arena.
        self.device_indicator_param = nn.Parameter(torch.empty(0))
        if self._device.type == "cuda":
            self.to(self._device)

    # ---- construction (fruit_nerf.py:78-177) -------------------------------------------------------------
    def populate_modules(self):
        cfg = self.config
        scene_contraction = None if cfg.disable_scene_contraction else SceneContraction(order=float("inf"))
        # fruit_nerf.py:88-103 — only these config fields reach the field
        self.field = FruitField(self.scene_aabb, num_levels=cfg.num_levels, max_res=cfg.max_res,
                                num_layers_semantic=cfg.num_layers_semantic,
                                hidden_dim_semantics=cfg.hidden_dim_semantics,
                                log2_hashmap_size=cfg.log2_hashmap_size, spatial_distortion=scene_contraction,
                                num_images=self.num_train_data, geo_feat_dim=cfg.geo_feat_dim,
                                use_average_appearance_embedding=cfg.use_average_appearance_embedding,
                                use_semantics=True, test_mode=self.test_mode, num_semantic_classes=1,
                                pass_semantic_gradients=cfg.pass_semantic_gradients,
                                mlp_precision=cfg.mlp_precision)
        self.density_fns = []
        num_prop_nets = cfg.num_proposal_iterations
        self.proposal_networks = nn.ModuleList()
        if cfg.use_same_proposal_network:
            assert len(cfg.proposal_net_args_list) == 1, "Only one proposal network is allowed."
            network = HashMLPDensityField(self.scene_aabb, spatial_distortion=scene_contraction,
                                          **cfg.proposal_net_args_list[0])
            self.proposal_networks.append(network)
            self.density_fns.extend([network.density_fn for _ in range(num_prop_nets)])
        else:
            for i in range(num_prop_nets):
                args = cfg.proposal_net_args_list[min(i, len(cfg.proposal_net_args_list) - 1)]
                self.proposal_networks.append(
                    HashMLPDensityField(self.scene_aabb, spatial_distortion=scene_contraction, **args))
            self.density_fns.extend([network.density_fn for network in self.proposal_networks])

        import functools

        @functools.lru_cache(maxsize=8192)     # (a pure function of the step, asked several times per iteration)
        def update_schedule(step):  # fruit_nerf.py:131-136
            return np.clip(np.interp(step, [0, cfg.proposal_warmup], [0, cfg.proposal_update_every]),
                           1, cfg.proposal_update_every)

        if cfg.proposal_initial_sampler == "uniform":
            # the reference leaves self.proposal_sampler undefined on this branch (fruit_nerf.py:145-149)
            raise NotImplementedError('proposal_initial_sampler="uniform" is broken in the reference; use "piecewise"')
        self.proposal_sampler = ProposalNetworkSampler(
            num_nerf_samples_per_ray=cfg.num_nerf_samples_per_ray,
            num_proposal_samples_per_ray=cfg.num_proposal_samples_per_ray,
            num_proposal_network_iterations=cfg.num_proposal_iterations,
            single_jitter=cfg.use_single_jitter, update_sched=update_schedule, initial_sampler=None)
        if cfg.background_color != "last_sample":
            raise NotImplementedError("only background_color='last_sample' (Nerfacto default) is built")
        if cfg.use_gradient_scaling:
            raise NotImplementedError("use_gradient_scaling=True is not built (Nerfacto default False)")

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._arena = None
        return out

    @property
    def device(self):
        return self.field.aabb.device

    def arena(self) -> ParamArena:
        """One flat fp32 arena for proposal networks + field (stable pointers, single all-reduce)."""
        if self._arena is None:
            dev = self.field.aabb.device
            if dev.type != "cuda":
                raise RuntimeError("FruitModel is on %s; move it to a HIP device (no CPU path)" % dev)
            self._arena = ParamArena([("proposal_networks", list(self.proposal_networks.parameters())),
                                      ("fields", list(self.field.parameters()))], dev)
            self.field.adopt_arena(self._arena)
        return self._arena

    # ---- reference API -------------------------------------------------------------------------------------
    def setup_inference(self, render_rgb, num_inference_samples, deterministic: bool = False):
        """fruit_nerf.py:179-183.  The default is the reference AS IT RUNS: the exporter calls eval_setup() first
        (scripts/exporter.py:86-94) and only then setup_inference(), which builds a fresh UniformSamplerWithNoise — a new
        nn.Module is in training mode, so the export jitters every bin edge with torch.rand (ray_samplers.py:79-87) and
        the exported point sets depend on the RNG stream (reproduced exactly given the same numbers: `jitter_fn`,
        tests/test_gpu_reference_pins.py::test_hip_export_matches_the_reference_export_as_run).
        deterministic=True: the sampler is put in eval mode (bin centres) — the lattice for which "identical exported
        point counts" is a meaningful bar, and the one the fused lattice kernels (fnr_hash_encode_lattice) implement."""
        self.render_rgb = render_rgb
        self.num_inference_samples = num_inference_samples
        self.proposal_sampler = UniformSamplerWithNoise(num_samples=self.num_inference_samples, single_jitter=False)
        if deterministic:
            self.proposal_sampler.eval()
        self.field.spatial_distortion = None

    def update_to_step(self, step: int) -> None:
        """nerfstudio Model.update_to_step: called by FruitPipeline.load_pipeline right before load_state_dict
        (fruit_pipeline.py:239).  Nerfacto-family models keep no step-dependent module state (the anneal and the proposal
        update schedule are driven by the training callbacks), so, like the base class, this changes no parameter; it only
        invalidates whatever was sampled ahead with the weights that are about to be replaced (lookahead_version)."""
        self.__dict__["_lookahead_epoch"] = self.__dict__.get("_lookahead_epoch", 0) + 1

    def get_param_groups(self) -> Dict[str, List[nn.Parameter]]:  # fruit_nerf.py:185-189
        return {"proposal_networks": list(self.proposal_networks.parameters()),
                "fields": list(self.field.parameters())}

    def anneal_at(self, step: int) -> float:  # fruit_nerf.py:199-207
        N = self.config.proposal_weights_anneal_max_num_iters
        train_frac = np.clip(step / N, 0, 1)

        def bias(x, b):
            return b * x / ((b - 1) * x + 1)

        return bias(train_frac, self.config.proposal_weights_anneal_slope)

    def set_anneal(self, step: int) -> None:  # the BEFORE_TRAIN_ITERATION callback, fruit_nerf.py:199-207
        self.proposal_sampler.set_anneal(self.anneal_at(step))

    def get_training_callbacks(self, training_callback_attributes=None) -> List["TrainingCallback"]:
        """fruit_nerf.py:191-223: the anneal callback before and the sampler's step callback after every training
        iteration, as TrainingCallback objects (the host's class inside Nerfstudio, engine/callbacks.py otherwise) — a
        Trainer drives them through `run_callback_at_location(step, location)`."""
        from .engine.callbacks import TrainingCallback, TrainingCallbackLocation
        callbacks = []
        if self.config.use_proposal_weight_anneal:
            callbacks.append(TrainingCallback(where_to_run=[TrainingCallbackLocation.BEFORE_TRAIN_ITERATION],
                                              update_every_num_iters=1, func=self.set_anneal))
            callbacks.append(TrainingCallback(where_to_run=[TrainingCallbackLocation.AFTER_TRAIN_ITERATION],
                                              update_every_num_iters=1, func=self.proposal_sampler.step_cb))
        return callbacks

    def level0_spec(self) -> Optional[dict]:
        """What a datamanager needs to pre-sample the proposal sampler's level 0 in the launch that draws the rays
        (PixelBatcher.sample(level0=...) -> RayBundle.presampled): training mode, piecewise spacing, single jitter."""
        sampler = self.proposal_sampler
        if not self.training or not isinstance(sampler, ProposalNetworkSampler):
            return None
        if sampler.num_proposal_network_iterations + 1 > L.FNR_TRAIN_PROLOGUE_MAX_JITTER:
            return None     # more jitters than fnr_train_prologue draws: the separate launches (torch.rand + sample_spaced)
        return {"S": sampler.num_proposal_samples_per_ray[0], "near": float(self.config.near_plane),
                "far": float(self.config.far_plane), "n_jitter": sampler.num_proposal_network_iterations + 1}

    def _collide(self, ray_bundle: RayBundle) -> RayBundle:  # NearFarCollider, fruit_nerf.py:161,382-383
        if ray_bundle.nears is not None and ray_bundle.fars is not None:
            if ray_bundle.__dict__.get("_collided_by") is self:
                return ray_bundle               # this collider's planes already (sample_ahead)
            if getattr(ray_bundle, "presampled", None) is not None:
                ray_bundle.presampled = None    # pre-sampled for the collider's planes, not for the caller's own
            return ray_bundle
        near_plane = self.config.near_plane if self.training else 0
        # constant per (shape, mode): built once instead of 4 launches per call; the hot path only reads them
        key = (tuple(ray_bundle.origins.shape[:-1]), float(near_plane), str(ray_bundle.origins.device))
        cache = self.__dict__.setdefault("_collider_cache", {})
        if key not in cache:
            if len(cache) > 8:
                cache.clear()
            ones = torch.ones_like(ray_bundle.origins[..., 0:1])
            cache[key] = (ones * near_plane, ones * self.config.far_plane)
        ray_bundle.nears, ray_bundle.fars = cache[key]
        ray_bundle.__dict__["_collided_by"] = self
        return ray_bundle

    # ---- the hot path ----------------------------------------------------------------------------------------
    def _render(self, ray_bundle: RayBundle, jitter: Optional[List[Tensor]] = None,
                save_input_jacobian: bool = False, loss_targets=None) -> Tuple[Dict, RenderContext]:
        """ProposalNetworkSampler.generate_ray_samples + field + weights + renderers (fruit_nerf.py:316-357).
        loss_targets = (image [R,3], fruit_mask [R,1], semantic loss weight), training steps that do not train the proposal networks:
        the compositing launch also runs its own backward under the rgb / semantic losses of get_loss_dict (fnr_composite_fwd_bwd_targets) and the context carries
        `composite_grads` = (d_density, d_rgb, d_logit) per sample."""
        self.arena()
        cfg = self.config
        sampler = self.proposal_sampler
        training = self.training
        if ray_bundle.origins.shape[0] == 0:  # empty batch: nothing to launch (zero-size tensors have no storage)
            return self._empty_render(ray_bundle)
        rays = K.RaysArg(ray_bundle.origins, ray_bundle.directions, ray_bundle.nears, ray_bundle.fars,
                         ray_bundle.camera_indices)
        dev = rays.device
        n_prop = sampler.num_proposal_network_iterations
        updated = sampler.updated_now()
        jit = list(jitter) if jitter is not None else [None] * (n_prop + 1)
        S0 = sampler.num_proposal_samples_per_ray[0]
        # level 0 + the jitters may come with the rays (fnr_train_prologue drew them in the launch that sampled the
        # pixels): used when they match this sampler, the collider's planes and no explicit jitter was passed
        pre = getattr(ray_bundle, "presampled", None) if (training and jitter is None) else None
        if pre is not None and not (pre.get("S0") == S0 and len(pre.get("jitter", ())) >= n_prop + 1
                                    and pre["spacing"].shape[0] == rays.n and pre.get("near") == float(cfg.near_plane)
                                    and pre.get("far") == float(cfg.far_plane)):
            pre = None
        if pre is not None:
            jit = list(pre["jitter"][:n_prop + 1])
        elif training:
            if any(j is None for j in jit):
                fresh = torch.rand(n_prop + 1, rays.n, device=dev)  # one launch for all sampling levels
                jit = [j if j is not None else fresh[i] for i, j in enumerate(jit)]
        else:
            jit = [None] * (n_prop + 1)
        ahead = pre.get("ahead") if pre is not None else None
        if ahead is not None and ahead["updated"] == updated and ahead["anneal"] == sampler._anneal \
                and ahead.get("param_version", None) in (None, self.lookahead_version()):
            # the proposal levels of these rays were sampled ahead (sample_ahead: enqueued at the end of the previous
            # iteration) under the schedule this pass sees
            levels, spacing, euclid, S = list(ahead["levels"]), ahead["spacing"], ahead["euclid"], ahead["S"]
            self.__dict__["_ahead_used"] = self.__dict__.get("_ahead_used", 0) + 1   # (tests: the look-ahead is what ran)
        else:
            levels, spacing, euclid, S = self._proposal_levels(rays, pre, jit, training and updated, sampler._anneal)
        if updated:
            sampler._steps_since_update = 0
        self._last_render_updated = bool(training and updated)   # did this pass keep the proposal nets' graph?

        fld = self.field
        net = fld.net_struct()
        # rays with autograd history (camera-pose optimisation) need the encode's input Jacobian in the backward pass
        want_jac = training and (save_input_jacobian or ray_bundle.origins.requires_grad
                                 or ray_bundle.directions.requires_grad)
        enc = K.hash_encode_fwd(net.grid, fld.warp_struct(), rays, euclid, S, want_jacobian=want_jac)
        feats, selector, jac = enc if want_jac else (enc[0], enc[1], None)
        mean_emb = fld._mean_embedding() if fld._uses_mean_embedding() else None
        if mean_emb is None and rays.cam is None:
            raise AttributeError("Camera indices are not provided.")
        density, rgb, logit, _, h_saved = K.field_mlp_fwd(net, rays, S, feats, selector, mean_emb, want_h=True)
        composite_grads = None
        # (not on the steps that train the proposal networks: there the second stream's chain — losses launch, proposal
        #  backward, their scatter — is the longer one, and it starts behind this launch)
        if loss_targets is not None and training and not updated:
            (weights, out_rgb, acc, depth, sem, label), composite_grads = K.composite_fwd_bwd_targets(
                rays, S, euclid, density, rgb, logit, loss_targets[0], loss_targets[1], loss_targets[2])
        else:
            weights, out_rgb, acc, depth, sem, label = K.composite_fwd(rays, S, euclid, density, rgb, logit, training)
        levels.append(dict(S=S, spacing=spacing, euclid=euclid, density=density.view(rays.n, S), weights=weights,
                           depth=depth, feats=None))
        ctx = RenderContext(rays=rays, levels=levels, updated=updated, training=training, field_feats=feats,
                            field_selector=selector, field_h=h_saved, sample_rgb=rgb, sample_logit=logit, sample_density=density,
                            weights=weights)
        ctx.labels = label[:, None]
        ctx.ray_bundle = ray_bundle
        ctx.field_jacobian = jac
        ctx.composite_grads = composite_grads
        outputs = {"rgb": out_rgb, "accumulation": acc[:, None], "depth": depth[:, None],
                   "semantics": sem[:, None]}
        for i in range(n_prop):
            outputs[f"prop_depth_{i}"] = levels[i]["depth"][:, None]
        return outputs, ctx

    def _proposal_levels(self, rays, pre, jit, save_feats: bool, anneal: float):
        """ProposalNetworkSampler.generate_ray_samples: level-0 bins (taken from the prologue's `pre` when given), then
        per proposal network density -> weights -> inverse-CDF bins of the next level -> (levels, spacing, euclid, S)."""
        cfg, sampler = self.config, self.proposal_sampler
        n_prop = sampler.num_proposal_network_iterations
        S = sampler.num_proposal_samples_per_ray[0]
        levels: List[dict] = []
        spacing, euclid = (pre["spacing"], pre["euclid"]) if pre is not None else K.sample_spaced(rays, 1, S, jit[0])
        for i in range(n_prop):
            net = self.proposal_networks[0 if cfg.use_same_proposal_network else i]
            density, feats = K.prop_density_fwd(net.prop_struct(), net.warp_struct(), rays, euclid, S,
                                                save_feats=save_feats)
            S_next = sampler.num_proposal_samples_per_ray[i + 1] if i + 1 < n_prop else sampler.num_nerf_samples_per_ray
            weights, depth, spacing_n, euclid_n = K.weights_pdf(rays, 1, S, S_next, density, spacing, euclid,
                                                                anneal, jit[i + 1])
            levels.append(dict(S=S, spacing=spacing, euclid=euclid, density=density, weights=weights, depth=depth,
                               feats=feats))
            spacing, euclid, S = spacing_n, euclid_n, S_next
        return levels, spacing, euclid, S

    @torch.no_grad()
    def sample_ahead(self, ray_bundle: RayBundle, step: int) -> bool:
        """The proposal sampling of the NEXT training iteration (`step` = the iteration that is finishing), enqueued
        ahead of it: neither the rays (drawn by fnr_train_prologue) nor the proposal networks depend on what the
        current iteration still has in flight once the proposal networks' and the cameras' optimiser steps are enqueued
        — the main table's scatter + optimiser step, the longest launch pair of the step (training.TrainingSteps).
        The result rides in ray_bundle.presampled["ahead"] together with the schedule it assumed (proposal update flag
        after step_cb(step), anneal of step + 1); _render recomputes when that is not the schedule it finds.
        Valid only while the proposal networks' parameters stay what they were when this ran."""
        sampler = self.proposal_sampler
        pre = getattr(ray_bundle, "presampled", None)
        if not self.training or pre is None or not isinstance(sampler, ProposalNetworkSampler) \
                or ray_bundle.origins.shape[0] == 0:
            return False
        self.arena()
        ray_bundle = self._collide(ray_bundle)
        pre = ray_bundle.presampled
        cfg = self.config
        n_prop = sampler.num_proposal_network_iterations
        if pre is None or not (pre.get("S0") == sampler.num_proposal_samples_per_ray[0]
                               and len(pre.get("jitter", ())) >= n_prop + 1
                               and pre["spacing"].shape[0] == ray_bundle.origins.shape[0]
                               and pre.get("near") == float(cfg.near_plane) and pre.get("far") == float(cfg.far_plane)):
            return False
        rays = K.RaysArg(ray_bundle.origins, ray_bundle.directions, ray_bundle.nears, ray_bundle.fars,
                         ray_bundle.camera_indices)
        updated, anneal = sampler.updated_after(step), self.anneal_at(step + 1)
        levels, spacing, euclid, S = self._proposal_levels(rays, pre, list(pre["jitter"][:n_prop + 1]), updated, anneal)
        pre["ahead"] = dict(levels=levels, spacing=spacing, euclid=euclid, S=S, updated=updated, anneal=anneal,
                            param_version=self.lookahead_version())
        return True

    def lookahead_version(self) -> tuple:
        """What a cached look-ahead is tied to besides the schedule: the identity of the parameter arena and torch's
        version counters of the proposal networks' parameters.  The library's own optimiser steps write through raw
        pointers and leave the counters alone (the look-ahead is enqueued behind them on purpose); anything torch does to
        the parameters in between — load_state_dict, an external optimiser's step, `with torch.no_grad(): p.copy_()` —
        bumps them, and _render / TrainingSteps then sample again instead of training on samples and saved features of
        the old weights.  (Writes through `p.data` bypass the counters: call TrainingSteps.drop_lookahead().)"""
        arena = self._arena
        cached = self.__dict__.get("_lookahead_params")
        if cached is None or cached[0] is not arena:     # (module traversal per call cost ~45 us of host time per step)
            cached = self.__dict__["_lookahead_params"] = (arena, list(self.proposal_networks.parameters()))
        return (id(arena), None if arena is None else arena.params.data_ptr(), self.__dict__.get("_lookahead_epoch", 0)) \
            + tuple(p._version for p in cached[1])

    def _empty_render(self, ray_bundle: RayBundle) -> Tuple[Dict, RenderContext]:
        dev = ray_bundle.origins.device
        sampler = self.proposal_sampler
        n_prop = sampler.num_proposal_network_iterations
        counts = list(sampler.num_proposal_samples_per_ray[:n_prop]) + [sampler.num_nerf_samples_per_ray]
        z = lambda *shape, dtype=torch.float32: torch.zeros(*shape, dtype=dtype, device=dev)  # noqa: E731
        levels = [dict(S=S, spacing=z(0, S + 1), euclid=z(0, S + 1), density=z(0, S), weights=z(0, S), depth=z(0),
                       feats=None) for S in counts]
        ctx = RenderContext(rays=None, levels=levels, updated=False, training=self.training)
        ctx.labels = z(0, 1, dtype=torch.long)
        ctx.ray_bundle = ray_bundle
        outputs = {"rgb": z(0, 3), "accumulation": z(0, 1), "depth": z(0, 1), "semantics": z(0, 1)}
        for i in range(n_prop):
            outputs[f"prop_depth_{i}"] = z(0, 1)
        return outputs, ctx

    def _samples_lists(self, ray_bundle: RayBundle, ctx: RenderContext):
        weights_list, ray_samples_list = [], []
        for lv in ctx.levels:
            weights_list.append(lv["weights"][..., None])
            e, s = lv["euclid"], lv["spacing"]
            rs = ray_bundle.get_ray_samples(bin_starts=e[..., :-1, None], bin_ends=e[..., 1:, None],
                                            spacing_starts=s[..., :-1, None], spacing_ends=s[..., 1:, None])
            rs._structured = (ctx.rays, e, lv["S"])
            ray_samples_list.append(rs)
        return weights_list, ray_samples_list

    def _finish_outputs(self, ray_bundle, outputs, ctx, repeat_colormap: bool):
        weights_list, ray_samples_list = self._samples_lists(ray_bundle, ctx)
        outputs["weights_list"] = weights_list
        outputs["ray_samples_list"] = ray_samples_list
        # semantics colormap (fruit_nerf.py:309-312, 351-355): heaviside(sigmoid(sem) - 0.9, 0) -> colormap lookup
        semantic_labels = ctx.labels  # heaviside(sigmoid(sem) - 0.9, 0), produced by the compositing kernel
        if self.colormap.device != semantic_labels.device:
            self.colormap = self.colormap.to(semantic_labels.device)  # once, not per call (H2D copies synchronise)
        cm = self.colormap[semantic_labels]
        outputs["semantics_colormap"] = cm.repeat(1, 3) if repeat_colormap else cm
        outputs["_ctx"] = ctx
        return outputs

    @torch.no_grad()
    def get_inference_outputs(self, ray_bundle: RayBundle):  # fruit_nerf.py:272-314
        outputs, ctx = self._render(ray_bundle)
        return self._finish_outputs(ray_bundle, outputs, ctx, repeat_colormap=True)

    def get_outputs(self, ray_bundle: RayBundle, jitter: Optional[List[Tensor]] = None):  # fruit_nerf.py:316-357
        if self.training and torch.is_grad_enabled():
            from .training import render_with_grad  # differentiable path (custom autograd.Function)
            outputs, ctx = render_with_grad(self, ray_bundle, jitter)
        else:
            with torch.no_grad():
                outputs, ctx = self._render(ray_bundle, jitter)
        return self._finish_outputs(ray_bundle, outputs, ctx, repeat_colormap=False)

    @torch.no_grad()
    def get_export_outputs(self, ray_bundle: RayBundle):  # fruit_nerf.py:251-269
        outputs = {}
        ray_samples = self.proposal_sampler(ray_bundle)
        field_outputs = self.field.forward(ray_samples)
        outputs["rgb"] = field_outputs[FieldHeadNames.RGB]
        outputs["point_location"] = ray_samples.frustums.get_positions()
        outputs["semantics"] = field_outputs[FieldHeadNames.SEMANTICS][..., 0]
        outputs["density"] = field_outputs[FieldHeadNames.DENSITY][..., 0]
        semantic_labels = torch.sigmoid(outputs["semantics"])
        threshold = 0.9
        semantic_labels = torch.heaviside(semantic_labels - threshold,
                                          torch.tensor(0.0, device=semantic_labels.device)).to(torch.long)
        outputs["semantics_colormap"] = semantic_labels
        return outputs

    @torch.no_grad()
    def export_lattice_batch(self, lat: "K.LatticeArg", direction: Tensor, ray_begin: int, n_rays: int):
        """Fused get_export_outputs for lattice rays [ray_begin, ray_begin+n_rays): per-sample density, rgb,
        logit without materialising positions / labels (they are re-derived inside fnr_export_compact)."""
        self.arena()
        fld = self.field
        net = fld.net_struct()
        feats, selector = K.hash_encode_lattice(net.grid, fld.warp_struct(), lat, ray_begin, n_rays)
        dirs = direction.reshape(1, 3).expand(n_rays, 3).contiguous()
        rays = K.RaysArg(dirs, dirs, None, None)  # the MLP stage only reads directions
        density, rgb, logit, _ = K.field_mlp_fwd(net, rays, lat.c.n_z, feats, selector, fld._mean_embedding())
        return density, rgb, logit

    def forward(self, ray_bundle: RayBundle, jitter: Optional[List[Tensor]] = None):  # fruit_nerf.py:374-394
        ray_bundle = self._collide(ray_bundle)
        if self.test_mode == "inference":
            return self.get_inference_outputs(ray_bundle)
        elif self.test_mode == "export":
            return self.get_export_outputs(ray_bundle)
        return self.get_outputs(ray_bundle, jitter=jitter)

    @torch.no_grad()
    def get_outputs_for_camera_ray_bundle(self, camera_ray_bundle: RayBundle, rank: int = 0,
                                          world_size: int = 1) -> Dict[str, Tensor]:
        """fruit_nerf.py:225-249: chunked full-image evaluation.  The reference moves every chunk to the CPU
        (`output.cpu()`, :245 — one blocking D2H per output and chunk); here the chunks stay on the device, are
        concatenated there and the image is returned on the device (get_image_metrics_and_images moves its inputs to
        the device anyway).  `config.eval_outputs_on_cpu = True` restores the reference's CPU tensors with ONE copy
        per output after the last chunk.

        world_size > 1: each rank renders a contiguous block of whole image rows with the same chunking;
        the blocks are all-gathered in rank order, so every rank returns the full image the single process returns."""
        num_rays_per_chunk = self.config.eval_num_rays_per_chunk
        image_height, image_width = camera_ray_bundle.origins.shape[:2]
        num_rays = len(camera_ray_bundle)
        outputs_lists = defaultdict(list)
        if world_size <= 1:
            for i in range(0, num_rays, num_rays_per_chunk):
                ray_bundle = camera_ray_bundle.get_row_major_sliced_ray_bundle(i, i + num_rays_per_chunk)
                outputs = self.forward(ray_bundle=ray_bundle)
                for output_name, output in outputs.items():
                    if not torch.is_tensor(output):
                        continue
                    outputs_lists[output_name].append(output)
            full = {k: torch.cat(v).view(image_height, image_width, -1) for k, v in outputs_lists.items()}
            return {k: v.cpu() for k, v in full.items()} if getattr(self.config, "eval_outputs_on_cpu", False) else full
        from .sharding import all_gather_rows, shard_range
        lo, hi = shard_range(num_rays, rank, world_size, granule=image_width)
        for i in range(lo, hi, num_rays_per_chunk):
            ray_bundle = camera_ray_bundle.get_row_major_sliced_ray_bundle(i, min(i + num_rays_per_chunk, hi))
            outputs = self.forward(ray_bundle=ray_bundle)
            for output_name, output in outputs.items():
                if torch.is_tensor(output):
                    outputs_lists[output_name].append(output)
        if not outputs_lists:   # a rank without rows (more ranks than image rows) still joins the collectives: the
            outputs_lists = self._eval_output_templates(camera_ray_bundle)   # key set is fixed by the model
        full = {k: all_gather_rows(torch.cat(v), world_size) for k, v in sorted(outputs_lists.items())}
        full = {k: v.view(image_height, image_width, -1) for k, v in full.items()}
        return {k: v.cpu() for k, v in full.items()} if getattr(self.config, "eval_outputs_on_cpu", False) else full

    def _eval_output_templates(self, camera_ray_bundle: RayBundle) -> Dict[str, list]:
        """Zero-row tensors with the keys / dtypes / trailing shapes of an eval forward (from a one-ray pass)."""
        one = self.forward(ray_bundle=camera_ray_bundle.get_row_major_sliced_ray_bundle(0, 1))
        return {k: [v[:0]] for k, v in one.items() if torch.is_tensor(v)}

    @torch.no_grad()
    def get_image_metrics_and_images(self, outputs: Dict[str, Tensor], batch: Dict[str, Tensor]):
        """fruit_nerf.py:403-458 on the device: one entry point (fnr_image_metrics, csrc/image_metrics.hip) for everything
        the reference computes with torchmetrics, no per-metric torch launches and ONE host read of eight sums.

        PSNR and SSIM (torchmetrics defaults: 11x11 gaussian, sigma 1.5, data_range None: the images' own value range; PSNR: data_range 1) and the reference's IoU
        (fruit_nerf.py:449-453): `F.softmax(outputs["semantics"])` WITHOUT a dim on the [H,W,1] map — torch's legacy
        implicit dim for a 3-D tensor is 0, so the softmax runs over image ROWS, every value is ~1/H < 0.5, and
        BinaryJaccardIndex (threshold 0.5) sees an all-False prediction: "iou" is ~0 whatever the model learned.
        Reproduced as is under "iou"; "iou_sigmoid" additionally reports the meaningful sigmoid(semantics) > 0.5 IoU.
        Pinned against the float64 restatement oracle/image_metrics.py (tests/test_gpu_properties.py).
        Not built: LPIPS (needs pretrained weights; reported as nan) and the matplotlib colormaps of
        nerfstudio.utils.colormaps (accumulation / depth images are returned as raw single-channel maps)."""
        dev = self.device
        image = batch["image"].to(dev)
        raw_rgb = outputs["rgb"].to(dev)
        rgb = torch.clamp(raw_rgb, min=0, max=1)
        acc = outputs["accumulation"].to(dev)
        depth = outputs["depth"].to(dev)
        images_dict = {"img": torch.cat([image, rgb], dim=1), "accumulation": acc, "depth": depth}
        for i in range(self.config.num_proposal_iterations):
            images_dict[f"prop_depth_{i}"] = outputs[f"prop_depth_{i}"].to(dev)
        sem = outputs["semantics"].to(dev)
        images_dict["semantics_colormap"] = torch.sigmoid(sem)
        mask = batch["fruit_mask"].to(dev)
        images_dict["fruit_mask"] = mask.repeat(1, 1, 3)
        if sem.dim() != 3 or sem.shape[-1] != 1:
            raise ValueError("get_image_metrics_and_images expects the [H, W, 1] semantics map of a full-image render")
        sums = K.image_metrics(raw_rgb, image, sem[..., 0], mask[..., 0]).tolist()   # the one device -> host read
        sse, ssim_sum, inter_sig, union_sig, inter_row, union_row, n_ssim, n_sse = sums
        metrics_dict = {"psnr": float(-10.0 * np.log10(sse / n_sse)), "ssim": float(ssim_sum / n_ssim),
                        "lpips": float("nan"),
                        "iou": float(inter_row / max(union_row, 1.0)),     # torchmetrics: 0 when the union is empty
                        "iou_sigmoid": float(inter_sig / max(union_sig, 1.0))}
        return metrics_dict, images_dict

    # ---- losses / metrics ----------------------------------------------------------------------------------------
    def get_loss_dict(self, outputs, batch, metrics_dict=None):  # fruit_nerf.py:359-372
        from .training import fused_losses
        return fused_losses(self, outputs, batch)

    def get_metrics_dict(self, outputs, batch):  # fruit_nerf.py:396-401
        from .training import metrics
        return metrics(self, outputs, batch)
