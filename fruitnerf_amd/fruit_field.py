"""FruitField — MI355X-native mirror of /root/reference/fruit_nerf/fruit_field.py:43-301.

Same constructor arguments, registered buffers, sub-module / parameter names and method signatures as the
reference class; the arithmetic (hash-grid lookup, base / semantic / colour MLPs, SH encoding, trunc_exp)
runs in libfruitnerf_hip.so (fnr_hash_encode_fwd + fnr_field_mlp_fwd).  There is no PyTorch fallback.
"""
from __future__ import annotations

from enum import Enum
from typing import Dict, Optional, Tuple

import os

import torch
from torch import Tensor, nn

from . import _kernels as K
from . import _lib as L
from .components.field_heads import SemanticFieldHead
from .params import Embedding, HashEncoding, MLP, ParamArena
from .rays import RaySamples


class FieldHeadNames(Enum):
    """nerfstudio.field_components.field_heads.FieldHeadNames (the three members FruitField emits)."""
    RGB = "rgb"
    DENSITY = "density"
    SEMANTICS = "semantics"


class SceneContraction(nn.Module):
    """Marker for nerfstudio SceneContraction(order=inf) (fruit_nerf.py:82-85): the contraction itself is
    evaluated inside the HIP encode kernels (warp mode 0)."""

    def __init__(self, order=float("inf")):
        super().__init__()
        assert order == float("inf"), "only the L-inf contraction of Nerfacto is built"
        self.order = order


# FNR_MLP_FP32: exact fp32 MFMA chains; FNR_MLP_BF16: bf16 operands (throughput mode, not parity grade);
# FNR_MLP_BF16X3: exact three-way bf16 split, fp32-grade results on the bf16 matrix pipe (csrc/field_bf16.hpp)
MLP_MODES = {"fp32": 0, "bf16": 1, "bf16x3": 3}
# "auto" = "bf16x3", the fastest PARITY-GRADE arithmetic on MI355X for both built shapes (the fp32
# MFMA runs at 1/16 of the bf16 rate on gfx950, the exact three-way split costs 6 / 3 bf16 products per fp32 product):
#   `fruit_nerf`      every MLP kernel, 1.065 -> 0.987 ms per 4096-ray step
#   `fruit_nerf_big`  the semantic branch (30 -> 128 -> 128 -> 64 -> head) forward and backward, weight-streamed, and the
#                     colour / base backward, 6.96 -> 5.01 ms per 8192-ray step; its base + colour FORWARD stays on fp32 MFMA
# "fp32" keeps the exact fmaf-chain kernels (v_mfma_f32_16x16x4_f32); both pass the same oracle-parity tests.


class FruitField(nn.Module):
    aabb: Tensor

    def __init__(self, aabb: Tensor, num_images: int, num_layers: int = 2, hidden_dim: int = 64,
                 geo_feat_dim: int = 15, num_levels: int = 16, base_res: int = 16, max_res: int = 2048,
                 log2_hashmap_size: int = 19, num_layers_color: int = 3, num_layers_semantic: int = 2,
                 features_per_level: int = 2, hidden_dim_color: int = 64, hidden_dim_semantics: int = 64,
                 hidden_dim_transient: int = 64, appearance_embedding_dim: int = 32, use_semantics: bool = False,
                 test_mode: Optional[str] = None, num_semantic_classes: int = 100,
                 pass_semantic_gradients: bool = False, use_average_appearance_embedding: bool = False,
                 spatial_distortion: Optional[nn.Module] = None, implementation: str = "hip",
                 mlp_precision: Optional[str] = None) -> None:
        super().__init__()
        # arithmetic of the MLP GEMMs (include/fruitnerf_hip.h: FNR_MLP_*); not a reference argument.  None -> the
        # FNR_MLP_PRECISION environment variable, else "auto" = bf16x3: the exact three-way bf16 split of every fp32
        # operand on the bf16 matrix pipe — six piece products in the forward pass and in the backward's forward recompute
        # (2^-27, fp32 grade: every output parity test passes in it), THREE in the backward's dX / dW (2^-17 per product;
        # measured against the oracle at the methods' real sizes: every gradient within 1e-4 of max |g|, bar 5e-4,
        # tests/test_gpu_training_parity.py::test_losses_and_all_gradients_at_the_real_configuration).  "fp32" = fp32 MFMA
        # chains, bit for bit an fmaf chain; "bf16" = plain bf16 operands (throughput mode, not parity grade).
        self.mlp_precision = mlp_precision or os.environ.get("FNR_MLP_PRECISION", "auto")
        if self.mlp_precision not in MLP_MODES and self.mlp_precision != "auto":
            raise ValueError(f"mlp_precision {self.mlp_precision!r}: one of {sorted(MLP_MODES) + ['auto']}")
        # fruit_field.py:98-113
        self.register_buffer("aabb", aabb.clone().float())
        self.geo_feat_dim = geo_feat_dim
        self.register_buffer("max_res", torch.tensor(max_res))
        self.register_buffer("num_levels", torch.tensor(num_levels))
        self.register_buffer("log2_hashmap_size", torch.tensor(log2_hashmap_size))
        self.spatial_distortion = spatial_distortion
        self.num_images = num_images
        self.appearance_embedding_dim = appearance_embedding_dim
        self.embedding_appearance = Embedding(num_images, appearance_embedding_dim)
        self.use_average_appearance_embedding = use_average_appearance_embedding
        self.use_semantics = use_semantics
        self.test_mode = test_mode
        self.pass_semantic_gradients = pass_semantic_gradients
        self.base_res = base_res
        if pass_semantic_gradients:
            raise NotImplementedError("pass_semantic_gradients=True is not built (reference default is False, "
                                      "fruit_nerf.py:56)")
        if not use_semantics:
            raise NotImplementedError("FruitModel always builds the field with use_semantics=True (fruit_nerf.py:98)")
        # fruit_field.py:124-166 — same module names => same state-dict keys
        self.mlp_base_grid = HashEncoding(num_levels=num_levels, min_res=base_res, max_res=max_res,
                                          log2_hashmap_size=log2_hashmap_size, features_per_level=features_per_level)
        self.mlp_base_mlp = MLP(in_dim=self.mlp_base_grid.get_out_dim(), num_layers=num_layers,
                                layer_width=hidden_dim, out_dim=1 + geo_feat_dim)
        self.mlp_base = nn.Sequential(self.mlp_base_grid, self.mlp_base_mlp)
        self.mlp_semantics = MLP(in_dim=geo_feat_dim, num_layers=num_layers_semantic,
                                 layer_width=hidden_dim_semantics, out_dim=hidden_dim_transient)
        self.field_head_semantics = SemanticFieldHead(in_dim=self.mlp_semantics.get_out_dim(),
                                                      num_classes=num_semantic_classes, activation=None)
        self.mlp_head = MLP(in_dim=16 + geo_feat_dim + appearance_embedding_dim, num_layers=num_layers_color,
                            layer_width=hidden_dim_color, out_dim=3)
        self._dims = dict(num_layers=num_layers, hidden_dim=hidden_dim, num_layers_color=num_layers_color,
                          hidden_dim_color=hidden_dim_color, num_layers_semantic=num_layers_semantic,
                          hidden_dim_semantics=hidden_dim_semantics, hidden_dim_transient=hidden_dim_transient,
                          num_semantic_classes=num_semantic_classes)
        if num_layers != 2 or num_layers_color != 3 or num_semantic_classes != 1:
            raise NotImplementedError("gfx950 field kernels are built for num_layers=2, num_layers_color=3, "
                                      "num_semantic_classes=1 (what FruitModel constructs, fruit_nerf.py:88-103)")
        # the two MLP shapes compiled into libfruitnerf_hip.so (csrc/field_layers.hpp: FieldCfgBase / FieldCfgBig):
        # fail at construction, not at the first kernel call
        shape = (geo_feat_dim, num_layers_semantic, hidden_dim_semantics)
        fixed = (num_levels, features_per_level, hidden_dim, hidden_dim_color, hidden_dim_transient,
                 appearance_embedding_dim)
        if shape not in ((15, 2, 64), (30, 3, 128)) or fixed != (16, 2, 64, 64, 64, 32):
            raise NotImplementedError(
                "gfx950 field kernels are built for the `fruit_nerf` shape (geo_feat_dim 15, num_layers_semantic 2, "
                "hidden_dim_semantics 64) and the `fruit_nerf_big` / `fruit_nerf_huge` shape (30, 3, 128) with 16 levels "
                "x 2 features, hidden_dim 64, hidden_dim_color 64, hidden_dim_transient 64, appearance 32 "
                f"(fruit_nerf_config.py:27-160); got {shape} / {fixed}")
        self._arena: Optional[ParamArena] = None
        self._net_c: Optional[L.fnr_field_net] = None
        self._net_ptr_key = None
        self._last_geo = None

    # ---- parameter arena / C descriptors -------------------------------------------------------------
    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._net_c = None  # storage may have moved
        if self._arena is not None and self._arena_owner:
            self._arena = None
        return out

    _arena_owner = True

    def adopt_arena(self, arena: ParamArena) -> None:
        """Called by FruitModel, which owns one arena for field + proposal networks."""
        self._arena = arena
        self._arena_owner = False
        self._net_c = None

    def _ensure_arena(self) -> None:
        dev = self.mlp_base_grid.hash_table.device
        if dev.type != "cuda":
            raise RuntimeError("FruitField parameters are on %s; move the module to a HIP device "
                               "(fruitnerf_amd has no CPU path)" % dev)
        if self._arena is None:
            self._arena = ParamArena([("fields", list(self.parameters()))], dev)
            self._arena_owner = True
            self._net_c = None

    # ---- deferred optimiser step (data-parallel training, training.DEFER_FIELD_UPDATE) ------------------------------
    def defer_update(self, finish) -> None:
        """`finish()` waits for the field's gradient collective and applies its optimiser step; it runs before the next
        use of the field's parameters (net_struct) or their export (state_dict)."""
        self.flush_deferred_update()
        self.__dict__["_deferred_update"] = finish

    def flush_deferred_update(self) -> None:
        finish = self.__dict__.pop("_deferred_update", None)
        if finish is not None:
            finish()

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self.flush_deferred_update()
        return super()._save_to_state_dict(destination, prefix, keep_vars)

    def net_struct(self, grads: bool = False) -> L.fnr_field_net:
        """fnr_field_net over the parameters (or, grads=True, over the arena's gradient views)."""
        self.flush_deferred_update()
        self._ensure_arena()
        def P(p: nn.Parameter):
            return (p.grad if grads else p.data).data_ptr()

        key = (P(self.mlp_base_grid.hash_table), P(self.mlp_head.layers[2].bias), self.mlp_precision)
        cache = self.__dict__.setdefault("_struct_cache", {})
        hit = cache.get(grads)
        if hit is not None and hit[0] == key and self._net_c is not None:
            return hit[1]

        g = self.mlp_base_grid
        net = L.fnr_field_net()
        net.grid = K.make_grid(g.hash_table.grad if grads else g.hash_table.data, g.num_levels, g.log2_hashmap_size,
                               g.scalings)
        d = self._dims
        net.geo_feat_dim = self.geo_feat_dim
        net.hidden_dim = d["hidden_dim"]
        net.hidden_dim_color = d["hidden_dim_color"]
        net.hidden_dim_semantics = d["hidden_dim_semantics"]
        net.num_layers_semantic = d["num_layers_semantic"]
        net.semantic_out_dim = d["hidden_dim_transient"]
        net.appearance_dim = self.appearance_embedding_dim
        net.n_images = self.num_images
        b = self.mlp_base_mlp.layers
        net.base_w0, net.base_b0 = P(b[0].weight), P(b[0].bias)
        net.base_w1, net.base_b1 = P(b[1].weight), P(b[1].bias)
        for i, lyr in enumerate(self.mlp_semantics.layers):
            net.sem_w[i], net.sem_b[i] = P(lyr.weight), P(lyr.bias)
        net.head_w, net.head_b = P(self.field_head_semantics.net.weight), P(self.field_head_semantics.net.bias)
        for i, lyr in enumerate(self.mlp_head.layers):
            net.col_w[i], net.col_b[i] = P(lyr.weight), P(lyr.bias)
        net.embedding = P(self.embedding_appearance.embedding.weight)
        net.mlp_mode = MLP_MODES[self.resolved_mlp_precision()]
        cache[grads] = (key, net)
        self._net_c = net  # non-None marks the cache valid (reset by _apply / adopt_arena)
        return net

    def resolved_mlp_precision(self) -> str:
        if self.mlp_precision != "auto":
            return self.mlp_precision
        return "bf16x3"

    def warp_struct(self) -> L.fnr_warp:
        # fruit_field.py:169-175: contraction + (x+2)/4, or aabb normalisation when spatial_distortion is None
        return K.make_warp(0 if self.spatial_distortion is not None else 1, self.aabb)

    def _uses_mean_embedding(self) -> bool:
        # fruit_field.py:293-295 (inference/export always mean) ; :250-262 (eval: mean iff use_average...)
        if self.test_mode in ("inference", "export"):
            return True
        return not self.training

    def _mean_embedding(self) -> Tensor:
        if self.test_mode in ("inference", "export") or self.use_average_appearance_embedding:
            return K.embedding_mean(self.embedding_appearance.embedding.weight.data)
        return torch.zeros(self.appearance_embedding_dim, device=self.aabb.device)  # fruit_field.py:258-261

    # ---- evaluation -------------------------------------------------------------------------------------
    def _flatten(self, ray_samples: RaySamples):
        """Generic RaySamples -> per-sample 'rays' with S=1 (euclid bins = [start, end])."""
        fr = ray_samples.frustums
        shape = tuple(fr.shape)
        starts, ends = fr.starts.reshape(-1, 1).float(), fr.ends.reshape(-1, 1).float()
        euclid = torch.cat([starts, ends], dim=-1).contiguous()
        cam = ray_samples.camera_indices
        rays = K.RaysArg(fr.origins.reshape(-1, 3), fr.directions.reshape(-1, 3), None, None,
                         None if cam is None else cam.reshape(-1))
        return rays, euclid, 1, shape

    def _evaluate(self, ray_samples: RaySamples, want_geo: bool):
        self._ensure_arena()
        if getattr(ray_samples, "_structured", None) is not None:
            rays, euclid, S = ray_samples._structured
            shape = (rays.n, S)
        else:
            rays, euclid, S, shape = self._flatten(ray_samples)
        mean_emb = self._mean_embedding() if self._uses_mean_embedding() else None
        if mean_emb is None and rays.cam is None:
            raise AttributeError("Camera indices are not provided.")  # fruit_field.py:240-241
        self._last_ray_samples = ray_samples
        if torch.is_grad_enabled() and self.training and mean_emb is None:
            # differentiable field query (autograd.Function over the same kernels; gradients land in the arena)
            from .training import field_with_grad
            density, rgb, logit, geo = field_with_grad(self, rays, euclid, S, want_geo)
        else:
            with torch.no_grad():
                net = self.net_struct()
                feats, selector = K.hash_encode_fwd(net.grid, self.warp_struct(), rays, euclid, S)
                density, rgb, logit, geo = K.field_mlp_fwd(net, rays, S, feats, selector, mean_emb, want_geo=want_geo)
        return density.view(*shape, 1), rgb.view(*shape, 3), logit.view(*shape, 1), \
            (None if geo is None else geo.view(*shape, self.geo_feat_dim))

    def get_density(self, ray_samples: RaySamples) -> Tuple[Tensor, Tensor]:
        """fruit_field.py:168-193 -> (density [...,1], base_mlp_out [...,geo]).

        The HIP field evaluates density, semantics and colour in ONE fused pass; the rgb / semantics of that pass are
        kept for the get_outputs call that follows (the reference's forward(), fruit_field.py:283-301).  In training
        mode with autograd enabled the pass is differentiable with respect to the field's parameters (density, rgb
        and semantics carry one shared autograd node); `base_mlp_out` is returned detached — the kernels route the
        colour branch's gradient into the base MLP internally, exactly as autograd would through the reference's graph."""
        density, rgb, logit, geo = self._evaluate(ray_samples, want_geo=True)
        self._last = (ray_samples, rgb, logit)
        self._last_geo = geo
        return density, geo

    # ---- side-effect attributes of the reference's get_density (fruit_field.py:180-186) -----------------------------
    # Nerfstudio reads them for normals (Field.get_normals); FruitNeRF's hot path never does, so they are produced on
    # demand from the last ray samples instead of costing a launch + 16 B/sample in every field query.
    @property
    def _sample_locations(self) -> Tensor:
        """The masked unit-cube positions of the last get_density call, requires_grad=True (fruit_field.py:169-182)."""
        rs = getattr(self, "_last_ray_samples", None)
        if rs is None:
            raise AttributeError("_sample_locations is set by get_density / forward")
        with torch.no_grad():
            positions = rs.frustums.get_positions().float()
            if self.spatial_distortion is not None:      # SceneContraction(order=inf), then (x + 2) / 4
                mag = positions.abs().amax(dim=-1, keepdim=True)
                positions = torch.where(mag < 1, positions, (2 - 1 / mag) * (positions / mag))
                positions = (positions + 2.0) / 4.0
            else:                                        # SceneBox.get_normalized_positions
                positions = (positions - self.aabb[0]) / (self.aabb[1] - self.aabb[0])
            selector = ((positions > 0.0) & (positions < 1.0)).all(dim=-1)
            positions = positions * selector[..., None]
        return positions.requires_grad_(True)

    @property
    def _density_before_activation(self) -> Tensor:
        """Raw density output of mlp_base for the last get_density call [..., 1] (fruit_field.py:185-186): the first
        column of the base MLP's output h, re-evaluated on demand."""
        rs = getattr(self, "_last_ray_samples", None)
        if rs is None:
            raise AttributeError("_density_before_activation is set by get_density / forward")
        with torch.no_grad():
            if getattr(rs, "_structured", None) is not None:
                rays, euclid, S = rs._structured
                shape = (rays.n, S)
            else:
                rays, euclid, S, shape = self._flatten(rs)
            net = self.net_struct()
            feats, selector = K.hash_encode_fwd(net.grid, self.warp_struct(), rays, euclid, S)
            emb = self._mean_embedding() if (self._uses_mean_embedding() or rays.cam is None) else None
            h = K.field_mlp_fwd(net, rays, S, feats, selector, emb, want_h=True)[4][0]
        return h[:, 0].reshape(*shape, 1)

    def get_normals(self) -> Tensor:
        """nerfstudio Field.get_normals differentiates `_density_before_activation` with respect to `_sample_locations`
        through autograd.  Here both are produced on demand by kernels (no autograd graph joins them), and no FruitNeRF
        code path asks for normals (the reference constructs the field without predict_normals): unsupported, loudly."""
        raise NotImplementedError("FruitField.get_normals: normals are not built (FruitNeRF never computes them; "
                                  "_sample_locations / _density_before_activation are kernel outputs without an autograd "
                                  "graph between them)")

    def release_last_samples(self) -> None:
        """Drop the references get_density keeps for _sample_locations / _density_before_activation / get_outputs (the
        last batch's ray samples and per-sample outputs stay alive until the next call otherwise)."""
        self._last_ray_samples = None
        self._last = None
        self._last_geo = None

    def _outputs_from_last(self, ray_samples, density_embedding):
        if density_embedding is None or density_embedding is not self._last_geo:
            raise NotImplementedError(
                "the HIP field evaluates density, semantics and colour in one fused pass; call get_outputs with "
                "the density_embedding returned by the immediately preceding get_density (or use forward())")
        _, rgb, logit = self._last
        return {FieldHeadNames.SEMANTICS: logit, FieldHeadNames.RGB: rgb}

    def get_outputs(self, ray_samples: RaySamples, density_embedding: Optional[Tensor] = None):
        assert density_embedding is not None  # fruit_field.py:237
        return self._outputs_from_last(ray_samples, density_embedding)

    def get_inference_outputs(self, ray_samples: RaySamples, density_embedding: Optional[Tensor] = None,
                              render_rgb: bool = False):
        return self._outputs_from_last(ray_samples, density_embedding)

    def forward(self, ray_samples: RaySamples) -> Dict[FieldHeadNames, Tensor]:
        """fruit_field.py:283-301.  Differentiable w.r.t. the field's parameters in training mode (see get_density);
        FruitModel's own training step uses the fused render function, which also covers compositing."""
        density, rgb, logit, _ = self._evaluate(ray_samples, want_geo=False)
        return {FieldHeadNames.SEMANTICS: logit, FieldHeadNames.RGB: rgb, FieldHeadNames.DENSITY: density}
