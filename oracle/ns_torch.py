"""Oracle part 1: restatement of the nerfstudio==0.3.2 torch-fallback components that
FruitNeRF's hot path executes (TEST INFRASTRUCTURE, parity unpinned: see oracle/__init__.py).

nerfstudio is NOT in /root/reference (pinned dependency, /root/reference/pyproject.toml:10) and
not installable here, so every class below restates the published 0.3.2 algorithm; the reference
call site that reaches it is cited instead.  Pure PyTorch, CPU, fp32.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor, nn

# --------------------------------------------------------------------------------------
# Ray containers  (nerfstudio.cameras.rays; used at /root/reference/fruit_nerf/fruit_nerf.py:19,
# components/ray_samplers.py:27, components/ray_generators.py:21)
# --------------------------------------------------------------------------------------


@dataclass
class Frustums:
    origins: Tensor  # [..., 3]
    directions: Tensor  # [..., 3]
    starts: Tensor  # [..., 1]
    ends: Tensor  # [..., 1]
    pixel_area: Tensor  # [..., 1]

    def get_positions(self) -> Tensor:
        # fruit_field.py:171,175 ; fruit_nerf.py:259
        return self.origins + self.directions * (self.starts + self.ends) / 2

    @property
    def shape(self):
        return self.origins.shape[:-1]


@dataclass
class RaySamples:
    frustums: Frustums
    camera_indices: Optional[Tensor] = None  # [..., 1]
    deltas: Optional[Tensor] = None  # [..., 1]
    spacing_starts: Optional[Tensor] = None
    spacing_ends: Optional[Tensor] = None
    spacing_to_euclidean_fn: Optional[Callable] = None

    @property
    def shape(self):
        return self.frustums.shape

    def get_weights(self, densities: Tensor) -> Tensor:
        # RaySamples.get_weights, called at fruit_nerf.py:283,325 and inside the proposal sampler
        delta_density = self.deltas * densities
        alphas = 1 - torch.exp(-delta_density)
        transmittance = torch.cumsum(delta_density[..., :-1, :], dim=-2)
        transmittance = torch.cat(
            [torch.zeros((*transmittance.shape[:1], 1, 1), device=densities.device), transmittance], dim=-2
        )
        transmittance = torch.exp(-transmittance)
        weights = alphas * transmittance
        weights = torch.nan_to_num(weights)
        return weights


@dataclass
class RayBundle:
    origins: Tensor  # [R, 3]
    directions: Tensor  # [R, 3]
    pixel_area: Tensor  # [R, 1]
    camera_indices: Optional[Tensor] = None  # [R, 1]
    nears: Optional[Tensor] = None  # [R, 1]
    fars: Optional[Tensor] = None  # [R, 1]

    def __len__(self):
        return self.origins.shape[0]

    def get_ray_samples(self, bin_starts, bin_ends, spacing_starts=None, spacing_ends=None,
                        spacing_to_euclidean_fn=None) -> RaySamples:
        deltas = bin_ends - bin_starts
        camera_indices = self.camera_indices[..., None] if self.camera_indices is not None else None
        S = bin_starts.shape[-2]
        frustums = Frustums(
            origins=self.origins[:, None, :].expand(-1, S, -1),
            directions=self.directions[:, None, :].expand(-1, S, -1),
            starts=bin_starts,
            ends=bin_ends,
            pixel_area=self.pixel_area[:, None, :].expand(-1, S, -1),
        )
        if camera_indices is not None:
            camera_indices = camera_indices.expand(-1, S, -1)
        # nerfstudio's RaySamples is a TensorDataclass: every field is broadcast to the batch shape [R, S]
        # (eval-mode spacing bins start life as [1, S+1])
        R = self.origins.shape[0]
        if spacing_starts is not None:
            spacing_starts = spacing_starts.expand(R, S, 1)
            spacing_ends = spacing_ends.expand(R, S, 1)
        return RaySamples(frustums, camera_indices, deltas, spacing_starts, spacing_ends, spacing_to_euclidean_fn)


# --------------------------------------------------------------------------------------
# Field components
# --------------------------------------------------------------------------------------


class HashEncoding(nn.Module):
    """nerfstudio HashEncoding, torch path.  Constructed at fruit_field.py:124-131 and inside
    HashMLPDensityField (fruit_nerf.py:111-127).  Every level is hashed (no dense levels), no +0.5
    offset, ceil/floor corner pairs, scalings from a float32 pow."""

    def __init__(self, num_levels=16, min_res=16, max_res=1024, log2_hashmap_size=19,
                 features_per_level=2, hash_init_scale=0.001):
        super().__init__()
        self.num_levels = num_levels
        self.features_per_level = features_per_level
        self.hash_table_size = 2 ** log2_hashmap_size
        levels = torch.arange(num_levels)
        self.growth_factor = np.exp((np.log(max_res) - np.log(min_res)) / (num_levels - 1)) if num_levels > 1 else 1
        self.scalings = torch.floor(min_res * self.growth_factor ** levels)
        self.hash_offset = levels * self.hash_table_size
        table = torch.rand(size=(self.hash_table_size * num_levels, features_per_level)) * 2 - 1
        table *= hash_init_scale
        self.hash_table = nn.Parameter(table)

    def get_out_dim(self) -> int:
        return self.num_levels * self.features_per_level

    def hash_fn(self, in_tensor: Tensor) -> Tensor:
        in_tensor = in_tensor * torch.tensor([1, 2654435761, 805459861])
        x = torch.bitwise_xor(in_tensor[..., 0], in_tensor[..., 1])
        x = torch.bitwise_xor(x, in_tensor[..., 2])
        x %= self.hash_table_size
        x += self.hash_offset
        return x

    def forward(self, in_tensor: Tensor) -> Tensor:
        assert in_tensor.shape[-1] == 3
        in_tensor = in_tensor[..., None, :]
        scaled = in_tensor * self.scalings.view(-1, 1)
        scaled_c = torch.ceil(scaled).type(torch.int32)
        scaled_f = torch.floor(scaled).type(torch.int32)
        offset = scaled - scaled_f
        c, f = scaled_c, scaled_f
        hashed_0 = self.hash_fn(c)
        hashed_1 = self.hash_fn(torch.cat([c[..., 0:1], f[..., 1:2], c[..., 2:3]], dim=-1))
        hashed_2 = self.hash_fn(torch.cat([f[..., 0:1], f[..., 1:2], c[..., 2:3]], dim=-1))
        hashed_3 = self.hash_fn(torch.cat([f[..., 0:1], c[..., 1:2], c[..., 2:3]], dim=-1))
        hashed_4 = self.hash_fn(torch.cat([c[..., 0:1], c[..., 1:2], f[..., 2:3]], dim=-1))
        hashed_5 = self.hash_fn(torch.cat([c[..., 0:1], f[..., 1:2], f[..., 2:3]], dim=-1))
        hashed_6 = self.hash_fn(f)
        hashed_7 = self.hash_fn(torch.cat([f[..., 0:1], c[..., 1:2], f[..., 2:3]], dim=-1))
        f_0 = self.hash_table[hashed_0]
        f_1 = self.hash_table[hashed_1]
        f_2 = self.hash_table[hashed_2]
        f_3 = self.hash_table[hashed_3]
        f_4 = self.hash_table[hashed_4]
        f_5 = self.hash_table[hashed_5]
        f_6 = self.hash_table[hashed_6]
        f_7 = self.hash_table[hashed_7]
        f_03 = f_0 * offset[..., 0:1] + f_3 * (1 - offset[..., 0:1])
        f_12 = f_1 * offset[..., 0:1] + f_2 * (1 - offset[..., 0:1])
        f_56 = f_5 * offset[..., 0:1] + f_6 * (1 - offset[..., 0:1])
        f_47 = f_4 * offset[..., 0:1] + f_7 * (1 - offset[..., 0:1])
        f0312 = f_03 * offset[..., 1:2] + f_12 * (1 - offset[..., 1:2])
        f4756 = f_47 * offset[..., 1:2] + f_56 * (1 - offset[..., 1:2])
        encoded_value = f0312 * offset[..., 2:3] + f4756 * (1 - offset[..., 2:3])
        return torch.flatten(encoded_value, start_dim=-2, end_dim=-1)


class MLP(nn.Module):
    """nerfstudio MLP, torch path (fruit_field.py:132-140,145-153,158-166): `num_layers` biased
    nn.Linear layers, activation after all but the last, then out_activation."""

    def __init__(self, in_dim, num_layers, layer_width, out_dim=None, activation=nn.ReLU(), out_activation=None):
        super().__init__()
        self.in_dim = in_dim
        self.out_dim = out_dim if out_dim is not None else layer_width
        self.activation = activation
        self.out_activation = out_activation
        layers = []
        if num_layers == 1:
            layers.append(nn.Linear(in_dim, self.out_dim))
        else:
            for i in range(num_layers - 1):
                layers.append(nn.Linear(in_dim if i == 0 else layer_width, layer_width))
            layers.append(nn.Linear(layer_width, self.out_dim))
        self.layers = nn.ModuleList(layers)

    def get_out_dim(self):
        return self.out_dim

    def forward(self, x: Tensor) -> Tensor:
        for i, layer in enumerate(self.layers):
            x = layer(x)
            if self.activation is not None and i < len(self.layers) - 1:
                x = self.activation(x)
        if self.out_activation is not None:
            x = self.out_activation(x)
        return x


def components_from_spherical_harmonics(levels: int, directions: Tensor) -> Tensor:
    """SH basis on (un-normalised, shifted) directions — SHEncoding(levels=4) torch path,
    fruit_field.py:115-118,208-210,243-245."""
    num_components = levels ** 2
    components = torch.zeros((*directions.shape[:-1], num_components), device=directions.device)
    assert 1 <= levels <= 4
    x = directions[..., 0]
    y = directions[..., 1]
    z = directions[..., 2]
    xx = x ** 2
    yy = y ** 2
    zz = z ** 2
    components[..., 0] = 0.28209479177387814
    if levels > 1:
        components[..., 1] = 0.4886025119029199 * y
        components[..., 2] = 0.4886025119029199 * z
        components[..., 3] = 0.4886025119029199 * x
    if levels > 2:
        components[..., 4] = 1.0925484305920792 * x * y
        components[..., 5] = 1.0925484305920792 * y * z
        components[..., 6] = 0.9461746957575601 * zz - 0.31539156525251999
        components[..., 7] = 1.0925484305920792 * x * z
        components[..., 8] = 0.5462742152960396 * (xx - yy)
    if levels > 3:
        components[..., 9] = 0.5900435899266435 * y * (3 * xx - yy)
        components[..., 10] = 2.890611442640554 * x * y * z
        components[..., 11] = 0.4570457994644658 * y * (5 * zz - 1)
        components[..., 12] = 0.3731763325901154 * z * (5 * zz - 3)
        components[..., 13] = 0.4570457994644658 * x * (5 * zz - 1)
        components[..., 14] = 1.445305721320277 * z * (xx - yy)
        components[..., 15] = 0.5900435899266435 * x * (xx - 3 * yy)
    return components


class SHEncoding(nn.Module):
    def __init__(self, levels=4):
        super().__init__()
        self.levels = levels

    def get_out_dim(self):
        return self.levels ** 2

    @torch.no_grad()
    def forward(self, in_tensor: Tensor) -> Tensor:
        return components_from_spherical_harmonics(levels=self.levels, directions=in_tensor)


def shift_directions_for_tcnn(directions: Tensor) -> Tensor:
    return (directions + 1.0) / 2.0


class Embedding(nn.Module):
    """nerfstudio Embedding (fruit_field.py:108,219,251,256)."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.embedding = nn.Embedding(in_dim, out_dim)

    def mean(self, dim=0):
        return self.embedding.weight.mean(dim)

    def forward(self, in_tensor: Tensor) -> Tensor:
        return self.embedding(in_tensor)


class _TruncExp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        x = ctx.saved_tensors[0]
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _TruncExp.apply  # fruit_field.py:191


class SceneContraction(nn.Module):
    """order=inf contraction (fruit_nerf.py:82-85; applied fruit_field.py:170-173)."""

    def __init__(self, order=float("inf")):
        super().__init__()
        self.order = order

    def forward(self, positions: Tensor) -> Tensor:
        mag = torch.linalg.norm(positions, ord=self.order, dim=-1)[..., None]
        return torch.where(mag < 1, positions, (2 - (1 / mag)) * (positions / mag))


def get_normalized_positions(positions: Tensor, aabb: Tensor) -> Tensor:
    # SceneBox.get_normalized_positions, fruit_field.py:175
    aabb_lengths = aabb[1] - aabb[0]
    return (positions - aabb[0]) / aabb_lengths


class HashMLPDensityField(nn.Module):
    """Proposal network (fruit_nerf.py:104-129): hash -> MLP -> trunc_exp, masked."""

    def __init__(self, aabb, num_layers=2, hidden_dim=64, spatial_distortion=None, use_linear=False,
                 num_levels=8, max_res=1024, base_res=16, log2_hashmap_size=18, features_per_level=2):
        super().__init__()
        assert not use_linear
        self.register_buffer("aabb", aabb)
        self.spatial_distortion = spatial_distortion
        self.encoding = HashEncoding(num_levels=num_levels, min_res=base_res, max_res=max_res,
                                     log2_hashmap_size=log2_hashmap_size, features_per_level=features_per_level)
        network = MLP(in_dim=self.encoding.get_out_dim(), num_layers=num_layers, layer_width=hidden_dim,
                      out_dim=1, activation=nn.ReLU(), out_activation=None)
        self.mlp_base = nn.Sequential(self.encoding, network)

    def density_fn(self, positions: Tensor) -> Tensor:
        # Field.density_fn wraps the points into zero-length frustums; get_positions() == positions
        if self.spatial_distortion is not None:
            positions = self.spatial_distortion(positions)
            positions = (positions + 2.0) / 4.0
        else:
            positions = get_normalized_positions(positions, self.aabb)
        selector = ((positions > 0.0) & (positions < 1.0)).all(dim=-1)
        positions = positions * selector[..., None]
        lead = positions.shape[:-1]
        density_before_activation = self.mlp_base(positions.view(-1, 3)).view(*lead, -1).to(positions)
        density = trunc_exp(density_before_activation)
        density = density * selector[..., None]
        return density


# --------------------------------------------------------------------------------------
# Samplers  (nerfstudio.model_components.ray_samplers; fruit_nerf.py:38,151-158)
# --------------------------------------------------------------------------------------


class SpacedSampler(nn.Module):
    def __init__(self, spacing_fn, spacing_fn_inv, num_samples=None, train_stratified=True, single_jitter=False):
        super().__init__()
        self.num_samples = num_samples
        self.train_stratified = train_stratified
        self.single_jitter = single_jitter
        self.spacing_fn = spacing_fn
        self.spacing_fn_inv = spacing_fn_inv

    def forward(self, ray_bundle: RayBundle, num_samples: Optional[int] = None, t_rand: Optional[Tensor] = None):
        """`t_rand` (not in nerfstudio): externally supplied jitter so that oracle and HIP path
        consume the same random numbers; None -> torch.rand like nerfstudio."""
        num_samples = num_samples or self.num_samples
        num_rays = ray_bundle.origins.shape[0]
        bins = torch.linspace(0.0, 1.0, num_samples + 1).to(ray_bundle.origins.device)[None, ...]
        if self.train_stratified and self.training:
            if t_rand is None:
                if self.single_jitter:
                    t_rand = torch.rand((num_rays, 1), dtype=bins.dtype, device=bins.device)
                else:
                    t_rand = torch.rand((num_rays, num_samples + 1), dtype=bins.dtype, device=bins.device)
            bin_centers = (bins[..., 1:] + bins[..., :-1]) / 2.0
            bin_upper = torch.cat([bin_centers, bins[..., -1:]], -1)
            bin_lower = torch.cat([bins[..., :1], bin_centers], -1)
            bins = bin_lower + (bin_upper - bin_lower) * t_rand
        s_near, s_far = (self.spacing_fn(x) for x in (ray_bundle.nears, ray_bundle.fars))

        def spacing_to_euclidean_fn(x):
            return self.spacing_fn_inv(x * s_far + (1 - x) * s_near)

        euclidean_bins = spacing_to_euclidean_fn(bins)
        return ray_bundle.get_ray_samples(
            bin_starts=euclidean_bins[..., :-1, None],
            bin_ends=euclidean_bins[..., 1:, None],
            spacing_starts=bins[..., :-1, None],
            spacing_ends=bins[..., 1:, None],
            spacing_to_euclidean_fn=spacing_to_euclidean_fn,
        )


class UniformLinDispPiecewiseSampler(SpacedSampler):
    def __init__(self, num_samples=None, train_stratified=True, single_jitter=False):
        super().__init__(
            num_samples=num_samples,
            spacing_fn=lambda x: torch.where(x < 1, x / 2, 1 - 1 / (2 * x)),
            spacing_fn_inv=lambda x: torch.where(x < 0.5, 2 * x, 1 / (2 - 2 * x)),
            train_stratified=train_stratified,
            single_jitter=single_jitter,
        )


class PDFSampler(nn.Module):
    def __init__(self, num_samples=None, train_stratified=True, single_jitter=False, include_original=True,
                 histogram_padding=0.01):
        super().__init__()
        self.num_samples = num_samples
        self.train_stratified = train_stratified
        self.include_original = include_original
        self.histogram_padding = histogram_padding
        self.single_jitter = single_jitter

    def forward(self, ray_bundle: RayBundle, ray_samples: RaySamples, weights: Tensor,
                num_samples: Optional[int] = None, eps: float = 1e-5, rand: Optional[Tensor] = None) -> RaySamples:
        num_samples = num_samples or self.num_samples
        num_bins = num_samples + 1
        weights = weights[..., 0] + self.histogram_padding
        weights_sum = torch.sum(weights, dim=-1, keepdim=True)
        padding = torch.relu(eps - weights_sum)
        weights = weights + padding / weights.shape[-1]
        weights_sum += padding
        pdf = weights / weights_sum
        cdf = torch.min(torch.ones_like(pdf), torch.cumsum(pdf, dim=-1))
        cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)
        if self.train_stratified and self.training:
            u = torch.linspace(0.0, 1.0 - (1.0 / num_bins), steps=num_bins, device=cdf.device)
            u = u.expand(size=(*cdf.shape[:-1], num_bins))
            if rand is None:
                if self.single_jitter:
                    rand = torch.rand((*cdf.shape[:-1], 1), device=cdf.device)
                else:
                    rand = torch.rand((*cdf.shape[:-1], num_samples + 1), device=cdf.device)
            u = u + rand / num_bins
        else:
            u = torch.linspace(0.0, 1.0 - (1.0 / num_bins), steps=num_bins, device=cdf.device)
            u = u + 1.0 / (2 * num_bins)
            u = u.expand(size=(*cdf.shape[:-1], num_bins))
        u = u.contiguous()
        existing_bins = torch.cat([ray_samples.spacing_starts[..., 0], ray_samples.spacing_ends[..., -1:, 0]], dim=-1)
        inds = torch.searchsorted(cdf, u, side="right")
        below = torch.clamp(inds - 1, 0, existing_bins.shape[-1] - 1)
        above = torch.clamp(inds, 0, existing_bins.shape[-1] - 1)
        cdf_g0 = torch.gather(cdf, -1, below)
        bins_g0 = torch.gather(existing_bins, -1, below)
        cdf_g1 = torch.gather(cdf, -1, above)
        bins_g1 = torch.gather(existing_bins, -1, above)
        t = torch.clip(torch.nan_to_num((u - cdf_g0) / (cdf_g1 - cdf_g0), 0), 0, 1)
        bins = bins_g0 + t * (bins_g1 - bins_g0)
        if self.include_original:
            bins, _ = torch.sort(torch.cat([existing_bins, bins], -1), -1)
        bins = bins.detach()
        euclidean_bins = ray_samples.spacing_to_euclidean_fn(bins)
        return ray_bundle.get_ray_samples(
            bin_starts=euclidean_bins[..., :-1, None],
            bin_ends=euclidean_bins[..., 1:, None],
            spacing_starts=bins[..., :-1, None],
            spacing_ends=bins[..., 1:, None],
            spacing_to_euclidean_fn=ray_samples.spacing_to_euclidean_fn,
        )


class ProposalNetworkSampler(nn.Module):
    """fruit_nerf.py:151-158.  `jitter` (extension): list of per-level random tensors
    [t_rand(R,1), rand1(R,1), rand2(R,1)] consumed instead of torch.rand."""

    def __init__(self, num_proposal_samples_per_ray=(64,), num_nerf_samples_per_ray=32,
                 num_proposal_network_iterations=2, single_jitter=False,
                 update_sched: Callable = lambda x: 1, initial_sampler=None):
        super().__init__()
        self.num_proposal_samples_per_ray = num_proposal_samples_per_ray
        self.num_nerf_samples_per_ray = num_nerf_samples_per_ray
        self.num_proposal_network_iterations = num_proposal_network_iterations
        self.update_sched = update_sched
        if initial_sampler is None:
            self.initial_sampler = UniformLinDispPiecewiseSampler(single_jitter=single_jitter)
        else:
            self.initial_sampler = initial_sampler
        self.pdf_sampler = PDFSampler(include_original=False, single_jitter=single_jitter)
        self._anneal = 1.0
        self._steps_since_update = 0
        self._step = 0

    def set_anneal(self, anneal: float) -> None:
        self._anneal = anneal

    def step_cb(self, step):
        self._step = step
        self._steps_since_update += 1

    def forward(self, ray_bundle: RayBundle, density_fns: List[Callable], jitter: Optional[List[Tensor]] = None
                ) -> Tuple[RaySamples, List, List]:
        weights_list = []
        ray_samples_list = []
        n = self.num_proposal_network_iterations
        weights = None
        ray_samples = None
        updated = self._steps_since_update > self.update_sched(self._step) or self._step < 10
        for i_level in range(n + 1):
            is_prop = i_level < n
            num_samples = self.num_proposal_samples_per_ray[i_level] if is_prop else self.num_nerf_samples_per_ray
            jit = None if jitter is None else jitter[i_level]
            if i_level == 0:
                ray_samples = self.initial_sampler(ray_bundle, num_samples=num_samples, t_rand=jit)
            else:
                annealed_weights = torch.pow(weights, self._anneal)
                ray_samples = self.pdf_sampler(ray_bundle, ray_samples, annealed_weights, num_samples=num_samples,
                                               rand=jit)
            if is_prop:
                if updated:
                    density = density_fns[i_level](ray_samples.frustums.get_positions())
                else:
                    with torch.no_grad():
                        density = density_fns[i_level](ray_samples.frustums.get_positions())
                weights = ray_samples.get_weights(density)
                weights_list.append(weights)
                ray_samples_list.append(ray_samples)
        if updated:
            self._steps_since_update = 0
        return ray_samples, weights_list, ray_samples_list


class NearFarCollider(nn.Module):
    # fruit_nerf.py:161,382-383
    def __init__(self, near_plane, far_plane):
        super().__init__()
        self.near_plane = near_plane
        self.far_plane = far_plane

    def forward(self, ray_bundle: RayBundle) -> RayBundle:
        if ray_bundle.nears is not None and ray_bundle.fars is not None:
            return ray_bundle
        ones = torch.ones_like(ray_bundle.origins[..., 0:1])
        near_plane = self.near_plane if self.training else 0
        ray_bundle.nears = ones * near_plane
        ray_bundle.fars = ones * self.far_plane
        return ray_bundle


# --------------------------------------------------------------------------------------
# Renderers  (fruit_nerf.py:164-168, 287-306, 329-348)
# --------------------------------------------------------------------------------------


def render_rgb_last_sample(rgb: Tensor, weights: Tensor, training: bool) -> Tensor:
    if not training:
        rgb = torch.nan_to_num(rgb)
    comp_rgb = torch.sum(weights * rgb, dim=-2)
    accumulated_weight = torch.sum(weights, dim=-2)
    background_color = rgb[..., -1, :]
    comp_rgb = comp_rgb + background_color * (1.0 - accumulated_weight)
    if not training:
        comp_rgb = torch.clamp(comp_rgb, min=0.0, max=1.0)
    return comp_rgb


def render_accumulation(weights: Tensor) -> Tensor:
    return torch.sum(weights, dim=-2)


def render_depth_median(weights: Tensor, ray_samples: RaySamples) -> Tensor:
    steps = (ray_samples.frustums.starts + ray_samples.frustums.ends) / 2
    cumulative_weights = torch.cumsum(weights[..., 0], dim=-1)
    split = torch.ones((*weights.shape[:-2], 1), device=weights.device) * 0.5
    median_index = torch.searchsorted(cumulative_weights, split, side="left")
    median_index = torch.clamp(median_index, 0, steps.shape[-2] - 1)
    return torch.gather(steps[..., 0], dim=-1, index=median_index)


def render_semantics(semantics: Tensor, weights: Tensor) -> Tensor:
    return torch.sum(weights * semantics, dim=-2)


# --------------------------------------------------------------------------------------
# Losses  (fruit_nerf.py:25-30, 359-372, 400)
# --------------------------------------------------------------------------------------

EPS = 1.0e-7


def ray_samples_to_sdist(ray_samples: RaySamples) -> Tensor:
    starts = ray_samples.spacing_starts
    ends = ray_samples.spacing_ends
    return torch.cat([starts[..., 0], ends[..., -1:, 0]], dim=-1)


def outer(t0_starts, t0_ends, t1_starts, t1_ends, y1):
    cy1 = torch.cat([torch.zeros_like(y1[..., :1]), torch.cumsum(y1, dim=-1)], dim=-1)
    idx_lo = torch.searchsorted(t1_starts.contiguous(), t0_starts.contiguous(), side="right") - 1
    idx_lo = torch.clamp(idx_lo, min=0, max=y1.shape[-1] - 1)
    idx_hi = torch.searchsorted(t1_ends.contiguous(), t0_ends.contiguous(), side="right")
    idx_hi = torch.clamp(idx_hi, min=0, max=y1.shape[-1] - 1)
    cy1_lo = torch.take_along_dim(cy1[..., :-1], idx_lo, dim=-1)
    cy1_hi = torch.take_along_dim(cy1[..., 1:], idx_hi, dim=-1)
    return cy1_hi - cy1_lo


def lossfun_outer(t, w, t_env, w_env):
    w_outer = outer(t[..., :-1], t[..., 1:], t_env[..., :-1], t_env[..., 1:], w_env)
    return torch.clip(w - w_outer, min=0) ** 2 / (w + EPS)


def interlevel_loss(weights_list, ray_samples_list) -> Tensor:
    c = ray_samples_to_sdist(ray_samples_list[-1]).detach()
    w = weights_list[-1][..., 0].detach()
    loss_interlevel = 0.0
    for ray_samples, weights in zip(ray_samples_list[:-1], weights_list[:-1]):
        sdist = ray_samples_to_sdist(ray_samples)
        cp = sdist
        wp = weights[..., 0]
        loss_interlevel += torch.mean(lossfun_outer(c, w, cp, wp))
    return loss_interlevel


def lossfun_distortion(t, w):
    ut = (t[..., 1:] + t[..., :-1]) / 2
    dut = torch.abs(ut[..., :, None] - ut[..., None, :])
    loss_inter = torch.sum(w * torch.sum(w[..., None, :] * dut, dim=-1), dim=-1)
    loss_intra = torch.sum(w ** 2 * (t[..., 1:] - t[..., :-1]), dim=-1) / 3
    return loss_inter + loss_intra


def distortion_loss(weights_list, ray_samples_list) -> Tensor:
    c = ray_samples_to_sdist(ray_samples_list[-1])
    w = weights_list[-1][..., 0]
    return torch.mean(lossfun_distortion(c, w))
