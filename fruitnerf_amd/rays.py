"""Ray containers exchanged across the FruitModel / FruitField API.

The reference takes these from nerfstudio (`nerfstudio.cameras.rays`, imported at
/root/reference/fruit_nerf/fruit_nerf.py:19 and components/ray_samplers.py:27); nerfstudio is not
installable here, so these are minimal stand-ins with IDENTICAL field names — the hot path only reads
attributes, so real nerfstudio objects can be passed instead (duck typing).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional

import torch
from torch import Tensor


_zero_cols = {}


@dataclass
class Frustums:
    origins: Tensor  # [..., 3]
    directions: Tensor  # [..., 3]
    starts: Tensor  # [..., 1]
    ends: Tensor  # [..., 1]
    pixel_area: Tensor  # [..., 1]

    def get_positions(self) -> Tensor:
        return self.origins + self.directions * (self.starts + self.ends) / 2

    @property
    def shape(self):
        return self.origins.shape[:-1]


@dataclass
class RaySamples:
    frustums: Frustums
    camera_indices: Optional[Tensor] = None  # [..., 1]
    deltas: Optional[Tensor] = None  # [..., 1]
    spacing_starts: Optional[Tensor] = None  # [..., S, 1]
    spacing_ends: Optional[Tensor] = None
    spacing_to_euclidean_fn: Optional[Callable] = None
    # structured fast-path handle: (ray_bundle, euclid_bins [R,S+1], spacing_bins [R,S+1]) when the samples
    # were produced by our own samplers; lets FruitField skip the per-sample flattening.
    _structured: Optional[tuple] = None

    @property
    def shape(self):
        return self.frustums.shape


@dataclass
class RayBundle:
    origins: Tensor  # [R, 3]
    directions: Tensor  # [R, 3]
    pixel_area: Optional[Tensor] = None  # [R, 1]
    camera_indices: Optional[Tensor] = None  # [R, 1]
    nears: Optional[Tensor] = None  # [R, 1]
    fars: Optional[Tensor] = None  # [R, 1]
    # Not a nerfstudio field: what fnr_train_prologue already produced for these rays in the launch that drew them
    # (level-0 bins of the proposal sampler and the samplers' jitters; data/synthetic_apple.py::PixelBatcher.sample with
    # level0=).  FruitModel uses it when it matches its sampler configuration and ignores it otherwise.
    presampled: Optional[dict] = None

    def __len__(self) -> int:
        return self.origins.shape[:-1].numel()

    def get_row_major_sliced_ray_bundle(self, start_idx: int, end_idx: int) -> "RayBundle":
        def sl(t):
            return None if t is None else t.reshape(-1, t.shape[-1])[start_idx:end_idx]

        return RayBundle(sl(self.origins), sl(self.directions), sl(self.pixel_area), sl(self.camera_indices),
                         sl(self.nears), sl(self.fars))

    def get_ray_samples(self, bin_starts: Tensor, bin_ends: Tensor, spacing_starts=None, spacing_ends=None,
                        spacing_to_euclidean_fn=None) -> RaySamples:
        S = bin_starts.shape[-2]
        exp = lambda t: None if t is None else t[:, None, :].expand(-1, S, -1)  # noqa: E731
        pixel_area = self.pixel_area
        if pixel_area is None:  # not read by the hot path; one cached zero column instead of a fill per call
            key = (self.origins.shape[0], str(self.origins.device))
            pixel_area = _zero_cols.get(key)
            if pixel_area is None:
                if len(_zero_cols) > 16:
                    _zero_cols.clear()
                pixel_area = _zero_cols[key] = torch.zeros_like(self.origins[:, :1])
        frustums = Frustums(exp(self.origins), exp(self.directions), bin_starts, bin_ends, exp(pixel_area))
        return RaySamples(frustums, exp(self.camera_indices), bin_ends - bin_starts, spacing_starts, spacing_ends,
                          spacing_to_euclidean_fn)
