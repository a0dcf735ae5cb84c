"""UniformSamplerWithNoise — mirror of /root/reference/fruit_nerf/components/ray_samplers.py:31-104.

Export-time sampler: `num_samples` uniform bins in [near, far] (spacing fn = identity); stratified jitter
only when `self.training`.  Bins are produced by fnr_sample_spaced.

Behavioural note (pinned by tests/test_reference_pins.py): the reference builds this sampler inside
`FruitModel.setup_inference` AFTER `eval_setup()` has put the pipeline in eval mode (scripts/exporter.py:86-94), so the
new module is still in training mode and the reference's export jitters every bin edge with `torch.rand`
(single_jitter=False -> t_rand [R, S+1]).  `FruitModel.setup_inference` here does the same by default (round 4); the
as-run export is reproduced exactly when the same jitter is supplied (`jitter_fn`;
tests/test_gpu_reference_pins.py::test_hip_export_matches_the_reference_export_as_run).
`setup_inference(..., deterministic=True)` puts the sampler in eval mode: bin centres, a deterministic lattice whose
exported point counts are reproducible (and which the fused lattice kernels implement).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from .. import _kernels as K
from ..rays import RayBundle, RaySamples


class UniformSamplerWithNoise(nn.Module):
    def __init__(self, num_samples: Optional[int] = None, train_stratified=True, single_jitter=False) -> None:
        super().__init__()
        self.num_samples = num_samples
        self.train_stratified = train_stratified
        self.single_jitter = single_jitter
        # source of the stratified jitter: callable(shape) -> tensor (any device); default torch.rand on the rays'
        # device.  `lambda shape: torch.rand(shape)` draws from the CPU generator exactly as the reference's CPU path
        # does (ray_samplers.py:80-83), which makes an export reproducible against it.
        self.jitter_fn = None

    def generate_ray_samples(self, ray_bundle: Optional[RayBundle] = None, num_samples: Optional[int] = None,
                             t_rand: Optional[torch.Tensor] = None) -> RaySamples:
        assert ray_bundle is not None
        assert ray_bundle.nears is not None
        assert ray_bundle.fars is not None
        num_samples = num_samples or self.num_samples
        assert num_samples is not None
        rays = K.RaysArg(ray_bundle.origins, ray_bundle.directions, ray_bundle.nears, ray_bundle.fars,
                         ray_bundle.camera_indices)
        if self.train_stratified and self.training:
            if t_rand is None:   # ray_samplers.py:80-83
                shape = (rays.n, 1) if self.single_jitter else (rays.n, num_samples + 1)
                t_rand = (torch.rand(shape, device=rays.device) if self.jitter_fn is None
                          else self.jitter_fn(shape).to(rays.device))
        else:
            t_rand = None
        spacing, euclid = K.sample_spaced(rays, 0, num_samples, t_rand)
        samples = ray_bundle.get_ray_samples(
            bin_starts=euclid[..., :-1, None], bin_ends=euclid[..., 1:, None],
            spacing_starts=spacing[..., :-1, None], spacing_ends=spacing[..., 1:, None])
        samples._structured = (rays, euclid, num_samples)
        return samples

    def forward(self, *args, **kwargs) -> RaySamples:
        return self.generate_ray_samples(*args, **kwargs)
