"""OrthographicRayGenerator: batches of parallel rays off the export lattice.

Written from SURVEY Appendix A.11 (the behaviour is the reference's
/root/reference/fruit_nerf/components/ray_generators.py:24-66, pinned in tests/test_reference_pins.py): batch `count`
(1-based) holds lattice rows [(count - 1) * B, min(count * B, P)); every ray starts at its lattice point, runs along the
normalised plane vector from near = 0 to far = |plane vector| and has zero pixel area.
"""
from __future__ import annotations

import torch
from torch import nn

from ..rays import RayBundle


class OrthographicRayGenerator(nn.Module):
    def __init__(self, surface_points, plane_normal, ray_batch_size, device, aabb) -> None:
        super().__init__()
        self.device = device
        self.aabb = aabb
        self.ray_batch_size = ray_batch_size
        self.surface_points = surface_points
        length = torch.linalg.norm(plane_normal)
        self.surface_vector_norm = length.to(device)                                    # far plane of every ray
        self.surface_normal = torch.nn.functional.normalize(plane_normal).to(device)    # [1,3] common direction

    def batch_range(self, count: int):
        """Lattice rows of the `count`-th batch (count >= 1), clipped to the lattice."""
        total = self.surface_points.shape[0]
        first = (count - 1) * self.ray_batch_size
        return first, min(first + self.ray_batch_size, total)

    def forward(self, count) -> RayBundle:
        first, last = self.batch_range(count)
        origins = self.surface_points[first:last]
        rows = origins.shape[0]
        column = torch.zeros(rows, 1).to(self.device)
        return RayBundle(origins=origins, directions=self.surface_normal.repeat(rows, 1).to(self.device),
                         pixel_area=column, nears=column.clone(), fars=(column + 1) * self.surface_vector_norm)
