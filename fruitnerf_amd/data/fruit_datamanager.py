"""Export geometry of the datamanager: the N_x x N_y x N lattice whose thresholded samples become the exported point
cloud (so it fixes the exact point counts).  Written from the specification in SURVEY Appendix A.11; the behaviour it
reproduces lives in /root/reference/fruit_nerf/data/fruit_datamanager.py (`get_corners_of_aabb` :42-68,
`sample_surface_points` :71-121, `setup_inference` :157-172, `next_sample_volume` :199-204) and is pinned bit for bit
against those functions executed here (tests/test_reference_pins.py, tests/golden/reference_pins.npz).

Specification (A.11):
  * corner i of the box takes x from (min, max)[i & 1], y from (min, max)[(i >> 1) & 1], z from (min, max)[i >> 2];
  * the sampled face is the z = z_min face: counts n_x = int(dx / dz * n), n_y = int(dy / dz * n) evaluated on float32
    tensors (so a 0.6 : 1 box at n = 1000 gives int(0.6 * 1000) in float32), x = linspace(x_min, x_max, n_x),
    y = linspace(y_min, y_max, n_y), points in 'ij' meshgrid order (index = i_x * n_y + i_y), z = z_min;
  * the rays run along the plane vector (0, 0, sign(z_max) * |z_min| + |z_max|) — the box height only while
    z_min <= 0 <= z_max; kept as is (the reference's exported clouds depend on it).

The lattice coordinates are evaluated with torch on the CPU (torch.linspace's ATen rounding is part of the contract) and
uploaded once.  The train / eval image pipeline of FruitDataManager (a VanillaDataManager subclass) is data loading and
out of scope; the pixel sampling + ray generation of training live in data/synthetic_apple.py /
csrc/pixel_sampler.hip.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch
from torch import Tensor

from ..components.ray_generators import OrthographicRayGenerator


def get_corners_of_aabb(aabb, device="cpu") -> Tensor:
    """[8,3] corners of the box ((x_min, y_min, z_min), (x_max, y_max, z_max)): bit 0 of the corner index selects x,
    bit 1 y, bit 2 z (0 = min, 1 = max)."""
    lo_hi = [[aabb[side][axis] for axis in range(3)] for side in range(2)]
    return torch.asarray([[lo_hi[(i >> axis) & 1][axis] for axis in range(3)] for i in range(8)], device=device)


@dataclass
class SurfaceLattice:
    """The z_min face of a box as the export samples it: coordinate vectors, the face's z and the plane vector."""
    x: Tensor              # [n_x] float32
    y: Tensor              # [n_y] float32
    z0: Tensor             # 0-d
    plane_vector: Tensor   # [1,3] float32

    def __iter__(self):    # (x, y, z0, plane_vector)
        return iter((self.x, self.y, self.z0, self.plane_vector))


def surface_lattice_axes(corners, n: int) -> SurfaceLattice:
    """The face lattice of the box whose corners are `corners` ([8,3] in get_corners_of_aabb order), CPU tensors."""
    corners = torch.as_tensor(corners).cpu()
    lo, hi = corners[0], corners[-1]
    extent = (corners.max(dim=0).values - corners.min(dim=0).values).abs()
    # the face spanned by corners 0, 1, 2 is the one on which a coordinate stays fixed: z for this corner order; a box
    # that is flat in x or y has no such unique axis (the reference divides 0 by 0 there)
    if bool(extent[0] == 0) or bool(extent[1] == 0) or bool(extent[2] == 0):
        raise ValueError(f"degenerate export box: extents {extent.tolist()}")
    counts = [int(extent[axis] / extent[2] * n) for axis in (0, 1)]      # float32 arithmetic, truncation
    x = torch.linspace(lo[0], hi[0], counts[0], dtype=torch.float32)
    y = torch.linspace(lo[1], hi[1], counts[1], dtype=torch.float32)
    height = torch.sign(hi[2]) * torch.abs(lo[2]) + torch.abs(hi[2])
    return SurfaceLattice(x, y, lo[2], torch.asarray([[0, 0, height]], dtype=torch.float32))


def sample_surface_points(aabb, n, device="cpu", noise=False) -> Tuple[Tensor, Tensor]:
    """-> (points [n_x * n_y, 3] on the z_min face in 'ij' order, plane vector [1,3]); `aabb` = the 8 corners."""
    lat = surface_lattice_axes(aabb, n)
    gx, gy = torch.meshgrid(lat.x, lat.y, indexing="ij")
    flat_x = gx.reshape(-1)
    points = torch.stack((flat_x, gy.reshape(-1), torch.full_like(flat_x, lat.z0)), dim=1)
    return points.to(device), lat.plane_vector.to(device)


class ExportDataManager:
    """The slice of FruitDataManager the volume export drives (setup_inference / next_sample_volume)."""

    class _Config:
        eval_num_rays_per_batch = 4096

    def __init__(self, device, eval_num_rays_per_batch: int = 32768):
        self.device = torch.device(device)
        self.config = ExportDataManager._Config()
        self.config.eval_num_rays_per_batch = eval_num_rays_per_batch
        self.train_count = 0
        self.orthographic_ray_generator: Optional[OrthographicRayGenerator] = None
        self.export_lattice = None

    def setup_inference(self, aabb, num_points):
        corners = get_corners_of_aabb(aabb=aabb, device="cpu")
        x, y, z0, plane_vector = surface_lattice_axes(corners, num_points)
        surface_points, plane_vector = sample_surface_points(corners, n=num_points, device=self.device)
        self.orthographic_ray_generator = OrthographicRayGenerator(
            surface_points=surface_points, plane_normal=plane_vector,
            ray_batch_size=self.config.eval_num_rays_per_batch, device=self.device, aabb=aabb)
        # host-evaluated lattice for the fused export path: sample k of every ray sits at
        # z = z0 + dir_z * (t_k + t_{k+1}) / 2 with t = bins * far + (1 - bins) * near, near = 0
        # (components/ray_samplers.py:76-94 and Frustums.get_positions)
        pv = plane_vector.cpu()
        far = torch.linalg.norm(pv)
        direction = torch.nn.functional.normalize(pv)[0]
        bins = torch.linspace(0.0, 1.0, num_points + 1)
        t = bins * far + (1 - bins) * torch.zeros(())
        zs = z0.float() + direction[2] * (t[:-1] + t[1:]) / 2
        self.export_lattice = dict(xs=x.to(self.device), ys=y.to(self.device), zs=zs.to(self.device),
                                   direction=direction.to(self.device), far=float(far), n_samples=num_points)
        self.train_count = 0
        return surface_points.shape[0]

    def next_sample_volume(self, step: int):
        self.train_count += 1
        ray_bundle = self.orthographic_ray_generator(count=self.train_count)
        return ray_bundle, None
