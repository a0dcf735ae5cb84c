"""Export geometry of FruitDataManager — mirror of /root/reference/fruit_nerf/data/fruit_datamanager.py:
get_corners_of_aabb (:42-68), sample_surface_points (:71-121), setup_inference (:157-172),
next_sample_volume (:199-204).  These define the N_x x N_y x N lattice whose thresholded samples become
the exported point cloud, so they fix the exact point counts.

The lattice coordinates are evaluated with torch on the CPU (the reference's CPU path: torch.linspace's
ATen rounding) and uploaded once; the train/eval image pipeline of FruitDataManager (VanillaDataManager
subclass, :124-155,174-197,206-215) is data loading and out of scope.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from ..components.ray_generators import OrthographicRayGenerator


def get_corners_of_aabb(aabb, device="cpu"):
    min_coords = aabb[0]
    max_coords = aabb[1]
    corners = torch.asarray([
        [min_coords[0], min_coords[1], min_coords[2]],
        [max_coords[0], min_coords[1], min_coords[2]],
        [min_coords[0], max_coords[1], min_coords[2]],
        [max_coords[0], max_coords[1], min_coords[2]],
        [min_coords[0], min_coords[1], max_coords[2]],
        [max_coords[0], min_coords[1], max_coords[2]],
        [min_coords[0], max_coords[1], max_coords[2]],
        [max_coords[0], max_coords[1], max_coords[2]],
    ], device=device)
    return corners


def surface_lattice_axes(aabb, n):
    """The x / y coordinate vectors, constant z and plane vector of sample_surface_points (:71-121), CPU."""
    corner_1 = aabb[0]
    corner_2 = aabb[1]
    corner_3 = aabb[2]
    dx_y_z = torch.abs(torch.max(aabb, axis=0).values - torch.min(aabb, axis=0).values)
    constant_axis_part_pos = int(torch.argmax(torch.logical_and((corner_1 == corner_2), (corner_2 == corner_3)).to(int)))
    start_x_pos = torch.argmax(torch.abs(corner_1 - corner_2))
    x = torch.linspace(corner_1[start_x_pos], corner_2[start_x_pos],
                       int(dx_y_z[0] / dx_y_z[constant_axis_part_pos] * n), dtype=torch.float32)
    start_y_pos = torch.argmax(torch.abs(corner_1 - corner_3))
    y = torch.linspace(corner_1[start_y_pos], corner_3[start_y_pos],
                       int(dx_y_z[1] / dx_y_z[constant_axis_part_pos] * n), dtype=torch.float32)
    z0 = corner_3[constant_axis_part_pos]
    corner_4 = aabb[-1]
    plane_vector = torch.asarray([[0, 0, torch.sign(corner_4[constant_axis_part_pos]) * torch.abs(
        corner_1[constant_axis_part_pos]) + torch.abs(corner_4[constant_axis_part_pos])]], dtype=torch.float32)
    return x, y, z0, plane_vector


def sample_surface_points(aabb, n, device="cpu", noise=False) -> Tuple[torch.Tensor, torch.Tensor]:
    x, y, z0, plane_vector = surface_lattice_axes(aabb.cpu() if torch.is_tensor(aabb) else aabb, n)
    xx, yy = torch.meshgrid(x, y, indexing="ij")
    surface_points = torch.column_stack((xx.flatten(), yy.flatten(), torch.full_like(xx.flatten(), z0)))
    return surface_points.clone().to(device), plane_vector.to(device)


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
