"""TEST INFRASTRUCTURE (oracle) — PARITY UNPINNED for the nerfstudio half: the train datamanager's PixelSampler +
RayGenerator (nerfstudio 0.3.2 `data/pixel_samplers.py::PixelSampler.sample_method`,
`cameras/cameras.py::Cameras._generate_rays_from_coords` for PERSPECTIVE cameras), as driven by
/root/reference/fruit_nerf/data/fruit_datamanager.py:188-197 (`next_train`: image batch -> pixel sampler -> ray generator),
restated in plain torch for the CPU.  The product's counterpart is fnr_sample_pixels / fnr_train_prologue
(fruitnerf_amd/csrc/pixel_sampler.hip); only tests, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.

Pinned on closed-form pinhole geometry (tests/test_oracle_closed_form.py::test_pixel_rays_pinhole_geometry): the ray through
the principal point is the camera's -z axis, a pixel fx pixels to its right leaves at 45 degrees towards +x, image rows
grow DOWN (-y), rays are unit length and start at the camera centre.

Conventions (nerfstudio): pixel (x, y) is sampled at its centre (x + 0.5, y + 0.5); camera looks along -z, +x right,
+y up; `u` in [0, 1)^3 picks (image slot, row, column) by floor(u * extent), as `torch.rand(...) * [n, H, W]` floored
does in PixelSampler.sample_method."""
from typing import Dict, Tuple

import torch
from torch import Tensor


def pixel_rays(c2w: Tensor, cam_idx: Tensor, y: Tensor, x: Tensor, fx: float, fy: float, cx: float, cy: float
               ) -> Tuple[Tensor, Tensor]:
    """c2w [M,3,4], cam_idx / y / x [N] (integer pixel coordinates) -> origins [N,3], unit directions [N,3]."""
    px = (x.to(torch.float32) + 0.5 - cx) / fx
    py = -(y.to(torch.float32) + 0.5 - cy) / fy
    d_cam = torch.stack([px, py, -torch.ones_like(px)], dim=-1)                    # [N,3]
    rot = c2w[cam_idx, :, :3]                                                      # [N,3,3]
    d_world = (rot * d_cam[:, None, :]).sum(dim=-1)
    d_world = d_world / torch.linalg.norm(d_world, dim=-1, keepdim=True)
    return c2w[cam_idx, :, 3], d_world


def sample_pixels(data: Dict, image_ids: Tensor, u: Tensor):
    """data: {"images" uint8 [M,H,W,3], "masks" uint8 [M,H,W], "c2w" [M,3,4], "H", "W", "fx", "fy", "cx", "cy"};
    image_ids [n_train]: training slot -> dataset image; u [R,3] in [0,1).
    -> origins [R,3], directions [R,3], camera_indices [R,1] (the slot: the appearance-embedding row),
       {"image" [R,3] in [0,1], "fruit_mask" [R,1] in {0,1}}."""
    n = image_ids.numel()
    k = (u[:, 0] * n).long().clamp_max(n - 1)
    y = (u[:, 1] * data["H"]).long().clamp_max(data["H"] - 1)
    x = (u[:, 2] * data["W"]).long().clamp_max(data["W"] - 1)
    img = image_ids[k]
    o, d = pixel_rays(data["c2w"], img, y, x, data["fx"], data["fy"], data["cx"], data["cy"])
    image = data["images"][img, y, x].to(torch.float32) / 255.0
    mask = data["masks"][img, y, x].to(torch.float32)[:, None]
    return o, d, k[:, None], {"image": image, "fruit_mask": mask}
