"""TEST INFRASTRUCTURE (oracle) — PARITY UNPINNED: nerfstudio 0.3.2 CameraOptimizer(mode="SO3xR3") restated from
memory (nerfstudio/cameras/camera_optimizers.py, cameras/lie_groups.py::exp_map_SO3xR3, utils/poses.py::multiply,
cameras/cameras.py::_generate_rays_from_coords), as configured at
/root/reference/fruit_nerf/fruit_nerf_config.py:39-43.  Plain torch on the CPU so that autograd provides the
reference gradient of the pose parameters.  Only tests import this."""
import torch
from torch import nn


def exp_map_SO3xR3(tangent_vector: torch.Tensor) -> torch.Tensor:
    """[N,6] (translation, so3 log-rotation) -> [N,3,4]; SO3 map 'grabbed from pytorch3d' with the 1e-4 clamp."""
    log_rot = tangent_vector[:, 3:]
    nrms = (log_rot * log_rot).sum(1)
    rot_angles = torch.clamp(nrms, 1e-4).sqrt()
    rot_angles_inv = 1.0 / rot_angles
    fac1 = rot_angles_inv * rot_angles.sin()
    fac2 = rot_angles_inv * rot_angles_inv * (1.0 - rot_angles.cos())
    skews = torch.zeros((log_rot.shape[0], 3, 3), dtype=log_rot.dtype)
    skews[:, 0, 1] = -log_rot[:, 2]
    skews[:, 0, 2] = log_rot[:, 1]
    skews[:, 1, 0] = log_rot[:, 2]
    skews[:, 1, 2] = -log_rot[:, 0]
    skews[:, 2, 0] = -log_rot[:, 1]
    skews[:, 2, 1] = log_rot[:, 0]
    skews_square = torch.bmm(skews, skews)
    ret = torch.zeros(tangent_vector.shape[0], 3, 4, dtype=tangent_vector.dtype)
    ret[:, :3, :3] = fac1[:, None, None] * skews + fac2[:, None, None] * skews_square + torch.eye(3)[None]
    ret[:, :3, 3] = tangent_vector[:, :3]
    return ret


def multiply(pose_a: torch.Tensor, pose_b: torch.Tensor) -> torch.Tensor:
    """utils/poses.py::multiply: compose [.,3,4] poses, R = R1 R2, t = t1 + R1 t2."""
    R1, t1 = pose_a[..., :3, :3], pose_a[..., :3, 3:]
    R2, t2 = pose_b[..., :3, :3], pose_b[..., :3, 3:]
    return torch.cat([R1.matmul(R2), t1 + R1.matmul(t2)], dim=-1)


class CameraOptimizer(nn.Module):
    def __init__(self, num_cameras: int, mode: str = "SO3xR3"):
        super().__init__()
        assert mode in ("off", "SO3xR3")
        self.mode = mode
        self.num_cameras = num_cameras
        if mode == "SO3xR3":
            self.pose_adjustment = nn.Parameter(torch.zeros((num_cameras, 6)))

    def forward(self, indices: torch.Tensor) -> torch.Tensor:
        if self.mode == "off":
            return torch.eye(4)[None, :3, :4].tile(indices.shape[0], 1, 1)
        return exp_map_SO3xR3(self.pose_adjustment[indices, :])


def generate_rays(c2w: torch.Tensor, camera_opt_to_camera: torch.Tensor, y: torch.Tensor, x: torch.Tensor, fx: float,
                  fy: float, cx: float, cy: float):
    """Cameras._generate_rays_from_coords for pinhole cameras: pixel centre +0.5, camera looks along -z,
    c2w = multiply(c2w, camera_opt_to_camera), directions = normalize(R d_cam), origins = t."""
    c2w = multiply(c2w, camera_opt_to_camera)
    d_cam = torch.stack([(x.float() + 0.5 - cx) / fx, -(y.float() + 0.5 - cy) / fy, -torch.ones_like(x).float()], -1)
    d = torch.sum(d_cam[..., None, :] * c2w[..., :3, :3], dim=-1)
    d = torch.nn.functional.normalize(d, dim=-1)
    return c2w[..., :3, 3], d
