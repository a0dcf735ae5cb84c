"""nerfstudio.cameras.camera_optimizers (0.3.2) as used by the reference's datamanager
(/root/reference/fruit_nerf/fruit_nerf_config.py:39-43: CameraOptimizerConfig(mode="SO3xR3",
optimizer=AdamOptimizerConfig(lr=6e-4, eps=1e-8, weight_decay=1e-2), scheduler=ExponentialDecay(lr_final=6e-6,
max_steps=200000))): a learnable [num_cameras, 6] tangent vector per training camera, applied to the camera-to-world
matrices before ray generation and trained from the ray gradients of the hot path.  The arithmetic runs in
libfruitnerf_hip.so (camera_opt.hip); there is no CPU path."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
from torch import Tensor, nn

from .. import _kernels as K


@dataclass
class CameraOptimizerConfig:
    mode: str = "off"                      # "off" | "SO3xR3"  ("SE3" is not built)
    position_noise_std: float = 0.0        # nerfstudio fields kept for config compatibility; noise is not built
    orientation_noise_std: float = 0.0
    lr: float = 6e-4                       # AdamOptimizerConfig(lr=6e-4, eps=1e-8, weight_decay=1e-2)
    eps: float = 1e-8
    weight_decay: float = 1e-2
    lr_final: Optional[float] = 6e-6       # ExponentialDecaySchedulerConfig(lr_final=6e-6, max_steps=200000)
    max_steps: int = 200000
    param_group: str = "camera_opt"

    def setup(self, num_cameras: int, device) -> "CameraOptimizer":
        return CameraOptimizer(self, num_cameras, device)


class CameraOptimizer(nn.Module):
    """Layer that modifies camera poses to be optimized (nerfstudio CameraOptimizer)."""

    def __init__(self, config: CameraOptimizerConfig, num_cameras: int, device) -> None:
        super().__init__()
        if config.mode not in ("off", "SO3xR3"):
            raise NotImplementedError(f"camera optimizer mode {config.mode!r} is not built (off | SO3xR3)")
        if config.position_noise_std != 0.0 or config.orientation_noise_std != 0.0:
            raise NotImplementedError("pose noise is not built")
        self.config = config
        self.num_cameras = num_cameras
        self.device = torch.device(device)
        if config.mode == "SO3xR3":
            # padded to a multiple of 4 floats for the fused Adam kernel (540 floats for 90 cameras already are)
            self.pose_adjustment = nn.Parameter(torch.zeros(num_cameras, 6, device=self.device))
            self.pose_adjustment.grad = torch.zeros_like(self.pose_adjustment)
        self._identity = None

    @property
    def enabled(self) -> bool:
        return self.config.mode != "off"

    def forward(self, indices: Tensor) -> Tensor:
        """[N,3,4] camera-to-camera corrections exp_map_SO3xR3(pose_adjustment[indices]) (identity when off)."""
        n = indices.shape[0]
        if not self.enabled:
            return torch.eye(4, device=self.device)[None, :3, :4].tile(n, 1, 1)
        if self._identity is None:
            eye = torch.eye(4, device=self.device)[None, :3, :4].tile(self.num_cameras, 1, 1).contiguous()
            self._identity = K.ImageSetArg(torch.zeros(1, 1, 1, 3, dtype=torch.uint8, device=self.device),
                                           torch.zeros(1, 1, 1, dtype=torch.uint8, device=self.device), eye, 1, 1, 0, 0)
            self._all = torch.arange(self.num_cameras, device=self.device)
        return K.camera_adjust(self._identity, self._all, self.pose_adjustment.data)[indices.long()]

    def adjusted_cameras(self, image_set: "K.ImageSetArg", train_ids: Tensor) -> Optional[Tensor]:
        """c2w' [n_train,3,4] = pose_utils.multiply(c2w[train_ids], forward(arange(n_train))); None when off."""
        if not self.enabled:
            return None
        return K.camera_adjust(image_set, train_ids, self.pose_adjustment.data)

    def get_param_groups(self) -> dict:
        return {self.config.param_group: list(self.parameters())} if self.enabled else {}


class CameraAdam:
    """The camera optimiser's torch.optim.Adam(lr, eps, weight_decay) + ExponentialDecay schedule as one fused launch
    (same kernel as the model's optimiser: fnr_adam_step with the L2 weight_decay term)."""

    def __init__(self, camera_optimizer: CameraOptimizer, betas=(0.9, 0.999), algorithm: str = "adam"):
        # algorithm="radam": the big/huge configs' camera optimiser (RAdamOptimizerConfig(lr=6e-4, eps=1e-8,
        # weight_decay=1e-3), fruit_nerf_config.py:77-80,125-128)
        if algorithm not in ("adam", "radam"):
            raise ValueError(f"unknown optimiser algorithm {algorithm!r}")
        self.algorithm = algorithm
        self.opt = camera_optimizer
        self.cfg = camera_optimizer.config
        self.betas = betas
        self.step_count = 0
        p = camera_optimizer.pose_adjustment
        if p.numel() % 4 != 0:
            raise NotImplementedError("pose table size must be a multiple of 4 floats (num_cameras even)")
        self.exp_avg = torch.zeros_like(p.data)
        self.exp_avg_sq = torch.zeros_like(p.data)

    def fused_step_args(self, grad_scale: float = 1.0):
        """Advance the step count and return the fnr_table_adam of THIS update for a kernel that applies it itself
        (fnr_camera_pose_grad_adam) — same learning rate / step as step() would use."""
        from ..training import exponential_decay_lr
        from .. import _lib as L
        lr, step = self.advance()
        c = self.cfg
        p = self.opt.pose_adjustment
        return L.table_adam(0 if self.algorithm == "adam" else 1, lr, self.betas[0], self.betas[1], c.eps,
                            step, grad_scale, c.weight_decay, L.ptr(p.data), L.ptr(self.exp_avg),
                            L.ptr(self.exp_avg_sq), None, slot=L.ADAM_SLOTS["camera_opt"])

    def advance(self):
        """Count one optimiser step -> (its learning rate, its 1-based step number): the host bookkeeping of
        fused_step_args() (a replayed step program patches these two into the recorded call)."""
        from ..training import exponential_decay_lr
        self.step_count += 1
        c = self.cfg
        lr = c.lr if c.lr_final is None else exponential_decay_lr(self.step_count - 1, c.lr, c.lr_final, c.max_steps)
        return lr, self.step_count

    def step(self, grad_scale: float = 1.0) -> None:
        from ..training import exponential_decay_lr
        self.step_count += 1
        c = self.cfg
        lr = c.lr if c.lr_final is None else exponential_decay_lr(self.step_count - 1, c.lr, c.lr_final, c.max_steps)
        p = self.opt.pose_adjustment
        fn = K.adam_step if self.algorithm == "adam" else K.radam_step
        fn(p.data.view(-1), p.grad.view(-1), self.exp_avg.view(-1), self.exp_avg_sq.view(-1), lr,
           self.betas[0], self.betas[1], c.eps, self.step_count, grad_scale, True, weight_decay=c.weight_decay)
