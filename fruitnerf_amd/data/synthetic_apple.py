"""Procedural synthetic apple tree: the benchmark scene.

No dataset, checkpoint or transforms.json exists in the container, so the "synthetic apple 800x800" workload of
BASELINE.json is generated: K red fruit spheres + green foliage blobs + a brown trunk inside the unit cube,
rendered analytically (ray/sphere intersection, Lambert shading) to RGB images + binary fruit masks from M
pinhole cameras on the unit sphere looking at the origin (fx = fy = 1111, cx = cy = 400 at 800x800 — Blender's
40 degree FOV), following the conventions the reference dataparser produces after auto-scaling
(/root/reference/fruit_nerf/data/fruitnerf_dataparser.py:194-223: camera origins within the unit box, scene box
+-1) and the Nerfstudio ray convention (camera looks along -z, y up, pixel centres at +0.5).
Everything is seeded; pure torch so it runs on the HIP device (bench) or the CPU (tests, small sizes).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import torch
from torch import Tensor


@dataclass
class SyntheticScene:
    centers: Tensor   # [M,3]
    radii: Tensor     # [M]
    colors: Tensor    # [M,3]
    is_fruit: Tensor  # [M] bool
    n_fruits: int


def make_scene(n_fruits: int = 32, n_foliage: int = 300, seed: int = 0, device="cpu") -> SyntheticScene:
    rng = np.random.default_rng(seed)
    fruit_c = rng.uniform(-0.35, 0.35, size=(n_fruits, 3))
    fruit_r = rng.uniform(0.03, 0.05, size=n_fruits)
    fruit_col = np.stack([rng.uniform(0.75, 0.95, n_fruits), rng.uniform(0.05, 0.2, n_fruits),
                          rng.uniform(0.05, 0.15, n_fruits)], -1)
    # foliage: blobs in a crown of radius 0.4 around (0,0,0.05)
    v = rng.normal(size=(n_foliage, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    fol_c = v * (0.4 * rng.uniform(0.2, 1.0, size=(n_foliage, 1)) ** (1 / 3)) + np.array([0, 0, 0.05])
    fol_r = rng.uniform(0.04, 0.09, size=n_foliage)
    fol_col = np.stack([rng.uniform(0.05, 0.2, n_foliage), rng.uniform(0.35, 0.7, n_foliage),
                        rng.uniform(0.05, 0.2, n_foliage)], -1)
    # trunk: chain of brown spheres below the crown
    n_trunk = 12
    trunk_c = np.stack([np.zeros(n_trunk), np.zeros(n_trunk), np.linspace(-0.75, -0.2, n_trunk)], -1)
    trunk_r = np.full(n_trunk, 0.045)
    trunk_col = np.tile(np.array([[0.35, 0.22, 0.1]]), (n_trunk, 1))
    centers = np.concatenate([fruit_c, fol_c, trunk_c])
    radii = np.concatenate([fruit_r, fol_r, trunk_r])
    colors = np.concatenate([fruit_col, fol_col, trunk_col])
    is_fruit = np.zeros(len(radii), dtype=bool)
    is_fruit[:n_fruits] = True
    f = lambda a, dt=torch.float32: torch.as_tensor(a, dtype=dt, device=device)  # noqa: E731
    return SyntheticScene(f(centers), f(radii), f(colors), torch.as_tensor(is_fruit, device=device), n_fruits)


def make_cameras(n_cameras: int = 100, seed: int = 0, device="cpu") -> Tensor:
    """camera-to-world [M,3,4]: origins on the unit sphere (upper 3/4), looking at the origin, z-up world."""
    rng = np.random.default_rng(seed + 1)
    c2w = np.zeros((n_cameras, 3, 4), dtype=np.float64)
    for i in range(n_cameras):
        phi = 2 * np.pi * (i + rng.uniform(0, 1)) / n_cameras * 7.0
        z = rng.uniform(-0.3, 0.85)
        r = np.sqrt(max(1 - z * z, 0.0))
        pos = np.array([r * np.cos(phi), r * np.sin(phi), z])
        fwd = -pos / np.linalg.norm(pos)           # viewing direction
        up = np.array([0.0, 0.0, 1.0])
        right = np.cross(fwd, up)
        right /= np.linalg.norm(right)
        true_up = np.cross(right, fwd)
        c2w[i, :, 0], c2w[i, :, 1], c2w[i, :, 2], c2w[i, :, 3] = right, true_up, -fwd, pos
    return torch.as_tensor(c2w, dtype=torch.float32, device=device)


def pixel_rays(c2w: Tensor, cam_idx: Tensor, y: Tensor, x: Tensor, fx: float, fy: float, cx: float, cy: float
               ) -> Tuple[Tensor, Tensor]:
    """Nerfstudio RayGenerator convention for pinhole cameras: pixel centre (x+0.5, y+0.5), camera -z forward."""
    dirs_cam = torch.stack([(x.float() + 0.5 - cx) / fx, -(y.float() + 0.5 - cy) / fy, -torch.ones_like(x).float()], -1)
    R = c2w[cam_idx, :, :3]
    # (a batched-GEMM einsum here costs a 58 us hipBLASLt launch per step; three fused multiply-adds do not)
    d = R[:, :, 0] * dirs_cam[:, 0:1] + R[:, :, 1] * dirs_cam[:, 1:2] + R[:, :, 2] * dirs_cam[:, 2:3]
    d = torch.nn.functional.normalize(d, dim=-1)
    o = c2w[cam_idx, :, 3]
    return o, d


def shade_rays(scene: SyntheticScene, o: Tensor, d: Tensor, chunk: int = 1 << 18) -> Tuple[Tensor, Tensor]:
    """Analytic render: rgb [N,3] in [0,1] and fruit mask [N] (float 0/1)."""
    light = torch.nn.functional.normalize(torch.tensor([0.4, 0.3, 0.85], device=o.device), dim=0)
    rgbs, masks = [], []
    for s in range(0, o.shape[0], chunk):
        oo, dd = o[s:s + chunk], d[s:s + chunk]
        oc = oo[:, None, :] - scene.centers[None]                      # [n,M,3]
        b = (oc * dd[:, None, :]).sum(-1)
        c = (oc * oc).sum(-1) - scene.radii[None] ** 2
        disc = b * b - c
        t = -b - torch.sqrt(disc.clamp_min(0))
        t = torch.where((disc > 0) & (t > 1e-4), t, torch.full_like(t, float("inf")))
        tmin, idx = t.min(dim=1)
        hit = torch.isfinite(tmin)
        p = oo + dd * tmin.nan_to_num(posinf=0.0)[:, None]
        n = torch.nn.functional.normalize(p - scene.centers[idx], dim=-1)
        lam = 0.35 + 0.65 * (n * light).sum(-1).clamp_min(0)
        col = scene.colors[idx] * lam[:, None]
        sky = 0.85 + 0.1 * dd[:, 2:3].clamp(-1, 1)
        bg = torch.cat([sky * 0.95, sky * 0.97, sky], -1).clamp(0, 1)
        rgbs.append(torch.where(hit[:, None], col, bg))
        masks.append((hit & scene.is_fruit[idx]).float())
    return torch.cat(rgbs), torch.cat(masks)


def render_dataset(scene: SyntheticScene, c2w: Tensor, H: int = 800, W: int = 800, fx: float = 1111.0,
                   fy: float = 1111.0) -> Dict[str, Tensor]:
    """uint8 images [M,H,W,3] and masks [M,H,W] (the on-device 'image batch' a datamanager would hold)."""
    dev = c2w.device
    cx, cy = W / 2.0, H / 2.0
    ys, xs = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    ys, xs = ys.reshape(-1), xs.reshape(-1)
    imgs = torch.empty(c2w.shape[0], H, W, 3, dtype=torch.uint8, device=dev)
    msks = torch.empty(c2w.shape[0], H, W, dtype=torch.uint8, device=dev)
    for i in range(c2w.shape[0]):
        ci = torch.full_like(ys, i)
        o, d = pixel_rays(c2w, ci, ys, xs, fx, fy, cx, cy)
        rgb, m = shade_rays(scene, o, d)
        imgs[i] = (rgb.view(H, W, 3) * 255.0 + 0.5).clamp(0, 255).to(torch.uint8)
        msks[i] = m.view(H, W).to(torch.uint8)
    return {"images": imgs, "masks": msks, "H": H, "W": W, "fx": fx, "fy": fy, "cx": cx, "cy": cy, "c2w": c2w}


class PixelBatcher:
    """PixelSampler + RayGenerator of the train datamanager (data/fruit_datamanager.py:188-197) on the device:
    uniform random (image, y, x) triples -> RayBundle tensors + {"image", "fruit_mask"} batch."""

    def __init__(self, data: Dict[str, Tensor], image_ids: Tensor, seed: int):
        self.data = data
        self.image_ids = image_ids          # dataset indices used for training (camera_indices = position here)
        self.gen = torch.Generator(device=data["images"].device)
        self.gen.manual_seed(seed)
        self._set = None

    def sample(self, n_rays: int, camera_optimizer=None, level0: Optional[dict] = None):
        """One fused kernel on the HIP device (fnr_sample_pixels / fnr_train_prologue); a batch on the CPU raises.
        camera_optimizer (cameras.camera_optimizers.CameraOptimizer, mode SO3xR3): rays come from the pose-corrected
        cameras; `last_draw` keeps what training.camera_backward_and_step needs to back-propagate into the poses.
        level0 (FruitModel.level0_spec(): S, near, far, n_jitter): the whole start of the step in ONE launch
        (fnr_train_prologue: counter-based random numbers, camera adjust, pixel sampling and the proposal sampler's
        level-0 bins); `last_presample` then holds what to hand to the model as RayBundle.presampled."""
        d = self.data
        dev = d["images"].device
        if level0 is not None and dev.type == "cuda":
            from .. import _kernels as K
            if self._set is None:
                self._set = K.ImageSetArg(d["images"], d["masks"], d["c2w"], d["fx"], d["fy"], d["cx"], d["cy"])
            pose = camera_optimizer.pose_adjustment.data if (camera_optimizer is not None and camera_optimizer.enabled) else None
            self._offset = getattr(self, "_offset", 0) + 1
            out = K.train_prologue(self._set, self.image_ids, n_rays, self.gen.initial_seed(), self._offset, pose,
                                   level0["near"], level0["far"], level0["S"], n_jitter=level0.get("n_jitter", 3))
            self.last_draw = {"u": out["u"], "cam": out["cam"], "c2w_adjusted": out["c2w_adjusted"]}
            self.last_presample = {"S0": out["S0"], "near": out["near"], "far": out["far"], "spacing": out["spacing"],
                                   "euclid": out["euclid"], "jitter": [out["jitter"][i] for i in range(out["jitter"].shape[0])]}
            return out["origins"], out["directions"], out["cam"][:, None], \
                {"image": out["image"], "fruit_mask": out["mask"][:, None]}
        u = torch.rand(n_rays, 3, device=dev, generator=self.gen)
        if dev.type == "cuda":
            from .. import _kernels as K
            if self._set is None:
                self._set = K.ImageSetArg(d["images"], d["masks"], d["c2w"], d["fx"], d["fy"], d["cx"], d["cy"])
            c2w_adj = camera_optimizer.adjusted_cameras(self._set, self.image_ids) if camera_optimizer is not None else None
            o, dirs, cam, image, mask = K.sample_pixels(self._set, self.image_ids, u, c2w_adj)
            self.last_draw = {"u": u, "cam": cam, "c2w_adjusted": c2w_adj}
            self.last_presample = None
            return o, dirs, cam[:, None], {"image": image, "fruit_mask": mask[:, None]}
        raise RuntimeError(f"PixelBatcher.sample: the image batch is on {dev}; pixel sampling + ray generation run in "
                           "libfruitnerf_hip.so on a HIP device (no CPU path; the CPU restatement is oracle/pixel_sampler.py)")
