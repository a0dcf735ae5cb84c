#!/usr/bin/env python3
"""Generates tests/golden/fruit_nerf_small.npz from the CPU oracle (seeded).  The reference ships no golden
vectors; these freeze the oracle's outputs so that (a) the oracle cannot drift silently and (b) the
HIP path can be checked on a GPU box without re-running the oracle.  Re-run only when the oracle is
deliberately changed:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ns_torch as ns  # noqa: E402
from tests import util  # noqa: E402


def big_config():
    """The parts of fruit_nerf_big that reach FruitField (fruit_nerf.py:88-103, fruit_nerf_config.py:82-95) on small
    tables: geo 30, semantic MLP 30->128->128->64, max_res 4096, anneal over 5000 steps."""
    cfg = util.small_config(log2=10, prop_log2=8, max_res=4096)
    cfg.geo_feat_dim, cfg.num_layers_semantic, cfg.hidden_dim_semantics = 30, 3, 128
    cfg.proposal_weights_anneal_max_num_iters = 5000
    return cfg


def main(cfg=None, file_name="fruit_nerf_small.npz", with_camera=True):
    cfg = util.small_config(log2=10, prop_log2=8) if cfg is None else cfg
    om = util.make_oracle(cfg, num_images=5, seed=123)
    R = 96
    o, d, pa, cam = util.random_rays(R, 5, seed=77)
    g = torch.Generator().manual_seed(9)
    jit = [torch.rand(R, 1, generator=g) for _ in range(3)]
    batch = {"image": torch.rand(R, 3, generator=g), "fruit_mask": (torch.rand(R, 1, generator=g) > 0.6).float()}
    out = {}
    for k, v in om.state_dict().items():
        out["sd::" + k] = v.numpy()
    out.update(origins=o.numpy(), directions=d.numpy(), cam=cam.numpy(), image=batch["image"].numpy(),
               fruit_mask=batch["fruit_mask"].numpy())
    for i, j in enumerate(jit):
        out[f"jitter{i}"] = j.numpy()
    # eval forward
    om.eval()
    with torch.no_grad():
        ev = om(ns.RayBundle(o, d, pa, camera_indices=cam))
    for k in ("rgb", "semantics", "accumulation", "depth"):
        out["eval::" + k] = ev[k].numpy()
    # train forward + losses + gradients (step 0: proposal nets updated)
    om.train()
    om.set_anneal(0)
    o_leaf, d_leaf = o.clone().requires_grad_(True), d.clone().requires_grad_(True)   # ray gradients (camera optimiser)
    tr = om(ns.RayBundle(o_leaf, d_leaf, pa, camera_indices=cam), jitter=jit)
    ld = om.get_loss_dict(tr, batch)
    sum(ld.values()).backward()
    out["grad::origins"], out["grad::directions"] = o_leaf.grad.numpy(), d_leaf.grad.numpy()
    for k in ("rgb", "semantics", "accumulation"):
        out["train::" + k] = tr[k].detach().numpy()
    for i in range(3):
        out[f"train::weights{i}"] = tr["weights_list"][i].detach().numpy()
    for k, v in ld.items():
        out["loss::" + k] = np.float32(v.item())
    for name, p in om.named_parameters():
        if p.grad is None:          # nerfstudio's zero-length device_indicator_param: part of the state dict, no gradient
            continue
        if "hash_table" in name:
            out["gradsum::" + name] = np.float64(p.grad.double().abs().sum().item())
        else:
            out["grad::" + name] = p.grad.numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), file_name)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    if with_camera:
        camera_golden()


def camera_golden():
    """CameraOptimizer(SO3xR3): corrected cameras, rays and the pose gradient of L = sum(o . Go + d . Gd)."""
    from oracle import camera_opt as oc
    from fruitnerf_amd.data import synthetic_apple as sa
    n_cam, HW, focal, R = 6, 32, 40.0, 128
    g = torch.Generator().manual_seed(31)
    c2w = sa.make_cameras(n_cam, seed=4)
    pose = torch.cat([torch.randn(n_cam, 3, generator=g) * 0.03, torch.randn(n_cam, 3, generator=g) * 0.05], dim=1)
    pose[1, 3:] = torch.tensor([2e-3, -1e-3, 3e-3])          # below the 1e-4 clamp of |w|^2
    u = torch.rand(R, 3, generator=g)
    Go, Gd = torch.randn(R, 3, generator=g), torch.randn(R, 3, generator=g)
    cam_opt = oc.CameraOptimizer(n_cam)
    with torch.no_grad():
        cam_opt.pose_adjustment.copy_(pose)
    k = (u[:, 0] * n_cam).long().clamp_max(n_cam - 1)
    y = (u[:, 1] * HW).long().clamp_max(HW - 1)
    x = (u[:, 2] * HW).long().clamp_max(HW - 1)
    o, d = oc.generate_rays(c2w[k], cam_opt(k), y, x, focal, focal, HW / 2.0, HW / 2.0)
    ((o * Go).sum() + (d * Gd).sum()).backward()
    with torch.no_grad():
        adj = oc.multiply(c2w, cam_opt(torch.arange(n_cam)))
    out = dict(c2w=c2w.numpy(), pose=pose.numpy(), u=u.numpy(), Go=Go.numpy(), Gd=Gd.numpy(), HW=np.int32(HW),
               focal=np.float32(focal), c2w_adjusted=adj.numpy(), origins=o.detach().numpy(), directions=d.detach().numpy(),
               pose_grad=cam_opt.pose_adjustment.grad.numpy())
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "camera_optimizer_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "big":      # python tests/golden/make_golden.py big
        main(big_config(), "fruit_nerf_big_small.npz", with_camera=False)
    else:
        main()
