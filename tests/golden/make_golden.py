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


def main():
    cfg = util.small_config(log2=10, prop_log2=8)
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
    tr = om(ns.RayBundle(o, d, pa, camera_indices=cam), jitter=jit)
    ld = om.get_loss_dict(tr, batch)
    sum(ld.values()).backward()
    for k in ("rgb", "semantics", "accumulation"):
        out["train::" + k] = tr[k].detach().numpy()
    for i in range(3):
        out[f"train::weights{i}"] = tr["weights_list"][i].detach().numpy()
    for k, v in ld.items():
        out["loss::" + k] = np.float32(v.item())
    for name, p in om.named_parameters():
        if "hash_table" in name:
            out["gradsum::" + name] = np.float64(p.grad.double().abs().sum().item())
        else:
            out["grad::" + name] = p.grad.numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fruit_nerf_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
