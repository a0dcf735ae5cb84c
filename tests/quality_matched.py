"""Matched-quality run (BASELINE.md §2 quality gate): the CPU oracle and the HIP model trained on the SAME synthetic
scene from the SAME initial weights with the SAME rays and the SAME sampler jitter at every step (fruit_nerf: 4096
rays / step, SO3xR3 camera optimiser, Adam + schedules), evaluated on the same held-out pixels at the same steps.
The two trajectories separate after a few hundred steps (float summation order; the training is chaotic), so the
claim this supports is statistical: held-out PSNR / semantic IoU of the HIP path vs the oracle at equal step counts.

    python -m tests.quality_matched --side oracle --out profiles/r02_raw/quality_oracle.json     # hours of CPU
    python -m tests.quality_matched --side hip    --out gpurun_out/quality_hip.json              # seconds of GPU

Test infrastructure (imports oracle/); not collected by pytest."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_CAMERAS, TRAIN_SPLIT, HW, R = 100, 0.9, 800, 4096
EVAL_AT = (250, 500, 1000, 1736)
EVAL_PIXELS = 32768          # per held-out view, 5 views


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", choices=["oracle", "hip"], required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--steps", type=int, default=EVAL_AT[-1])
    ap.add_argument("--threads", type=int, default=0)
    args = ap.parse_args()
    from bench import METHODS, split_indices
    M = METHODS["fruit_nerf"]
    from fruitnerf_amd.data import synthetic_apple as sa
    from oracle import camera_opt as oc
    from oracle import fruit_oracle as fo
    from oracle import ns_torch as ns
    from fruitnerf_amd.hostinfo import usable_cpus
    torch.set_num_threads(args.threads or usable_cpus())
    hip = args.side == "hip"
    dev = torch.device("cuda:0") if hip else torch.device("cpu")
    focal = 1111.0 * HW / 800.0
    scene = sa.make_scene(seed=0, device=dev)
    c2w = sa.make_cameras(N_CAMERAS, seed=0, device=dev)
    data = sa.render_dataset(scene, c2w, H=HW, W=HW, fx=focal, fy=focal)
    i_train, i_eval = split_indices(N_CAMERAS, TRAIN_SPLIT)
    train_ids = torch.as_tensor(i_train, device=dev)
    n_train = len(i_train)
    torch.manual_seed(0)
    om = fo.FruitModel(fo.FruitNerfModelConfig(), num_train_data=n_train)       # identical initial weights
    g = torch.Generator().manual_seed(2024)                                     # rays + jitter stream (CPU generator)
    ge = torch.Generator().manual_seed(7)                                       # held-out pixels
    eval_px = [(int(img), torch.randint(0, HW, (EVAL_PIXELS,), generator=ge), torch.randint(0, HW, (EVAL_PIXELS,), generator=ge))
               for img in i_eval[:5]]
    results = {"side": args.side, "rays_per_step": R, "evals": []}

    if hip:
        from tests import util
        from fruitnerf_amd import _kernels as K
        from fruitnerf_amd.cameras.camera_optimizers import CameraAdam, CameraOptimizerConfig
        from fruitnerf_amd.rays import RayBundle
        from fruitnerf_amd.training import FusedAdam, fused_train_iteration
        model = util.make_hip_like(om, dev)
        model.train()
        opt = FusedAdam(model, group_lr={k: dict(v) for k, v in M["groups"].items()})
        cm = M["camera"]
        cam_opt = CameraOptimizerConfig(mode="SO3xR3", lr=cm["lr"], eps=cm["eps"], weight_decay=cm["weight_decay"],
                                        lr_final=cm["lr_final"], max_steps=cm["max_steps"]).setup(n_train, dev)
        cadam = CameraAdam(cam_opt, algorithm=cm["algorithm"])
        batcher = sa.PixelBatcher(data, train_ids, seed=0)
        batcher._set = K.ImageSetArg(data["images"], data["masks"], data["c2w"], data["fx"], data["fy"], data["cx"], data["cy"])
    else:
        om.train()
        groups = om.get_param_groups()
        ocam = oc.CameraOptimizer(n_train)
        cm = M["camera"]
        hyper = [M["groups"]["proposal_networks"], M["groups"]["fields"], cm]
        opts = [torch.optim.Adam(groups["proposal_networks"], lr=hyper[0]["lr"], eps=1e-15),
                torch.optim.Adam(groups["fields"], lr=hyper[1]["lr"], eps=1e-15),
                torch.optim.Adam(ocam.parameters(), lr=cm["lr"], eps=cm["eps"], weight_decay=cm["weight_decay"])]

        def decay(h):   # nerfstudio ExponentialDecay without warm-up (fruit_nerf_config.py:47-56)
            return lambda s: float(np.exp(np.log(h["lr_final"] / h["lr"]) * min(s / h["max_steps"], 1.0)))
        scheds = [torch.optim.lr_scheduler.LambdaLR(o_, decay(h)) for o_, h in zip(opts, hyper) if h.get("lr_final")]
        cb = sa.PixelBatcher(data, train_ids, seed=0)

    def evaluate(step):
        psnrs, inter, union = [], 0.0, 0.0
        with torch.no_grad():
            for img, y, x in eval_px:
                ci = torch.full((EVAL_PIXELS,), img)
                o, d = sa.pixel_rays(c2w, ci.to(dev), y.to(dev), x.to(dev), focal, focal, HW / 2.0, HW / 2.0)
                tgt = data["images"][ci.to(dev), y.to(dev), x.to(dev)].float() / 255.0
                msk = data["masks"][ci.to(dev), y.to(dev), x.to(dev)].float()
                if hip:
                    model.eval()
                    out = model(RayBundle(o, d, None, None))
                    model.train()
                else:
                    om.eval()
                    out = om(ns.RayBundle(o, d, torch.ones(EVAL_PIXELS, 1), camera_indices=torch.zeros(EVAL_PIXELS, 1, dtype=torch.long)))
                    om.train()
                mse = torch.mean((out["rgb"] - tgt) ** 2)
                psnrs.append(float(-10.0 * torch.log10(mse)))
                pred = (torch.sigmoid(out["semantics"][:, 0]) > 0.5).float()
                inter += float((pred * msk).sum())
                union += float(((pred + msk) > 0).float().sum())
        rec = {"step": step, "psnr_heldout": round(float(np.mean(psnrs)), 3),
               "semantic_iou_heldout": round(inter / max(union, 1.0), 4)}
        results["evals"].append(rec)
        print(json.dumps(rec), flush=True)
        with open(args.out, "w") as f:
            json.dump(results, f, indent=1)

    t0 = time.time()
    for step in range(args.steps):
        u = torch.rand(R, 3, generator=g)
        jit = [torch.rand(R, 1, generator=g) for _ in range(3)]
        if hip:
            ud = u.to(dev)
            c2w_adj = cam_opt.adjusted_cameras(batcher._set, batcher.image_ids)
            o, d, ci, image, mask = K.sample_pixels(batcher._set, batcher.image_ids, ud, c2w_adj)
            batcher.last_draw = {"u": ud, "cam": ci, "c2w_adjusted": c2w_adj}
            ld, _ = fused_train_iteration(model, opt, RayBundle(o, d, None, ci[:, None]),
                                          {"image": image, "fruit_mask": mask[:, None]}, step,
                                          jitter=[j.to(dev) for j in jit], camera=(cam_opt, cadam, batcher))
        else:
            o, d, cam, batch = cb.sample_torch(u)
            kk = cam[:, 0]
            yy = (u[:, 1] * HW).long().clamp_max(HW - 1)
            xx = (u[:, 2] * HW).long().clamp_max(HW - 1)
            o, d = oc.generate_rays(data["c2w"][train_ids[kk]], ocam(kk), yy, xx, focal, focal, HW / 2.0, HW / 2.0)
            om.set_anneal(step)
            for op_ in opts:
                op_.zero_grad()
            out = om(ns.RayBundle(o, d, torch.ones(R, 1), camera_indices=cam), jitter=jit)
            ld = om.get_loss_dict(out, batch)
            sum(ld.values()).backward()
            for op_ in opts:
                op_.step()
            for sc in scheds:
                sc.step()
            om.proposal_sampler.step_cb(step)
        if (step + 1) % 50 == 0:
            print(f"step {step + 1} {time.time() - t0:.0f}s " + " ".join(f"{k}={float(v):.5f}" for k, v in ld.items()), flush=True)
        if step + 1 in EVAL_AT:
            evaluate(step + 1)
    results["train_seconds"] = round(time.time() - t0, 1)
    with open(args.out, "w") as f:
        json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
