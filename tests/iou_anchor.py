"""Anchor for the late semantic-IoU drop of bench.py's quality run (IoU 0.89 at 10 k steps, 0.75 at
30 k on the bench's seed stream): is the drift after 10 k steps the METHOD's behaviour on this scene or the HIP path's
(2-piece backward, Philox stream)?

The HIP model is trained with bench.py's own loop to `--pre-steps` (same seeds: the state bench.py's quality gate passes
through), then the COMPLETE training state — parameters, both Adam moments and step counts of both parameter groups, the
camera poses with their optimiser state, the proposal sampler's schedule counters — is copied into the CPU oracle
(torch.optim.Adam + LambdaLR at the same step), and both sides continue for `--steps` iterations on IDENTICAL rays and
sampler jitter (one CPU generator stream feeds both), in bf16x3 and in strict fp32 arithmetic on the HIP side.  Held-out
PSNR / semantic IoU (bench.py's definition) of all arms at the same steps, plus each arm's batch losses.

    python -m tests.iou_anchor --pre-steps 10000 --steps 200 --eval-every 50 --out gpurun_out/r04/iou_anchor.json

Test infrastructure (imports oracle/); not collected by pytest."""
import argparse
import copy
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pre-steps", type=int, default=10000)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--eval-every", type=int, default=50)
    ap.add_argument("--eval-pixels", type=int, default=16384, help="held-out pixels per view (5 views)")
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--no-oracle", action="store_true", help="HIP arms only (seconds)")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    import bench
    from fruitnerf_amd import _kernels as K
    from fruitnerf_amd.cameras.camera_optimizers import CameraAdam
    from fruitnerf_amd.data import synthetic_apple as sa
    from fruitnerf_amd.hostinfo import usable_cpus
    from fruitnerf_amd.rays import RayBundle
    from fruitnerf_amd.training import FusedAdam, fused_train_iteration
    from oracle import camera_opt as oc
    from oracle import fruit_oracle as fo
    from oracle import ns_torch as ns
    torch.set_num_threads(usable_cpus())
    dev = torch.device("cuda:0")
    HW, focal, R = 800, 1111.0, args.rays
    scene = sa.make_scene(seed=0, device=dev)
    c2w = sa.make_cameras(bench.N_CAMERAS, seed=0, device=dev)
    data = sa.render_dataset(scene, c2w, H=HW, W=HW, fx=focal, fy=focal)
    i_train, i_eval = bench.split_indices(bench.N_CAMERAS, bench.TRAIN_SPLIT)
    train_ids = torch.as_tensor(i_train, device=dev)
    n_train = len(i_train)
    M = bench.METHODS["fruit_nerf"]

    # ---- bench.py's own loop to the checkpoint ---------------------------------------------------------------------------
    t0 = time.time()
    run = bench.MethodRun("fruit_nerf", "bf16x3", "SO3xR3", dev, 0, 1, data, train_ids, n_train)
    for _ in range(args.pre_steps):
        run.one_step(want_metrics=False)
    torch.cuda.synchronize()
    print(f"pre-training: {args.pre_steps} steps in {time.time() - t0:.1f} s", flush=True)
    model, opt = run.model, run.opt
    cam_opt, cadam = run.camera[0], run.camera[1]
    run.steps.drop_lookahead()
    start = run.step_idx
    sampler = model.proposal_sampler
    snap = {"params": model.arena().params.clone(), "m": opt.exp_avg.clone(), "v": opt.exp_avg_sq.clone(),
            "step_count": opt.step_count, "group_steps": dict(opt.group_steps),
            "pose": cam_opt.pose_adjustment.data.clone(), "cm": cadam.exp_avg.clone(), "cv": cadam.exp_avg_sq.clone(),
            "cstep": cadam.step_count, "sampler": (sampler._step, sampler._steps_since_update)}

    ge = torch.Generator().manual_seed(7)
    eval_px = [(int(img), torch.randint(0, HW, (args.eval_pixels,), generator=ge),
                torch.randint(0, HW, (args.eval_pixels,), generator=ge)) for img in i_eval[:5]]
    c2w_cpu, images_cpu, masks_cpu = c2w.cpu(), data["images"].cpu(), data["masks"].cpu()

    def heldout(forward, device):
        psnrs, inter, union = [], 0.0, 0.0
        with torch.no_grad():
            for img, y, x in eval_px:
                ci = torch.full((args.eval_pixels,), img)
                o, d = sa.pixel_rays(c2w_cpu, ci, y, x, focal, focal, HW / 2.0, HW / 2.0)
                tgt = images_cpu[ci, y, x].float() / 255.0
                msk = masks_cpu[ci, y, x].float()
                out = forward(o.to(device), d.to(device))
                rgb, sem = out["rgb"].cpu(), out["semantics"][:, 0].cpu()
                psnrs.append(float(-10.0 * torch.log10(torch.mean((rgb - tgt) ** 2))))
                pred = (torch.sigmoid(sem) > 0.5).float()
                inter += float((pred * msk).sum())
                union += float(((pred + msk) > 0).float().sum())
        return round(float(np.mean(psnrs)), 3), round(inter / max(union, 1.0), 4)

    def stream():
        """The shared ray / jitter stream: (u [R,3], jitter [3 x [R,1]]) per step from one CPU generator."""
        g = torch.Generator().manual_seed(4242)
        for _ in range(args.steps):
            yield torch.rand(R, 3, generator=g), [torch.rand(R, 1, generator=g) for _ in range(3)]

    results = {"pre_steps": start, "steps": args.steps, "rays_per_step": R, "arms": {}}

    # ---- HIP arms: restore the snapshot, continue on the shared stream -------------------------------------------------
    def hip_arm(precision):
        with torch.no_grad():
            model.arena().params.copy_(snap["params"])
            opt.exp_avg.copy_(snap["m"])
            opt.exp_avg_sq.copy_(snap["v"])
            model.arena().grads.zero_()
            cam_opt.pose_adjustment.data.copy_(snap["pose"])
            cam_opt.pose_adjustment.grad.zero_()
            cadam.exp_avg.copy_(snap["cm"])
            cadam.exp_avg_sq.copy_(snap["cv"])
        opt.step_count, opt.group_steps = snap["step_count"], dict(snap["group_steps"])
        cadam.step_count = snap["cstep"]
        sampler._step, sampler._steps_since_update = snap["sampler"]
        model.field.mlp_precision = precision
        model.train()
        batcher = run.batcher
        traj = []

        def ev(step):
            model.eval()
            p, i = heldout(lambda o, d: model(RayBundle(o, d, None, None)), dev)
            model.train()
            traj.append({"step": step, "psnr_heldout": p, "semantic_iou_heldout": i})
            print(precision, json.dumps(traj[-1]), flush=True)
        ev(start)
        losses = []
        for k, (u, jit) in enumerate(stream()):
            ud = u.to(dev)
            c2w_adj = cam_opt.adjusted_cameras(batcher._set, batcher.image_ids)
            o, d, ci, image, mask = K.sample_pixels(batcher._set, batcher.image_ids, ud, c2w_adj)
            batcher.last_draw = {"u": ud, "cam": ci, "c2w_adjusted": c2w_adj}
            ld, _ = fused_train_iteration(model, opt, RayBundle(o, d, None, ci[:, None]),
                                          {"image": image, "fruit_mask": mask[:, None]}, start + k,
                                          jitter=[j.to(dev) for j in jit], camera=(cam_opt, cadam, batcher))
            if (k + 1) % 10 == 0:
                losses.append([start + k + 1] + [round(float(v), 7) for v in ld.values()])
            if (k + 1) % args.eval_every == 0:
                ev(start + k + 1)
        results["arms"][f"hip_{precision}"] = {"trajectory": traj, "batch_losses_every_10": losses}

    hip_arm("bf16x3")
    hip_arm("fp32")

    # ---- oracle arm: the same state in torch.optim on the CPU -----------------------------------------------------------
    if not args.no_oracle:
        ocfg = fo.FruitNerfModelConfig()
        om = fo.FruitModel(ocfg, num_train_data=n_train)
        with torch.no_grad():
            model.arena().params.copy_(snap["params"])
        om.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()}, strict=True)
        om.train()
        om.proposal_sampler._step, om.proposal_sampler._steps_since_update = snap["sampler"]
        groups = om.get_param_groups()
        ocam = oc.CameraOptimizer(n_train)
        with torch.no_grad():
            ocam.pose_adjustment.copy_(snap["pose"].cpu())
        cm = M["camera"]
        hyper = [M["groups"]["proposal_networks"], M["groups"]["fields"], cm]
        opts = [torch.optim.Adam(groups["proposal_networks"], lr=hyper[0]["lr"], eps=1e-15),
                torch.optim.Adam(groups["fields"], lr=hyper[1]["lr"], eps=1e-15),
                torch.optim.Adam(ocam.parameters(), lr=cm["lr"], eps=cm["eps"], weight_decay=cm["weight_decay"])]
        # Adam state: the arena's moment slices per parameter (the arena lists the groups' parameters in the order of
        # get_param_groups / named_parameters, which the oracle shares), step counts per group
        named_h = dict(model.named_parameters())
        arena = model.arena()
        offsets = {id(p): (off, n) for _, p, off, n in arena.entries}
        m_cpu, v_cpu = snap["m"].cpu(), snap["v"].cpu()
        for (gname, o_), steps_taken in zip((("proposal_networks", opts[0]), ("fields", opts[1])),
                                            (snap["group_steps"]["proposal_networks"], snap["group_steps"]["fields"])):
            prefix = "proposal_networks." if gname == "proposal_networks" else "field."
            by_name = {n: p for n, p in om.named_parameters() if n.startswith(prefix)}
            for n, p in by_name.items():
                off, cnt = offsets[id(named_h[n])]
                o_.state[p] = {"step": torch.tensor(float(steps_taken)), "exp_avg": m_cpu[off:off + cnt].view(p.shape).clone(),
                               "exp_avg_sq": v_cpu[off:off + cnt].view(p.shape).clone()}
        opts[2].state[ocam.pose_adjustment] = {"step": torch.tensor(float(snap["cstep"])), "exp_avg": snap["cm"].cpu().clone(),
                                               "exp_avg_sq": snap["cv"].cpu().clone()}

        def decay(h):
            return lambda s: float(np.exp(np.log(h["lr_final"] / h["lr"]) * min(s / h["max_steps"], 1.0)))
        scheds = []
        for o_, h in zip(opts, hyper):
            if h.get("lr_final"):
                sc = torch.optim.lr_scheduler.LambdaLR(o_, decay(h))
                sc.last_epoch = start
                for g_, lr in zip(o_.param_groups, [h["lr"] * decay(h)(start)]):
                    g_["lr"] = lr
                scheds.append(sc)
        data_cpu = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in data.items()}
        from oracle import pixel_sampler as ops
        cb = sa.PixelBatcher(data_cpu, train_ids.cpu(), seed=0)     # (holds data / image_ids for the oracle's sampler)
        tids = train_ids.cpu()
        traj, losses = [], []

        def evo(step):
            om.eval()
            p, i = heldout(lambda o, d: om(ns.RayBundle(o, d, torch.ones(len(o), 1),
                                                         camera_indices=torch.zeros(len(o), 1, dtype=torch.long))), "cpu")
            om.train()
            traj.append({"step": step, "psnr_heldout": p, "semantic_iou_heldout": i})
            print("oracle", json.dumps(traj[-1]), flush=True)
        evo(start)
        t1 = time.time()
        for k, (u, jit) in enumerate(stream()):
            o, d, cam, batch = ops.sample_pixels(cb.data, cb.image_ids, u)
            kk = cam[:, 0]
            yy = (u[:, 1] * HW).long().clamp_max(HW - 1)
            xx = (u[:, 2] * HW).long().clamp_max(HW - 1)
            o, d = oc.generate_rays(data_cpu["c2w"][tids[kk]], ocam(kk), yy, xx, focal, focal, HW / 2.0, HW / 2.0)
            om.set_anneal(start + k)
            for op_ in opts:
                op_.zero_grad()
            out = om(ns.RayBundle(o, d, torch.ones(R, 1), camera_indices=cam), jitter=jit)
            ld = om.get_loss_dict(out, batch)
            sum(ld.values()).backward()
            for op_ in opts:
                op_.step()
            for sc in scheds:
                sc.step()
            om.proposal_sampler.step_cb(start + k)
            if (k + 1) % 10 == 0:
                losses.append([start + k + 1] + [round(float(v), 7) for v in ld.values()])
                print(f"oracle step {k + 1} {time.time() - t1:.0f}s", flush=True)
            if (k + 1) % args.eval_every == 0:
                evo(start + k + 1)
        results["arms"]["oracle_fp32_cpu"] = {"trajectory": traj, "batch_losses_every_10": losses,
                                              "train_seconds": round(time.time() - t1, 1)}
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(results, f, indent=1)
    print(json.dumps({k: v["trajectory"] for k, v in results["arms"].items()}))


if __name__ == "__main__":
    main()
