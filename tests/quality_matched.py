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
    ap.add_argument("--method", default="fruit_nerf", help="bench.py METHODS key (fruit_nerf | fruit_nerf_big | ...)")
    ap.add_argument("--rays", type=int, default=R, help="rays per step (a bounded batch keeps the oracle side affordable)")
    ap.add_argument("--eval-at", default=",".join(map(str, EVAL_AT)))
    ap.add_argument("--data-cache", default="", help="torch.save / load the rendered dataset here (the analytic renderer "
                    "takes ~8 min for 100 x 800 x 800 on 8 host cores, seconds on the GPU)")
    ap.add_argument("--eval-pixels", type=int, default=EVAL_PIXELS, help="held-out pixels per view (5 views)")
    ap.add_argument("--save-state", default="", help="write the final state dict (torch.save) here")
    ap.add_argument("--load-state", default="", help="skip training: load this state dict (e.g. the oracle's trained "
                    "weights on the hip side, for an export / count comparison on IDENTICAL weights)")
    ap.add_argument("--count", action="store_true",
                    help="after the last step: 128^3 volume export + first-stage fruit count of the semantic set "
                         "(clustering_base.py:183-207) on both sides")
    args = ap.parse_args()
    from bench import METHODS, split_indices
    M = METHODS[args.method]
    globals()["R"] = args.rays
    globals()["EVAL_AT"] = tuple(int(x) for x in args.eval_at.split(","))
    globals()["EVAL_PIXELS"] = args.eval_pixels
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
    if args.data_cache and os.path.exists(args.data_cache):
        data = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in torch.load(args.data_cache).items()}
    else:
        data = sa.render_dataset(scene, c2w, H=HW, W=HW, fx=focal, fy=focal)
        if args.data_cache:
            torch.save({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in data.items()}, args.data_cache)
    i_train, i_eval = split_indices(N_CAMERAS, TRAIN_SPLIT)
    train_ids = torch.as_tensor(i_train, device=dev)
    n_train = len(i_train)
    torch.manual_seed(0)
    ocfg = fo.FruitNerfModelConfig()
    for k, v in M["model"].items():
        setattr(ocfg, k, v)
    om = fo.FruitModel(ocfg, num_train_data=n_train)                            # identical initial weights
    g = torch.Generator().manual_seed(2024)                                     # rays + jitter stream (CPU generator)
    ge = torch.Generator().manual_seed(7)                                       # held-out pixels
    eval_px = [(int(img), torch.randint(0, HW, (EVAL_PIXELS,), generator=ge), torch.randint(0, HW, (EVAL_PIXELS,), generator=ge))
               for img in i_eval[:5]]
    results = {"side": args.side, "method": args.method, "rays_per_step": R, "evals": []}

    if hip:
        from tests import util
        from fruitnerf_amd import _kernels as K
        from fruitnerf_amd.cameras.camera_optimizers import CameraAdam, CameraOptimizerConfig
        from fruitnerf_amd.rays import RayBundle
        from fruitnerf_amd.training import FusedAdam, fused_train_iteration
        model = util.make_hip_like(om, dev)
        model.train()
        opt = FusedAdam(model, algorithm=M["algorithm"], group_lr={k: dict(v) for k, v in M["groups"].items()})
        cm = M["camera"]
        cam_opt = CameraOptimizerConfig(mode="SO3xR3", lr=cm["lr"], eps=cm["eps"], weight_decay=cm["weight_decay"],
                                        lr_final=cm["lr_final"], max_steps=cm["max_steps"] or 1).setup(n_train, dev)
        cadam = CameraAdam(cam_opt, algorithm=cm["algorithm"])
        batcher = sa.PixelBatcher(data, train_ids, seed=0)
        batcher._set = K.ImageSetArg(data["images"], data["masks"], data["c2w"], data["fx"], data["fy"], data["cx"], data["cy"])
    else:
        om.train()
        groups = om.get_param_groups()
        ocam = oc.CameraOptimizer(n_train)
        cm = M["camera"]
        hyper = [M["groups"]["proposal_networks"], M["groups"]["fields"], cm]
        Opt = torch.optim.Adam if M["algorithm"] == "adam" else torch.optim.RAdam
        OptC = torch.optim.Adam if cm["algorithm"] == "adam" else torch.optim.RAdam
        opts = [Opt(groups["proposal_networks"], lr=hyper[0]["lr"], eps=1e-15),
                Opt(groups["fields"], lr=hyper[1]["lr"], eps=1e-15),
                OptC(ocam.parameters(), lr=cm["lr"], eps=cm["eps"], weight_decay=cm["weight_decay"])]

        def decay(h):   # nerfstudio ExponentialDecay without warm-up (fruit_nerf_config.py:47-56)
            return lambda s: float(np.exp(np.log(h["lr_final"] / h["lr"]) * min(s / h["max_steps"], 1.0)))
        scheds = [torch.optim.lr_scheduler.LambdaLR(o_, decay(h)) for o_, h in zip(opts, hyper) if h.get("lr_final")]
        from oracle import pixel_sampler as ops
        cb = sa.PixelBatcher(data, train_ids, seed=0)               # (holds data / image_ids for the oracle's sampler)

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

    def count_fruits():
        """128^3 volume export of the current weights (bin-centre lattice) -> sizes of the three point sets and the
        first-stage fruit count of the semantic set (radius-outlier removal -> voxel down-sampling -> DBSCAN ->
        centre-distance merge, clustering_base.py:183-258; parameters as in bench.py, scaled to the lattice pitch)."""
        N_EXP = 128
        spacing = 2.0 / N_EXP * 2.0
        kw = dict(nb_points=2, radius=1.8 * spacing, voxel_size=spacing / 4, eps=1.8 * spacing, min_samples=4)
        aabb = ((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0))
        from fruitnerf_amd.clustering import Clustering
        fruit = scene.is_fruit.cpu().numpy()
        fc = Clustering(template_path=None, voxel_size_down_sample=kw["voxel_size"], remove_outliers_nb_points=kw["nb_points"],
                        remove_outliers_radius=kw["radius"], min_samples=kw["min_samples"], apple_template_size=1.0,
                        cluster_merge_distance=0.04, gt_cluster=scene.centers.cpu().numpy()[fruit].astype(np.float64) * 2.0,
                        gt_count=int(scene.n_fruits), template_radius=float(scene.radii.cpu().numpy()[fruit].mean()) * 2.0)
        fc.alpha_surface = min(100.0, 0.6 / spacing)   # the reference's 100 presumes a 1 mm voxel grid
        fc.icp_max_correspondence_distance = max(0.01, 1.5 * spacing)
        full = None
        if hip:
            import copy
            from fruitnerf_amd.clustering import PointCloud
            from fruitnerf_amd.data.fruit_datamanager import ExportDataManager
            from fruitnerf_amd.data.semantics import apple_metadata
            from fruitnerf_amd.export.exporter_utils import sample_volume
            from fruitnerf_amd.fruit_nerf import FruitModel
            em = FruitModel(copy.deepcopy(model.config), apple_metadata(), num_train_data=n_train, device=dev, test_mode="export")
            em.load_state_dict(model.state_dict(), strict=True)
            em.eval()

            class _Pipe:
                pass
            pipe = _Pipe()
            pipe.model, pipe.datamanager = em, ExportDataManager(dev, eval_num_rays_per_batch=32768)
            em.setup_inference(True, N_EXP, deterministic=True)
            n_rays = pipe.datamanager.setup_inference(aabb=aabb, num_points=N_EXP)
            sets = sample_volume(pipe, n_rays, transform_json={"scale": 1.0})
            pts = sets["semantic"]["points"]
            count = 0
            if pts.shape[0] >= 5:
                full = fc.count(PointCloud(pts, None, dev), eps=kw["eps"])
                count = fc.counter - fc.fuse_counter
        else:
            from oracle import cloud as ocl
            emo = fo.FruitModel(ocfg, num_train_data=n_train, test_mode="export")
            emo.load_state_dict(om.state_dict(), strict=True)
            emo.eval()
            emo.setup_inference(True, N_EXP)
            sets = fo.sample_volume(emo, torch.tensor(aabb), N_EXP, 32768)
            pts = sets["semantic"]["points"].numpy()
            count = 0
            if pts.shape[0] >= 5:
                X, _, labels = ocl.cluster_front_end(pts, None, kw["nb_points"], kw["radius"], kw["voxel_size"], kw["eps"],
                                                     kw["min_samples"])
                Xs, ls = fc.merge_small_clusters(X, None, labels)
                count = fc.counter - fc.fuse_counter
                full = fc.split_large_cluster(Xs, None, ls)
        import hashlib
        rec = {"export_lattice": f"{N_EXP}^3", "export_points": {k: int(v["points"].shape[0]) for k, v in sets.items()},
               "export_points_sha1": {k: hashlib.sha1(np.ascontiguousarray(
                   (v["points"].cpu().numpy() if torch.is_tensor(v["points"]) else v["points"]).astype(np.float64)).tobytes()).hexdigest()
                   for k, v in sets.items()},
               "fruit_count": None if full is None else int(full), "fruit_count_first_stage": int(count),
               "fruit_count_scene": int(scene.n_fruits)}
        results["count"] = rec
        print(json.dumps(rec), flush=True)

    if args.load_state:
        sd = torch.load(args.load_state, map_location="cpu")
        if hip:
            model.load_state_dict({k: v.to(dev) for k, v in sd.items()}, strict=True)
        else:
            om.load_state_dict(sd, strict=True)
        results["loaded_state"] = args.load_state
        evaluate(-1)
        if args.count:
            count_fruits()
        with open(args.out, "w") as f:
            json.dump(results, f, indent=1)
        return

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
            o, d, cam, batch = ops.sample_pixels(cb.data, cb.image_ids, u)
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
    if args.save_state:
        sd = (model if hip else om).state_dict()
        torch.save({k: v.detach().cpu() for k, v in sd.items()}, args.save_state)
    if args.count:
        count_fruits()
    with open(args.out, "w") as f:
        json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
