"""Which state diverges first between two training runs from one seed?  (Round 6: the hunt for what `nt` loads of the Jacobian
in k_field_mlp_bwd_base_coop do to run-to-run reproducibility; run with FNR_LIB_PATH pointing at a variant library.)
usage: python tests/diagnostics/nt_jac_probe.py [steps = 60] [runs = 3]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(dev, steps):
    import fruitnerf_amd.training as T
    from fruitnerf_amd.cameras.camera_optimizers import CameraAdam, CameraOptimizerConfig
    from fruitnerf_amd.data import synthetic_apple as sa
    from fruitnerf_amd.data.semantics import apple_metadata
    from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
    HW, focal, n_train = 96, 1111.0 * 96 / 800, 40
    scene = sa.make_scene(seed=0, device=dev)
    c2w = sa.make_cameras(n_train, seed=0, device=dev)
    data = sa.render_dataset(scene, c2w, H=HW, W=HW, fx=focal, fy=focal)
    batcher = sa.PixelBatcher(data, torch.arange(n_train, device=dev), seed=1)
    torch.manual_seed(0)
    hm = FruitModel(FruitNerfModelConfig(mlp_precision="bf16x3"), apple_metadata(), num_train_data=n_train, device=dev)
    hm.train()
    opt = T.FusedAdam(hm)
    cam_opt = CameraOptimizerConfig(mode="SO3xR3").setup(n_train, dev)
    loop = T.TrainingSteps(hm, opt, batcher, 4096, camera=(cam_opt, CameraAdam(cam_opt)))
    arena = hm.arena()
    groups = {"proposal_networks": arena.group_ranges["proposal_networks"], "fields": arena.group_ranges["fields"]}
    table = hm.field.mlp_base_grid.hash_table
    ta, tn = [(off, n) for _, p, off, n in arena.entries if p is table][0]
    rec = torch.zeros(steps, 5, dtype=torch.int64, device=dev)
    for i in range(steps):
        loop.step(want_metrics=False)
        P = arena.params.view(torch.int32)
        rec[i, 0] = P[ta:ta + tn].sum(dtype=torch.int64)                                   # main hash table
        fa, fb = groups["fields"]
        rec[i, 1] = P[fa:fb].sum(dtype=torch.int64) - rec[i, 0]                            # field MLP weights + embedding
        pa, pb = groups["proposal_networks"]
        rec[i, 2] = P[pa:pb].sum(dtype=torch.int64)                                        # proposal networks
        rec[i, 3] = cam_opt.pose_adjustment.data.view(torch.int32).sum(dtype=torch.int64)  # camera poses
        rec[i, 4] = cam_opt.pose_adjustment.grad.view(torch.int32).sum(dtype=torch.int64)
    torch.cuda.synchronize()
    return rec.cpu(), dict(loop.stats)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device("cuda:0")
    names = ["hash table", "field MLPs", "proposal nets", "camera poses", "pose grad"]
    ref, stats = run(dev, steps)
    print("lib", os.environ.get("FNR_LIB_PATH", "default"), "sequencer", os.environ.get("FNR_NATIVE_SEQUENCER", "1"), stats)
    for r in range(1, runs):
        got, _ = run(dev, steps)
        diff = (got != ref)
        if not diff.any():
            print(f"run {r}: identical to run 0 over {steps} steps")
            continue
        first = int(diff.any(dim=1).nonzero()[0])
        print(f"run {r}: first difference after step {first}: " + ", ".join(n for n, d in zip(names, diff[first].tolist()) if d)
              + f"; at the end: " + ", ".join(n for n, d in zip(names, diff[-1].tolist()) if d))


if __name__ == "__main__":
    main()
