"""fruit_nerf_big at its real configuration: are two runs from one seed bit-identical, and does the second stream /
sampling ahead change anything?  (tests/test_gpu_determinism.py covers fruit_nerf.)"""
import sys, torch
sys.path.insert(0, ".")
import bench
import fruitnerf_amd.training as T
from fruitnerf_amd.data import synthetic_apple as sa

dev = torch.device("cuda", 0)
scene = sa.make_scene(seed=0, device=dev)
c2w = sa.make_cameras(bench.N_CAMERAS, seed=0, device=dev)
data = sa.render_dataset(scene, c2w, H=800, W=800, fx=1111.0, fy=1111.0)
i_train, _ = bench.split_indices(bench.N_CAMERAS, bench.TRAIN_SPLIT)
method = sys.argv[1] if len(sys.argv) > 1 else "fruit_nerf_big"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
eval_at = {int(a) for a in sys.argv[3].split(',')} if len(sys.argv) > 3 else set()


def run(overlap, ahead):
    T.OVERLAP_PROPOSAL_BACKWARD, T.SAMPLE_AHEAD = overlap, ahead
    r = bench.MethodRun(method, "bf16x3", "SO3xR3", dev, 0, 1, data, torch.as_tensor(i_train, device=dev), len(i_train))
    from fruitnerf_amd.rays import RayBundle
    for i in range(steps):
        r.one_step()
        if eval_at and i + 1 in eval_at:     # what bench.py's quality gate does between two training steps
            r.model.eval()
            with torch.no_grad():
                g = torch.Generator(device=dev); g.manual_seed(7)
                n = 32768
                y = torch.randint(0, 800, (n,), device=dev, generator=g); x = torch.randint(0, 800, (n,), device=dev, generator=g)
                o, d = sa.pixel_rays(c2w, torch.full((n,), 3, device=dev), y, x, 1111.0, 1111.0, 400.0, 400.0)
                for _ in range(3):
                    r.model(RayBundle(o, d, None, None))
            r.model.train()
    torch.cuda.synchronize()
    return r.model.arena().params.clone(), r.opt.exp_avg.clone(), r.camera[0].pose_adjustment.data.clone()


ref = run(False, False)
for name, (o, a) in {"same again": (False, False), "second stream": (True, False), "sampling ahead": (False, True),
                     "both": (True, True), "both again": (True, True)}.items():
    got = run(o, a)
    print(name, [int((x != y).sum()) for x, y in zip(ref, got)], "entries differ (params, exp_avg, poses)", flush=True)
