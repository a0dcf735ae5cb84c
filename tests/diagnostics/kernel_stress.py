"""Is one forward + backward of a method a pure function of its inputs?  The same batch, sampler jitters and weights
through training.fused_forward_backward (gradients materialised, no optimiser step) `reps` times, while a third stream
keeps the GPU busy with unrelated GEMMs that perturb wave scheduling; every repetition's gradient arena, ray gradients
and losses must have the digest of the first.  A kernel with an intra-kernel race shows up here as a rare odd one out.
usage: kernel_stress.py [method] [reps] [state: init | trained<steps>]"""
import sys, torch
sys.path.insert(0, ".")
import bench
import fruitnerf_amd.training as T
from fruitnerf_amd.data import synthetic_apple as sa
from fruitnerf_amd.rays import RayBundle

dev = torch.device("cuda", 0)
scene = sa.make_scene(seed=0, device=dev)
c2w = sa.make_cameras(bench.N_CAMERAS, seed=0, device=dev)
data = sa.render_dataset(scene, c2w, H=800, W=800, fx=1111.0, fy=1111.0)
i_train, _ = bench.split_indices(bench.N_CAMERAS, bench.TRAIN_SPLIT)
method = sys.argv[1] if len(sys.argv) > 1 else "fruit_nerf_big"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
state = sys.argv[3] if len(sys.argv) > 3 else "trained300"
r = bench.MethodRun(method, "bf16x3", "SO3xR3", dev, 0, 1, data, torch.as_tensor(i_train, device=dev), len(i_train))
if state.startswith("trained"):
    for _ in range(int(state[7:])):
        r.one_step(want_metrics=False)
r.steps.drop_lookahead()
model = r.model
o, d, cam, batch = r.batcher.sample(r.rays, r.camera[0], level0=model.level0_spec())
pre = r.batcher.last_presample
model.proposal_sampler._steps_since_update = 100      # every repetition trains the proposal networks too
busy = torch.cuda.Stream(device=dev)
a = torch.randn(2048, 2048, device=dev)


def digest(*ts):
    """Order-sensitive 64-bit checksum computed on the device (a .cpu() copy of the 300 MB arena per repetition cost
    0.2 s: 9 GPU-minutes for 2400 repetitions)."""
    out = []
    for t in ts:
        v = t.detach().contiguous().view(-1).view(torch.int32).to(torch.int64)
        w = torch.arange(1, v.numel() + 1, device=v.device, dtype=torch.int64) % 1000003
        out.append((int(v.sum()), int((v * w).sum())))
    return tuple(out)


first, odd = None, []
for overlap in (False, True):
    for k in range(reps):
        model.arena().zero_grad()
        if k % 3:
            with torch.cuda.stream(busy):
                for _ in range(1 + k % 4):
                    a @ a
        rg = {}
        model.proposal_sampler._steps_since_update = 100
        ld, md = T.fused_forward_backward(model, RayBundle(o, d, None, cam, presampled=dict(pre)), batch, ray_grads=rg,
                                          overlap_proposal_backward=overlap)
        torch.cuda.synchronize()
        dg = (digest(model.arena().grads), digest(rg["origins"], rg["directions"]),
              digest(ld["rgb_loss"], ld["semantics_loss"], ld["interlevel_loss"], md["psnr"], md["distortion"]))
        if first is None:
            first = dg
        elif dg != first:
            odd.append((overlap, k, [i for i in range(3) if dg[i] != first[i]]))
    print(f"{method} {state}, second stream {overlap}: {reps} repetitions, odd ones out so far: {odd[:8]} ({len(odd)})", flush=True)
