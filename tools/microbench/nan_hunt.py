"""Find the first step whose gradients contain non-finite values and say where."""
import sys, torch
sys.path.insert(0, '/root/repo')
from fruitnerf_amd.data import synthetic_apple as sa
from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
from fruitnerf_amd.data.semantics import apple_metadata
from fruitnerf_amd.rays import RayBundle
from fruitnerf_amd.training import FusedAdam, fused_forward_backward
dev = torch.device('cuda:0')
HW = 200; focal = 1111.0 * HW / 800
scene = sa.make_scene(seed=0, device=dev); c2w = sa.make_cameras(100, seed=0, device=dev)
data = sa.render_dataset(scene, c2w, H=HW, W=HW, fx=focal, fy=focal)
batcher = sa.PixelBatcher(data, torch.arange(90, device=dev), seed=1)
torch.manual_seed(0)
model = FruitModel(FruitNerfModelConfig(), apple_metadata(), num_train_data=90, device=dev); model.train()
opt = FusedAdam(model)
arena = model.arena()
import fruitnerf_amd._kernels as K
_wb, _pb, _il = K.weights_bwd, K.prop_density_bwd, K.interlevel_fwd
def chk(tag, t):
    if t is not None and torch.is_tensor(t) and t.is_floating_point() and not torch.isfinite(t).all():
        bad = ~torch.isfinite(t)
        print("   NONFINITE", tag, tuple(t.shape), "count", int(bad.sum()), "first idx", bad.nonzero()[0].tolist())
        return True
    return False
def wb(S, euclid, density, weights, d_w, up):
    out = _wb(S, euclid, density, weights, d_w, up)
    if CHECK[0]:
        for tag, t in (("wb.euclid", euclid), ("wb.density", density), ("wb.weights", weights), ("wb.d_w", d_w), ("wb.out", out)):
            chk(f"S={S} {tag}", t)
        if not torch.isfinite(out).all():
            bad = (~torch.isfinite(out)).nonzero()[0].tolist()
            r = bad[0] if len(bad) > 1 else bad[0] // S
            print("      ray", r, "euclid", euclid.view(-1, S + 1)[r][:8].tolist(), "...", euclid.view(-1, S + 1)[r][-4:].tolist())
            print("      density max", float(density.view(-1, S)[r].max()), "weights sum", float(weights.view(-1, S)[r].sum()), "d_w absmax", float(d_w.view(-1, S)[r].abs().max()))
    return out
def pb(net, grads, warp, rays, euclid, S, feats, d_density, want_position_grad=False):
    if CHECK[0]:
        chk(f"pb S={S} feats", feats); chk(f"pb S={S} d_density", d_density)
    return _pb(net, grads, warp, rays, euclid, S, feats, d_density, want_position_grad)
K.weights_bwd, K.prop_density_bwd = wb, pb
CHECK = [False]
for step in range(8000):
    o, d, cam, batch = batcher.sample(4096)
    model.set_anneal(step)
    ld, md = fused_forward_backward(model, RayBundle(o, d, None, cam), batch)
    CHECK[0] = step >= 2100
    if step >= 2100:
        bad = ~torch.isfinite(arena.grads)
        gmax = float(arena.grads.abs().max())
        if bool(bad.any()) or gmax > 1e6 or not all(torch.isfinite(v) for v in ld.values()):
            print("step", step, "non-finite grads", int(bad.sum()), "max|g|", gmax, {k: float(v) for k, v in ld.items()})
            for name, p in model.named_parameters():
                g = p.grad
                if not torch.isfinite(g).all() or float(g.abs().max()) > 1e6:
                    print("   ", name, "nonfinite", int((~torch.isfinite(g)).sum()), "max", float(g.abs().max()))
            break

    opt.step()
    model.proposal_sampler.step_cb(step)
else:
    print("no non-finite gradient in 8000 steps")
