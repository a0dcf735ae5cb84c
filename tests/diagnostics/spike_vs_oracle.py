"""Train until the gradient spikes, then replay THAT step in the CPU oracle (same weights, rays, jitter, batch) and
compare the gradients: is the spike a property of the model or of the HIP path?"""
import sys, torch, collections
sys.path.insert(0, '/root/repo')
from fruitnerf_amd.data import synthetic_apple as sa
from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
from fruitnerf_amd.data.semantics import apple_metadata
from fruitnerf_amd.rays import RayBundle
from fruitnerf_amd.training import FusedAdam, fused_forward_backward
from oracle import fruit_oracle as fo, ns_torch as ns
dev = torch.device('cuda:0')
HW = 200; focal = 1111.0 * HW / 800
scene = sa.make_scene(seed=0, device=dev); c2w = sa.make_cameras(100, seed=0, device=dev)
data = sa.render_dataset(scene, c2w, H=HW, W=HW, fx=focal, fy=focal)
batcher = sa.PixelBatcher(data, torch.arange(90, device=dev), seed=1)
torch.manual_seed(0)
model = FruitModel(FruitNerfModelConfig(), apple_metadata(), num_train_data=90, device=dev); model.train()
opt = FusedAdam(model); arena = model.arena()
R = 4096
med = collections.deque(maxlen=200)
for step in range(12000):
    o, d, cam, batch = batcher.sample(R)
    jit = [torch.rand(R, 1, device=dev) for _ in range(3)]
    model.set_anneal(step)
    samp = model.proposal_sampler
    state = (samp._step, samp._steps_since_update)
    ld, md = fused_forward_backward(model, RayBundle(o, d, None, cam), batch, jitter=jit)
    if step >= 1000:
        gmax = float(arena.grads.abs().max())
        if len(med) >= 100 and gmax > 30 * sorted(med)[len(med) // 2]:
            print("spike at step", step, "max|g|", gmax, "median", sorted(med)[len(med) // 2], {k: float(v) for k, v in ld.items()})
            hip_grads = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()}
            om = fo.FruitModel(fo.FruitNerfModelConfig(), num_train_data=90)
            om.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()}, strict=True)
            om.train()
            om.proposal_sampler._step, om.proposal_sampler._steps_since_update = state
            om.set_anneal(step)
            torch.set_num_threads(32)
            out = om(ns.RayBundle(o.cpu(), d.cpu(), torch.ones(R, 1), camera_indices=cam.cpu().long()), jitter=[j.cpu() for j in jit])
            b = {k: v.cpu() for k, v in batch.items()}
            old = om.get_loss_dict(out, b)
            sum(old.values()).backward()
            print("oracle losses", {k: float(v) for k, v in old.items()})
            with torch.no_grad():
                samp._step, samp._steps_since_update = state
                hout, hctx = model._render(model._collide(RayBundle(o, d, None, cam)), jit)
            for k in ("rgb", "semantics", "accumulation", "depth"):
                df = (hout[k].cpu() - out[k].detach()).abs()
                print(f"   forward {k}: max diff {float(df.max()):.3e} at ray {int(df.view(R, -1).max(1)[0].argmax())}, mean {float(df.mean()):.3e}")
            r = int((hout["rgb"].cpu() - out["rgb"].detach()).abs().view(R, -1).max(1)[0].argmax())
            S = 48
            print("   worst ray", r, "hip rgb", hout["rgb"][r].tolist(), "oracle rgb", out["rgb"][r].tolist())
            wl = out["weights_list"][-1][r, :, 0].detach()
            print("   oracle weights top", wl.topk(5))
            print("   hip    weights top", hctx.weights.view(R, S)[r].cpu().topk(5))
            hs = hctx.sample_rgb.view(R, S, 3)[r].cpu(); hd = hctx.sample_density.view(R, S)[r].cpu()
            print("   hip density max", float(hd.max()), "finite", bool(torch.isfinite(hd).all()), "rgb finite", bool(torch.isfinite(hs).all()))
            rs = out["ray_samples_list"][-1]
            fo_ = om.field.forward(rs[r:r + 1]) if hasattr(rs, "__getitem__") else None
            if fo_ is not None:
                orgb = fo_[list(fo_.keys())[0]]
                for kk, vv in fo_.items():
                    print("   oracle field", kk, tuple(vv.shape), "absmax", float(vv.detach().abs().max()))
                import fruitnerf_amd.fruit_field as ff
                orgb = fo_[ff.FieldHeadNames.RGB][0].detach() if ff.FieldHeadNames.RGB in fo_ else None
                if orgb is None:
                    orgb = [v for k_, v in fo_.items() if "rgb" in str(k_).lower()][0][0].detach()
                dsamp = (hs - orgb).abs().max(1)[0]
                kbad = int(dsamp.argmax())
                print("   per-sample rgb max diff", float(dsamp.max()), "at sample", kbad, "hip", hs[kbad].tolist(), "oracle", orgb[kbad].tolist(), "hip density there", float(hd[kbad]))
            for n, p in om.named_parameters():
                gr = p.grad if p.grad is not None else torch.zeros_like(p)
                gh = hip_grads[n]
                print(f"   {n:55s} max|oracle| {float(gr.abs().max()):.4e}  max|hip| {float(gh.abs().max()):.4e}  max|diff| {float((gr - gh).abs().max()):.4e}")
            break
        med.append(gmax)
    opt.step()
    samp.step_cb(step)
else:
    print("no spike in 12000 steps")
