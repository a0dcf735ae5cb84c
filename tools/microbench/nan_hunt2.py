"""Find the first training step where the loss jumps / parameters or gradients become abnormal."""
import sys, torch, collections
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
opt = FusedAdam(model); arena = model.arena()
groups = {}
for gname, p, off, n in arena.entries:
    key = ("table" if "hash_table" in [k for k, v in model.named_parameters() if v is p][0] else "mlp") + ":" + gname
    groups.setdefault(key, []).append((off, n))
def stats(t):
    out = {}
    for k, spans in groups.items():
        m = max(float(t[a:a + n].abs().max()) for a, n in spans)
        nf = sum(int((~torch.isfinite(t[a:a + n])).sum()) for a, n in spans)
        out[k] = (round(m, 6), nf)
    return out
hist = collections.deque(maxlen=3)
for step in range(9000):
    o, d, cam, batch = batcher.sample(4096)
    model.set_anneal(step)
    ld, md = fused_forward_backward(model, RayBundle(o, d, None, cam), batch)
    if step >= 1500:
        rec = (step, {k: round(float(v), 6) for k, v in ld.items()}, "grad", stats(arena.grads), "param", stats(arena.params))
        bad = rec[1]["rgb_loss"] > 0.004 or any(v[1] for v in rec[3].values()) or any(v[0] > 50 for v in rec[5].values()) or any(v[0] > 10 for v in rec[3].values())
        if bad:
            for h in hist: print(h)
            print("BAD", rec)
            break
        hist.append(rec)
    opt.step()
    model.proposal_sampler.step_cb(step)
else:
    print("clean run")
