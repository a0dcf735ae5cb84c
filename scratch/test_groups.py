# level-group scatter == single scatter (same gradients)
import sys, torch
sys.path.insert(0, '/root/repo')
from tests import util
from fruitnerf_amd.rays import RayBundle
from fruitnerf_amd.training import fused_forward_backward, _FieldGradientExchange
import fruitnerf_amd.training as T
dev = torch.device('cuda:0')
cfg = util.small_config(log2=15, prop_log2=13)
om = util.make_oracle(cfg, seed=9)
R = 192
o, d, pa, cam = util.random_rays(R, 7, seed=4)
jit = [torch.rand(R, 1).to(dev) for _ in range(3)]
g = torch.Generator().manual_seed(8)
hb = {"image": torch.rand(R, 3, generator=g).to(dev), "fruit_mask": (torch.rand(R, 1, generator=g) > 0.6).float().to(dev)}
res = []
T.start_gradient_sync = lambda arena, span, world, bucket_elems=0: [(span[0], span[1], None)]
for grouped in (False, True):
    hm = util.make_hip_like(om, dev); hm.train(); hm.set_anneal(0)
    rb = RayBundle(o.to(dev), d.to(dev), pa.to(dev), cam.to(dev))
    ex = _FieldGradientExchange(hm, 2) if grouped else None
    fused_forward_backward(hm, rb, hb, jitter=jit, exchange=ex)
    torch.cuda.synchronize()
    if ex: print("pending spans", [(a, b) for a, b, _ in ex.pending], hm.arena().group_ranges)
    res.append(hm.arena().grads.clone())
print("max diff", (res[0]-res[1]).abs().max().item(), "scale", res[0].abs().max().item(), "nnz", int((res[0]!=0).sum()), int((res[1]!=0).sum()))
