import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from tests import util
from tests.test_golden import _load, _inputs
from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
from fruitnerf_amd.data.semantics import apple_metadata
from fruitnerf_amd.rays import RayBundle
from fruitnerf_amd.training import fused_forward_backward
import fruitnerf_amd.training as T
from oracle import ns_torch as ns
dev = torch.device('cuda:0')
g, sd = _load()
oc = util.small_config(log2=10, prop_log2=8)
cfg = FruitNerfModelConfig()
for k, v in vars(oc).items():
    if hasattr(cfg, k): setattr(cfg, k, v)
o, d, pa, cam, jit, batch = _inputs(g, dev)
ref_o, ref_d = torch.from_numpy(g["grad::origins"]), torch.from_numpy(g["grad::directions"])
def run(tag, mult=None, no_jac=False):
    hm = FruitModel(cfg, apple_metadata(), num_train_data=5, device=dev); hm.load_state_dict(sd, strict=True); hm.train(); hm.set_anneal(0)
    if mult is not None: hm.config.interlevel_loss_mult = mult
    if no_jac:
        orig = T._field_ray_grads
        def f(model, rctx, d_feats, a, b):
            rctx.field_jacobian = None
            return orig(model, rctx, d_feats, a, b)
        T._field_ray_grads = f
    got = {}
    fused_forward_backward(hm, RayBundle(o, d, pa, cam), batch, jitter=jit, ray_grads=got)
    if no_jac: T._field_ray_grads = orig
    torch.cuda.synchronize()
    return got["origins"].cpu(), got["directions"].cpu()
a = run("full")
print("full      vs golden: o err", (a[0]-ref_o).abs().max().item(), "d err", (a[1]-ref_d).abs().max().item(), "scale", ref_o.abs().max().item(), ref_d.abs().max().item())
b = run("gather", no_jac=True)
print("gather    vs golden: o err", (b[0]-ref_o).abs().max().item(), "jac vs gather", (a[0]-b[0]).abs().max().item())
# oracle without interlevel contribution
om = util.make_oracle(oc, num_images=5, seed=0, randomize=False); om.load_state_dict(sd, strict=True); om.train(); om.set_anneal(0)
oc_, dc_ = o.cpu().clone().requires_grad_(True), d.cpu().clone().requires_grad_(True)
tr = om(ns.RayBundle(oc_, dc_, pa.cpu(), camera_indices=cam.cpu()), jitter=[j.cpu() for j in jit])
ld = om.get_loss_dict(tr, {k: v.cpu() for k, v in batch.items()})
(ld["rgb_loss"] + ld["semantics_loss"]).backward()
c = run("nointer", mult=0.0)
print("no-interlevel: hip vs oracle o err", (c[0]-oc_.grad).abs().max().item(), "scale", oc_.grad.abs().max().item())
r = (a[0]-ref_o).abs().max(1)[0].argmax().item()
print("worst ray", r, "hip", a[0][r].tolist(), "golden", ref_o[r].tolist())

# ---- per-sample gradient w.r.t. the unit-cube positions: oracle (retain_grad) vs HIP gather kernel ----
from fruitnerf_amd import _kernels as K
om = util.make_oracle(oc, num_images=5, seed=0, randomize=False); om.load_state_dict(sd, strict=True); om.train(); om.set_anneal(0)
oc_, dc_ = o.cpu().clone().requires_grad_(True), d.cpu().clone().requires_grad_(True)
tr = om(ns.RayBundle(oc_, dc_, pa.cpu(), camera_indices=cam.cpu()), jitter=[j.cpu() for j in jit])
xw = om.field._sample_locations
xw.retain_grad()
ld = om.get_loss_dict(tr, {k: v.cpu() for k, v in batch.items()})
(ld["rgb_loss"] + ld["semantics_loss"]).backward()
gx_ref = xw.grad            # [R, S, 3]
hm = FruitModel(cfg, apple_metadata(), num_train_data=5, device=dev); hm.load_state_dict(sd, strict=True); hm.train(); hm.set_anneal(0)
hm.config.interlevel_loss_mult = 0.0
captured = {}
orig = T._field_ray_grads
def f(model, rctx, d_feats, a, b):
    fld = model.field; lv = rctx.levels[-1]
    captured["partial"] = K.hash_encode_input_grad(fld.net_struct().grid, fld.warp_struct(), rctx.rays, lv["euclid"], lv["S"], d_feats).sum(0)
    captured["euclid"] = lv["euclid"]
    return orig(model, rctx, d_feats, a, b)
T._field_ray_grads = f
got = {}
fused_forward_backward(hm, RayBundle(o, d, pa, cam), batch, jitter=jit, ray_grads=got)
torch.cuda.synchronize()
gx_hip = captured["partial"][:, :3].view(gx_ref.shape[0], -1, 3).cpu()
dif = (gx_hip - gx_ref).abs()
print("unit-cube position grads: max|ref|", gx_ref.abs().max().item(), "max err", dif.max().item(), "at", np.unravel_index(dif.argmax().item(), dif.shape))
r = 36
print("ray 36 per-sample err", dif[r].max(1)[0].tolist()[:48])
k = dif[r].max(1)[0].argmax().item()
print("worst sample", k, "hip", gx_hip[r, k].tolist(), "ref", gx_ref[r, k].tolist(), "xw", xw[r, k].tolist())

# ---- brute force for the worst sample: oracle HashEncoding autograd in float32 and float64 with HIP's d_feats ----
def f2(model, rctx, d_feats, a, b):
    captured["d_feats"] = d_feats.clone()
    return orig(model, rctx, d_feats, a, b)
T._field_ray_grads = f2
hm2 = FruitModel(cfg, apple_metadata(), num_train_data=5, device=dev); hm2.load_state_dict(sd, strict=True); hm2.train(); hm2.set_anneal(0)
hm2.config.interlevel_loss_mult = 0.0
fused_forward_backward(hm2, RayBundle(o, d, pa, cam), batch, jitter=jit, ray_grads={})
torch.cuda.synchronize()
S = 48
n = r * S + k
df = captured["d_feats"][:, n, :].cpu()            # [L, 2]
enc = om.field.mlp_base_grid
for dt in (torch.float32, torch.float64):
    x = xw[r, k].detach().to(dt).clone().requires_grad_(True)
    enc_d = enc if dt == torch.float32 else None
    if dt == torch.float64:
        import copy
        enc_d = copy.deepcopy(enc).double()
        enc_d.scalings = enc.scalings.double() if hasattr(enc, "scalings") and torch.is_tensor(enc.scalings) else enc.scalings
    feat = enc_d(x[None, :])                       # [1, L*2]
    (feat.view(-1, 2) * df.to(dt)).sum().backward()
    print(dt, "d/dx via oracle HashEncoding autograd:", x.grad.tolist())
print("hip gather kernel:", gx_hip[r, k].tolist(), " oracle model autograd:", gx_ref[r, k].tolist())
sc = enc.scalings if torch.is_tensor(enc.scalings) else torch.tensor(enc.scalings)
print("scaled y per level:", (xw[r, k, 1].detach() * sc.float()).tolist())
