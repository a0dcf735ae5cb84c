import sys, torch
sys.path.insert(0, '/root/repo')
from fruitnerf_amd import _lib as L, _kernels as K
from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
from fruitnerf_amd.data.semantics import apple_metadata
dev = torch.device('cuda:0')
m = FruitModel(FruitNerfModelConfig(), apple_metadata(), num_train_data=10, device=dev); m.train(); m.arena()
fld = m.field
import os
for mode, R in [(md, int(x)) for md in os.environ.get("MODES", "fp32,bf16x3,bf16").split(",")
                for x in os.environ.get("RS", "16,512,4096").split(",")]:
    fld.mlp_precision = mode
    S = 48; N = R * S
    o = torch.randn(R, 3, device=dev) * 0.3; d = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
    cam = torch.randint(0, 10, (R,), device=dev)
    rays = K.RaysArg(o, d, torch.full((R,), 0.05, device=dev), torch.full((R,), 4.0, device=dev), cam)
    sp, eu = K.sample_spaced(rays, 1, S, None)
    net, gnet = fld.net_struct(), fld.net_struct(grads=True)
    feats, sel = K.hash_encode_fwd(net.grid, fld.warp_struct(), rays, eu, S)
    L.profile_enable(True)
    for _ in range(20):
        den, rgb, lg, _, h = K.field_mlp_fwd(net, rays, S, feats, sel, None, want_h=True)
        K.field_mlp_bwd(net, gnet, rays, S, feats, h, sel, den, rgb, lg)
    torch.cuda.synchronize()
    recs = L.profile_collect(); L.profile_enable(False)
    import collections
    agg = collections.defaultdict(list)
    for op, u, ms in recs: agg[op].append(ms)
    print(mode, R, {k: round(sorted(v)[len(v)//2]*1e3, 1) for k, v in agg.items()}, "us")
