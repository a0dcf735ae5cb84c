import sys, torch
sys.path.insert(0, '/root/repo')
from fruitnerf_amd import _lib as L, _kernels as K
from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
from fruitnerf_amd.data.semantics import apple_metadata
dev = torch.device('cuda:0')
for log2 in (19, 16, 12):
    m = FruitModel(FruitNerfModelConfig(log2_hashmap_size=log2), apple_metadata(), num_train_data=10, device=dev); m.train(); m.arena()
    fld = m.field
    R, S = 4096, 48
    o = torch.randn(R, 3, device=dev) * 0.3; d = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
    cam = torch.randint(0, 10, (R,), device=dev)
    rays = K.RaysArg(o, d, torch.full((R,), 0.05, device=dev), torch.full((R,), 4.0, device=dev), cam)
    sp, eu = K.sample_spaced(rays, 1, S, None)
    net = fld.net_struct()
    L.profile_enable(True)
    for _ in range(20):
        feats, sel = K.hash_encode_fwd(net.grid, fld.warp_struct(), rays, eu, S)
    torch.cuda.synchronize()
    recs = L.profile_collect(); L.profile_enable(False)
    ms = sorted(r[2] for r in recs)
    print("log2_T", log2, "encode median us", round(ms[len(ms)//2]*1e3, 1))
