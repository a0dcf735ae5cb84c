"""Which inputs make the binned scatter overflow a queue (fnr_debug_scatter_overflows)?"""
import sys, torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import util
from fruitnerf_amd import _kernels as K, _lib as L
from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
from fruitnerf_amd.data.semantics import apple_metadata

dev = torch.device("cuda", 0)
m = FruitModel(FruitNerfModelConfig(), apple_metadata(), num_train_data=16, device=dev)
m.train(); m.arena()
fld = m.field
R, S = 4096, 48
g = torch.Generator(device=dev).manual_seed(1)
d_feats = torch.randn(16, R * S, 2, device=dev, generator=g) * 1e-3
gnet = fld.net_struct(grads=True)
o1, d1, _, _ = util.random_rays(1, 16, seed=11)
cases = {}
o, d, _, cam = util.random_rays(R, 16, seed=9)
cases["random rays"] = (o.to(dev), d.to(dev), 0.05, 1000.0)
cases["4096 copies of one ray"] = (o1.to(dev).expand(R, 3).contiguous(), d1.to(dev).expand(R, 3).contiguous(), 0.05, 1000.0)
cases["one ray, near = far (one point)"] = (o1.to(dev).expand(R, 3).contiguous(), d1.to(dev).expand(R, 3).contiguous(), 0.5, 0.5)
cases["one ray, samples within 0.3..0.31"] = (o1.to(dev).expand(R, 3).contiguous(), d1.to(dev).expand(R, 3).contiguous(), 0.3, 0.31)
# one sample per ray, rays alternating between two fixed ones: neighbouring samples never share a cell (no run to pre-sum),
# yet every record of a level lands in the same one or two bins
N1 = R * S
o2, d2, _, _ = util.random_rays(2, 16, seed=5)
alt = (torch.arange(N1, device=dev) % 2)
cases["two alternating points, S = 1"] = (o2.to(dev)[alt].contiguous(), d2.to(dev)[alt].contiguous(), 0.4, 0.4)
for name, (oo, dd, near, far) in cases.items():
    R, S = (N1, 1) if "S = 1" in name else (4096, 48)
    rays = K.RaysArg(oo, dd, torch.full((R, 1), near, device=dev), torch.full((R, 1), far, device=dev),
                     torch.zeros(R, 1, dtype=torch.long, device=dev))
    _, eu = K.sample_spaced(rays, 1, S, None)
    m.arena().grads.zero_()
    L.scatter_overflows(reset=True)
    K.hash_encode_bwd(gnet.grid, fld.warp_struct(), rays, eu, S, d_feats)
    n = L.scatter_overflows(reset=True)
    tg = fld.mlp_base_grid.hash_table.grad.view(16, 1 << 19, 2)
    err = float(((tg.double().sum(1) - d_feats.double().sum(1)).abs() / d_feats.abs().double().sum(1)).max())
    print(f"{name}: {n} overflowed records; rows touched per level {[int((tg[l] != 0).any(1).sum()) for l in (0, 1, 4, 15)]}; "
          f"column-sum error {err:.2e}", flush=True)
