"""prop_density_bwd (k_prop_bwd + k_prop_reduce + binned scatter) for the two proposal networks of `fruit_nerf`, with and
without the position gradient (camera optimiser), uniformly spaced samples on random rays."""
import sys

import torch

sys.path.insert(0, __import__('os').getcwd())   # run from the root of the build under test
from fruitnerf_amd import _kernels as K   # noqa: E402
from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig   # noqa: E402
from fruitnerf_amd.data.semantics import apple_metadata   # noqa: E402

dev = torch.device('cuda:0')
m = FruitModel(FruitNerfModelConfig(), apple_metadata(), num_train_data=10, device=dev)
m.train()
m.arena()
R = 4096
o = torch.randn(R, 3, device=dev) * 0.3
d = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
rays = K.RaysArg(o, d, torch.full((R,), 0.05, device=dev), torch.full((R,), 4.0, device=dev), None)
for i, S in ((0, 256), (1, 96)):
    pn = m.proposal_networks[i]
    sp, eu = K.sample_spaced(rays, 1, S, None)
    dens, feats = K.prop_density_fwd(pn.prop_struct(), pn.warp_struct(), rays, eu, S, save_feats=True)
    d_density = torch.randn(R, S, device=dev) * 1e-3
    for pos in (False, True):
        for _ in range(3):
            K.prop_density_bwd(pn.prop_struct(), pn.prop_struct(grads=True), pn.warp_struct(), rays, eu, S, feats,
                               d_density, want_position_grad=pos)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(10):
            K.prop_density_bwd(pn.prop_struct(), pn.prop_struct(grads=True), pn.warp_struct(), rays, eu, S, feats,
                               d_density, want_position_grad=pos)
        t1.record()
        torch.cuda.synchronize()
        print(f"proposal net {i}, {R}x{S} samples, position gradient {pos}: {t0.elapsed_time(t1) / 10 * 1e3:.1f} us")
