"""k_hash_encode alone at the training shape (4096 rays x 48 samples, 16 levels, T = 2^19), with and without the
Jacobian, on uniformly spaced samples of random rays."""
import sys

import torch

sys.path.insert(0, '/root/repo')
from fruitnerf_amd import _kernels as K   # noqa: E402
from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig   # noqa: E402
from fruitnerf_amd.data.semantics import apple_metadata   # noqa: E402

dev = torch.device('cuda:0')
m = FruitModel(FruitNerfModelConfig(), apple_metadata(), num_train_data=10, device=dev)
m.train()
R, S = 4096, 48
o = torch.randn(R, 3, device=dev) * 0.3
d = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
rays = K.RaysArg(o, d, torch.full((R,), 0.05, device=dev), torch.full((R,), 4.0, device=dev), None)
sp, eu = K.sample_spaced(rays, 1, S, None)
grid, warp = m.field.net_struct().grid, m.field.warp_struct()
for jac in (False, True):
    for _ in range(5):
        K.hash_encode_fwd(grid, warp, rays, eu, S, jac)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        K.hash_encode_fwd(grid, warp, rays, eu, S, jac)
    t1.record()
    torch.cuda.synchronize()
    print(f"hash_encode_fwd jacobian={jac}: {t0.elapsed_time(t1) / 20 * 1e3:.1f} us")
