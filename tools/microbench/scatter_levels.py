"""Per-level cost of the main-field scatter: hash_encode_bwd restricted to one level at a time (and to level groups),
uniformly spaced samples on random rays through the scene box.  Shows which levels the call's time goes to."""
import sys

import torch

sys.path.insert(0, '/root/repo')
from fruitnerf_amd import _kernels as K   # noqa: E402
from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig   # noqa: E402
from fruitnerf_amd.data.semantics import apple_metadata   # noqa: E402

dev = torch.device('cuda:0')
m = FruitModel(FruitNerfModelConfig(), apple_metadata(), num_train_data=10, device=dev)
m.train()
m.arena()
R, S = 4096, 48
o = torch.randn(R, 3, device=dev) * 0.3
d = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
rays = K.RaysArg(o, d, torch.full((R,), 0.05, device=dev), torch.full((R,), 4.0, device=dev), None)
sp, eu = K.sample_spaced(rays, 1, S, None)
grid, warp = m.field.net_struct(grads=True).grid, m.field.warp_struct()
d_feats = torch.randn(16, R * S, 2, device=dev) * 1e-3


def timed(l0, n, reps=10):
    for _ in range(3):
        K.hash_encode_bwd(grid, warp, rays, eu, S, d_feats, l0, n)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        K.hash_encode_bwd(grid, warp, rays, eu, S, d_feats, l0, n)
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps * 1e3


print("all 16 levels: %.1f us" % timed(0, 16))
for l in range(16):
    print("level %2d alone: %6.1f us" % (l, timed(l, 1)))
for l0, n in ((0, 4), (4, 4), (8, 4), (12, 4), (0, 8), (8, 8), (4, 12), (5, 11)):
    print("levels %2d..%2d: %6.1f us" % (l0, l0 + n - 1, timed(l0, n)))
