"""What does the table scatter of one step cost the NEXT step's encode?  A loop of k_hash_encode (with the Jacobian) followed by
the fused scatter (emit + accumulate with the table's Adam step) at the training shape, next to the encode alone — run it under
rocprofv3 (tools/gpu_call.sh ktpy) and read k_hash_encode's median; with FNR_LIB_PATH pointing at a build whose emit kernel never
writes its queue (a three-line patch of hash_scatter.hip's copy-out: the two queue stores behind a condition that is never true)
the difference between the two libraries is what the record queue's stores do to the caches.  (No training here: garbage
gradients cannot feed back into the positions or the timing.)
Round 6, call 45 (profiles/r06_raw/queue_store_premise.log): k_hash_encode median 74.9 us with the queue stores, 72.6 us without;
alone (no scatter between the calls) 75.3 us per call.  The record queue's stores are NOT what the encode's misses come from."""
import sys

import torch

sys.path.insert(0, '/root/repo')
from fruitnerf_amd import _kernels as K   # noqa: E402
from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig   # noqa: E402
from fruitnerf_amd.data.semantics import apple_metadata   # noqa: E402
from fruitnerf_amd.training import FusedAdam   # noqa: E402

dev = torch.device('cuda:0')
m = FruitModel(FruitNerfModelConfig(), apple_metadata(), num_train_data=10, device=dev)
m.train()
m.arena()
opt = FusedAdam(m)
R, S = 4096, 48
g = torch.Generator(device='cpu').manual_seed(0)
fld = m.field
grid, ggrid, warp = fld.net_struct().grid, fld.net_struct(grads=True).grid, fld.warp_struct()
d_feats = (torch.randn(16, R * S, 2, generator=g) * 1e-4).to(dev)


def rays_of(seed):
    gg = torch.Generator(device='cpu').manual_seed(seed)
    o = (torch.randn(R, 3, generator=gg) * 0.3).to(dev)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=gg), dim=-1).to(dev)
    rays = K.RaysArg(o, d, torch.full((R,), 0.05, device=dev), torch.full((R,), 4.0, device=dev), None)
    _, eu = K.sample_spaced(rays, 1, S, None)
    return rays, eu


batches = [rays_of(s) for s in range(8)]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
# 1. encode alone (fresh rays every call: the table is re-read from wherever it lives)
for i in range(40):
    if i == 8: ev[0].record()
    rays, eu = batches[i % 8]
    K.hash_encode_fwd(grid, warp, rays, eu, S, True)
ev[1].record()
# 2. encode + scatter (+ the table's optimiser step), as a training step has them
for i in range(40):
    if i == 8: ev[2].record()
    rays, eu = batches[i % 8]
    K.hash_encode_fwd(grid, warp, rays, eu, S, True)
    opt.begin_step() if hasattr(opt, "begin_step") else None
    adam, _ = opt.table_adam_args(fld.mlp_base_grid.hash_table)
    K.hash_encode_bwd_adam(ggrid, warp, rays, eu, S, d_feats, adam)
ev[3].record()
torch.cuda.synchronize()
print(f"encode alone: {ev[0].elapsed_time(ev[1]) / 32 * 1e3:.1f} us per call; encode + scatter: {ev[2].elapsed_time(ev[3]) / 32 * 1e3:.1f} us per pair")
