"""Phase breakdown of k_scatter_emit from the debug build (make -C fruitnerf_amd/csrc EXTRA=-DFNR_EMIT_TIMING
OUT=../lib/libfruitnerf_hip_dbg.so OBJDIR=../../build/obj_dbg; run with FNR_LIB_PATH pointing at it).
Prints the share of wave-0 shader clocks per phase for the main-field scatter and the two proposal-net scatters."""
import ctypes as C
import sys

import torch

sys.path.insert(0, '/root/repo')
from fruitnerf_amd import _lib as L, _kernels as K   # noqa: E402
from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig   # noqa: E402
from fruitnerf_amd.data.semantics import apple_metadata   # noqa: E402

dev = torch.device('cuda:0')
m = FruitModel(FruitNerfModelConfig(), apple_metadata(), num_train_data=10, device=dev)
m.train()
m.arena()
lib = L.load()
lib.fnr_debug_emit_phases.restype = C.c_int
NAMES = ["init", "load+corners+scan+count", "max+scan1", "scan2+reserve(global atomics)", "place in LDS", "copy out",
         "  (of phase 1) until loads + warp done", "  (of phase 1) until run sums done"]


def phases(reset=True):
    buf = (C.c_ulonglong * 8)()
    assert lib.fnr_debug_emit_phases(buf, 1 if reset else 0) == 0
    return list(buf)


R = 4096
o = torch.randn(R, 3, device=dev) * 0.3
d = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
rays = K.RaysArg(o, d, torch.full((R,), 0.05, device=dev), torch.full((R,), 4.0, device=dev), None)
for name, S, grid_of, levels in (("main field 4096x48, 16 levels", 48, lambda: m.field.net_struct(grads=True).grid, 16),
                                 ("proposal net 0, 4096x256, 5 levels", 256, None, 5),
                                 ("proposal net 1, 4096x96, 5 levels", 96, None, 5)):
    sp, eu = K.sample_spaced(rays, 1, S, None)
    N = R * S
    if grid_of is not None:
        grid, warp = grid_of(), m.field.warp_struct()
    else:
        pn = m.proposal_networks[0 if S == 256 else 1]
        grid, warp = pn.prop_struct(grads=True).grid, pn.warp_struct()
    d_feats = torch.randn(levels, N, 2, device=dev) * 1e-3
    for _ in range(3):
        K.hash_encode_bwd(grid, warp, rays, eu, S, d_feats)
    torch.cuda.synchronize()
    phases(reset=True)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10):
        K.hash_encode_bwd(grid, warp, rays, eu, S, d_feats)
    t1.record()
    torch.cuda.synchronize()
    p = phases()
    tot = sum(p[:6])
    print(f"{name}: {t0.elapsed_time(t1) / 10 * 1e3:.1f} us per call (emit + accumulate); emit phases:")
    for n, v in zip(NAMES, p):
        print(f"    {n:34s} {100.0 * v / tot:5.1f} %   {v / 10 / ((N + 511) // 512 * levels):9.0f} clk / workgroup")
