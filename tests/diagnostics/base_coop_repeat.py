"""The field-MLP backward with the hash grid's input gradient (fnr_field_mlp_bwd_rays: k_field_mlp_bwd_base_coop<.., POSGRAD>)
called REPEATEDLY on identical inputs: d_position and d_feats must come out bit-identical every time.  Round 6's hunt: with
`nt` loads of the Jacobian they do not — which entries differ (lane group, wave, batch), and by how much?
usage: FNR_LIB_PATH=<variant .so> python tests/diagnostics/base_coop_repeat.py [repeats = 20] [rays = 4096]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    from fruitnerf_amd import _kernels as K
    from fruitnerf_amd.data.semantics import apple_metadata
    from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
    from fruitnerf_amd.rays import RayBundle
    from tests import util
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    hm = FruitModel(FruitNerfModelConfig(mlp_precision="bf16x3"), apple_metadata(), num_train_data=40, device=dev)
    hm.train()
    with torch.no_grad():
        hm.field.mlp_base_grid.hash_table.mul_(300.0)          # features of O(0.3): gradients of a visible size
    o, d, _, cam = util.random_rays(R, 40, seed=3)
    rb = hm._collide(RayBundle(o.to(dev), d.to(dev), None, cam.to(dev)))
    with torch.no_grad():
        outputs, rctx = hm._render(rb, None, save_input_jacobian=True)
    rays, fin = rctx.rays, rctx.levels[-1]
    S = fin["S"]
    N = rays.n * S
    g = torch.Generator(device=dev).manual_seed(1)
    d_density = torch.randn(N, device=dev, generator=g) * 1e-3
    d_rgb = torch.randn(N, 3, device=dev, generator=g) * 1e-3
    d_logit = torch.randn(N, device=dev, generator=g) * 1e-3
    fld = hm.field
    hm.arena()
    net, gnet = fld.net_struct(), fld.net_struct(grads=True)
    ref = None
    print("lib", os.environ.get("FNR_LIB_PATH", "default"), "N", N)
    for r in range(reps):
        d_feats, d_pos = K.field_mlp_bwd(net, gnet, rays, S, rctx.field_feats, rctx.field_h, rctx.field_selector, d_density,
                                         d_rgb, d_logit, jacobian=rctx.field_jacobian)
        torch.cuda.synchronize()
        # what d_position must be, from the kernel's own d_feats and the Jacobian: per level l and axis a,
        # c[l, a, n] = d_feats[l, n, :] . jac[l, a, n, :]; d_position[n, a] = sum_l c[l, a, n] (selector-free: the encode zeroes it)
        c = (d_feats[:, None, :, :] * rctx.field_jacobian).sum(dim=-1).double()            # [L, 3, N]
        want = c.sum(dim=0).t()                                                             # [N, 3]
        err = (d_pos[:, :3].double() - want).abs().max(dim=1).values
        tol = 1e-4 * want.abs().max()
        wrong = (err > tol).nonzero()[:, 0]
        if ref is None:
            ref = (d_feats.clone(), d_pos.clone())
            print("d_pos |max|", float(d_pos.abs().max()), "finite", bool(torch.isfinite(d_pos).all()))
        if len(wrong):
            i = wrong[0]
            res = (d_pos[i, :3].double() - want[i])                                           # what is missing / extra
            by_m = torch.stack([c[4 * m:4 * m + 4, :, i].sum(dim=0) for m in range(4)])       # level groups of a load round
            by_g = torch.stack([c[g::4, :, i].sum(dim=0) for g in range(4)])                  # levels of one lane group
            print(f"call {r}: {len(wrong)} samples off the recomputed value (waves {sorted(set(((wrong // 16) % 8).tolist()))}); sample {int(i)}: "
                  f"got {d_pos[i, :3].tolist()} want {want[i].tolist()}\n    residual {res.tolist()}\n    -(levels 4m..4m+3), m = 0..3: "
                  f"{(-by_m).tolist()}\n    -(levels g, 4+g, 8+g, 12+g), g = 0..3: {(-by_g).tolist()}")
            # is the wrong value the right value of ANOTHER sample?
            near = ((want - d_pos[i, :3].double()).abs().max(dim=1).values < tol).nonzero()[:, 0]
            print("    samples whose correct value this is:", near[:8].tolist())
        else:
            print(f"call {r}: every d_position equals its recomputed value")
        if r == 0:
            continue
        bad_f = int((d_feats != ref[0]).sum())
        diff = (d_pos != ref[1]).any(dim=1)
        nbad = int(diff.sum())
        if nbad == 0 and bad_f == 0:
            print(f"repeat {r}: identical")
            continue
        idx = diff.nonzero()[:, 0]
        j, wave, batch = idx % 16, (idx // 16) % 8, idx // 128
        rel = ((d_pos[idx] - ref[1][idx]).abs().max(dim=1).values / ref[1][idx].abs().max(dim=1).values.clamp_min(1e-30))
        print(f"repeat {r}: d_feats entries differing {bad_f}; d_pos samples differing {nbad} of {N}; "
              f"waves {sorted(set(wave.tolist()))[:8]} j {sorted(set(j.tolist()))[:16]} batches {len(set(batch.tolist()))} "
              f"(first {sorted(set(batch.tolist()))[:6]}); rel. size median {float(rel.median()):.2e} max {float(rel.max()):.2e}; "
              f"first samples {idx[:6].tolist()}")


if __name__ == "__main__":
    main()
