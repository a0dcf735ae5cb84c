"""The bf16 matrix-pipe modes of the field MLPs (include/fruitnerf_hip.h: FNR_MLP_BF16X3 / FNR_MLP_BF16,
csrc/field_bf16.hpp) on the GPU.

bf16x3 is parity grade and FruitField's default ("auto"): the oracle-parity tests of the default path are re-run
here with the field switched explicitly to fp32 (exact fmaf chains) and to bf16x3 (FNR_MLP_PRECISION is read by
FruitField at construction), at the SAME tolerances.
bf16 (plain) is a throughput mode: it is compared with the fp32 HIP path at bf16-sized tolerances."""
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def _rerun(monkeypatch, fn, *args):
    """Re-run an oracle-parity test of the default path with the field MLPs in BOTH parity-grade arithmetics: whichever
    of them is FruitField's default ("auto"), the other one's kernels are covered here at the same tolerances."""
    for precision in ("fp32", "bf16x3"):
        monkeypatch.setenv("FNR_MLP_PRECISION", precision)
        fn(*args)


@pytest.mark.parametrize("training", [False, True])
def test_bf16x3_field_forward_per_sample(dev, monkeypatch, training):
    from tests import test_gpu_forward_parity as t
    _rerun(monkeypatch, t.test_field_forward_per_sample, dev, training, "fruit_nerf")


@pytest.mark.parametrize("training", [False, True])
def test_bf16x3_model_forward_end_to_end(dev, monkeypatch, training):
    from tests import test_gpu_forward_parity as t
    _rerun(monkeypatch, t.test_model_forward_end_to_end, dev, training)


@pytest.mark.parametrize("step,n_samples", [(0, 48), (12, 48), (0, 40)])
def test_bf16x3_losses_and_all_gradients(dev, monkeypatch, step, n_samples):
    from tests import test_gpu_training_parity as t
    _rerun(monkeypatch, t.test_losses_and_all_gradients, dev, step, n_samples, "fruit_nerf")


@pytest.mark.parametrize("step,n_samples", [(0, 128), (12, 40)])
def test_bf16x3_big_shape_losses_and_all_gradients(dev, monkeypatch, step, n_samples):
    """fruit_nerf_big: the bf16x3 mode runs the semantic branch's backward (30 -> 128 -> 128 -> 64 -> head) as the
    weight-streamed cooperative-dW kernel on the bf16 pipe; forward, colour and base stay on fp32 MFMA."""
    from tests import test_gpu_training_parity as t
    _rerun(monkeypatch, t.test_losses_and_all_gradients, dev, step, n_samples, "fruit_nerf_big")


def test_bf16x3_big_shape_fused_step_matches_the_autograd_step(dev, monkeypatch):
    from tests import test_gpu_training_parity as t
    _rerun(monkeypatch, t.test_fused_step_matches_the_autograd_step, dev, "fruit_nerf_big")


def test_bf16x3_three_training_steps_track_the_oracle(dev, monkeypatch):
    from tests import test_gpu_training_parity as t
    _rerun(monkeypatch, t.test_three_training_steps_track_the_oracle, dev)


def test_bf16x3_fused_step_matches_the_autograd_step(dev, monkeypatch):
    from tests import test_gpu_training_parity as t
    _rerun(monkeypatch, t.test_fused_step_matches_the_autograd_step, dev, "fruit_nerf")


def test_bf16x3_step_at_a_trained_state_matches_the_oracle(dev, monkeypatch):
    from tests import test_gpu_training_parity as t
    _rerun(monkeypatch, t.test_step_at_a_trained_state_matches_the_oracle, dev)


def test_bf16x3_export_at_a_trained_state_matches_the_oracle(dev, monkeypatch):
    from tests import test_gpu_training_parity as t
    _rerun(monkeypatch, t.test_export_at_a_trained_state_matches_the_oracle, dev)


def test_bf16x3_model_matches_the_reference_model(dev, monkeypatch):
    from tests import test_gpu_reference_pins as t
    _rerun(monkeypatch, t.test_hip_model_matches_the_reference_model, dev)


def _field_pair(dev, precision, seed=2, shape="fruit_nerf"):
    cfg = util.small_config(log2=16) if shape == "fruit_nerf" else util.big_config(log2=16)
    om = util.make_oracle(cfg, seed=seed)
    ref = util.make_hip_like(om, dev)
    alt = util.make_hip_like(om, dev)
    ref.field.mlp_precision = "fp32"
    alt.field.mlp_precision = precision
    return ref, alt


def _samples(dev, R=128, S=48, seed=7):
    from fruitnerf_amd.rays import RayBundle
    o, d, pa, cam = util.random_rays(R, 7, seed=seed)
    g = torch.Generator().manual_seed(seed)
    euclid = torch.sort(torch.rand(R, S + 1, generator=g) * 1.6 + 0.2, dim=-1).values
    hb = RayBundle(o.to(dev), d.to(dev), pa.to(dev), cam.to(dev))
    return hb.get_ray_samples(euclid[:, :-1, None].to(dev), euclid[:, 1:, None].to(dev))


@pytest.mark.parametrize("precision,tol_rgb,tol_logit,tol_dens", [("bf16x3", 5e-6, 2e-5, 2e-5), ("bf16", 3e-2, 2e-1, 1e-1)])
@pytest.mark.parametrize("training,shape", [(False, "fruit_nerf"), (True, "fruit_nerf"), (True, "fruit_nerf_big")])
def test_bf16_modes_vs_the_fp32_kernels_forward(dev, precision, tol_rgb, tol_logit, tol_dens, training, shape):
    """Same weights, same samples: per-sample rgb / logit / density of the bf16-pipe kernels vs the exact fp32 MFMA
    kernels.  bf16x3 agrees to fp32 rounding; plain bf16 to ~2^-8 of the activations."""
    from fruitnerf_amd.fruit_field import FieldHeadNames as H
    ref, alt = _field_pair(dev, precision, shape=shape)
    ref.train(training)
    alt.train(training)
    rs = _samples(dev)
    a, b = ref.field(rs), alt.field(rs)
    e_rgb = float((a[H.RGB] - b[H.RGB]).abs().max())
    e_log = float((a[H.SEMANTICS] - b[H.SEMANTICS]).abs().max())
    e_den = float(((a[H.DENSITY] - b[H.DENSITY]).abs() / a[H.DENSITY].abs().clamp_min(1e-3)).max())
    print(f"[{precision} train={training}] rgb {e_rgb:.3e} logit {e_log:.3e} density(rel) {e_den:.3e} "
          f"(|logit| max {float(a[H.SEMANTICS].abs().max()):.2f})")
    assert e_rgb <= tol_rgb and e_log <= tol_logit and e_den <= tol_dens
    if precision == "bf16":   # the mode really is a different arithmetic (fruit_nerf_big: only its semantic branch)
        assert (e_rgb if shape == "fruit_nerf" else e_log) > 1e-6


@pytest.mark.parametrize("precision,tol", [("bf16x3", 5e-4), ("bf16", 0.2)])
@pytest.mark.parametrize("S,shape", [(48, "fruit_nerf"), (40, "fruit_nerf"), (24, "fruit_nerf"), (8, "fruit_nerf"), (1, "fruit_nerf"),
                                     (48, "fruit_nerf_big"), (37, "fruit_nerf_big")])
def test_bf16_modes_vs_the_fp32_kernels_backward(dev, precision, tol, S, shape):
    """d_feats and every MLP / embedding gradient of fnr_field_mlp_bwd in the bf16-pipe modes vs the fp32 kernels
    (S = 40, 24: tiles straddle two rays; S = 8, 1 — the plugin API's per-sample queries — a tile holds many).  bf16x3 (three piece products in the backward pass): max error relative to each
    tensor's max |g| within the 5e-4 bar of the oracle tests (measured 5e-6 .. 2e-4).  Plain bf16: L2-relative error —
    the random zero-mean upstream gradients of this test make the weight gradients sums with ~100x cancellation and
    bf16-sized pre-activation errors flip ReLU gates, so max-norm errors of single elements reach 10-25 %."""
    from fruitnerf_amd import _kernels as K
    ref, alt = _field_pair(dev, precision, seed=4, shape=shape)
    R = 96
    N = R * S
    o, d, pa, cam = util.random_rays(R, 7, seed=9)
    g = torch.Generator().manual_seed(1)
    euclid = torch.sort(torch.rand(R, S + 1, generator=g) * 1.6 + 0.2, dim=-1).values.to(dev).contiguous()
    d_density = (torch.randn(N, generator=g) * 1e-2).to(dev)
    d_rgb = torch.randn(N, 3, generator=g).to(dev) * 1e-1
    d_logit = torch.randn(N, generator=g).to(dev) * 1e-1
    outs = []
    for m in (ref, alt):
        m.train()
        m.arena()
        fld = m.field
        m.arena().grads.zero_()
        net, gnet = fld.net_struct(), fld.net_struct(grads=True)
        rays = K.RaysArg(o.to(dev), d.to(dev), None, None, cam.to(dev))
        feats, selector = K.hash_encode_fwd(net.grid, fld.warp_struct(), rays, euclid, S)
        dens, rgb, logit, _, saved = K.field_mlp_fwd(net, rays, S, feats, selector, None, want_h=True)
        d_feats = K.field_mlp_bwd(net, gnet, rays, S, feats, saved, selector, d_density, d_rgb, d_logit)
        torch.cuda.synchronize()
        grads = {n: p.grad.detach().clone() for n, p in fld.named_parameters()
                 if p.grad is not None and "hash_table" not in n}
        outs.append((d_feats.clone(), grads))
    (df0, g0), (df1, g1) = outs
    def err(a, b):
        if precision == "bf16":
            return float((a - b).double().norm() / a.double().norm())
        return float((a - b).abs().max() / a.abs().max())
    worst = err(df0, df1)
    print(f"[{precision} S={S}] d_feats err {worst:.3e}")
    for n in g0:
        scale = float(g0[n].abs().max())
        if scale == 0.0:
            assert float(g1[n].abs().max()) == 0.0, n
            continue
        rel = err(g0[n], g1[n])
        print(f"[{precision} S={S}] {n}: max|g| {scale:.3e} err {rel:.3e}")
        worst = max(worst, rel)
    assert worst <= tol


def test_transposing_lds_read_returns_what_the_per_wave_backward_assumes(dev, tmp_path):
    """tools/microbench/lds_tr16_transpose.hip on this box: ds_read_b64_tr_b16 over the shipped chunk order hands lane t of
    16-lane group g feature t of samples 4g .. 4g+3 (the dW operands of field_mlp_bwd_pw.hip are built on it)."""
    import os
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("hipcc not available")
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "microbench", "lds_tr16_transpose.hip")
    exe = tmp_path / "lds_tr16_transpose"
    subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", src, "-o", str(exe)], check=True, capture_output=True, timeout=600)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "OK: 0 mismatches of 512" in out.stdout, out.stdout[-400:]


@pytest.mark.parametrize("shape", ["fruit_nerf", "fruit_nerf_big"])
def test_per_wave_backward_is_reproducible_and_its_position_gradient_is_its_own_contraction(dev, shape):
    """The MLP backward with the hash grid's input gradient (fnr_field_mlp_bwd_rays -> k_field_mlp_bwd_base_pw<.., POSGRAD>) at the
    training size, called repeatedly on identical inputs: d_feats and d_position bit-identical call to call (round 5's
    irreproducibility sat in exactly this reduction of the cooperative kernel), and every d_position equal to the contraction
    of the kernel's own d_feats with the encode's Jacobian recomputed in float64."""
    from fruitnerf_amd import _kernels as K
    from fruitnerf_amd.data.semantics import apple_metadata
    from fruitnerf_amd.fruit_nerf import FruitModel
    from fruitnerf_amd.fruit_nerf_config import model_config
    from fruitnerf_amd.rays import RayBundle
    cfg = model_config(shape, mlp_precision="bf16x3")
    torch.manual_seed(0)
    hm = FruitModel(cfg, apple_metadata(), num_train_data=40, device=dev)
    hm.train()
    with torch.no_grad():
        hm.field.mlp_base_grid.hash_table.mul_(300.0)          # features of O(0.3): gradients of a visible size
    R = 4096 if shape == "fruit_nerf" else 1024
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
    for r in range(6):
        d_feats, d_pos = K.field_mlp_bwd(net, gnet, rays, S, rctx.field_feats, rctx.field_h, rctx.field_selector, d_density,
                                         d_rgb, d_logit, jacobian=rctx.field_jacobian)
        torch.cuda.synchronize()
        if ref is None:
            ref = (d_feats.clone(), d_pos.clone())
            want = (d_feats[:, None, :, :] * rctx.field_jacobian).sum(dim=-1).double().sum(dim=0).t()      # [N, 3]
            err = (d_pos[:, :3].double() - want).abs().max()
            assert float(want.abs().max()) > 0 and float(err) <= 1e-4 * float(want.abs().max()), (float(err), float(want.abs().max()))
        else:
            assert torch.equal(d_feats, ref[0]), f"call {r}: {int((d_feats != ref[0]).sum())} d_feats entries differ"
            assert torch.equal(d_pos, ref[1]), f"call {r}: {int((d_pos != ref[1]).any(dim=1).sum())} d_position rows differ"
