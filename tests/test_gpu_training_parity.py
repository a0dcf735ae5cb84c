"""GPU parity of the training path: losses, every parameter gradient, the Adam update and a short
multi-step run, HIP (through the C ABI) vs the CPU oracle's autograd."""
import copy

import pytest
import torch

from oracle import fruit_oracle as fo
from oracle import ns_torch as ns
from tests import util

pytestmark = pytest.mark.gpu


def _batch(R, seed):
    g = torch.Generator().manual_seed(seed)
    return {"image": torch.rand(R, 3, generator=g), "fruit_mask": (torch.rand(R, 1, generator=g) > 0.6).float()}


def _oracle_step(om, o, d, pa, cam, jit, batch, step):
    om.set_anneal(step)
    out = om(ns.RayBundle(o.clone(), d.clone(), pa.clone(), camera_indices=cam.clone()), jitter=jit)
    md = om.get_metrics_dict(out, batch)
    ld = om.get_loss_dict(out, batch)
    loss = sum(ld.values())
    loss.backward()
    return out, ld, md


def _grad_report(om, hm, tag="", with_aggregate=False):
    """Worst max-norm relative gradient error over the parameters (and, with_aggregate, the worst L1-relative error:
    sum|hip - ref| / sum|ref| per parameter)."""
    worst, worst_agg = 0.0, 0.0
    named_h = dict(hm.named_parameters())
    for name, p in util.named_trainable(om):
        g_ref = p.grad if p.grad is not None else torch.zeros_like(p)
        g_hip = named_h[name].grad.detach().cpu()
        scale = g_ref.abs().max().item()
        diff = (g_hip - g_ref).abs()
        err = diff.max().item()
        rel = err / max(scale, 1e-12)
        agg = diff.double().sum().item() / max(g_ref.abs().double().sum().item(), 1e-30)
        nnz_ref = int((g_ref != 0).sum())
        nnz_hip = int((g_hip != 0).sum())
        print(f"[grad{tag}] {name}: max|ref| {scale:.3e} max_err {err:.3e} rel {rel:.3e} L1-rel {agg:.3e} "
              f"nnz ref/hip {nnz_ref}/{nnz_hip}")
        if scale > 0:
            worst = max(worst, rel)
            worst_agg = max(worst_agg, agg)
        else:
            assert err == 0.0, f"{name}: oracle grad is zero but HIP grad is not"
    return (worst, worst_agg) if with_aggregate else worst


@pytest.mark.parametrize("step,n_samples,shape", [(0, 48, "fruit_nerf"), (12, 48, "fruit_nerf"), (0, 40, "fruit_nerf"),
                                                  (0, 64, "fruit_nerf_huge"),
                                                  (0, 128, "fruit_nerf_big"), (12, 40, "fruit_nerf_big")])
def test_losses_and_all_gradients(dev, step, n_samples, shape):
    """step 0: proposal nets are 'updated' (interlevel gradient flows); step 12 with a fresh sampler
    state: not updated -> proposal-network gradients must be exactly zero on both sides.
    40 samples per ray: 16-sample MFMA tiles straddle rays (per-ray colour terms take their slow path).
    fruit_nerf_big: the second built MLP shape (geo 30, semantic 30 -> 128 -> 128 -> 64; FieldCfgBig).
    fruit_nerf_huge: that field with the huge method's proposal networks (5 levels -> 512, 7 levels -> 2048)."""
    from fruitnerf_amd.rays import RayBundle
    cfg = {"fruit_nerf": util.small_config, "fruit_nerf_big": util.big_config,
           "fruit_nerf_huge": util.huge_config}[shape](log2=15, prop_log2=13)
    cfg.num_nerf_samples_per_ray = n_samples
    # this test is about the MLP shape: keep the sampler in the regime of the fruit_nerf cases (an anneal exponent of
    # 0.02 at step 12 of 5000 makes the PDF sampler's inverse CDF ill-conditioned, and max_res 4096 on a RANDOM table
    # turns 1e-6 of sample position into 4e-3 of a finest-level cell)
    cfg.proposal_weights_anneal_max_num_iters, cfg.max_res = 1000, 2048
    om = util.make_oracle(cfg, seed=5)
    hm = util.make_hip_like(om, dev)
    om.train()
    hm.train()
    for m in (om, hm):
        m.proposal_sampler._step = step
        m.proposal_sampler._steps_since_update = 0
    R = 160
    o, d, pa, cam = util.random_rays(R, 7, seed=21)
    jit = [torch.rand(R, 1) for _ in range(3)]
    batch = _batch(R, 3)
    out, ld_ref, md_ref = _oracle_step(om, o, d, pa, cam, jit, batch, step)

    hm.set_anneal(step)
    hout = hm(RayBundle(o.to(dev), d.to(dev), pa.to(dev), cam.to(dev)), jitter=[j.to(dev) for j in jit])
    hb = {k: v.to(dev) for k, v in batch.items()}
    md = hm.get_metrics_dict(hout, hb)
    ld = hm.get_loss_dict(hout, hb)
    sum(ld.values()).backward()
    torch.cuda.synchronize()
    for k in ld_ref:
        a, b = float(ld[k]), float(ld_ref[k])
        print(f"[loss step={step}] {k}: hip {a:.8e} oracle {b:.8e}")
        # interlevel: a mean of squared CLIPPED histogram overlaps (~1e-4 here), ill-conditioned in the sampled bin
        # edges (sampler parity 2e-6) when the anneal exponent is tiny (fruit_nerf_big anneals over 5000 steps): 1e-3
        tol = 1e-3 * abs(b) + 1e-8 if (k == "interlevel_loss" and shape != "fruit_nerf") else 2e-5 * max(abs(b), 1e-3)
        assert abs(a - b) <= tol, k
    for k in md_ref:
        a, b = float(md[k]), float(md_ref[k])
        print(f"[metric step={step}] {k}: hip {a:.6e} oracle {b:.6e}")
        assert abs(a - b) <= 1e-4 * max(abs(b), 1e-3), k
    worst, worst_agg = _grad_report(om, hm, f" step={step}", with_aggregate=True)
    if shape == "fruit_nerf":
        assert worst <= 2e-3, f"worst relative gradient error {worst}"
    else:
        # 20 480 samples x 448 hidden units: a ReLU pre-activation within fp32 rounding of 0 is expected for ~1 of them;
        # that unit takes the other branch than in the oracle (MFMA K-order) and ONE sample's contribution moves one
        # weight row and, through dX, that sample's 128 table rows — visible in the max norm (gradients ~1e-6), invisible
        # in L1.  A wrong kernel moves every entry: bound both.
        assert worst <= 2e-2 and worst_agg <= 2e-3, f"gradient error: max-norm {worst}, L1 {worst_agg}"


@pytest.mark.parametrize("shape,step,tables", [("fruit_nerf_big", 0, "white"), ("fruit_nerf_big", 2500, "smooth"),
                                               ("fruit_nerf_huge", 2500, "smooth"), ("fruit_nerf_huge", 0, "white")])
def test_losses_and_all_gradients_at_the_real_configuration(dev, shape, step, tables):
    """The gradient legs of `fruit_nerf_big` / `fruit_nerf_huge` WITHOUT the shrinking of the test above: the methods'
    own sizes (fruit_nerf_config.py:82-95 / 113-164 — T = 2^21, max_res 4096 / 8192, 512/256/128 and 512/512/64 samples,
    the 5- and 7-level proposal grids at T = 2^17, anneal over 5000 iterations).  step 0: anneal exponent 0 (flat
    proposal PDFs); step 2500: exponent 0.909.  Both are 'updated' steps (proposal networks get gradients).
    tables = "white": uniform random entries at every level — at max_res 4096 / 8192 the finest levels then encode
    noise with a slope of thousands per unit length, so the 1e-6 sampler noise is visible in single entries;
    "smooth": the same entries scaled by base_res / res_l per level (every level the same slope, as in a trained
    field), where the well-conditioned bar must hold."""
    from fruitnerf_amd.rays import RayBundle
    cfg = {"fruit_nerf_big": util.fruit_nerf_big_config, "fruit_nerf_huge": util.fruit_nerf_huge_config}[shape]()
    om = util.make_oracle(cfg, seed=5)
    if tables == "smooth":
        util.smooth_tables_(om)
    hm = util.make_hip_like(om, dev)
    om.train()
    hm.train()
    for m in (om, hm):
        m.proposal_sampler._step = step
        m.proposal_sampler._steps_since_update = 100
    R = 96
    o, d, pa, cam = util.random_rays(R, 7, seed=21)
    jit = [torch.rand(R, 1) for _ in range(3)]
    batch = _batch(R, 3)
    out, ld_ref, md_ref = _oracle_step(om, o, d, pa, cam, jit, batch, step)
    hm.set_anneal(step)
    hout = hm(RayBundle(o.to(dev), d.to(dev), pa.to(dev), cam.to(dev)), jitter=[j.to(dev) for j in jit])
    hb = {k: v.to(dev) for k, v in batch.items()}
    md = hm.get_metrics_dict(hout, hb)
    ld = hm.get_loss_dict(hout, hb)
    sum(ld.values()).backward()
    torch.cuda.synchronize()
    for k in ld_ref:
        a, b = float(ld[k]), float(ld_ref[k])
        print(f"[real {shape} step={step} {tables}] {k}: hip {a:.8e} oracle {b:.8e} rel {abs(a - b) / max(abs(b), 1e-12):.2e}")
        tol = 1e-3 * abs(b) + 1e-8 if k == "interlevel_loss" else 1e-4 * max(abs(b), 1e-3)
        assert abs(a - b) <= tol, k
    for k in ("rgb", "semantics", "accumulation"):
        err = (hout[k].detach().cpu() - out[k].detach()).abs().max().item()
        print(f"[real {shape} step={step} {tables}] output {k}: max abs err {err:.3e}")
        assert err <= 1e-5, k            # measured 1e-6 (round 3): an order inside the 1e-4 bar
    worst, worst_agg = _grad_report(om, hm, f" real {shape} step={step} {tables}", with_aggregate=True)
    # measured (round 3, bf16x3 default): max-norm <= 1e-4, L1 <= 5e-5 on all four legs — the round-1 bar of 5e-4 holds
    # at the real sizes for white and smooth tables alike
    assert worst <= 5e-4 and worst_agg <= 2e-4, f"gradient error: max-norm {worst}, L1 {worst_agg}"


def test_adam_matches_torch(dev):
    from fruitnerf_amd import _kernels as K
    torch.manual_seed(0)
    n = 4096 * 4
    p0 = torch.randn(n)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-2, eps=1e-15)
    p = p0.clone().to(dev)
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    for step in range(1, 6):
        g = torch.randn(n) * (torch.rand(n) > 0.5)  # half the entries see zero gradients
        ref.grad = g.clone()
        opt.step()
        gd = (g * 2.0).to(dev)  # all-reduced SUM of 2 ranks -> scale 0.5
        K.adam_step(p, gd, m, v, 1e-2, 0.9, 0.999, 1e-15, step, grad_scale=0.5, zero_grad=True)
        assert float(gd.abs().max()) == 0.0
    a, _ = util.report("adam.params", p, ref.detach())
    assert a <= 2e-6


def test_three_training_steps_track_the_oracle(dev):
    """forward + backward + Adam for 3 steps from identical weights and identical jitter."""
    from fruitnerf_amd.rays import RayBundle
    from fruitnerf_amd.training import FusedAdam, train_iteration
    cfg = util.small_config(log2=14, prop_log2=12)
    om = util.make_oracle(cfg, seed=8)
    hm = util.make_hip_like(om, dev)
    om.train()
    hm.train()
    groups = om.get_param_groups()
    opts = [torch.optim.Adam(groups["proposal_networks"], lr=1e-2, eps=1e-15),
            torch.optim.Adam(groups["fields"], lr=1e-2, eps=1e-15)]
    hopt = FusedAdam(hm)
    p0 = {n: p.detach().clone() for n, p in util.named_trainable(om)}
    R = 128
    for step in range(3):
        o, d, pa, cam = util.random_rays(R, 7, seed=100 + step)
        jit = [torch.rand(R, 1) for _ in range(3)]
        batch = _batch(R, 50 + step)
        for op in opts:
            op.zero_grad()
        _, ld_ref, _ = _oracle_step(om, o, d, pa, cam, jit, batch, step)
        for op in opts:
            op.step()
        om.proposal_sampler.step_cb(step)
        ld, _ = train_iteration(hm, hopt, RayBundle(o.to(dev), d.to(dev), pa.to(dev), cam.to(dev)),
                                {k: v.to(dev) for k, v in batch.items()}, step, jitter=[j.to(dev) for j in jit])
        for k in ld_ref:
            a, b = float(ld[k]), float(ld_ref[k])
            print(f"[train step {step}] {k}: hip {a:.8e} oracle {b:.8e}")
            # Adam (eps=1e-15) turns every non-zero gradient into a +-lr step on the first iterations, so
            # entries whose gradient is rounding noise take different signs on the two sides: trajectories
            # agree to ~1e-3 (the tiny interlevel term to a few %), not to fp32 rounding.
            tol = 5e-2 if k == "interlevel_loss" else 2e-3
            assert abs(a - b) <= tol * max(abs(b), 1e-3), (step, k)
    torch.cuda.synchronize()
    named_h = dict(hm.named_parameters())
    for name, p in util.named_trainable(om):
        diff = (named_h[name].detach().cpu() - p.detach()).abs().max().item()
        print(f"[params after 3 steps] {name}: max_abs_diff {diff:.3e}")
        # Adam's first steps move every touched entry by ~lr regardless of gradient magnitude, so entries
        # whose gradient is rounding-level noise may differ by O(lr); bound the bulk instead of the max
        d_abs = (named_h[name].detach().cpu() - p.detach()).abs()
        moved = (p.detach() - p0[name]).abs()
        print(f"    median diff {d_abs.median().item():.3e}  mean diff {d_abs.mean().item():.3e}  "
              f"mean |update| {moved.mean().item():.3e}")
        # the bulk of the entries must agree far better than the size of the update itself
        assert d_abs.median().item() <= 0.05 * max(moved.mean().item(), 1e-9) + 1e-7, name
        assert d_abs.max().item() <= 3 * 3 * 1e-2 + 1e-6, name  # nothing moves further than steps * lr apart


@pytest.mark.parametrize("shape", ["fruit_nerf", "fruit_nerf_big"])
def test_field_api_is_differentiable(dev, shape):
    """FruitField.forward / get_density -> get_outputs in training mode carry autograd history w.r.t. the field's
    parameters (fruit_field.py:168-301 is an ordinary differentiable nn.Module in the reference): a loss on the
    per-sample density, rgb and semantics back-propagates into every field parameter like the oracle's autograd."""
    from fruitnerf_amd.fruit_field import FieldHeadNames
    from fruitnerf_amd.rays import RayBundle
    cfg = (util.small_config if shape == "fruit_nerf" else util.big_config)(log2=14)
    om = util.make_oracle(cfg, seed=41)
    hm = util.make_hip_like(om, dev)
    om.train()
    hm.train()
    R, S = 64, 24
    o, d, pa, cam = util.random_rays(R, 7, seed=5)
    euclid = torch.sort(torch.rand(R, S + 1) * 1.6 + 0.2, dim=-1).values
    g = torch.Generator().manual_seed(3)
    wd, wr, ws = torch.rand(R, S, 1, generator=g) * 1e-3, torch.randn(R, S, 3, generator=g), torch.randn(R, S, 1, generator=g)
    rs = ns.RayBundle(o, d, pa, camera_indices=cam).get_ray_samples(euclid[:, :-1, None], euclid[:, 1:, None])
    ref = om.field(rs)
    ((ref["density"] * wd).sum() + (ref["rgb"] * wr).sum() + (ref["semantics"] * ws).sum()).backward()
    hb = RayBundle(o.to(dev), d.to(dev), pa.to(dev), cam.to(dev))
    hs = hb.get_ray_samples(euclid[:, :-1, None].to(dev), euclid[:, 1:, None].to(dev))
    for two_calls in (False, True):
        hm.arena().grads.zero_()
        if two_calls:                                           # the Field base class's own forward()
            dens, emb = hm.field.get_density(hs)
            out = hm.field.get_outputs(hs, density_embedding=emb)
            rgb, sem = out[FieldHeadNames.RGB], out[FieldHeadNames.SEMANTICS]
            assert not emb.requires_grad
        else:
            out = hm.field(hs)
            dens, rgb, sem = out[FieldHeadNames.DENSITY], out[FieldHeadNames.RGB], out[FieldHeadNames.SEMANTICS]
        assert dens.requires_grad and rgb.requires_grad and sem.requires_grad
        ((dens * wd.to(dev)).sum() + (rgb * wr.to(dev)).sum() + (sem * ws.to(dev)).sum()).backward()
        torch.cuda.synchronize()
        named_h = dict(hm.field.named_parameters())
        for name, p in om.field.named_parameters():
            g_ref = p.grad if p.grad is not None else torch.zeros_like(p)
            diff = (named_h[name].grad.cpu() - g_ref).abs()
            scale = g_ref.abs().max().item()
            l1 = diff.double().sum().item() / max(g_ref.abs().double().sum().item(), 1e-30)
            print(f"[field api two_calls={two_calls}] {name}: max|ref| {scale:.3e} max_err {diff.max().item():.3e} L1-rel {l1:.3e}")
            # `fruit_nerf`: the round-1 bar (2e-3 of max |g|) in every arithmetic; the big shape's 448 hidden units x
            # 20 k samples on a white 2^14-row table produce the occasional ReLU gate on the other side of 0 (see the
            # big-shape note above): max-norm 2e-2 AND L1 2e-3 there
            tol = 2e-3 if shape == "fruit_nerf" else 2e-2
            assert diff.max().item() <= tol * scale + 1e-12 and l1 <= 2e-3, name
    # the side-effect attributes of get_density (fruit_field.py:180-186)
    loc, dba = hm.field._sample_locations, hm.field._density_before_activation
    assert loc.shape == (R, S, 3) and loc.requires_grad and dba.shape == (R, S, 1)
    with torch.no_grad():
        om.field.get_density(rs)
    assert (loc.detach().cpu() - om.field._sample_locations.detach()).abs().max().item() <= 1e-6
    assert (dba.cpu() - om.field._density_before_activation.detach()).abs().max().item() <= 1e-4
    # eval mode: the same API without autograd history
    hm.eval()
    assert not hm.field(hs)[FieldHeadNames.RGB].requires_grad


def test_trainer_shaped_loop_over_the_plugin_api(dev):
    """Nerfstudio's Trainer.train_iteration, spelled out over the plugin surface only (callbacks by location, forward,
    get_metrics_dict, get_loss_dict, reduce(add), backward, optimiser) — must leave the model exactly where
    fused_train_iteration() (what bench.py times) leaves its twin."""
    import functools
    from fruitnerf_amd.engine.callbacks import TrainingCallbackAttributes, TrainingCallbackLocation as Loc
    from fruitnerf_amd.rays import RayBundle
    from fruitnerf_amd.training import FusedAdam, fused_train_iteration, skipped_groups
    cfg = util.small_config(log2=13, prop_log2=11)
    om = util.make_oracle(cfg, seed=23)
    a, b = util.make_hip_like(om, dev), util.make_hip_like(om, dev)
    a.train()
    b.train()
    opt_a, opt_b = FusedAdam(a), FusedAdam(b)
    callbacks = a.get_training_callbacks(TrainingCallbackAttributes(optimizers=None, grad_scaler=None, pipeline=None))
    R = 128
    for step in range(12):                                       # crosses the end of the every-step update phase
        o, d, pa, cam = util.random_rays(R, 7, seed=500 + step)
        batch = {k: v.to(dev) for k, v in _batch(R, 90 + step).items()}
        for cb in callbacks:
            cb.run_callback_at_location(step, location=Loc.BEFORE_TRAIN_ITERATION)
        torch.manual_seed(1000 + step)                           # the model draws its jitter from the device generator
        outputs = a(RayBundle(o.to(dev), d.to(dev), pa.to(dev), cam.to(dev)))
        metrics_dict = a.get_metrics_dict(outputs, batch)
        loss_dict = a.get_loss_dict(outputs, batch, metrics_dict)
        functools.reduce(torch.add, loss_dict.values()).backward()
        opt_a.step(skip=skipped_groups(a))
        for cb in callbacks:
            cb.run_callback_at_location(step, location=Loc.AFTER_TRAIN_ITERATION)
        torch.manual_seed(1000 + step)
        ld_b, md_b = fused_train_iteration(b, opt_b, RayBundle(o.to(dev), d.to(dev), pa.to(dev), cam.to(dev)), batch, step)
        for k in loss_dict:
            # same kernels on both sides; only the order of a few float atomics differs, and Adam (eps 1e-15) turns
            # rounding-level gradient noise into +-lr steps, so the twins agree to ~1e-3, not to fp32 rounding
            tol = 5e-2 if k == "interlevel_loss" else 2e-3
            assert abs(float(loss_dict[k]) - float(ld_b[k])) <= tol * max(abs(float(ld_b[k])), 1e-6), (step, k)
    torch.cuda.synchronize()
    pa_, pb_ = a.arena().params, b.arena().params
    moved = (pb_ - util.make_hip_like(om, dev).arena().params).abs()
    assert (pa_ - pb_).abs().median().item() <= 0.05 * moved.mean().item() + 1e-7
    assert (pa_ - pb_).abs().max().item() <= 3 * 12 * 1e-2
    assert a.proposal_sampler._step == b.proposal_sampler._step == 11
    assert opt_a.group_steps == opt_b.group_steps


@pytest.mark.parametrize("fused", [False, True])
def test_optimizer_skips_the_proposal_networks_on_steps_that_do_not_update_them(dev, fused):
    """Steps 9..12 cross the end of the every-step phase (ProposalNetworkSampler: step < 10).  On iteration 11 the
    proposal densities are computed under no_grad, the reference's zero_grad() leaves those .grad = None and
    torch.optim skips the parameters: no movement, no moment decay, no step-count advance (bias correction).  The
    oracle side is driven with torch.optim exactly as nerfstudio's Optimizers does; the HIP side must show the same
    pattern and the same per-group step counts."""
    from fruitnerf_amd.rays import RayBundle
    from fruitnerf_amd.training import FusedAdam, fused_train_iteration, train_iteration
    cfg = util.small_config(log2=12, prop_log2=10)
    om = util.make_oracle(cfg, seed=31)
    hm = util.make_hip_like(om, dev)
    om.train()
    hm.train()
    groups = om.get_param_groups()
    opts = {"proposal_networks": torch.optim.Adam(groups["proposal_networks"], lr=1e-2, eps=1e-15),
            "fields": torch.optim.Adam(groups["fields"], lr=1e-2, eps=1e-15)}
    hopt = FusedAdam(hm)
    for m in (om, hm):                       # as after 9 warm-up iterations
        m.proposal_sampler._step = 8
        m.proposal_sampler._steps_since_update = 1
    hopt.step_count = 9
    hopt.group_steps = {k: 9 for k in hopt.group_steps}
    R = 96
    a, b = hm.arena().group_ranges["proposal_networks"]
    moved_ref, moved_hip = [], []
    for step in range(9, 13):
        o, d, pa, cam = util.random_rays(R, 7, seed=300 + step)
        jit = [torch.rand(R, 1) for _ in range(3)]
        batch = _batch(R, 70 + step)
        before_ref = [p.detach().clone() for p in groups["proposal_networks"]]
        for op in opts.values():
            op.zero_grad()                  # torch >= 2.0: set_to_none=True
        _oracle_step(om, o, d, pa, cam, jit, batch, step)
        for op in opts.values():
            op.step()
        om.proposal_sampler.step_cb(step)
        moved_ref.append(any(not torch.equal(x, p.detach()) for x, p in zip(before_ref, groups["proposal_networks"])))
        before = hm.arena().params[a:b].clone()
        m_before, v_before = hopt.exp_avg[a:b].clone(), hopt.exp_avg_sq[a:b].clone()
        fn = fused_train_iteration if fused else train_iteration
        fn(hm, hopt, RayBundle(o.to(dev), d.to(dev), pa.to(dev), cam.to(dev)), {k: v.to(dev) for k, v in batch.items()},
           step, jitter=[j.to(dev) for j in jit])
        torch.cuda.synchronize()
        moved_hip.append(not torch.equal(before, hm.arena().params[a:b]))
        if not moved_hip[-1]:
            assert torch.equal(m_before, hopt.exp_avg[a:b]) and torch.equal(v_before, hopt.exp_avg_sq[a:b])
        assert float(hm.arena().grads.abs().max()) == 0.0       # zero_grad fused into the step, skipped span included
    print("[optimizer skip] proposal networks moved on steps 9..12: oracle", moved_ref, "hip", moved_hip)
    # nerfstudio's rule: updated iff steps_since_update > update_sched(step) or step < 10, evaluated with the step
    # number the AFTER_TRAIN_ITERATION callback stored -> iterations 9 and 10 still update, 11 does not, 12 does
    assert moved_ref == [True, True, False, True] and moved_hip == moved_ref
    ref_steps = {int(opts["proposal_networks"].state[p]["step"]) for p in groups["proposal_networks"]}
    assert ref_steps == {3} and hopt.group_steps["proposal_networks"] == 9 + 3
    assert hopt.group_steps["fields"] == 9 + 4 and hopt.step_count == 9 + 4


def test_optimizer_can_step_groups_without_gradient_like_torch_1_13(dev):
    """FusedAdam(skip_groups_without_grad=False): nerfstudio 0.3.2 also runs on torch 1.13, whose zero_grad() leaves ZERO
    tensors — Adam then still decays the moments, moves the proposal networks along them and advances their step count
    on the iterations that evaluate them under no_grad.  Oracle: torch.optim fed explicit zero gradients."""
    from fruitnerf_amd.rays import RayBundle
    from fruitnerf_amd.training import FusedAdam, fused_train_iteration
    cfg = util.small_config(log2=12, prop_log2=10)
    om = util.make_oracle(cfg, seed=31)
    hm = util.make_hip_like(om, dev)
    om.train()
    hm.train()
    groups = om.get_param_groups()
    opts = {"proposal_networks": torch.optim.Adam(groups["proposal_networks"], lr=1e-2, eps=1e-15),
            "fields": torch.optim.Adam(groups["fields"], lr=1e-2, eps=1e-15)}
    hopt = FusedAdam(hm, skip_groups_without_grad=False)
    for m in (om, hm):
        m.proposal_sampler._step = 8
        m.proposal_sampler._steps_since_update = 1
    R = 96
    a, b = hm.arena().group_ranges["proposal_networks"]
    moved = []
    named_h = {n: p for n, p in hm.named_parameters() if n.startswith("proposal_networks")}
    named_o = {n: p for n, p in om.named_parameters() if n.startswith("proposal_networks")}
    for step in range(9, 13):
        o, d, pa, cam = util.random_rays(R, 7, seed=300 + step)
        jit = [torch.rand(R, 1) for _ in range(3)]
        batch = _batch(R, 70 + step)
        for op in opts.values():
            op.zero_grad(set_to_none=False)            # torch 1.13's default
        for p in groups["proposal_networks"]:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        before_o = {n: p.detach().clone() for n, p in named_o.items()}
        before_h = {n: p.detach().clone() for n, p in named_h.items()}
        _oracle_step(om, o, d, pa, cam, jit, batch, step)
        g_max = max(float(p.grad.abs().max()) for p in groups["proposal_networks"])
        for op in opts.values():
            op.step()
        om.proposal_sampler.step_cb(step)
        before = hm.arena().params[a:b].clone()
        fused_train_iteration(hm, hopt, RayBundle(o.to(dev), d.to(dev), pa.to(dev), cam.to(dev)),
                              {k: v.to(dev) for k, v in batch.items()}, step, jitter=[j.to(dev) for j in jit])
        torch.cuda.synchronize()
        moved.append(not torch.equal(before, hm.arena().params[a:b]))
        if step == 11:
            # no gradient on either side, and both move along their momentum by the same amounts.  (Entry-wise equality
            # of the parameters is not the criterion: Adam's first step is -lr * sign(g), so a gradient entry whose sign
            # is rounding noise differs by 2 lr from the first iteration on, on any two implementations.)
            assert g_max == 0.0
            for n in named_o:
                mo = (named_o[n].detach() - before_o[n]).abs()
                mh = (named_h[n].detach() - before_h[n]).abs().cpu()
                assert float(mo.max()) > 1e-3 and abs(float(mh.max()) - float(mo.max())) <= 1e-3 * float(mo.max()), n
                assert abs(float(mh.mean()) - float(mo.mean())) <= 2e-2 * float(mo.mean()) + 1e-9, n
    assert moved == [True, True, True, True]           # iteration 11 has no gradient and still moves (momentum)
    assert hopt.group_steps["proposal_networks"] == 4
    assert {int(opts["proposal_networks"].state[p]["step"]) for p in groups["proposal_networks"]} == {4}


@pytest.mark.parametrize("shape", ["fruit_nerf", "fruit_nerf_big"])
def test_fused_step_matches_the_autograd_step(dev, shape):
    """fused_forward_backward() (no autograd engine, what bench.py times) must leave the same losses, metrics and
    gradients as model(...) -> get_metrics_dict -> get_loss_dict -> sum -> backward."""
    from fruitnerf_amd.rays import RayBundle
    from fruitnerf_amd.training import fused_forward_backward
    cfg = (util.small_config if shape == "fruit_nerf" else util.big_config)(log2=15, prop_log2=13)
    om = util.make_oracle(cfg, seed=9)
    R = 192
    o, d, pa, cam = util.random_rays(R, 7, seed=4)
    jit = [torch.rand(R, 1).to(dev) for _ in range(3)]
    hb = {k: v.to(dev) for k, v in _batch(R, 8).items()}
    results = []
    for fused in (False, True):
        hm = util.make_hip_like(om, dev)
        hm.train()
        hm.set_anneal(0)
        rb = RayBundle(o.to(dev), d.to(dev), pa.to(dev), cam.to(dev))
        if fused:
            ld, md = fused_forward_backward(hm, rb, hb, jitter=jit)
        else:
            out = hm(rb, jitter=jit)
            md = hm.get_metrics_dict(out, hb)
            ld = hm.get_loss_dict(out, hb)
            sum(ld.values()).backward()
        torch.cuda.synchronize()
        results.append((ld, md, hm.arena().grads.clone()))
    (ld_a, md_a, g_a), (ld_f, md_f, g_f) = results
    for k in ld_a:
        assert abs(float(ld_a[k]) - float(ld_f[k])) <= 1e-6 * max(abs(float(ld_a[k])), 1e-6), k
    for k in md_a:
        assert abs(float(md_a[k]) - float(md_f[k])) <= 1e-6 * max(abs(float(md_a[k])), 1e-6), k
    scale = g_a.abs().max().item()
    assert scale > 0
    # same kernels, same inputs; only the order of a few float atomics (weight-gradient partials) may differ
    assert (g_a - g_f).abs().max().item() <= 1e-5 * scale
    assert int((g_a != 0).sum()) == int((g_f != 0).sum())


def test_level_group_scatter_matches_single_scatter(dev, monkeypatch):
    """Data-parallel training scatters the hash-grid gradient per group of levels (so each group's rows can be
    all-reduced early): same gradients as the single scatter, and the exchanged spans tile the field's parameters."""
    from fruitnerf_amd.rays import RayBundle
    import fruitnerf_amd.training as T
    monkeypatch.setattr(T, "start_gradient_sync", lambda arena, span, world, bucket_elems=0, **kw: [(span[0], span[1], None)])
    monkeypatch.setattr(T, "SHARDED_FIELD_OPTIMIZER", False)     # (the level groups' spans as all-reduce buckets)
    cfg = util.small_config(log2=15, prop_log2=13)
    om = util.make_oracle(cfg, seed=9)
    R = 160
    o, d, pa, cam = util.random_rays(R, 7, seed=4)
    jit = [torch.rand(R, 1).to(dev) for _ in range(3)]
    hb = {k: v.to(dev) for k, v in _batch(R, 8).items()}
    grads, spans = [], None
    for grouped in (False, True):
        hm = util.make_hip_like(om, dev)
        hm.train()
        hm.set_anneal(0)
        ex = T._FieldGradientExchange(hm, 2, level_groups=4) if grouped else None
        T.fused_forward_backward(hm, RayBundle(o.to(dev), d.to(dev), pa.to(dev), cam.to(dev)), hb, jitter=jit,
                                 exchange=ex)
        torch.cuda.synchronize()
        grads.append(hm.arena().grads.clone())
        if ex is not None:
            spans = sorted((a, b) for a, b, _ in ex.pending)
            f0, f1 = hm.arena().group_ranges["fields"]
    assert spans[0][0] == f0 and spans[-1][1] == f1
    assert all(spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1)), spans
    assert len(spans) == 4      # one collective per level group: the MLP weights ride with the first / last group
    scale = grads[0].abs().max().item()
    assert (grads[0] - grads[1]).abs().max().item() <= 1e-5 * scale
    assert int((grads[0] != 0).sum()) == int((grads[1] != 0).sum())


@pytest.mark.parametrize("log2,prop_log2", [(15, 13), (19, 17)])
def test_scatter_is_invariant_to_the_levels_per_workgroup(dev, monkeypatch, log2, prop_log2):
    """k_scatter_emit takes a workgroup's 512 samples through `lpb` consecutive levels (hash_scatter.hip); the launch
    heuristic only picks lpb > 1 at bench-sized calls, so the test forces it (FNR_EMIT_LPB).  The table gradients are
    sums in 64-bit block fixed point, i.e. order-independent: bit-equal across lpb = 1, 2, 3 (ragged: 16 = 5 x 3 + 1 and
    5 = 3 + 2 levels), 5.  Small tables take the per-corner path, full-size tables the corner-PAIR path."""
    from fruitnerf_amd.rays import RayBundle
    import fruitnerf_amd.training as T
    cfg = util.small_config(log2=log2, prop_log2=prop_log2)
    om = util.make_oracle(cfg, seed=9)
    R = 96
    o, d, pa, cam = util.random_rays(R, 7, seed=4)
    jit = [torch.rand(R, 1).to(dev) for _ in range(3)]
    hb = {k: v.to(dev) for k, v in _batch(R, 8).items()}
    grads = []
    for lpb in ("1", "1", "2", "3", "5"):
        monkeypatch.setenv("FNR_EMIT_LPB", lpb)
        hm = util.make_hip_like(om, dev)
        hm.train()
        hm.set_anneal(0)
        T.fused_forward_backward(hm, RayBundle(o.to(dev), d.to(dev), pa.to(dev), cam.to(dev)), hb, jitter=jit)
        torch.cuda.synchronize()
        # every gradient: the hash tables are fixed-point sums, the MLP weights fixed-order sums (k_reduce_dw)
        grads.append([hm.arena().grads.detach().clone()])
    assert all(float(t.abs().max()) > 0 for t in grads[0])
    for which, g in enumerate(grads[1:]):
        for t, ref in zip(g, grads[0]):
            diff = (t - ref).abs().max().item()
            assert torch.equal(t, ref), (which, diff, float(ref.abs().max()))


@pytest.mark.parametrize("shape,algorithm,log2", [("fruit_nerf", "adam", 15), ("fruit_nerf", "adam", 19),
                                                  ("fruit_nerf_big", "radam", 15)])
def test_fused_table_optimizer_matches_the_separate_step(dev, shape, algorithm, log2):
    """fnr_hash_encode_bwd_adam: the main hash table's Adam / RAdam step applied inside the scatter's accumulate kernel
    (single-process training; the gradient table is never written) vs scatter + fnr_adam_step / fnr_radam_step.
    First step from identical states: the table's parameters and both moments are BIT-identical (same fixed-point
    sums, same operation order), every row took its step (moment decay of untouched rows included), the table's
    gradient stays zero.  Then 11 more steps across step 10 (proposal nets no longer updated every step)."""
    from fruitnerf_amd.rays import RayBundle
    from fruitnerf_amd.training import FusedAdam, fused_train_iteration
    cfg = {"fruit_nerf": util.small_config, "fruit_nerf_big": util.big_config}[shape](log2=log2, prop_log2=13)
    om = util.make_oracle(cfg, seed=3)
    R = 128
    o, d, pa, cam = util.random_rays(R, 7, seed=2)
    hb = {k: v.to(dev) for k, v in _batch(R, 5).items()}
    g = torch.Generator().manual_seed(0)
    jits = [[torch.rand(R, 1, generator=g).to(dev) for _ in range(3)] for _ in range(12)]
    runs = []
    for fuse in (False, True, False):
        hm = util.make_hip_like(om, dev)
        hm.train()
        opt = FusedAdam(hm, algorithm=algorithm)
        snaps = []
        for step in range(12):
            fused_train_iteration(hm, opt, RayBundle(o.to(dev), d.to(dev), pa.to(dev), cam.to(dev)), hb, step,
                                  jitter=jits[step], fuse_table_optimizer=fuse)
            if step in (0, 11):
                torch.cuda.synchronize()
                snaps.append((hm.arena().params.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone()))
        table = hm.field.mlp_base_grid.hash_table
        a, n = [(off, k) for _, p, off, k in hm.arena().entries if p is table][0]
        assert float(hm.arena().grads.abs().max()) == 0.0          # zero_grad everywhere, the table never written
        runs.append((snaps, (a, a + n)))
    (s0, (a, b)), (s1, _), (s2, _) = runs
    # Every cross-workgroup sum has a fixed order since round 3 (k_reduce_dw / k_prop_reduce: single writers), so the
    # comparison is exact: the fused run — table step inside the scatter's accumulate kernel, MLP-weight and embedding
    # steps inside k_reduce_dw / k_embedding_grad (fnr_field_mlp_bwd_adam) — is BIT-identical to the separate-step run,
    # parameters and both moments, after the first step and after 12 (across step 10, where the proposal networks stop
    # being updated every step and their optimiser step is skipped).
    for snap_sep, snap_fused, snap_sep2 in zip(s0, s1, s2):
        for x, y, z in zip(snap_sep, snap_fused, snap_sep2):
            assert torch.equal(x, z), "two separate-step runs differ: training is no longer reproducible"
            n_diff = int((x != y).sum())
            assert n_diff == 0, f"{n_diff} entries differ between the fused and the separate optimiser steps " \
                                f"(max {float((x - y).abs().max()):.3e})"
    p0 = s0[0][0][a:b]
    assert int((p0 != runs[0][0][1][0][a:b]).sum()) > 0            # the table did move between step 1 and step 12


def test_sparse_touch_bitmap_survives_unfused_table_steps(dev):
    """the sparse-touch bitmap is only maintained by the fused table kernels; a step of the table that goes
    through fnr_adam_step instead (fuse_table_optimizer=False here; the exchange path, scaler_step) makes moments non-zero
    without setting bits.  FusedAdam drops the bitmap of any span stepped that way and rebuilds it from the moments at the
    next fused step: fused / unfused steps interleaved, each step on OTHER rays (so that rows get their first gradient on
    the unfused steps), with the skipping on against every row swept — bit-identical parameters and moments."""
    import fruitnerf_amd.training as T
    from fruitnerf_amd.rays import RayBundle
    cfg = util.small_config(log2=15, prop_log2=13)
    om = util.make_oracle(cfg, seed=3)
    R = 96
    hb = {k: v.to(dev) for k, v in _batch(R, 5).items()}
    g = torch.Generator().manual_seed(0)
    pattern = (True, False, True, True, False, True)
    jits = [[torch.rand(R, 1, generator=g).to(dev) for _ in range(3)] for _ in pattern]
    bundles = []
    for i in range(len(pattern)):
        o, d, pa, cam = util.random_rays(R, 7, seed=20 + i)
        bundles.append(RayBundle(o.to(dev), d.to(dev), pa.to(dev), cam.to(dev)))
    saved = T.SPARSE_TOUCH_SKIPPING
    states = []
    try:
        for sparse in (True, False):
            T.SPARSE_TOUCH_SKIPPING = sparse
            hm = util.make_hip_like(om, dev)
            hm.train()
            opt = T.FusedAdam(hm)
            for step, fuse in enumerate(pattern):
                T.fused_train_iteration(hm, opt, bundles[step], hb, step, jitter=jits[step], fuse_table_optimizer=fuse)
                if sparse:
                    assert bool(opt._touched) == fuse      # dropped by the unfused step, rebuilt by the next fused one
            torch.cuda.synchronize()
            states.append((hm.arena().params.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone()))
    finally:
        T.SPARSE_TOUCH_SKIPPING = saved
    for name, x, y in zip(("parameters", "exp_avg", "exp_avg_sq"), *states):
        assert torch.equal(x, y), f"{name}: {int((x != y).sum())} entries differ between sparse-touch and dense sweeps"


def test_fused_radam_step_of_fruit_nerf_big_tracks_torch_optim_radam(dev):
    """`fruit_nerf_big` trains with RAdam (fruit_nerf_config.py:97-106).  Eight fused steps of that shape — the table's step
    inside the scatter's accumulate kernel, the MLP weights' inside k_reduce_dw / k_embedding_grad, the proposal networks'
    inside their backward (fnr_*_adam entry points, algorithm = radam), across the rectification threshold (rho_t > 5 from
    step 6 on) — against the SAME HIP gradient kernels stepped by torch.optim.RAdam itself (a second model whose optimiser
    hands the arena's gradient to torch.optim.RAdam(lr = the scheduler's, eps 1e-15) and copies the result back).  Equal
    gradients in, so what differs is fp32 rounding inside the update: through the five unrectified steps the parameters agree
    to rounding (measured: max 7e-9), behind the threshold on average to a few 1e-3 of the distance they moved (measured at
    step 7: mean difference 8e-8 against a mean update of 1.6e-4; 0.4 % of the entries — gradients at rounding level, which
    the rectified update turns into a step of either sign — further apart than 1e-6)."""
    import fruitnerf_amd.training as T
    from fruitnerf_amd.rays import RayBundle
    cfg = util.big_config(log2=15, prop_log2=13)
    om = util.make_oracle(cfg, seed=3)
    R = 128
    hb = {k: v.to(dev) for k, v in _batch(R, 5).items()}
    g = torch.Generator().manual_seed(0)
    n_steps = 8                                     # < 10: the proposal networks are updated on every one of them
    jits = [[torch.rand(R, 1, generator=g).to(dev) for _ in range(3)] for _ in range(n_steps)]
    bundles = []
    for i in range(n_steps):
        o, d, pa, cam = util.random_rays(R, 7, seed=40 + i)
        bundles.append(RayBundle(o.to(dev), d.to(dev), pa.to(dev), cam.to(dev)))

    class TorchRAdam(T.FusedAdam):
        """FusedAdam's bookkeeping (schedulers, counters), torch.optim.RAdam's arithmetic."""

        def __init__(self, model):
            super().__init__(model, algorithm="radam")
            self.ref = self.arena.params.detach().clone().requires_grad_(True)
            self.torch_opt = torch.optim.RAdam([self.ref], lr=1e-2, eps=self.eps, betas=self.betas)

        def step(self, grad_scale=1.0, skip=(), done=()):
            assert not skip and not done
            lrs = self.begin_step(skip)
            assert len(set(lrs.values())) == 1
            self.torch_opt.param_groups[0]["lr"] = next(iter(lrs.values()))
            self.ref.grad = self.arena.grads.detach().clone() * grad_scale
            self.torch_opt.step()
            self.arena.params.copy_(self.ref.detach())
            self.arena.grads.zero_()

    hm_f, hm_t = util.make_hip_like(om, dev), util.make_hip_like(om, dev)
    hm_f.train()
    hm_t.train()
    p0 = hm_f.arena().params.clone()
    opt_f, opt_t = T.FusedAdam(hm_f, algorithm="radam"), TorchRAdam(hm_t)
    for step in range(n_steps):
        T.fused_train_iteration(hm_f, opt_f, bundles[step], hb, step, jitter=jits[step])
        T.fused_train_iteration(hm_t, opt_t, bundles[step], hb, step, jitter=jits[step], fuse_table_optimizer=False)
        torch.cuda.synchronize()
        a, b = hm_f.arena().params, hm_t.arena().params
        diff, moved = (a - b).abs(), (b - p0).abs()
        print(f"[fused RAdam vs torch.optim.RAdam] step {step + 1}: max diff {float(diff.max()):.3e}  mean diff "
              f"{float(diff.mean()):.3e}  mean |moved| {float(moved.mean()):.3e}  max |moved| {float(moved.max()):.3e}")
        # steps 1 - 5 (rho_t <= 5: p -= lr m_hat, no division) agree to rounding.  From step 6 on the update is
        # lr r_t m_hat / (sqrt(v_hat) + 1e-15), which turns ANY non-zero gradient into a step of the order of lr r_t: entries
        # whose gradient is rounding noise take either sign on the two sides (the Adam test above has the same caveat), so
        # the bulk is bounded — mean difference against the mean update, fraction of entries apart by more than 1e-6 —
        # and the worst entry by the number of rectified steps times the step size
        assert float(diff.mean()) <= (1e-3 if step < 5 else 5e-3) * float(moved.mean()) + 1e-9, step
        if step < 5:
            assert float(diff.max()) <= 1e-3 * float(moved.max()) + 1e-7, step
        else:
            assert float((diff > 1e-6).float().mean()) <= 2e-2, step      # (measured 3.6e-3 of the arena at step 7)
            assert float(diff.max()) <= (step - 4) * 1e-2, step
    assert float(moved.max()) > 1e-3                 # the rectified steps did move the parameters
    # moments too: torch keeps them in its state, the fused path in FusedAdam's arenas
    st = opt_t.torch_opt.state[opt_t.ref]
    for name, x, y in (("exp_avg", opt_f.exp_avg, st["exp_avg"]), ("exp_avg_sq", opt_f.exp_avg_sq, st["exp_avg_sq"])):
        d = float((x - y).abs().max())
        dm = float((x - y).abs().mean())
        print(f"[fused RAdam vs torch.optim.RAdam] {name}: max diff {d:.3e} of max {float(y.abs().max()):.3e}, mean diff "
              f"{dm:.3e} of mean {float(y.abs().mean()):.3e}")
        assert d <= 1e-2 * float(y.abs().max()) and dm <= 1e-3 * float(y.abs().mean()) + 1e-20, name


def test_ray_gradient_paths_agree(dev):
    """The saved-Jacobian path (forward encode stores d feats / d x) and the gather path (backward re-reads the table)
    of the hash grid's input gradient give the same ray gradients."""
    from fruitnerf_amd import _kernels as K
    from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
    from fruitnerf_amd.data.semantics import apple_metadata
    torch.manual_seed(0)
    m = FruitModel(FruitNerfModelConfig(log2_hashmap_size=15), apple_metadata(), num_train_data=4, device=dev)
    with torch.no_grad():
        m.field.mlp_base_grid.hash_table.uniform_(-0.5, 0.5)
    m.train()
    m.arena()
    fld = m.field
    R, S = 96, 48
    o, d, pa, cam = util.random_rays(R, 4, seed=2)
    rays = K.RaysArg(o.to(dev), d.to(dev), torch.full((R, 1), 0.05, device=dev), torch.full((R, 1), 6.0, device=dev),
                     cam.to(dev))
    _, eu = K.sample_spaced(rays, 1, S, None)
    net = fld.net_struct()
    feats, sel, jac = K.hash_encode_fwd(net.grid, fld.warp_struct(), rays, eu, S, want_jacobian=True)
    d_feats = torch.randn_like(feats)
    a = [torch.zeros(R, 3, device=dev) for _ in range(2)]
    b = [torch.zeros(R, 3, device=dev) for _ in range(2)]
    K.position_grad_from_jacobian(fld.warp_struct(), rays, eu, S, jac, d_feats, a[0], a[1])
    partial = K.hash_encode_input_grad(net.grid, fld.warp_struct(), rays, eu, S, d_feats)
    K.position_grad_reduce(fld.warp_struct(), rays, eu, S, partial, b[0], b[1])
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert float(y.abs().max()) > 0
        assert float((x - y).abs().max()) <= 1e-4 * float(y.abs().max())


@pytest.mark.parametrize("shape", ["fruit_nerf", "fruit_nerf_big"])
@pytest.mark.parametrize("mode", ["fp32", "bf16x3", "bf16"])
def test_input_gradient_inside_the_mlp_backward_is_the_separate_launch(dev, shape, mode):
    """fnr_field_mlp_bwd_rays: the base-branch kernel contracts the dL/dfeats it holds in registers with the encode's
    saved Jacobian; finished by fnr_position_grad_reduce this must give the ray gradients of
    fnr_position_grad_from_jacobian on the d_feats the same call wrote (only the summation order over levels differs),
    and every other output of the backward must be bit-identical to the call without a Jacobian."""
    from fruitnerf_amd import _kernels as K
    cfg = (util.small_config if shape == "fruit_nerf" else util.big_config)(log2=15, prop_log2=13)
    om = util.make_oracle(cfg, seed=3)
    hm = util.make_hip_like(om, dev)
    hm.field.mlp_precision = mode
    hm.train()
    hm.arena()
    fld = hm.field
    R, S = 200, 40                      # 8000 samples: a ragged last batch of the 128-sample workgroups
    o, d, pa, cam = util.random_rays(R, 7, seed=4)
    rays = K.RaysArg(o.to(dev), d.to(dev), torch.full((R, 1), 0.05, device=dev), torch.full((R, 1), 6.0, device=dev),
                     cam.to(dev))
    _, eu = K.sample_spaced(rays, 1, S, None)
    net, gnet = fld.net_struct(), fld.net_struct(grads=True)
    feats, sel, jac = K.hash_encode_fwd(net.grid, fld.warp_struct(), rays, eu, S, want_jacobian=True)
    density, rgb, logit, _, saved = K.field_mlp_fwd(net, rays, S, feats, sel, None, want_h=True)
    g = torch.Generator().manual_seed(1)
    N = R * S
    dd, dr, dl = (torch.randn(N, generator=g).to(dev), torch.randn(N, 3, generator=g).to(dev), torch.randn(N, generator=g).to(dev))
    hm.arena().grads.zero_()
    d_feats0 = K.field_mlp_bwd(net, gnet, rays, S, feats, saved, sel, dd, dr, dl)
    g0 = hm.arena().grads.clone()
    hm.arena().grads.zero_()
    d_feats1, d_pos = K.field_mlp_bwd(net, gnet, rays, S, feats, saved, sel, dd, dr, dl, jacobian=jac)
    torch.cuda.synchronize()
    assert torch.equal(d_feats0, d_feats1) and torch.equal(g0, hm.arena().grads)
    a = [torch.zeros(R, 3, device=dev) for _ in range(2)]
    b = [torch.zeros(R, 3, device=dev) for _ in range(2)]
    K.position_grad_reduce(fld.warp_struct(), rays, eu, S, d_pos.view(1, N, 4), a[0], a[1])
    K.position_grad_from_jacobian(fld.warp_struct(), rays, eu, S, jac, d_feats1, b[0], b[1])
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert float(y.abs().max()) > 0
        err = float((x - y).abs().max()) / float(y.abs().max())
        print(f"[input grad in mlp bwd {shape} {mode}] rel err {err:.3e}")
        assert err <= 2e-6


@pytest.mark.parametrize("fused", [False, True])
def test_ray_gradients_match_autograd(dev, fused):
    """d(loss)/d(origins), d(loss)/d(directions): the gradient a camera-pose optimiser consumes
    (fruit_nerf_config.py:39-43).  Oracle: plain autograd with the rays as leaves."""
    from fruitnerf_amd.rays import RayBundle
    from fruitnerf_amd.training import fused_forward_backward
    cfg = util.small_config(log2=15, prop_log2=13)
    om = util.make_oracle(cfg, seed=11)
    hm = util.make_hip_like(om, dev)
    om.train()
    hm.train()
    R = 160
    o, d, pa, cam = util.random_rays(R, 7, seed=33)
    jit = [torch.rand(R, 1) for _ in range(3)]
    batch = _batch(R, 5)
    o_ref, d_ref = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
    om.set_anneal(0)
    out = om(ns.RayBundle(o_ref, d_ref, pa.clone(), camera_indices=cam.clone()), jitter=jit)
    sum(om.get_loss_dict(out, batch).values()).backward()

    hm.set_anneal(0)
    hb = {k: v.to(dev) for k, v in batch.items()}
    hjit = [j.to(dev) for j in jit]
    if fused:
        got = {}
        fused_forward_backward(hm, RayBundle(o.to(dev), d.to(dev), pa.to(dev), cam.to(dev)), hb, jitter=hjit,
                               ray_grads=got)
        g_o, g_d = got["origins"], got["directions"]
    else:
        o_h, d_h = o.to(dev).requires_grad_(True), d.to(dev).requires_grad_(True)
        hout = hm(RayBundle(o_h, d_h, pa.to(dev), cam.to(dev)), jitter=hjit)
        sum(hm.get_loss_dict(hout, hb).values()).backward()
        g_o, g_d = o_h.grad, d_h.grad
    torch.cuda.synchronize()
    for name, got_g, ref_g in (("origins", g_o, o_ref.grad), ("directions", g_d, d_ref.grad)):
        scale = ref_g.abs().max().item()
        err = (got_g.cpu() - ref_g).abs().max().item()
        print(f"[ray grad fused={fused}] {name}: max|ref| {scale:.3e} max_err {err:.3e} rel {err / scale:.3e}")
        assert scale > 0 and err <= 5e-3 * scale, name  # directions: sum of t * g with t up to the far plane


def test_camera_optimizer_pose_gradients_and_step(dev):
    """CameraOptimizer(SO3xR3) (fruit_nerf_config.py:39-43): corrected cameras, rays, d(loss)/d(pose_adjustment) and the
    Adam(weight_decay) update, HIP vs the oracle's autograd through exp_map -> multiply -> ray generation -> model."""
    from oracle import camera_opt as oc
    from fruitnerf_amd import _kernels as K
    from fruitnerf_amd.cameras.camera_optimizers import CameraAdam, CameraOptimizerConfig
    from fruitnerf_amd.data import synthetic_apple as sa
    from fruitnerf_amd.rays import RayBundle
    from fruitnerf_amd.training import fused_forward_backward
    n_cam, HW, focal, R = 8, 64, 90.0, 192
    cfg = util.small_config(log2=15, prop_log2=13)
    om = util.make_oracle(cfg, num_images=n_cam, seed=13)
    hm = util.make_hip_like(om, dev)
    om.train()
    hm.train()
    scene = sa.make_scene(seed=0)
    c2w = sa.make_cameras(n_cam, seed=0)
    data = sa.render_dataset(scene, c2w, H=HW, W=HW, fx=focal, fy=focal)
    train_ids = torch.arange(n_cam)
    g = torch.Generator().manual_seed(3)
    pose0 = torch.cat([torch.randn(n_cam, 3, generator=g) * 0.02, torch.randn(n_cam, 3, generator=g) * 0.03], dim=1)
    pose0[0, 3:] = 0.0   # one camera below the 1e-4 clamp of the rotation angle
    u = torch.rand(R, 3, generator=g)
    jit = [torch.rand(R, 1) for _ in range(3)]

    # ---- oracle: autograd from the losses to the pose parameters ----
    ocam = oc.CameraOptimizer(n_cam)
    with torch.no_grad():
        ocam.pose_adjustment.copy_(pose0)
    k = (u[:, 0] * n_cam).long().clamp_max(n_cam - 1)
    y = (u[:, 1] * HW).long().clamp_max(HW - 1)
    x = (u[:, 2] * HW).long().clamp_max(HW - 1)
    o_ref, d_ref = oc.generate_rays(c2w[train_ids[k]], ocam(k), y, x, focal, focal, HW / 2.0, HW / 2.0)
    batch = {"image": data["images"][train_ids[k], y, x].float() / 255.0,
             "fruit_mask": data["masks"][train_ids[k], y, x].float()[:, None]}
    om.set_anneal(0)
    out = om(ns.RayBundle(o_ref, d_ref, torch.ones(R, 1), camera_indices=k[:, None]), jitter=jit)
    sum(om.get_loss_dict(out, batch).values()).backward()
    oopt = torch.optim.Adam(ocam.parameters(), lr=6e-4, eps=1e-8, weight_decay=1e-2)
    g_ref = ocam.pose_adjustment.grad.clone()
    with torch.no_grad():
        delta_ref = ocam(torch.arange(n_cam)).clone()
        c2w_ref = oc.multiply(c2w, delta_ref)
    oopt.step()

    # ---- HIP ----
    hcam = CameraOptimizerConfig(mode="SO3xR3").setup(n_cam, dev)
    with torch.no_grad():
        hcam.pose_adjustment.copy_(pose0.to(dev))
    ddev = {kk: (v.to(dev) if torch.is_tensor(v) else v) for kk, v in data.items()}
    batcher = sa.PixelBatcher(ddev, train_ids.to(dev), seed=0)
    batcher._set = K.ImageSetArg(ddev["images"], ddev["masks"], ddev["c2w"], focal, focal, HW / 2.0, HW / 2.0)
    c2w_adj = hcam.adjusted_cameras(batcher._set, batcher.image_ids)
    a, _ = util.report("camera.c2w_adjusted", c2w_adj, c2w_ref)
    assert a <= 1e-6
    b, _ = util.report("camera.forward", hcam(torch.arange(n_cam, device=dev)), delta_ref)
    assert b <= 1e-6
    o_h, d_h, cam_h, image, mask = K.sample_pixels(batcher._set, batcher.image_ids, u.to(dev), c2w_adj)
    assert util.report("camera.origins", o_h, o_ref.detach())[0] <= 1e-6
    assert util.report("camera.directions", d_h, d_ref.detach())[0] <= 1e-6
    batcher.last_draw = {"u": u.to(dev), "cam": cam_h, "c2w_adjusted": c2w_adj}
    hm.set_anneal(0)
    got = {}
    fused_forward_backward(hm, RayBundle(o_h, d_h, None, cam_h[:, None]), {"image": image, "fruit_mask": mask[:, None]},
                           jitter=[j.to(dev) for j in jit], ray_grads=got)
    hadam = CameraAdam(hcam)
    hcam.pose_adjustment.grad.zero_()
    K.camera_pose_grad(batcher._set, batcher.image_ids, batcher.last_draw["u"], cam_h, hcam.pose_adjustment.data, c2w_adj,
                       got["origins"], got["directions"], hcam.pose_adjustment.grad)
    torch.cuda.synchronize()
    g_hip = hcam.pose_adjustment.grad.cpu().clone()
    scale = g_ref.abs().max().item()
    err = (g_hip - g_ref).abs().max().item()
    print(f"[camera] pose grad: max|ref| {scale:.3e} max_err {err:.3e} rel {err / scale:.3e}")
    assert scale > 0 and err <= 2e-3 * scale
    assert (g_ref[0, 3:].abs().max() > 0) and (g_hip[0, 3:] - g_ref[0, 3:]).abs().max() <= 2e-3 * scale
    hadam.step()
    torch.cuda.synchronize()
    assert util.report("camera.pose_after_adam", hcam.pose_adjustment.data, ocam.pose_adjustment.data)[0] <= 2e-6
    assert float(hcam.pose_adjustment.grad.abs().max()) == 0.0
    # fnr_camera_pose_grad_adam (gradient + optimiser step in one launch, the single-process training path) from the
    # same state: pose and both moments bit-identical to the two launches above, gradient left zero
    from fruitnerf_amd.training import camera_backward_and_step
    hcam2 = CameraOptimizerConfig(mode="SO3xR3").setup(n_cam, dev)
    with torch.no_grad():
        hcam2.pose_adjustment.copy_(pose0.to(dev))
    hadam2 = CameraAdam(hcam2)
    camera_backward_and_step(hcam2, hadam2, batcher, got, world_size=1)
    torch.cuda.synchronize()
    assert torch.equal(hcam2.pose_adjustment.data, hcam.pose_adjustment.data)
    assert torch.equal(hadam2.exp_avg, hadam.exp_avg) and torch.equal(hadam2.exp_avg_sq, hadam.exp_avg_sq)
    assert float(hcam2.pose_adjustment.grad.abs().max()) == 0.0 and hadam2.step_count == hadam.step_count == 1


def test_step_at_a_trained_state_matches_the_oracle(dev):
    """Parity where it is hardest: after 2500 HIP training steps on the synthetic scene (full `fruit_nerf` sizes) the
    densities are sharp (delta*sigma up to ~1e8, saturated sigmoids, peaky PDF samples).  One more step is replayed in
    the CPU oracle with the same weights, rays, jitter and batch: losses and every gradient must agree.  (This is
    the check that exposed the cancelling exclusive scan; the random-weight tests above cannot.)"""
    from fruitnerf_amd.data import synthetic_apple as sa
    from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
    from fruitnerf_amd.data.semantics import apple_metadata
    from fruitnerf_amd.rays import RayBundle
    from fruitnerf_amd.training import FusedAdam, fused_forward_backward, fused_train_iteration
    HW, focal, n_train = 96, 1111.0 * 96 / 800, 40
    scene = sa.make_scene(seed=0, device=dev)
    c2w = sa.make_cameras(n_train, seed=0, device=dev)
    data = sa.render_dataset(scene, c2w, H=HW, W=HW, fx=focal, fy=focal)
    batcher = sa.PixelBatcher(data, torch.arange(n_train, device=dev), seed=1)
    torch.manual_seed(0)
    hm = FruitModel(FruitNerfModelConfig(), apple_metadata(), num_train_data=n_train, device=dev)
    hm.train()
    opt = FusedAdam(hm)
    steps = 2500
    for step in range(steps):
        o, d, cam, batch = batcher.sample(4096)
        fused_train_iteration(hm, opt, RayBundle(o, d, None, cam), batch, step, want_metrics=False)
    R = 768
    o, d, cam, batch = batcher.sample(R)
    jit = [torch.rand(R, 1, device=dev) for _ in range(3)]
    hm.set_anneal(steps)
    samp = hm.proposal_sampler
    samp._steps_since_update = 100          # make this an "updated" step: proposal-network gradients are exercised too
    state = (samp._step, samp._steps_since_update)
    ray_grads = {}
    ld, md = fused_forward_backward(hm, RayBundle(o, d, None, cam), batch, jitter=jit, ray_grads=ray_grads)
    torch.cuda.synchronize()
    assert float(ld["rgb_loss"]) < 2e-3, "the model did not train"

    om = fo.FruitModel(fo.FruitNerfModelConfig(), num_train_data=n_train)
    om.load_state_dict({k: v.detach().cpu() for k, v in hm.state_dict().items()}, strict=True)
    om.train()
    om.proposal_sampler._step, om.proposal_sampler._steps_since_update = state
    om.set_anneal(steps)
    o_ref, d_ref = o.cpu().clone().requires_grad_(True), d.cpu().clone().requires_grad_(True)
    out = om(ns.RayBundle(o_ref, d_ref, torch.ones(R, 1), camera_indices=cam.cpu().long()), jitter=[j.cpu() for j in jit])
    b = {k: v.cpu() for k, v in batch.items()}
    ld_ref = om.get_loss_dict(out, b)
    md_ref = om.get_metrics_dict(out, b)
    sum(ld_ref.values()).backward()
    for name, got_g, ref_g in (("origins", ray_grads["origins"], o_ref.grad), ("directions", ray_grads["directions"], d_ref.grad)):
        diff = (got_g.cpu() - ref_g).abs()
        scale = ref_g.abs().max().item()
        per_ray = diff.max(dim=1)[0]
        off = (per_ray > 1e-2 * scale).float().mean().item()
        keep = per_ray.argsort()[: int(0.97 * R)]                 # all but the worst 3 % of the rays
        agg = diff[keep].sum().item() / ref_g[keep].abs().sum().item()
        print(f"[trained state] d loss / d {name}: max|ref| {scale:.3e} max_err {diff.max().item():.3e} rays off {off:.2%} "
              f"aggregate rel (97 % of rays) {agg:.3e}")
        # A sample within ~1e-4 of a cell face takes the neighbouring cell's slope (piecewise-linear encoding, see
        # tests/test_golden.py); at a trained state one such sample at a fine level (slope ~ 2048 x table value x d_feat,
        # times t for the direction) can exceed the whole ray's gradient.  A few rays may be off; the rest must match.
        assert off <= 5e-3 and agg <= 1e-2, name      # measured (round 3): 0.00 % of rays off, aggregate 9e-5
    for k in ld_ref:
        a, r = float(ld[k]), float(ld_ref[k])
        print(f"[trained state] {k}: hip {a:.8e} oracle {r:.8e}")
        # usually 6 digits; a PDF sample that lands on the other side of a sharp surface (sampler parity is 2e-6, the
        # trained density changes by orders of magnitude within 1e-4) moves one ray's output and the batch mean by ~1e-3
        assert abs(a - r) <= 1e-2 * max(abs(r), 1e-4), k   # measured (round 3): <= 2.5e-3
    for k in md_ref:
        a, r = float(md[k]), float(md_ref[k])
        assert abs(a - r) <= 1e-2 * max(abs(r), 1e-3), k
    _grad_report(om, hm, " trained state")
    # Training is bit-reproducible since round 3, so the trained state and these numbers are the same on every run of a
    # build.  Criterion per tensor, at ~5x what round 3 measured (worst L1-rel 3.7e-4 / max-norm rel 9.4e-4 here, 2.5e-3 /
    # 1.9e-3 over both trained-state replays, profiles/r03_raw): L1-rel <= 1e-2 AND max-norm rel <= 1e-2 — a kernel
    # regression of 10x fails.  (The same-samples leg in tests/test_gpu_trained_state.py holds 5e-4 of max |g|.)
    named_h = dict(hm.named_parameters())
    for name, p in util.named_trainable(om):
        ref = p.grad if p.grad is not None else torch.zeros_like(p)
        got = named_h[name].grad.detach().cpu()
        denom = ref.abs().double().sum().item()
        agg = (got - ref).abs().double().sum().item() / max(denom, 1e-30)
        mx = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
        assert denom == 0 and float(got.abs().sum()) == 0 or (agg <= 1e-2 and mx <= 1e-2), \
            f"{name}: gradient error L1-rel {agg} max-norm rel {mx}"


def test_export_at_a_trained_state_matches_the_oracle(dev):
    """Exact export counts are the north star's second parity bar: after 2000 HIP training steps (full sizes) the three
    thresholded point sets of a 64^3 lattice are identical, HIP vs oracle, counts and ordered coordinates."""
    import numpy as np
    from fruitnerf_amd.data import synthetic_apple as sa
    from fruitnerf_amd.data.fruit_datamanager import ExportDataManager
    from fruitnerf_amd.export.exporter_utils import sample_volume
    from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
    from fruitnerf_amd.data.semantics import apple_metadata
    from fruitnerf_amd.rays import RayBundle
    from fruitnerf_amd.training import FusedAdam, fused_train_iteration
    HW, focal, n_train = 96, 1111.0 * 96 / 800, 40
    scene = sa.make_scene(seed=0, device=dev)
    c2w = sa.make_cameras(n_train, seed=0, device=dev)
    data = sa.render_dataset(scene, c2w, H=HW, W=HW, fx=focal, fy=focal)
    batcher = sa.PixelBatcher(data, torch.arange(n_train, device=dev), seed=1)
    torch.manual_seed(0)
    hm = FruitModel(FruitNerfModelConfig(), apple_metadata(), num_train_data=n_train, device=dev)
    hm.train()
    opt = FusedAdam(hm)
    for step in range(2000):
        o, d, cam, batch = batcher.sample(4096)
        fused_train_iteration(hm, opt, RayBundle(o, d, None, cam), batch, step, want_metrics=False)
    N = 64
    em = FruitModel(copy.deepcopy(hm.config), apple_metadata(), num_train_data=n_train, device=dev, test_mode="export")
    em.load_state_dict(hm.state_dict(), strict=True)
    em.eval()

    class Pipe:
        pass

    pipe = Pipe()
    pipe.model = em
    pipe.datamanager = ExportDataManager(dev, eval_num_rays_per_batch=1000)
    em.setup_inference(True, N, deterministic=True)
    aabb = ((-0.5, -0.5, -0.5), (0.5, 0.5, 0.5))
    n_rays = pipe.datamanager.setup_inference(aabb=aabb, num_points=N)
    got = sample_volume(pipe, n_rays, transform_json={"scale": 1.0})
    om = fo.FruitModel(fo.FruitNerfModelConfig(), num_train_data=n_train, test_mode="export")
    om.load_state_dict({k: v.detach().cpu() for k, v in hm.state_dict().items()}, strict=True)
    om.field.test_mode = "export"
    om.eval()
    om.setup_inference(True, N)
    ref = fo.sample_volume(om, aabb, N, num_rays_per_batch=1000, dataparser_scale=1.0)
    assert ref["density"]["points"].shape[0] > 500 and ref["semantic"]["points"].shape[0] > 20
    for name in ("semantic_colormap", "semantic", "density"):
        a, b = got[name]["points"], ref[name]["points"].numpy()
        print(f"[trained export] {name}: hip {a.shape[0]} oracle {b.shape[0]}")
        assert a.shape == b.shape and np.array_equal(a, b), name
    # ... and therefore the same fruit count (north star: "identical fruit count"): the first-stage count of the semantic
    # set — radius-outlier removal -> voxel down-sampling -> DBSCAN -> centre-distance merge (clustering_base.py:183-258)
    # — through the GPU front-end on the HIP export and through the CPU libraries' restatement on the oracle's export
    from fruitnerf_amd.clustering import FruitClustering, PointCloud
    from oracle import cloud as ocl
    spacing = 1.0 / N * 2.0                       # lattice pitch of the [-0.5, 0.5]^3 box after sample_volume's x2 scaling
    kw = dict(nb_points=2, radius=1.8 * spacing, voxel_size=spacing / 4, eps=1.8 * spacing, min_samples=4)
    fc = FruitClustering(voxel_size_down_sample=kw["voxel_size"], remove_outliers_nb_points=kw["nb_points"],
                         remove_outliers_radius=kw["radius"], cluster_merge_distance=0.04)
    count_hip = fc.first_stage_count(PointCloud(got["semantic"]["points"], None, dev), eps=kw["eps"], min_samples=kw["min_samples"])
    X, _, labels = ocl.cluster_front_end(ref["semantic"]["points"].numpy(), None, kw["nb_points"], kw["radius"], kw["voxel_size"],
                                         kw["eps"], kw["min_samples"])
    fc2 = FruitClustering(cluster_merge_distance=0.04)
    count_ref = 0
    if not isinstance(X, int):
        fc2.merge_small_clusters(X, None, labels)
        count_ref = fc2.counter - fc2.fuse_counter
    print(f"[trained export] first-stage fruit count: hip {count_hip} oracle {count_ref}")
    assert count_hip == count_ref and count_hip >= 1


def test_radam_matches_torch(dev):
    """fnr_radam_step vs torch.optim.RAdam (CPU) across the rectification threshold (rho_t > 5 from step 6 on for
    beta2 = 0.999), with and without weight decay."""
    from fruitnerf_amd import _kernels as K
    for wd in (0.0, 1e-3):
        g0 = torch.Generator().manual_seed(7)
        p_ref = torch.randn(4096, generator=g0).requires_grad_(True)
        opt = torch.optim.RAdam([p_ref], lr=1e-2, eps=1e-15, weight_decay=wd)
        p = p_ref.detach().clone().to(dev)
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        for step in range(1, 13):
            grad = torch.randn(4096, generator=g0) * (10.0 ** float(torch.randint(-6, 1, (1,), generator=g0)))
            p_ref.grad = grad.clone()
            opt.step()
            gd = (grad * 2.0).to(dev)
            K.radam_step(p, gd, m, v, 1e-2, 0.9, 0.999, 1e-15, step, grad_scale=0.5, zero_grad=True, weight_decay=wd)
            assert float(gd.abs().max()) == 0.0
            err = (p.cpu() - p_ref.detach()).abs().max().item()
            assert err <= 5e-6, (wd, step, err)


@pytest.mark.parametrize("algorithm", ["adam", "radam"])
def test_adam_step_spans_is_the_separate_launches(dev, algorithm):
    """fnr_adam_step_spans: three spans of one arena (own learning rate and step count each, gaps between them) in one
    launch vs one fnr_adam_step / fnr_radam_step launch per span — bit-identical parameters and moments, gradients
    zeroed inside the spans only, nothing touched outside."""
    from fruitnerf_amd import _kernels as K
    g0 = torch.Generator().manual_seed(11)
    n = 1 << 16
    spans = [(0, 540, 6e-4, 3), (1024, 16856 // 4 * 4, 1e-2, 9), (40000, 20000, 3e-3, 1)]
    state = [torch.randn(n, generator=g0), torch.randn(n, generator=g0), torch.randn(n, generator=g0) * 0.1,
             torch.rand(n, generator=g0) * 0.01]
    outs = []
    for fused in (False, True):
        p, g, m, v = [t.clone().to(dev) for t in state]
        if fused:
            K.adam_step_spans(p, g, m, v, spans, algorithm, 0.9, 0.999, 1e-15, grad_scale=0.5, zero_grad=True,
                              weight_decay=1e-3)
        else:
            fn = K.adam_step if algorithm == "adam" else K.radam_step
            for a, cnt, lr, step in spans:
                fn(p[a:a + cnt], g[a:a + cnt], m[a:a + cnt], v[a:a + cnt], lr, 0.9, 0.999, 1e-15, step,
                   grad_scale=0.5, zero_grad=True, weight_decay=1e-3)
        outs.append((p, g, m, v))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    p, g, m, v = outs[1]
    inside = torch.zeros(n, dtype=torch.bool)
    for a, cnt, _, _ in spans:
        inside[a:a + cnt] = True
    assert float(g.cpu()[inside].abs().max()) == 0.0
    for t, t0 in zip((p, g, m, v), state):
        assert torch.equal(t.cpu()[~inside], t0[~inside])
    assert not torch.equal(p.cpu()[inside], state[0][inside])


@pytest.mark.parametrize("R,n_levels,want_distortion", [(4096, 2, True), (301, 1, False), (64, 0, True)])
def test_train_losses_is_the_separate_launches(dev, R, n_levels, want_distortion):
    """fnr_train_losses (every loss and metric of a step + the slot sums in one launch) vs fnr_losses_fwd +
    fnr_interlevel_fwd per level + fnr_distortion: all gradients bit-identical, the five scalars to 1e-6 (float sums in
    a different order); repeated on a re-zeroed accumulator (completion counters start from zero each time)."""
    from fruitnerf_amd import _kernels as K, _lib as L
    g0 = torch.Generator().manual_seed(5)
    S_f, S_ps = 48, [256, 96][:n_levels]

    def level(S):
        sp = torch.sort(torch.rand(R, S + 1, generator=g0), dim=-1).values
        w = torch.rand(R, S, generator=g0)
        return sp.to(dev), (w / w.sum(-1, keepdim=True)).to(dev)
    sp_f, w_f = level(S_f)
    props = [(S,) + level(S) for S in S_ps]
    rgb, img = torch.rand(R, 3, generator=g0).to(dev), torch.rand(R, 3, generator=g0).to(dev)
    sem = torch.randn(R, 1, generator=g0).to(dev)
    msk = (torch.rand(R, 1, generator=g0) > 0.5).float().to(dev)
    ref_l, ref_drgb, ref_dsem = K.losses_fwd(rgb, img, sem, msk, 2.0)
    slots = torch.zeros(2, L.FNR_LOSS_SLOTS, device=dev)
    ref_dwp = [K.interlevel_fwd(S_f, sp_f, w_f, S, sp, w, 1.0, slots[0]) for S, sp, w in props]
    if want_distortion:
        K.distortion(S_f, sp_f, w_f, out=slots[1])
    ref_sums = slots.sum(dim=1)
    for _ in range(2):
        accum = torch.zeros(L.FNR_TRAIN_LOSSES_ACCUM_FLOATS, device=dev)
        losses, d_rgb, d_sem, d_wps = K.train_losses(rgb, img, sem, msk, 2.0, S_f, sp_f, w_f, props, 1.0,
                                                     want_distortion, accum)
        assert torch.equal(d_rgb, ref_drgb) and torch.equal(d_sem, ref_dsem)
        for a, b in zip(d_wps, ref_dwp):
            assert torch.equal(a, b)
        want = torch.stack([ref_l[0], ref_l[1], ref_l[2], ref_sums[0], ref_sums[1]])
        err = ((losses - want).abs() / want.abs().clamp_min(1e-3)).max().item()
        assert err <= 1e-6 * 5, (losses.tolist(), want.tolist())
    # the proposal levels' weights backward fused in: d(loss)/d(density) bit-identical to fnr_weights_bwd(d_wp, 1)
    if n_levels:
        full = []
        for S, sp, w in props:
            eu = torch.cumsum(torch.rand(R, S + 1, generator=g0) * 0.05, dim=-1).to(dev)
            dens = (torch.rand(R, S, generator=g0) * 30.0).to(dev)
            full.append((S, sp, w, eu, dens))
        accum = torch.zeros(L.FNR_TRAIN_LOSSES_ACCUM_FLOATS, device=dev)
        losses2, d_rgb2, d_sem2, d_dens = K.train_losses(rgb, img, sem, msk, 2.0, S_f, sp_f, w_f, full, 1.0,
                                                         want_distortion, accum, fuse_weights_bwd=True)
        assert torch.equal(d_rgb2, ref_drgb) and torch.equal(d_sem2, ref_dsem)
        one = torch.ones(1, device=dev)
        for (S, sp, w, eu, dens), dwp, got in zip(full, ref_dwp, d_dens):
            ref = K.weights_bwd(S, eu, dens, w, dwp, one)
            assert float(ref.abs().max()) > 0
            assert torch.equal(got, ref)
    # the accumulator comes back zeroed (the loss slots are self-cleaning: a training loop launches no fill per step),
    # and a second call on the buffer the first one left behind gives the same numbers
    accum = torch.zeros(L.FNR_TRAIN_LOSSES_ACCUM_FLOATS, device=dev)
    for _ in range(2):
        losses3, d_rgb3, _, _ = K.train_losses(rgb, img, sem, msk, 2.0, S_f, sp_f, w_f, props, 1.0, want_distortion, accum)
        torch.cuda.synchronize()
        assert torch.equal(d_rgb3, ref_drgb) and torch.equal(losses3, losses)
        assert float(accum.abs().max()) == 0.0


@pytest.mark.parametrize("R,S", [(4096, 48), (301, 128), (5, 17)])
def test_composite_bwd_targets_is_losses_then_composite_bwd(dev, R, S):
    """fnr_composite_bwd_targets (round 5: the per-ray MSE / BCE gradients formed inside the composite backward, so that a
    training step need not wait for the losses launch) against fnr_train_losses -> fnr_composite_bwd on the same composited
    outputs: d_density, d_rgb, d_logit bit-identical.  Densities up to 1e3 behind a surface, logits of both signs and sizes,
    ragged ray counts."""
    from fruitnerf_amd import _kernels as K, _lib as L
    g0 = torch.Generator().manual_seed(R * 31 + S)
    o, d, pa, cam = util.random_rays(R, 7, seed=3)
    rays = K.RaysArg(o.to(dev), d.to(dev), torch.full((R, 1), 0.05, device=dev), torch.full((R, 1), 1000.0, device=dev),
                     cam.to(dev))
    euclid = torch.cumsum(torch.rand(R, S + 1, generator=g0) * 0.07 + 1e-3, dim=-1).to(dev)
    density = (torch.rand(R, S, generator=g0) ** 6 * 1e3).to(dev)
    rgb_s = torch.rand(R, S, 3, generator=g0).to(dev)
    logit_s = (torch.randn(R, S, generator=g0) * 6.0).to(dev)
    weights, out_rgb, acc, depth, out_sem, labels = K.composite_fwd(rays, S, euclid, density, rgb_s, logit_s, training=True)
    image = torch.rand(R, 3, generator=g0).to(dev)
    mask = (torch.rand(R, 1, generator=g0) > 0.5).float().to(dev)
    sem_w = 2.0
    sp_f = torch.sort(torch.rand(R, S + 1, generator=g0), dim=-1).values.to(dev)
    accum = torch.zeros(L.FNR_TRAIN_LOSSES_ACCUM_FLOATS, device=dev)
    _, g_rgb, g_sem, _ = K.train_losses(out_rgb, image, out_sem, mask, sem_w, S, sp_f, weights.view(R, S), [], 1.0, False, accum)
    ref = K.composite_bwd(rays, S, euclid, density, rgb_s, weights, g_rgb, g_sem)
    got = K.composite_bwd_targets(rays, S, euclid, density, rgb_s, weights, out_rgb, image, out_sem, mask, sem_w)
    for name, a, b in zip(("d_density", "d_rgb", "d_logit"), got, ref):
        assert float(b.abs().max()) > 0, name
        assert torch.equal(a, b), f"{name}: {int((a != b).sum())} of {a.numel()} entries differ"


@pytest.mark.parametrize("R,S", [(4096, 48), (1001, 48), (64, 96), (7, 129), (33, 256)])
def test_fused_composite_forward_backward_is_the_two_launches(dev, R, S):
    """fnr_composite_fwd_bwd_targets (round 6: the compositing launch of a training step runs its own backward) against
    fnr_composite_fwd -> fnr_composite_bwd_targets: weights, every composited output and the three gradients bit-identical."""
    from fruitnerf_amd import _kernels as K
    g0 = torch.Generator().manual_seed(R * 17 + S)
    o, d, pa, cam = util.random_rays(R, 7, seed=4)
    rays = K.RaysArg(o.to(dev), d.to(dev), torch.full((R, 1), 0.05, device=dev), torch.full((R, 1), 1000.0, device=dev),
                     cam.to(dev))
    euclid = torch.cumsum(torch.rand(R, S + 1, generator=g0) * 0.07 + 1e-3, dim=-1).to(dev)
    density = (torch.rand(R, S, generator=g0) ** 6 * 1e3).to(dev)
    rgb_s = torch.rand(R, S, 3, generator=g0).to(dev)
    logit_s = (torch.randn(R, S, generator=g0) * 6.0).to(dev)
    image = torch.rand(R, 3, generator=g0).to(dev)
    mask = (torch.rand(R, 1, generator=g0) > 0.5).float().to(dev)
    sem_w = 2.0
    fwd = K.composite_fwd(rays, S, euclid, density, rgb_s, logit_s, training=True)
    weights, out_rgb, acc, depth, out_sem, labels = fwd
    bwd = K.composite_bwd_targets(rays, S, euclid, density, rgb_s, weights, out_rgb, image, out_sem, mask, sem_w)
    fwd2, bwd2 = K.composite_fwd_bwd_targets(rays, S, euclid, density, rgb_s, logit_s, image, mask, sem_w)
    for name, a, b in zip(("weights", "rgb", "accumulation", "depth", "semantics", "labels"), fwd2, fwd):
        assert torch.equal(a, b), f"{name}: {int((a != b).sum())} of {a.numel()} entries differ"
    for name, a, b in zip(("d_density", "d_rgb", "d_logit"), bwd2, bwd):
        assert float(b.abs().max()) > 0, name
        assert torch.equal(a, b), f"{name}: {int((a != b).sum())} of {a.numel()} entries differ"


def test_proposal_backward_on_a_second_stream_changes_nothing(dev):
    """training.OVERLAP_PROPOSAL_BACKWARD (FNR_OVERLAP_PROPOSAL_BACKWARD=1): the proposal-network backward runs on a
    second HIP stream underneath the field backward.  One full step (camera optimiser included, ray gradients from both
    chains) from identical states with and without it: every hash table (main + proposal), their moments and the
    poses are bit-identical, the MLP weights (float atomics) agree to 1e-6."""
    import fruitnerf_amd.training as T
    from fruitnerf_amd import _kernels as K
    from fruitnerf_amd.cameras.camera_optimizers import CameraAdam, CameraOptimizerConfig
    from fruitnerf_amd.data import synthetic_apple as sa
    from fruitnerf_amd.rays import RayBundle
    n_cam, HW, focal, R = 8, 64, 90.0, 512
    cfg = util.small_config(log2=15, prop_log2=13)
    om = util.make_oracle(cfg, num_images=n_cam, seed=5)
    scene = sa.make_scene(seed=0)
    data = {k: (v.to(dev) if torch.is_tensor(v) else v)
            for k, v in sa.render_dataset(scene, sa.make_cameras(n_cam, seed=0), H=HW, W=HW, fx=focal, fy=focal).items()}
    g = torch.Generator().manual_seed(4)
    u = torch.rand(R, 3, generator=g).to(dev)
    jit = [torch.rand(R, 1, generator=g).to(dev) for _ in range(3)]
    outs = []
    saved = T.OVERLAP_PROPOSAL_BACKWARD
    try:
        for overlap in (False, True):
            T.OVERLAP_PROPOSAL_BACKWARD = overlap
            hm = util.make_hip_like(om, dev)
            hm.train()
            opt = T.FusedAdam(hm)
            cam = CameraOptimizerConfig(mode="SO3xR3").setup(n_cam, dev)
            cadam = CameraAdam(cam)
            batcher = sa.PixelBatcher(data, torch.arange(n_cam, device=dev), seed=0)
            batcher._set = K.ImageSetArg(data["images"], data["masks"], data["c2w"], focal, focal, HW / 2.0, HW / 2.0)
            c2w_adj = cam.adjusted_cameras(batcher._set, batcher.image_ids)
            o, d, ci, image, mask = K.sample_pixels(batcher._set, batcher.image_ids, u, c2w_adj)
            batcher.last_draw = {"u": u, "cam": ci, "c2w_adjusted": c2w_adj}
            T.fused_train_iteration(hm, opt, RayBundle(o, d, None, ci[:, None]), {"image": image, "fruit_mask": mask[:, None]},
                                    0, jitter=jit, camera=(cam, cadam, batcher))
            torch.cuda.synchronize()
            tables = [hm.field.mlp_base_grid.hash_table] + [p.encoding.hash_table for p in hm.proposal_networks]
            spans = [[(off, off + k) for _, q, off, k in hm.arena().entries if q is t][0] for t in tables]
            outs.append((hm.arena().params.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(),
                         cam.pose_adjustment.data.clone(), spans))
    finally:
        T.OVERLAP_PROPOSAL_BACKWARD = saved
    a, b = outs
    for x, y in zip(a[:3], b[:3]):
        for lo, hi in a[4]:
            assert torch.equal(x[lo:hi], y[lo:hi])
        assert float((x - y).abs().max()) <= 1e-6 * max(1.0, float(x.abs().max()))
    assert torch.equal(a[3], b[3])
    fresh = util.make_hip_like(om, dev).arena().params
    for lo, hi in a[4]:
        assert int((a[0][lo:hi] != fresh[lo:hi]).sum()) > 0      # every table moved (proposal nets train at step 0)

