"""Parity at TRAINED states, held to the north-star bar (RGB + semantic outputs within 1e-4 of the reference CPU path,
/root/reference/fruit_nerf/fruit_nerf.py:316-357) — the regime the random-weight tests cannot reach: sharp densities
(delta*sigma up to 1e8), saturated sigmoids, peaky proposal PDFs.

Both built method shapes are trained on the HIP path at their REAL configuration (fruit_nerf: T = 2^19, max_res 2048,
256/96/48 samples, anneal 1000; fruit_nerf_big: T = 2^21, max_res 4096, 512/256/128 samples, geo 30, semantic MLP
3 x 128, anneal 5000, RAdam — fruit_nerf_config.py:27-110), the weights are loaded into the CPU oracle and

  * an eval-mode forward of 4096 rays is compared per ray, end to end AND stage by stage: (i) the sample bins each
    proposal level produced, (ii) field + renderers on IDENTICAL final samples (the oracle's bins fed to the HIP
    kernels) — where the 1e-4 bar must hold for every ray, (iii) end to end, where a ray's output additionally carries
    the fp32 noise of the proposal networks through the inverse-CDF sampler (the same noise two CPUs with different
    BLAS kernels would show); outliers are printed with the sample that moved;
  * (fruit_nerf_big) one more training step is replayed in the oracle: losses and every gradient."""
import pytest
import torch

from oracle import fruit_oracle as fo
from oracle import ns_torch as ns
from tests import util

pytestmark = pytest.mark.gpu

BAR = 1e-4          # north star: RGB + semantic outputs within 1e-4
# one-step replays at a trained state (end to end: the HIP path samples for itself).  <= 5x what round 3 measured:
LOSS_REL = 1e-2     # losses / metrics, relative (measured <= 2.5e-3)
GRAD_L1 = 1e-2      # sum|hip - oracle| / sum|oracle| per parameter tensor (measured <= 2.5e-3)
GRAD_MAX = 1e-2     # max|hip - oracle| / max|oracle| per parameter tensor (measured <= 1.9e-3)
RAYS_OFF = 5e-3     # fraction of rays whose origin / direction gradient is off by > 1 % of the largest (measured 0)
SAME_SAMPLES_GRAD = 5e-4   # every gradient entry within this of max|g| when both sides use the oracle's sample bins

SHAPES = {
    # name: (oracle config factory, rays / step, training steps, optimiser, group lr table)
    "fruit_nerf": (util.full_config, 4096, 2500, "adam", None),
    "fruit_nerf_big": (util.fruit_nerf_big_config, 8192, 1200, "radam",
                       {"proposal_networks": dict(lr=1e-2, lr_final=None, max_steps=None),
                        "fields": dict(lr=1e-2, lr_final=1e-4, max_steps=50000)}),
}
_CACHE = {}


def _scene(dev):
    from fruitnerf_amd.data import synthetic_apple as sa
    HW, focal, n_train = 96, 1111.0 * 96 / 800, 40
    scene = sa.make_scene(seed=0, device=dev)
    c2w = sa.make_cameras(n_train, seed=0, device=dev)
    data = sa.render_dataset(scene, c2w, H=HW, W=HW, fx=focal, fy=focal)
    return sa, data, n_train


def _trained(dev, shape):
    """HIP model of `shape` trained on the synthetic scene at its real configuration (cached per test session),
    its optimiser, batcher and the matching oracle config."""
    if shape in _CACHE:
        return _CACHE[shape]
    from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
    from fruitnerf_amd.data.semantics import apple_metadata
    from fruitnerf_amd.rays import RayBundle
    from fruitnerf_amd.training import FusedAdam, fused_train_iteration
    make_cfg, rays, steps, algo, group_lr = SHAPES[shape]
    ocfg = make_cfg()
    sa, data, n_train = _scene(dev)
    batcher = sa.PixelBatcher(data, torch.arange(n_train, device=dev), seed=1)
    cfg = FruitNerfModelConfig()
    for k, v in vars(ocfg).items():
        if hasattr(cfg, k):
            setattr(cfg, k, v)
    torch.manual_seed(0)
    hm = FruitModel(cfg, apple_metadata(), num_train_data=n_train, device=dev)
    hm.train()
    opt = FusedAdam(hm, algorithm=algo, group_lr=group_lr)
    for step in range(steps):
        o, d, cam, batch = batcher.sample(rays)
        ld, _ = fused_train_iteration(hm, opt, RayBundle(o, d, None, cam), batch, step, want_metrics=False)
    torch.cuda.synchronize()
    print(f"[trained {shape}] {steps} steps x {rays} rays: " + " ".join(f"{k} {float(v):.3e}" for k, v in ld.items()))
    assert float(ld["rgb_loss"]) < 5e-3, "the model did not train"
    _CACHE[shape] = (hm, opt, batcher, ocfg, n_train, steps)
    return _CACHE[shape]


def _oracle_of(hm, ocfg, n_train):
    om = fo.FruitModel(ocfg, num_train_data=n_train)
    om.load_state_dict({k: v.detach().cpu() for k, v in hm.state_dict().items()}, strict=True)
    return om


def _bins(rs):
    """[R, S+1] euclidean / spacing bin edges of an oracle RaySamples."""
    e = torch.cat([rs.frustums.starts[..., 0], rs.frustums.ends[:, -1:, 0]], dim=-1)
    s = torch.cat([rs.spacing_starts[..., 0], rs.spacing_ends[:, -1:, 0]], dim=-1)
    return e.contiguous(), s.contiguous()


def _per_ray_err(got, ref):
    """max over a ray's channels of |got - ref| / max(1, |ref|) -> [R]"""
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return ((got - ref).abs() / ref.abs().clamp_min(1.0)).reshape(ref.shape[0], -1).max(dim=1)[0]


@pytest.mark.parametrize("shape", ["fruit_nerf", "fruit_nerf_big"])
def test_eval_outputs_at_a_trained_state_meet_the_output_bar(dev, shape):
    from fruitnerf_amd import _kernels as K
    from fruitnerf_amd.rays import RayBundle
    hm, opt, batcher, ocfg, n_train, steps = _trained(dev, shape)
    R = 4096
    o, d, cam, batch = batcher.sample(R)
    hm.eval()
    hm.set_anneal(steps)         # the sampler keeps the exponent its last BEFORE_TRAIN_ITERATION callback set
    with torch.no_grad():        # (fruit_nerf_big anneals over 5000 iterations: 0.76 here, not the constructor's 1.0)
        hout = hm(RayBundle(o, d, None, cam))
    torch.cuda.synchronize()
    om = _oracle_of(hm, ocfg, n_train)
    om.eval()
    om.set_anneal(steps)
    with torch.no_grad():
        oout = om(ns.RayBundle(o.cpu(), d.cpu(), torch.ones(R, 1), camera_indices=cam.cpu().long()))
    ctx = hout["_ctx"]

    # ---- (i) the sampler chain: bins of every level, HIP vs oracle (spacing domain: [0, 1]) -------------------------
    worst_bin = []
    for i, (lv, rs) in enumerate(zip(ctx.levels, oout["ray_samples_list"])):
        e_ref, s_ref = _bins(rs)
        ds = (lv["spacing"].cpu() - s_ref).abs().max(dim=1)[0]
        worst_bin.append(ds)
        print(f"[trained {shape}] level {i}: S {lv['S']}  max |spacing bin - oracle| {ds.max().item():.3e}  "
              f"median over rays {ds.median().item():.3e}  rays > 1e-5: {(ds > 1e-5).float().mean().item():.4%}")
    # level 0 is the deterministic piecewise spacing: identical up to rounding
    assert worst_bin[0].max().item() <= 2e-6

    # ---- (ii) field + renderers on IDENTICAL final samples: the oracle's bins through the HIP kernels ---------------
    e_ref, s_ref = _bins(oout["ray_samples_list"][-1])
    S = ctx.levels[-1]["S"]
    with torch.no_grad():
        rays = K.RaysArg(o, d, None, None, cam)
        euclid = e_ref.to(dev).contiguous()
        fld = hm.field
        net = fld.net_struct()
        feats, selector = K.hash_encode_fwd(net.grid, fld.warp_struct(), rays, euclid, S)
        density, rgb, logit, _, _ = K.field_mlp_fwd(net, rays, S, feats, selector, fld._mean_embedding(), want_h=True)
        weights, out_rgb, acc, depth, sem, label = K.composite_fwd(rays, S, euclid, density, rgb, logit, False)
    torch.cuda.synchronize()
    same = {"rgb": _per_ray_err(out_rgb, oout["rgb"]), "semantics": _per_ray_err(sem[:, None], oout["semantics"]),
            "accumulation": _per_ray_err(acc[:, None], oout["accumulation"])}
    for k, v in same.items():
        print(f"[trained {shape}] same samples  {k}: max {v.max().item():.3e}  rays > {BAR:g}: {(v > BAR).sum().item()} / {R}")
    w_ref = oout["weights_list"][-1][..., 0]
    print(f"[trained {shape}] same samples  weights: max abs {((weights.cpu() - w_ref).abs()).max().item():.3e}; "
          f"largest delta*sigma {float((density.view(R, S) * (euclid[:, 1:] - euclid[:, :-1])).max()):.3e}")
    for k, v in same.items():
        assert v.max().item() <= BAR, f"{k}: field + renderers on identical samples must meet the bar for every ray"
    assert torch.equal(label.cpu().long().reshape(-1), oout["semantics_colormap"].reshape(-1)) or \
        (label.cpu().long().reshape(-1) != oout["semantics_colormap"].reshape(-1)).float().mean().item() <= 1e-3

    # ---- (iii) end to end ------------------------------------------------------------------------------------------
    errs = {k: _per_ray_err(hout[k], oout[k]) for k in ("rgb", "semantics", "accumulation")}
    worst = torch.stack(list(errs.values())).max(dim=0)[0]
    frac_ok = (worst <= BAR).float().mean().item()
    for k, v in errs.items():
        print(f"[trained {shape}] end to end  {k}: max {v.max().item():.3e}  median {v.median().item():.3e}  "
              f"rays > {BAR:g}: {(v > BAR).sum().item()} / {R}  > 1e-3: {(v > 1e-3).sum().item()}  > 1e-2: {(v > 1e-2).sum().item()}")
    print(f"[trained {shape}] end to end  rays with every output within {BAR:g}: {frac_ok:.4%}")
    # the outliers, with the sample that moved: the final-level bin that differs most and what the density does there
    e_h = ctx.levels[-1]["euclid"].cpu()
    dens_ref = None
    for r in worst.argsort(descending=True)[:6].tolist():
        k = int((e_h[r] - e_ref[r]).abs().argmax())
        if dens_ref is None:
            with torch.no_grad():
                dens_ref = (density.view(R, S)).cpu()
        kk = min(k, S - 1)
        print(f"   ray {r}: err {worst[r].item():.3e}  final-level bin {k} moved by {(e_h[r, k] - e_ref[r, k]).item():+.3e} "
              f"(t = {e_ref[r, k].item():.5f}); density there {dens_ref[r, kk].item():.3e}, next {dens_ref[r, min(kk + 1, S - 1)].item():.3e}; "
              f"level-1/2 spacing-bin error {worst_bin[1][r].item():.2e} / {worst_bin[2][r].item():.2e}")
    # Residual: a ray is off only where a last-bit difference of a proposal density moved a PDF sample
    # across a density jump of orders of magnitude — the reference on another CPU / BLAS shows the same noise.
    assert frac_ok >= 0.995, f"only {frac_ok:.4%} of the rays within {BAR:g} end to end"
    assert worst.median().item() <= 5e-6 and worst.max().item() <= 5e-3
    hm.train()


def test_fruit_nerf_big_step_at_a_trained_state_matches_the_oracle(dev):
    """fruit_nerf_big at its real configuration after 1200 HIP training steps: one more step (an 'updated' one, so the
    proposal networks get gradients) replayed in the CPU oracle with the same weights, rays, jitter and batch."""
    from fruitnerf_amd.rays import RayBundle
    from fruitnerf_amd.training import fused_forward_backward
    hm, opt, batcher, ocfg, n_train, steps = _trained(dev, "fruit_nerf_big")
    hm.train()
    R = 512
    o, d, cam, batch = batcher.sample(R)
    jit = [torch.rand(R, 1, device=dev) for _ in range(3)]
    hm.set_anneal(steps)
    samp = hm.proposal_sampler
    samp._steps_since_update = 100
    state = (samp._step, samp._steps_since_update)
    hm.arena().grads.zero_()
    ray_grads = {}
    ld, md = fused_forward_backward(hm, RayBundle(o, d, None, cam), batch, jitter=jit, ray_grads=ray_grads)
    torch.cuda.synchronize()

    om = _oracle_of(hm, ocfg, n_train)
    om.train()
    om.proposal_sampler._step, om.proposal_sampler._steps_since_update = state
    om.set_anneal(steps)
    o_ref, d_ref = o.cpu().clone().requires_grad_(True), d.cpu().clone().requires_grad_(True)
    out = om(ns.RayBundle(o_ref, d_ref, torch.ones(R, 1), camera_indices=cam.cpu().long()), jitter=[j.cpu() for j in jit])
    b = {k: v.cpu() for k, v in batch.items()}
    ld_ref = om.get_loss_dict(out, b)
    md_ref = om.get_metrics_dict(out, b)
    sum(ld_ref.values()).backward()
    for k in ld_ref:
        a, r = float(ld[k]), float(ld_ref[k])
        print(f"[trained big] {k}: hip {a:.8e} oracle {r:.8e} rel {abs(a - r) / max(abs(r), 1e-12):.2e}")
        assert abs(a - r) <= LOSS_REL * max(abs(r), 1e-4), k     # measured (round 3): <= 2.5e-3
    for k in md_ref:
        a, r = float(md[k]), float(md_ref[k])
        print(f"[trained big] {k}: hip {a:.8e} oracle {r:.8e}")
        assert abs(a - r) <= LOSS_REL * max(abs(r), 1e-3), k
    named_h = dict(hm.named_parameters())
    for name, p in util.named_trainable(om):
        ref = p.grad if p.grad is not None else torch.zeros_like(p)
        got = named_h[name].grad.detach().cpu()
        denom = ref.abs().double().sum().item()
        agg = (got - ref).abs().double().sum().item() / max(denom, 1e-30)
        mx = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
        print(f"[trained big] grad {name}: max|ref| {ref.abs().max().item():.3e} max-norm rel {mx:.3e} L1-rel {agg:.3e}")
        # measured (round 3, profiles/r03_raw/gputest_final.log): worst L1 2.5e-3, worst max-norm 1.9e-3 -> bars at ~5x
        assert denom == 0 and float(got.abs().sum()) == 0 or (agg <= GRAD_L1 and mx <= GRAD_MAX), \
            f"{name}: gradient error L1-rel {agg} max-norm rel {mx}"
    for name, got_g, ref_g in (("origins", ray_grads["origins"], o_ref.grad), ("directions", ray_grads["directions"], d_ref.grad)):
        diff = (got_g.cpu() - ref_g).abs()
        scale = ref_g.abs().max().item()
        per_ray = diff.max(dim=1)[0]
        off = (per_ray > 1e-2 * scale).float().mean().item()
        print(f"[trained big] d loss / d {name}: max|ref| {scale:.3e} max_err {diff.max().item():.3e} rays off {off:.2%}")
        assert off <= RAYS_OFF, name


def _bundle_on_oracle_bins(hm, o, d, cam, oout, updated, anneal):
    """A HIP RayBundle whose proposal sampling is replaced by the ORACLE's bins at every level (through the look-ahead
    slot `presampled["ahead"]`, the mechanism TrainingSteps uses): the HIP proposal networks are evaluated on the
    oracle's level-0 / level-1 bins, their weights are formed by the HIP weights kernel, the inverse-CDF bins it also
    produces are thrown away in favour of the oracle's.  Both sides then differentiate the losses at IDENTICAL sample
    positions, so what remains between their gradients is the arithmetic of the backward kernels alone."""
    from fruitnerf_amd import _kernels as K
    from fruitnerf_amd.rays import RayBundle
    dev = o.device
    cfg = hm.config
    sampler = hm.proposal_sampler
    n_prop = sampler.num_proposal_network_iterations
    R = o.shape[0]
    rb = hm._collide(RayBundle(o, d, None, cam))
    rays = K.RaysArg(rb.origins, rb.directions, rb.nears, rb.fars, rb.camera_indices)
    bins = [tuple(t.to(dev).contiguous() for t in _bins(rs)) for rs in oout["ray_samples_list"]]   # (euclid, spacing)
    counts = list(sampler.num_proposal_samples_per_ray[:n_prop]) + [sampler.num_nerf_samples_per_ray]
    zero = torch.zeros(R, device=dev)
    levels = []
    with torch.no_grad():
        for i in range(n_prop):
            net = hm.proposal_networks[i]
            euclid, spacing = bins[i]
            density, feats = K.prop_density_fwd(net.prop_struct(), net.warp_struct(), rays, euclid, counts[i],
                                                save_feats=updated)
            weights, depth, _, _ = K.weights_pdf(rays, 1, counts[i], counts[i + 1], density, spacing, euclid, anneal, zero)
            levels.append(dict(S=counts[i], spacing=spacing, euclid=euclid, density=density, weights=weights, depth=depth,
                               feats=feats))
    pre = dict(S0=counts[0], jitter=[zero] * (n_prop + 1), spacing=bins[0][1], euclid=bins[0][0],
               near=float(cfg.near_plane), far=float(cfg.far_plane),
               ahead=dict(levels=levels, spacing=bins[-1][1], euclid=bins[-1][0], S=counts[-1], updated=updated,
                          anneal=anneal))
    return RayBundle(o, d, None, cam, presampled=pre)


@pytest.mark.parametrize("shape", ["fruit_nerf", "fruit_nerf_big"])
def test_gradients_on_identical_samples_at_a_trained_state(dev, shape):
    """The gradient leg of 'field + renderers on IDENTICAL samples' (VERDICT r03, weak #2): one training step (an
    'updated' one: the proposal networks get gradients too) at the trained state, the oracle sampling for itself and the
    HIP path differentiating at the oracle's bins.  Bar: every entry of every parameter gradient within 5e-4 of that
    tensor's max |g| (the round-1 bar of the random-weight tests, now at sharp densities and saturated sigmoids), losses
    and metrics to 1e-4 relative; the ray gradients (camera optimiser) are reported and held to 5e-3 of their max."""
    from fruitnerf_amd.training import fused_forward_backward
    hm, opt, batcher, ocfg, n_train, steps = _trained(dev, shape)
    hm.train()
    R = 768 if shape == "fruit_nerf" else 512
    o, d, cam, batch = batcher.sample(R)
    jit = [torch.rand(R, 1, device=dev) for _ in range(3)]
    samp = hm.proposal_sampler
    hm.set_anneal(steps)
    samp._steps_since_update = 100
    state = (samp._step, samp._steps_since_update)

    om = _oracle_of(hm, ocfg, n_train)
    om.train()
    om.proposal_sampler._step, om.proposal_sampler._steps_since_update = state
    om.set_anneal(steps)
    o_ref, d_ref = o.cpu().clone().requires_grad_(True), d.cpu().clone().requires_grad_(True)
    out = om(ns.RayBundle(o_ref, d_ref, torch.ones(R, 1), camera_indices=cam.cpu().long()), jitter=[j.cpu() for j in jit])
    b = {k: v.cpu() for k, v in batch.items()}
    ld_ref = om.get_loss_dict(out, b)
    md_ref = om.get_metrics_dict(out, b)
    sum(ld_ref.values()).backward()

    assert samp.updated_now()
    rb = _bundle_on_oracle_bins(hm, o, d, cam, out, True, samp._anneal)
    hm.arena().grads.zero_()
    used = hm.__dict__.get("_ahead_used", 0)
    ray_grads = {}
    ld, md = fused_forward_backward(hm, rb, batch, ray_grads=ray_grads)
    torch.cuda.synchronize()
    assert hm.__dict__.get("_ahead_used", 0) == used + 1, "the HIP pass did not run on the oracle's bins"
    for k in ld_ref:
        a, r = float(ld[k]), float(ld_ref[k])
        print(f"[same samples {shape}] {k}: hip {a:.8e} oracle {r:.8e} rel {abs(a - r) / max(abs(r), 1e-12):.2e}")
        # 1e-4 relative + 5e-8 absolute: at a trained state the semantic loss is a MEAN of 6e-7 .. 6e-5 over rays whose
        # logits sit at +-15 .. 30, where the two BCE-with-logits formulas (torch: (1 - z) x + m + log(e^-m + e^(-x-m));
        # HIP: max(x, 0) - x z + log1p(e^-|x|)) round intermediates of size |x|: ~1e-6 per saturated ray (measured
        # round 4: |hip - oracle| = 1.2e-8 .. 1.7e-8 on the mean, rgb and interlevel losses 3e-7 .. 5e-7 relative)
        assert abs(a - r) <= 1e-4 * abs(r) + 5e-8, k
    for k in md_ref:
        a, r = float(md[k]), float(md_ref[k])
        print(f"[same samples {shape}] {k}: hip {a:.8e} oracle {r:.8e}")
        assert abs(a - r) <= 1e-4 * max(abs(r), 1e-3), k
    named_h = dict(hm.named_parameters())
    worst = ("", 0.0)
    ref_grads = {name: (p.grad if p.grad is not None else torch.zeros_like(p)) for name, p in util.named_trainable(om)}
    for name, p in util.named_trainable(om):
        ref = ref_grads[name]
        got = named_h[name].grad.detach().cpu()
        scale = ref.abs().max().item()
        if ref.numel() == 1 and name.endswith(".bias"):
            # a scalar output bias is ONE sum of signed per-sample terms (d loss / d out) that nearly cancel (|g| ~ 1e-6
            # where the layer's weight gradient, the same terms weighted by O(1) activations, is ~5e-5): its own magnitude
            # is not the scale of its rounding error, the layer's weight gradient is
            scale = max(scale, ref_grads[name[:-len("bias")] + "weight"].abs().max().item())
        mx = (got - ref).abs().max().item() / max(scale, 1e-30)
        agg = (got - ref).abs().double().sum().item() / max(ref.abs().double().sum().item(), 1e-30)
        print(f"[same samples {shape}] grad {name}: max|ref| {scale:.3e} max-norm rel {mx:.3e} L1-rel {agg:.3e}")
        if scale == 0:
            assert float(got.abs().sum()) == 0, name
            continue
        worst = max(worst, (name, mx), key=lambda t: t[1])
        assert mx <= SAME_SAMPLES_GRAD, f"{name}: {mx:.3e} of max|g| on identical samples"
    print(f"[same samples {shape}] worst gradient: {worst[0]} {worst[1]:.3e} of max|g| (bar {SAME_SAMPLES_GRAD:g})")
    for name, got_g, ref_g in (("origins", ray_grads["origins"], o_ref.grad), ("directions", ray_grads["directions"], d_ref.grad)):
        diff = (got_g.cpu() - ref_g).abs()
        scale = ref_g.abs().max().item()
        off = (diff.max(dim=1)[0] > 5e-3 * scale).float().mean().item()
        print(f"[same samples {shape}] d loss / d {name}: max|ref| {scale:.3e} max_err {diff.max().item():.3e} "
              f"({diff.max().item() / scale:.2e} of max) rays beyond 5e-3: {off:.2%}")
        assert off <= RAYS_OFF, name
    hm.arena().grads.zero_()
