"""GPU parity: HIP forward path (through the C ABI) vs the CPU oracle, stage by stage and end to end.

Tolerances (north_star: RGB + semantic outputs within 1e-4 of the reference CPU path):
  * hash-grid features, selector, sampler bins: bit-exact / 1e-6 (same op order, non-contracted fp32)
  * densities: relative 2e-5 (exp amplifies the fp32 summation-order difference of the MFMA chain)
  * rgb / logit per sample and composited: absolute 1e-4
"""
import pytest
import torch

from oracle import fruit_oracle as fo
from oracle import ns_torch as ns
from tests import util

pytestmark = pytest.mark.gpu


def _bundle(o, d, pa, cam, near=None, far=None):
    R = o.shape[0]
    nears = None if near is None else torch.full((R, 1), near)
    fars = None if far is None else torch.full((R, 1), far)
    return ns.RayBundle(o.clone(), d.clone(), pa.clone(), camera_indices=cam.clone(), nears=nears, fars=fars)


def _hip_bundle(o, d, pa, cam, dev, near=None, far=None):
    from fruitnerf_amd.rays import RayBundle
    R = o.shape[0]
    nears = None if near is None else torch.full((R, 1), near, device=dev)
    fars = None if far is None else torch.full((R, 1), far, device=dev)
    return RayBundle(o.to(dev), d.to(dev), pa.to(dev), cam.to(dev), nears, fars)


def test_device_is_gfx950():
    from fruitnerf_amd import _lib as L
    info = L.device_check()
    print("device:", info)
    assert info["arch"].startswith("gfx950") and info["cus"] > 0


@pytest.mark.parametrize("mode", ["contract", "aabb"])
def test_hash_encode_bit_exact(dev, mode):
    from fruitnerf_amd import _kernels as K
    torch.manual_seed(0)
    L_, log2 = 16, 15
    enc = ns.HashEncoding(num_levels=L_, min_res=16, max_res=2048, log2_hashmap_size=log2)
    with torch.no_grad():
        enc.hash_table.copy_(torch.rand_like(enc.hash_table) * 2 - 1)
    R, S = 300, 7
    o, d, pa, cam = util.random_rays(R, 4, seed=3)
    euclid = torch.sort(torch.rand(R, S + 1) * 3.0, dim=-1).values
    pos = o[:, None, :] + d[:, None, :] * (euclid[:, :-1, None] + euclid[:, 1:, None]) / 2
    aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    if mode == "contract":
        x = (ns.SceneContraction()(pos) + 2.0) / 4.0
    else:
        x = ns.get_normalized_positions(pos, aabb)
    sel = ((x > 0.0) & (x < 1.0)).all(dim=-1)
    x = x * sel[..., None]
    ref = enc(x.view(-1, 3))  # [N, 32]
    grid = K.make_grid(enc.hash_table.data.to(dev).contiguous(), L_, log2, [int(v) for v in enc.scalings.tolist()])
    rays = K.RaysArg(o.to(dev), d.to(dev), None, None)
    warp = K.make_warp(0 if mode == "contract" else 1, aabb)
    table_dev = enc.hash_table.data.to(dev).contiguous()
    grid.table = table_dev.data_ptr()
    feats, selector = K.hash_encode_fwd(grid, warp, rays, euclid.to(dev).contiguous(), S)
    got = feats.permute(1, 0, 2).reshape(R * S, 2 * L_).cpu()
    util.report(f"hash_encode[{mode}]", got, ref)
    assert torch.equal(selector.cpu().bool(), sel.view(-1)), "selector mask differs"
    assert torch.equal(got, ref), "hash-grid features are not bit-exact"


def test_sampler_chain_and_prop_density(dev):
    """level-0 piecewise bins, proposal density, get_weights + PDF resampling (train jitter and eval)."""
    from fruitnerf_amd import _kernels as K
    cfg = util.small_config()
    om = util.make_oracle(cfg, seed=1)
    hm = util.make_hip_like(om, dev)
    R = 257
    o, d, pa, cam = util.random_rays(R, 7, seed=5)
    for training in (True, False):
        om.train(training)
        near = 0.05 if training else 0.0
        rb = _bundle(o, d, pa, cam, near, 1000.0)
        jit = [torch.rand(R, 1), torch.rand(R, 1)] if training else [None, None]
        samp0 = ns.UniformLinDispPiecewiseSampler(single_jitter=True)
        samp0.train(training)
        rs0 = samp0(rb, num_samples=256, t_rand=jit[0])
        with torch.no_grad():
            dens0 = om.proposal_networks[0].density_fn(rs0.frustums.get_positions())
        w0 = rs0.get_weights(dens0)
        pdf = ns.PDFSampler(include_original=False, single_jitter=True)
        pdf.train(training)
        anneal = 0.37
        rs1 = pdf(rb, rs0, torch.pow(w0, anneal), num_samples=96, rand=jit[1])

        rays = K.RaysArg(o.to(dev), d.to(dev), torch.full((R,), near, device=dev),
                         torch.full((R,), 1000.0, device=dev))
        sp, eu = K.sample_spaced(rays, 1, 256, None if jit[0] is None else jit[0].to(dev))
        ref_sp = torch.cat([rs0.spacing_starts[..., 0], rs0.spacing_ends[..., -1:, 0]], -1)
        ref_eu = torch.cat([rs0.frustums.starts[..., 0], rs0.frustums.ends[..., -1:, 0]], -1)
        a, _ = util.report(f"spacing0[train={training}]", sp, ref_sp)
        _, r = util.report(f"euclid0[train={training}]", eu, ref_eu)
        assert a <= 1e-7 and r <= 1e-5
        net = hm.proposal_networks[0]
        hm.arena()
        dens, _ = K.prop_density_fwd(net.prop_struct(), net.warp_struct(), rays, eu, 256)
        _, r = util.report(f"prop_density0[train={training}]", dens, dens0[..., 0])
        assert r <= 5e-5
        # feed the ORACLE's density/bins so the comparison isolates weights + PDF sampling
        w, depth, sp1, eu1 = K.weights_pdf(rays, 1, 256, 96, dens0[..., 0].to(dev).contiguous(),
                                           ref_sp.to(dev).contiguous(), ref_eu.to(dev).contiguous(), anneal,
                                           None if jit[1] is None else jit[1].to(dev))
        a, _ = util.report(f"weights0[train={training}]", w, w0[..., 0])
        assert a <= 2e-6
        ref_sp1 = torch.cat([rs1.spacing_starts[..., 0], rs1.spacing_ends[..., -1:, 0]], -1)
        ref_eu1 = torch.cat([rs1.frustums.starts[..., 0], rs1.frustums.ends[..., -1:, 0]], -1)
        a, _ = util.report(f"pdf_spacing1[train={training}]", sp1, ref_sp1)
        assert a <= 2e-5
        _, r = util.report(f"pdf_euclid1[train={training}]", eu1, ref_eu1)
        assert r <= 2e-3  # 1/(2-2s) amplifies spacing error near s -> 1 (far plane 1000)
        ref_depth = ns.render_depth_median(w0, rs0)
        a, r = util.report(f"prop_depth0[train={training}]", depth, ref_depth[..., 0])
        # median index can flip between adjacent samples when cumsum ~ 0.5; count mismatches instead
        bad = ((depth.cpu() - ref_depth[..., 0]).abs() > 1e-4 * ref_depth[..., 0].abs().clamp_min(1)).sum().item()
        assert bad <= 1, f"{bad} median-depth mismatches"


@pytest.mark.parametrize("shape", ["fruit_nerf", "fruit_nerf_big"])
@pytest.mark.parametrize("training", [True, False])
def test_field_forward_per_sample(dev, training, shape):
    """FruitField.forward on generic RaySamples: density / rgb / semantic logit per sample, for both built MLP shapes."""
    cfg = util.small_config(log2=16) if shape == "fruit_nerf" else util.big_config(log2=16)
    om = util.make_oracle(cfg, seed=2)
    hm = util.make_hip_like(om, dev)
    om.train(training)
    hm.train(training)
    R, S = 96, 48
    o, d, pa, cam = util.random_rays(R, 7, seed=7)
    rb = _bundle(o, d, pa, cam)
    euclid = torch.sort(torch.rand(R, S + 1) * 1.6 + 0.2, dim=-1).values
    rs = rb.get_ray_samples(euclid[:, :-1, None], euclid[:, 1:, None])
    with torch.no_grad():
        ref = om.field(rs)
    from fruitnerf_amd.rays import RayBundle
    hb = RayBundle(o.to(dev), d.to(dev), pa.to(dev), cam.to(dev))
    hs = hb.get_ray_samples(euclid[:, :-1, None].to(dev), euclid[:, 1:, None].to(dev))
    from fruitnerf_amd.fruit_field import FieldHeadNames
    got = hm.field(hs)
    _, r = util.report(f"field.density[train={training}]", got[FieldHeadNames.DENSITY], ref["density"])
    a1, _ = util.report(f"field.rgb[train={training}]", got[FieldHeadNames.RGB], ref["rgb"])
    a2, _ = util.report(f"field.semantics[train={training}]", got[FieldHeadNames.SEMANTICS], ref["semantics"])
    assert r <= 2e-5 and a1 <= 1e-4 and a2 <= 1e-4
    dens, geo = hm.field.get_density(hs)
    with torch.no_grad():
        rd, rg = om.field.get_density(rs)
    a, _ = util.report("field.geo", geo, rg)
    assert a <= 1e-5 and geo.shape[-1] == cfg.geo_feat_dim


@pytest.mark.parametrize("training", [False, True])
def test_model_forward_end_to_end(dev, training):
    """FruitModel.forward: sampler chain + field + renderers, full-size `fruit_nerf` grids."""
    cfg = util.full_config()
    om = util.make_oracle(cfg, seed=3)
    hm = util.make_hip_like(om, dev)
    om.train(training)
    hm.train(training)
    om.proposal_sampler._anneal = 0.6
    hm.proposal_sampler._anneal = 0.6
    R = 192
    o, d, pa, cam = util.random_rays(R, 7, seed=11)
    jit = [torch.rand(R, 1) for _ in range(3)] if training else None
    with torch.no_grad():
        ref = om(_bundle(o, d, pa, cam), jitter=jit)
    with torch.no_grad():
        got = hm(_hip_bundle(o, d, pa, cam, dev), jitter=None if jit is None else [j.to(dev) for j in jit])
    a_rgb, _ = util.report(f"model.rgb[train={training}]", got["rgb"], ref["rgb"])
    a_sem, _ = util.report(f"model.semantics[train={training}]", got["semantics"], ref["semantics"])
    a_acc, _ = util.report(f"model.accumulation[train={training}]", got["accumulation"], ref["accumulation"])
    util.report(f"model.depth[train={training}]", got["depth"], ref["depth"])
    for i in range(3):
        util.report(f"model.weights[{i}]", got["weights_list"][i], ref["weights_list"][i])
    assert a_rgb <= 1e-4 and a_sem <= 1e-4 and a_acc <= 1e-4
    bad = ((got["depth"].cpu() - ref["depth"]).abs() > 1e-3 * ref["depth"].abs().clamp_min(1)).sum().item()
    assert bad <= 2, f"{bad} median-depth mismatches"
    assert torch.equal(got["semantics_colormap"].cpu().float().view(-1), ref["semantics_colormap"].float().view(-1))


@pytest.mark.parametrize("training", [False, True])
def test_model_forward_end_to_end_fruit_nerf_big(dev, training):
    """FruitModel.forward at the real `fruit_nerf_big` sizes (fruit_nerf_config.py:82-95): hash 16 x 2^21, max_res 4096,
    samples 512/256/128, geo 30, semantic 30 -> 128 -> 128 -> 64."""
    cfg = util.fruit_nerf_big_config()
    om = util.make_oracle(cfg, seed=13)
    hm = util.make_hip_like(om, dev)
    om.train(training)
    hm.train(training)
    om.proposal_sampler._anneal = 0.4
    hm.proposal_sampler._anneal = 0.4
    R = 80
    o, d, pa, cam = util.random_rays(R, 7, seed=17)
    jit = [torch.rand(R, 1) for _ in range(3)] if training else None
    with torch.no_grad():
        ref = om(_bundle(o, d, pa, cam), jitter=jit)
        got = hm(_hip_bundle(o, d, pa, cam, dev), jitter=None if jit is None else [j.to(dev) for j in jit])
    a_rgb, _ = util.report(f"big.rgb[train={training}]", got["rgb"], ref["rgb"])
    a_sem, _ = util.report(f"big.semantics[train={training}]", got["semantics"], ref["semantics"])
    a_acc, _ = util.report(f"big.accumulation[train={training}]", got["accumulation"], ref["accumulation"])
    for i in range(3):
        assert got["weights_list"][i].shape == ref["weights_list"][i].shape == (R, (512, 256, 128)[i], 1)
        util.report(f"big.weights[{i}]", got["weights_list"][i], ref["weights_list"][i])
    assert a_rgb <= 1e-4 and a_sem <= 1e-4 and a_acc <= 1e-4
    assert torch.equal(got["semantics_colormap"].cpu().float().view(-1), ref["semantics_colormap"].float().view(-1))


def test_pixel_sampler_matches_the_oracle(dev):
    """fnr_sample_pixels (PixelSampler + RayGenerator) vs oracle/pixel_sampler.py (pinned on closed-form pinhole geometry,
    tests/test_oracle_closed_form.py; what bench.py's CPU baseline draws its rays with)."""
    from fruitnerf_amd.data import synthetic_apple as sa
    from oracle import pixel_sampler as ops
    scene = sa.make_scene(seed=0)
    c2w = sa.make_cameras(6, seed=0)
    data = sa.render_dataset(scene, c2w, H=48, W=40, fx=61.0, fy=59.0)
    ids = torch.tensor([0, 2, 3, 5])
    gdata = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data.items()}
    u = torch.rand(5000, 3)
    u[0] = torch.tensor([0.999999, 0.999999, 0.999999])
    u[1] = 0.0
    o, d, cam, batch = ops.sample_pixels(data, ids, u)
    with pytest.raises(RuntimeError, match="no CPU path"):       # the product has no CPU sampler
        sa.PixelBatcher(data, ids, seed=1).sample(16)
    from fruitnerf_amd import _kernels as K
    st = K.ImageSetArg(gdata["images"], gdata["masks"], gdata["c2w"], data["fx"], data["fy"], data["cx"], data["cy"])
    go, gd, gcam, gimg, gmask = K.sample_pixels(st, ids.to(dev), u.to(dev))
    assert torch.equal(gcam.cpu().long(), cam[:, 0])
    assert torch.equal(go.cpu(), o)
    a, _ = util.report("pixel_sampler.directions", gd, d)
    assert a <= 2e-7
    assert torch.equal(gimg.cpu(), batch["image"]) and torch.equal(gmask.cpu(), batch["fruit_mask"][:, 0])


@pytest.mark.parametrize("shape", ["fruit_nerf", "fruit_nerf_big"])
def test_export_counts_and_points_identical(dev, shape):
    """Volume export: identical point counts and identical ordered point lists for the three sets."""
    from fruitnerf_amd.data.fruit_datamanager import ExportDataManager
    from fruitnerf_amd.export.exporter_utils import sample_volume
    cfg = util.small_config(log2=15) if shape == "fruit_nerf" else util.big_config(log2=15)
    om = util.make_oracle(cfg, seed=4, test_mode="export")
    util.randomize_(om, 9, density_boost=3.0)
    om.field.test_mode = "export"
    aabb = ((-1.0, -0.6, -1.0), (1.0, 0.6, 1.0))  # non-cubic: n_y = int(0.6 * N)
    util.straddle_export_thresholds(om, aabb)   # the three thresholds cut through the middle of the lattice's samples
    om.eval()
    hm = util.make_hip_like(om, dev, test_mode="export")
    hm.eval()
    N = 40
    om.setup_inference(True, N)
    ref = fo.sample_volume(om, aabb, N, num_rays_per_batch=333, dataparser_scale=0.7)

    class Pipe:
        pass

    for fused in (True, False):
        pipe = Pipe()
        pipe.model = hm
        pipe.datamanager = ExportDataManager(dev, eval_num_rays_per_batch=333)
        hm.setup_inference(True, N, deterministic=True)
        num_rays = pipe.datamanager.setup_inference(aabb=aabb, num_points=N)
        if not fused:
            pipe.datamanager.export_lattice = None
        got = sample_volume(pipe, num_rays, transform_json={"scale": 0.7})
        near = {}
        for name in ("semantic_colormap", "semantic", "density"):
            n_ref, n_got = ref[name]["points"].shape[0], got[name]["points"].shape[0]
            print(f"[export fused={fused}] {name}: oracle {n_ref} points, hip {n_got} points")
            near[name] = (n_ref, n_got)
        n_total = N * int(0.6 * N) * N
        assert 0.2 * n_total < ref["density"]["points"].shape[0] < 0.8 * n_total, "density threshold is not discriminating"
        assert 0.1 * ref["density"]["points"].shape[0] < ref["semantic"]["points"].shape[0] < \
            ref["semantic_colormap"]["points"].shape[0] < 0.9 * ref["density"]["points"].shape[0]
        for name in ("semantic_colormap", "semantic", "density"):
            assert near[name][0] == near[name][1], f"{name}: count mismatch {near[name]}"
            assert torch.equal(torch.from_numpy(got[name]["points"]), ref[name]["points"]), f"{name}: points differ"
            a, _ = util.report(f"export.{name}.colors", torch.from_numpy(got[name]["colors"]), ref[name]["colors"])
            assert a <= 1e-4


@pytest.mark.parametrize("S", [48, 96, 256])
def test_weights_behind_a_huge_density_spike(dev, S):
    """A sharp surface: delta*sigma ~ 1e8 after a moderate prefix.  The transmittance in front of the spike must
    survive (an exclusive scan computed as `inclusive - own` cancels it to 0 and the spike gets weight 1 on top of the
    earlier weights — that blew training up after a few thousand steps).  Checked against float64 closed form for the
    compositing kernel, the PDF sampler's weights and the weight backward."""
    from fruitnerf_amd import _kernels as K
    R = 64
    g = torch.Generator().manual_seed(S)
    edges = torch.cumsum(torch.rand(R, S + 1, generator=g) * 0.05 + 0.01, dim=1)
    density = torch.rand(R, S, generator=g) * 3.0
    spike = torch.randint(S // 3, S - 2, (R,), generator=g)
    density[torch.arange(R), spike] = 10.0 ** (6 + 4 * torch.rand(R, generator=g))     # 1e6 .. 1e10
    rgb = torch.rand(R, S, 3, generator=g)
    logit = torch.randn(R, S, generator=g)
    dd = (edges[:, 1:] - edges[:, :-1]).double() * density.double()
    T = torch.exp(-torch.cat([torch.zeros(R, 1, dtype=torch.float64), torch.cumsum(dd, 1)[:, :-1]], 1))
    w_ref = ((1 - torch.exp(-dd)) * T).float()
    o, d, pa, cam = util.random_rays(R, 4, seed=1)
    rays = K.RaysArg(o.to(dev), d.to(dev), torch.zeros(R, 1, device=dev), torch.ones(R, 1, device=dev), cam.to(dev))
    w, out_rgb, acc, depth, sem, label = K.composite_fwd(rays, S, edges.to(dev).contiguous(), density.to(dev).view(-1),
                                                         rgb.to(dev).view(-1, 3), logit.to(dev).view(-1), True)
    assert util.report(f"spike[{S}].composite.weights", w, w_ref)[0] <= 2e-6
    assert float(acc.max()) <= 1.0 + 1e-5
    if S + 1 <= 257:
        spacing = (edges / edges[:, -1:]).to(dev).contiguous()
        w2, _, _, _ = K.weights_pdf(rays, 1, S, 32, density.to(dev).contiguous(), spacing, edges.to(dev).contiguous(), 1.0,
                                    None)
        assert util.report(f"spike[{S}].pdf.weights", w2, w_ref)[0] <= 2e-6
    # backward of the weights: d(sum_k g_k w_k)/d sigma vs float64 autograd
    gw = torch.randn(R, S, generator=g)
    dref = density.double().clone().requires_grad_(True)
    ddr = (edges[:, 1:] - edges[:, :-1]).double() * dref
    Tr = torch.exp(-torch.cat([torch.zeros(R, 1, dtype=torch.float64), torch.cumsum(ddr, 1)[:, :-1]], 1))
    ((1 - torch.exp(-ddr)) * Tr * gw.double()).sum().backward()
    d_sigma = K.weights_bwd(S, edges.to(dev).contiguous(), density.to(dev).contiguous(), w, gw.to(dev).contiguous(),
                            torch.ones(1, device=dev))
    ref = dref.grad.float()
    got = d_sigma.view(R, S).cpu()
    err = (got - ref).abs().max().item()
    assert err <= 1e-5 * max(1.0, ref.abs().max().item()), err



def test_weight_gradient_behind_a_surface_is_relatively_exact(dev):
    """Behind a surface (transmittance ~1e-6) dL/dsigma is tiny, but the trunc_exp backward multiplies it by sigma (up
    to e^15): it has to be right in RELATIVE terms.  A suffix sum computed as `total - prefix` is not (total and
    prefix agree to 7 digits there); the kernel uses a reverse scan like autograd's cumsum backward."""
    from fruitnerf_amd import _kernels as K
    R, S = 64, 48
    g = torch.Generator().manual_seed(4)
    edges = torch.cumsum(torch.rand(R, S + 1, generator=g) * 0.05 + 0.01, dim=1)
    density = torch.rand(R, S, generator=g) * 2.0
    spike = torch.randint(S // 4, S // 2, (R,), generator=g)
    delta = edges[:, 1:] - edges[:, :-1]
    density[torch.arange(R), spike] = (12.0 + 6.0 * torch.rand(R, generator=g)) / delta[torch.arange(R), spike]
    gw = torch.randn(R, S, generator=g)
    dref = density.double().clone().requires_grad_(True)
    dd = delta.double() * dref
    T = torch.exp(-torch.cat([torch.zeros(R, 1, dtype=torch.float64), torch.cumsum(dd, 1)[:, :-1]], 1))
    w64 = (1 - torch.exp(-dd)) * T
    (w64 * gw.double()).sum().backward()
    ref = dref.grad
    d_sigma = K.weights_bwd(S, edges.to(dev).contiguous(), density.to(dev).contiguous(), w64.detach().float().to(dev).contiguous(),
                            gw.to(dev).contiguous(), torch.ones(1, device=dev))
    got = d_sigma.view(R, S).cpu().double()
    behind = torch.arange(S)[None, :] > (spike[:, None] + 1)
    rel = ((got - ref).abs() / ref.abs().clamp_min(1e-300))[behind]
    print(f"[behind surface] |ref| range {float(ref[behind].abs().min()):.2e}..{float(ref[behind].abs().max()):.2e} "
          f"max rel err {float(rel.max()):.2e}")
    assert float(ref[behind].abs().max()) < 1e-3 and float(rel.max()) <= 2e-3
