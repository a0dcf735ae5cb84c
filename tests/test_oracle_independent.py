"""oracle/ns_torch.py (nerfstudio 0.3.2 torch path, restated from memory) vs oracle/independent.py (float64 NumPy
derivations from the published definitions that share no code with it) on random inputs.  What this cannot pin is
listed in oracle/independent.py's header and DESIGN.md section 2."""
import numpy as np
import pytest
import torch

from oracle import independent as ind
from oracle import ns_torch as ns


@pytest.mark.parametrize("levels,min_res,max_res,log2", [(16, 16, 2048, 19), (16, 16, 4096, 21), (5, 16, 128, 17),
                                                         (7, 16, 2048, 17)])
def test_hash_grid_matches_the_instant_ngp_definition(levels, min_res, max_res, log2):
    torch.manual_seed(levels + log2)
    enc = ns.HashEncoding(num_levels=levels, min_res=min_res, max_res=max_res, log2_hashmap_size=log2)
    with torch.no_grad():
        enc.hash_table.copy_(torch.rand_like(enc.hash_table) * 2 - 1)   # O(1) entries: a sharper test than 1e-3
    g = torch.Generator().manual_seed(1)
    x = torch.rand(500, 3, generator=g)
    x[:8] = torch.tensor([[0.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.25, 0.75, 0.125], [1.0, 1.0, 1.0],   # lattice hits,
                          [0.999999, 0.5, 1e-7], [0.0625, 0.0625, 0.0625], [0.3, 0.0, 1.0], [1.0, 0.0, 0.5]])  # faces
    with torch.no_grad():
        got = enc(x).numpy()
    want = ind.hash_grid(x.numpy(), enc.hash_table.detach().numpy(), [int(s) for s in enc.scalings.tolist()],
                         2 ** log2)
    assert np.abs(got - want).max() < 5e-6


def test_hash_index_equals_uint32_hash_for_all_grid_coordinates():
    enc = ns.HashEncoding(num_levels=1, min_res=16, max_res=16, log2_hashmap_size=19)
    g = torch.Generator().manual_seed(0)
    v = torch.randint(0, 8193, (4096, 3), generator=g, dtype=torch.int32)
    got = enc.hash_fn(v[:, None, :])[:, 0].numpy()
    want = ind.spatial_hash(v[:, 0].numpy(), v[:, 1].numpy(), v[:, 2].numpy(), 2 ** 19)
    assert np.array_equal(got, want)


def test_sh16_equals_scipy_real_spherical_harmonics_on_the_unit_sphere():
    g = torch.Generator().manual_seed(2)
    d = torch.randn(400, 3, generator=g, dtype=torch.float64)
    d = d / d.norm(dim=-1, keepdim=True)
    got = ns.components_from_spherical_harmonics(4, d.float()).numpy()
    want = ind.real_sh16(d.numpy())
    assert np.abs(got - want).max() < 2e-6


def test_render_weights_equal_the_nerf_quadrature_and_its_closed_form():
    g = torch.Generator().manual_seed(3)
    R, S = 7, 48
    sigma = torch.rand(R, S, 1, generator=g) * 30
    edges = torch.sort(torch.rand(R, S + 1, generator=g) * 4, dim=-1).values
    rs = ns.RaySamples(frustums=None, deltas=(edges[:, 1:] - edges[:, :-1])[..., None])
    got = rs.get_weights(sigma)[..., 0].numpy()
    want = ind.render_weights(sigma[..., 0].numpy(), rs.deltas[..., 0].numpy())
    assert np.abs(got - want).max() < 2e-6
    # homogeneous medium: w_i = exp(-s t_i) - exp(-s t_{i+1}) with t measured from the first edge
    s = 2.5
    got = rs.get_weights(torch.full((R, S, 1), s))[..., 0].double().numpy()
    t = (edges - edges[:, :1]).double().numpy()
    assert np.abs(got - (np.exp(-s * t[:, :-1]) - np.exp(-s * t[:, 1:]))).max() < 2e-6


def test_scene_contraction_equals_mipnerf360_eq10_with_the_inf_norm():
    g = torch.Generator().manual_seed(4)
    p = torch.randn(1000, 3, generator=g) * 3
    got = ns.SceneContraction(order=float("inf"))(p).numpy()
    assert np.abs(got - ind.contract_linf(p.numpy())).max() < 1e-6
    assert np.abs(got).max() < 2.0


@pytest.mark.parametrize("training", [False, True])
def test_pdf_sampler_inverts_the_padded_histogram_cdf(training):
    """Inverse-transform sampling: CDF(bin_k) must equal u_k for the step density  p ~ w + padding  on the previous
    level's bins.  (padding = 0.01 is nerfstudio's constant — recalled, stated in DESIGN.)"""
    g = torch.Generator().manual_seed(5)
    R, S_prev, S = 6, 32, 16
    rb = ns.RayBundle(torch.zeros(R, 3), torch.tensor([[0.0, 0.0, 1.0]]).repeat(R, 1), torch.ones(R, 1),
                      nears=torch.full((R, 1), 0.05), fars=torch.full((R, 1), 1000.0))
    first = ns.UniformLinDispPiecewiseSampler(num_samples=S_prev, single_jitter=True)
    first.eval()
    prev = first(rb)
    w = torch.rand(R, S_prev, 1, generator=g) ** 4
    w[0] = 0.0   # an empty ray: the padded histogram is flat
    sampler = ns.PDFSampler(num_samples=S, include_original=False, single_jitter=True)
    sampler.train(training)
    rand = torch.rand(R, 1, generator=g)
    out = sampler(rb, prev, w, rand=rand if training else None)
    bins = torch.cat([out.spacing_starts[..., 0], out.spacing_ends[..., -1:, 0]], -1).double().numpy()
    edges = torch.cat([prev.spacing_starts[..., 0], prev.spacing_ends[..., -1:, 0]], -1).double().numpy()
    nb = S + 1
    u = np.arange(nb) / nb + (rand.double().numpy() / nb if training else 1.0 / (2 * nb))
    for r in range(R):
        mass = w[r, :, 0].double().numpy() + 0.01
        mass = mass / mass.sum()
        cdf_at_bins = ind.histogram_cdf(edges[r], mass, bins[r])
        assert np.abs(cdf_at_bins - u[r] if training else cdf_at_bins - u).max() < 2e-5
    # and the euclidean bins are the piecewise warp of the s-space bins
    eu = torch.cat([out.frustums.starts[..., 0], out.frustums.ends[..., -1:, 0]], -1).double().numpy()
    want = np.vectorize(lambda s: ind.lin_disp_piecewise(s, 0.05, 1000.0))(bins)
    assert np.abs(eu / want - 1).max() < 5e-4   # float32 1/(2-2s) near s = 1 amplifies rounding


def test_piecewise_sampler_is_uniform_in_distance_then_in_disparity():
    S = 64
    rb = ns.RayBundle(torch.zeros(1, 3), torch.tensor([[0.0, 0.0, 1.0]]), torch.ones(1, 1),
                      nears=torch.full((1, 1), 0.0), fars=torch.full((1, 1), 1000.0))
    smp = ns.UniformLinDispPiecewiseSampler(num_samples=S)
    smp.eval()
    out = smp(rb)
    eu = torch.cat([out.frustums.starts[0, :, 0], out.frustums.ends[0, -1:, 0]]).double().numpy()
    inner = eu[eu <= 1.0 + 1e-6]
    assert np.abs(np.diff(inner) - np.diff(inner)[0]).max() < 1e-6          # equal steps in distance up to t = 1
    outer = eu[eu >= 1.0 - 1e-6][:-1]
    d = np.diff(1.0 / outer)
    assert np.abs(d / d[0] - 1).max() < 1e-3                                 # equal steps in 1/t beyond
    want = np.array([ind.lin_disp_piecewise(k / S, 0.0, 1000.0) for k in range(S + 1)])
    assert np.abs(eu[:-1] / np.maximum(want[:-1], 1e-12) - 1)[1:].max() < 1e-4


def _rs_from_sdist(sd):
    sd = torch.as_tensor(sd, dtype=torch.float32)
    return ns.RaySamples(frustums=ns.Frustums(None, None, sd[:, :-1, None], sd[:, 1:, None], None),
                         spacing_starts=sd[:, :-1, None], spacing_ends=sd[:, 1:, None])


def test_interlevel_loss_equals_the_brute_force_outer_measure():
    rng = np.random.default_rng(6)
    R, n, m0, m1 = 5, 12, 20, 9
    def hist(k):
        t = np.sort(rng.uniform(0, 1, (R, k + 1)), axis=-1)
        t[:, 0], t[:, -1] = 0.0, 1.0     # common range: inverse-CDF sampling nests every level inside the previous one, so a fine interval is never
        # outside the envelope's range (the one case where the searchsorted form and the overlap definition differ)
        w = rng.uniform(0, 1, (R, k)) ** 3
        return t, w / w.sum(-1, keepdims=True) * rng.uniform(0.3, 1.0, (R, 1))
    t, w = hist(n)
    (t0, w0), (t1, w1) = hist(m0), hist(m1)
    got = float(ns.interlevel_loss(
        [torch.tensor(w0, dtype=torch.float32)[..., None], torch.tensor(w1, dtype=torch.float32)[..., None],
         torch.tensor(w, dtype=torch.float32)[..., None]],
        [_rs_from_sdist(t0), _rs_from_sdist(t1), _rs_from_sdist(t)]))
    want = 0.0
    for te, we in ((t0, w0), (t1, w1)):
        want += np.mean([ind.outer_measure_loss(t[r], w[r], te[r], we[r], eps=1e-7) for r in range(R)])
    assert abs(got - want) < 1e-6 * max(1.0, abs(want))
    assert want > 1e-4   # the case is not vacuous


def test_distortion_loss_equals_the_double_sum():
    rng = np.random.default_rng(7)
    R, n = 4, 10
    t = np.sort(rng.uniform(0, 1, (R, n + 1)), axis=-1)
    w = rng.uniform(0, 0.2, (R, n))
    got = float(ns.distortion_loss([torch.tensor(w, dtype=torch.float32)[..., None]], [_rs_from_sdist(t)]))
    want = np.mean([ind.distortion(t[r], w[r]) for r in range(R)])
    assert abs(got - want) < 1e-6


def test_median_depth_is_the_weighted_median_of_the_sample_midpoints():
    rng = np.random.default_rng(8)
    R, n = 6, 15
    t = np.sort(rng.uniform(0, 5, (R, n + 1)), axis=-1)
    w = rng.uniform(0, 1, (R, n))
    w = w / w.sum(-1, keepdims=True) * np.array([1.0, 0.9, 0.7, 0.4, 0.2, 1.0])[:, None]   # rays that never reach 1/2
    rs = _rs_from_sdist(t)
    got = ns.render_depth_median(torch.tensor(w, dtype=torch.float32)[..., None], rs)[:, 0].numpy()
    mids = (t[:, :-1] + t[:, 1:]) / 2
    want = np.array([ind.weighted_median(mids[r], w[r]) for r in range(R)])
    assert np.abs(got - want).max() < 1e-6


def test_image_metrics_oracle_against_an_independent_ssim():
    """oracle/image_metrics.py (separable filtering of the reflect-padded images, then the crop) against the definition
    evaluated another way: a full 121-tap 2-D convolution (scipy) of the UNPADDED images restricted to the windows that fit
    — torchmetrics' pad-then-crop leaves exactly those.  Plus the closed forms: SSIM(a, a) = 1, SSIM of two constant images
    = (2 ab + c1) / (a^2 + b^2 + c1), and the window is normalised and symmetric."""
    import numpy as np
    from scipy.signal import convolve2d
    from oracle import image_metrics as om
    g = om.gaussian_window()
    assert g.dtype == np.float32 and abs(float(g.sum(dtype=np.float64)) - 1.0) < 1e-6 and np.array_equal(g, g[::-1])
    rng = np.random.default_rng(0)
    a = rng.random((40, 37, 3))
    b = np.clip(a + 0.1 * rng.standard_normal(a.shape), 0, 1)
    assert abs(om.ssim(a, a) - 1.0) < 1e-12
    w = np.outer(g.astype(np.float64), g.astype(np.float64))

    def f(x):
        return np.stack([convolve2d(x[..., c], w, mode="valid") for c in range(3)])

    mu_a, mu_b = f(a), f(b)
    saa, sbb, sab = f(a * a) - mu_a ** 2, f(b * b) - mu_b ** 2, f(a * b) - mu_a * mu_b
    m = ((2 * mu_a * mu_b + 1e-4) * (2 * sab + 9e-4)) / ((mu_a ** 2 + mu_b ** 2 + 1e-4) * (saa + sbb + 9e-4))
    assert m.shape == (3, 30, 27) and abs(om.ssim(a, b, data_range=1.0) - float(m.mean())) < 1e-12
    # data_range None (what the reference's call leaves torchmetrics with): the larger of the two images' value ranges
    R = max(a.max() - a.min(), b.max() - b.min())
    c1, c2 = (0.01 * R) ** 2, (0.03 * R) ** 2
    m = ((2 * mu_a * mu_b + c1) * (2 * sab + c2)) / ((mu_a ** 2 + mu_b ** 2 + c1) * (saa + sbb + c2))
    assert abs(om.ssim(a, b) - float(m.mean())) < 1e-12
    a2, b2 = 0.2 + 0.5 * a, 0.2 + 0.5 * b                     # squeezed images: the constants follow the range
    assert abs(om.ssim(a2, b2) - om.ssim(a2, b2, data_range=0.5 * R)) < 1e-12
    assert abs(om.ssim(a2, b2) - om.ssim(a2, b2, data_range=1.0)) > 1e-4
    ca, cb = np.full((20, 20, 3), 0.3), np.full((20, 20, 3), 0.7)
    # (the float32 window sums to 1 within 1e-7, so the 'variances' of a constant image are ~1e-8 against c2 = 9e-4)
    assert abs(om.ssim(ca, cb, data_range=1.0) - (2 * 0.3 * 0.7 + 1e-4) / (0.09 + 0.49 + 1e-4)) < 2e-5
    assert abs(om.psnr(ca, cb) - 10 * np.log10(1 / 0.16)) < 1e-9
    sem = np.ones((12, 11, 1))
    sem[2, 1, 0] = 10.0                                   # dominates its COLUMN: the row softmax is > 0.5 only there
    mask = np.zeros((12, 11, 1))
    mask[2, 1, 0] = mask[0, 0, 0] = 1.0
    r = om.image_metrics(np.zeros((12, 11, 3)) + 0.5, np.zeros((12, 11, 3)) + 0.5 + 1e-3, sem, mask)
    assert r["iou"] == 0.5 and r["iou_sigmoid"] == 2 / 132    # sigmoid(1) > 0.5 everywhere
