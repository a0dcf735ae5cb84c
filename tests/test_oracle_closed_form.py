"""CPU: pins for the oracle.  The reference ships no tests or golden vectors ("parity
unpinned"), so these closed-form checks are what anchors the restatement in oracle/."""

import numpy as np
import pytest
import torch

from oracle import fruit_oracle as fo
from oracle import ns_torch as ns


def test_hash_scalings_match_survey_appendix_a2():
    enc = ns.HashEncoding(num_levels=16, min_res=16, max_res=2048, log2_hashmap_size=4)
    assert enc.scalings.tolist() == [16, 22, 30, 42, 58, 80, 111, 153, 212, 294, 406, 561, 776, 1072, 1482, 2047]
    assert ns.HashEncoding(5, 16, 128, 4).scalings.tolist() == [16, 26, 45, 76, 128]
    assert ns.HashEncoding(5, 16, 256, 4).scalings.tolist() == [16, 32, 64, 128, 256]
    assert ns.HashEncoding(16, 16, 4096, 4).scalings.tolist()[-2:] == [2830, 4095]


def test_hash_encoding_returns_table_rows_at_lattice_points():
    torch.manual_seed(0)
    enc = ns.HashEncoding(num_levels=3, min_res=4, max_res=16, log2_hashmap_size=8)
    T = enc.hash_table_size
    for level, s in enumerate(enc.scalings.tolist()):
        s = int(s)
        ijk = torch.tensor([[1, 2, 3], [0, 0, 0], [s, s - 1, 1]])
        x = ijk.float() / s  # scaled = exactly integral for powers of two; use exactness check below
        scaled = x * s
        if not torch.equal(scaled, ijk.float()):
            continue
        out = enc(x)[:, 2 * level:2 * level + 2]
        h = ((ijk[:, 0] * 1) ^ (ijk[:, 1] * 2654435761) ^ (ijk[:, 2] * 805459861)) % T + level * T
        assert torch.equal(out, enc.hash_table[h])


def test_hash_is_uint32_compatible():
    enc = ns.HashEncoding(num_levels=1, min_res=16, max_res=16, log2_hashmap_size=19)
    v = torch.tensor([[2047, 2047, 2047], [1, 2046, 7]], dtype=torch.int32)
    got = enc.hash_fn(v[:, None, :])[:, 0]
    x = v.numpy().astype(np.uint32)
    want = (x[:, 0] ^ (x[:, 1] * np.uint32(2654435761)) ^ (x[:, 2] * np.uint32(805459861))) & np.uint32(2 ** 19 - 1)
    assert got.tolist() == want.tolist()


def test_hash_encoding_trilinear_midpoint():
    enc = ns.HashEncoding(num_levels=1, min_res=4, max_res=4, log2_hashmap_size=10)
    x = torch.tensor([[0.375, 0.625, 0.125]])  # cell (1,2,0) centre at res 4
    f = [1, 2, 0]
    rows = []
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                rows.append(((f[0] + dx) ^ ((f[1] + dy) * 2654435761) ^ ((f[2] + dz) * 805459861)) % 1024)
    want = enc.hash_table[rows].mean(0)
    assert torch.allclose(enc(x)[0], want, atol=1e-9)


def test_sh16_analytic_values():
    d = torch.tensor([[0.0, 0.0, 1.0], [1.0, 0.0, 0.0], [0.5, 0.5, 1.0]])
    c = ns.components_from_spherical_harmonics(4, d)
    assert c[0, 0].item() == pytest.approx(0.28209479177387814)
    assert c[0, 2].item() == pytest.approx(0.4886025119029199)
    assert c[0, 6].item() == pytest.approx(0.9461746957575601 - 0.31539156525251999, rel=1e-6)
    assert c[0, 12].item() == pytest.approx(0.3731763325901154 * 2, rel=1e-6)
    assert c[1, 3].item() == pytest.approx(0.4886025119029199)
    assert c[1, 15].item() == pytest.approx(0.5900435899266435, rel=1e-6)
    # export direction (0,0,1) is shifted to (0.5,0.5,1): constant vector
    assert c[2, 4].item() == pytest.approx(1.0925484305920792 * 0.25, rel=1e-6)
    assert c[2, 8].item() == 0.0


def test_scene_contraction_linf():
    c = ns.SceneContraction()
    x = torch.tensor([[0.5, -0.2, 0.1], [1.0, 0.0, 0.0], [2.0, -1.0, 0.5], [1e6, 0.0, 0.0]])
    y = c(x)
    assert torch.equal(y[0], x[0])
    assert torch.allclose(y[1], x[1])
    assert torch.allclose(y[2], torch.tensor([1.5, -0.75, 0.375]))
    assert y[3, 0].item() == pytest.approx(2.0, abs=1e-5)


def test_trunc_exp_backward_clamps():
    x = torch.tensor([0.0, 20.0, -20.0], requires_grad=True)
    ns.trunc_exp(x).sum().backward()
    assert torch.allclose(x.grad, torch.exp(torch.tensor([0.0, 15.0, -15.0])))


def _bundle(R, near=0.05, far=1000.0):
    o = torch.zeros(R, 3)
    d = torch.nn.functional.normalize(torch.ones(R, 3), dim=-1)
    return ns.RayBundle(o, d, torch.ones(R, 1), camera_indices=torch.zeros(R, 1, dtype=torch.long),
                        nears=torch.full((R, 1), near), fars=torch.full((R, 1), far))


def test_piecewise_sampler_spacing():
    s = ns.UniformLinDispPiecewiseSampler()
    s.eval()
    rs = s(_bundle(2), num_samples=8)
    sp = torch.cat([rs.spacing_starts[..., 0], rs.spacing_ends[..., -1:, 0]], -1)
    assert torch.allclose(sp[0], torch.linspace(0, 1, 9))
    eu = torch.cat([rs.frustums.starts[..., 0], rs.frustums.ends[..., -1:, 0]], -1)
    assert eu[0, 0].item() == pytest.approx(0.05, rel=1e-6)
    assert eu[0, -1].item() == pytest.approx(1000.0, rel=2e-3)
    assert (eu[0, 1:] > eu[0, :-1]).all()


def test_pdf_sampler_flat_weights_gives_uniform_bins():
    s0 = ns.UniformLinDispPiecewiseSampler()
    s0.eval()
    rb = _bundle(3)
    rs = s0(rb, num_samples=16)
    pdf = ns.PDFSampler(include_original=False)
    pdf.eval()
    out = pdf(rb, rs, torch.ones(3, 16, 1), num_samples=7)
    sp = torch.cat([out.spacing_starts[..., 0], out.spacing_ends[..., -1:, 0]], -1)
    u = torch.linspace(0, 1 - 1 / 8, 8) + 1 / 16
    assert torch.allclose(sp[0], u, atol=1e-6)


def test_pdf_sampler_one_hot_concentrates():
    s0 = ns.UniformLinDispPiecewiseSampler()
    s0.eval()
    rb = _bundle(1)
    rs = s0(rb, num_samples=10)
    w = torch.zeros(1, 10, 1)
    w[0, 4] = 50.0
    pdf = ns.PDFSampler(include_original=False)
    pdf.eval()
    out = pdf(rb, rs, w, num_samples=20)
    sp = torch.cat([out.spacing_starts[..., 0], out.spacing_ends[..., -1:, 0]], -1)[0]
    assert ((sp >= 0.4 - 1e-6) & (sp <= 0.5 + 1e-6)).float().mean() > 0.9


def test_weights_sum_identity():
    rb = _bundle(4, near=0.0, far=4.0)
    s = fo.UniformSamplerWithNoise(num_samples=32)
    s.eval()
    rs = s(rb)
    sigma = torch.rand(4, 32, 1) * 3
    w = rs.get_weights(sigma)
    total = (rs.deltas * sigma).sum(-2)
    assert torch.allclose(w.sum(-2), 1 - torch.exp(-total), atol=1e-6)
    assert (w >= 0).all() and (w.sum(-2) <= 1 + 1e-6).all()


def test_renderers():
    w = torch.tensor([[[0.2], [0.3], [0.1]]])
    rgb = torch.tensor([[[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]]])
    out = ns.render_rgb_last_sample(rgb, w, training=True)
    assert torch.allclose(out, torch.tensor([[0.2, 0.3, 0.1 + 0.4]]))
    assert ns.render_accumulation(w).item() == pytest.approx(0.6)
    assert ns.render_semantics(torch.tensor([[[1.0], [2.0], [3.0]]]), w).item() == pytest.approx(1.1)


def test_interlevel_loss_zero_when_proposal_bounds_final():
    rb = _bundle(2)
    s0 = ns.UniformLinDispPiecewiseSampler()
    s0.eval()
    prop = s0(rb, num_samples=8)
    fine = s0(rb, num_samples=8)
    wp = torch.full((2, 8, 1), 0.1)
    wf = torch.full((2, 8, 1), 0.05)
    assert ns.interlevel_loss([wp, wf], [prop, fine]).item() == pytest.approx(0.0, abs=1e-9)
    assert ns.interlevel_loss([wf, wp], [prop, fine]).item() > 0


@pytest.mark.parametrize("aabb,n,want", [
    (((-1, -1, -1), (1, 1, 1)), 1000, (1000, 1000)),
    (((-1.0, -0.6, -1.0), (1.0, 0.6, 1.0)), 1000, (1000, int(torch.tensor(1.2) / torch.tensor(2.0) * 1000))),
    (((-0.5, -0.5, -1.0), (0.5, 0.5, 1.0)), 64, (32, 32)),
])
def test_export_lattice_sizes(aabb, n, want):
    corners = fo.get_corners_of_aabb(aabb)
    pts, vec = fo.sample_surface_points(corners, n)
    assert pts.shape[0] == want[0] * want[1]
    assert vec.tolist() == [[0.0, 0.0, 2.0]]
    assert pts[:, 2].unique().tolist() == [float(aabb[0][2])]
    # x-major ordering: index = i_x * n_y + i_y
    assert pts[1, 0] == pts[0, 0] and pts[1, 1] > pts[0, 1]


def test_export_boundary_rays_are_masked_and_counts_on_analytic_field():
    """Lattice endpoints normalise to exactly 0 / 1 and are killed by the strict selector."""
    cfg = fo.FruitNerfModelConfig(log2_hashmap_size=8, max_res=64)
    cfg.proposal_net_args_list = [dict(hidden_dim=16, log2_hashmap_size=6, num_levels=5, max_res=32, use_linear=False)] * 2
    torch.manual_seed(0)
    m = fo.FruitModel(cfg, num_train_data=2, test_mode="export")
    m.eval()
    with torch.no_grad():  # constant field: density = e^5 = 148 >= 70 everywhere inside, logit = 4
        for p in m.field.parameters():
            p.zero_()
        m.field.mlp_base_mlp.layers[1].bias[0] = 5.0
        m.field.field_head_semantics.net.bias[0] = 4.0
    N = 12
    m.setup_inference(True, N)
    out = fo.sample_volume(m, ((-1, -1, -1), (1, 1, 1)), N, num_rays_per_batch=50)
    inside = (N - 2) * (N - 2) * N  # boundary rows/cols of the N x N ray grid are excluded, all N depths kept
    assert out["density"]["points"].shape[0] == inside
    assert out["semantic"]["points"].shape[0] == inside
    assert out["semantic_colormap"]["points"].shape[0] == inside
    assert out["density"]["points"].abs().max().item() <= 2.0 + 1e-9  # x2 rescale (exporter_utils.py:191)


def test_camera_optimizer_exp_map_is_a_rotation_and_matches_matrix_exp():
    """oracle/camera_opt.py::exp_map_SO3xR3 (restated nerfstudio lie_groups): above the 1e-4 clamp the rotation block
    is exp(skew(w)) (independent formula: torch.matrix_exp), it is orthonormal with det 1, the translation is copied;
    below the clamp it is the small-angle series around the clamped angle; multiply() composes poses."""
    import torch
    from oracle import camera_opt as oc
    g = torch.Generator().manual_seed(0)
    tv = torch.cat([torch.randn(16, 3, generator=g), torch.randn(16, 3, generator=g) * 0.7], dim=1).double()
    M = oc.exp_map_SO3xR3(tv)
    R, t = M[:, :, :3], M[:, :, 3]
    assert torch.equal(t, tv[:, :3])
    w = tv[:, 3:]
    K = torch.zeros(16, 3, 3, dtype=torch.float64)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -w[:, 2], w[:, 1], w[:, 2], -w[:, 0], -w[:, 1], w[:, 0]
    assert torch.allclose(R, torch.matrix_exp(K), atol=1e-12)
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3, dtype=torch.float64).expand(16, 3, 3), atol=1e-12)
    assert torch.allclose(torch.linalg.det(R), torch.ones(16, dtype=torch.float64), atol=1e-12)
    # below the clamp (|w|^2 < 1e-4): theta is clamped to 0.01, R = I + f1 K + f2 K^2 with the clamped factors
    small = torch.zeros(1, 6, dtype=torch.float64)
    small[0, 3:] = torch.tensor([1e-3, -2e-3, 5e-4])
    Rs = oc.exp_map_SO3xR3(small)[0, :, :3]
    Ks = torch.zeros(3, 3, dtype=torch.float64)
    Ks[0, 1], Ks[0, 2], Ks[1, 0], Ks[1, 2], Ks[2, 0], Ks[2, 1] = -5e-4, -2e-3, 5e-4, -1e-3, 2e-3, 1e-3
    th = torch.tensor(0.01, dtype=torch.float64)
    want = torch.eye(3, dtype=torch.float64) + torch.sin(th) / th * Ks + (1 - torch.cos(th)) / th ** 2 * (Ks @ Ks)
    assert torch.allclose(Rs, want, atol=1e-15)
    # zero tangent -> identity; composition
    I = oc.exp_map_SO3xR3(torch.zeros(2, 6, dtype=torch.float64))
    assert torch.allclose(I[:, :, :3], torch.eye(3, dtype=torch.float64).expand(2, 3, 3)) and float(I[:, :, 3].abs().max()) == 0
    A, B = M[:8], M[8:]
    AB = oc.multiply(A, B)
    assert torch.allclose(AB[:, :, :3], A[:, :, :3] @ B[:, :, :3]) and torch.allclose(
        AB[:, :, 3], A[:, :, 3] + (A[:, :, :3] @ B[:, :, 3:]).squeeze(-1))


def test_camera_ray_generation_conventions():
    """Pinhole rays through the (identity-corrected) camera: pixel centre +0.5, -z forward, unit directions, origins =
    camera position — and a pure translation tangent moves the origins by R1 t only."""
    import torch
    from oracle import camera_opt as oc
    c2w = torch.eye(4)[None, :3, :4].clone()
    c2w[0, :, 3] = torch.tensor([0.1, -0.2, 0.3])
    y, x = torch.tensor([5]), torch.tensor([7])
    o, d = oc.generate_rays(c2w, oc.exp_map_SO3xR3(torch.zeros(1, 6)), y, x, 10.0, 10.0, 8.0, 8.0)
    v = torch.tensor([(7 + 0.5 - 8) / 10, -(5 + 0.5 - 8) / 10, -1.0])
    assert torch.allclose(d[0], v / v.norm(), atol=1e-7) and torch.allclose(o[0], c2w[0, :, 3])
    tv = torch.zeros(1, 6)
    tv[0, :3] = torch.tensor([0.01, 0.02, -0.03])
    o2, d2 = oc.generate_rays(c2w, oc.exp_map_SO3xR3(tv), y, x, 10.0, 10.0, 8.0, 8.0)
    assert torch.allclose(o2[0] - o[0], tv[0, :3], atol=1e-7) and torch.allclose(d2, d, atol=1e-6)


def test_pixel_rays_pinhole_geometry():
    """oracle/pixel_sampler.py on closed forms: with an identity pose the ray through the principal point (pixel centre
    exactly on (cx, cy)) is (0, 0, -1); fx pixels to the right it leaves at 45 degrees towards +x; fy pixels DOWN the image
    towards -y; all rays are unit length and start at the camera centre; a rotated / translated camera rotates and moves
    them rigidly.  The integer pixel (x, y) is sampled at (x + 0.5, y + 0.5)."""
    from oracle import pixel_sampler as ops
    fx, fy, cx, cy = 50.0, 40.0, 32.5, 24.5            # pixel (32, 24) has its centre on the principal point
    eye = torch.eye(3, 4)[None]
    cam = torch.zeros(4, dtype=torch.long)
    x = torch.tensor([32, 82, 32, 0])
    y = torch.tensor([24, 24, 64, 0])
    o, d = ops.pixel_rays(eye, cam, y, x, fx, fy, cx, cy)
    s = 0.5 ** 0.5
    assert torch.equal(o, torch.zeros(4, 3))
    assert torch.allclose(d[0], torch.tensor([0.0, 0.0, -1.0]), atol=1e-7)
    assert torch.allclose(d[1], torch.tensor([s, 0.0, -s]), atol=1e-7)          # +fx pixels -> +x at 45 degrees
    assert torch.allclose(d[2], torch.tensor([0.0, -s, -s]), atol=1e-7)         # +fy pixels (down) -> -y at 45 degrees
    want3 = torch.tensor([(0.5 - cx) / fx, -(0.5 - cy) / fy, -1.0])
    assert torch.allclose(d[3], want3 / want3.norm(), atol=1e-7)
    assert torch.allclose(d.norm(dim=-1), torch.ones(4), atol=1e-6)
    # rigid motion: 90 degrees about +y (camera now looks along -x), centre at (1, 2, 3)
    R = torch.tensor([[0.0, 0.0, 1.0], [0.0, 1.0, 0.0], [-1.0, 0.0, 0.0]])
    c2w = torch.cat([R, torch.tensor([[1.0], [2.0], [3.0]])], dim=1)[None]
    o2, d2 = ops.pixel_rays(c2w, cam, y, x, fx, fy, cx, cy)
    assert torch.equal(o2, torch.tensor([1.0, 2.0, 3.0]).expand(4, 3))
    assert torch.allclose(d2, d @ R.T, atol=1e-7)
    assert torch.allclose(d2[0], torch.tensor([-1.0, 0.0, 0.0]), atol=1e-7)


def test_pixel_sampler_picks_floor_of_u_times_extent():
    """sample_pixels: (slot, row, column) = floor(u * (n_train, H, W)) clamped to the last index; the slot (not the dataset
    image) is the camera index; colours are uint8 / 255, the mask is 0 / 1."""
    from oracle import pixel_sampler as ops
    M, H, W = 5, 6, 7
    g = torch.Generator().manual_seed(0)
    data = {"images": torch.randint(0, 256, (M, H, W, 3), dtype=torch.uint8, generator=g),
            "masks": torch.randint(0, 2, (M, H, W), dtype=torch.uint8, generator=g),
            "c2w": torch.eye(3, 4).expand(M, 3, 4).clone(), "H": H, "W": W, "fx": 9.0, "fy": 9.0, "cx": W / 2, "cy": H / 2}
    data["c2w"][:, :, 3] = torch.arange(M, dtype=torch.float32)[:, None]          # camera i sits at (i, i, i)
    ids = torch.tensor([4, 1, 3])
    u = torch.tensor([[0.0, 0.0, 0.0], [0.34, 0.5, 0.99], [0.999999, 0.999999, 0.999999], [0.67, 0.17, 0.43]])
    o, d, cam, batch = ops.sample_pixels(data, ids, u)
    assert cam[:, 0].tolist() == [0, 1, 2, 2]
    assert o[:, 0].tolist() == [4.0, 1.0, 3.0, 3.0]                                # slots -> dataset images 4, 1, 3, 3
    rows, cols = [0, 3, 5, 1], [0, 6, 6, 3]
    for i, (k, r, c) in enumerate(zip([4, 1, 3, 3], rows, cols)):
        assert torch.equal(batch["image"][i], data["images"][k, r, c].float() / 255.0)
        assert batch["fruit_mask"][i, 0] == float(data["masks"][k, r, c])
    assert batch["image"].dtype == torch.float32 and batch["fruit_mask"].shape == (4, 1)
