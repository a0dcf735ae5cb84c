"""Size-independent properties of the HIP path at the FULL `fruit_nerf` sizes (2^19-row tables, 4096 rays, samples
256/96/48) where the CPU oracle is too slow, plus ragged / tiny / empty batches (SURVEY §8c "substitute pins" 3)."""
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def _full_model(dev, seed=0, num_images=16):
    from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
    from fruitnerf_amd.data.semantics import apple_metadata
    torch.manual_seed(seed)
    m = FruitModel(FruitNerfModelConfig(), apple_metadata(), num_train_data=num_images, device=dev)
    g = torch.Generator(device=dev).manual_seed(seed + 1)
    with torch.no_grad():  # 'trained-like': O(1) features, non-trivial densities
        m.field.mlp_base_grid.hash_table.copy_(
            (torch.rand(m.field.mlp_base_grid.hash_table.shape, device=dev, generator=g) * 2 - 1) * 0.5)
        for net in m.proposal_networks:
            net.encoding.hash_table.copy_((torch.rand(net.encoding.hash_table.shape, device=dev, generator=g) * 2 - 1) * 0.8)
        m.field.mlp_base_mlp.layers[1].bias[0].add_(2.0)
    return m


def _rays(R, dev, seed=3, num_images=16):
    from fruitnerf_amd.rays import RayBundle
    o, d, pa, cam = util.random_rays(R, num_images, seed=seed)
    return RayBundle(o.to(dev), d.to(dev), pa.to(dev), cam.to(dev))


def _sub(rb, idx):
    from fruitnerf_amd.rays import RayBundle
    return RayBundle(rb.origins[idx].contiguous(), rb.directions[idx].contiguous(), rb.pixel_area[idx].contiguous(),
                     rb.camera_indices[idx].contiguous())


KEYS = ("rgb", "semantics", "accumulation", "depth", "prop_depth_0", "prop_depth_1")


def test_permutation_and_batch_split_invariance_full_size(dev):
    """Rays are independent given the parameters: permuting or splitting the batch permutes / splits the outputs
    bit for bit (eval mode: no jitter)."""
    m = _full_model(dev)
    m.eval()
    R = 4096
    rb = _rays(R, dev)
    with torch.no_grad():
        ref = m(rb)
        perm = torch.randperm(R, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
        got = m(_sub(rb, perm))
        for k in KEYS:
            assert torch.equal(got[k], ref[k][perm]), k
        parts = [m(_sub(rb, torch.arange(a, b, device=dev))) for a, b in ((0, 1000), (1000, 1001), (1001, 4096))]
        for k in KEYS:
            assert torch.equal(torch.cat([p[k] for p in parts]), ref[k]), k
    assert float(ref["accumulation"].mean()) > 0.05 and torch.isfinite(ref["rgb"]).all()


@pytest.mark.parametrize("R", [1, 63, 65, 4097])
def test_ragged_batches_match_the_oracle_rows(dev, R):
    """Batch sizes around the wave (64) and tile (16-sample) boundaries against the CPU oracle on a small model."""
    from oracle import ns_torch as ns
    from fruitnerf_amd.rays import RayBundle
    cfg = util.small_config(log2=14, prop_log2=12)
    om = util.make_oracle(cfg, seed=2)
    hm = util.make_hip_like(om, dev)
    om.eval()
    hm.eval()
    Rc = min(R, 130)  # the oracle only checks the first rows of the big batch (rays are independent)
    o, d, pa, cam = util.random_rays(R, 7, seed=R)
    with torch.no_grad():
        ref = om(ns.RayBundle(o[:Rc], d[:Rc], pa[:Rc], camera_indices=cam[:Rc]))
        got = hm(RayBundle(o.to(dev), d.to(dev), pa.to(dev), cam.to(dev)))
    for k in ("rgb", "semantics", "accumulation"):
        assert got[k].shape[0] == R
        assert util.report(f"ragged[{R}].{k}", got[k][:Rc], ref[k])[0] <= 1e-4


def test_empty_batch(dev):
    """Zero rays: every entry point returns without launching; the model returns empty outputs."""
    m = _full_model(dev)
    m.eval()
    rb = _rays(4, dev)
    with torch.no_grad():
        out = m(_sub(rb, torch.arange(0, 0, device=dev)))
    for k in KEYS:
        assert out[k].shape[0] == 0


def test_scatter_conserves_the_feature_gradient_full_size(dev):
    """Checksum of checksums for the hash-grid backward: the 8 trilinear weights of a sample sum to 1, so for every
    level the column sums of the gradient table equal the column sums of d_feats — at 2^19 rows x 16 levels and
    196 608 samples, through the binned scatter (queues, block fixed point) and the level-group variant."""
    from fruitnerf_amd import _kernels as K
    m = _full_model(dev)
    m.train()
    arena = m.arena()
    fld = m.field
    R, S = 4096, 48
    rb = _rays(R, dev, seed=9)
    rays = K.RaysArg(rb.origins, rb.directions, torch.full((R, 1), 0.05, device=dev), torch.full((R, 1), 1000.0, device=dev),
                     rb.camera_indices)
    _, eu = K.sample_spaced(rays, 1, S, None)
    g = torch.Generator(device=dev).manual_seed(1)
    d_feats = torch.randn(16, R * S, 2, device=dev, generator=g) * 1e-3
    gnet = fld.net_struct(grads=True)
    T = 1 << 19
    want = d_feats.double().sum(dim=1)                               # [16, 2]
    for groups in (1, 4):
        arena.grads.zero_()
        per = 16 // groups
        for lb in range(0, 16, per):
            K.hash_encode_bwd(gnet.grid, fld.warp_struct(), rays, eu, S, d_feats, lb, per)
        table_grad = fld.mlp_base_grid.hash_table.grad.view(16, T, 2)
        got = table_grad.double().sum(dim=1)
        scale = d_feats.abs().double().sum(dim=1)
        assert float(((got - want).abs() / scale).max()) <= 1e-5, groups
        assert int((table_grad != 0).any(dim=2).sum()) > 1_000_000


def test_scatter_with_overflowing_queues_still_conserves_the_gradient(dev):
    """One sample per ray, 196 608 rays alternating between two fixed ones: neighbouring samples never share a cell (no run
    for the emit kernel to pre-sum: 4096 copies of one 48-sample ray leave 8 records per workgroup and level), yet every
    record of a level lands in the same one or two bins — far beyond the queues' capacity (3 x the level's mean), so the
    fallback (global atomics into the gradient table, merged by the accumulate kernel) carries most of the call.  The
    column sums must still match, and fnr_debug_scatter_overflows must report the fallback; it reports none for random
    rays (training never takes it: tests/diagnostics/scatter_overflows.py, bench.py's quality.scatter_queue_overflows)."""
    from fruitnerf_amd import _kernels as K, _lib as L
    m = _full_model(dev)
    m.train()
    arena = m.arena()
    fld = m.field
    N = 4096 * 48
    two = _rays(2, dev, seed=5)
    alt = torch.arange(N, device=dev) % 2
    rays = K.RaysArg(two.origins[alt].contiguous(), two.directions[alt].contiguous(), torch.full((N, 1), 0.4, device=dev),
                     torch.full((N, 1), 0.4, device=dev), torch.zeros(N, 1, dtype=torch.long, device=dev))
    _, eu = K.sample_spaced(rays, 1, 1, None)
    g = torch.Generator(device=dev).manual_seed(1)
    d_feats = torch.randn(16, N, 2, device=dev, generator=g) * 1e-3
    gnet = fld.net_struct(grads=True)
    want = d_feats.double().sum(dim=1)
    scale = d_feats.abs().double().sum(dim=1)
    arena.grads.zero_()
    L.scatter_overflows(reset=True)
    K.hash_encode_bwd(gnet.grid, fld.warp_struct(), rays, eu, 1, d_feats)
    n_over = L.scatter_overflows(reset=True)
    table_grad = fld.mlp_base_grid.hash_table.grad.view(16, 1 << 19, 2)
    got = table_grad.double().sum(dim=1)
    print(f"[scatter overflow] {n_over} of {16 * N * 8} contributions went through the atomic fallback")
    assert n_over > 1_000_000
    assert int((table_grad != 0).any(dim=2).sum()) <= 16 * 16          # two points x 8 corners per level
    assert float(((got - want).abs() / scale).max()) <= 1e-6
    # random rays: the same entry point never overflows
    R, S = 4096, 48
    rb = _rays(R, dev, seed=9)
    rays2 = K.RaysArg(rb.origins, rb.directions, torch.full((R, 1), 0.05, device=dev),
                      torch.full((R, 1), 1000.0, device=dev), rb.camera_indices)
    _, eu2 = K.sample_spaced(rays2, 1, S, None)
    arena.grads.zero_()
    K.hash_encode_bwd(gnet.grid, fld.warp_struct(), rays2, eu2, S, d_feats)
    assert L.scatter_overflows(reset=True) == 0


def test_sparse_touch_bitmap_survives_an_overflowing_scatter(dev):
    """rows whose FIRST gradient arrives on a step that overflows a queue take their optimiser step in the
    accumulate kernel's row-by-row sweep (the level's gradient partly sits in the table, put there by atomics); that sweep
    has to set the sparse-touch bits too, or the next wide sweep skips those pairs and their moments stop decaying.
    Two fused steps — an overflowing one (two points, 196 608 samples), then random rays — with the bitmap against the
    same two steps with every row swept: parameters and both moments bit-identical."""
    import fruitnerf_amd.training as T
    from fruitnerf_amd import _kernels as K, _lib as L
    N = 4096 * 48
    R, S = 4096, 48
    g = torch.Generator(device=dev).manual_seed(1)
    d_feats = torch.randn(16, N, 2, device=dev, generator=g) * 1e-3
    two = _rays(2, dev, seed=5)
    alt = torch.arange(N, device=dev) % 2
    rb = _rays(R, dev, seed=9)
    saved = T.SPARSE_TOUCH_SKIPPING
    states = []
    try:
        for sparse in (True, False):
            T.SPARSE_TOUCH_SKIPPING = sparse
            m = _full_model(dev)
            m.train()
            fld = m.field
            opt = T.FusedAdam(m)
            table = fld.mlp_base_grid.hash_table
            gnet = fld.net_struct(grads=True)
            rays = K.RaysArg(two.origins[alt].contiguous(), two.directions[alt].contiguous(),
                             torch.full((N, 1), 0.4, device=dev), torch.full((N, 1), 0.4, device=dev),
                             torch.zeros(N, 1, dtype=torch.long, device=dev))
            _, eu = K.sample_spaced(rays, 1, 1, None)
            L.scatter_overflows(reset=True)
            args, (a, b) = opt.table_adam_args(table)
            assert (args.touched is not None) == sparse
            K.hash_encode_bwd_adam(gnet.grid, fld.warp_struct(), rays, eu, 1, d_feats, args)
            assert L.scatter_overflows(reset=True) > 1_000_000
            opt.begin_step()
            rays2 = K.RaysArg(rb.origins, rb.directions, torch.full((R, 1), 0.05, device=dev),
                              torch.full((R, 1), 1000.0, device=dev), rb.camera_indices)
            _, eu2 = K.sample_spaced(rays2, 1, S, None)
            for _ in range(2):      # the first step's rows only decay here (random rays do not hit those 256 rows)
                args, _ = opt.table_adam_args(table)
                K.hash_encode_bwd_adam(gnet.grid, fld.warp_struct(), rays2, eu2, S, d_feats, args)
                assert L.scatter_overflows(reset=True) == 0
                opt.begin_step()
            torch.cuda.synchronize()
            states.append((m.arena().params[a:b].clone(), opt.exp_avg[a:b].clone(), opt.exp_avg_sq[a:b].clone()))
    finally:
        T.SPARSE_TOUCH_SKIPPING = saved
    # The overflow fallback adds floats with global atomics, so the 256 hot rows' sums (2 points x 8 corners x 16 levels)
    # differ in their last digits from run to run (include/fruitnerf_hip.h, fnr_debug_scatter_overflows; measured: up to
    # 1.5e-4 relative on a parameter after three steps): those entries are compared to 2e-3 — a row whose moments had
    # stopped decaying would be 19 % off after two steps (beta1^2 = 0.81), its parameter a whole step (2.5 %) —
    # and every other entry bit for bit.
    for name, x, y in zip(("parameters", "exp_avg", "exp_avg_sq"), *states):
        n_diff = int((x != y).sum())
        assert n_diff <= 512, f"{name}: {n_diff} entries differ between the sparse-touch and the dense sweeps"
        assert torch.allclose(x, y, rtol=2e-3, atol=1e-6), \
            f"{name}: max rel diff {float(((x - y).abs() / y.abs().clamp_min(1e-12)).max()):.3e} — rows frozen by a missing bit?"
    assert int((states[0][1] != 0).sum()) > 100_000


def test_adam_leaves_untouched_parameters_alone_and_is_idempotent_on_zero_grad(dev):
    """Zero gradients with zero moments leave parameters bit-identical (19.4 M-element arena), and a second step
    with zero gradients only decays the moments."""
    from fruitnerf_amd.training import FusedAdam
    m = _full_model(dev)
    m.train()
    opt = FusedAdam(m)
    arena = m.arena()
    before = arena.params.clone()
    arena.grads.zero_()
    opt.step()
    assert torch.equal(arena.params, before)
    arena.grads[123456] = 1.0
    opt.step()
    changed = (arena.params != before).nonzero().flatten()
    assert changed.tolist() == [123456]
    assert float(arena.grads.abs().max()) == 0.0


def test_full_image_eval_and_image_metrics(dev):
    """get_outputs_for_camera_ray_bundle (chunked, outputs [H,W,.] on the device; CPU with eval_outputs_on_cpu) + get_image_metrics_and_images
    (fruit_nerf.py:225-249, 403-458) on a small image."""
    from fruitnerf_amd.rays import RayBundle
    m = _full_model(dev)
    m.eval()
    m.config.eval_num_rays_per_chunk = 500   # several ragged chunks
    H, W = 24, 40
    rb = _rays(H * W, dev, seed=12)
    cam_rb = RayBundle(rb.origins.view(H, W, 3), rb.directions.view(H, W, 3), None, None)
    out = m.get_outputs_for_camera_ray_bundle(cam_rb)
    assert out["rgb"].shape == (H, W, 3) and out["rgb"].device.type == "cuda"
    whole = m(RayBundle(rb.origins, rb.directions, None, None))
    assert torch.equal(out["rgb"].view(-1, 3), whole["rgb"])
    m.config.eval_outputs_on_cpu = True     # the reference's behaviour (fruit_nerf.py:245)
    out_cpu = m.get_outputs_for_camera_ray_bundle(cam_rb)
    assert out_cpu["rgb"].device.type == "cpu" and torch.equal(out_cpu["rgb"], out["rgb"].cpu())
    m.config.eval_outputs_on_cpu = False
    g = torch.Generator().manual_seed(0)
    batch = {"image": torch.rand(H, W, 3, generator=g), "fruit_mask": (torch.rand(H, W, 1, generator=g) > 0.7).float()}
    metrics, images = m.get_image_metrics_and_images(out, batch)
    assert set(metrics) == {"psnr", "ssim", "lpips", "iou", "iou_sigmoid"}
    assert 0 < metrics["psnr"] < 60 and -1 <= metrics["ssim"] <= 1
    # every metric against the float64 restatement of torchmetrics' algorithm (oracle/image_metrics.py)
    from oracle import image_metrics as oim
    want_all = oim.image_metrics(out["rgb"].cpu().numpy(), batch["image"].numpy(), out["semantics"].cpu().numpy(),
                                 batch["fruit_mask"].numpy())
    print("[image metrics]", metrics, want_all)
    assert abs(metrics["ssim"] - want_all["ssim"]) <= 1e-5 and abs(metrics["psnr"] - want_all["psnr"]) <= 1e-4
    assert abs(metrics["iou"] - want_all["iou"]) <= 1e-9 and abs(metrics["iou_sigmoid"] - want_all["iou_sigmoid"]) <= 1e-9
    # the reference's quirk (fruit_nerf.py:451): F.softmax without dim on [H,W,1] runs over image rows (implicit dim 0)
    sem, tgt = out["semantics"].cpu(), batch["fruit_mask"][..., 0] > 0.5
    pred = torch.softmax(sem, dim=0)[..., 0] > 0.5
    want = float((pred & tgt).sum()) / max(float((pred | tgt).sum()), 1.0)
    assert abs(metrics["iou"] - want) < 1e-6
    pred = torch.sigmoid(sem)[..., 0] > 0.5
    assert abs(metrics["iou_sigmoid"] - float((pred & tgt).sum()) / max(float((pred | tgt).sum()), 1.0)) < 1e-6
    assert images["img"].shape == (H, 2 * W, 3) and images["fruit_mask"].shape == (H, W, 3)


@pytest.mark.parametrize("fused", [True, False])
def test_export_is_invariant_to_the_ray_batch_size(dev, fused):
    """sample_volume on a 48^3 lattice with the full-size field: the three point sets (counts AND ordered coordinates)
    do not depend on eval_num_rays_per_batch — 7 ragged batches vs one (SURVEY §8c pin 3)."""
    import numpy as np
    from fruitnerf_amd.data.fruit_datamanager import ExportDataManager
    from fruitnerf_amd.export.exporter_utils import sample_volume
    m = _full_model(dev)
    with torch.no_grad():
        m.field.field_head_semantics.net.weight.mul_(8.0)
        m.field.field_head_semantics.net.bias.add_(1.0)
        m.field.mlp_base_mlp.layers[1].weight[0].mul_(3.0)   # densities around the export threshold (sigma >= 70)
        m.field.mlp_base_mlp.layers[1].bias[0].add_(2.5)
    m.test_mode = "export"
    m.field.test_mode = "export"
    m.eval()
    N = 48

    class Pipe:
        pass

    results = []
    for per_batch in (N * N, 333):
        pipe = Pipe()
        pipe.model = m
        pipe.datamanager = ExportDataManager(dev, eval_num_rays_per_batch=per_batch)
        m.setup_inference(True, N, deterministic=True)
        n_rays = pipe.datamanager.setup_inference(aabb=((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)), num_points=N)
        if not fused:
            pipe.datamanager.export_lattice = None
        results.append(sample_volume(pipe, n_rays, transform_json={"scale": 1.0}))
    a, b = results
    assert a["density"]["points"].shape[0] > 50
    for name in ("semantic_colormap", "semantic", "density"):
        assert a[name]["points"].shape == b[name]["points"].shape, name
        assert np.array_equal(a[name]["points"], b[name]["points"]), name
        assert np.array_equal(a[name]["colors"], b[name]["colors"]), name


def test_train_prologue_is_the_separate_launches(dev):
    """fnr_train_prologue (random numbers + camera adjust + pixel sampling + level-0 spaced sampling in one launch): given
    the numbers it drew, every output is bit-identical to fnr_camera_adjust / fnr_sample_pixels / fnr_sample_spaced; the
    numbers are uniform in [0, 1), differ between rays, words and steps, and repeat for the same (seed, offset)."""
    import torch
    from fruitnerf_amd import _kernels as K
    from fruitnerf_amd.cameras.camera_optimizers import CameraOptimizerConfig
    from fruitnerf_amd.data import synthetic_apple as sa
    n_cam, HW, focal, R, S0 = 12, 64, 90.0, 5000, 256
    scene = sa.make_scene(seed=0, device=dev)
    c2w = sa.make_cameras(n_cam, seed=0, device=dev)
    data = sa.render_dataset(scene, c2w, H=HW, W=HW, fx=focal, fy=focal)
    iset = K.ImageSetArg(data["images"], data["masks"], data["c2w"], focal, focal, HW / 2.0, HW / 2.0)
    ids = torch.arange(n_cam, device=dev)
    cam_opt = CameraOptimizerConfig(mode="SO3xR3").setup(n_cam, dev)
    with torch.no_grad():
        cam_opt.pose_adjustment.copy_(0.02 * torch.randn(n_cam, 6, device=dev))
    for pose in (cam_opt.pose_adjustment.data, None):
        out = K.train_prologue(iset, ids, R, seed=1234, offset=7, pose_adjustment=pose, near=0.05, far=1000.0, S0=S0)
        again = K.train_prologue(iset, ids, R, seed=1234, offset=7, pose_adjustment=pose, near=0.05, far=1000.0, S0=S0)
        other = K.train_prologue(iset, ids, R, seed=1234, offset=8, pose_adjustment=pose, near=0.05, far=1000.0, S0=S0)
        torch.cuda.synchronize()
        rnd = torch.cat([out["u"].t(), out["jitter"]])                       # [6, R]
        assert torch.equal(out["u"], again["u"]) and torch.equal(out["jitter"], again["jitter"])
        assert float(rnd.min()) >= 0.0 and float(rnd.max()) < 1.0
        assert abs(float(rnd.mean()) - 0.5) < 0.01 and abs(float(rnd.var()) - 1.0 / 12.0) < 0.005
        assert int(torch.unique(rnd).numel()) > 0.99 * rnd.numel()          # 24-bit numbers, 30 000 of them
        assert float((rnd != torch.cat([other["u"].t(), other["jitter"]])).float().mean()) > 0.99
        c = torch.corrcoef(rnd)
        assert float((c - torch.eye(6, device=dev)).abs().max()) < 0.06     # words of a ray are uncorrelated
        c2w_adj = K.camera_adjust(iset, ids, pose) if pose is not None else None
        if pose is not None:
            assert torch.equal(out["c2w_adjusted"], c2w_adj)
        o, d, cam, image, mask = K.sample_pixels(iset, ids, out["u"], c2w_adj)
        for name, got, ref in (("origins", out["origins"], o), ("directions", out["directions"], d), ("cam", out["cam"], cam),
                               ("image", out["image"], image), ("mask", out["mask"], mask)):
            assert torch.equal(got, ref), name
        rays = K.RaysArg(o, d, torch.full((R, 1), 0.05, device=dev), torch.full((R, 1), 1000.0, device=dev), cam)
        spacing, euclid = K.sample_spaced(rays, 1, S0, out["jitter"][0])
        assert torch.equal(out["spacing"], spacing) and torch.equal(out["euclid"], euclid)


@pytest.mark.parametrize("shape,span", [((11, 11), 1.0), ((64, 75), 1.0), ((64, 75), 0.55), ((203, 131), 1.0)])
def test_image_metrics_kernel_matches_the_float64_oracle(dev, shape, span):
    """fnr_image_metrics (csrc/image_metrics.hip) on synthetic images with structure (smooth ramps + noise, a prediction
    with out-of-range values so that the clamp matters, saturated and near-zero logits): SSIM within 1e-5 and PSNR within
    1e-4 dB of the float64 restatement of torchmetrics' algorithm, both IoUs exact; (11, 11) is the smallest image with an
    SSIM value (one window), the others have ragged 32 x 32 tiles.  span < 1: both images squeezed into [0.2, 0.2 + span] —
    the reference passes no data_range, so torchmetrics derives c1 / c2 from the images' own value range."""
    import numpy as np
    from fruitnerf_amd import _kernels as K
    from oracle import image_metrics as oim
    H, W = shape
    g = torch.Generator().manual_seed(H * 1000 + W)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    image = torch.stack([0.5 + 0.4 * torch.sin(6 * xx + 2 * yy), yy * xx, 0.3 + 0.2 * torch.cos(9 * yy)], dim=-1)
    image = (image + 0.05 * torch.rand(H, W, 3, generator=g)).clamp(0, 1)
    rgb = image + 0.15 * torch.randn(H, W, 3, generator=g)          # leaves [0, 1]: the metrics see clamp(rgb)
    if span != 1.0:
        image, rgb = 0.2 + span * image, 0.2 + span * rgb.clamp(0, 1)
    sem = 8.0 * torch.randn(H, W, 1, generator=g)
    mask = (torch.rand(H, W, 1, generator=g) > 0.6).float()
    sums = K.image_metrics(rgb.to(dev), image.to(dev), sem.to(dev)[..., 0], mask.to(dev)[..., 0]).tolist()
    want = oim.image_metrics(rgb.numpy(), image.numpy(), sem.numpy(), mask.numpy())
    got = {"psnr": -10.0 * np.log10(sums[0] / sums[7]), "ssim": sums[1] / sums[6],
           "iou_sigmoid": sums[2] / max(sums[3], 1.0), "iou": sums[4] / max(sums[5], 1.0)}
    print(f"[image metrics {H}x{W}]", got, want)
    assert sums[6] == 3 * (H - 10) * (W - 10) and sums[7] == 3 * H * W
    assert abs(got["ssim"] - want["ssim"]) <= 1e-5
    assert abs(got["psnr"] - want["psnr"]) <= 1e-4
    assert got["iou"] == pytest.approx(want["iou"], abs=1e-12) and got["iou_sigmoid"] == pytest.approx(want["iou_sigmoid"], abs=1e-12)
    # without semantics: only the image sums
    sums2 = K.image_metrics(rgb.to(dev), image.to(dev)).tolist()
    assert sums2[0] == sums[0] and sums2[1] == sums[1] and sums2[2:6] == [0.0, 0.0, 0.0, 0.0]


@pytest.mark.parametrize("case", ["nan_in_render", "nan_in_image"])
def test_image_metrics_value_range_follows_torch_on_nan(dev, case):
    """torchmetrics' SSIM with data_range=None takes max() - min() of the (clamped) images: torch's reductions and clamp
    propagate NaN, so one NaN in the render (or the image) makes the range, c1 / c2 and the SSIM NaN — fminf / fmaxf would
    drop it and report a finite SSIM.  Compared with the float64 restatement (NumPy propagates NaN the same
    way).  (A constant image pair — range 0, c1 = c2 = 0 — makes every SSIM value 0 / 0 up to the rounding of the window
    sums: garbage in torchmetrics, in the oracle and here alike; nothing to pin.)"""
    import numpy as np
    from fruitnerf_amd import _kernels as K
    from oracle import image_metrics as oim
    H, W = 40, 33
    g = torch.Generator().manual_seed(11)
    image = torch.rand(H, W, 3, generator=g)
    rgb = (image + 0.1 * torch.randn(H, W, 3, generator=g))
    if case == "nan_in_render":
        rgb[17, 5, 1] = float("nan")
    elif case == "nan_in_image":
        image[3, 30, 2] = float("nan")
    sem = torch.randn(H, W, 1, generator=g)
    mask = (torch.rand(H, W, 1, generator=g) > 0.5).float()
    sums = K.image_metrics(rgb.to(dev), image.to(dev), sem.to(dev)[..., 0], mask.to(dev)[..., 0]).tolist()
    with np.errstate(all="ignore"):
        want = oim.image_metrics(rgb.numpy(), image.numpy(), sem.numpy(), mask.numpy())
    ssim = sums[1] / sums[6]
    print(f"[image metrics {case}] ssim {ssim} oracle {want['ssim']}")
    assert np.isnan(want["ssim"]), "the oracle (NumPy max / min / clip) propagates NaN"
    assert np.isnan(ssim)
    # the IoUs do not depend on the images
    assert sums[2] / max(sums[3], 1.0) == pytest.approx(want["iou_sigmoid"], abs=1e-12)
