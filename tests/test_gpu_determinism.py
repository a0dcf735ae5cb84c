"""Training on the HIP path is bit-reproducible: the reference's CPU path is deterministic, and so is this one — every
sum that crosses workgroups has a fixed order (per-workgroup partial images reduced by a single writer:
k_reduce_dw / k_prop_reduce; the hash-grid scatter adds 64-bit fixed-point values, which commute; pose and embedding
gradients gather their rays by ballot)."""
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def _run(dev, steps, mlp_precision, overlap=None, prologue=True, ahead=None, torch113=False, fuse_weights=True,
         eval_after=()):
    # overlap / ahead None: the product defaults (on)
    import fruitnerf_amd.training as T
    from fruitnerf_amd.cameras.camera_optimizers import CameraAdam, CameraOptimizerConfig
    from fruitnerf_amd.data import synthetic_apple as sa
    from fruitnerf_amd.data.semantics import apple_metadata
    from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
    from fruitnerf_amd.rays import RayBundle
    HW, focal, n_train = 96, 1111.0 * 96 / 800, 40
    scene = sa.make_scene(seed=0, device=dev)
    c2w = sa.make_cameras(n_train, seed=0, device=dev)
    data = sa.render_dataset(scene, c2w, H=HW, W=HW, fx=focal, fy=focal)
    batcher = sa.PixelBatcher(data, torch.arange(n_train, device=dev), seed=1)
    torch.manual_seed(0)
    hm = FruitModel(FruitNerfModelConfig(mlp_precision=mlp_precision), apple_metadata(), num_train_data=n_train, device=dev)
    hm.train()
    opt = T.FusedAdam(hm, skip_groups_without_grad=not torch113)
    cam_opt = CameraOptimizerConfig(mode="SO3xR3").setup(n_train, dev)
    camera = (cam_opt, CameraAdam(cam_opt), batcher)
    saved = T.OVERLAP_PROPOSAL_BACKWARD, T.SAMPLE_AHEAD, T.FUSE_WEIGHT_OPTIMIZER
    T.FUSE_WEIGHT_OPTIMIZER = fuse_weights
    if overlap is not None:
        T.OVERLAP_PROPOSAL_BACKWARD = overlap
    if ahead is not None:
        T.SAMPLE_AHEAD = ahead
    losses = []
    try:
        if prologue:
            # what bench.py runs: fnr_train_prologue (the step's random numbers, corrected cameras, rays and level-0 bins
            # in one launch) and the next step's sampling enqueued at the end of the current one (training.TrainingSteps)
            loop = T.TrainingSteps(hm, opt, batcher, 4096, camera=(cam_opt, camera[1]))
        for step in range(steps):
            if prologue:
                ld, md = loop.step()
            else:   # torch.rand + the separate entry points
                o, d, cam, batch = batcher.sample(4096, cam_opt, level0=None)
                rb = RayBundle(o, d, None, cam, presampled=batcher.last_presample)
                ld, md = T.fused_train_iteration(hm, opt, rb, batch, step, camera=camera)
            if step + 1 in eval_after:   # an eval pass between two iterations (a Trainer's eval_image / bench.py's quality gate)
                hm.eval()
                with torch.no_grad():
                    eo, ed, _, ecam = util.random_rays(2048, n_train, seed=step)
                    for _ in range(2):   # the first eval render may reset the sampler's update counter, the second not
                        hm(RayBundle(eo.to(dev), ed.to(dev), None, ecam.to(dev)))
                hm.train()
            if step % 20 == 19 or step == steps - 1:
                losses.append(torch.stack([ld["rgb_loss"], ld["semantics_loss"], ld["interlevel_loss"], md["psnr"],
                                           md["distortion"]]).clone())
        if prologue and T.SAMPLE_AHEAD and not eval_after:   # every step but the first ran on what the previous one sampled ahead
            assert hm.__dict__.get("_ahead_used", 0) == steps - 1
        if prologue and T.SAMPLE_AHEAD and eval_after:
            # an eval pass resets the sampler's counter when an update was due (nerfstudio does): the look-ahead assumed an
            # update that the next iteration then does not see, and the model samples again
            assert steps - 1 - len(eval_after) <= hm.__dict__.get("_ahead_used", 0) <= steps - 1
    finally:
        T.OVERLAP_PROPOSAL_BACKWARD, T.SAMPLE_AHEAD, T.FUSE_WEIGHT_OPTIMIZER = saved
    torch.cuda.synchronize()
    return (hm.arena().params.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), cam_opt.pose_adjustment.data.clone(),
            torch.stack(losses))


@pytest.mark.parametrize("mlp_precision", ["bf16x3", "fp32"])
def test_two_training_runs_from_one_seed_end_bit_identical(dev, mlp_precision):
    """200 full training steps (4096 rays, SO3xR3 camera optimiser, proposal-network updates, fused table optimiser)
    twice from the same seeds: every parameter, both Adam moments, the camera poses and the logged losses / metrics
    are bit-identical."""
    a = _run(dev, 200, mlp_precision)
    b = _run(dev, 200, mlp_precision)
    names = ("parameters", "exp_avg", "exp_avg_sq", "camera poses", "losses + metrics")
    for name, x, y in zip(names, a, b):
        n_diff = int((x != y).sum())
        print(f"[determinism {mlp_precision}] {name}: {n_diff} of {x.numel()} entries differ "
              f"(max abs diff {float((x - y).abs().max()):.3e})")
    for name, x, y in zip(names, a, b):
        assert torch.equal(x, y), f"{name} differ between two runs from one seed"
    assert float(a[4][-1][0]) < 5e-3, "the runs did train"


def test_runs_without_the_prologue_launch_are_reproducible_too(dev):
    a = _run(dev, 40, "bf16x3", prologue=False)
    b = _run(dev, 40, "bf16x3", prologue=False)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_paired_proposal_levels_are_the_separate_calls(dev):
    """fnr_prop_density_bwd_pair (both proposal levels through one entry point, their accumulate launches as one) against
    one fnr_prop_density_bwd(_adam) call per level: identical states after 30 steps."""
    import fruitnerf_amd.training as T
    a = _run(dev, 30, "bf16x3")
    saved, T.PAIR_PROPOSAL_LEVELS = T.PAIR_PROPOSAL_LEVELS, False
    try:
        b = _run(dev, 30, "bf16x3")
    finally:
        T.PAIR_PROPOSAL_LEVELS = saved
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("mlp_precision", ["bf16x3", "fp32"])
def test_losses_on_the_second_stream_change_nothing(dev, mlp_precision):
    """training.LOSSES_ON_SIDE (round 5): the losses launch heads the second stream's segment and the composite backward
    forms the per-ray loss gradients itself (fnr_composite_bwd_targets) — against the losses launch ahead of the backward
    on the launch stream (fnr_train_losses -> fnr_composite_bwd).  60 steps (every step trains the proposal networks until
    step 10, every other one after: both step shapes): identical parameters, moments, poses and logged losses."""
    import fruitnerf_amd.training as T
    on = _run(dev, 60, mlp_precision)
    saved, T.LOSSES_ON_SIDE = T.LOSSES_ON_SIDE, False
    try:
        inline = _run(dev, 60, mlp_precision)
    finally:
        T.LOSSES_ON_SIDE = saved
    assert saved is True                     # the default under test
    for name, x, y in zip(("parameters", "exp_avg", "exp_avg_sq", "camera poses", "losses + metrics"), on, inline):
        assert torch.equal(x, y), f"{name} differ between the losses launch on the second stream and on the launch stream"


def test_sampling_ahead_is_sampling_at_the_start_of_the_step(dev):
    """training.SAMPLE_AHEAD: the next iteration's prologue + proposal sampling enqueued at the end of the current one
    (on the second stream, underneath the table scatter) against every iteration sampling at its own start; with and
    without the second stream.  120 steps cover the every-step (< 10) and every-other-step proposal update schedule."""
    ref = _run(dev, 120, "bf16x3", overlap=False, ahead=False)
    for overlap in (False, True):
        got = _run(dev, 120, "bf16x3", overlap=overlap, ahead=True)
        for name, x, y in zip(("parameters", "exp_avg", "exp_avg_sq", "camera poses", "losses + metrics"), ref, got):
            assert torch.equal(x, y), f"{name} differ (second stream {overlap})"


def test_sampling_ahead_survives_eval_passes_between_iterations(dev):
    """Eval renders between two training iterations change the proposal sampler's update counter (nerfstudio resets it in
    eval mode too), i.e. the schedule the look-ahead assumed: the model must notice and sample again — same states as
    without the look-ahead.  Eval passes after steps where an update was due (12, 30) and where it was not (21, 41)."""
    ref = _run(dev, 60, "bf16x3", overlap=False, ahead=False, eval_after=(12, 21, 30, 41))
    got = _run(dev, 60, "bf16x3", overlap=True, ahead=True, eval_after=(12, 21, 30, 41))
    for x, y in zip(ref, got):
        assert torch.equal(x, y)


@pytest.mark.parametrize("config", ["torch113", "unfused"])
def test_sampling_ahead_waits_for_an_unfused_proposal_step(dev, config):
    """When the proposal networks' optimiser step is NOT part of their backward (FUSE_WEIGHT_OPTIMIZER off), or happens on
    every iteration (torch 1.13 semantics: zero gradients still move the parameters through the moments), the look-ahead
    must follow optimizer.step(): same states as sampling at the start of the next iteration."""
    kw = dict(torch113=True) if config == "torch113" else dict(fuse_weights=False)
    ref = _run(dev, 40, "bf16x3", overlap=False, ahead=False, **kw)
    got = _run(dev, 40, "bf16x3", overlap=True, ahead=True, **kw)
    for x, y in zip(ref, got):
        assert torch.equal(x, y)


def test_second_stream_run_is_bit_identical_too(dev):
    """The proposal-network backward on a second HIP stream (FNR_OVERLAP_PROPOSAL_BACKWARD) changes the interleaving of
    kernels, not the arithmetic."""
    a = _run(dev, 100, "bf16x3", overlap=False)
    b = _run(dev, 100, "bf16x3", overlap=True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_stream_safe_mode_changes_nothing(dev):
    """FNR_STREAM_SAFE (training.STREAM_SAFE): every tensor that crosses between the launch stream and the second stream
    is also registered with the consuming stream (Tensor.record_stream), so the caching allocator cannot recycle a block
    under a kernel that still reads it.  The default relies on the fork / join structure alone (training._ForkJoin); if
    that structure had a hole, the two modes would eventually part ways.  150 steps (every-step and every-other-step
    proposal updates), eval passes in between; the switch must also have found tensors to register."""
    import fruitnerf_amd.training as T
    ref = _run(dev, 150, "bf16x3", eval_after=(12, 41))
    saved, T.STREAM_SAFE = T.STREAM_SAFE, True
    counted = []
    orig = T.crosses_to

    def counting(stream, *objs):
        n = orig(stream, *objs)
        counted.append(n)
        return n

    T.crosses_to = counting
    try:
        got = _run(dev, 150, "bf16x3", eval_after=(12, 41))
    finally:
        T.STREAM_SAFE, T.crosses_to = saved, orig
    assert sum(counted) > 150 * 10, "the switch registered no cross-stream tensors"
    for name, x, y in zip(("parameters", "exp_avg", "exp_avg_sq", "camera poses", "losses + metrics"), ref, got):
        assert torch.equal(x, y), f"{name} differ with FNR_STREAM_SAFE"


def test_fused_composite_backward_changes_nothing(dev):
    """training.FUSE_COMPOSITE_BACKWARD (round 6): on the steps without a proposal backward the compositing launch runs its own
    backward (fnr_composite_fwd_bwd_targets).  150 steps (every-step and every-other-step proposal updates, so both kinds of
    step occur), eval passes in between: parameters, moments, poses and the logged losses bit-identical with the switch off,
    and the fused entry point is what ran."""
    import fruitnerf_amd.training as T
    from fruitnerf_amd import _kernels as K
    calls = []
    orig = K.composite_fwd_bwd_targets

    def counting(*a, **k):
        calls.append(1)
        return orig(*a, **k)

    K.composite_fwd_bwd_targets = counting
    try:
        ref = _run(dev, 150, "bf16x3", eval_after=(12, 41))
        n_fused = len(calls)
        saved, T.FUSE_COMPOSITE_BACKWARD = T.FUSE_COMPOSITE_BACKWARD, False
        try:
            got = _run(dev, 150, "bf16x3", eval_after=(12, 41))
        finally:
            T.FUSE_COMPOSITE_BACKWARD = saved
    finally:
        K.composite_fwd_bwd_targets = orig
    assert n_fused > 0 and len(calls) == n_fused, (n_fused, len(calls))   # interpreted steps of the first run only (replays bypass Python)
    for name, x, y in zip(("parameters", "exp_avg", "exp_avg_sq", "camera poses", "losses + metrics"), ref, got):
        assert torch.equal(x, y), f"{name} differ with FNR_FUSE_COMPOSITE_BACKWARD=0"


def test_lookahead_is_dropped_when_parameters_change_between_steps(dev):
    """a cached look-ahead carries the next step's proposal samples and saved proposal features; it is tied to
    torch's version counters of the proposal networks' parameters and of the camera poses, so a load_state_dict or a pose
    edit between two TrainingSteps.step calls makes the next step sample again — same states as without the look-ahead."""
    import fruitnerf_amd.training as T

    def run(ahead):
        from fruitnerf_amd.cameras.camera_optimizers import CameraAdam, CameraOptimizerConfig
        from fruitnerf_amd.data import synthetic_apple as sa
        from fruitnerf_amd.data.semantics import apple_metadata
        from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
        HW, focal, n_train = 96, 1111.0 * 96 / 800, 40
        scene = sa.make_scene(seed=0, device=dev)
        c2w = sa.make_cameras(n_train, seed=0, device=dev)
        data = sa.render_dataset(scene, c2w, H=HW, W=HW, fx=focal, fy=focal)
        batcher = sa.PixelBatcher(data, torch.arange(n_train, device=dev), seed=1)
        torch.manual_seed(0)
        hm = FruitModel(FruitNerfModelConfig(), apple_metadata(), num_train_data=n_train, device=dev)
        hm.train()
        opt = T.FusedAdam(hm)
        cam_opt = CameraOptimizerConfig(mode="SO3xR3").setup(n_train, dev)
        loop = T.TrainingSteps(hm, opt, batcher, 2048, camera=(cam_opt, CameraAdam(cam_opt)))
        saved, T.SAMPLE_AHEAD = T.SAMPLE_AHEAD, ahead
        try:
            for step in range(16):
                loop.step()
                if step == 5:      # a checkpoint reload with different proposal networks
                    state = {k: (v * 1.01 if k.startswith("proposal_networks") else v) for k, v in hm.state_dict().items()}
                    hm.load_state_dict(state, strict=True)
                if step == 9:      # the poses edited from outside
                    with torch.no_grad():
                        cam_opt.pose_adjustment.mul_(0.5)
        finally:
            T.SAMPLE_AHEAD = saved
        torch.cuda.synchronize()
        return hm.arena().params.clone(), opt.exp_avg.clone(), cam_opt.pose_adjustment.data.clone(), hm.__dict__.get("_ahead_used", 0)

    ref = run(False)
    got = run(True)
    assert got[3] == 15 - 2, f"look-aheads used: {got[3]} (two of the 15 must have been dropped)"
    for name, x, y in zip(("parameters", "exp_avg", "camera poses"), ref, got):
        assert torch.equal(x, y), f"{name} differ: a stale look-ahead was used"


def test_sparse_touch_skipping_changes_nothing(dev):
    """fnr_table_adam.touched (round 4): the fused table steps skip pairs of rows that never received a gradient — their
    moments are zero, so torch's update of them is exactly zero.  60 steps with the skipping on (the default), with every
    row swept (training.SPARSE_TOUCH_SKIPPING off) and with the bitmaps thrown away and rebuilt from the moments half way:
    identical parameters, moments, poses, losses; and the bitmap says what the moments say — a pair is marked iff one of
    its 4 moment entries is non-zero — with a sizeable part of the table never touched (every level is hashed: the coarse
    levels use (res + 1)^3 of their 2^19 rows)."""
    import fruitnerf_amd.training as T
    ref = _run(dev, 60, "bf16x3")
    saved = T.SPARSE_TOUCH_SKIPPING
    T.SPARSE_TOUCH_SKIPPING = False
    try:
        dense = _run(dev, 60, "bf16x3")
    finally:
        T.SPARSE_TOUCH_SKIPPING = saved
    for name, x, y in zip(("parameters", "exp_avg", "exp_avg_sq", "camera poses", "losses + metrics"), ref, dense):
        assert torch.equal(x, y), f"{name} differ between sparse-touch skipping and the dense sweep"
    # the bitmap against the moments, on a short run of the same loop
    from fruitnerf_amd.cameras.camera_optimizers import CameraAdam, CameraOptimizerConfig
    from fruitnerf_amd.data import synthetic_apple as sa
    from fruitnerf_amd.data.semantics import apple_metadata
    from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
    HW, focal, n_train = 96, 1111.0 * 96 / 800, 40
    scene = sa.make_scene(seed=0, device=dev)
    c2w = sa.make_cameras(n_train, seed=0, device=dev)
    data = sa.render_dataset(scene, c2w, H=HW, W=HW, fx=focal, fy=focal)
    batcher = sa.PixelBatcher(data, torch.arange(n_train, device=dev), seed=1)
    torch.manual_seed(0)
    hm = FruitModel(FruitNerfModelConfig(), apple_metadata(), num_train_data=n_train, device=dev)
    hm.train()
    opt = T.FusedAdam(hm)
    cam_opt = CameraOptimizerConfig(mode="SO3xR3").setup(n_train, dev)
    loop = T.TrainingSteps(hm, opt, batcher, 4096, camera=(cam_opt, CameraAdam(cam_opt)))
    for step in range(12):
        loop.step()
        if step == 5:
            opt.rebuild_touched()            # e.g. after loading moments: rebuilt from them at the next fused step
    torch.cuda.synchronize()
    table = hm.field.mlp_base_grid.hash_table
    a, n = [(off, k) for _, p, off, k in hm.arena().entries if p is table][0]
    bm = opt._touched[a][1]                     # keyed by the table's arena offset: (length, bitmap)
    bits = ((bm.to(torch.int64)[:, None] >> torch.arange(32, device=dev)) & 1).bool().reshape(-1)
    m, v = opt.exp_avg[a:a + n].view(-1, 4), opt.exp_avg_sq[a:a + n].view(-1, 4)
    nonzero = ((m != 0) | (v != 0)).any(dim=1)
    assert torch.equal(bits, nonzero), "the touched bitmap and the moments disagree"
    frac = float(nonzero.float().mean())
    print(f"[sparse touch] pairs of rows ever touched after 12 steps: {frac:.3f} of the table")
    assert 0.5 < frac < 0.8            # levels 0 - 4 mostly untouched, levels 5 - 15 dense
    untouched = ~nonzero
    assert bool((opt.exp_avg[a:a + n].view(-1, 4)[untouched] == 0).all())
