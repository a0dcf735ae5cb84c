"""The native step sequencer (training.TrainingSteps + fnr_program_*, ABI 12): a training step whose launch sequence has
been recorded is replayed by ONE call of the C ABI.  Same entry points, same arguments, same streams, same order — so
training must be bit-identical with the sequencer on and off, the host-side counters (schedules, optimiser step counts,
the batcher's random-number counter) must end where the interpreted loop leaves them, and anything the recording cannot
hold must send the step back to the interpreter instead of into a wrong replay."""
import pytest
import torch

from tests import util
from tests.test_gpu_determinism import _run

pytestmark = pytest.mark.gpu

NAMES = ("parameters", "exp_avg", "exp_avg_sq", "camera poses", "losses + metrics")


def _interpreted(fn):
    import fruitnerf_amd.training as T
    saved, T.NATIVE_SEQUENCER = T.NATIVE_SEQUENCER, False
    try:
        return fn()
    finally:
        T.NATIVE_SEQUENCER = saved


@pytest.mark.parametrize("mlp_precision", ["bf16x3", "fp32"])
def test_replayed_training_is_the_interpreted_training(dev, mlp_precision):
    """120 steps of the headline loop (4096 rays, camera optimiser, two streams, look-ahead): every-step proposal updates
    until step 10, every other step after — all four step shapes per parity get recorded and replayed."""
    import fruitnerf_amd.training as T
    assert T.NATIVE_SEQUENCER, "the sequencer is the default under test"
    native = _run(dev, 120, mlp_precision)
    ref = _interpreted(lambda: _run(dev, 120, mlp_precision))
    for name, x, y in zip(NAMES, native, ref):
        assert torch.equal(x, y), f"{name} differ between replayed and interpreted steps"


def test_replay_survives_eval_passes_and_profiled_steps(dev):
    """Eval renders between two iterations move the sampler's update counter (the look-ahead's schedule no longer holds: the
    step is interpreted and samples again) and use the allocator, not the step arenas; steps whose launches are bracketed
    by HIP events (bench.py's roofline leg) are interpreted with the streams serialised.  Neither may disturb the replayed
    steps around them."""
    import fruitnerf_amd.training as T
    from fruitnerf_amd import _lib as L

    def run():
        return _run(dev, 60, "bf16x3", eval_after=(12, 21, 30, 41))
    native = run()
    ref = _interpreted(run)
    for name, x, y in zip(NAMES, native, ref):
        assert torch.equal(x, y), f"{name} differ (eval passes between replayed steps)"


def _loop(dev, n_rays=2048, camera=True, precision=None, big=False):
    import fruitnerf_amd.training as T
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
    cfg = FruitNerfModelConfig(mlp_precision=precision)
    if big:      # the fruit_nerf_big shape (other MLP kernels, RAdam) on small tables
        bc = util.big_config(log2=15, prop_log2=13)
        for k, v in vars(bc).items():
            if hasattr(cfg, k):
                setattr(cfg, k, v)
        cfg.num_nerf_samples_per_ray, cfg.num_proposal_samples_per_ray = 64, (128, 96)
    hm = FruitModel(cfg, apple_metadata(), num_train_data=n_train, device=dev)
    hm.train()
    opt = T.FusedAdam(hm, algorithm="radam" if big else "adam")
    cam = None
    if camera:
        cam_opt = CameraOptimizerConfig(mode="SO3xR3").setup(n_train, dev)
        cam = (cam_opt, CameraAdam(cam_opt, algorithm="radam" if big else "adam"))
    return T.TrainingSteps(hm, opt, batcher, n_rays, camera=cam), hm, opt, batcher, cam


def _host_state(loop, hm, opt, batcher, cam):
    s = hm.proposal_sampler
    return {"step_idx": loop.step_idx, "opt.step_count": opt.step_count, "opt.group_steps": dict(opt.group_steps),
            "sampler._step": s._step, "sampler._steps_since_update": s._steps_since_update, "sampler._anneal": float(s._anneal),
            "batcher._offset": batcher._offset, "camera.step_count": None if cam is None else cam[1].step_count,
            "_ahead_used": hm.__dict__.get("_ahead_used", 0), "_last_render_updated": hm._last_render_updated}


@pytest.mark.parametrize("big", [False, True])
def test_host_state_and_statistics_after_replayed_steps(dev, big):
    """The counters Python keeps — optimiser / scheduler steps per group, the sampler's schedule state, the batcher's
    random-number counter, the camera optimiser's step count — end where the interpreted loop leaves them; most steps were
    replays; both built field shapes record (the fruit_nerf_big shape: other MLP kernels, RAdam's host-side rectification)."""
    def run(steps=40):
        loop, hm, opt, batcher, cam = _loop(dev, big=big)
        losses = []
        for i in range(steps):
            ld, md = loop.step(want_metrics=(i % 3 != 0))
            losses.append(torch.stack(list(ld.values())).clone())
        torch.cuda.synchronize()
        return _host_state(loop, hm, opt, batcher, cam), hm.arena().params.clone(), torch.stack(losses), dict(loop.stats)
    st_n, p_n, l_n, stats = run()
    st_i, p_i, l_i, stats_i = _interpreted(run)
    assert st_n == st_i
    assert torch.equal(p_n, p_i) and torch.equal(l_n, l_i)
    print("[sequencer] stats", stats)
    assert stats_i["replayed"] == 0 and stats_i["recorded"] == 0
    assert stats["replayed"] >= 20 and stats["recorded"] >= 4 and stats["record_failed"] == 0
    assert stats["replayed"] + stats["interpreted"] == 40


def test_outside_changes_send_the_step_back_to_the_interpreter(dev):
    """A checkpoint reload, a pose edit, a dropped look-ahead, the camera optimiser switched off and on, another MLP
    arithmetic: every one of them changes what a recorded program assumed — the affected steps are interpreted (and new
    shapes recorded), and the states stay those of the interpreted loop."""
    def run():
        loop, hm, opt, batcher, cam = _loop(dev)
        for step in range(48):
            loop.step()
            if step == 13:
                state = {k: (v * 1.01 if k.startswith("proposal_networks") else v) for k, v in hm.state_dict().items()}
                hm.update_to_step(step)
                hm.load_state_dict(state, strict=True)
            if step == 19:
                with torch.no_grad():
                    cam[0].pose_adjustment.mul_(0.5)
            if step == 24:
                loop.drop_lookahead()
            if step == 29:
                loop.camera = None
                loop.drop_lookahead()
            if step == 35:
                loop.camera = (cam[0], cam[1], batcher)
                loop.drop_lookahead()
            if step == 40:
                hm.field.mlp_precision = "fp32"
        torch.cuda.synchronize()
        return hm.arena().params.clone(), opt.exp_avg.clone(), cam[0].pose_adjustment.data.clone(), dict(loop.stats)
    native = run()
    ref = _interpreted(run)
    for name, x, y in zip(("parameters", "exp_avg", "camera poses"), native, ref):
        assert torch.equal(x, y), f"{name} differ"
    assert native[3]["replayed"] >= 12 and native[3]["record_failed"] == 0, native[3]


def test_unrecordable_configurations_stay_interpreted(dev):
    """A step that calls an entry point without a recording hook (here: the MLP weights' optimiser steps NOT fused into the
    backward kernels, so the step runs fnr_field_mlp_bwd_rays and FusedAdam.step launches fnr_adam_step_spans) poisons its
    recording: the shape is marked with the entry point's name, every such step is interpreted, nothing is replayed."""
    import fruitnerf_amd.training as T

    def run():
        saved, T.FUSE_WEIGHT_OPTIMIZER = T.FUSE_WEIGHT_OPTIMIZER, False
        try:
            loop, hm, opt, batcher, cam = _loop(dev)
            for _ in range(16):
                loop.step()
            torch.cuda.synchronize()
            return hm.arena().params.clone(), dict(loop.stats), dict(loop._unrecordable)
        finally:
            T.FUSE_WEIGHT_OPTIMIZER = saved
    p_n, stats, why = run()
    p_i, _, _ = _interpreted(run)
    assert torch.equal(p_n, p_i)
    assert stats["replayed"] == 0 and stats["record_failed"] >= 1
    assert why and all("ran while recording and cannot be replayed" in r for r in why.values()), why
    assert any("fnr_field_mlp_bwd_rays" in r or "fnr_adam_step" in r for r in why.values()), why


def test_replayed_step_enqueues_faster_than_the_interpreter(dev):
    """What the sequencer is for: host time per enqueued step (no synchronisation inside the window) at least 1.8x lower than
    the interpreted loop's on the same box — measured on 1024 rays, where the GPU side is short enough not to throttle
    the host through a full queue."""
    import time

    def enqueue_ms(steps=60):
        loop, hm, opt, batcher, cam = _loop(dev, n_rays=1024)
        for _ in range(30):
            loop.step()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(steps):
                loop.step()
            best = min(best, (time.perf_counter() - t0) / steps * 1e3)
            torch.cuda.synchronize()
        return best, dict(loop.stats)
    native, stats = enqueue_ms()
    interp, _ = _interpreted(enqueue_ms)
    print(f"[sequencer] host enqueue per step: replayed {native:.3f} ms, interpreted {interp:.3f} ms, stats {stats}")
    assert stats["replayed"] > 150
    assert native * 1.8 <= interp
