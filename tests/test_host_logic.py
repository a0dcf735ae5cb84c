"""CPU: host-side logic of the Python mirror (no kernel launches): checkpoint contract, export lattice,
ray batches, proposal update schedule / anneal, flat parameter arena, learning-rate schedule."""
import numpy as np
import pytest
import torch

from oracle import fruit_oracle as fo
from tests import util


def _models(cfg=None):
    from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
    from fruitnerf_amd.data.semantics import apple_metadata
    cfg = cfg or util.small_config(log2=8, prop_log2=6)
    om = util.make_oracle(cfg, num_images=5, seed=0, randomize=False)
    hc = FruitNerfModelConfig()
    for k, v in vars(cfg).items():
        if hasattr(hc, k):
            setattr(hc, k, v)
    hm = FruitModel(hc, apple_metadata(), num_train_data=5, device="cpu")
    return om, hm


def test_state_dict_keys_and_shapes_match_the_torch_layout():
    om, hm = _models()
    so, sh = om.state_dict(), hm.state_dict()
    assert list(so.keys()) == list(sh.keys())  # SURVEY Appendix C incl. the mlp_base.0 / mlp_base.1 aliases
    for k in so:
        assert tuple(so[k].shape) == tuple(sh[k].shape), k
    hm.load_state_dict(so, strict=True)  # fruit_pipeline.py:240
    assert set(hm.get_param_groups()) == {"proposal_networks", "fields"}  # fruit_nerf.py:185-189
    n_o = sum(p.numel() for p in om.parameters())
    n_h = sum(p.numel() for p in hm.parameters())
    assert n_o == n_h


def test_full_config_parameter_count():
    from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
    from fruitnerf_amd.data.semantics import apple_metadata
    hm = FruitModel(FruitNerfModelConfig(), apple_metadata(), num_train_data=90, device="cpu")
    n = sum(p.numel() for p in hm.parameters())
    assert n == 16 * 2 ** 19 * 2 + 2 * (5 * 2 ** 17 * 2 + 16 * 10 + 16 + 16 + 1) + (64 * 32 + 64 + 16 * 64 + 16) \
        + (64 * 15 + 64 + 64 * 64 + 64 + 64 + 1) + (64 * 63 + 64 + 64 * 64 + 64 + 3 * 64 + 3) + 90 * 32
    assert hm.field.mlp_base_grid.scalings == [16, 22, 30, 42, 58, 80, 111, 153, 212, 294, 406, 561, 776, 1072, 1482,
                                               2047]
    assert hm.proposal_networks[0].encoding.scalings == [16, 26, 45, 76, 128]
    assert hm.proposal_networks[1].encoding.scalings == [16, 32, 64, 128, 256]


def test_ignored_config_fields_do_not_reach_the_field():
    """hidden_dim / hidden_dim_color / appearance_embed_dim of fruit_nerf_big are silently ignored."""
    from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
    from fruitnerf_amd.data.semantics import apple_metadata
    cfg = FruitNerfModelConfig(log2_hashmap_size=6, hidden_dim=128, hidden_dim_color=128, appearance_embed_dim=128)
    cfg.proposal_net_args_list = util.small_config(prop_log2=5).proposal_net_args_list
    hm = FruitModel(cfg, apple_metadata(), num_train_data=3, device="cpu")
    assert hm.field.mlp_base_mlp.layers[0].weight.shape == (64, 32)
    assert hm.field.mlp_head.layers[0].weight.shape == (64, 16 + 15 + 32)
    assert hm.field.embedding_appearance.embedding.weight.shape == (3, 32)


@pytest.mark.parametrize("aabb,n", [(((-1, -1, -1), (1, 1, 1)), 50), (((-1.0, -0.6, -1.0), (1.0, 0.6, 1.0)), 40),
                                    (((-0.5, -0.25, -1.0), (0.5, 0.25, 0.5)), 33)])
def test_export_lattice_matches_the_oracle(aabb, n):
    from fruitnerf_amd.data.fruit_datamanager import ExportDataManager, get_corners_of_aabb, sample_surface_points
    pts_o, vec_o = fo.sample_surface_points(fo.get_corners_of_aabb(aabb), n)
    pts_h, vec_h = sample_surface_points(get_corners_of_aabb(aabb), n)
    assert torch.equal(pts_o, pts_h) and torch.equal(vec_o, vec_h)
    dm = ExportDataManager("cpu", eval_num_rays_per_batch=97)
    num = dm.setup_inference(aabb=aabb, num_points=n)
    assert num == pts_o.shape[0]
    lat = dm.export_lattice
    assert lat["xs"].numel() * lat["ys"].numel() == num and lat["zs"].numel() == n
    # fused-path lattice == positions of the generic path (origins + dir * (t0 + t1) / 2)
    gen_o = fo.OrthographicRayGenerator(pts_o, vec_o, 97)
    total = 0
    for count in range(1, (num + 96) // 97 + 1):
        rb_o = gen_o(count)
        rb_h, _ = dm.next_sample_volume(0)
        assert torch.equal(rb_o.origins, rb_h.origins) and torch.equal(rb_o.fars, rb_h.fars)
        assert torch.equal(rb_o.directions, rb_h.directions)
        total += rb_h.origins.shape[0]
    assert total == num
    s = fo.UniformSamplerWithNoise(num_samples=n)
    s.eval()
    rs = s(gen_o(1))
    pos = rs.frustums.get_positions()
    n_y = lat["ys"].numel()
    ray = torch.arange(pos.shape[0])
    assert torch.equal(pos[0, :, 2], lat["zs"])
    assert torch.equal(pos[:, 0, 0], lat["xs"][ray // n_y]) and torch.equal(pos[:, 0, 1], lat["ys"][ray % n_y])


def test_proposal_update_schedule_and_anneal_match_the_oracle():
    om, hm = _models()
    for step in list(range(0, 40)) + [999, 1000, 2500, 5000, 7000]:
        om.set_anneal(step)
        hm.set_anneal(step)
        assert om.proposal_sampler._anneal == pytest.approx(hm.proposal_sampler._anneal, abs=0)
    upd_o, upd_h = [], []
    for step in range(60):
        so, sh = om.proposal_sampler, hm.proposal_sampler
        uo = so._steps_since_update > so.update_sched(so._step) or so._step < 10
        uh = sh.updated_now()
        upd_o.append(bool(uo))
        upd_h.append(uh)
        if uo:
            so._steps_since_update = 0
        if uh:
            sh._steps_since_update = 0
        so.step_cb(step)
        sh.step_cb(step)
    assert upd_o == upd_h
    assert all(upd_h[:10]) and not all(upd_h[10:])


def test_param_arena_views_and_groups():
    from fruitnerf_amd.params import ParamArena
    om, hm = _models()
    arena = ParamArena([("proposal_networks", list(hm.proposal_networks.parameters())),
                        ("fields", list(hm.field.parameters()))], "cpu")
    assert arena.numel % 4 == 0 and arena.numel >= sum(p.numel() for p in hm.parameters())
    a, b = arena.group_ranges["proposal_networks"]
    c, d = arena.group_ranges["fields"]
    assert a == 0 and b == c and d == arena.numel
    p = hm.field.mlp_head.layers[2].bias
    assert p.data_ptr() >= arena.params.data_ptr() and p.grad.data_ptr() >= arena.grads.data_ptr()
    arena.grads.fill_(3.0)
    assert float(p.grad.sum()) == 9.0
    arena.params.zero_()
    assert float(hm.field.mlp_base_grid.hash_table.abs().sum()) == 0.0
    p.grad = None
    arena.reattach_grads()
    assert p.grad is not None and p.grad.data_ptr() >= arena.grads.data_ptr()


def test_fused_adam_run_planning():
    """FusedAdam.plan_runs: which arena spans one optimiser step launches — groups skipped when they got no gradient
    (torch.optim leaves grad=None parameters alone), adjacent groups merged only while learning rate AND step count
    coincide, and the span a fused kernel already updated (the main hash table inside the scatter) cut out."""
    from fruitnerf_amd.training import FusedAdam

    class _Arena:
        group_ranges = {"proposal_networks": (0, 1000), "fields": (1000, 5000)}
        params = torch.zeros(5000)
    opt = FusedAdam.__new__(FusedAdam)
    opt.arena = _Arena()
    sched = dict(lr=1e-2, lr_final=1e-4, max_steps=1000)        # fruit_nerf: both groups on the same schedule
    opt.groups = {"proposal_networks": dict(sched), "fields": dict(sched)}
    opt.step_count, opt.group_steps = 0, {"proposal_networks": 0, "fields": 0}
    lrs = opt.begin_step()
    lr0 = lrs["fields"]
    assert lrs["proposal_networks"] == lr0 == pytest.approx(1e-2) and opt.step_count == 1
    # step 1: same lr, same step count -> one run over both groups; the table span [1200, 4200) cut out of it
    assert opt.plan_runs(lrs) == [[0, 5000, lr0, "proposal_networks"]]
    assert opt.plan_runs(lrs, done=((1200, 4200),)) == [[0, 1200, lr0, "proposal_networks"],
                                                         [4200, 5000, lr0, "proposal_networks"]]
    # a step that did not train the proposal networks: the group is left out and its step count does not advance
    lrs = opt.begin_step(skip=("proposal_networks",))
    assert opt.group_steps == {"proposal_networks": 1, "fields": 2}
    runs = opt.plan_runs(lrs, skip=("proposal_networks",), done=((1200, 4200),))
    assert [r[:2] for r in runs] == [[1000, 1200], [4200, 5000]] and all(r[3] == "fields" for r in runs)
    assert runs[0][2] == pytest.approx(1e-2 * (1e-2) ** (1 / 1000))          # update k uses lr(k - 1)
    # afterwards the groups differ in step count: same lr, but no merge (bias corrections differ)
    lrs = opt.begin_step()
    assert lrs["proposal_networks"] == lrs["fields"]
    assert opt.plan_runs(lrs, done=((1000, 5000),)) == [[0, 1000, lrs["fields"], "proposal_networks"]]
    assert [r[:2] + r[3:] for r in opt.plan_runs(lrs)] == [[0, 1000, "proposal_networks"], [1000, 5000, "fields"]]
    # fruit_nerf_big: the proposal group has no schedule -> different learning rates from the second step on
    opt.groups["proposal_networks"] = dict(lr=1e-2, lr_final=None, max_steps=None)
    opt.step_count, opt.group_steps = 5, {"proposal_networks": 5, "fields": 5}
    lrs = opt.begin_step()
    assert lrs["proposal_networks"] == 1e-2 and lrs["fields"] < 1e-2 and len(opt.plan_runs(lrs)) == 2


def test_usable_cpus_respects_the_cgroup_quota(tmp_path, monkeypatch):
    """hostinfo.usable_cpus(): the CFS quota of the container caps the hardware thread count (cgroup v2 cpu.max)."""
    import builtins
    import os
    from fruitnerf_amd import hostinfo
    n = hostinfo.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    real_open = builtins.open
    for text, want in (("1600000 100000\n", 16), ("max 100000\n", 256), ("150000 100000\n", 2), ("garbage", 256)):
        def fake_open(path, *a, **k):
            if path == "/sys/fs/cgroup/cpu.max":
                f = tmp_path / "cpu.max"
                f.write_text(text)
                return real_open(f, *a, **k)
            if str(path).startswith("/sys/fs/cgroup/cpu/"):
                raise OSError("no cgroup v1 here")
            return real_open(path, *a, **k)
        monkeypatch.setattr(builtins, "open", fake_open)
        monkeypatch.setattr(os, "cpu_count", lambda: 256)
        assert hostinfo.usable_cpus() == want, text
        monkeypatch.undo()


def test_exponential_decay_schedule():
    from fruitnerf_amd.training import exponential_decay_lr
    assert exponential_decay_lr(0, 1e-2, 1e-4, 200000) == pytest.approx(1e-2)
    assert exponential_decay_lr(200000, 1e-2, 1e-4, 200000) == pytest.approx(1e-4)
    assert exponential_decay_lr(100000, 1e-2, 1e-4, 200000) == pytest.approx(1e-3)
    assert exponential_decay_lr(300000, 1e-2, 1e-4, 200000) == pytest.approx(1e-4)


def test_unsupported_configurations_fail_loudly():
    from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
    from fruitnerf_amd.data.semantics import apple_metadata
    with pytest.raises(NotImplementedError):
        FruitModel(FruitNerfModelConfig(log2_hashmap_size=4, proposal_initial_sampler="uniform"), apple_metadata(), num_train_data=1,
                   device="cpu")
    with pytest.raises(NotImplementedError):
        FruitModel(FruitNerfModelConfig(log2_hashmap_size=4, pass_semantic_gradients=True), apple_metadata(), num_train_data=1,
                   device="cpu")
    hm = FruitModel(FruitNerfModelConfig(log2_hashmap_size=4), apple_metadata(), num_train_data=1, device="cpu")
    with pytest.raises(RuntimeError, match="no CPU path"):
        hm.arena()


def test_synthetic_scene_is_seeded_and_sane():
    from fruitnerf_amd.data import synthetic_apple as sa
    s1, s2 = sa.make_scene(seed=0), sa.make_scene(seed=0)
    assert torch.equal(s1.centers, s2.centers) and s1.n_fruits == 32 and int(s1.is_fruit.sum()) == 32
    c2w = sa.make_cameras(8, seed=0)
    assert torch.allclose(c2w[:, :, 3].norm(dim=-1), torch.ones(8), atol=1e-5)
    data = sa.render_dataset(s1, c2w, H=40, W=40, fx=55.0, fy=55.0)
    assert data["images"].shape == (8, 40, 40, 3) and data["masks"].max() == 1
    frac = data["masks"].float().mean().item()
    assert 0.001 < frac < 0.3
    with pytest.raises(RuntimeError, match="no CPU path"):      # the product samples pixels on the HIP device only
        sa.PixelBatcher(data, torch.arange(8), seed=3).sample(64)
    from oracle import pixel_sampler as ops
    o, d, cam, batch = ops.sample_pixels(data, torch.arange(8), torch.rand(64, 3, generator=torch.Generator().manual_seed(3)))
    assert o.shape == (64, 3) and torch.allclose(d.norm(dim=-1), torch.ones(64), atol=1e-5)
    assert batch["image"].shape == (64, 3) and batch["fruit_mask"].shape == (64, 1) and cam.max() < 8


def test_ply_sink_roundtrip(tmp_path):
    """Open3D-layout binary PLY (double xyz + uchar rgb): header, size and quantisation."""
    import numpy as np
    from fruitnerf_amd.export import ply
    rng = np.random.default_rng(0)
    pts = rng.normal(size=(1000, 3))
    cols = rng.uniform(-0.2, 1.2, size=(1000, 3))
    path = str(tmp_path / "scene" / "semantic.ply")
    counts = ply.write_point_clouds({"semantic": {"points": pts, "colors": cols, "path": path},
                                     "density": {"points": pts[:0], "colors": cols[:0], "path": None}})
    assert counts == {"semantic": 1000, "density": 0}
    raw = open(path, "rb").read()
    head = raw[:raw.index(b"end_header\n") + len(b"end_header\n")]
    assert head.startswith(b"ply\nformat binary_little_endian 1.0\n") and b"element vertex 1000\n" in head
    assert b"property double x" in head and b"property uchar blue" in head
    assert len(raw) == len(head) + 1000 * 27
    p2, c2 = ply.read_point_cloud(path)
    assert np.array_equal(p2, pts)
    assert np.array_equal(c2, np.rint(np.clip(cols, 0, 1) * 255) / 255.0)


def test_ssim_oracle_basic_properties():
    """The SSIM oracle (oracle/image_metrics.py; the product's SSIM is the HIP kernel fnr_image_metrics, compared with it
    in tests/test_gpu_properties.py): 1 for identical images, symmetric, lower for noisier images."""
    import numpy as np
    from oracle.image_metrics import ssim
    rng = np.random.default_rng(0)
    a = rng.random((48, 40, 3))
    assert abs(ssim(a, a) - 1.0) < 1e-12
    b = np.clip(a + 0.05 * rng.standard_normal(a.shape), 0, 1)
    c = np.clip(a + 0.25 * rng.standard_normal(a.shape), 0, 1)
    assert abs(ssim(a, b) - ssim(b, a)) < 1e-12
    assert 0.0 < ssim(a, c) < ssim(a, b) < 1.0


def test_product_package_never_imports_the_oracle_or_the_tests():
    """oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch
    it.  No module under fruitnerf_amd/ (or tools/) may import it (or tests/), not even lazily inside a function."""
    import ast
    import os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    offenders = []
    for root in (os.path.join(repo, "fruitnerf_amd"), os.path.join(repo, "tools")):   # tools/: profiling + microbenchmarks
        for dirpath, _, files in os.walk(root):
            for f in files:
                if not f.endswith(".py"):
                    continue
                path = os.path.join(dirpath, f)
                for node in ast.walk(ast.parse(open(path).read())):
                    mods = []
                    if isinstance(node, ast.Import):
                        mods = [a.name for a in node.names]
                    elif isinstance(node, ast.ImportFrom) and node.level == 0:
                        mods = [node.module or ""]
                    for m in mods:
                        if m.split(".")[0] in ("oracle", "tests"):
                            offenders.append(f"{os.path.relpath(path, repo)}:{node.lineno} imports {m}")
    assert not offenders, offenders


def test_model_is_constructed_and_driven_the_way_fruit_pipeline_and_the_trainer_do():
    """FruitPipeline builds the model with exactly these keyword arguments (fruit_pipeline.py:104-112); Nerfstudio's
    Trainer then asks for the training callbacks and runs them by location around every iteration
    (fruit_nerf.py:191-223).  No GPU needed: construction, parameter groups and the callback protocol are host logic."""
    from fruitnerf_amd.data.semantics import Semantics
    from fruitnerf_amd.engine.callbacks import TrainingCallback, TrainingCallbackAttributes, TrainingCallbackLocation
    from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig

    class SceneBox:                      # nerfstudio.data.scene_box.SceneBox: the model reads .aabb
        aabb = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])

    cfg = FruitNerfModelConfig(log2_hashmap_size=6)
    cfg.proposal_net_args_list = [dict(a, log2_hashmap_size=5) for a in cfg.proposal_net_args_list]
    metadata = {"semantics": Semantics(filenames=[], classes=["apple", "stuff"], colors=torch.tensor([0.0, 1.0]),
                                       mask_classes=["apple", "stuff"])}
    model = cfg.setup(scene_box=SceneBox(), num_train_data=7, metadata=metadata, device="cpu", grad_scaler=None,
                      test_mode="val", render_rgb_inference=True)
    assert isinstance(model, FruitModel) and model.test_mode == "val" and model.field.num_images == 7
    assert set(model.get_param_groups()) == {"proposal_networks", "fields"}       # optimiser keys, fruit_nerf_config.py:47-56
    with pytest.raises(AssertionError):                                           # fruit_nerf.py:72
        FruitModel(cfg, {}, num_train_data=1, device="cpu")
    with pytest.raises(TypeError):                                                # metadata is a required argument
        FruitModel(cfg, num_train_data=1, device="cpu")                           # noqa

    callbacks = model.get_training_callbacks(TrainingCallbackAttributes(optimizers=None, grad_scaler=None, pipeline=None))
    assert len(callbacks) == 2 and all(isinstance(c, TrainingCallback) for c in callbacks)
    assert callbacks[0].where_to_run == [TrainingCallbackLocation.BEFORE_TRAIN_ITERATION]
    assert callbacks[1].where_to_run == [TrainingCallbackLocation.AFTER_TRAIN_ITERATION]
    assert all(c.update_every_num_iters == 1 for c in callbacks)
    N, slope = cfg.proposal_weights_anneal_max_num_iters, cfg.proposal_weights_anneal_slope
    smp = model.proposal_sampler
    for step in (0, 1, 250, 999, 1000, 5000):                                     # Trainer.train_iteration's skeleton
        for c in callbacks:
            c.run_callback_at_location(step, location=TrainingCallbackLocation.BEFORE_TRAIN_ITERATION)
        frac = min(max(step / N, 0), 1)
        assert smp._anneal == pytest.approx(slope * frac / ((slope - 1) * frac + 1))   # fruit_nerf.py:199-207
        before = smp._steps_since_update
        for c in callbacks:
            c.run_callback_at_location(step, location=TrainingCallbackLocation.AFTER_TRAIN_ITERATION)
        assert smp._step == step and smp._steps_since_update == before + 1        # ProposalNetworkSampler.step_cb
    cfg2 = FruitNerfModelConfig(log2_hashmap_size=6, use_proposal_weight_anneal=False)
    cfg2.proposal_net_args_list = cfg.proposal_net_args_list
    assert FruitModel(cfg2, metadata, num_train_data=1, device="cpu").get_training_callbacks(None) == []


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    """`--gpus N` is a promise about the number of ranks: inside a launcher's environment (WORLD_SIZE set) a mismatch is
    an error, never a silent N = 1 run that prints n_gpus: 1 (without WORLD_SIZE bench.py starts the N ranks itself:
    tests/test_gpu_distributed.py::test_bench_gpus_2_starts_two_ranks)."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline", "--no-quality"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "--gpus 2 but 1 rank(s)" in (out.stderr + out.stdout)


def test_bench_roofline_bookkeeping():
    """The `roofline` entry is the dominant HBM- / MFMA-bound entry point by total time over the timed window (ties by
    name), the proposal networks' L2-resident kernels are never candidates, and `frac` follows from the line's own
    numbers; `traffic` comes from the committed PMC file when it has that entry point."""
    import importlib.util
    import os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(repo, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    alg = bench.alg_table(33024.0)
    assert alg["prop_density_bwd"][0] == "l2" and alg["prop_density_fwd"][0] == "l2"
    recs = [("prop_density_bwd", 1048576, 0.30), ("hash_encode_bwd", 196608, 0.20), ("field_mlp_bwd", 196608, 0.16),
            ("hash_encode_fwd", 196608, 0.09)] * 5
    first, other = bench.pick_rooflines(recs, alg, 5)
    assert first == ("hash_encode_bwd", 196608) and other == ("field_mlp_bwd", 196608)
    # tie -> by name, deterministically
    first, _ = bench.pick_rooflines([("hash_encode_fwd", 8, 1.0), ("adam_step", 8, 1.0)], alg, 1)
    assert first == ("adam_step", 8)
    e = bench.roofline_entry("hash_encode_bwd", 196608, 0.2, 5, alg, fixed_bytes=28.0 * 16777216, fixed_note="x")
    want = (2176.0 * 196608 + 28.0 * 16777216) / 0.2e-3 / 1e9
    assert abs(e["achieved"] - want) < 1e-2 and abs(e["frac"] - want / 8000.0) < 1e-4 and e["traffic"] is None
    pmc = {"source": "s", "build": "b", "entry_points": {"fruit_nerf": {"hash_encode_bwd[196608]": {
        "bytes_per_launch": 8.0e8, "kernels": ["k_scatter_emit", "k_scatter_accumulate"]}}}}
    e = bench.roofline_entry("hash_encode_bwd", 196608, 0.2, 5, alg, 28.0 * 16777216, "x", pmc=pmc)
    assert e["traffic"] == 8.0e8 and abs(e["traffic_over_algorithmic"] - 8.0e8 / (want * 0.2e-3 * 1e9)) < 1e-3
    l2 = bench.roofline_entry("prop_density_fwd", 1048576, 0.0343, 1, alg)
    assert l2["bound"] == "l2" and l2["peak"] == bench.L2_PEAK_GBS and l2["frac"] < 1.0


def test_sampler_look_ahead_is_the_schedule():
    """ProposalNetworkSampler.updated_after(step) / FruitModel.anneal_at(step + 1) — what FruitModel.sample_ahead assumes
    about the NEXT iteration while the current one is finishing — against what that iteration then sees
    (updated_now() after step_cb(step), set_anneal(step + 1)), over the warm-up of the update schedule."""
    import numpy as np
    from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig, ProposalNetworkSampler
    cfg = FruitNerfModelConfig()

    def sched(step):   # fruit_nerf.py:131-136
        return np.clip(np.interp(step, [0, cfg.proposal_warmup], [0, cfg.proposal_update_every]), 1, cfg.proposal_update_every)
    s = ProposalNetworkSampler(num_nerf_samples_per_ray=48, num_proposal_samples_per_ray=(256, 96), single_jitter=True,
                               update_sched=sched)
    n_updates = 0
    for step in range(7000):
        updated = s.updated_now()
        if updated:
            s._steps_since_update = 0     # what the forward pass does
            n_updates += 1
        predicted = s.updated_after(step)
        s.step_cb(step)
        assert predicted == s.updated_now(), step
    assert 1000 < n_updates < 7000

    class M:   # anneal_at only reads the config
        config = cfg
    for step in (0, 1, 10, 500, 999, 1000, 5000):
        a = FruitModel.anneal_at(M, step)
        assert 0.0 <= a <= 1.0 and (step < cfg.proposal_weights_anneal_max_num_iters or a == 1.0)


def test_training_steps_draw_and_lookahead_bookkeeping(monkeypatch):
    """training.TrainingSteps on stubs (no device): iteration i runs on draw i + 1 of the batcher; the look-ahead for
    iteration i + 1 is made inside iteration i with `finishing step` i; a dropped look-ahead costs its draw and the next
    step samples at its own start; SAMPLE_AHEAD off / eval mode: no look-ahead at all."""
    import torch
    import fruitnerf_amd.training as T

    log = []

    class Model:
        training = True
        device = torch.device("cpu")

        def level0_spec(self):
            return {"S": 4, "near": 0.05, "far": 10.0, "n_jitter": 3}

        def sample_ahead(self, rb, step):
            log.append(("ahead", step, rb.presampled["draw"]))
            rb.presampled["ahead"] = {"for": step + 1}
            return True

    class Batcher:
        draws = 0
        last_presample = None

        def sample(self, n, cam, level0=None):
            assert level0 is not None
            Batcher.draws += 1
            self.last_presample = {"draw": Batcher.draws}
            z = torch.zeros(n, 3)
            return z, z, torch.zeros(n, 1, dtype=torch.long), {"image": z, "fruit_mask": torch.zeros(n, 1)}

    def fake_iteration(model, optimizer, rb, batch, step, world_size=1, want_metrics=True, camera=None, ahead=None):
        log.append(("iter", step, rb.presampled["draw"], rb.presampled.get("ahead")))
        if ahead is not None:
            ahead()
        return {}, {}

    monkeypatch.setattr(T, "fused_train_iteration", fake_iteration)
    monkeypatch.setattr(T, "SAMPLE_AHEAD", True)
    loop = T.TrainingSteps(Model(), None, Batcher(), 8)
    loop.step(); loop.step(); loop.step()
    assert log == [("iter", 0, 1, None), ("ahead", 0, 2), ("iter", 1, 2, {"for": 1}), ("ahead", 1, 3),
                   ("iter", 2, 3, {"for": 2}), ("ahead", 2, 4)]
    del log[:]
    loop.drop_lookahead()                       # draw 4 is gone; iteration 3 samples at its start
    loop.step()
    assert log == [("iter", 3, 5, None), ("ahead", 3, 6)]
    del log[:]
    monkeypatch.setattr(T, "SAMPLE_AHEAD", False)
    loop.step(); loop.step()                    # consumes the pending look-ahead, then samples per step
    assert log == [("iter", 4, 6, {"for": 4}), ("iter", 5, 7, None)]
    assert loop.step_idx == 6


def test_lookahead_is_ordered_after_the_proposal_networks_step(monkeypatch):
    """fused_train_iteration on stubs: the look-ahead (which reads the proposal networks) is handed to the backward — to be
    enqueued on the second stream — only when the networks' step of this iteration is fused into their backward or does
    not happen; when optimizer.step() still has to take it (unfused weights, torch-1.13 stepping without gradients) the
    look-ahead comes after that call."""
    import types
    import torch
    import fruitnerf_amd.training as T

    log = []

    class Sampler:
        def __init__(self, updated):
            self.updated = updated

        def updated_now(self):
            return self.updated

        def step_cb(self, step):
            log.append("step_cb")

    def make_model(updated):
        m = types.SimpleNamespace()
        m.training = True
        m.config = types.SimpleNamespace(use_same_proposal_network=False)
        m.proposal_sampler = Sampler(updated)
        m.set_anneal = lambda step: None
        m.arena = lambda: types.SimpleNamespace(group_ranges={"fields": (0, 8), "proposal_networks": (8, 12)})
        m.field = types.SimpleNamespace(mlp_base_grid=types.SimpleNamespace(hash_table=None))
        m._last_render_updated = updated
        return m

    class Opt:
        def __init__(self, skip):
            self.skip_groups_without_grad = skip

        def table_adam_args(self, table, group):
            return object(), (0, 4)

        def weight_adam_args(self, group):
            return (object(), None), (0, 8)

        def step(self, skip=(), done=()):
            log.append(("optimizer.step", tuple(skip)))

    def fake_fb(model, rb, batch, jitter, want_metrics, exchange, ray_grads, **kw):
        log.append(("backward", "fused proposal step" if kw["proposal_optimizer"] is not None else "no fused proposal step"))
        if kw["ahead"] is not None:
            kw["ahead"]()
        return {}, {}

    monkeypatch.setattr(T, "fused_forward_backward", fake_fb)
    ahead = lambda: log.append("look-ahead")   # noqa: E731

    def run(updated, skip=True, fuse_weights=True):
        del log[:]
        monkeypatch.setattr(T, "FUSE_WEIGHT_OPTIMIZER", fuse_weights)
        T.fused_train_iteration(make_model(updated), Opt(skip), None, None, 7, ahead=ahead)
        return [e if isinstance(e, str) else e[0] for e in log]

    inside = ["backward", "look-ahead", "optimizer.step", "step_cb"]
    after = ["backward", "optimizer.step", "look-ahead", "step_cb"]
    assert run(updated=True) == inside                        # the networks step inside their backward
    assert run(updated=False) == inside                       # no gradient -> the group is skipped altogether
    assert run(updated=False, skip=False) == after            # torch 1.13: the group steps on zero gradients
    assert run(updated=True, skip=False) == inside            # ... but a fused step is a fused step
    assert run(updated=True, fuse_weights=False) == after     # unfused: optimizer.step() takes the networks' step
    assert run(updated=False, fuse_weights=False) == inside


def test_method_config_entry_points_resolve_and_match_the_reference_table():
    """pyproject.toml's `nerfstudio.method_configs` entry points (reference pyproject.toml:24-27) name objects that exist
    in fruitnerf_amd.fruit_nerf_config, and the three method tables carry the reference's hyper-parameters
    (fruit_nerf_config.py:27-164): batch sizes, iteration counts, optimisers + schedulers per group, camera optimiser,
    the model fields FruitModel forwards to FruitField."""
    import importlib
    import os
    import tomli
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "pyproject.toml"), "rb") as f:
        proj = tomli.load(f)
    eps = proj["project"]["entry-points"]["nerfstudio.method_configs"]
    assert sorted(eps) == ["fruit_nerf", "fruit_nerf_big", "fruit_nerf_huge"]
    for method, target in eps.items():
        mod, attr = target.split(":")
        spec = getattr(importlib.import_module(mod), attr)
        cfg = spec.config
        name = cfg["method_name"] if isinstance(cfg, dict) else cfg.method_name
        assert name == method
        assert spec.description.startswith("Base config for FruitNeRF")
    from fruitnerf_amd import fruit_nerf_config as FC
    base, big, huge = (FC.METHODS[k] for k in ("fruit_nerf", "fruit_nerf_big", "fruit_nerf_huge"))
    assert [m["datamanager"]["train_num_rays_per_batch"] for m in (base, big, huge)] == [4096, 8192, 16384]
    assert [m["trainer"]["max_num_iterations"] for m in (base, big, huge)] == [30000, 100000, 100000]
    assert all(m["trainer"]["mixed_precision"] for m in (base, big, huge))
    assert base["optimizers"]["fields"] == dict(algorithm="adam", lr=1e-2, eps=1e-15,
                                                scheduler=dict(kind="exponential_decay", lr_final=1e-4, max_steps=200000))
    assert big["optimizers"]["proposal_networks"] == dict(algorithm="radam", lr=1e-2, eps=1e-15, scheduler=None)
    assert big["optimizers"]["fields"]["scheduler"]["max_steps"] == 50000
    assert base["camera_optimizer"]["weight_decay"] == 1e-2 and big["camera_optimizer"]["weight_decay"] == 1e-3
    assert big["camera_optimizer"]["scheduler"] is None and huge["camera_optimizer"]["scheduler"]["lr_final"] == 6e-5
    mc = FC.model_config("fruit_nerf_big")
    assert (mc.num_nerf_samples_per_ray, mc.num_proposal_samples_per_ray, mc.geo_feat_dim, mc.hidden_dim_semantics,
            mc.num_layers_semantic, mc.max_res, mc.log2_hashmap_size, mc.proposal_weights_anneal_max_num_iters) == \
        (128, (512, 256), 30, 128, 3, 4096, 21, 5000)
    mh = FC.model_config("fruit_nerf_huge")
    assert [a["num_levels"] for a in mh.proposal_net_args_list] == [5, 7] and mh.max_res == 8192
    assert FC.model_config("fruit_nerf").max_res == 2048 and FC.model_config("fruit_nerf").eval_num_rays_per_chunk == 1 << 15
    # bench.py reads its method table from here
    import bench
    assert bench.METHODS["fruit_nerf_big"]["groups"] == FC.group_schedules("fruit_nerf_big")
    assert bench.METHODS["fruit_nerf"]["rays"] == 4096 and bench.METHODS["fruit_nerf_huge"]["samples"] == (512, 512, 64)


def test_scaler_step_unscales_skips_on_inf_and_feeds_the_scalers_update():
    """training.scaler_step: what `grad_scaler.step(optimizer)` means for FusedAdam — gradients of a scaled loss are
    unscaled inside the step (grad_scale = 1 / scale), a non-finite gradient skips the step and zeroes the gradients, and
    the inf check is visible to GradScaler.update() (scale backs off after an inf, grows after `growth_interval` clean
    steps)."""
    import torch
    import fruitnerf_amd.training as T

    class Arena:
        def __init__(self):
            self.grads = torch.zeros(8)

        def zero_grad(self):
            self.grads.zero_()

    class Opt:
        def __init__(self):
            self.arena = Arena()
            self.calls = []

        def step(self, grad_scale=1.0, skip=(), done=()):
            self.calls.append((grad_scale, tuple(skip)))

    opt = Opt()
    assert T.scaler_step(opt, None, skip=("proposal_networks",)) and opt.calls == [(1.0, ("proposal_networks",))]
    scaler = torch.amp.GradScaler("cpu", init_scale=1024.0, growth_interval=2)
    loss = torch.ones(1, requires_grad=True).sum()
    scaler.scale(loss)                        # initialises the scale tensor, as a Trainer's scale(loss).backward() does
    opt.arena.grads.fill_(1024.0)
    assert T.scaler_step(opt, scaler)
    assert opt.calls[-1] == (1.0 / 1024.0, ())
    scaler.update()
    assert scaler.get_scale() == 1024.0
    opt.arena.grads[3] = float("inf")
    n = len(opt.calls)
    assert not T.scaler_step(opt, scaler)
    assert len(opt.calls) == n and float(opt.arena.grads.abs().sum()) == 0.0
    scaler.update()
    assert scaler.get_scale() == 512.0       # backoff_factor 0.5
    for _ in range(2):
        opt.arena.grads.fill_(1.0)
        assert T.scaler_step(opt, scaler)
        scaler.update()
    assert scaler.get_scale() == 1024.0      # growth after growth_interval clean steps


def test_scatter_workspace_cache_is_bounded_per_entry_point():
    """_kernels._WorkspaceCache: LRU, at most two buffers per (device, entry point); a hit reports the buffer as clean, a
    miss (first use, or after eviction) as not clean."""
    from fruitnerf_amd._kernels import _WorkspaceCache
    made = []

    def make():
        made.append(object())
        return made[-1]

    c = _WorkspaceCache(per_tag=2)
    k = lambda stream, tag, n=64: ("cuda", 0, stream, n, tag)     # noqa: E731
    b1, clean = c.get(k(1, "prop0"), make)
    assert clean == 0 and c.get(k(1, "prop0"), make) == (b1, 1)
    b2, _ = c.get(k(2, "prop0"), make)
    assert len(c) == 2 and c.get(k(1, "prop0"), make) == (b1, 1)        # 1 is now the most recent
    b3, clean = c.get(k(3, "prop0"), make)                               # evicts stream 2 (least recent), keeps 1
    assert clean == 0 and len(c) == 2
    assert c.get(k(1, "prop0"), make) == (b1, 1)
    assert c.get(k(2, "prop0"), make)[1] == 0                           # gone: a new buffer, not clean
    c.get(k(7, "field"), make)                                           # other entry points are counted separately
    c.get(k(7, "prop1"), make)
    assert len(c) == 4
    c.get(k(1, "prop0", n=128), make)                                    # another size of the same entry point counts too
    assert sum(1 for key, _ in c.items() if key[4] == "prop0") == 2


def test_a_failed_scatter_call_drops_the_cached_workspaces():
    """A scatter entry point that fails may have enqueued its emit launch without the accumulate launch that puts the
    queue counters back to zero: every cached workspace of that device is dropped, so that the next call zeroes its own."""
    import pytest as _pytest
    import torch as _torch
    from fruitnerf_amd import _kernels as K
    saved = dict(K._SCATTER_WS.entries)
    try:
        K._SCATTER_WS.clear()
        K._SCATTER_WS.entries[("cuda", 0, 11, 64, "field")] = object()
        K._SCATTER_WS.entries[("cuda", 0, 12, 64, "prop0")] = object()
        K._SCATTER_WS.entries[("cuda", 1, 11, 64, "field")] = object()
        K._scatter_check(0, "ok", _torch.device("cuda", 0))                       # success: nothing is dropped
        assert len(K._SCATTER_WS) == 3
        with _pytest.raises(RuntimeError):
            K._scatter_check(3, "hash_encode_bwd", _torch.device("cuda", 0))      # failure: device 0's buffers go, then it raises
        assert [k for k, _ in K._SCATTER_WS.items()] == [("cuda", 1, 11, 64, "field")]
    finally:
        K._SCATTER_WS.clear()
        K._SCATTER_WS.entries.update(saved)


def test_clustering_template_fails_loudly_and_falls_back_only_for_lfs_pointers(tmp_path):
    """a mistyped or corrupt template path must not silently become the 0.1-radius sphere (the template's
    volume drives the split / prune thresholds); None and a Git-LFS pointer file (what the reference's *_template.ply are
    in its repository snapshot) do fall back, and say so."""
    import numpy as np
    import pytest
    from fruitnerf_amd.clustering.clustering_base import Clustering
    assert Clustering(template_path=None).template_fallback is True
    lfs = tmp_path / "apple_template.ply"
    lfs.write_text("version https://git-lfs.github.com/spec/v1\noid sha256:0\nsize 1\n")
    assert Clustering(template_path=lfs).template_fallback is True
    with pytest.raises(FileNotFoundError):
        Clustering(template_path=tmp_path / "appel_template.ply")
    bad = tmp_path / "corrupt.ply"
    bad.write_bytes(b"ply\nformat binary_little_endian 1.0\nelement vertex 5\nproperty double x\nend_header\n\x00")
    with pytest.raises(Exception):
        Clustering(template_path=bad)
    # a readable PLY is used as is
    from fruitnerf_amd.export import ply
    good = tmp_path / "good.ply"
    pts = np.random.default_rng(0).standard_normal((50, 3)) * 0.05
    ply.write_point_cloud(str(good), pts, np.zeros((50, 3)))
    c = Clustering(template_path=good)
    assert c.template_fallback is False


def test_touched_bitmaps_are_keyed_by_arena_offset_and_dropped_by_unfused_steps(monkeypatch):
    """FusedAdam's sparse-touch bitmaps: keyed by the table's arena offset (not by id(), which can be reused),
    rebuilt from the moments, and dropped by any optimiser step over the table's span that does not go through the fused
    table kernels — on CPU tensors, with the Adam launch itself replaced (the bookkeeping is host logic)."""
    import types
    import torch
    import fruitnerf_amd.training as T
    table_a, table_b = torch.zeros(256 * 2), torch.zeros(128 * 2)
    params = torch.zeros(64 + table_a.numel() + table_b.numel())
    arena = types.SimpleNamespace(params=params, grads=torch.zeros_like(params),
                                  entries=[("w", None, 0, 64), ("a", table_a, 64, table_a.numel()),
                                           ("b", table_b, 64 + table_a.numel(), table_b.numel())],
                                  group_ranges={"fields": (0, params.numel())})
    model = types.SimpleNamespace(arena=lambda: arena)
    opt = T.FusedAdam(model)
    launches = []
    monkeypatch.setattr(T.K, "adam_step", lambda *a, **k: launches.append(a[0].numel()))
    monkeypatch.setattr(T, "SPARSE_TOUCH_SKIPPING", True)
    opt.exp_avg[64 + 8] = 1.0                                     # pair 2 of table a has a moment
    bm_a = opt.touched_bitmap(table_a, 64, table_a.numel())
    bm_b = opt.touched_bitmap(table_b, 64 + table_a.numel(), table_b.numel())
    assert bm_a.numel() == table_a.numel() // 128 and int(bm_a[0]) == 1 << 2 and int(bm_b.abs().sum()) == 0
    assert opt.touched_bitmap(table_a, 64, table_a.numel()) is bm_a             # cached under the offset
    assert set(opt._touched) == {64, 64 + table_a.numel()}
    opt.begin_step()
    opt.step_span(0, 64, 1e-2, group="fields")                    # the MLP weights only: no table span touched
    assert set(opt._touched) == {64, 64 + table_a.numel()} and launches == [64]
    opt.step_span(32, 64 + 16, 1e-2, group="fields")              # overlaps table a's head
    assert set(opt._touched) == {64 + table_a.numel()}
    assert opt.touched_bitmap(table_a, 64, table_a.numel()) is not bm_a         # rebuilt from the moments
    opt.rebuild_touched()
    assert not opt._touched
    # skipping does not apply with weight decay or to a span that is not a whole number of 128-float words
    assert T.FusedAdam(model, weight_decay=1e-3).touched_bitmap(table_a, 64, table_a.numel()) is None
    assert opt.touched_bitmap(table_a, 64, 100) is None
