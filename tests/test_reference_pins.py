"""CPU: pins produced by EXECUTING THE REFERENCE'S OWN CODE (tests/golden/reference_pins.npz, generator:
tests/golden/make_reference_golden.py, which imports /root/reference with its third-party imports stubbed).  Both the
oracle's restatement and the product's host mirror must reproduce them bit for bit: export lattice
(fruit_datamanager.py:42-121), orthographic ray batches (ray_generators.py:46-66), the uniform export sampler
(ray_samplers.py:54-104) and the centre-distance cluster merge (clustering_base.py:209-258)."""
import os

import numpy as np
import pytest
import torch

PINS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_pins.npz")
AABB_NAMES = ("cube", "slab", "tree", "above")


@pytest.fixture(scope="module")
def pins():
    return np.load(PINS)


def _aabb(pins, name):
    a = pins[f"aabb::{name}"]
    return tuple(map(float, a[0])), tuple(map(float, a[1]))


@pytest.mark.parametrize("name", AABB_NAMES)
def test_export_lattice_matches_the_reference(pins, name):
    from oracle import fruit_oracle as fo
    from fruitnerf_amd.data import fruit_datamanager as hd
    aabb = _aabb(pins, name)
    for corners in (fo.get_corners_of_aabb(aabb), hd.get_corners_of_aabb(aabb, device="cpu")):
        assert np.array_equal(corners.numpy(), pins[f"corners::{name}"])
    corners = fo.get_corners_of_aabb(aabb)
    for n in (8, 10):
        for pts, vec in (fo.sample_surface_points(corners, n), hd.sample_surface_points(corners, n, device="cpu")):
            assert np.array_equal(pts.numpy(), pins[f"surface::{name}::{n}"])
            assert np.array_equal(vec.numpy(), pins[f"plane_vector::{name}::{n}"])
    for n in (256, 1000):                                   # int(dx/dz * n) in float32, incl. the 0.6 * 1000 case
        want = int(pins[f"surface_count::{name}::{n}"])
        x, y, z0, vec = hd.surface_lattice_axes(corners, n)
        assert x.numel() * y.numel() == want
        pts, _ = fo.sample_surface_points(corners, n)
        assert pts.shape[0] == want
        assert np.array_equal(torch.stack([pts[0], pts[1], pts[-1]]).numpy(), pins[f"surface_ends::{name}::{n}"])


def test_export_datamanager_counts_match_the_reference(pins):
    from fruitnerf_amd.data.fruit_datamanager import ExportDataManager
    for name in AABB_NAMES:
        dm = ExportDataManager("cpu", eval_num_rays_per_batch=4096)
        assert dm.setup_inference(_aabb(pins, name), 256) == int(pins[f"surface_count::{name}::256"])
        lat = dm.export_lattice
        assert lat["xs"].numel() * lat["ys"].numel() == int(pins[f"surface_count::{name}::256"])


def test_orthographic_ray_batches_match_the_reference(pins):
    from oracle import fruit_oracle as fo
    from fruitnerf_amd.components.ray_generators import OrthographicRayGenerator
    aabb = _aabb(pins, "tree")
    corners = fo.get_corners_of_aabb(aabb)
    pts, vec = fo.sample_surface_points(corners, 10)
    assert pts.shape[0] == int(pins["gen::n_points"])
    gens = (fo.OrthographicRayGenerator(pts, vec, 8),
            OrthographicRayGenerator(surface_points=pts, plane_normal=vec, ray_batch_size=8, device="cpu", aabb=aabb))
    for gen in gens:
        for count in range(1, int(pins["gen::n_batches"]) + 1):
            rb = gen(count)
            for k in ("origins", "directions", "pixel_area", "nears", "fars"):
                assert np.array_equal(getattr(rb, k).numpy(), pins[f"gen::{count}::{k}"]), (count, k)
    start, end = gens[1].batch_range(3)
    assert (start, end) == (16, 20)


@pytest.mark.parametrize("mode", ["eval", "train", "train_single"])
def test_uniform_sampler_matches_the_reference(pins, mode):
    from oracle import fruit_oracle as fo
    from oracle import ns_torch as ns
    R = pins["smp::nears"].shape[0]
    rb = ns.RayBundle(torch.zeros(R, 3), torch.ones(R, 3), torch.zeros(R, 1), nears=torch.from_numpy(pins["smp::nears"]),
                      fars=torch.from_numpy(pins["smp::fars"]))
    smp = fo.UniformSamplerWithNoise(num_samples=6, single_jitter=(mode == "train_single"))
    smp.train(mode != "eval")
    t_rand = None if mode == "eval" else torch.from_numpy(pins[f"smp::{mode}::t_rand"])
    rs = smp(rb, t_rand=t_rand)
    fr = rs.frustums
    assert np.array_equal(fr.starts.numpy(), pins[f"smp::{mode}::bin_starts"])
    assert np.array_equal(fr.ends.numpy(), pins[f"smp::{mode}::bin_ends"])
    assert np.array_equal(rs.spacing_starts.numpy(), np.broadcast_to(pins[f"smp::{mode}::spacing_starts"],
                                                                      rs.spacing_starts.shape))
    assert np.array_equal(rs.spacing_ends.numpy(), np.broadcast_to(pins[f"smp::{mode}::spacing_ends"],
                                                                    rs.spacing_ends.shape))
    probe = torch.linspace(0, 1, 4)[None, :].expand(R, -1)
    assert np.array_equal(rs.spacing_to_euclidean_fn(probe).numpy(), pins[f"smp::{mode}::spacing_to_euclidean(probe)"])


def test_merge_small_clusters_matches_the_reference(pins):
    from fruitnerf_amd.clustering.clustering_base import FruitClustering
    fc = FruitClustering(cluster_merge_distance=0.04)
    Xs, ls = fc.merge_small_clusters(pins["merge::X"], None, pins["merge::labels"])
    assert len(Xs) == int(pins["merge::n_clusters"])
    assert fc.counter == int(pins["merge::counter"]) and fc.fuse_counter == int(pins["merge::fuse_counter"])
    for i, (x, lab) in enumerate(zip(Xs, ls)):
        assert np.array_equal(x, pins[f"merge::cluster::{i}"])
        assert np.array_equal(lab, pins[f"merge::cluster_labels::{i}"])
    assert np.array_equal(np.vstack(fc.cluster_center), pins["merge::centres"])


# ---- the reference's own FruitField class, run over oracle/ns_torch.py (tests/golden/reference_field.npz) -----------

FIELD_PINS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_field.npz")


def _field_case(g, name):
    from oracle import ns_torch as ns
    t = lambda k: torch.from_numpy(g[f"{name}::in::{k}"])  # noqa: E731
    edges = t("edges")
    rb = ns.RayBundle(t("origins"), t("directions"), torch.full((edges.shape[0], 1), 1e-6), camera_indices=t("cam"),
                      nears=edges[:, :1], fars=edges[:, -1:])
    return rb.get_ray_samples(bin_starts=edges[:, :-1, None], bin_ends=edges[:, 1:, None])


@pytest.mark.parametrize("name,test_mode,contract,training", [("train", None, True, True), ("eval", None, True, False),
                                                               ("inference", "inference", True, False),
                                                               ("export", "export", False, False)])
def test_oracle_field_is_the_reference_field(name, test_mode, contract, training):
    """oracle/fruit_oracle.py::FruitField vs the outputs of /root/reference/fruit_nerf/fruit_field.py::FruitField
    (executed by tests/golden/make_reference_field_golden.py over the same nerfstudio restatement): bit for bit, in
    every mode, including the gradients of a fixed scalar through all branches."""
    from oracle import fruit_oracle as fo
    from oracle import ns_torch as ns
    g = np.load(FIELD_PINS)
    small = np.load(os.path.join(os.path.dirname(FIELD_PINS), "fruit_nerf_small.npz"))
    sd = {k[len("sd::field."):]: torch.from_numpy(small[k]) for k in small.files if k.startswith("sd::field.")}
    field = fo.FruitField(sd["aabb"], num_images=int(g["n_images"]), log2_hashmap_size=10, test_mode=test_mode,
                          use_average_appearance_embedding=True,
                          spatial_distortion=ns.SceneContraction(order=float("inf")) if contract else None)
    assert sorted(field.state_dict().keys()) == list(g["state_dict_keys"])
    field.load_state_dict(sd, strict=True)
    field.train(training)
    res = field(_field_case(g, name))
    for head, v in res.items():
        assert np.array_equal(v.detach().numpy(), g[f"{name}::out::{head}"]), (name, head)
    assert np.array_equal(field._sample_locations.detach().numpy(), g[f"{name}::out::sample_locations"])
    assert np.array_equal(field._density_before_activation.detach().numpy(),
                          g[f"{name}::out::density_before_activation"])
    if not training:
        return
    gen = torch.Generator().manual_seed(9)
    loss = sum((res[k] * torch.rand(res[k].shape, generator=gen)).sum() for k in ("semantics", "rgb", "density"))
    loss.backward()
    assert loss.item() == float(g[f"{name}::loss"])
    for pname, p in field.named_parameters():
        if "hash_table" in pname:
            want = float(g[f"{name}::gradsum::{pname}"])
            assert abs(p.grad.double().abs().sum().item() - want) <= 1e-6 * want
        elif p.grad is not None:
            assert np.array_equal(p.grad.numpy(), g[f"{name}::grad::{pname}"]), pname
    assert np.array_equal(field._sample_locations.grad.numpy(), g[f"{name}::grad::sample_locations"])


# ---- the reference's own FruitModel class, run over oracle/ns_torch.py (tests/golden/reference_model.npz) -----------

MODEL_PINS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_model.npz")


def _oracle_model(test_mode):
    from oracle import fruit_oracle as fo
    from tests import util
    small = np.load(os.path.join(os.path.dirname(MODEL_PINS), "fruit_nerf_small.npz"))
    sd = {k[4:]: torch.from_numpy(small[k]) for k in small.files if k.startswith("sd::")}
    m = fo.FruitModel(util.small_config(log2=10, prop_log2=8), num_train_data=5, aabb=sd["field.aabb"],
                      test_mode=test_mode)
    m.load_state_dict(sd, strict=True)
    return m


def _model_inputs():
    from tests.golden.make_reference_model_golden import inputs
    return inputs()


def test_oracle_model_trains_like_the_reference_model():
    """14 training iterations driven the way the reference's callbacks drive them (anneal before, step_cb after):
    losses, distortion metric, anneal value and the proposal networks' update schedule at every step; outputs, labels,
    weights and per-parameter gradient sums at steps 0, 11 and 13 — all bit for bit."""
    from oracle import ns_torch as ns
    g = np.load(MODEL_PINS)
    o, d, pa, cam, batch = _model_inputs()
    m = _oracle_model("val")
    m.train()
    assert sorted(m.get_param_groups().keys()) == list(g["param_groups"])
    updated = []
    for step in range(14):
        m.set_anneal(step)
        torch.manual_seed(1000 + step)
        res = m(ns.RayBundle(o, d, pa, camera_indices=cam))
        ld = m.get_loss_dict(res, batch)
        md = m.get_metrics_dict(res, batch)
        m.zero_grad()
        sum(ld.values()).backward()
        assert set(ld) == {"rgb_loss", "semantics_loss", "interlevel_loss"}
        for k, v in ld.items():
            assert np.float32(v.item()) == g[f"train::{step}::loss::{k}"], (step, k)
        assert np.float32(md["distortion"].item()) == g[f"train::{step}::distortion"]
        assert m.proposal_sampler._anneal == float(g[f"train::{step}::anneal"])
        assert res["weights_list"][0].requires_grad == bool(g[f"train::{step}::prop_has_grad"])
        updated.append(bool(g[f"train::{step}::prop_has_grad"]))
        if step in (0, 11, 13):
            for k in ("rgb", "accumulation", "depth", "semantics", "prop_depth_0", "prop_depth_1"):
                assert np.array_equal(res[k].detach().numpy(), g[f"train::{step}::{k}"]), (step, k)
            assert np.array_equal(res["semantics_colormap"].numpy(), g[f"train::{step}::labels"])
            for i in range(3):
                assert np.array_equal(res["weights_list"][i].detach().numpy(), g[f"train::{step}::weights{i}"])
            for n, p in m.named_parameters():
                if n == "device_indicator_param":       # zero-length, never part of the graph (nerfstudio Model base)
                    continue
                got = p.grad.double().abs().sum().item() if p.grad is not None else 0.0
                want = float(g[f"train::{step}::gradsum::{n}"])
                # (index_add on the CPU accumulates in thread order: the sums agree to rounding, not to the bit)
                assert abs(got - want) <= 1e-6 * max(abs(want), 1e-30), (step, n)
        m.proposal_sampler.step_cb(step)
    assert all(updated[:10]) and not all(updated[10:])      # every step during the first 10, then on the schedule


@pytest.mark.parametrize("name,test_mode", [("eval", "val"), ("inference", "inference")])
def test_oracle_model_evaluates_like_the_reference_model(name, test_mode):
    from oracle import ns_torch as ns
    g = np.load(MODEL_PINS)
    o, d, pa, cam, batch = _model_inputs()
    m = _oracle_model(test_mode)
    m.eval()
    with torch.no_grad():
        res = m(ns.RayBundle(o, d, pa, camera_indices=cam))
    for k in ("rgb", "accumulation", "depth", "semantics", "prop_depth_0", "prop_depth_1"):
        assert np.array_equal(res[k].numpy(), g[f"{name}::{k}"]), k
    # the reference looks the {0, 1} label up in its colormap [0, 1] (inference mode repeats it to 3 channels)
    cm = torch.tensor([0.0, 1.0])[res["semantics_colormap"]]
    assert np.array_equal((cm.repeat(1, 3) if name == "inference" else cm).numpy(), g[f"{name}::colormap"])
    ld = m.get_loss_dict(res, batch)
    assert set(ld) == {"rgb_loss", "semantics_loss"}
    for k, v in ld.items():
        assert np.float32(v.item()) == g[f"{name}::loss::{k}"]


@pytest.mark.parametrize("name", ["export", "export_centres"])
def test_oracle_model_exports_like_the_reference_model(name):
    """"export": the flow exactly as scripts/exporter.py runs it — setup_inference() builds the sampler AFTER the
    pipeline went to eval mode, so the sampler stays in training mode and jitters every bin edge;
    "export_centres": the same with the sampler in eval mode (bin centres), which is what the product's lattice export
    and the oracle's default implement."""
    from oracle import fruit_oracle as fo
    g = np.load(MODEL_PINS)
    assert bool(g["export::sampler_training"]) and bool(g["export_centres::sampler_training"])
    m = _oracle_model("export")
    m.eval()
    m.setup_inference(render_rgb=True, num_inference_samples=9, sampler_mode_as_in_reference=(name == "export"))
    assert m.proposal_sampler.training == (name == "export")
    corners = fo.get_corners_of_aabb(((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)))
    pts, vec = fo.sample_surface_points(corners, 6)
    torch.manual_seed(77)
    with torch.no_grad():
        res = m(fo.OrthographicRayGenerator(pts, vec, 64)(1))
    assert sorted(res.keys()) == list(g[f"{name}::keys"])
    for k in ("rgb", "point_location", "semantics", "density", "semantics_colormap"):
        assert np.array_equal(res[k].numpy(), g[f"{name}::{k}"]), k
    if name == "export":        # the jitter moves samples off the lattice planes: z differs between rays
        z = g["export::point_location"][..., 2]
        assert np.ptp(z[:, 0]) > 0 and np.ptp(g["export_centres::point_location"][..., 2][:, 0]) == 0


# ---- the reference's own export loop (tests/golden/reference_export.npz) ----------------------------------------------

EXPORT_PINS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_export.npz")


@pytest.mark.parametrize("name", ["as_run", "centres"])
def test_oracle_sample_volume_is_the_reference_sample_volume(name):
    """exporter_utils.py::sample_volume driven by the reference's datamanager methods, ray generator and model:
    identical point lists, colours (incl. the per-set normalisation) and batch order — with the sampler as the
    reference leaves it (training mode, jitter from the seeded generator) and with bin centres."""
    from oracle import fruit_oracle as fo
    from tests import util
    from tests.golden.make_reference_export_golden import export_state_dict
    g = np.load(EXPORT_PINS)
    sd = export_state_dict()
    m = fo.FruitModel(util.small_config(log2=10, prop_log2=8), num_train_data=5, aabb=sd["field.aabb"],
                      test_mode="export")
    m.load_state_dict(sd, strict=True)
    m.eval()
    m.setup_inference(render_rgb=True, num_inference_samples=int(g["n_side"]),
                      sampler_mode_as_in_reference=(name == "as_run"))
    aabb = tuple(map(tuple, g["aabb"].tolist()))
    torch.manual_seed(123)
    sets = fo.sample_volume(m, aabb, int(g["n_side"]), int(g["batch"]), dataparser_scale=float(g["scale"]))
    counts = {}
    for set_name in ("semantic_colormap", "semantic", "density"):
        assert np.array_equal(sets[set_name]["points"].numpy(), g[f"{name}::{set_name}::points"]), set_name
        assert np.array_equal(sets[set_name]["colors"].numpy(), g[f"{name}::{set_name}::colors"]), set_name
        counts[set_name] = sets[set_name]["points"].shape[0]
    assert counts["density"] > counts["semantic_colormap"] > counts["semantic"] > 0
    assert str(g[f"{name}::density::path"]).endswith("scene/density.ply")


def test_reference_export_counts_depend_on_its_sampler_noise():
    g = np.load(EXPORT_PINS)
    a = [g[f"as_run::{s}::points"].shape[0] for s in ("semantic_colormap", "semantic", "density")]
    b = [g[f"centres::{s}::points"].shape[0] for s in ("semantic_colormap", "semantic", "density")]
    assert a != b


def test_second_counting_stage_matches_the_reference_split_large_cluster():
    """clustering/clustering_base.py:260-511 — the reference's own merge_small_clusters + split_large_cluster were executed
    over the oracle's restatements of alphashape / Open3D ICP / hausdorff (tests/golden/make_reference_split_golden.py);
    fruitnerf_amd.clustering (its own alpha shapes, KD-tree ICP and Hausdorff, SciPy Ward tree) must reach the same count
    through the same decisions: first-stage counters, which clusters are split / pruned, the six hypothesis distances of
    every split cluster, the final count and detection rate."""
    import time
    from fruitnerf_amd.clustering import FruitClustering
    from fruitnerf_amd.clustering import shapes
    from tests.golden.make_reference_split_golden import make_scene
    g = np.load(os.path.join(os.path.dirname(PINS), "reference_split.npz"))
    X, labels = make_scene()
    fc = FruitClustering(cluster_merge_distance=0.04)
    fc.set_template(shapes.sphere_template(float(g["template_radius"]), int(g["template_points"])))
    fc.gt_count = 11
    assert abs(fc.fruit_alpha_shape_.volume - float(g["template_volume"])) <= 1e-12
    t0 = time.time()
    Xs, ls = fc.merge_small_clusters(X, None, labels)
    assert (fc.counter, fc.fuse_counter, len(Xs)) == (int(g["counter"]), int(g["fuse_counter"]), int(g["n_merged_clusters"]))
    assert [len(c) for c in Xs] == g["merged_sizes"].tolist()
    count = fc.split_large_cluster(Xs, None, ls, seed=int(g["seed"]))
    print(f"[second stage] count {count} (reference {int(g['count'])}) in {time.time() - t0:.1f} s; decisions "
          f"{[d['fruits'] for d in fc.cluster_decisions]}")
    got = np.array([d["distances"] for d in fc.cluster_decisions if "distances" in d])
    want = g["hypothesis_distances"]
    assert got.shape == want.shape and np.abs(got - want).max() <= 1e-9
    assert [int(np.argmin(r)) + 1 for r in want] == [d["fruits"] for d in fc.cluster_decisions if "distances" in d]
    assert count == int(g["count"]) == fc.counter - fc.fuse_counter + fc.additional_count - fc.prune_counter
    assert fc.prune_counter == 2 and abs(fc.detection_rate - float(g["detection_rate"])) <= 1e-12
    # the reference's own summary lines
    text = str(g["stdout"])
    assert f"Second stage clustering count: {count}" in text
    assert f"First clustering stage count after fused (tiny) clusters: {fc.counter - fc.fuse_counter}" in text


# ---- the checkpoint contract: the reference FruitModel's state dict and its FruitPipeline.load_pipeline ---------------
# (tests/golden/reference_pipeline.npz, written by make_reference_pipeline_golden.py from the reference's own classes)

PIPELINE_PINS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_pipeline.npz")


def _product_model_small():
    from tests.golden.make_reference_pipeline_golden import product_model
    return product_model()


def test_product_state_dict_keys_are_the_reference_models():
    """Key set, order-independent, BOTH ways, with shapes and dtypes: what the reference FruitModel (over nerfstudio's
    Model base, incl. its zero-length `device_indicator_param`) puts into a checkpoint is exactly what the product model
    emits and accepts."""
    g = np.load(PIPELINE_PINS)
    want = {k: (s, d) for k, s, d in zip(g["state_keys"].tolist(), g["state_shapes"].tolist(), g["state_dtypes"].tolist())}
    sd = _product_model_small().state_dict()
    got = {k: (",".join(str(int(s)) for s in v.shape), str(v.dtype)) for k, v in sd.items()}
    assert sorted(set(want) - set(got)) == [], "keys of a reference checkpoint the product model does not have"
    assert sorted(set(got) - set(want)) == [], "keys the product model emits that the reference model does not"
    assert got == want
    assert "device_indicator_param" in got and got["device_indicator_param"][0] == "0"


@pytest.mark.parametrize("prefix", ["", "module."])
def test_reference_shaped_checkpoint_loads_the_way_load_pipeline_loads_it(prefix):
    """FruitPipeline.load_pipeline's three steps (fruit_pipeline.py:235-240) on a checkpoint with the reference's keys:
    strip `module.`, `model.update_to_step(step)`, `load_state_dict(state, strict=True)`."""
    g = np.load(PIPELINE_PINS)
    assert bool(g["loaded::plain"]) and bool(g["loaded::ddp"])       # the reference's own method ran at generation time
    gen = torch.Generator().manual_seed(3)
    ckpt = {}
    for k, s, d in zip(g["state_keys"].tolist(), g["state_shapes"].tolist(), g["state_dtypes"].tolist()):
        shape = tuple(int(x) for x in s.split(",")) if s else ()
        dtype = getattr(torch, d.replace("torch.", ""))
        ckpt[prefix + k] = (torch.rand(shape, generator=gen) if dtype.is_floating_point
                            else torch.randint(1, 30, shape, generator=gen)).to(dtype)
    model = _product_model_small()
    # (parameters under two names in every checkpoint — `mlp_base = Sequential(mlp_base_grid, mlp_base_mlp)`,
    #  fruit_field.py:141 — hold one value)
    first = {}
    for k, v in model.state_dict().items():
        if v.numel():
            ckpt[prefix + k] = ckpt[prefix + first.setdefault(v.data_ptr(), k)]
    state = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in ckpt.items()}
    model.update_to_step(1234)
    missing, unexpected = model.load_state_dict(state, strict=True)
    assert not missing and not unexpected
    after = model.state_dict()
    for k, v in state.items():
        assert torch.equal(after[k], v), k
    # the zero-length parameter belongs to no optimiser group and never reaches the arena
    assert all(p.numel() > 0 for ps in model.get_param_groups().values() for p in ps)


def test_update_to_step_invalidates_what_was_sampled_ahead():
    model = _product_model_small()
    before = model.lookahead_version()
    model.update_to_step(10)
    assert model.lookahead_version() != before


@pytest.mark.skipif(not os.path.isdir("/root/reference/fruit_nerf"), reason="needs the reference sources (build box only)")
@pytest.mark.parametrize("prefix", ["", "module."])
def test_reference_load_pipeline_runs_on_the_product_model(prefix):
    """The reference's OWN FruitPipeline.load_pipeline, executed live on a pipeline whose `_model` is the product model
    (in a subprocess: the stub import machinery it needs must not leak into this test session)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, torch; sys.path.insert(0, %r)\n"
            "from tests.golden import make_reference_pipeline_golden as m\n"
            "ref = m.reference_model().state_dict()\n"
            "pipe, loaded = m.run_reference_load_pipeline(%r, ref)\n"
            "got = pipe._model.state_dict()\n"
            "assert set(got) == set(ref), (set(got) ^ set(ref))\n"
            "assert all(torch.equal(got[k], v) for k, v in ref.items())\n"
            "print('loaded', len(ref))\n" % (root, prefix))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "loaded 42" in res.stdout
