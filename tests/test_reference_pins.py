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
            assert p.grad.double().abs().sum().item() == float(g[f"{name}::gradsum::{pname}"])
        elif p.grad is not None:
            assert np.array_equal(p.grad.numpy(), g[f"{name}::grad::{pname}"]), pname
    assert np.array_equal(field._sample_locations.grad.numpy(), g[f"{name}::grad::sample_locations"])
