#!/usr/bin/env python3
"""Generates tests/golden/reference_pins.npz by EXECUTING THE REFERENCE'S OWN CODE (read from /root/reference at
generation time only; tests read the committed .npz).  The reference's hot-path modules import nerfstudio / nerfacc /
jaxtyping / open3d / alphashape / hausdorff at module top, none of which is installable here, so those imports are
satisfied by inert stub modules; the functions pinned below are the ones whose bodies are plain torch / numpy and do
not depend on anything the stubs would have to compute:

  * fruit_nerf/data/fruit_datamanager.py::get_corners_of_aabb, sample_surface_points     (:42-121)
  * fruit_nerf/components/ray_generators.py::OrthographicRayGenerator.forward            (:46-66)
  * fruit_nerf/components/ray_samplers.py::UniformSamplerWithNoise.generate_ray_samples  (:54-104)
  * clustering/clustering_base.py::FruitClustering.merge_small_clusters                  (:209-258)

Stubs that carry state (and nothing else): `RayBundle` keeps its keyword arguments and returns the arguments of
`get_ray_samples` (so the bins the sampler computes are observable); `SpacedSampler` is an nn.Module that stores its
constructor arguments (what nerfstudio 0.3.2's Sampler/SpacedSampler __init__ does).  For the stratified (training)
branch `torch.rand` is replaced by a recorded tensor so that the jitter is an input of the fixture.

    python tests/golden/make_reference_golden.py"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch
from torch import nn

REFERENCE = "/root/reference"
STUBBED = ("nerfstudio", "nerfacc", "jaxtyping", "open3d", "alphashape", "hausdorff", "tinycudann", "pymeshlab",
           "tyro", "torchmetrics", "torchtyping")


class _Any:
    """Inert value for class attributes of stubbed classes that are touched at import time."""
    def __call__(self, *a, **k):
        return self

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return self

    def __getitem__(self, item):
        return self

    def __add__(self, other):
        return other

    __radd__ = __add__

    def __iter__(self):
        return iter(())

    def __hash__(self):
        return 0


class _DummyMeta(type):
    def __getitem__(cls, item):       # Generic[...] style subscripts
        return cls

    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Any()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = _DummyMeta(name, (), {"__init__": lambda self, *a, **k: None})
        setattr(self, name, cls)
        return cls


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in STUBBED:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        module.__path__ = []


class RayBundle:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def get_ray_samples(self, **kw):
        return kw


class SpacedSampler(nn.Module):
    def __init__(self, num_samples=None, spacing_fn=None, spacing_fn_inv=None, train_stratified=True,
                 single_jitter=False):
        super().__init__()
        self.num_samples, self.spacing_fn, self.spacing_fn_inv = num_samples, spacing_fn, spacing_fn_inv
        self.train_stratified, self.single_jitter = train_stratified, single_jitter


def install_stubs():
    sys.meta_path.insert(0, _StubFinder())
    import nerfstudio.cameras.rays as rays
    import nerfstudio.model_components.ray_samplers as samplers
    rays.RayBundle = RayBundle
    samplers.SpacedSampler = SpacedSampler
    sys.path.insert(0, REFERENCE)


AABBS = {
    "cube": ((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)),
    "slab": ((-0.6, -1.0, -1.0), (0.6, 1.0, 1.0)),          # dx / dz = 0.6 -> int(0.6 * n) columns in float32
    "tree": ((-0.45, -0.3, -0.2), (0.35, 0.5, 1.4)),
    "above": ((-1.0, -1.0, 0.25), (1.0, 1.0, 1.75)),        # z_min > 0: the plane vector's length quirk (:117-119)
}


def main():
    install_stubs()
    from fruit_nerf.data import fruit_datamanager as ref_dm
    from fruit_nerf.components import ray_generators as ref_gen
    from fruit_nerf.components import ray_samplers as ref_smp
    out = {}

    # ---- AABB corners + surface lattice ----------------------------------------------------------------------------
    for name, aabb in AABBS.items():
        out[f"aabb::{name}"] = np.array(aabb, dtype=np.float64)
        corners = ref_dm.get_corners_of_aabb(aabb=aabb, device="cpu")
        out[f"corners::{name}"] = corners.numpy()
        for n in (8, 10):
            pts, vec = ref_dm.sample_surface_points(corners, n=n, device="cpu")
            out[f"surface::{name}::{n}"] = pts.numpy()
            out[f"plane_vector::{name}::{n}"] = vec.numpy()
        for n in (256, 1000):                                 # too large to store: sizes and end points only
            pts, vec = ref_dm.sample_surface_points(corners, n=n, device="cpu")
            out[f"surface_count::{name}::{n}"] = np.int64(pts.shape[0])
            out[f"surface_ends::{name}::{n}"] = torch.stack([pts[0], pts[1], pts[-1]]).numpy()

    # ---- orthographic ray batches -----------------------------------------------------------------------------------
    corners = ref_dm.get_corners_of_aabb(aabb=AABBS["tree"], device="cpu")
    pts, vec = ref_dm.sample_surface_points(corners, n=10, device="cpu")
    gen = ref_gen.OrthographicRayGenerator(surface_points=pts, plane_normal=vec, ray_batch_size=8, device="cpu",
                                           aabb=AABBS["tree"])
    out["gen::n_points"] = np.int64(pts.shape[0])
    n_batches = -(-pts.shape[0] // 8)
    out["gen::n_batches"] = np.int64(n_batches)
    for count in range(1, n_batches + 1):
        rb = gen(count)
        for k in ("origins", "directions", "pixel_area", "nears", "fars"):
            out[f"gen::{count}::{k}"] = getattr(rb, k).numpy()

    # ---- UniformSamplerWithNoise ------------------------------------------------------------------------------------
    R, N = 5, 6
    g = torch.Generator().manual_seed(11)
    bundle = RayBundle(origins=torch.rand(R, 3, generator=g), directions=torch.rand(R, 3, generator=g),
                       nears=torch.rand(R, 1, generator=g) * 0.2, fars=1.0 + torch.rand(R, 1, generator=g))
    out["smp::nears"], out["smp::fars"] = bundle.nears.numpy(), bundle.fars.numpy()
    real_rand = torch.rand
    for mode, single in (("eval", False), ("train", False), ("train_single", True)):
        smp = ref_smp.UniformSamplerWithNoise(num_samples=N, single_jitter=single)
        smp.train(mode != "eval")
        if mode != "eval":
            t_rand = real_rand((R, 1) if single else (R, N + 1), generator=g)
            out[f"smp::{mode}::t_rand"] = t_rand.numpy()
            torch.rand = lambda *a, **k: t_rand              # noqa: E731  (the jitter becomes a fixture input)
        try:
            rs = smp.generate_ray_samples(bundle)
        finally:
            torch.rand = real_rand
        for k in ("bin_starts", "bin_ends", "spacing_starts", "spacing_ends"):
            out[f"smp::{mode}::{k}"] = rs[k].numpy()
        probe = torch.linspace(0, 1, 4)[None, :].expand(R, -1)
        out[f"smp::{mode}::spacing_to_euclidean(probe)"] = rs["spacing_to_euclidean_fn"](probe).numpy()

    # ---- merge_small_clusters ---------------------------------------------------------------------------------------
    import logging
    level = logging.getLogger().level
    from clustering import clustering_base as ref_cl         # (sets the root logger to ERROR on import)
    logging.getLogger().setLevel(level)
    rng = np.random.default_rng(5)
    centres = np.array([[0, 0, 0], [0.02, 0.0, 0.0], [0.5, 0, 0], [0.5, 0.03, 0.0], [0.52, 0.05, 0.0], [1.0, 1.0, 1.0]])
    sizes = [40, 10, 25, 30, 8, 12]
    X = np.concatenate([rng.normal(c, 0.002, (k, 3)) for c, k in zip(centres, sizes)] + [rng.uniform(2, 3, (7, 3))])
    labels = np.concatenate([np.full(k, i) for i, k in enumerate(sizes)] + [np.full(7, -1)])
    perm = rng.permutation(len(X))
    X, labels = X[perm], labels[perm]
    fc = ref_cl.FruitClustering(cluster_merge_distance=0.04)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        Xs, ls = fc.merge_small_clusters(X, None, labels)
    out["merge::X"], out["merge::labels"] = X, labels
    out["merge::n_clusters"] = np.int64(len(Xs))
    out["merge::counter"], out["merge::fuse_counter"] = np.int64(fc.counter), np.int64(fc.fuse_counter)
    for i, (x, lab) in enumerate(zip(Xs, ls)):
        out[f"merge::cluster::{i}"] = x
        out[f"merge::cluster_labels::{i}"] = lab
    out["merge::centres"] = np.vstack(fc.cluster_center)

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_pins.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
