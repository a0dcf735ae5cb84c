#!/usr/bin/env python3
"""Generates tests/golden/cloud_small.npz: a seeded lattice "export" cloud and what the reference's CPU libraries make
of it — scikit-learn's DBSCAN labels (the real dependency, imported here) and the oracle's restatement of Open3D's
radius-outlier mask and voxel down-sampling.  Re-run only when the oracle is deliberately changed:
    python tests/golden/make_cloud_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cloud as oc  # noqa: E402

PARAMS = dict(nb_points=20, radius=0.012, voxel_size=0.003, eps=0.012, min_samples=15)


def make_cloud():
    rng = np.random.default_rng(2024)
    h = 0.004
    parts = []
    for c in rng.uniform(-0.3, 0.3, (6, 3)):
        g = np.stack(np.meshgrid(*[np.arange(-6, 7)] * 3, indexing="ij"), -1).reshape(-1, 3) * h
        parts.append(np.round(c / h) * h + g[(g * g).sum(1) <= (0.02 + 0.004 * rng.random()) ** 2])
    parts.append(np.round(rng.uniform(-0.4, 0.4, (500, 3)) / h) * h)
    X = np.concatenate(parts)
    rng.shuffle(X)
    return X, np.clip(np.abs(X) * 2.5, 0, 1)


def main():
    import sklearn
    X, C = make_cloud()
    keep = oc.remove_radius_outlier(X, PARAMS["nb_points"], PARAMS["radius"])
    vx, vc = oc.voxel_down_sample(X[keep], C[keep], PARAMS["voxel_size"])
    labels = oc.dbscan(vx, PARAMS["eps"], PARAMS["min_samples"])
    out = dict(points=X, colors=C, counts_strict=oc.radius_neighbor_counts(X, PARAMS["radius"], False),
               counts_inclusive=oc.radius_neighbor_counts(X, PARAMS["radius"], True), keep=keep, voxel_points=vx,
               voxel_colors=vc, labels=labels, sklearn_version=np.array(sklearn.__version__),
               **{"param_" + k: np.array(v) for k, v in PARAMS.items()})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cloud_small.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: {len(X)} points, {keep.sum()} kept, {len(vx)} voxels, {labels.max() + 1} clusters, "
          f"{(labels == -1).sum()} noise, sklearn {sklearn.__version__}")


if __name__ == "__main__":
    main()
