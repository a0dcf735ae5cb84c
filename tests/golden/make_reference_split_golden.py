"""Pins for the second counting stage: the reference's OWN `FruitClustering.merge_small_clusters` +
`split_large_cluster` (/root/reference/clustering/clustering_base.py:209-511) executed in this container.

The module imports open3d, alphashape and hausdorff at the top; none of them is installable here.  They are replaced by
thin modules whose functions are the oracle's restatements of those libraries' published algorithms (oracle/cloud.py:
alphashape_3d, registration_icp, hausdorff_distance, the Open3D-shaped point-cloud container) — so the control flow that
is FruitNeRF's own (the 0.9 / 0.3 volume tests against the template's alpha shape, the one-template ICP hypothesis against
the 2..6-way Ward splits scored by Hausdorff distance, argmin, `count = counter - fuse_counter + additional - prune`) is
the reference's code running, not a restatement.  scikit-learn (AgglomerativeClustering), matplotlib and tqdm are the real
packages.  Open3D draws the 1000 surface samples from a clock-seeded generator; the stand-in draws them from
np.random.default_rng(seed + cluster index), the convention fruitnerf_amd.clustering uses.

    python tests/golden/make_reference_split_golden.py        ->  tests/golden/reference_split.npz
"""
import os
import sys
import types

import numpy as np

REFERENCE = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
TEMPLATE_RADIUS, TEMPLATE_POINTS, SEED = 0.08, 1500, 11


def ball(rng, centre, radius, n):
    p = rng.normal(size=(n, 3))
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    return np.asarray(centre) + p * radius * rng.random((n, 1)) ** (1.0 / 3.0)


def make_scene(seed: int = 5):
    """The labelled cloud handed to merge_small_clusters: single fruits, two doubles, a triple, crumbs that are pruned,
    satellites that are fused into their neighbour, noise.  -> (X [n,3], labels [n])."""
    rng = np.random.default_rng(seed)
    R = TEMPLATE_RADIUS
    parts = []
    for k in range(4):                                             # four single fruits, a little smaller / larger than the template
        parts.append(ball(rng, [0.6 * k, 0.0, 0.0], R * (0.92 + 0.05 * k), 2600))
    parts.append(np.vstack([ball(rng, [0.0, 0.8, 0.0], R, 2600), ball(rng, [1.45 * R, 0.8, 0.0], R, 2600)]))        # double
    parts.append(np.vstack([ball(rng, [0.8, 0.8, 0.0], R, 2400), ball(rng, [0.8, 0.8 + 1.5 * R, 0.02], R, 2400)]))  # double
    parts.append(np.vstack([ball(rng, [1.6 + 1.5 * R * i, 0.8, 0.0], R, 2200) for i in range(3)]))                  # triple
    parts.append(ball(rng, [0.0, 1.6, 0.0], 0.35 * R, 700))         # crumbs: far below 0.3 template volumes
    parts.append(ball(rng, [0.5, 1.6, 0.0], 0.4 * R, 700))
    parts.append(ball(rng, [0.6 + 0.03, 0.0, 0.005], 0.2 * R, 150))  # satellites inside cluster_merge_distance of fruit 1 / 2
    parts.append(ball(rng, [1.2 - 0.02, 0.01, 0.0], 0.2 * R, 150))
    X = np.vstack(parts + [rng.uniform(2.0, 3.0, (40, 3))])
    labels = np.concatenate([np.full(len(p), i) for i, p in enumerate(parts)] + [np.full(40, -1)])
    perm = rng.permutation(len(X))
    return X[perm], labels[perm]


def install_library_stand_ins(seed: int):
    sys.path.insert(0, ROOT)
    from oracle import cloud as oc
    o3d = types.ModuleType("open3d")
    o3d.geometry = types.SimpleNamespace(PointCloud=oc.O3dPointCloud, TriangleMesh=object)
    o3d.utility = types.SimpleNamespace(Vector3dVector=lambda a: np.asarray(a, dtype=np.float64),
                                        Vector3iVector=lambda a: np.asarray(a))

    class Estimation:
        def __init__(self, with_scaling=False):
            self.with_scaling = with_scaling

    class Criteria:
        def __init__(self, relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=30):
            self.relative_fitness, self.relative_rmse, self.max_iteration = relative_fitness, relative_rmse, max_iteration

    o3d.pipelines = types.SimpleNamespace(registration=types.SimpleNamespace(
        registration_icp=oc.registration_icp, TransformationEstimationPointToPoint=Estimation,
        ICPConvergenceCriteria=Criteria))
    o3d.io = types.SimpleNamespace(write_point_cloud=lambda *a, **k: True, read_point_cloud=None)
    o3d.visualization = types.SimpleNamespace(draw_geometries=lambda *a, **k: None)
    sys.modules["open3d"] = o3d
    state = {"surface_calls": 0}
    alpha_mod = types.ModuleType("alphashape")

    def alphashape(points, alpha):
        s = 0
        if alpha == 100:                      # the surface that gets sampled: one such call per cluster, in cluster order
            s = seed + state["surface_calls"]
            state["surface_calls"] += 1
        return oc.alphashape_3d(np.asarray(points), alpha, seed=s)

    alpha_mod.alphashape = alphashape
    sys.modules["alphashape"] = alpha_mod
    h = types.ModuleType("hausdorff")
    h.hausdorff_distance = oc.hausdorff_distance
    sys.modules["hausdorff"] = h
    sys.path.insert(0, REFERENCE)
    return oc


def sphere_template(radius, n):
    """Same construction as fruitnerf_amd.clustering.shapes.sphere_template (restated: the generator must not depend on
    the product)."""
    k = np.arange(n) + 0.5
    z = 1.0 - 2.0 * k / n
    phi = k * np.pi * (3.0 - np.sqrt(5.0))
    rho = np.sqrt(1.0 - z * z)
    return radius * np.stack([rho * np.cos(phi), rho * np.sin(phi), z], axis=1)


def main():
    import contextlib
    import io
    import logging
    oc = install_library_stand_ins(SEED)
    level = logging.getLogger().level
    from clustering import clustering_base as ref        # the reference's module (sets the root logger to ERROR on import)
    logging.getLogger().setLevel(level)
    X, labels = make_scene()
    fc = ref.FruitClustering(cluster_merge_distance=0.04)
    fc.fruit_template = oc.O3dPointCloud()
    fc.fruit_template.points = sphere_template(TEMPLATE_RADIUS, TEMPLATE_POINTS)
    fc.fruit_template.translate(-fc.fruit_template.get_center())           # run_clustering.py:43
    import alphashape
    fc.fruit_alpha_shape_ = alphashape.alphashape(np.asarray(fc.fruit_template.points), 10)   # run_clustering.py:44
    fc.gt_cluster, fc.gt_count, fc.pcd_path = None, 11, "/tmp/reference_split/semantic.ply"
    decisions = []
    orig_argmin = np.argmin

    def spying_argmin(a, *args, **kw):        # the reference keeps the six hypothesis distances only inside argmin
        if isinstance(a, list) and len(a) == 6 and not args and not kw:
            decisions.append([float(v) for v in a])
        return orig_argmin(a, *args, **kw)

    with contextlib.redirect_stdout(io.StringIO()) as out:
        Xs, ls = fc.merge_small_clusters(X, None, labels)
        ref.np.argmin = spying_argmin
        try:
            count = fc.split_large_cluster(Xs, None, ls)
        finally:
            ref.np.argmin = orig_argmin
    text = out.getvalue()
    res = {"count": np.int64(count), "counter": np.int64(fc.counter), "fuse_counter": np.int64(fc.fuse_counter),
           "n_merged_clusters": np.int64(len(Xs)), "template_volume": np.float64(fc.fruit_alpha_shape_.volume),
           "hypothesis_distances": np.array(decisions, dtype=np.float64), "detection_rate": np.float64(fc.detection_rate),
           "seed": np.int64(SEED), "template_radius": np.float64(TEMPLATE_RADIUS), "template_points": np.int64(TEMPLATE_POINTS),
           "merged_sizes": np.array([len(c) for c in Xs], dtype=np.int64),
           "stdout": np.array(text)}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_split.npz")
    np.savez_compressed(path, **res)
    print(text)
    print({k: (v.tolist() if v.size < 50 else v.shape) for k, v in res.items() if k != "stdout"})
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
