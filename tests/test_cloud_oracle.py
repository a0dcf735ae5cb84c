"""CPU: pins of the counting-stage oracle (oracle/cloud.py).  DBSCAN is pinned on the reference's real dependency
(scikit-learn); the Open3D restatements are checked against closed forms and scipy's KD-tree (Open3D is absent)."""
import numpy as np
import pytest

from oracle import cloud as oc


def blobs(seed, n_blobs=10, per=120, sigma=0.03, noise=250):
    rng = np.random.default_rng(seed)
    cent = rng.uniform(-1, 1, (n_blobs, 3))
    X = np.concatenate([c + sigma * rng.standard_normal((per, 3)) for c in cent] + [rng.uniform(-1, 1, (noise, 3))])
    rng.shuffle(X)
    return X


@pytest.mark.parametrize("seed,eps,min_samples", [(0, 0.03, 10), (1, 0.05, 30), (2, 0.02, 5), (3, 0.08, 4)])
def test_restated_dbscan_is_sklearn(seed, eps, min_samples):
    X = blobs(seed)
    a, b = oc.dbscan(X, eps, min_samples), oc.dbscan_restated(X, eps, min_samples)
    assert a.max() >= 3
    assert np.array_equal(a, b)


def border_case():
    """Two 45-point rods (every rod point is core for eps 0.1 / min_samples 44), 0.12 apart end to end, and one point
    between them that is within eps of 21 points of each rod: 43 neighbours, not core — a border point of BOTH."""
    xs = np.arange(45) * 0.002
    a = np.stack([xs, np.zeros(45), np.zeros(45)], 1)
    b = np.stack([0.2075 + xs, np.zeros(45), np.zeros(45)], 1)
    mid = np.array([[0.1475, 0.0, 0.0]])
    return a, b, mid


def test_border_point_joins_the_lower_numbered_cluster():
    """scikit-learn gives a shared border point to the cluster that was discovered first in index order — whichever
    way the input is ordered; the restated rule (smallest adjacent cluster number) says the same."""
    a, b, mid = border_case()
    for order in ([a, b, mid], [b, mid, a], [mid, b, a], [a[:20], mid, b, a[20:]]):
        X = np.concatenate(order)
        lab = oc.dbscan(X, 0.1, 44)
        assert lab.max() == 1 and (lab >= 0).all()
        assert np.array_equal(lab, oc.dbscan_restated(X, 0.1, 44))
        m = int(np.flatnonzero((X == mid[0]).all(1))[0])
        assert lab[m] == 0
        assert oc.radius_neighbor_counts(X, 0.1, inclusive=True)[m] == 43


def test_radius_counts_against_kdtree_and_strictness():
    from scipy.spatial import cKDTree
    X = blobs(7)
    got = oc.radius_neighbor_counts(X, 0.05, inclusive=True)
    ref = np.array([len(v) for v in cKDTree(X).query_ball_point(X, 0.05)])
    assert np.array_equal(got, ref)
    # exact ties: lattice points at distance exactly r are neighbours only for the inclusive (scikit-learn) test
    g = np.stack(np.meshgrid(*[np.arange(4.0)] * 3, indexing="ij"), -1).reshape(-1, 3)
    strict = oc.radius_neighbor_counts(g, 1.0, inclusive=False)
    incl = oc.radius_neighbor_counts(g, 1.0, inclusive=True)
    assert strict.max() == 1 and incl.max() == 7 and incl.min() == 4
    keep = oc.remove_radius_outlier(g, 1, 1.0)
    assert not keep.any()                      # Open3D: strictly inside the radius, MORE than nb_points
    # just outside the tie: corners see 3 neighbours + themselves = 4, which is not MORE than 4
    assert oc.remove_radius_outlier(g, 4, 1.0001).sum() == 64 - 8

def test_voxel_down_sample_closed_form():
    pts = np.array([[0.0, 0, 0], [0.04, 0.01, 0.0], [1.0, 1.0, 1.0], [0.02, 0.0, 0.03], [1.01, 0.99, 1.0]])
    col = np.array([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0], [1.0, 1.0, 1.0], [0.5, 0.5, 0.5]])
    x, c = oc.voxel_down_sample(pts, col, 0.2)
    assert x.shape == (2, 3)
    assert np.allclose(x[0], pts[[0, 1, 3]].mean(0)) and np.allclose(c[0], col[[0, 1, 3]].mean(0))
    assert np.allclose(x[1], pts[[2, 4]].mean(0)) and np.allclose(c[1], col[[2, 4]].mean(0))
    # a voxel smaller than every gap keeps all points (sorted by voxel key), colours attached to their points
    x, c = oc.voxel_down_sample(pts, col, 1e-3)
    assert x.shape == (5, 3)
    for p, q in zip(x, c):
        i = int(np.flatnonzero(np.isclose(pts, p).all(1))[0])
        assert np.array_equal(col[i], q)
    x, c = oc.voxel_down_sample(np.zeros((0, 3)), None, 0.1)
    assert x.shape == (0, 3) and c is None


def test_merge_small_clusters_host_logic():
    from fruitnerf_amd.clustering.clustering_base import FruitClustering
    fc = FruitClustering(cluster_merge_distance=0.04)
    rng = np.random.default_rng(0)
    a = rng.normal([0, 0, 0], 0.002, (40, 3))
    b = rng.normal([0.02, 0, 0], 0.002, (10, 3))       # within 0.04 of a's centre: fused into it
    c = rng.normal([0.5, 0, 0], 0.002, (25, 3))
    X = np.concatenate([a, b, c, [[9.0, 9, 9]]])
    labels = np.array([0] * 40 + [1] * 10 + [2] * 25 + [-1])
    Xs, ls = fc.merge_small_clusters(X, None, labels)
    assert fc.counter == 3 and fc.fuse_counter == 1 and len(Xs) == 2
    assert len(Xs[0]) == 50 and len(Xs[1]) == 25 and set(ls[1]) == {1}
    assert np.allclose(fc.cluster_center[0], (a.mean(0) + b.mean(0)) / 2)


def test_second_stage_geometry_matches_the_oracle_restatements():
    """fruitnerf_amd.clustering.shapes (vectorised circumradii + KD-tree nearest neighbours + SciPy Ward tree) against the
    oracle's loop / brute-force restatements of alphashape, Open3D's ICP and surface sampling, hausdorff, and against
    scikit-learn's AgglomerativeClustering itself."""
    import numpy as np
    from sklearn.cluster import AgglomerativeClustering
    from fruitnerf_amd.clustering import shapes as S
    from oracle import cloud as oc
    rng = np.random.default_rng(1)
    p = rng.normal(size=(1200, 3))
    p = 0.08 * p / np.linalg.norm(p, axis=1, keepdims=True) * rng.random((1200, 1)) ** (1 / 3)
    for alpha in (10, 40):
        a, b = oc.alphashape_3d(p, alpha, seed=3), S.alpha_shape(p, alpha)
        assert abs(a.volume - b.volume) <= 1e-12 and np.array_equal(a.faces, b.faces)
        assert abs(b.volume - 4 / 3 * np.pi * 0.08 ** 3) <= 0.2 * 4 / 3 * np.pi * 0.08 ** 3      # a ball, roughly
        v, f = b.vertices, b.faces        # closed, outward-wound surface: the divergence theorem gives the same volume
        assert abs(np.einsum("ij,ij->i", v[f[:, 0]], np.cross(v[f[:, 1]], v[f[:, 2]])).sum() / 6 - b.volume) <= 1e-12
    assert np.array_equal(a.sample_points_uniformly(300).points, b.sample_points_uniformly(300, seed=3))
    assert S.alpha_shape(p[:3], 10).volume == 0.0 and S.alpha_shape(p, 1e6).faces.shape[0] == 0
    A, B = p[:300], p[300:800] + 0.01
    assert abs(oc.hausdorff_distance(A, B) - S.hausdorff_distance(A, B)) <= 1e-15
    # ICP: a scaled, shifted copy of a sphere template; both loops must land on the same similarity transform
    t = S.sphere_template(0.08, 800)
    tgt = oc.O3dPointCloud()
    tgt.points = 1.04 * t + np.array([0.01, 0.003, -0.002])
    src = oc.O3dPointCloud()
    src.points = t
    init = np.eye(4)
    init[:3, 3] = [0.005, 0.0, 0.0]

    class Est:
        with_scaling = True

    class Crit:
        max_iteration, relative_fitness, relative_rmse = 2000, 1e-6, 1e-6

    ro = oc.registration_icp(src, tgt, 0.01, init, Est(), Crit())
    rs = S.registration_icp(t, tgt.points, 0.01, init, with_scaling=True, max_iteration=2000)
    assert np.abs(ro.transformation - rs.transformation).max() <= 1e-9 and abs(ro.fitness - rs.fitness) <= 1e-12
    assert abs(rs.transformation[0, 0] - 1.04) <= 1e-6 and rs.fitness == 1.0
    # Ward: the same partition as scikit-learn's AgglomerativeClustering (numbering aside)
    blobs3 = np.vstack([t + c for c in ([0, 0, 0], [0.2, 0, 0], [0, 0.25, 0.05])])
    for k in (2, 3, 5):
        ours, theirs = S.ward_labels(blobs3, k), AgglomerativeClustering(n_clusters=k).fit_predict(blobs3)
        pairs = {(a, b) for a, b in zip(ours.tolist(), theirs.tolist())}
        assert len(pairs) == k == len(set(ours.tolist()))
