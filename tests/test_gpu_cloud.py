"""GPU: the counting stage's point-cloud front-end (fnr_cloud_* through the C ABI) against oracle/cloud.py.
Integer results (neighbour counts, kept indices, DBSCAN labels) and the fp64 voxel means must be BIT-EXACT."""
import numpy as np
import pytest
import torch

from oracle import cloud as oc
from tests.test_cloud_oracle import blobs, border_case

pytestmark = pytest.mark.gpu


def to_dev(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)


@pytest.mark.parametrize("radius", [0.03, 0.05, 0.3])
def test_radius_counts_match_the_oracle(dev, radius):
    from fruitnerf_amd import _kernels as K
    X = blobs(11)
    for inclusive in (False, True):
        got = K.cloud_radius_count(to_dev(X, dev), radius, inclusive).cpu().numpy()
        assert np.array_equal(got, oc.radius_neighbor_counts(X, radius, inclusive))


def test_radius_counts_on_exact_ties_and_degenerate_clouds(dev):
    from fruitnerf_amd import _kernels as K
    g = np.stack(np.meshgrid(*[np.arange(4.0)] * 3, indexing="ij"), -1).reshape(-1, 3)
    for inclusive in (False, True):
        got = K.cloud_radius_count(to_dev(g, dev), 1.0, inclusive).cpu().numpy()
        assert np.array_equal(got, oc.radius_neighbor_counts(g, 1.0, inclusive))
    one = np.array([[0.3, -0.2, 0.9]])
    assert K.cloud_radius_count(to_dev(one, dev), 0.01, False).tolist() == [1]
    same = np.repeat(one, 100, 0)                       # all points identical: zero extent
    assert K.cloud_radius_count(to_dev(same, dev), 0.01, False).tolist() == [100] * 100
    assert K.cloud_radius_count(to_dev(np.zeros((0, 3)), dev), 0.01, False).numel() == 0
    plane = np.concatenate([g[:, :2], np.zeros((64, 1))], 1)   # flat in z
    assert np.array_equal(K.cloud_radius_count(to_dev(plane, dev), 1.5, True).cpu().numpy(),
                          oc.radius_neighbor_counts(plane, 1.5, True))


def test_remove_radius_outlier_keeps_the_same_points(dev):
    from fruitnerf_amd.clustering import PointCloud
    X = blobs(3, noise=600)
    pcd = PointCloud(X, np.abs(np.sin(X)), dev)
    out, idx = pcd.remove_radius_outlier(nb_points=12, radius=0.04)
    keep = oc.remove_radius_outlier(X, 12, 0.04)
    assert 0 < keep.sum() < len(X)
    assert np.array_equal(idx.cpu().numpy(), np.flatnonzero(keep))
    assert np.array_equal(out.points.cpu().numpy(), X[keep])
    assert np.array_equal(out.colors.cpu().numpy(), np.abs(np.sin(X))[keep])
    with pytest.raises(ValueError):
        pcd.remove_radius_outlier(nb_points=0, radius=0.04)


@pytest.mark.parametrize("voxel", [0.2, 0.05, 1e-3])
def test_voxel_down_sample_is_bit_exact(dev, voxel):
    from fruitnerf_amd.clustering import PointCloud
    X = blobs(4)
    C = np.random.default_rng(1).uniform(0, 1, X.shape)
    ref_x, ref_c = oc.voxel_down_sample(X, C, voxel)
    out = PointCloud(X, C, dev).voxel_down_sample(voxel)
    assert np.array_equal(out.points.cpu().numpy(), ref_x)
    assert np.array_equal(out.colors.cpu().numpy(), ref_c)
    no_col = PointCloud(X, None, dev).voxel_down_sample(voxel)
    assert no_col.colors is None and np.array_equal(no_col.points.cpu().numpy(), ref_x)


def test_voxel_down_sample_edge_cases(dev):
    from fruitnerf_amd.clustering import PointCloud
    empty = PointCloud(np.zeros((0, 3)), None, dev).voxel_down_sample(0.1)
    assert len(empty) == 0
    X = blobs(5)
    one = PointCloud(X, None, dev).voxel_down_sample(100.0)            # everything in one voxel: sequential mean
    ref, _ = oc.voxel_down_sample(X, None, 100.0)
    assert len(one) == 1 and np.array_equal(one.points.cpu().numpy(), ref)
    with pytest.raises(ValueError):
        PointCloud(X, None, dev).voxel_down_sample(0.0)
    with pytest.raises(RuntimeError, match="2\\^21"):
        PointCloud(X, None, dev).voxel_down_sample(1e-8)


@pytest.mark.parametrize("seed,eps,min_samples", [(0, 0.03, 10), (1, 0.05, 30), (2, 0.02, 5), (3, 0.08, 4),
                                                  (4, 0.5, 3), (5, 0.01, 1)])
def test_dbscan_labels_are_sklearn_labels(dev, seed, eps, min_samples):
    from fruitnerf_amd import _kernels as K
    X = blobs(seed)
    labels, n_clusters = K.cloud_dbscan(to_dev(X, dev), eps, min_samples)
    ref = oc.dbscan(X, eps, min_samples)
    assert np.array_equal(labels.cpu().numpy().astype(np.int64), ref)
    assert int(n_clusters.item()) == ref.max() + 1


def test_dbscan_border_rule_and_all_noise(dev):
    from fruitnerf_amd.clustering.clustering_base import dbscan_labels
    a, b, mid = border_case()
    for order in ([a, b, mid], [b, mid, a], [mid, b, a], [a[:20], mid, b, a[20:]]):
        X = np.concatenate(order)
        got = dbscan_labels(to_dev(X, dev), 0.1, 44).cpu().numpy()
        assert np.array_equal(got, oc.dbscan(X, 0.1, 44))
    X = blobs(6)
    assert (dbscan_labels(to_dev(X, dev), 1e-4, 5) == -1).all()
    assert dbscan_labels(to_dev(np.zeros((0, 3)), dev), 0.1, 5).numel() == 0


def fruit_cloud(seed, n_fruits=14, per=500, clutter=1500):
    """An exported semantic cloud in miniature: lattice samples (spacing h) inside spheres + lattice clutter."""
    rng = np.random.default_rng(seed)
    h = 0.004
    cent = rng.uniform(-0.4, 0.4, (n_fruits, 3))
    parts = []
    for c in cent:
        g = np.stack(np.meshgrid(*[np.arange(-6, 7)] * 3, indexing="ij"), -1).reshape(-1, 3) * h
        g = g[(g * g).sum(1) <= (0.022 + 0.004 * rng.random()) ** 2]
        parts.append(np.round(c / h) * h + g)
    parts.append(np.round(rng.uniform(-0.5, 0.5, (clutter, 3)) / h) * h)
    X = np.concatenate(parts)
    rng.shuffle(X)
    return X, np.clip(np.abs(X) * 2, 0, 1)


def test_cluster_front_end_matches_the_oracle_pipeline(dev):
    """FruitClustering.cluster = remove_outliers -> voxel_down_sample -> DBSCAN on lattice-structured input."""
    from fruitnerf_amd.clustering import FruitClustering, PointCloud
    X, C = fruit_cloud(0)
    params = dict(nb_points=20, radius=0.012, voxel_size=0.003, eps=0.012, min_samples=15)
    ref_X, ref_C, ref_labels = oc.cluster_front_end(X, C, **params)
    fc = FruitClustering(voxel_size_down_sample=params["voxel_size"], remove_outliers_nb_points=params["nb_points"],
                         remove_outliers_radius=params["radius"], cluster_merge_distance=0.04)
    got_X, got_C, got_labels = fc.cluster(PointCloud(X, C, dev), eps=params["eps"], min_sampled=params["min_samples"])
    assert ref_labels.max() + 1 >= 10
    assert np.array_equal(got_X, ref_X) and np.array_equal(got_C, ref_C)
    assert np.array_equal(got_labels, ref_labels)
    assert 10 <= fc.first_stage_count(PointCloud(X, C, dev), params["eps"], params["min_samples"]) <= 14
    empty = fc.cluster(PointCloud(X[:5], C[:5], dev), eps=0.01, min_sampled=5)
    assert empty == (-1, -1, -1)


def canonical(labels: np.ndarray, index: np.ndarray) -> np.ndarray:
    """Relabel clusters by the smallest `index` value among their members (-1 stays)."""
    out = np.full(len(labels), -1, dtype=np.int64)
    for lab in np.unique(labels[labels >= 0]):
        m = labels == lab
        out[m] = index[m].min()
    return out


def test_large_cloud_invariants(dev):
    """400k points (no oracle at this size): pair symmetry of the counts, permutation invariance of the DBSCAN
    partition on core points, conservation of the point sum under voxel averaging."""
    from fruitnerf_amd import _kernels as K
    rng = np.random.default_rng(9)
    cent = rng.uniform(-1, 1, (60, 3))
    X = np.concatenate([c + 0.02 * rng.standard_normal((6000, 3)) for c in cent] + [rng.uniform(-1, 1, (40000, 3))])
    n = len(X)
    x = to_dev(X, dev)
    eps, ms = 0.01, 20
    counts = K.cloud_radius_count(x, eps, True)
    assert int(counts.min()) >= 1 and (int(counts.sum()) - n) % 2 == 0
    labels, k = K.cloud_dbscan(x, eps, ms)
    labels = labels.cpu().numpy()
    assert 1 <= int(k.item()) == labels.max() + 1
    core = (counts >= ms).cpu().numpy()
    assert (labels[core] >= 0).all()
    perm = rng.permutation(n)
    labels_p, k_p = K.cloud_dbscan(to_dev(X[perm], dev), eps, ms)
    assert int(k_p.item()) == int(k.item())
    back = np.empty(n, dtype=np.int64)
    back[perm] = labels_p.cpu().numpy()
    idx = np.arange(n)
    assert np.array_equal(canonical(labels, idx)[core], canonical(back, idx)[core])
    assert np.array_equal(labels == -1, back == -1)
    # first label of each cluster appears in increasing order of first core index (scikit-learn numbering)
    first_core = [np.flatnonzero(core & (labels == c))[0] for c in range(labels.max() + 1)]
    assert first_core == sorted(first_core)
    vx, _ = K.cloud_voxel_down_sample(x, None, 0.004)
    keys = torch.floor((x - (x.amin(0) - 0.002)) / 0.004).to(torch.int64)
    assert vx.shape[0] == torch.unique(keys, dim=0).shape[0]
    assert float(vx.amin()) >= float(x.amin()) and float(vx.amax()) <= float(x.amax())


def test_end_to_end_count_matches_the_cpu_pipeline(dev):
    """Clustering.count (clustering_base.py:513-538: cluster -> merge_small_clusters -> split_large_cluster) on a lattice
    cloud of fruits, two of them touching: the GPU front-end (radius-outlier removal, voxel down-sampling, DBSCAN) feeding
    the host second stage must give the count the all-CPU pipeline gives, with the same per-cluster decisions, and find
    every fruit.  CPU leg: SciPy's KD-tree for the radius counts (1.8 h is not a lattice distance, so `<` and `<=` agree),
    voxels of h / 4 hold one lattice point each (the down-sampled cloud is the cleaned cloud in ascending (z, y, x)
    order), scikit-learn's DBSCAN — the reference's own call — then the same second stage."""
    from scipy.spatial import cKDTree
    from fruitnerf_amd.clustering import Clustering, PointCloud
    rng = np.random.default_rng(3)
    h, r = 0.009, 0.08
    centres = np.array([[0.0, 0.0, 0.0], [0.5, 0.0, 0.0], [0.0, 0.5, 0.0], [0.5, 0.5, 0.1],            # singles
                        [-0.5, 0.0, 0.0], [-0.5 + 1.5 * r, 0.0, 0.0],                                  # a touching pair
                        [0.0, -0.5, 0.0]])
    k = int(np.ceil(r / h))
    gi = np.stack(np.meshgrid(*[np.arange(-k, k + 1)] * 3, indexing="ij"), -1).reshape(-1, 3)          # lattice INDICES
    ball = gi[((gi * h) ** 2).sum(1) <= r * r]
    parts = [np.round(c / h).astype(np.int64) + ball for c in centres]
    crumb = gi[((gi * h) ** 2).sum(1) <= (0.3 * r) ** 2] + np.round(np.array([0.3, -0.5, 0.0]) / h).astype(np.int64)   # pruned
    clutter = np.round(rng.uniform(-1, 1, (300, 3)) / h).astype(np.int64)
    # unique in index space: the touching balls share lattice nodes, and two float sums for one node may differ in an ulp
    X = np.unique(np.concatenate(parts + [crumb, clutter]), axis=0).astype(np.float64) * h
    rng.shuffle(X)
    kw = dict(voxel_size_down_sample=h / 4, remove_outliers_nb_points=2, remove_outliers_radius=1.8 * h, min_samples=4,
              apple_template_size=1.0, cluster_merge_distance=0.04, gt_cluster=centres, gt_count=len(centres),
              template_radius=r)
    cl = Clustering(template_path=None, **kw)
    cl.alpha_surface = 60.0                            # 1 / 60 > the lattice cell's circumradius (0.87 h)
    count = cl.count(PointCloud(X, None, dev), eps=1.8 * h, seed=2)
    ref = Clustering(template_path=None, **kw)
    ref.alpha_surface = 60.0
    keep = cKDTree(X).query_ball_point(X, 1.8 * h, return_length=True) > 2
    Xo = X[keep]
    Xo = Xo[np.lexsort((Xo[:, 0], Xo[:, 1], Xo[:, 2]))]
    labels = oc.dbscan(Xo, 1.8 * h, 4)
    assert np.array_equal(Xo, cl.pcd_downsampled_cleaned.points.cpu().numpy())
    Xs, ls = ref.merge_small_clusters(Xo, None, labels)
    want = ref.split_large_cluster(Xs, None, ls, seed=2)
    print(f"[count] gpu front-end {count} cpu front-end {want}; first stage {cl.counter - cl.fuse_counter}, additional "
          f"{cl.additional_count}, pruned {cl.prune_counter}; TP {cl.true_positive} FP {cl.false_positive} FN {cl.false_negative}")
    assert count == want
    assert [d["fruits"] for d in cl.cluster_decisions] == [d["fruits"] for d in ref.cluster_decisions]
    assert (cl.counter, cl.fuse_counter, cl.additional_count, cl.prune_counter) == \
        (ref.counter, ref.fuse_counter, ref.additional_count, ref.prune_counter)
    assert cl.prune_counter == 1                       # the crumb
    # every fruit is found; the touching pair is the one cluster the template hypotheses may over-split (the reference's
    # Hausdorff score prefers three or four small-error templates to two there)
    assert cl.true_positive == len(centres) and cl.false_negative == 0 and len(centres) <= count <= len(centres) + 2
