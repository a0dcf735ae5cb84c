"""TEST INFRASTRUCTURE (oracle) — point-cloud front-end of the fruit counting stage, CPU / numpy float64.

The reference's `FruitClustering.cluster` (/root/reference/clustering/clustering_base.py:183-207) runs, on the exported
semantic point cloud:  Open3D `remove_radius_outlier` (:141-143)  ->  Open3D `voxel_down_sample` (:138-139)  ->
`sklearn.cluster.DBSCAN(eps, min_samples).fit(X).labels_` (:199-200).

* DBSCAN: scikit-learn IS importable here, so `dbscan()` below is the reference's own call (parity PINNED on the real
  dependency); `dbscan_restated()` is the numbering rule written out (tests check it against scikit-learn).
* Open3D (`open3d`, unpinned in /root/reference/pyproject.toml:6) is NOT importable: `remove_radius_outlier` and
  `voxel_down_sample` restate its published algorithm (cpp/open3d/geometry/PointCloud.cpp: RemoveRadiusOutliers,
  VoxelDownSample; nanoflann's RadiusResultSet keeps dist^2 < r^2) — PARITY UNPINNED for these two.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module."""
import numpy as np


def squared_distances(p: np.ndarray, q: np.ndarray) -> np.ndarray:
    """[n,3] x [m,3] -> [n,m]; summed in coordinate order like nanoflann's L2 adaptor and scikit-learn's
    EuclideanDistance.rdist: ((dx*dx) + dy*dy) + dz*dz, every op rounded (no FMA)."""
    d = p[:, None, :] - q[None, :, :]
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]


def radius_neighbor_counts(xyz: np.ndarray, radius: float, inclusive: bool, chunk: int = 512) -> np.ndarray:
    """counts[i] = #{j : |p_i - p_j|^2 < r^2}  (<= when inclusive), the point itself included.  Brute force."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    r2 = np.float64(radius) * np.float64(radius)
    out = np.zeros(len(xyz), dtype=np.int32)
    for a in range(0, len(xyz), chunk):
        d2 = squared_distances(xyz[a:a + chunk], xyz)
        out[a:a + chunk] = ((d2 <= r2) if inclusive else (d2 < r2)).sum(1)
    return out


def remove_radius_outlier(xyz: np.ndarray, nb_points: int, radius: float) -> np.ndarray:
    """Open3D PointCloud::RemoveRadiusOutliers -> keep mask [n].  A point stays when the radius search around it
    (strict dist^2 < r^2, the point itself included) returns MORE than nb_points indices; survivors keep their order."""
    return radius_neighbor_counts(xyz, radius, inclusive=False) > nb_points


def voxel_down_sample(xyz: np.ndarray, rgb, voxel_size: float):
    """Open3D PointCloud::VoxelDownSample: voxel index = floor((p - (min_bound - voxel_size/2)) / voxel_size) per axis,
    one output point per occupied voxel = the mean (sum in point order, / count) of its points and colours.
    Open3D emits voxels in unordered_map order (unspecified); here and in the HIP path they are emitted in ascending
    (iz, iy, ix) order."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    if len(xyz) == 0:
        return xyz.reshape(0, 3), (None if rgb is None else np.zeros((0, 3)))
    vmin = xyz.min(0) - np.float64(voxel_size) * 0.5
    vmax = xyz.max(0) + np.float64(voxel_size) * 0.5
    idx = np.floor((xyz - vmin) / np.float64(voxel_size)).astype(np.int64)
    dims = np.floor((vmax - vmin) / np.float64(voxel_size)).astype(np.int64) + 1
    key = (idx[:, 2] * dims[1] + idx[:, 1]) * dims[0] + idx[:, 0]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    heads = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
    ends = np.r_[heads[1:], len(ks)]
    out_xyz = np.empty((len(heads), 3))
    out_rgb = None if rgb is None else np.empty((len(heads), 3))
    for v, (a, b) in enumerate(zip(heads, ends)):
        acc = np.zeros(3)
        for i in order[a:b]:            # sequential sum in point order, like AccumulatedPoint::AddPoint
            acc = acc + xyz[i]
        out_xyz[v] = acc / np.float64(b - a)
        if rgb is not None:
            acc = np.zeros(3)
            for i in order[a:b]:
                acc = acc + rgb[i]
            out_rgb[v] = acc / np.float64(b - a)
    return out_xyz, out_rgb


def dbscan(X: np.ndarray, eps: float, min_samples: int) -> np.ndarray:
    """The reference's call (clustering_base.py:199-200) -> labels [n] int64 (-1 = noise)."""
    from sklearn.cluster import DBSCAN
    if len(X) == 0:
        return np.zeros(0, dtype=np.int64)
    return DBSCAN(eps=eps, min_samples=min_samples).fit(np.asarray(X, dtype=np.float64)).labels_


def dbscan_restated(X: np.ndarray, eps: float, min_samples: int) -> np.ndarray:
    """What scikit-learn's labels are, as a rule the parallel kernel can follow:
    core_i = (#{j: d2_ij <= eps^2}, itself included) >= min_samples; clusters = connected components of the core
    points under d2 <= eps^2, numbered by their smallest member index; a non-core point within eps of core points takes
    the SMALLEST such cluster number (the first cluster to reach it in dbscan_inner's index-ordered expansion); the
    rest is noise (-1)."""
    X = np.asarray(X, dtype=np.float64)
    n = len(X)
    r2 = np.float64(eps) * np.float64(eps)
    adj = squared_distances(X, X) <= r2
    core = adj.sum(1) >= min_samples
    parent = np.arange(n)

    def find(a):
        while parent[a] != a:
            a = parent[a]
        return a
    for i in np.flatnonzero(core):
        for j in np.flatnonzero(adj[i] & core):
            a, b = find(i), find(j)
            if a != b:
                parent[max(a, b)] = min(a, b)
    root = np.array([find(i) for i in range(n)])
    ids = -np.ones(n, dtype=np.int64)
    roots = np.flatnonzero(core & (root == np.arange(n)))
    ids[roots] = np.arange(len(roots))
    labels = -np.ones(n, dtype=np.int64)
    labels[core] = ids[root[core]]
    for i in np.flatnonzero(~core):
        nb = np.flatnonzero(adj[i] & core)
        if len(nb):
            labels[i] = ids[root[nb]].min()
    return labels


def cluster_front_end(xyz, rgb, nb_points, radius, voxel_size, eps, min_samples):
    """FruitClustering.cluster (clustering_base.py:183-207) -> X, C, labels."""
    keep = remove_radius_outlier(xyz, nb_points, radius)
    X, Cc = voxel_down_sample(xyz[keep], None if rgb is None else rgb[keep], voxel_size)
    if len(X) == 0:
        return -1, -1, -1
    return X, Cc, dbscan(X, eps, min_samples)
