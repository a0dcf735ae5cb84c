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


# ======================================================================================================================
# Second counting stage (clustering_base.py:260-511): the third-party calls it makes, restated.  `alphashape` (1.3.1),
# `hausdorff` (0.2.6) and `open3d` are absent here (not vendored, not installable): PARITY UNPINNED for these restatements
# of their published algorithms; scikit-learn's AgglomerativeClustering IS importable and is called as the reference does.
# The classes at the end are Open3D-shaped containers so that the reference's OWN split_large_cluster can be executed over
# these restatements (tests/golden/make_reference_split_golden.py) — its control flow, thresholds and count formula are
# then the reference's code, not a restatement.
# ======================================================================================================================


def circumradius_tetra(p0, p1, p2, p3) -> float:
    """Radius of the sphere through four points, via the Cayley-Menger style linear system alphashape solves
    (alphashape.circumcenter: barycentric coordinates from [[2 P P^T, 1], [1^T, 0]] x = [diag(P P^T), 1])."""
    P = np.array([p0, p1, p2, p3], dtype=np.float64)
    A = np.zeros((5, 5))
    A[:4, :4] = 2.0 * P @ P.T
    A[:4, 4] = 1.0
    A[4, :4] = 1.0
    b = np.r_[(P * P).sum(1), 1.0]
    try:
        bary = np.linalg.solve(A, b)[:4]
    except np.linalg.LinAlgError:
        return np.inf
    centre = bary @ P
    return float(np.linalg.norm(P[0] - centre))


class AlphaMesh:
    """What alphashape.alphashape returns for 3-D input (a trimesh.Trimesh) as far as the reference uses it: `.volume`
    and `.as_open3d.sample_points_uniformly(n)`."""

    def __init__(self, vertices, faces, volume, seed=0):
        self.vertices, self.faces, self.volume, self.seed = vertices, faces, volume, seed

    @property
    def as_open3d(self):
        return self

    def sample_points_uniformly(self, number_of_points):
        """Open3D TriangleMesh::SamplePointsUniformly: triangle drawn by area, point a (1 - sqrt(r1)) + b sqrt(r1)(1 - r2)
        + c sqrt(r1) r2.  Open3D's generator is clock seeded; here np.random.default_rng(seed) with the draws laid out as
        [n,3] = (triangle, r1, r2) per point."""
        rng = np.random.default_rng(self.seed)
        v, f = self.vertices, self.faces
        area = np.array([0.5 * np.linalg.norm(np.cross(v[t[1]] - v[t[0]], v[t[2]] - v[t[0]])) for t in f])
        cdf = np.cumsum(area / area.sum())
        out = np.empty((number_of_points, 3))
        draws = rng.random((number_of_points, 3))
        for k in range(number_of_points):
            t = f[min(int(np.searchsorted(cdf, draws[k, 0], side="left")), len(f) - 1)]
            s = np.sqrt(draws[k, 1])
            out[k] = (1 - s) * v[t[0]] + s * (1 - draws[k, 2]) * v[t[1]] + s * draws[k, 2] * v[t[2]]
        pc = O3dPointCloud()
        pc.points = out
        return pc


def alphashape_3d(points, alpha, seed=0) -> AlphaMesh:
    """alphashape.alphashape(points, alpha), 3-D branch: Delaunay triangulation, every tetrahedron whose circumradius is
    below 1 / alpha contributes its four faces, faces seen twice cancel (they are interior), the rest is the surface.
    Volume: trimesh's enclosed volume of that surface = the kept tetrahedra's volumes summed (closed surface)."""
    from scipy.spatial import Delaunay
    pts = np.asarray(points, dtype=np.float64)
    faces, volume = {}, 0.0
    for simplex in Delaunay(pts).simplices:
        p = pts[simplex]
        if circumradius_tetra(*p) < 1.0 / alpha:
            vol = np.linalg.det(np.c_[p[1] - p[0], p[2] - p[0], p[3] - p[0]]) / 6.0
            volume += abs(vol)
            order = simplex if vol > 0 else simplex[[1, 0, 2, 3]]
            a, b, c, d = (int(i) for i in order)
            for tri in ((a, c, b), (a, b, d), (b, c, d), (a, d, c)):     # outward for a positively oriented tetrahedron
                key = tuple(sorted(tri))
                if key in faces:
                    del faces[key]
                else:
                    faces[key] = tri
    # canonical face list (sampling walks it in order): smallest vertex index first within a face, faces sorted by their
    # sorted vertex triple
    ordered = []
    for key in sorted(faces):
        tri = faces[key]
        k = tri.index(min(tri))
        ordered.append(tri[k:] + tri[:k])
    return AlphaMesh(pts, np.array(ordered, dtype=int).reshape(-1, 3), volume, seed)


def hausdorff_distance(XA, XB, distance="euclidean") -> float:
    """hausdorff.hausdorff_distance: the symmetric Hausdorff distance, brute force."""
    assert distance == "euclidean"
    XA, XB = np.asarray(XA, dtype=np.float64), np.asarray(XB, dtype=np.float64)
    d = np.sqrt(squared_distances(XA, XB))
    return float(max(d.min(axis=1).max(), d.min(axis=0).max()))


def umeyama_similarity(src, dst, with_scaling=True):
    """Eigen::umeyama (what Open3D's TransformationEstimationPointToPoint calls): dst ~ c R src + t."""
    n = len(src)
    ms, md = src.mean(0), dst.mean(0)
    sigma = (dst - md).T @ (src - ms) / n
    U, d, Vt = np.linalg.svd(sigma)
    S = np.ones(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2] = -1
    R = U @ np.diag(S) @ Vt
    c = 1.0
    if with_scaling:
        c = float((d * S).sum() / ((src - ms) ** 2).sum() * n)
    T = np.eye(4)
    T[:3, :3] = c * R
    T[:3, 3] = md - c * R @ ms
    return T


def registration_icp(source, target, max_correspondence_distance, init, estimation=None, criteria=None):
    """open3d.pipelines.registration.registration_icp (RegistrationICP in Registration.cpp): brute-force nearest neighbours."""
    with_scaling = bool(getattr(estimation, "with_scaling", False))
    max_iteration = int(getattr(criteria, "max_iteration", 30))
    rel_fit, rel_rmse = float(getattr(criteria, "relative_fitness", 1e-6)), float(getattr(criteria, "relative_rmse", 1e-6))
    src = np.asarray(source.points, dtype=np.float64)
    tgt = np.asarray(target.points, dtype=np.float64)
    T = np.array(init, dtype=np.float64)

    def evaluate(moved):
        d2 = squared_distances(moved, tgt)
        j = d2.argmin(axis=1)
        dmin = d2[np.arange(len(moved)), j]
        ok = dmin < max_correspondence_distance ** 2
        n = int(ok.sum())
        return ok, j, (n / len(moved) if n else 0.0), (float(np.sqrt(dmin[ok].sum() / n)) if n else 0.0)

    moved = src @ T[:3, :3].T + T[:3, 3]
    ok, j, fitness, rmse = evaluate(moved)
    for _ in range(max_iteration):
        if ok.sum() < 3:
            break
        update = umeyama_similarity(moved[ok], tgt[j[ok]], with_scaling)
        T = update @ T
        moved = moved @ update[:3, :3].T + update[:3, 3]
        ok, j, new_fitness, new_rmse = evaluate(moved)
        stop = abs(fitness - new_fitness) < rel_fit and abs(rmse - new_rmse) < rel_rmse
        fitness, rmse = new_fitness, new_rmse
        if stop:
            break

    class Result:
        pass

    res = Result()
    res.transformation, res.fitness, res.inlier_rmse = T, fitness, rmse
    return res


class O3dPointCloud:
    """o3d.geometry.PointCloud as far as clustering_base.py touches it."""

    def __init__(self):
        self.points = np.zeros((0, 3))
        self.colors = np.zeros((0, 3))

    def get_center(self):
        return np.asarray(self.points, dtype=np.float64).mean(axis=0)

    def translate(self, t):
        self.points = np.asarray(self.points, dtype=np.float64) + np.asarray(t, dtype=np.float64)
        return self

    def transform(self, T):
        p = np.asarray(self.points, dtype=np.float64)
        self.points = p @ np.asarray(T)[:3, :3].T + np.asarray(T)[:3, 3]
        return self

    def scale(self, s, center=(0, 0, 0)):
        c = np.asarray(center, dtype=np.float64)
        self.points = (np.asarray(self.points, dtype=np.float64) - c) * s + c
        return self

    def paint_uniform_color(self, color):
        return self

    def __add__(self, other):
        out = O3dPointCloud()
        out.points = np.vstack([np.asarray(self.points).reshape(-1, 3), np.asarray(other.points).reshape(-1, 3)])
        return out

    def __deepcopy__(self, memo):
        out = O3dPointCloud()
        out.points = np.array(self.points, dtype=np.float64, copy=True)
        return out
