"""Geometry of the counting stage's second half (/root/reference/clustering/clustering_base.py:260-511): what the
reference obtains from `alphashape`, Open3D's ICP / surface sampling, the `hausdorff` package and scikit-learn's
AgglomerativeClustering, as host code over NumPy + SciPy (the reference runs this stage offline on the CPU; its inputs
— the cleaned, clustered cloud — come from the GPU front-end in clustering_base.py).  Nothing here imports oracle/.

Library semantics reproduced (each pinned against an independent restatement, tests/test_cloud_oracle.py, and — through
the reference's own `split_large_cluster` executed over those restatements — tests/test_reference_pins.py):
  * `alphashape.alphashape(points, alpha)` in 3-D (alphashape 1.3.1): Delaunay tetrahedra with circumradius < 1 / alpha;
    the shape's boundary = triangles that belong to exactly one kept tetrahedron; `.volume` = the enclosed volume
    (trimesh's signed-volume sum over the outward-oriented boundary = the sum of the kept tetrahedra's volumes);
  * `TriangleMesh.sample_points_uniformly(n)` (Open3D): triangle by area, then (1 - sqrt(r1), sqrt(r1)(1 - r2),
    sqrt(r1) r2) barycentric weights; Open3D seeds its generator from the clock — here a NumPy Generator with a seed;
  * `registration_icp(source, target, max_correspondence_distance, init, TransformationEstimationPointToPoint(
    with_scaling=True), ICPConvergenceCriteria(max_iteration))`: nearest target point of every transformed source point
    within the distance, Umeyama similarity estimate on those pairs, until fitness and inlier RMSE both move by less
    than 1e-6 (Open3D's default relative criteria) or max_iteration;
  * `hausdorff_distance(A, B, distance="euclidean")`: max(max_a min_b |a - b|, max_b min_a |a - b|);
  * `AgglomerativeClustering(n_clusters=k)` (Ward linkage on the points): SciPy's Ward tree cut into k clusters — the same
    tree scikit-learn builds for unstructured data.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np


# ---- alpha shapes -----------------------------------------------------------------------------------------------


def tetra_circumradii(pts: np.ndarray, simplices: np.ndarray) -> np.ndarray:
    """Circumradius of every tetrahedron (pts [n,3], simplices [m,4]); inf for degenerate (flat) ones.
    Centre c relative to vertex 0 solves 2 (v_i - v_0) . c = |v_i - v_0|^2, i = 1..3."""
    a = pts[simplices[:, 0]]
    E = pts[simplices[:, 1:]] - a[:, None, :]                      # [m,3,3] edge vectors from vertex 0
    rhs = 0.5 * np.einsum("mij,mij->mi", E, E)
    det = np.linalg.det(E)
    ok = np.abs(det) > 1e-300
    centre = np.zeros_like(rhs)
    centre[ok] = np.linalg.solve(E[ok], rhs[ok][..., None])[..., 0]
    r = np.linalg.norm(centre, axis=1)
    r[~ok] = np.inf
    return r


@dataclass
class AlphaShape:
    """The alpha shape of a 3-D point set: the kept tetrahedra and their boundary surface."""
    vertices: np.ndarray      # [n,3] the input points
    tetrahedra: np.ndarray    # [t,4] kept Delaunay simplices
    faces: np.ndarray         # [f,3] boundary triangles, wound so that their normals point out of the shape
    volume: float

    def face_areas(self) -> np.ndarray:
        v = self.vertices
        return 0.5 * np.linalg.norm(np.cross(v[self.faces[:, 1]] - v[self.faces[:, 0]],
                                             v[self.faces[:, 2]] - v[self.faces[:, 0]]), axis=1)

    def sample_points_uniformly(self, number_of_points: int, seed: int = 0) -> np.ndarray:
        """[number_of_points, 3] points on the boundary surface, uniform by area."""
        if self.faces.shape[0] == 0:
            raise ValueError("the alpha shape has no surface: alpha is too large for this cloud's point spacing")
        rng = np.random.default_rng(seed)
        area = self.face_areas()
        cdf = np.cumsum(area / area.sum())
        r = rng.random((number_of_points, 3))
        tri = np.minimum(np.searchsorted(cdf, r[:, 0], side="left"), len(cdf) - 1)
        s = np.sqrt(r[:, 1])
        wa, wb, wc = 1.0 - s, s * (1.0 - r[:, 2]), s * r[:, 2]
        v, f = self.vertices, self.faces[tri]
        return wa[:, None] * v[f[:, 0]] + wb[:, None] * v[f[:, 1]] + wc[:, None] * v[f[:, 2]]


_FACE_OF_TETRA = np.array([[1, 2, 3], [0, 3, 2], [0, 1, 3], [0, 2, 1]])   # face i is opposite vertex i


def alpha_shape(points: np.ndarray, alpha: float) -> AlphaShape:
    """alphashape.alphashape(points, alpha) for a 3-D cloud."""
    from scipy.spatial import Delaunay
    pts = np.ascontiguousarray(points, dtype=np.float64)
    if pts.shape[0] < 4:
        return AlphaShape(pts, np.zeros((0, 4), int), np.zeros((0, 3), int), 0.0)
    try:
        simplices = Delaunay(pts).simplices
    except Exception:                                              # coplanar / degenerate input: no volume
        return AlphaShape(pts, np.zeros((0, 4), int), np.zeros((0, 3), int), 0.0)
    keep = simplices[tetra_circumradii(pts, simplices) < 1.0 / alpha]
    if keep.shape[0] == 0:
        return AlphaShape(pts, keep, np.zeros((0, 3), int), 0.0)
    a, b, c, d = (pts[keep[:, i]] for i in range(4))
    signed = np.einsum("ij,ij->i", np.cross(b - a, c - a), d - a) / 6.0
    # orient every tetrahedron positively, so that _FACE_OF_TETRA's windings point outwards
    flip = signed < 0
    keep = keep.copy()
    keep[flip, 0], keep[flip, 1] = keep[flip, 1].copy(), keep[flip, 0].copy()
    faces = keep[:, _FACE_OF_TETRA].reshape(-1, 3)                  # [4t,3]
    key = np.sort(faces, axis=1)
    _, inverse, counts = np.unique(key, axis=0, return_inverse=True, return_counts=True)
    boundary = faces[counts[inverse.reshape(-1)] == 1]
    # canonical order (the surface sampling consumes the faces in order): every triangle rotated so that its smallest
    # vertex index comes first (winding kept), triangles sorted by their sorted vertex triple
    first = np.argmin(boundary, axis=1)
    boundary = np.take_along_axis(boundary, (first[:, None] + np.arange(3)[None, :]) % 3, axis=1)
    boundary = boundary[np.lexsort(np.sort(boundary, axis=1).T[::-1])]
    return AlphaShape(pts, keep, boundary, float(np.abs(signed).sum()))


# ---- point-set distances, registration, splitting --------------------------------------------------------------------


def nearest(src: np.ndarray, tgt: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """For every row of src: (distance to, index of) its nearest row of tgt."""
    from scipy.spatial import cKDTree
    d, i = cKDTree(tgt).query(src, k=1)
    return d, i


def hausdorff_distance(A: np.ndarray, B: np.ndarray) -> float:
    return float(max(nearest(A, B)[0].max(), nearest(B, A)[0].max()))


def umeyama(src: np.ndarray, dst: np.ndarray, with_scaling: bool = True) -> np.ndarray:
    """Least-squares similarity transform dst ~ s R src + t (Umeyama 1991) as a 4 x 4 matrix."""
    mu_s, mu_d = src.mean(axis=0), dst.mean(axis=0)
    xs, xd = src - mu_s, dst - mu_d
    cov = xd.T @ xs / src.shape[0]
    U, D, Vt = np.linalg.svd(cov)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1.0
    R = U @ S @ Vt
    scale = 1.0
    if with_scaling:
        var = (xs ** 2).sum() / src.shape[0]
        scale = float(np.trace(np.diag(D) @ S) / var) if var > 0 else 1.0
    T = np.eye(4)
    T[:3, :3] = scale * R
    T[:3, 3] = mu_d - scale * R @ mu_s
    return T


def transform_points(T: np.ndarray, pts: np.ndarray) -> np.ndarray:
    return pts @ T[:3, :3].T + T[:3, 3]


@dataclass
class RegistrationResult:
    transformation: np.ndarray
    fitness: float
    inlier_rmse: float
    iterations: int


def registration_icp(source: np.ndarray, target: np.ndarray, max_correspondence_distance: float, init: np.ndarray,
                     with_scaling: bool = True, max_iteration: int = 30, relative_fitness: float = 1e-6,
                     relative_rmse: float = 1e-6) -> RegistrationResult:
    """Point-to-point ICP of `source` onto `target` (Open3D's loop: evaluate, estimate on the current pairs, re-evaluate;
    stop when fitness and RMSE both change by less than the relative criteria)."""
    from scipy.spatial import cKDTree
    tree = cKDTree(target)
    T = np.array(init, dtype=np.float64)

    def evaluate(T):
        moved = transform_points(T, source)
        d, i = tree.query(moved, k=1, distance_upper_bound=max_correspondence_distance)
        ok = np.isfinite(d)
        n = int(ok.sum())
        if n == 0:
            return moved, ok, i, 0.0, 0.0
        return moved, ok, i, n / source.shape[0], float(np.sqrt((d[ok] ** 2).sum() / n))

    moved, ok, idx, fitness, rmse = evaluate(T)
    it = 0
    for it in range(1, max_iteration + 1):
        if ok.sum() < 3:
            break
        update = umeyama(moved[ok], target[idx[ok]], with_scaling)
        T = update @ T
        moved, ok, idx, new_fitness, new_rmse = evaluate(T)
        done = abs(fitness - new_fitness) < relative_fitness and abs(rmse - new_rmse) < relative_rmse
        fitness, rmse = new_fitness, new_rmse
        if done:
            break
    return RegistrationResult(T, fitness, rmse, it)


def ward_labels(points: np.ndarray, n_clusters: int) -> np.ndarray:
    """AgglomerativeClustering(n_clusters).fit_predict(points) up to the numbering of the clusters."""
    from scipy.cluster import hierarchy
    n_clusters = min(n_clusters, points.shape[0])
    return hierarchy.fcluster(hierarchy.ward(points), n_clusters, criterion="maxclust") - 1


def sphere_template(radius: float, n_points: int = 2000) -> np.ndarray:
    """A fruit template when the reference's template meshes are not available (its *_template.ply files are Git-LFS
    pointers): n points on a sphere, Fibonacci lattice (deterministic, near-uniform), centred at the origin."""
    k = np.arange(n_points) + 0.5
    z = 1.0 - 2.0 * k / n_points
    phi = k * np.pi * (3.0 - np.sqrt(5.0))
    rho = np.sqrt(1.0 - z * z)
    return radius * np.stack([rho * np.cos(phi), rho * np.sin(phi), z], axis=1)
