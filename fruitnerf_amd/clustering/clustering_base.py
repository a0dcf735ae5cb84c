"""Host mirror of the reference's `FruitClustering` (/root/reference/clustering/clustering_base.py) and of its
`Clustering` subclass (/root/reference/clustering/run_clustering.py:21-67): `remove_outliers` (:141-143),
`voxel_down_sample` (:138-139), `pcd2points_and_color` (:145-149), `cluster` (:183-207), the centre-distance pass
`merge_small_clusters` (:209-258), the template-matching `split_large_cluster` (:260-511) and `count` (:513-538).
The three library calls of the front half — Open3D `remove_radius_outlier`, Open3D `voxel_down_sample`,
`sklearn.cluster.DBSCAN` — run as HIP kernels on the MI355X (fruitnerf_amd/csrc/cloud.hip through the C ABI); there is no
CPU fallback for them.  The second stage works on a few thousand points per cluster and is host code in the reference
too (alpha shapes, ICP, Hausdorff distance, Ward splitting: clustering/shapes.py).

`PointCloud` stands in for `o3d.geometry.PointCloud` (open3d is not a dependency): float64 points and colours resident
on the GPU, with the two Open3D methods the reference calls, same argument names and return shapes."""
from __future__ import annotations

from pathlib import Path
from typing import List, Optional, Tuple, Union

import numpy as np
import torch

from .. import _kernels as K
from ..export import ply
from . import shapes


class PointCloud:
    """points / colors: float64 [n,3] device tensors (colours in [0,1], optional)."""

    def __init__(self, points, colors=None, device: Union[str, torch.device] = "cuda"):
        dev = torch.device(device)
        self.points = torch.as_tensor(points, dtype=torch.float64, device=dev).reshape(-1, 3).contiguous()
        self.colors = (None if colors is None else
                       torch.as_tensor(colors, dtype=torch.float64, device=dev).reshape(-1, 3).contiguous())
        if self.colors is not None and self.colors.shape != self.points.shape:
            raise ValueError(f"{self.points.shape[0]} points but {self.colors.shape[0]} colours")

    @classmethod
    def read(cls, path: str, device: Union[str, torch.device] = "cuda") -> "PointCloud":
        """o3d.io.read_point_cloud for the binary PLY layout the exporter writes."""
        pts, cols = ply.read_point_cloud(path)
        return cls(pts, cols, device)

    def __len__(self) -> int:
        return self.points.shape[0]

    def select_by_index(self, indices: torch.Tensor) -> "PointCloud":
        out = PointCloud.__new__(PointCloud)
        out.points = self.points[indices].contiguous()
        out.colors = None if self.colors is None else self.colors[indices].contiguous()
        return out

    def remove_radius_outlier(self, nb_points: int, radius: float) -> Tuple["PointCloud", torch.Tensor]:
        """Open3D: keep the points with MORE than nb_points points (itself included) strictly inside `radius`;
        -> (filtered cloud, kept indices in ascending order)."""
        if nb_points < 1 or radius <= 0:
            raise ValueError("Illegal input parameters, number of points and radius must be positive")  # Open3D's check
        counts = K.cloud_radius_count(self.points, radius, inclusive=False)
        keep = torch.nonzero(counts > nb_points).flatten()
        return self.select_by_index(keep), keep

    def voxel_down_sample(self, voxel_size: float) -> "PointCloud":
        """Open3D: one point per occupied voxel, the mean of its points / colours."""
        if voxel_size <= 0:
            raise ValueError("voxel_size <= 0.")                                                        # Open3D's check
        out = PointCloud.__new__(PointCloud)
        out.points, out.colors = K.cloud_voxel_down_sample(self.points, self.colors, voxel_size)
        return out


def dbscan_labels(points: torch.Tensor, eps: float, min_samples: int) -> torch.Tensor:
    """sklearn.cluster.DBSCAN(eps=eps, min_samples=min_samples).fit(X).labels_ on the device -> int64 [n]."""
    labels, _ = K.cloud_dbscan(points, eps, min_samples)
    return labels.to(torch.int64)


class FruitClustering:
    def __init__(self, voxel_size_down_sample: float = 0.00005, remove_outliers_nb_points: int = 100,
                 remove_outliers_radius: float = 0.01, cluster_merge_distance: float = 0.04):
        self.voxel_size_down_sample = voxel_size_down_sample
        self.remove_outliers_nb_points = remove_outliers_nb_points
        self.remove_outliers_radius = remove_outliers_radius
        self.cluster_merge_distance = cluster_merge_distance
        self.pcd_path: Optional[str] = None
        self.pcd: Optional[PointCloud] = None
        self.pcd_down_sampled: Optional[PointCloud] = None
        self.pcd_downsampled_cleaned: Optional[PointCloud] = None

    def voxel_down_sample(self, pcd: PointCloud) -> PointCloud:
        return pcd.voxel_down_sample(voxel_size=self.voxel_size_down_sample)

    def remove_outliers(self, pcd: PointCloud):
        return pcd.remove_radius_outlier(nb_points=self.remove_outliers_nb_points, radius=self.remove_outliers_radius)

    def pcd2points_and_color(self, pcd: PointCloud):
        return pcd.points.cpu().numpy(), (None if pcd.colors is None else pcd.colors.cpu().numpy())

    def cluster(self, pcd: PointCloud, **kwargs):
        """-> X [m,3], C [m,3], labels [m] (numpy, like the reference); (-1, -1, -1) for an empty cleaned cloud.
        kwargs: eps, min_sampled (the reference's spelling)."""
        self.pcd = pcd
        self.pcd_down_sampled, _ = self.remove_outliers(pcd=self.pcd)
        self.pcd_downsampled_cleaned = self.voxel_down_sample(pcd=self.pcd_down_sampled)
        if len(self.pcd_downsampled_cleaned) == 0:
            return -1, -1, -1
        labels = dbscan_labels(self.pcd_downsampled_cleaned.points, kwargs["eps"], kwargs["min_sampled"])
        X, C = self.pcd2points_and_color(pcd=self.pcd_downsampled_cleaned)
        if self.pcd_path is not None:
            ply.write_point_cloud(str(Path(self.pcd_path).parents[0] / "semantic_cleaned_down_sampled.ply"), X,
                                  C if C is not None else np.zeros_like(X))
        return X, C, labels.cpu().numpy()

    def merge_small_clusters(self, X: np.ndarray, C, labels: np.ndarray) -> Tuple[List[np.ndarray], List[np.ndarray]]:
        """Greedy centre-distance fusion in label order: a cluster whose centroid lies within cluster_merge_distance
        of the nearest already accepted centre is appended to that cluster (whose centre becomes the midpoint of its
        current point mean and the newcomer's centroid); otherwise it opens a new one.  -> (list of point arrays,
        list of label arrays)."""
        self.cluster_center: List[np.ndarray] = []
        self.assigned_cluster: List[np.ndarray] = []
        self.counter = 0
        self.fuse_counter = 0
        for lab in np.unique(labels):
            if lab == -1:
                continue
            self.counter += 1
            pts = X[labels == lab]
            centre = pts.mean(axis=0)
            if self.cluster_center:
                dist = np.linalg.norm(np.vstack(self.cluster_center) - centre, axis=1)
                near = int(np.argmin(dist))
                if dist[near] < self.cluster_merge_distance:
                    self.cluster_center[near] = (self.assigned_cluster[near].mean(axis=0) + centre) / 2
                    self.assigned_cluster[near] = np.vstack([self.assigned_cluster[near], pts])
                    self.fuse_counter += 1
                    continue
            self.cluster_center.append(centre)
            self.assigned_cluster.append(pts)
        return (list(self.assigned_cluster),
                [np.full(len(c), i, dtype=int) for i, c in enumerate(self.assigned_cluster)])

    # ---- second stage (:260-511) ---------------------------------------------------------------------------------
    # alphashape's alpha of the volume test (:371) and of the surface the fits run on (:382); ICP's correspondence
    # distance and iteration cap (:265-268); the candidate fruit numbers per cluster (:411-424)
    alpha_volume: float = 10.0
    alpha_surface: float = 100.0
    icp_max_correspondence_distance: float = 0.01
    icp_max_iteration: int = 2000
    surface_samples: int = 1000
    max_fruits_per_cluster: int = 6

    def set_template(self, template_points: np.ndarray) -> None:
        """The fruit template (centred at the origin) and its alpha shape, whose volume is the unit of the size test
        (run_clustering.py:43-46)."""
        pts = np.asarray(template_points, dtype=np.float64)
        self.fruit_template = pts - pts.mean(axis=0)
        self.fruit_alpha_shape_ = shapes.alpha_shape(self.fruit_template, self.alpha_volume)

    def fit_fruits(self, surface: np.ndarray) -> Tuple[int, List[np.ndarray], List[float]]:
        """How many template fruits explain `surface` (points sampled from a cluster's alpha shape) best: one template
        registered by scaled point-to-point ICP (:262-279), or k = 2 .. max templates placed at the centres of a Ward
        split of the surface into k parts (:296-322); each hypothesis is scored by the Hausdorff distance between the
        surface and its template points, the smallest wins (first one on ties, :427).
        -> (fruits, their template point sets, all the distances)."""
        init = np.eye(4)
        init[:3, 3] = surface.mean(axis=0)
        reg = shapes.registration_icp(self.fruit_template, surface, self.icp_max_correspondence_distance, init,
                                      with_scaling=True, max_iteration=self.icp_max_iteration)
        hypotheses = [[shapes.transform_points(reg.transformation, self.fruit_template)]]
        for k in range(2, self.max_fruits_per_cluster + 1):
            parts = shapes.ward_labels(surface, k)
            hypotheses.append([self.fruit_template + surface[parts == part].mean(axis=0) for part in np.unique(parts)])
        dists = [shapes.hausdorff_distance(surface, np.vstack(h)) for h in hypotheses]
        best = int(np.argmin(dists))
        return best + 1, hypotheses[best], dists

    def split_large_cluster(self, X: List[np.ndarray], C, labels, seed: int = 0) -> int:
        """The second counting stage on merge_small_clusters' clusters: a cluster whose alpha-shape volume exceeds the
        template's by more than 1 / 0.9 holds several fruits — fit_fruits says how many; one below 0.3 template volumes is
        dropped; the rest count once.  -> count = first-stage clusters - fused + additional fruits - pruned (:486).
        seed: of the surface sampling (Open3D draws from a clock-seeded generator there; cluster i uses seed + i)."""
        if getattr(self, "fruit_template", None) is None:
            raise RuntimeError("split_large_cluster needs a fruit template: set_template(points) / Clustering(...)")
        unit = self.fruit_alpha_shape_.volume
        self.additional_count = self.prune_counter = 0
        self.valid_clusters: List[np.ndarray] = []
        self.cluster_decisions: List[dict] = []
        for i, cluster in enumerate(X):
            volume = shapes.alpha_shape(cluster, self.alpha_volume).volume
            record = {"points": int(cluster.shape[0]), "volume": volume, "fruits": 1}
            if unit < 0.9 * volume:
                surface = shapes.alpha_shape(cluster, self.alpha_surface).sample_points_uniformly(self.surface_samples,
                                                                                                  seed=seed + i)
                fruits, fits, dists = self.fit_fruits(surface)
                self.additional_count += fruits - 1                 # one fruit of the cluster is counted already
                self.valid_clusters.extend(fits)
                record.update(fruits=fruits, distances=dists)
            elif 0.3 * unit > abs(volume):
                self.prune_counter += 1
                record["fruits"] = 0
            else:
                self.valid_clusters.append(cluster)
            self.cluster_decisions.append(record)
        self.cluster_centers = [c.mean(axis=0) for c in self.valid_clusters]
        count = self.counter - self.fuse_counter + self.additional_count - self.prune_counter
        gt = getattr(self, "gt_cluster_center", None)
        if gt is not None:
            self._score_against(np.asarray(gt, dtype=np.float64), count)
        if getattr(self, "gt_count", None):
            self.detection_rate = count / self.gt_count
        return count

    def _score_against(self, gt_centers: np.ndarray, count: int, match_distance: float = 0.15) -> None:
        """Greedy matching of the estimated fruit centres to ground-truth centres (:463-509): each estimate takes the
        nearest still unmatched ground-truth centre if it is closer than match_distance."""
        left = gt_centers.copy()
        self.true_positive = self.false_positive = 0
        for centre in self.cluster_centers:
            if left.shape[0] == 0:
                self.false_positive += 1
                continue
            d = np.linalg.norm(left - centre, axis=1)
            j = int(np.argmin(d))
            if d[j] < match_distance:
                self.true_positive += 1
                left = np.delete(left, j, axis=0)
            else:
                self.false_positive += 1
        self.false_negative = int(left.shape[0])
        self.real_count = self.true_positive
        tp, fp, fn = self.true_positive, self.false_positive, self.false_negative
        self.precision = tp / (tp + fp) if tp + fp else 0.0
        self.recall = tp / (tp + fn) if tp + fn else 0.0
        self.F1 = 2 * self.precision * self.recall / (self.precision + self.recall) if self.precision + self.recall else 0.0

    def count(self, pcd: Union[str, PointCloud], eps: float = 0.01, seed: int = 0) -> int:
        """cluster -> merge_small_clusters -> split_large_cluster (:513-538); 0 for a missing file or an empty cloud."""
        import os
        if isinstance(pcd, str):
            if not os.path.exists(pcd):
                self.real_count = 0
                return 0
            self.pcd_path = pcd
            pcd = PointCloud.read(pcd)
        if len(pcd) == 0:
            self.real_count = 0
            return 0
        X, C, labels = self.cluster(pcd=pcd, eps=eps, min_sampled=self.min_samples)
        if isinstance(X, int) and X == -1:
            return 0
        X, labels = self.merge_small_clusters(X, C, labels)
        return self.split_large_cluster(X, C, labels, seed=seed)

    def first_stage_count(self, pcd: PointCloud, eps: float, min_samples: int) -> int:
        """Clusters after `cluster` + `merge_small_clusters`: the count the reference prints as "First clustering
        stage count after fused (tiny) clusters" (:250), before the template-matching split."""
        X, C, labels = self.cluster(pcd, eps=eps, min_sampled=min_samples)
        if isinstance(X, int):
            return 0
        self.merge_small_clusters(X, C, labels)
        return self.counter - self.fuse_counter


class Clustering(FruitClustering):
    """run_clustering.py:21-67: FruitClustering + the fruit template and the evaluation inputs.  template_path: a PLY of
    the template (Open3D-readable binary layout); None or a Git-LFS pointer file (the reference's *_template.ply in its
    repository snapshot) -> a sphere of `template_radius` (shapes.sphere_template), `template_fallback` True; any other
    unreadable file raises.  The template is scaled by
    apple_template_size about the origin, then centred (:41-43)."""

    def __init__(self, template_path: Union[str, Path, None] = None, voxel_size_down_sample: float = 0.00005,
                 remove_outliers_nb_points: int = 800, remove_outliers_radius: float = 0.02, min_samples: int = 60,
                 apple_template_size: float = 0.8, cluster_merge_distance: float = 0.04, gt_cluster=None,
                 gt_count: Optional[int] = None, template_radius: float = 0.1):
        super().__init__(voxel_size_down_sample=voxel_size_down_sample,
                         remove_outliers_nb_points=remove_outliers_nb_points,
                         remove_outliers_radius=remove_outliers_radius, cluster_merge_distance=cluster_merge_distance)
        self.template_path = template_path
        self.min_samples = min_samples
        # The sphere stands in ONLY for "no template given" and for a Git-LFS pointer file (what the reference's
        # *_template.ply are in its repository snapshot); a missing, mistyped or corrupt file raises like the reference's
        # o3d.io.read_point_cloud + asserts would: the template's volume drives the split and prune thresholds, so a silent
        # substitute would give plausible but wrong counts.
        template = None
        self.template_fallback = True
        if template_path is not None:
            with open(str(template_path), "rb") as f:            # FileNotFoundError for a mistyped path
                head = f.read(64)
            if not head.startswith(b"version https://git-lfs"):
                template, _ = ply.read_point_cloud(str(template_path))     # raises on a corrupt PLY
                self.template_fallback = False
        if template is None:
            template = shapes.sphere_template(template_radius)
        self.set_template(np.asarray(template, dtype=np.float64) * apple_template_size)
        # ground truth: an [n,3] array of fruit centres (the reference reads them from an OBJ / line set, :48-59)
        self.gt_cluster = gt_cluster
        self.gt_cluster_center = None if gt_cluster is None else np.asarray(gt_cluster, dtype=np.float64).reshape(-1, 3)
        self.gt_count = gt_count
