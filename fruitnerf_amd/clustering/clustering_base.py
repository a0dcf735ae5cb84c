"""Host mirror of the front half of the reference's `FruitClustering` (/root/reference/clustering/clustering_base.py):
`remove_outliers` (:141-143), `voxel_down_sample` (:138-139), `pcd2points_and_color` (:145-149), `cluster` (:183-207)
and the centre-distance pass `merge_small_clusters` (:209-258).  The three library calls the reference makes there
— Open3D `remove_radius_outlier`, Open3D `voxel_down_sample`, `sklearn.cluster.DBSCAN` — run as HIP kernels on the
MI355X (fruitnerf_amd/csrc/cloud.hip through the C ABI); there is no CPU fallback.  The template-matching split stage
behind it (ICP / alpha shapes / Hausdorff, :260-511) is the reference's and consumes the X, labels this returns.

`PointCloud` stands in for `o3d.geometry.PointCloud` (open3d is not a dependency): float64 points and colours resident
on the GPU, with the two Open3D methods the reference calls, same argument names and return shapes."""
from __future__ import annotations

from pathlib import Path
from typing import List, Optional, Tuple, Union

import numpy as np
import torch

from .. import _kernels as K
from ..export import ply


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

    def first_stage_count(self, pcd: PointCloud, eps: float, min_samples: int) -> int:
        """Clusters after `cluster` + `merge_small_clusters`: the count the reference prints as "First clustering
        stage count after fused (tiny) clusters" (:250), before the template-matching split."""
        X, C, labels = self.cluster(pcd, eps=eps, min_sampled=min_samples)
        if isinstance(X, int):
            return 0
        self.merge_small_clusters(X, C, labels)
        return self.counter - self.fuse_counter
