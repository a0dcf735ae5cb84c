"""Binary PLY sink for the exported point sets — the file layout Open3D's `o3d.io.write_point_cloud` produces for a
PointCloud with points + colors (what /root/reference/fruit_nerf/scripts/exporter.py:116-119 calls on the three
sets of `sample_volume`), so the reference's clustering stage (clustering/*.py: `o3d.io.read_point_cloud`) can read
our exports unchanged.  open3d is not installed here: vertex = double x, y, z + uchar red, green, blue,
little-endian, colours quantised as round(clamp(c, 0, 1) * 255)."""
from __future__ import annotations

import os
from typing import Dict, Tuple

import numpy as np

_VERTEX = np.dtype([("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])


def write_point_cloud(path: str, points: np.ndarray, colors: np.ndarray) -> None:
    points = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    colors = np.asarray(colors, dtype=np.float64).reshape(-1, 3)
    if points.shape[0] != colors.shape[0]:
        raise ValueError(f"{points.shape[0]} points but {colors.shape[0]} colours")
    v = np.empty(points.shape[0], dtype=_VERTEX)
    v["x"], v["y"], v["z"] = points[:, 0], points[:, 1], points[:, 2]
    q = np.rint(np.clip(colors, 0.0, 1.0) * 255.0).astype(np.uint8)
    v["red"], v["green"], v["blue"] = q[:, 0], q[:, 1], q[:, 2]
    header = ("ply\nformat binary_little_endian 1.0\ncomment Created by Open3D\n"
              f"element vertex {points.shape[0]}\n"
              "property double x\nproperty double y\nproperty double z\n"
              "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n")
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(v.tobytes())


def read_point_cloud(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """Reader for files written by write_point_cloud (tests, count checks): points float64 [n,3], colours in [0,1]."""
    with open(path, "rb") as f:
        n = None
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: no end_header")
            line = line.decode("ascii").strip()
            if line.startswith("format") and "binary_little_endian" not in line:
                raise ValueError(f"{path}: only binary_little_endian PLY is supported")
            if line.startswith("element vertex"):
                n = int(line.split()[-1])
            if line == "end_header":
                break
        v = np.frombuffer(f.read(n * _VERTEX.itemsize), dtype=_VERTEX, count=n)
    pts = np.stack([v["x"], v["y"], v["z"]], axis=1)
    cols = np.stack([v["red"], v["green"], v["blue"]], axis=1).astype(np.float64) / 255.0
    return pts, cols


def write_point_clouds(pcd_list: Dict[str, dict]) -> Dict[str, int]:
    """The export script's final loop (scripts/exporter.py:116-119) over sample_volume()'s result: every set that has
    a path is written; returns the point count per set."""
    counts = {}
    for name, entry in pcd_list.items():
        counts[name] = int(entry["points"].shape[0])
        if entry.get("path"):
            write_point_cloud(entry["path"], entry["points"], entry["colors"])
    return counts
