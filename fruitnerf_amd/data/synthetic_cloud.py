"""Synthetic stand-in for an exported semantic point cloud (there is no network for the reference's PLY files): lattice
samples of pitch h inside `n_fruits` spheres plus uniform lattice clutter, shuffled.  With the defaults a point has ~300
neighbours inside the reference's synthetic-apple search radius 0.01 (clustering/config_synthetic.py:2-15), like a
1024^3 export of a fruit tree."""
import numpy as np


def make_export_cloud(n_fruits: int = 80, seed: int = 0, h: float = 0.0024, r: float = 0.035,
                      clutter: float = 0.05) -> np.ndarray:
    rng = np.random.default_rng(seed)
    cent = rng.uniform(-0.8, 0.8, (n_fruits, 3))
    k = int(np.ceil(r / h))
    g = np.stack(np.meshgrid(*[np.arange(-k, k + 1)] * 3, indexing="ij"), -1).reshape(-1, 3) * h
    ball = g[(g * g).sum(1) <= r * r]
    parts = [np.round(c / h) * h + ball for c in cent]
    n = sum(len(p) for p in parts)
    parts.append(np.round(rng.uniform(-1, 1, (int(n * clutter), 3)) / h) * h)
    X = np.concatenate(parts)
    rng.shuffle(X)
    return X
