"""Counting-stage front-end on a synthetic exported cloud: lattice samples (pitch h) inside N spheres + clutter, with
the reference's synthetic-apple parameters (clustering/config_synthetic.py:2-15: radius 0.01 / nb_points 200,
voxel 0.001, eps 0.01 / min_samples 100).  GPU time per stage; optional CPU comparison (scipy KD-tree + sklearn).
  python tools/microbench/cloud_time.py [n_fruits=80] [cpu]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fruitnerf_amd import _kernels as K
from fruitnerf_amd.clustering import FruitClustering, PointCloud
from fruitnerf_amd.data.synthetic_cloud import make_export_cloud as make_cloud


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); out = fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return out, best * 1e3


if __name__ == "__main__":
    n_fruits = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    X = make_cloud(n_fruits)
    dev = torch.device("cuda:0")
    x = torch.as_tensor(X, device=dev)
    K.cloud_radius_count(x[:1000].contiguous(), 0.01, False)           # warm-up (module load)
    counts, t_cnt = timed(lambda: K.cloud_radius_count(x, 0.01, False))
    keep = torch.nonzero(counts > 200).flatten()
    xk = x[keep].contiguous()
    (vx, _), t_vox = timed(lambda: K.cloud_voxel_down_sample(xk, None, 0.001))
    (labels, k), t_db = timed(lambda: K.cloud_dbscan(vx, 0.01, 100))
    fc = FruitClustering(0.001, 200, 0.01, 0.04)
    cnt, t_all = timed(lambda: fc.first_stage_count(PointCloud(x, None, dev), 0.01, 100), reps=2)
    pairs = float(counts.double().sum())
    print(f"n={len(X)} fruits={n_fruits} mean neighbours {pairs / len(X):.0f} | radius_count {t_cnt:.2f} ms "
          f"({pairs / t_cnt / 1e6:.1f} G hits/s) | kept {len(keep)} | voxel {t_vox:.2f} ms -> {len(vx)} | dbscan {t_db:.2f} ms "
          f"-> {int(k)} clusters | FruitClustering first stage {t_all:.1f} ms -> count {cnt}", flush=True)
    if "cpu" in sys.argv:
        from scipy.spatial import cKDTree
        from sklearn.cluster import DBSCAN
        t = time.perf_counter(); tree = cKDTree(X); c = tree.query_ball_point(X, 0.01, return_length=True, workers=-1)
        t_c = time.perf_counter() - t
        Xv = vx.cpu().numpy()
        t = time.perf_counter(); lab = DBSCAN(eps=0.01, min_samples=100, n_jobs=-1).fit(Xv).labels_
        t_d = time.perf_counter() - t
        print(f"CPU: cKDTree counts {t_c:.2f} s (all cores) | sklearn DBSCAN {t_d:.2f} s | labels equal "
              f"{np.array_equal(lab, labels.cpu().numpy())} | counts equal(<=) n/a", flush=True)
