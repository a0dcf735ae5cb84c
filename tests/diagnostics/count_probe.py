"""The second counting stage on a dumped 512^3 semantic export of bench.py's trained model (FNR_BENCH_DUMP_CLOUD=...),
CPU only: scipy radius-outlier removal + scikit-learn DBSCAN stand in for the GPU front-end (labels equal, see
tests/test_gpu_cloud.py), then the product's merge_small_clusters + split_large_cluster.
    PYTHONPATH=. python tests/diagnostics/count_probe.py [dump.npz]"""
import numpy as np, time, sys
from sklearn.cluster import DBSCAN
from fruitnerf_amd.clustering import Clustering
from scipy.spatial import cKDTree
d=np.load(sys.argv[1] if len(sys.argv) > 1 else 'profiles/r04_raw/semantic_cloud_512.npz')
P=d['points'].astype(np.float64); C=d['centres']/2; R=d['radii']/2; pitch=float(d['pitch'])   # the r04 dump stored centres and radii x2 (bench.py's mistake of that build); pitch=float(d['pitch'])
t=cKDTree(P); cnt=t.query_ball_point(P, 1.8*pitch, return_length=True)-1
Q=P[cnt>=2]
lab=DBSCAN(eps=1.8*pitch,min_samples=4).fit_predict(Q)
print('front-end', len(P), len(Q), lab.max()+1)
for merge, tr, asurf in [(0.04, 1.0, 100.0), (0.04,1.0,60.0), (0.02,1.0,100.0)]:
    cl=Clustering(template_path=None, voxel_size_down_sample=pitch/4, remove_outliers_nb_points=2, remove_outliers_radius=1.8*pitch,
              min_samples=4, apple_template_size=tr, cluster_merge_distance=merge, gt_cluster=C, gt_count=32, template_radius=float(R.mean()))
    cl.alpha_surface=asurf
    t0=time.time()
    Xm,lm=cl.merge_small_clusters(Q, None, lab)
    try:
        n=cl.split_large_cluster(Xm,None,lm,seed=0)
    except Exception as e:
        print('ERR',merge,tr,asurf,repr(e)); continue
    print('merge',merge,'tmpl',tr,'alpha_s',asurf,'count',n,'first',cl.counter-cl.fuse_counter,'add',cl.additional_count,'pruned',cl.prune_counter,'TP',cl.true_positive,'FP',cl.false_positive,'FN',cl.false_negative,'F1',round(cl.F1,3),'%.1fs'%(time.time()-t0))
