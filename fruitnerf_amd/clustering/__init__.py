"""The fruit counting stage (reference: /root/reference/clustering/clustering_base.py, run_clustering.py)."""
from .clustering_base import Clustering, FruitClustering, PointCloud  # noqa: F401
