"""Front-end of the fruit counting stage (reference: /root/reference/clustering/clustering_base.py:118-258)."""
from .clustering_base import FruitClustering, PointCloud  # noqa: F401
