"""SemanticFieldHead — mirror of /root/reference/fruit_nerf/components/field_heads.py:29-40.

In the reference this is nerfstudio's FieldHead: an `nn.Linear(in_dim, num_classes)` stored as `.net`
plus an optional activation (None here).  It is a parameter holder: the 64->1 product is the last GEMM
of the semantic branch inside fnr_field_mlp_fwd.
"""
from __future__ import annotations

from typing import Optional

from torch import nn


class SemanticFieldHead(nn.Module):
    def __init__(self, num_classes: int, in_dim: Optional[int] = None, activation=None) -> None:
        super().__init__()
        assert in_dim is not None and activation is None
        self.in_dim = in_dim
        self.out_dim = num_classes
        self.net = nn.Linear(in_dim, num_classes)

    def forward(self, *_):
        raise RuntimeError("SemanticFieldHead is a parameter holder; it is fused into the HIP semantic branch")
