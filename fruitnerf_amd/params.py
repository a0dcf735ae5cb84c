"""Parameter holders with nerfstudio-0.3.2 (torch layout) names, and the flat parameter arena.

Checkpoint contract (FruitPipeline.load_pipeline uses load_state_dict(strict=True),
/root/reference/fruit_nerf/fruit_pipeline.py:229-240): module / parameter names below reproduce the
state-dict keys of the reference's torch layout, e.g. `field.mlp_base_grid.hash_table`,
`field.mlp_base.0.hash_table` (alias), `field.mlp_semantics.layers.0.weight`,
`field.field_head_semantics.net.weight`, `proposal_networks.0.mlp_base.1.layers.1.bias`.

The modules hold parameters only; all arithmetic runs in the HIP library.  `ParamArena` re-homes every
parameter of a module tree into ONE contiguous fp32 buffer (plus a same-shaped gradient buffer) so that
  * the kernels get stable raw pointers,
  * the data-parallel exchange is a single RCCL all-reduce over one buffer (fruit_pipeline.py:116-118),
  * the fused Adam step is one launch over the arena.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
from torch import nn

from ._kernels import hash_scalings


class HashEncoding(nn.Module):
    """Parameter holder for nerfstudio HashEncoding (fruit_field.py:124-131)."""

    def __init__(self, num_levels=16, min_res=16, max_res=1024, log2_hashmap_size=19, features_per_level=2,
                 hash_init_scale=0.001):
        super().__init__()
        assert features_per_level == 2, "gfx950 kernels are built for features_per_level=2"
        self.num_levels = num_levels
        self.min_res = min_res
        self.max_res = max_res
        self.log2_hashmap_size = log2_hashmap_size
        self.features_per_level = features_per_level
        self.hash_table_size = 2 ** log2_hashmap_size
        self.scalings = hash_scalings(num_levels, min_res, max_res)
        table = torch.rand(size=(self.hash_table_size * num_levels, features_per_level)) * 2 - 1
        table *= hash_init_scale
        self.hash_table = nn.Parameter(table)

    def get_out_dim(self) -> int:
        return self.num_levels * self.features_per_level

    def forward(self, *_):
        raise RuntimeError("HashEncoding is a parameter holder; evaluation is fused into the HIP field kernels")


class MLP(nn.Module):
    """Parameter holder for nerfstudio MLP (torch path): `num_layers` biased nn.Linear layers."""

    def __init__(self, in_dim, num_layers, layer_width, out_dim=None):
        super().__init__()
        self.in_dim = in_dim
        self.out_dim = out_dim if out_dim is not None else layer_width
        layers = []
        if num_layers == 1:
            layers.append(nn.Linear(in_dim, self.out_dim))
        else:
            for i in range(num_layers - 1):
                layers.append(nn.Linear(in_dim if i == 0 else layer_width, layer_width))
            layers.append(nn.Linear(layer_width, self.out_dim))
        self.layers = nn.ModuleList(layers)

    def get_out_dim(self) -> int:
        return self.out_dim

    def forward(self, *_):
        raise RuntimeError("MLP is a parameter holder; evaluation is fused into the HIP field kernels")


class Embedding(nn.Module):
    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.embedding = nn.Embedding(in_dim, out_dim)

    def mean(self, dim=0):
        return self.embedding.weight.mean(dim)


class ParamArena:
    """Flat fp32 parameter + gradient storage for a list of (group_name, parameters)."""

    def __init__(self, groups: List[Tuple[str, List[nn.Parameter]]], device):
        self.device = torch.device(device)
        seen = set()
        self.entries: List[Tuple[str, nn.Parameter, int, int]] = []
        self.group_ranges: Dict[str, Tuple[int, int]] = {}
        off = 0
        for gname, params in groups:
            start = off
            for p in params:
                if id(p) in seen:
                    continue
                seen.add(id(p))
                n = p.numel()
                self.entries.append((gname, p, off, n))
                off += (n + 3) // 4 * 4  # keep every tensor 16-byte aligned
            self.group_ranges[gname] = (start, off)
        self.numel = off
        self.params = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(off, dtype=torch.float32, device=self.device)
        for _, p, o, n in self.entries:
            view = self.params[o:o + n].view(p.shape)
            view.copy_(p.data.to(self.device, torch.float32))
            p.data = view
            p.grad = self.grads[o:o + n].view(p.shape)

    def group_slice(self, name: str) -> slice:
        a, b = self.group_ranges[name]
        return slice(a, b)

    def zero_grad(self) -> None:
        self.grads.zero_()

    def reattach_grads(self) -> None:
        """Optimisers may set .grad = None (zero_grad(set_to_none=True)); re-point them at the arena."""
        for _, p, o, n in self.entries:
            if p.grad is None or p.grad.data_ptr() != self.grads.data_ptr() + 4 * o:
                p.grad = self.grads[o:o + n].view(p.shape)
