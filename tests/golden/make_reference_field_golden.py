#!/usr/bin/env python3
"""Generates tests/golden/reference_field.npz by running the reference's OWN `FruitField` class
(/root/reference/fruit_nerf/fruit_field.py, read at generation time only) — its constructor wiring, `get_density`,
`get_outputs`, `get_inference_outputs` and `forward` — on top of the oracle's restatement of the nerfstudio 0.3.2
components (oracle/ns_torch.py stands in for `nerfstudio.field_components.*`, which cannot be installed here).

What this pins: every line of field logic that lives in the reference repository itself (contraction vs. AABB
normalisation, the selector mask, density split + trunc_exp, detached semantic branch, appearance-embedding rules of the
three modes, concatenation order of the colour MLP input, the sub-module names of the checkpoint contract) for
oracle/fruit_oracle.py::FruitField, which restates it.  What it does NOT pin: the nerfstudio components themselves
(hash grid, MLP, SH, embedding) — those stay a restatement (oracle/ns_torch.py header).

    python tests/golden/make_reference_field_golden.py"""
import enum
import os
import sys

import numpy as np
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ns_torch as ns  # noqa: E402
from tests.golden import make_reference_golden as stubs  # noqa: E402  (the stub import machinery)


class FieldHeadNames(enum.Enum):        # nerfstudio.field_components.field_heads.FieldHeadNames (members used here)
    RGB = "rgb"
    DENSITY = "density"
    SEMANTICS = "semantics"


class FieldHead(nn.Module):
    """nerfstudio FieldHead: `self.net = nn.Linear(in_dim, out_dim)`, optional activation on the output."""

    def __init__(self, out_dim, field_head_name, in_dim=None, activation=None):
        super().__init__()
        self.out_dim, self.activation, self.field_head_name = out_dim, activation, field_head_name
        self.net = nn.Linear(in_dim, out_dim)

    def forward(self, in_tensor):
        out = self.net(in_tensor)
        return self.activation(out) if self.activation else out


def _drop_implementation(cls):
    class Adapter(cls):
        def __init__(self, *a, implementation=None, **k):
            super().__init__(*a, **k)
    Adapter.__name__ = cls.__name__
    return Adapter


class _Unused(nn.Module):               # NeRFEncoding: constructed by the reference (fruit_field.py:120-122), never called
    def __init__(self, *a, **k):
        super().__init__()


class _SceneBox:
    get_normalized_positions = staticmethod(ns.get_normalized_positions)


def install():
    stubs.install_stubs()
    import nerfstudio.cameras.rays as m
    m.RaySamples = ns.RaySamples
    import nerfstudio.data.scene_box as m
    m.SceneBox = _SceneBox
    import nerfstudio.field_components.activations as m
    m.trunc_exp = ns.trunc_exp
    import nerfstudio.field_components.encodings as m
    m.HashEncoding, m.SHEncoding, m.NeRFEncoding = (_drop_implementation(ns.HashEncoding),
                                                    _drop_implementation(ns.SHEncoding), _Unused)
    import nerfstudio.field_components.embedding as m
    m.Embedding = ns.Embedding
    import nerfstudio.field_components.field_heads as m
    m.FieldHeadNames, m.FieldHead = FieldHeadNames, FieldHead
    import nerfstudio.field_components.mlp as m
    m.MLP = _drop_implementation(ns.MLP)
    import nerfstudio.fields.base_field as m
    m.Field, m.shift_directions_for_tcnn = nn.Module, ns.shift_directions_for_tcnn


def ray_samples(seed, R, S, n_images, span):
    g = torch.Generator().manual_seed(seed)
    o = (torch.rand(R, 3, generator=g) * 2 - 1) * span
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    cam = torch.randint(0, n_images, (R, 1), generator=g)
    edges = torch.sort(torch.rand(R, S + 1, generator=g) * 2.5, dim=-1).values
    rb = ns.RayBundle(o, d, torch.full((R, 1), 1e-6), camera_indices=cam, nears=edges[:, :1], fars=edges[:, -1:])
    rs = rb.get_ray_samples(bin_starts=edges[:, :-1, None], bin_ends=edges[:, 1:, None])
    return dict(origins=o, directions=d, cam=cam, edges=edges), rs


def main():
    install()
    from fruit_nerf.fruit_field import FruitField          # the reference's class
    small = np.load(os.path.join(ROOT, "tests", "golden", "fruit_nerf_small.npz"))
    sd = {k[len("sd::field."):]: torch.from_numpy(small[k]) for k in small.files if k.startswith("sd::field.")}
    n_images = sd["embedding_appearance.embedding.weight"].shape[0]
    aabb = sd["aabb"]
    kw = dict(num_levels=16, max_res=2048, num_layers_semantic=2, hidden_dim_semantics=64, log2_hashmap_size=10,
              num_images=n_images, geo_feat_dim=15, use_average_appearance_embedding=True, use_semantics=True,
              num_semantic_classes=1, pass_semantic_gradients=False, implementation="torch")
    out = {"n_images": np.int64(n_images)}
    cases = [("train", None, ns.SceneContraction(order=float("inf")), True),
             ("eval", None, ns.SceneContraction(order=float("inf")), False),
             ("inference", "inference", ns.SceneContraction(order=float("inf")), False),
             ("export", "export", None, False)]
    for name, test_mode, distortion, training in cases:
        field = FruitField(aabb, test_mode=test_mode, spatial_distortion=distortion, **kw)
        field.load_state_dict(sd, strict=True)              # the checkpoint contract: same keys, same shapes
        field.train(training)
        inp, rs = ray_samples(seed=40 + len(out), R=24, S=7, n_images=n_images, span=0.9 if name == "export" else 1.6)
        for k, v in inp.items():
            out[f"{name}::in::{k}"] = v.numpy()
        res = field(rs)
        for head, v in res.items():
            out[f"{name}::out::{head.value}"] = v.detach().numpy()
        out[f"{name}::out::sample_locations"] = field._sample_locations.detach().numpy()
        out[f"{name}::out::density_before_activation"] = field._density_before_activation.detach().numpy()
        if training:                                        # gradients of a fixed scalar through every branch
            g = torch.Generator().manual_seed(9)
            loss = sum((v * torch.rand(v.shape, generator=g)).sum() for v in res.values())
            loss.backward()
            out[f"{name}::loss"] = np.float64(loss.item())
            for pname, p in field.named_parameters():
                if p.grad is None:
                    continue
                if "hash_table" in pname:
                    out[f"{name}::gradsum::{pname}"] = np.float64(p.grad.double().abs().sum().item())
                else:
                    out[f"{name}::grad::{pname}"] = p.grad.numpy()
            out[f"{name}::grad::sample_locations"] = field._sample_locations.grad.numpy()
    out["state_dict_keys"] = np.array(sorted(field.state_dict().keys()))
    path = os.path.join(ROOT, "tests", "golden", "reference_field.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
