#!/usr/bin/env python3
"""Generates tests/golden/reference_model.npz by running the reference's OWN `FruitModel`
(/root/reference/fruit_nerf/fruit_nerf.py, read at generation time only): `populate_modules`, the training callbacks
(weight anneal, proposal update schedule), `forward` -> `get_outputs` / `get_inference_outputs` / `get_export_outputs`
after `setup_inference`, `get_loss_dict`, `get_metrics_dict` — on top of the oracle's restatement of the nerfstudio
0.3.2 components (oracle/ns_torch.py stands in for `nerfstudio.*`; thin adapters below give those functions the class
shapes the reference constructs).

What this pins for oracle/fruit_oracle.py::FruitModel: which config fields reach the field, the proposal-network
wiring and update schedule, sampler -> field -> weights -> renderers, the detached semantic weights, the 0.9 label
threshold, the loss / metric dictionaries and the export outputs.  Not pinned: the nerfstudio components themselves and
PSNR (torchmetrics is absent; excluded from the fixture).

    python tests/golden/make_reference_model_golden.py"""
import dataclasses
import enum
import os
import sys

import numpy as np
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fruit_oracle as fo  # noqa: E402
from oracle import ns_torch as ns  # noqa: E402
from tests import util  # noqa: E402
from tests.golden import make_reference_field_golden as field_stubs  # noqa: E402

FRUIT_ONLY = ("semantic_loss_weight", "pass_semantic_gradients", "num_layers_semantic", "hidden_dim_semantics",
              "geo_feat_dim")


def nerfacto_config_class():
    """NerfactoModelConfig stand-in: the nerfstudio 0.3.2 defaults the oracle restates plus the
    switches FruitModel reads that keep their defaults."""
    fields = [(f.name, f.type, f) for f in dataclasses.fields(fo.FruitNerfModelConfig) if f.name not in FRUIT_ONLY]
    spec = []
    for name, typ, f in fields:
        if f.default is not dataclasses.MISSING:
            spec.append((name, typ, dataclasses.field(default=f.default)))
        else:
            spec.append((name, typ, dataclasses.field(default_factory=f.default_factory)))
    for name, default in (("background_color", "last_sample"), ("implementation", "torch"),
                          ("use_gradient_scaling", False), ("use_same_proposal_network", False),
                          ("proposal_initial_sampler", "piecewise")):
        spec.append((name, type(default), dataclasses.field(default=default)))
    return dataclasses.make_dataclass("NerfactoModelConfig", spec)


class Model(nn.Module):
    """nerfstudio.models.base_model.Model: the constructor protocol FruitModel relies on."""

    def __init__(self, config, scene_box, num_train_data, **kwargs):
        super().__init__()
        self.config, self.scene_box, self.num_train_data, self.kwargs = config, scene_box, num_train_data, kwargs
        self.collider = None
        self.populate_modules()
        self.callbacks = None
        self.device_indicator_param = nn.Parameter(torch.empty(0))

    @property
    def device(self):
        return self.device_indicator_param.device

    def populate_modules(self):
        pass


class SceneBox:
    def __init__(self, aabb):
        self.aabb = aabb


class Semantics:
    def __init__(self, colors):
        self.colors = colors


class RGBRenderer(nn.Module):
    def __init__(self, background_color="random"):
        super().__init__()
        assert background_color == "last_sample"
        self.background_color = background_color

    def forward(self, rgb, weights):
        return ns.render_rgb_last_sample(rgb, weights, self.training)


class AccumulationRenderer(nn.Module):
    def forward(self, weights):
        return ns.render_accumulation(weights)


class DepthRenderer(nn.Module):
    def forward(self, weights, ray_samples):
        return ns.render_depth_median(weights, ray_samples)


class SemanticRenderer(nn.Module):
    def forward(self, semantics, weights):
        return ns.render_semantics(semantics, weights)


class _Inert(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, *a, **k):
        return torch.tensor(float("nan"))


class TrainingCallbackLocation(enum.Enum):
    BEFORE_TRAIN_ITERATION = 1
    AFTER_TRAIN_ITERATION = 2


class TrainingCallback:
    def __init__(self, where_to_run, func, update_every_num_iters=None, iters=None, args=None, kwargs=None):
        self.where_to_run, self.func, self.update_every_num_iters = where_to_run, func, update_every_num_iters


class HashMLPDensityField(ns.HashMLPDensityField):
    def __init__(self, aabb, implementation=None, **kw):
        super().__init__(aabb, **kw)


def install():
    field_stubs.install()
    import nerfstudio.models.base_model as m
    m.Model = Model
    import nerfstudio.models.nerfacto as m
    m.NerfactoModelConfig = nerfacto_config_class()
    import nerfstudio.data.dataparsers.base_dataparser as m
    m.Semantics = Semantics
    import nerfstudio.engine.callbacks as m
    m.TrainingCallback, m.TrainingCallbackLocation = TrainingCallback, TrainingCallbackLocation
    import nerfstudio.field_components.spatial_distortions as m
    m.SceneContraction = ns.SceneContraction
    import nerfstudio.fields.density_fields as m
    m.HashMLPDensityField = HashMLPDensityField
    import nerfstudio.model_components.losses as m
    m.MSELoss, m.distortion_loss, m.interlevel_loss = nn.MSELoss, ns.distortion_loss, ns.interlevel_loss
    import nerfstudio.model_components.renderers as m
    m.AccumulationRenderer, m.DepthRenderer, m.RGBRenderer = AccumulationRenderer, DepthRenderer, RGBRenderer
    m.SemanticRenderer, m.UncertaintyRenderer = SemanticRenderer, _Inert
    import nerfstudio.model_components.ray_samplers as m
    m.ProposalNetworkSampler, m.SpacedSampler = ns.ProposalNetworkSampler, ns.SpacedSampler
    import nerfstudio.model_components.scene_colliders as m
    m.NearFarCollider = ns.NearFarCollider
    import nerfstudio.cameras.rays as m
    m.RayBundle, m.RaySamples, m.Frustums = ns.RayBundle, ns.RaySamples, ns.Frustums
    import torchmetrics as m
    m.PeakSignalNoiseRatio, m.JaccardIndex = _Inert, _Inert
    import torchmetrics.image.lpip as m
    m.LearnedPerceptualImagePatchSimilarity = _Inert


N_IMAGES, R = 5, 40
STEPS = 14                      # crosses step 10, where the proposal networks stop being updated every step


def inputs():
    o, d, pa, cam = util.random_rays(R, N_IMAGES, seed=21)
    g = torch.Generator().manual_seed(22)
    batch = {"image": torch.rand(R, 3, generator=g), "fruit_mask": (torch.rand(R, 1, generator=g) > 0.6).float()}
    return o, d, pa, cam, batch


def small_model_config(cls):
    oc = util.small_config(log2=10, prop_log2=8)
    cfg = cls()
    for k, v in vars(oc).items():
        if hasattr(cfg, k):
            setattr(cfg, k, v)
    return cfg


def main():
    install()
    from fruit_nerf.fruit_nerf import FruitModel, FruitNerfModelConfig          # the reference's classes
    small = np.load(os.path.join(ROOT, "tests", "golden", "fruit_nerf_small.npz"))
    sd = {k[4:]: torch.from_numpy(small[k]) for k in small.files if k.startswith("sd::")}
    aabb = sd["field.aabb"]
    semantics = Semantics(colors=torch.tensor([0.0, 255.0]) / 255.0)      # fruitnerf_dataparser.py:251-258

    def build(test_mode):
        m = FruitModel(config=small_model_config(FruitNerfModelConfig), metadata={"semantics": semantics},
                       scene_box=SceneBox(aabb), num_train_data=N_IMAGES, device="cpu", grad_scaler=None,
                       test_mode=test_mode)
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert set(missing) == {"device_indicator_param"} and not unexpected, (missing, unexpected)
        return m

    out = {}
    o, d, pa, cam, batch = inputs()
    # ---- training: the reference's own callbacks drive the anneal and the proposal update schedule ------------------
    model = build("val")
    model.train()
    cbs = model.get_training_callbacks(None)
    before = [c for c in cbs if TrainingCallbackLocation.BEFORE_TRAIN_ITERATION in c.where_to_run]
    after = [c for c in cbs if TrainingCallbackLocation.AFTER_TRAIN_ITERATION in c.where_to_run]
    out["param_groups"] = np.array(sorted(model.get_param_groups().keys()))
    for step in range(STEPS):
        for c in before:
            c.func(step)
        torch.manual_seed(1000 + step)
        res = model(ns.RayBundle(o, d, pa, camera_indices=cam))
        ld = model.get_loss_dict(res, batch)
        md = model.get_metrics_dict(res, batch)
        model.zero_grad()
        sum(ld.values()).backward()
        for k, v in ld.items():
            out[f"train::{step}::loss::{k}"] = np.float32(v.item())
        out[f"train::{step}::distortion"] = np.float32(md["distortion"].item())
        out[f"train::{step}::anneal"] = np.float64(model.proposal_sampler._anneal)
        out[f"train::{step}::prop_has_grad"] = np.bool_(res["weights_list"][0].requires_grad)
        if step in (0, 11, STEPS - 1):
            for k in ("rgb", "accumulation", "depth", "semantics", "prop_depth_0", "prop_depth_1"):
                out[f"train::{step}::{k}"] = res[k].detach().numpy()
            out[f"train::{step}::labels"] = res["semantics_colormap"].numpy()
            for i in range(3):
                out[f"train::{step}::weights{i}"] = res["weights_list"][i].detach().numpy()
            gs = {n: (p.grad.double().abs().sum().item() if p.grad is not None else 0.0)
                  for n, p in model.named_parameters() if n != "device_indicator_param"}
            for n, v in gs.items():
                out[f"train::{step}::gradsum::{n}"] = np.float64(v)
        for c in after:
            c.func(step)
    # ---- eval / inference --------------------------------------------------------------------------------------------
    for name, test_mode in (("eval", "val"), ("inference", "inference")):
        model = build(test_mode)
        model.eval()
        with torch.no_grad():
            res = model(ns.RayBundle(o, d, pa, camera_indices=cam))
        for k in ("rgb", "accumulation", "depth", "semantics", "prop_depth_0", "prop_depth_1"):
            out[f"{name}::{k}"] = res[k].numpy()
        out[f"{name}::colormap"] = res["semantics_colormap"].numpy()
        ld = model.get_loss_dict(res, batch)
        assert set(ld) == {"rgb_loss", "semantics_loss"}                       # no interlevel term outside training
        for k, v in ld.items():
            out[f"{name}::loss::{k}"] = np.float32(v.item())
    # ---- export: setup_inference swaps the sampler and drops the contraction ------------------------------------------
    # The exporter (scripts/exporter.py:86-94) calls eval_setup(), which puts the pipeline in eval mode, and THEN
    # model.setup_inference(), which constructs a fresh UniformSamplerWithNoise: a new nn.Module is in training mode,
    # so the reference's export sampler jitters every bin edge (`train_stratified and self.training`,
    # ray_samplers.py:79-87).  Both behaviours are pinned: "export" = the flow as the reference runs it (jitter drawn
    # from the seeded global generator), "export_centres" = the same model with the sampler switched to eval mode
    # (bin centres: what the product's lattice export implements).
    corners = fo.get_corners_of_aabb(((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)))
    pts, vec = fo.sample_surface_points(corners, 6)
    rb = fo.OrthographicRayGenerator(pts, vec, 64)(1)
    for name in ("export", "export_centres"):
        model = build("export")
        model.eval()
        model.setup_inference(render_rgb=True, num_inference_samples=9)
        out[f"{name}::sampler_training"] = np.bool_(model.proposal_sampler.training)
        if name == "export_centres":
            model.proposal_sampler.eval()
        torch.manual_seed(77)
        with torch.no_grad():
            res = model(rb)
        for k in ("rgb", "point_location", "semantics", "density", "semantics_colormap"):
            out[f"{name}::{k}"] = res[k].numpy()
        out[f"{name}::keys"] = np.array(sorted(res.keys()))
    path = os.path.join(ROOT, "tests", "golden", "reference_model.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
