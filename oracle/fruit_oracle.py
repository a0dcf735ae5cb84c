"""Oracle part 2: restatement of the in-tree FruitNeRF hot path on top of oracle/ns_torch.py
(TEST INFRASTRUCTURE; see oracle/__init__.py).  FruitField, the export lattice / ray batches / export sampler are
pinned against the reference's own code executed over ns_torch (tests/golden/make_reference_*golden.py ->
tests/test_reference_pins.py); the nerfstudio components underneath remain an unpinned restatement.

Each function cites the reference file:line (relative to /root/reference/) it follows.
Pure PyTorch, CPU, fp32 (nerfstudio disables mixed precision on CPU).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor, nn

from . import ns_torch as ns


# --------------------------------------------------------------------------------------
# FruitField  — fruit_nerf/fruit_field.py:43-301
# --------------------------------------------------------------------------------------


class FruitField(nn.Module):
    def __init__(self, aabb: Tensor, num_images: int, num_layers=2, hidden_dim=64, geo_feat_dim=15, num_levels=16,
                 base_res=16, max_res=2048, log2_hashmap_size=19, num_layers_color=3, num_layers_semantic=2,
                 features_per_level=2, hidden_dim_color=64, hidden_dim_semantics=64, hidden_dim_transient=64,
                 appearance_embedding_dim=32, test_mode=None, num_semantic_classes=1,
                 pass_semantic_gradients=False, use_average_appearance_embedding=False, spatial_distortion=None):
        super().__init__()
        # fruit_field.py:98-113
        self.register_buffer("aabb", aabb)
        self.geo_feat_dim = geo_feat_dim
        self.register_buffer("max_res", torch.tensor(max_res))
        self.register_buffer("num_levels", torch.tensor(num_levels))
        self.register_buffer("log2_hashmap_size", torch.tensor(log2_hashmap_size))
        self.spatial_distortion = spatial_distortion
        self.num_images = num_images
        self.appearance_embedding_dim = appearance_embedding_dim
        self.embedding_appearance = ns.Embedding(num_images, appearance_embedding_dim)
        self.use_average_appearance_embedding = use_average_appearance_embedding
        self.test_mode = test_mode
        self.pass_semantic_gradients = pass_semantic_gradients
        # fruit_field.py:115-166
        self.direction_encoding = ns.SHEncoding(levels=4)
        self.mlp_base_grid = ns.HashEncoding(num_levels=num_levels, min_res=base_res, max_res=max_res,
                                             log2_hashmap_size=log2_hashmap_size,
                                             features_per_level=features_per_level)
        self.mlp_base_mlp = ns.MLP(in_dim=self.mlp_base_grid.get_out_dim(), num_layers=num_layers,
                                   layer_width=hidden_dim, out_dim=1 + geo_feat_dim, activation=nn.ReLU())
        self.mlp_base = nn.Sequential(self.mlp_base_grid, self.mlp_base_mlp)
        self.mlp_semantics = ns.MLP(in_dim=geo_feat_dim, num_layers=num_layers_semantic,
                                    layer_width=hidden_dim_semantics, out_dim=hidden_dim_transient,
                                    activation=nn.ReLU())
        # SemanticFieldHead == FieldHead == nn.Linear wrapped as `.net` (components/field_heads.py:29-40)
        self.field_head_semantics = nn.Module()
        self.field_head_semantics.net = nn.Linear(self.mlp_semantics.get_out_dim(), num_semantic_classes)
        self.mlp_head = ns.MLP(in_dim=self.direction_encoding.get_out_dim() + geo_feat_dim + appearance_embedding_dim,
                               num_layers=num_layers_color, layer_width=hidden_dim_color, out_dim=3,
                               activation=nn.ReLU(), out_activation=nn.Sigmoid())

    def get_density(self, ray_samples: ns.RaySamples) -> Tuple[Tensor, Tensor]:
        # fruit_field.py:168-193
        if self.spatial_distortion is not None:
            positions = ray_samples.frustums.get_positions()
            positions = self.spatial_distortion(positions)
            positions = (positions + 2.0) / 4.0
        else:
            positions = ns.get_normalized_positions(ray_samples.frustums.get_positions(), self.aabb)
        selector = ((positions > 0.0) & (positions < 1.0)).all(dim=-1)
        positions = positions * selector[..., None]
        self._sample_locations = positions
        if not self._sample_locations.requires_grad:      # fruit_field.py:181-182
            self._sample_locations.requires_grad = True
        positions_flat = positions.view(-1, 3)
        h = self.mlp_base(positions_flat).view(*ray_samples.frustums.shape, -1)
        density_before_activation, base_mlp_out = torch.split(h, [1, self.geo_feat_dim], dim=-1)
        self._density_before_activation = density_before_activation
        density = ns.trunc_exp(density_before_activation.to(positions))
        density = density * selector[..., None]
        return density, base_mlp_out

    def _semantics(self, density_embedding, outputs_shape):
        semantics_input = density_embedding.view(-1, self.geo_feat_dim)
        if not self.pass_semantic_gradients:
            semantics_input = semantics_input.detach()
        x = self.mlp_semantics(semantics_input).view(*outputs_shape, -1)
        return self.field_head_semantics.net(x)

    def get_inference_outputs(self, ray_samples, density_embedding) -> Dict[str, Tensor]:
        # fruit_field.py:195-232 : always the mean appearance embedding
        outputs = {}
        outputs_shape = ray_samples.frustums.directions.shape[:-1]
        outputs["semantics"] = self._semantics(density_embedding, outputs_shape).to(torch.float32)
        directions = ns.shift_directions_for_tcnn(ray_samples.frustums.directions)
        d = self.direction_encoding(directions.reshape(-1, 3))
        embedded_appearance = torch.ones((*directions.shape[:-1], self.appearance_embedding_dim)
                                         ) * self.embedding_appearance.mean(dim=0)
        h = torch.cat([d, density_embedding.view(-1, self.geo_feat_dim),
                       embedded_appearance.view(-1, self.appearance_embedding_dim)], dim=-1)
        outputs["rgb"] = self.mlp_head(h).view(*outputs_shape, -1)
        return outputs

    def get_outputs(self, ray_samples, density_embedding) -> Dict[str, Tensor]:
        # fruit_field.py:234-281
        assert density_embedding is not None
        outputs = {}
        if ray_samples.camera_indices is None:
            raise AttributeError("Camera indices are not provided.")
        camera_indices = ray_samples.camera_indices.squeeze()
        directions = ns.shift_directions_for_tcnn(ray_samples.frustums.directions)
        d = self.direction_encoding(directions.reshape(-1, 3))
        outputs_shape = ray_samples.frustums.directions.shape[:-1]
        if self.training:
            embedded_appearance = self.embedding_appearance(camera_indices)
        else:
            if self.use_average_appearance_embedding:
                embedded_appearance = torch.ones((*directions.shape[:-1], self.appearance_embedding_dim)
                                                 ) * self.embedding_appearance.mean(dim=0)
            else:
                embedded_appearance = torch.zeros((*directions.shape[:-1], self.appearance_embedding_dim))
        outputs["semantics"] = self._semantics(density_embedding, outputs_shape)
        h = torch.cat([d, density_embedding.view(-1, self.geo_feat_dim),
                       embedded_appearance.view(-1, self.appearance_embedding_dim)], dim=-1)
        outputs["rgb"] = self.mlp_head(h).view(*outputs_shape, -1)
        return outputs

    def forward(self, ray_samples) -> Dict[str, Tensor]:
        # fruit_field.py:283-301
        density, density_embedding = self.get_density(ray_samples)
        if self.test_mode == "inference" or self.test_mode == "export":
            field_outputs = self.get_inference_outputs(ray_samples, density_embedding)
        else:
            field_outputs = self.get_outputs(ray_samples, density_embedding)
        field_outputs["density"] = density
        return field_outputs


# --------------------------------------------------------------------------------------
# Export-time sampler / ray source — components/ray_samplers.py:31-104, components/ray_generators.py:24-66,
# data/fruit_datamanager.py:42-121,157-172,199-204
# --------------------------------------------------------------------------------------


class UniformSamplerWithNoise(ns.SpacedSampler):
    def __init__(self, num_samples=None, train_stratified=True, single_jitter=False):
        super().__init__(num_samples=num_samples, spacing_fn=lambda x: x, spacing_fn_inv=lambda x: x,
                         train_stratified=train_stratified, single_jitter=single_jitter)


def get_corners_of_aabb(aabb) -> Tensor:
    # fruit_datamanager.py:42-68
    min_coords = aabb[0]
    max_coords = aabb[1]
    return torch.asarray([
        [min_coords[0], min_coords[1], min_coords[2]],
        [max_coords[0], min_coords[1], min_coords[2]],
        [min_coords[0], max_coords[1], min_coords[2]],
        [max_coords[0], max_coords[1], min_coords[2]],
        [min_coords[0], min_coords[1], max_coords[2]],
        [max_coords[0], min_coords[1], max_coords[2]],
        [min_coords[0], max_coords[1], max_coords[2]],
        [max_coords[0], max_coords[1], max_coords[2]],
    ])


def sample_surface_points(aabb: Tensor, n: int) -> Tuple[Tensor, Tensor]:
    # fruit_datamanager.py:71-121  (aabb here = the 8 corners)
    corner_1 = aabb[0]
    corner_2 = aabb[1]
    corner_3 = aabb[2]
    dx_y_z = torch.abs(torch.max(aabb, axis=0).values - torch.min(aabb, axis=0).values)
    constant_axis_part_pos = int(torch.argmax(torch.logical_and((corner_1 == corner_2), (corner_2 == corner_3)).to(int)))
    start_x_pos = torch.argmax(torch.abs(corner_1 - corner_2))
    x = torch.linspace(corner_1[start_x_pos], corner_2[start_x_pos],
                       int(dx_y_z[0] / dx_y_z[constant_axis_part_pos] * n), dtype=torch.float32)
    start_y_pos = torch.argmax(torch.abs(corner_1 - corner_3))
    y = torch.linspace(corner_1[start_y_pos], corner_3[start_y_pos],
                       int(dx_y_z[1] / dx_y_z[constant_axis_part_pos] * n), dtype=torch.float32)
    xx, yy = torch.meshgrid(x, y, indexing="ij")
    surface_points = torch.column_stack(
        (xx.flatten(), yy.flatten(), torch.full_like(xx.flatten(), corner_3[constant_axis_part_pos])))
    corner_4 = aabb[-1]
    plane_vector = torch.asarray([[0, 0, torch.sign(corner_4[constant_axis_part_pos]) * torch.abs(
        corner_1[constant_axis_part_pos]) + torch.abs(corner_4[constant_axis_part_pos])]], dtype=torch.float32)
    return surface_points.clone(), plane_vector


class OrthographicRayGenerator:
    # components/ray_generators.py:24-66
    def __init__(self, surface_points, plane_normal, ray_batch_size):
        self.surface_points = surface_points
        self.surface_normal = torch.nn.functional.normalize(plane_normal)
        self.surface_vector_norm = torch.linalg.norm(plane_normal)
        self.ray_batch_size = ray_batch_size

    def __call__(self, count: int) -> ns.RayBundle:
        start = self.ray_batch_size * (count - 1)
        end = self.ray_batch_size * count
        if self.ray_batch_size * count >= self.surface_points.shape[0]:
            end = self.surface_points.shape[0]
        num_points = self.surface_points[start:end].shape[0]
        return ns.RayBundle(origins=self.surface_points[start:end],
                            directions=self.surface_normal.repeat(num_points, 1),
                            pixel_area=torch.zeros(num_points, 1),
                            nears=torch.zeros(num_points, 1),
                            fars=torch.ones(num_points, 1) * self.surface_vector_norm)


# --------------------------------------------------------------------------------------
# FruitModel — fruit_nerf/fruit_nerf.py:50-458 (+ resolved Nerfacto 0.3.2 defaults)
# --------------------------------------------------------------------------------------


@dataclass
class FruitNerfModelConfig:
    near_plane: float = 0.05
    far_plane: float = 1000.0
    num_levels: int = 16
    max_res: int = 2048
    log2_hashmap_size: int = 19
    num_proposal_samples_per_ray: Tuple[int, ...] = (256, 96)
    num_nerf_samples_per_ray: int = 48
    proposal_update_every: int = 5
    proposal_warmup: int = 5000
    num_proposal_iterations: int = 2
    proposal_net_args_list: List[Dict] = field(default_factory=lambda: [
        {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 128, "use_linear": False},
        {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 256, "use_linear": False},
    ])
    interlevel_loss_mult: float = 1.0
    use_proposal_weight_anneal: bool = True
    proposal_weights_anneal_slope: float = 10.0
    proposal_weights_anneal_max_num_iters: int = 1000
    use_single_jitter: bool = True
    use_average_appearance_embedding: bool = True
    disable_scene_contraction: bool = False
    eval_num_rays_per_chunk: int = 1 << 15
    # FruitNerfModelConfig, fruit_nerf.py:50-59
    semantic_loss_weight: float = 1.0
    pass_semantic_gradients: bool = False
    num_layers_semantic: int = 2
    hidden_dim_semantics: int = 64
    geo_feat_dim: int = 15


class FruitModel(nn.Module):
    def __init__(self, config: FruitNerfModelConfig, num_train_data: int, aabb: Optional[Tensor] = None,
                 test_mode: Optional[str] = None):
        super().__init__()
        self.config = config
        self.test_mode = test_mode
        if aabb is None:
            aabb = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])  # fruitnerf_dataparser.py:218-223
        scene_contraction = None if config.disable_scene_contraction else ns.SceneContraction(order=float("inf"))
        # fruit_nerf.py:88-103 — only these fields reach the field
        self.field = FruitField(aabb, num_levels=config.num_levels, max_res=config.max_res,
                                num_layers_semantic=config.num_layers_semantic,
                                hidden_dim_semantics=config.hidden_dim_semantics,
                                log2_hashmap_size=config.log2_hashmap_size, spatial_distortion=scene_contraction,
                                num_images=num_train_data, geo_feat_dim=config.geo_feat_dim,
                                use_average_appearance_embedding=config.use_average_appearance_embedding,
                                test_mode=test_mode, num_semantic_classes=1,
                                pass_semantic_gradients=config.pass_semantic_gradients)
        # fruit_nerf.py:104-129
        self.proposal_networks = nn.ModuleList()
        for i in range(config.num_proposal_iterations):
            args = dict(config.proposal_net_args_list[min(i, len(config.proposal_net_args_list) - 1)])
            self.proposal_networks.append(ns.HashMLPDensityField(aabb, spatial_distortion=scene_contraction, **args))
        self.density_fns = [net.density_fn for net in self.proposal_networks]

        def update_schedule(step):  # fruit_nerf.py:131-136
            return np.clip(np.interp(step, [0, config.proposal_warmup], [0, config.proposal_update_every]),
                           1, config.proposal_update_every)

        self.proposal_sampler = ns.ProposalNetworkSampler(
            num_nerf_samples_per_ray=config.num_nerf_samples_per_ray,
            num_proposal_samples_per_ray=config.num_proposal_samples_per_ray,
            num_proposal_network_iterations=config.num_proposal_iterations,
            single_jitter=config.use_single_jitter, update_sched=update_schedule, initial_sampler=None)
        self.collider = ns.NearFarCollider(near_plane=config.near_plane, far_plane=config.far_plane)
        self.rgb_loss = nn.MSELoss()
        self.binary_cross_entropy_loss = nn.BCEWithLogitsLoss(reduction="mean")
        # nerfstudio Model.__init__ (the base class of the reference's FruitModel, fruit_nerf.py:62) registers this
        # zero-length parameter after populate_modules(); it is part of every checkpoint (tests/golden/reference_pipeline.npz
        # holds the reference model's own key list)
        self.device_indicator_param = nn.Parameter(torch.empty(0))

    def update_to_step(self, step: int) -> None:
        """nerfstudio Model.update_to_step: a no-op for this model (called by FruitPipeline.load_pipeline, fruit_pipeline.py:239)."""

    # fruit_nerf.py:179-183
    def setup_inference(self, render_rgb, num_inference_samples, sampler_mode_as_in_reference: bool = False):
        """The reference constructs the sampler here, after eval_setup() has already put the pipeline in eval mode
        (scripts/exporter.py:86-94), and a fresh nn.Module is in TRAINING mode: as the reference runs it, the export
        sampler jitters every bin edge (ray_samplers.py:79-87) — pinned by tests/test_reference_pins.py with
        sampler_mode_as_in_reference=True.  The default follows the model's mode instead (bin centres when evaluating):
        the deterministic lattice the product's export implements and the point-count parity tests rely on."""
        self.render_rgb = render_rgb
        self.num_inference_samples = num_inference_samples
        self.proposal_sampler = UniformSamplerWithNoise(num_samples=num_inference_samples, single_jitter=False)
        if not sampler_mode_as_in_reference:
            self.proposal_sampler.train(self.training)
        self.field.spatial_distortion = None

    def get_param_groups(self):  # fruit_nerf.py:185-189
        return {"proposal_networks": list(self.proposal_networks.parameters()),
                "fields": list(self.field.parameters())}

    def set_anneal(self, step):  # fruit_nerf.py:199-207
        N = self.config.proposal_weights_anneal_max_num_iters
        train_frac = np.clip(step / N, 0, 1)

        def bias(x, b):
            return b * x / ((b - 1) * x + 1)

        self.proposal_sampler.set_anneal(bias(train_frac, self.config.proposal_weights_anneal_slope))

    def get_export_outputs(self, ray_bundle):  # fruit_nerf.py:251-269
        outputs = {}
        ray_samples = self.proposal_sampler(ray_bundle)
        field_outputs = self.field.forward(ray_samples)
        outputs["rgb"] = field_outputs["rgb"]
        outputs["point_location"] = ray_samples.frustums.get_positions()
        outputs["semantics"] = field_outputs["semantics"][..., 0]
        outputs["density"] = field_outputs["density"][..., 0]
        semantic_labels = torch.sigmoid(outputs["semantics"])
        semantic_labels = torch.heaviside(semantic_labels - 0.9, torch.tensor(0.0)).to(torch.long)
        outputs["semantics_colormap"] = semantic_labels
        return outputs

    def get_outputs(self, ray_bundle, jitter=None):  # fruit_nerf.py:316-357 (== get_inference_outputs :272-314)
        ray_samples, weights_list, ray_samples_list = self.proposal_sampler(
            ray_bundle, density_fns=self.density_fns, jitter=jitter)
        field_outputs = self.field.forward(ray_samples)
        weights = ray_samples.get_weights(field_outputs["density"])
        weights_list.append(weights)
        ray_samples_list.append(ray_samples)
        rgb = ns.render_rgb_last_sample(field_outputs["rgb"], weights, self.training)
        depth = ns.render_depth_median(weights, ray_samples)
        accumulation = ns.render_accumulation(weights)
        outputs = {"rgb": rgb, "accumulation": accumulation, "depth": depth,
                   "weights_list": weights_list, "ray_samples_list": ray_samples_list}
        for i in range(self.config.num_proposal_iterations):
            outputs[f"prop_depth_{i}"] = ns.render_depth_median(weights_list[i], ray_samples_list[i])
        semantic_weights = weights
        if not self.config.pass_semantic_gradients:
            semantic_weights = semantic_weights.detach()
        outputs["semantics"] = ns.render_semantics(field_outputs["semantics"], semantic_weights)
        semantic_labels = torch.sigmoid(outputs["semantics"].detach())
        semantic_labels = torch.heaviside(semantic_labels - 0.9, torch.tensor(0.0)).to(torch.long)
        outputs["semantics_colormap"] = semantic_labels  # colormap lookup is host cosmetics (out of scope)
        return outputs

    def forward(self, ray_bundle, jitter=None):  # fruit_nerf.py:374-394
        ray_bundle = self.collider(ray_bundle)
        if self.test_mode == "export":
            return self.get_export_outputs(ray_bundle)
        return self.get_outputs(ray_bundle, jitter=jitter)

    def get_loss_dict(self, outputs, batch):  # fruit_nerf.py:359-372
        loss_dict = {}
        loss_dict["rgb_loss"] = self.rgb_loss(batch["image"], outputs["rgb"])
        loss_dict["semantics_loss"] = self.config.semantic_loss_weight * self.binary_cross_entropy_loss(
            outputs["semantics"], batch["fruit_mask"])
        if self.training:
            loss_dict["interlevel_loss"] = self.config.interlevel_loss_mult * ns.interlevel_loss(
                outputs["weights_list"], outputs["ray_samples_list"])
        return loss_dict

    def get_metrics_dict(self, outputs, batch):  # fruit_nerf.py:396-401
        mse = torch.mean((outputs["rgb"] - batch["image"]) ** 2)
        return {"psnr": 10.0 * torch.log10(1.0 / mse),  # torchmetrics PSNR(data_range=1.0)
                "distortion": ns.distortion_loss(outputs["weights_list"], outputs["ray_samples_list"])}


# --------------------------------------------------------------------------------------
# sample_volume — export/exporter_utils.py:47-258 (masks + gathers; Open3D objects out of scope)
# --------------------------------------------------------------------------------------


def sample_volume(model: FruitModel, aabb, num_points_per_side: int, num_rays_per_batch: int,
                  dataparser_scale: float = 1.0) -> Dict[str, Dict[str, Tensor]]:
    corners = get_corners_of_aabb(aabb)
    surface_points, plane_vector = sample_surface_points(corners, n=num_points_per_side)
    gen = OrthographicRayGenerator(surface_points, plane_vector, num_rays_per_batch)
    num_rays = surface_points.shape[0]
    pts = {"semantic_colormap": [], "semantic": [], "density": []}
    cols = {"semantic_colormap": [], "semantic": [], "density": []}
    done, count = 0, 0
    while done < num_rays:  # exporter_utils.py:94-95,172
        count += 1
        with torch.no_grad():
            outputs = model(gen(count))
        points_3d = outputs["point_location"].reshape((-1, 3))
        semantic = outputs["semantics"].reshape((-1, 1)).repeat((1, 3))
        semantics_colormap = outputs["semantics_colormap"].reshape((-1, 1)).repeat((1, 3))
        density = outputs["density"].reshape((-1, 1)).repeat((1, 3))
        rgb = outputs["rgb"].reshape((-1, 3))
        mask_sem = (semantic >= 3).sum(dim=1).to(bool)  # exporter_utils.py:111-114
        mask_den = (density >= 70).sum(dim=1).to(bool)
        mask_cm = (semantics_colormap >= 0.999).sum(dim=1).to(bool)
        for name, m, fourth in (("semantic_colormap", mask_cm & mask_den, semantic),
                                ("semantic", mask_sem & mask_den, semantic),
                                ("density", mask_den, density)):
            pts[name].append(points_3d[m])
            cols[name].append(torch.hstack([rgb[m], torch.sigmoid(fourth[m][:, 0]).unsqueeze(-1)]))
        done += outputs["point_location"].shape[0]
    out = {}
    for name in pts:
        p = torch.cat(pts[name], dim=0)
        c = torch.cat(cols[name], dim=0)
        if name != "semantic_colormap" and c.shape[0] != 0:
            c = c / c.max()  # exporter_utils.py:202-203,227-228
        p = p.double() * (1.0 / dataparser_scale) * 2.0  # exporter_utils.py:190-191
        out[name] = {"points": p, "colors": c.double()[:, :3]}
    return out
