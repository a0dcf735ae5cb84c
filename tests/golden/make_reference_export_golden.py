#!/usr/bin/env python3
"""Generates tests/golden/reference_export.npz by running the reference's OWN export loop
(/root/reference/fruit_nerf/export/exporter_utils.py::sample_volume, :47-258) driven by its own datamanager methods
(data/fruit_datamanager.py::FruitDataManager.setup_inference / next_sample_volume, :157-172,199-204), its own
OrthographicRayGenerator and its own FruitModel in 'export' mode — over oracle/ns_torch.py for the nerfstudio
components (see make_reference_model_golden.py) and a four-line stand-in for Open3D's PointCloud (points, colours,
`scale(s, center)`: p -> (p - center) * s + center), which the loop only uses as a container.

Pins oracle/fruit_oracle.py::sample_volume: thresholds (density >= 70, logit >= 3, label >= 0.999), the three point
sets and their colour columns, per-set colour normalisation, the 1/scale * 2 rescaling, batch order.

    python tests/golden/make_reference_export_golden.py"""
import os
import pathlib
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.golden import make_reference_model_golden as model_stubs  # noqa: E402

AABB = ((-0.9, -0.8, -1.0), (0.7, 0.8, 1.0))
N_SIDE, BATCH, SCALE = 12, 50, 0.5


class PointCloud:
    def __init__(self):
        self.points, self.colors = None, None

    def scale(self, s, center):
        self.points = (np.asarray(self.points) - center) * s + center
        return self


def export_state_dict():
    """fruit_nerf_small.npz weights with the density and fruit logits pushed up so that all three sets are non-empty
    and different on a 12 x 9 x 12 lattice."""
    small = np.load(os.path.join(ROOT, "tests", "golden", "fruit_nerf_small.npz"))
    sd = {k[4:]: torch.from_numpy(small[k]).clone() for k in small.files if k.startswith("sd::")}
    # (mlp_base is nn.Sequential(mlp_base_grid, mlp_base_mlp): both aliases of a tensor are in the state dict)
    for k in ("field.mlp_base_grid.hash_table", "field.mlp_base.0.hash_table"):
        sd[k] *= 8.0
    for k in ("field.mlp_base_mlp.layers.1.bias", "field.mlp_base.1.layers.1.bias"):
        sd[k][0] += 1.5
    sd["field.field_head_semantics.net.weight"] *= 2.0
    sd["field.field_head_semantics.net.bias"] += 2.9
    return sd


def install():
    model_stubs.install()
    import open3d as o3d
    o3d.geometry.PointCloud = PointCloud
    o3d.utility.Vector3dVector = lambda a: np.array(a, dtype=np.float64)
    import rich.console
    import nerfstudio.utils.rich_utils as ru
    ru.CONSOLE = rich.console.Console(quiet=True)


def main():
    install()
    from fruit_nerf.fruit_nerf import FruitModel, FruitNerfModelConfig
    from fruit_nerf.data.fruit_datamanager import FruitDataManager
    from fruit_nerf.export.exporter_utils import sample_volume
    sd = export_state_dict()
    semantics = model_stubs.Semantics(colors=torch.tensor([0.0, 255.0]) / 255.0)
    out = {"aabb": np.array(AABB), "n_side": np.int64(N_SIDE), "batch": np.int64(BATCH), "scale": np.float64(SCALE)}
    for name in ("as_run", "centres"):
        model = FruitModel(config=model_stubs.small_model_config(FruitNerfModelConfig), metadata={"semantics": semantics},
                           scene_box=model_stubs.SceneBox(sd["field.aabb"]), num_train_data=model_stubs.N_IMAGES,
                           device="cpu", grad_scaler=None, test_mode="export")
        model.load_state_dict(sd, strict=False)
        model.eval()                                              # eval_setup(): pipeline.eval() comes first ...
        model.setup_inference(render_rgb=True, num_inference_samples=N_SIDE)    # ... then the sampler is created
        if name == "centres":
            model.proposal_sampler.eval()
        dm = types.SimpleNamespace(device="cpu", train_count=0,
                                   config=types.SimpleNamespace(eval_num_rays_per_batch=BATCH))
        dm.setup_inference = types.MethodType(FruitDataManager.setup_inference, dm)
        dm.next_sample_volume = types.MethodType(FruitDataManager.next_sample_volume, dm)
        num_points = dm.setup_inference(aabb=AABB, num_points=N_SIDE)
        pipeline = types.SimpleNamespace(model=model, datamanager=dm)
        torch.manual_seed(123)
        pcds = sample_volume(pipeline=pipeline, num_points=num_points, output_dir=pathlib.Path("/tmp/fnr_export"),
                             config=types.SimpleNamespace(load_dir=pathlib.Path("outputs/scene/fruit_nerf/run")),
                             transform_json={"transform": np.eye(4)[:3].tolist(), "scale": SCALE})
        out[f"{name}::num_rays"] = np.int64(num_points)
        out[f"{name}::batches"] = np.int64(dm.train_count)
        for set_name, entry in pcds.items():
            out[f"{name}::{set_name}::points"] = np.asarray(entry["pcd"].points)
            out[f"{name}::{set_name}::colors"] = np.asarray(entry["pcd"].colors)
            out[f"{name}::{set_name}::path"] = np.array(entry["path"])
        print(name, {k: len(v["pcd"].points) for k, v in pcds.items()}, "rays", num_points, "batches", dm.train_count)
    path = os.path.join(ROOT, "tests", "golden", "reference_export.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
