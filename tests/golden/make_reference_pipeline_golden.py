#!/usr/bin/env python3
"""Generates tests/golden/reference_pipeline.npz: the checkpoint contract of the drop-in boundary, pinned by running the
reference's OWN code (/root/reference, read at generation time only):

  * the reference `FruitModel` (fruit_nerf/fruit_nerf.py over the stubs of make_reference_model_golden.py, whose `Model`
    base registers nerfstudio's zero-length `device_indicator_param`) -> the full `state_dict()` key list with shapes:
    what a checkpoint written by a Nerfstudio Trainer holds under `pipeline._model.*`;
  * the reference `FruitPipeline.load_pipeline` (fruit_nerf/fruit_pipeline.py:229-240: strip `module.`,
    `model.update_to_step(step)`, `load_state_dict(state, strict=True)`) executed on a pipeline whose `_model` is the
    PRODUCT model (fruitnerf_amd.fruit_nerf.FruitModel, on the CPU: no kernel runs) and whose datamanager holds the
    product's camera optimiser, with a checkpoint produced from the reference model's state dict — once as a single
    process wrote it and once `module.`-prefixed, as DDP wrote it (`fruit_pipeline.py:116`).

The fixture keeps the key list, the shapes and which load variants went through; tests/test_reference_pins.py checks the
product's key set against it both ways and repeats the load with the fixture's keys (and, where /root/reference exists,
runs `run_reference_load_pipeline()` below live).

    python tests/golden/make_reference_pipeline_golden.py"""
import os
import sys

import numpy as np
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.golden import make_reference_model_golden as model_stubs  # noqa: E402

N_IMAGES = model_stubs.N_IMAGES


class Pipeline(nn.Module):
    """nerfstudio.pipelines.base_pipeline.Pipeline (0.3.2): an nn.Module holding `_model` and `datamanager`; `model`
    unwraps DDP (`module_wrapper`)."""

    @property
    def model(self):
        m = self._model
        return m.module if isinstance(m, nn.parallel.DistributedDataParallel) else m

    @property
    def device(self):
        return self.model.device


_installed = False


def install():
    global _installed
    if _installed:
        return
    model_stubs.install()
    import nerfstudio.pipelines.base_pipeline as m
    m.Pipeline = Pipeline
    import nerfstudio.configs.base_config as m      # (a real module object, so that `cfg.InstantiateConfig` is a class)
    m.InstantiateConfig = type("InstantiateConfig", (), {})
    _installed = True


def reference_model():
    """The reference's FruitModel at the small test configuration (tables of 2^10 / 2^8 rows)."""
    install()
    from fruit_nerf.fruit_nerf import FruitModel, FruitNerfModelConfig
    semantics = model_stubs.Semantics(colors=torch.tensor([0.0, 255.0]) / 255.0)
    aabb = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
    torch.manual_seed(5)
    return FruitModel(config=model_stubs.small_model_config(FruitNerfModelConfig), metadata={"semantics": semantics},
                      scene_box=model_stubs.SceneBox(aabb), num_train_data=N_IMAGES, device="cpu", grad_scaler=None,
                      test_mode="val")


def product_model():
    from fruitnerf_amd.data.semantics import apple_metadata
    from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
    from tests import util
    cfg = FruitNerfModelConfig()
    for k, v in vars(util.small_config(log2=10, prop_log2=8)).items():
        if hasattr(cfg, k):
            setattr(cfg, k, v)
    torch.manual_seed(6)
    return FruitModel(cfg, apple_metadata(), num_train_data=N_IMAGES, device="cpu", test_mode="val")


class _DataManager(nn.Module):
    """What of nerfstudio's VanillaDataManager reaches the pipeline's state dict: the camera optimiser's poses."""

    def __init__(self):
        super().__init__()
        from fruitnerf_amd.cameras.camera_optimizers import CameraOptimizer, CameraOptimizerConfig
        self.train_camera_optimizer = CameraOptimizer(CameraOptimizerConfig(mode="SO3xR3"), N_IMAGES, "cpu")


def run_reference_load_pipeline(prefix: str, ref_state=None):
    """-> (product model after the load, the checkpoint that was loaded).  Raises what the reference's load raises."""
    install()
    from fruit_nerf.fruit_pipeline import FruitPipeline
    if ref_state is None:
        ref_state = reference_model().state_dict()
    pipe = FruitPipeline.__new__(FruitPipeline)
    nn.Module.__init__(pipe)
    pipe._model = product_model()
    pipe.datamanager = _DataManager()
    g = torch.Generator().manual_seed(7)
    loaded = {f"{prefix}_model.{k}": v.clone() for k, v in ref_state.items()}
    loaded[f"{prefix}datamanager.train_camera_optimizer.pose_adjustment"] = torch.randn(N_IMAGES, 6, generator=g) * 1e-3
    FruitPipeline.load_pipeline(pipe, loaded, 1234)
    return pipe, loaded


def main():
    ref = reference_model()
    sd = ref.state_dict()
    out = {"state_keys": np.array(list(sd.keys())),
           "state_shapes": np.array([",".join(str(int(s)) for s in v.shape) for v in sd.values()]),
           "state_dtypes": np.array([str(v.dtype) for v in sd.values()])}
    for name, prefix in (("plain", ""), ("ddp", "module.")):
        pipe, loaded = run_reference_load_pipeline(prefix, sd)
        got = pipe._model.state_dict()
        for k, v in sd.items():
            assert torch.equal(got[k], v), k
        assert torch.equal(pipe.datamanager.train_camera_optimizer.pose_adjustment.data,
                           loaded[f"{prefix}datamanager.train_camera_optimizer.pose_adjustment"])
        out[f"loaded::{name}"] = np.bool_(True)
    path = os.path.join(ROOT, "tests", "golden", "reference_pipeline.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(sd), "state-dict keys; load_pipeline ok: plain, ddp")


if __name__ == "__main__":
    main()
