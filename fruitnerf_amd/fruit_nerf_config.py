"""Method configurations of the plugin: `fruit_nerf`, `fruit_nerf_big`, `fruit_nerf_huge`.

The reference registers three `MethodSpecification(config=TrainerConfig(...))` objects as `nerfstudio.method_configs`
entry points (/root/reference/pyproject.toml:24-27 -> /root/reference/fruit_nerf/fruit_nerf_config.py:27,63,113).  This
module is the drop-in's counterpart:

  * `METHODS` — the hyper-parameters of the three methods as plain data (what reaches the hot path: batch sizes, the model
    fields `FruitModel` forwards to `FruitField`, optimiser + scheduler per parameter group, the camera optimiser).  It is
    the single source for `bench.py`, the tests and `training_setup()`; nothing here needs Nerfstudio.
  * `fruit_nerf_method`, `fruit_nerf_method_big`, `fruit_nerf_method_huge` — the entry-point objects of this repo's
    `pyproject.toml`.  Inside a Nerfstudio installation they are real `MethodSpecification`s whose pipeline builds
    `fruitnerf_amd.fruit_nerf.FruitModel` (same Trainer / optimiser / scheduler configuration as the reference's, with the
    reference's datamanager + dataparser when its package is importable); without Nerfstudio they are `MethodSpec`
    stand-ins carrying the same data, so the registration itself is testable here.  They are built on first access
    (module `__getattr__`): importing this module never imports Nerfstudio.
  * `training_setup(method, model, ...)` — the optimisers of a method for this repo's own loop (`training.TrainingSteps`):
    `FusedAdam` over the model's parameter groups and `CameraOptimizer` + `CameraAdam`, configured from `METHODS`.

Only the model fields that reach the hot path differ between the methods: `hidden_dim`, `hidden_dim_color` and
`appearance_embed_dim` of the big / huge configs are never forwarded to `FruitField` (fruit_nerf.py:88-103);
they are kept under `model_ignored` for the record.
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field
from typing import Any, Dict, Optional


def _exp_decay(lr_final: float, max_steps: int) -> dict:
    return {"kind": "exponential_decay", "lr_final": lr_final, "max_steps": max_steps}


# fruit_nerf_config.py:27-61 / :63-111 / :113-164.  optimizers: nerfstudio AdamOptimizerConfig / RAdamOptimizerConfig
# (betas (0.9, 0.999), weight_decay 0 unless given) + ExponentialDecaySchedulerConfig or no scheduler.
METHODS: Dict[str, dict] = {
    "fruit_nerf": dict(
        description="Base config for FruitNeRF",
        trainer=dict(steps_per_eval_batch=500, steps_per_save=2000, max_num_iterations=30000, mixed_precision=True,
                     viewer_num_rays_per_chunk=1 << 13),
        datamanager=dict(train_num_rays_per_batch=4096, eval_num_rays_per_batch=4096),
        camera_optimizer=dict(mode="SO3xR3", algorithm="adam", lr=6e-4, eps=1e-8, weight_decay=1e-2,
                              scheduler=_exp_decay(6e-6, 200000)),
        model=dict(eval_num_rays_per_chunk=1 << 15),
        model_ignored={},
        optimizers={
            "proposal_networks": dict(algorithm="adam", lr=1e-2, eps=1e-15, scheduler=_exp_decay(1e-4, 200000)),
            "fields": dict(algorithm="adam", lr=1e-2, eps=1e-15, scheduler=_exp_decay(1e-4, 200000)),
        }),
    "fruit_nerf_big": dict(
        description="Base config for FruitNeRF-Big",
        trainer=dict(steps_per_eval_batch=500, steps_per_save=2000, max_num_iterations=100000, mixed_precision=True,
                     viewer_num_rays_per_chunk=1 << 15),
        datamanager=dict(train_num_rays_per_batch=4096 * 2, eval_num_rays_per_batch=4096,
                         train_num_images_to_sample_from=200, train_num_times_to_repeat_images=1000,
                         dataparser=dict(train_split_fraction=0.99)),
        camera_optimizer=dict(mode="SO3xR3", algorithm="radam", lr=6e-4, eps=1e-8, weight_decay=1e-3, scheduler=None),
        model=dict(eval_num_rays_per_chunk=1 << 15, num_nerf_samples_per_ray=128,
                   num_proposal_samples_per_ray=(512, 256), geo_feat_dim=30, hidden_dim_semantics=128,
                   num_layers_semantic=3, max_res=4096, proposal_weights_anneal_max_num_iters=5000,
                   log2_hashmap_size=21),
        model_ignored=dict(hidden_dim=128, hidden_dim_color=128, appearance_embed_dim=128),
        optimizers={
            "proposal_networks": dict(algorithm="radam", lr=1e-2, eps=1e-15, scheduler=None),
            "fields": dict(algorithm="radam", lr=1e-2, eps=1e-15, scheduler=_exp_decay(1e-4, 50000)),
        }),
    "fruit_nerf_huge": dict(
        description="Base config for FruitNeRF-Huge",
        trainer=dict(steps_per_eval_batch=500, steps_per_save=2000, max_num_iterations=100000, mixed_precision=True,
                     viewer_num_rays_per_chunk=1 << 15),
        datamanager=dict(train_num_rays_per_batch=4096 * 4, eval_num_rays_per_batch=4096),
        camera_optimizer=dict(mode="SO3xR3", algorithm="radam", lr=6e-4, eps=1e-8, weight_decay=1e-3,
                              scheduler=_exp_decay(6e-5, 50000)),
        model=dict(eval_num_rays_per_chunk=1 << 15, num_nerf_samples_per_ray=64,
                   num_proposal_samples_per_ray=(512, 512),
                   proposal_net_args_list=[
                       {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 512, "use_linear": False},
                       {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 7, "max_res": 2048, "use_linear": False}],
                   geo_feat_dim=30, hidden_dim_semantics=128, num_layers_semantic=3, max_res=8192,
                   proposal_weights_anneal_max_num_iters=5000, log2_hashmap_size=21),
        model_ignored=dict(hidden_dim=256, hidden_dim_color=256, appearance_embed_dim=32),
        optimizers={
            "proposal_networks": dict(algorithm="radam", lr=1e-2, eps=1e-15, scheduler=None),
            "fields": dict(algorithm="radam", lr=1e-2, eps=1e-15, scheduler=_exp_decay(1e-4, 50000)),
        }),
}

ENTRY_POINTS = {"fruit_nerf_method": "fruit_nerf", "fruit_nerf_method_big": "fruit_nerf_big",
                "fruit_nerf_method_huge": "fruit_nerf_huge"}


def model_config(method: str, **overrides):
    """FruitNerfModelConfig of a method (hot-path fields only; `overrides` on top, e.g. mlp_precision)."""
    from .fruit_nerf import FruitNerfModelConfig
    kw = copy.deepcopy(METHODS[method]["model"])
    kw.update(overrides)
    return FruitNerfModelConfig(**kw)


def group_schedules(method: str) -> Dict[str, dict]:
    """FusedAdam's `group_lr` table of a method: {group: {lr, lr_final | None, max_steps | None}}."""
    out = {}
    for name, o in METHODS[method]["optimizers"].items():
        s = o.get("scheduler")
        out[name] = dict(lr=o["lr"], lr_final=None if s is None else s["lr_final"],
                         max_steps=None if s is None else s["max_steps"])
    return out


def camera_optimizer_config(method: str, mode: Optional[str] = None):
    """cameras.camera_optimizers.CameraOptimizerConfig of a method (+ its optimiser algorithm)."""
    from .cameras.camera_optimizers import CameraOptimizerConfig
    c = METHODS[method]["camera_optimizer"]
    s = c.get("scheduler")
    cfg = CameraOptimizerConfig(mode=c["mode"] if mode is None else mode, lr=c["lr"], eps=c["eps"],
                                weight_decay=c["weight_decay"], lr_final=None if s is None else s["lr_final"],
                                max_steps=1 if s is None else s["max_steps"])
    return cfg, c["algorithm"]


def training_setup(method: str, model, num_cameras: Optional[int] = None, camera_mode: Optional[str] = None):
    """The optimisers of `method` for this repo's training loop -> (FusedAdam, (CameraOptimizer, CameraAdam) | None).
    Both groups of a method share one algorithm and eps (fruit_nerf_config.py:47-56,97-106,148-160)."""
    from .training import FusedAdam
    opts = METHODS[method]["optimizers"]
    algos = {o["algorithm"] for o in opts.values()}
    eps = {o["eps"] for o in opts.values()}
    assert len(algos) == 1 and len(eps) == 1, "one optimiser family per method"
    optimizer = FusedAdam(model, eps=eps.pop(), algorithm=algos.pop(), group_lr=group_schedules(method))
    camera = None
    cfg, algorithm = camera_optimizer_config(method, camera_mode)
    if cfg.mode != "off":
        from .cameras.camera_optimizers import CameraAdam
        cam = cfg.setup(num_cameras if num_cameras is not None else model.num_train_data, model.device)
        camera = (cam, CameraAdam(cam, algorithm=algorithm))
    return optimizer, camera


# ---- entry-point objects --------------------------------------------------------------------------------------------


@dataclass
class MethodSpec:
    """Stand-in for nerfstudio.plugins.types.MethodSpecification when Nerfstudio is not installed: `config` holds the
    method's table (a deep copy of METHODS[name] + `method_name` + the model config object), `description` the text."""
    config: Dict[str, Any] = field(default_factory=dict)
    description: str = ""


def _nerfstudio_spec(name: str):
    """The real thing: a TrainerConfig whose pipeline builds this package's FruitModel.  Raises ImportError without
    Nerfstudio.  The datamanager / dataparser / pipeline are data loading and stay the reference's own classes
    (`fruit_nerf.*`, the GPL package this plugin sits next to) when importable, Nerfstudio's vanilla ones otherwise."""
    from nerfstudio.cameras.camera_optimizers import CameraOptimizerConfig
    from nerfstudio.configs.base_config import ViewerConfig
    from nerfstudio.engine.optimizers import AdamOptimizerConfig, RAdamOptimizerConfig
    from nerfstudio.engine.schedulers import ExponentialDecaySchedulerConfig
    from nerfstudio.engine.trainer import TrainerConfig
    from nerfstudio.plugins.types import MethodSpecification
    try:
        from fruit_nerf.data.fruit_datamanager import FruitDataManagerConfig as DataManagerConfig
        from fruit_nerf.data.fruitnerf_dataparser import FruitNerfDataParserConfig as DataParserConfig
        from fruit_nerf.fruit_pipeline import FruitPipelineConfig as PipelineConfig
    except ImportError:
        from nerfstudio.data.datamanagers.base_datamanager import VanillaDataManagerConfig as DataManagerConfig
        from nerfstudio.data.dataparsers.nerfstudio_dataparser import NerfstudioDataParserConfig as DataParserConfig
        from nerfstudio.pipelines.base_pipeline import VanillaPipelineConfig as PipelineConfig
    M = METHODS[name]

    def optimizer(o):
        cls = AdamOptimizerConfig if o["algorithm"] == "adam" else RAdamOptimizerConfig
        kw = dict(lr=o["lr"], eps=o["eps"])
        if o.get("weight_decay"):
            kw["weight_decay"] = o["weight_decay"]
        return cls(**kw)

    def scheduler(s):
        return None if s is None else ExponentialDecaySchedulerConfig(lr_final=s["lr_final"], max_steps=s["max_steps"])

    cam = M["camera_optimizer"]
    cam_kw = dict(mode=cam["mode"], optimizer=optimizer(cam))
    if cam.get("scheduler") is not None:
        cam_kw["scheduler"] = scheduler(cam["scheduler"])
    dm = dict(M["datamanager"])
    parser_kw = dm.pop("dataparser", {})
    t = M["trainer"]
    return MethodSpecification(
        config=TrainerConfig(
            method_name=name, steps_per_eval_batch=t["steps_per_eval_batch"], steps_per_save=t["steps_per_save"],
            max_num_iterations=t["max_num_iterations"], mixed_precision=t["mixed_precision"],
            pipeline=PipelineConfig(
                datamanager=DataManagerConfig(dataparser=DataParserConfig(**parser_kw),
                                              camera_optimizer=CameraOptimizerConfig(**cam_kw), **dm),
                model=model_config(name)),
            optimizers={g: {"optimizer": optimizer(o), "scheduler": scheduler(o.get("scheduler"))}
                        for g, o in M["optimizers"].items()},
            viewer=ViewerConfig(num_rays_per_chunk=t["viewer_num_rays_per_chunk"]), vis="viewer"),
        description=M["description"])


def method_specification(name: str):
    """The entry-point object of method `name` (MethodSpecification inside Nerfstudio, MethodSpec otherwise)."""
    try:
        return _nerfstudio_spec(name)
    except ImportError:
        cfg = copy.deepcopy(METHODS[name])
        cfg["method_name"] = name
        cfg["model_config"] = model_config(name)
        return MethodSpec(config=cfg, description=cfg["description"])


_built: Dict[str, Any] = {}


def __getattr__(attr: str):   # PEP 562: the three entry-point names, built on first access
    if attr in ENTRY_POINTS:
        if attr not in _built:
            _built[attr] = method_specification(ENTRY_POINTS[attr])
        return _built[attr]
    raise AttributeError(f"module {__name__!r} has no attribute {attr!r}")
