"""ctypes binding of include/fruitnerf_hip.h (libfruitnerf_hip.so, gfx950).

There is NO CPU fallback: if the shared library is missing or a call fails, a RuntimeError carrying
`fnr_last_error()` is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FNR_LIB_PATH") or os.path.join(_HERE, "lib", "libfruitnerf_hip.so")

FNR_MAX_LEVELS = 16
FNR_MAX_SEM_LAYERS = 4
FNR_LOSS_SLOTS = 1024
ABI_VERSION = 13     # include/fruitnerf_hip.h: FNR_ABI_VERSION

c_float_p = C.POINTER(C.c_float)


class fnr_grid(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("log2_hashmap_size", C.c_int32),
                ("scalings", C.c_int32 * FNR_MAX_LEVELS), ("table", C.c_void_p)]


class fnr_rays(C.Structure):
    _fields_ = [("n_rays", C.c_int64), ("origins", C.c_void_p), ("directions", C.c_void_p),
                ("nears", C.c_void_p), ("fars", C.c_void_p), ("camera_indices", C.c_void_p)]


class fnr_warp(C.Structure):
    _fields_ = [("mode", C.c_int32), ("aabb", C.c_float * 6)]


class fnr_prop_net(C.Structure):
    _fields_ = [("grid", fnr_grid), ("hidden_dim", C.c_int32), ("w0", C.c_void_p), ("b0", C.c_void_p),
                ("w1", C.c_void_p), ("b1", C.c_void_p)]


class fnr_field_net(C.Structure):
    _fields_ = [("grid", fnr_grid), ("geo_feat_dim", C.c_int32), ("hidden_dim", C.c_int32),
                ("hidden_dim_color", C.c_int32), ("hidden_dim_semantics", C.c_int32),
                ("num_layers_semantic", C.c_int32), ("semantic_out_dim", C.c_int32),
                ("appearance_dim", C.c_int32), ("n_images", C.c_int32),
                ("base_w0", C.c_void_p), ("base_b0", C.c_void_p), ("base_w1", C.c_void_p), ("base_b1", C.c_void_p),
                ("sem_w", C.c_void_p * FNR_MAX_SEM_LAYERS), ("sem_b", C.c_void_p * FNR_MAX_SEM_LAYERS),
                ("head_w", C.c_void_p), ("head_b", C.c_void_p),
                ("col_w", C.c_void_p * 3), ("col_b", C.c_void_p * 3), ("embedding", C.c_void_p),
                ("mlp_mode", C.c_int32)]


class fnr_table_adam(C.Structure):
    # slot: which fnr_step_scalars.adam entry supplies lr / step when a recorded step program replays the call (0: none)
    _fields_ = [("algorithm", C.c_int32), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("eps", C.c_float), ("slot", C.c_int32), ("step", C.c_int64), ("grad_scale", C.c_float),
                ("weight_decay", C.c_float), ("params", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("touched", C.c_void_p)]


def table_adam(algorithm: int, lr: float, beta1: float, beta2: float, eps: float, step: int, grad_scale: float,
               weight_decay: float, params, exp_avg, exp_avg_sq, touched, slot: int = 0) -> "fnr_table_adam":
    return fnr_table_adam(algorithm, lr, beta1, beta2, eps, slot, step, grad_scale, weight_decay, params, exp_avg,
                          exp_avg_sq, touched)


FNR_PROGRAM_ADAM_SLOTS = 4
ADAM_SLOTS = {"fields": 1, "proposal_networks": 2, "camera_opt": 3}     # optimiser group -> fnr_table_adam.slot


class fnr_adam_scalars(C.Structure):
    _fields_ = [("lr", C.c_float), ("reserved", C.c_int32), ("step", C.c_int64)]


class fnr_step_scalars(C.Structure):
    _fields_ = [("prologue_offset", C.c_uint64), ("anneal", C.c_float), ("reserved", C.c_int32),
                ("adam", fnr_adam_scalars * FNR_PROGRAM_ADAM_SLOTS), ("losses", C.c_void_p)]


class fnr_adam_span(C.Structure):
    _fields_ = [("offset", C.c_int64), ("count", C.c_int64), ("step", C.c_int64), ("lr", C.c_float),
                ("reserved", C.c_int32)]


FNR_MAX_ADAM_SPANS = 8
FNR_MAX_PROPOSAL_LEVELS = 4
FNR_MAX_POSITION_SOURCES = 4         # fnr_position_grad_reduce_multi
FNR_TRAIN_PROLOGUE_MAX_JITTER = 5    # fnr_train_prologue: n_jitter in 1..5
FNR_TRAIN_LOSSES_ACCUM_FLOATS = 4 * FNR_LOSS_SLOTS + 33 * 32


class fnr_lattice(C.Structure):
    _fields_ = [("n_x", C.c_int32), ("n_y", C.c_int32), ("n_z", C.c_int32),
                ("xs", C.c_void_p), ("ys", C.c_void_p), ("zs", C.c_void_p)]


class fnr_image_set(C.Structure):
    _fields_ = [("n_images", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("images", C.c_void_p),
                ("masks", C.c_void_p), ("c2w", C.c_void_p), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float)]


P = C.POINTER
_vp = C.c_void_p
_i = C.c_int
_i64 = C.c_int64
_f = C.c_float

# name -> (restype, argtypes); every symbol include/fruitnerf_hip.h declares
SIGNATURES = {
    "fnr_abi_version": (_i, []),
    "fnr_last_error": (C.c_char_p, []),
    "fnr_device_check": (_i, [P(C.c_int), C.c_char_p, _i]),
    "fnr_profile_enable": (_i, [_i, C.c_uint64]),
    "fnr_profile_pause": (_i, [_i]),
    "fnr_profile_collect": (_i64, [P(C.c_int32), P(C.c_int64), P(C.c_float), _i64]),
    "fnr_debug_scatter_overflows": (_i, [P(C.c_uint64), _i]),
    "fnr_debug_scatter_records": (_i, [P(C.c_uint64), _i]),
    "fnr_sample_pixels": (_i, [P(fnr_image_set), _vp, _i, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fnr_train_prologue": (_i, [P(fnr_image_set), _vp, _i, _i64, C.c_uint64, C.c_uint64, _vp, _vp, _vp, _vp, _i, _vp, _vp,
                                _vp, _vp, _vp, _f, _f, _i, _i, _vp, _vp, _vp, _vp]),
    "fnr_sample_spaced": (_i, [P(fnr_rays), _i, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "fnr_weights_pdf": (_i, [P(fnr_rays), _i, _i, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fnr_prop_density_fwd": (_i, [P(fnr_prop_net), P(fnr_warp), P(fnr_rays), _vp, _i, _vp, _vp, _vp]),
    "fnr_hash_encode_fwd": (_i, [P(fnr_grid), P(fnr_warp), P(fnr_rays), _vp, _i, _vp, _vp, _vp, _vp]),
    "fnr_position_grad_from_jacobian": (_i, [P(fnr_warp), P(fnr_rays), _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "fnr_hash_encode_lattice": (_i, [P(fnr_grid), P(fnr_warp), P(fnr_lattice), _i64, _i64, _vp, _vp, _vp]),
    "fnr_field_mlp_fwd_workspace_bytes": (C.c_size_t, [_i64]),
    "fnr_field_h_dim": (_i, [P(fnr_field_net)]),
    "fnr_field_mlp_fwd": (_i, [P(fnr_field_net), P(fnr_rays), _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                               C.c_size_t, _vp]),
    "fnr_embedding_mean": (_i, [_vp, _i, _i, _vp, _vp]),
    "fnr_composite_fwd": (_i, [P(fnr_rays), _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fnr_losses_fwd": (_i, [_i64, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    "fnr_interlevel_fwd": (_i, [_i64, _i, _vp, _vp, _i, _vp, _vp, _f, _vp, _vp, _vp]),
    "fnr_distortion": (_i, [_i64, _i, _vp, _vp, _vp, _vp]),
    "fnr_composite_bwd": (_i, [P(fnr_rays), _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fnr_composite_fwd_bwd_targets": (_i, [P(fnr_rays), _i, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp,
                                           _vp, _vp, _vp, _vp]),
    "fnr_composite_bwd_targets": (_i, [P(fnr_rays), _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp,
                                       _vp]),
    "fnr_weights_bwd": (_i, [_i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fnr_field_mlp_bwd_workspace_bytes": (C.c_size_t, [_i64, _i]),
    "fnr_field_mlp_bwd": (_i, [P(fnr_field_net), P(fnr_field_net), P(fnr_rays), _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                               _vp, _vp, _vp, C.c_size_t, _vp]),
    "fnr_field_mlp_bwd_rays": (_i, [P(fnr_field_net), P(fnr_field_net), P(fnr_rays), _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                    _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "fnr_field_mlp_bwd_adam": (_i, [P(fnr_field_net), P(fnr_field_net), P(fnr_rays), _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                    _vp, _vp, _vp, _vp, P(fnr_table_adam), _vp, _vp, C.c_size_t, _vp]),
    "fnr_hash_scatter_workspace_bytes": (C.c_size_t, [_i64, _i, _i]),
    "fnr_hash_encode_bwd": (_i, [P(fnr_grid), P(fnr_warp), P(fnr_rays), _vp, _i, _vp, _i, _i, _vp, C.c_size_t, _i, _vp]),
    "fnr_hash_encode_bwd_adam": (_i, [P(fnr_grid), P(fnr_warp), P(fnr_rays), _vp, _i, _vp, _vp, C.c_size_t, _i,
                                      P(fnr_table_adam), _vp]),
    "fnr_prop_density_bwd_workspace_bytes": (C.c_size_t, [_i64, _i, _i]),
    "fnr_prop_density_bwd": (_i, [P(fnr_prop_net), P(fnr_prop_net), P(fnr_warp), P(fnr_rays), _vp, _i, _vp, _vp, _vp,
                                  _vp, C.c_size_t, _i, _vp]),
    "fnr_prop_density_bwd_adam": (_i, [P(fnr_prop_net), P(fnr_prop_net), P(fnr_warp), P(fnr_rays), _vp, _i, _vp, _vp, _vp,
                                       P(fnr_table_adam), P(fnr_table_adam), _vp, _vp, C.c_size_t, _i, _vp]),
    "fnr_prop_density_bwd_pair": (_i, [P(P(fnr_prop_net)), P(P(fnr_prop_net)), P(P(fnr_warp)), P(fnr_rays), P(C.c_void_p),
                                       P(C.c_int), P(C.c_void_p), P(C.c_void_p), P(C.c_void_p), P(P(fnr_table_adam)),
                                       P(fnr_table_adam), _vp, P(C.c_void_p), P(C.c_size_t), P(C.c_int), _vp]),
    "fnr_prop_density_bwd_pair_split": (_i, [P(P(fnr_prop_net)), P(P(fnr_prop_net)), P(P(fnr_warp)), P(fnr_rays), P(C.c_void_p),
                                             P(C.c_int), P(C.c_void_p), P(C.c_void_p), P(C.c_void_p), P(P(fnr_table_adam)),
                                             P(fnr_table_adam), _vp, P(C.c_void_p), P(C.c_size_t), P(C.c_int), _vp, _vp]),
    "fnr_hash_encode_input_grad": (_i, [P(fnr_grid), P(fnr_warp), P(fnr_rays), _vp, _i, _vp, _vp, _vp]),
    "fnr_position_grad_reduce": (_i, [P(fnr_warp), P(fnr_rays), _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "fnr_position_grad_reduce_multi": (_i, [_i, P(P(fnr_warp)), P(fnr_rays), P(C.c_void_p), P(C.c_int), P(C.c_int),
                                            P(C.c_void_p), _i, _vp, _vp, _vp]),
    "fnr_train_losses": (_i, [_i64, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _vp, _vp, _i, P(C.c_int), P(C.c_void_p),
                              P(C.c_void_p), P(C.c_void_p), P(C.c_void_p), P(C.c_void_p), P(C.c_void_p), _f, _i, _vp,
                              _vp, _vp]),
    "fnr_adam_step": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _i64, _f, _f, _i, _vp]),
    "fnr_radam_step": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _i64, _f, _f, _i, _vp]),
    "fnr_adam_step_spans": (_i, [_vp, _vp, _vp, _vp, _i, P(fnr_adam_span), _i, _f, _f, _f, _f, _f, _i, _vp]),
    "fnr_cloud_workspace_bytes": (C.c_size_t, [_i64]),
    "fnr_cloud_bounds": (_i, [_vp, _i64, _vp, _vp, C.c_size_t, _vp]),
    "fnr_cloud_radius_count": (_i, [_vp, _i64, P(C.c_double), P(C.c_double), C.c_double, _i, _vp, _vp, C.c_size_t,
                                    _vp]),
    "fnr_cloud_dbscan": (_i, [_vp, _i64, P(C.c_double), P(C.c_double), C.c_double, C.c_int32, _vp, _vp, _vp,
                              C.c_size_t, _vp]),
    "fnr_cloud_voxel_down_sample": (_i, [_vp, _vp, _i64, P(C.c_double), P(C.c_double), C.c_double, _vp, _vp, _vp, _vp,
                                         C.c_size_t, _vp]),
    "fnr_camera_adjust": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "fnr_camera_pose_grad": (_i, [P(fnr_image_set), _vp, _i, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fnr_camera_pose_grad_adam": (_i, [P(fnr_image_set), _vp, _i, _i64, _vp, _vp, _vp, _vp, _vp, _vp, P(fnr_table_adam),
                                       _vp]),
    "fnr_image_metrics_workspace_bytes": (C.c_size_t, [_i, _i]),
    "fnr_image_metrics": (_i, [_i, _i, _vp, _vp, _vp, _vp, P(C.c_float), _vp, _vp, C.c_size_t, _vp]),
    "fnr_program_create": (_i, [P(_vp)]),
    "fnr_program_destroy": (_i, [_vp]),
    "fnr_program_begin": (_i, [_vp]),
    "fnr_program_end": (_i, [_vp]),
    "fnr_program_abort": (_i, [_vp]),
    "fnr_program_size": (_i64, [_vp]),
    "fnr_program_op_name": (C.c_char_p, [_vp, _i64]),
    "fnr_program_replay": (_i, [_vp, P(fnr_step_scalars)]),
    "fnr_event_create": (_i, [P(_vp)]),
    "fnr_event_destroy": (_i, [_vp]),
    "fnr_event_record": (_i, [_vp, _vp]),
    "fnr_stream_wait_event": (_i, [_vp, _vp]),
    "fnr_stream_wait_stream": (_i, [_vp, _vp]),
    "fnr_export_workspace_bytes": (C.c_size_t, [_i64]),
    "fnr_export_compact": (_i, [P(fnr_lattice), _i64, _i64, _vp, _i64, _vp, _vp, _vp, P(_vp), P(_vp), _i64, _vp, _vp,
                                _vp]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libfruitnerf_hip.so and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). fruitnerf_amd has no CPU or PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError -> loud failure on a stale library
        fn.restype = res
        fn.argtypes = args
    if lib.fnr_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libfruitnerf_hip.so ABI {lib.fnr_abi_version()} != {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def last_error() -> str:
    return load().fnr_last_error().decode()


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise RuntimeError(f"fruitnerf_hip {what} failed (rc={rc}): {last_error()}")


_ptr_keep: Optional[list] = None     # while a step program is being recorded: every tensor whose pointer is handed out


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "fruitnerf_hip needs contiguous tensors"
    if _ptr_keep is not None:
        _ptr_keep.append(t)          # a recorded program replays this pointer: its tensor must outlive the program
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device: torch.device) -> Optional[int]:
    """The HIP stream PyTorch currently enqueues on for `device` (raw handle: building a torch.cuda.Stream object per
    launch cost ~4 us x 25 launches of host time per training step)."""
    if _raw_stream is not None:
        idx = device.index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream


def device_check() -> dict:
    lib = load()
    n = C.c_int(0)
    buf = C.create_string_buffer(64)
    check(lib.fnr_device_check(C.byref(n), buf, 64), "device_check")
    return {"arch": buf.value.decode(), "cus": n.value}


def require_gpu_tensor(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}; fruitnerf_amd runs only on a HIP device (no CPU path)")


PROFILE_OPS = ["sample_spaced", "weights_pdf", "prop_density_fwd", "hash_encode_fwd", "hash_encode_lattice",
               "field_mlp_fwd", "composite_fwd", "losses_fwd", "interlevel_fwd", "distortion", "composite_bwd",
               "weights_bwd", "field_mlp_bwd", "hash_encode_bwd", "prop_density_bwd", "adam_step", "export_compact",
               "position_grad", "cloud"]


_profile_state = {"on": False, "paused": False}


def profile_enable(on: bool, ops=None) -> None:
    mask = (1 << 64) - 1 if ops is None else sum(1 << PROFILE_OPS.index(o) for o in ops)
    check(load().fnr_profile_enable(1 if on else 0, mask), "profile_enable")
    _profile_state["on"], _profile_state["paused"] = bool(on), False


def profile_pause(paused: bool) -> None:
    check(load().fnr_profile_pause(1 if paused else 0), "profile_pause")
    _profile_state["paused"] = bool(paused)


def profile_recording() -> bool:
    """True while entry points bracket their launches with HIP events (a step program recorded or replayed then would
    carry / drop those events: training.TrainingSteps takes the interpreted path on such steps)."""
    return _profile_state["on"] and not _profile_state["paused"]


class _CallLog:
    """Stands in for the loaded library while a step program is being recorded: every entry point called from Python is
    noted by name, so that the recording can be checked against what Python asked for (a call the C side did not record
    would silently be missing from every replay)."""

    def __init__(self, lib):
        self._lib, self.names = lib, []

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        names = self.names

        def logged(*a):
            names.append(name)
            return fn(*a)
        return logged


def begin_call_log(keep: Optional[list] = None) -> "_CallLog":
    """keep: receives every tensor whose device pointer ptr() hands out while the log is active."""
    global _lib, _ptr_keep
    lib = load()
    if isinstance(lib, _CallLog):
        raise RuntimeError("a call log is already active")
    _lib = _CallLog(lib)
    _ptr_keep = keep
    return _lib


def end_call_log(log: "_CallLog") -> list:
    global _lib, _ptr_keep
    if _lib is log:
        _lib = log._lib
    _ptr_keep = None
    return log.names


def scatter_overflows(reset: bool = False) -> int:
    """Records of the binned scatter that overflowed their bin's queue since the last reset (fnr_debug_scatter_overflows)."""
    n = C.c_uint64(0)
    check(load().fnr_debug_scatter_overflows(C.byref(n), 1 if reset else 0), "debug_scatter_overflows")
    return int(n.value)


def scatter_records(reset: bool = False):
    """(records summed into the field's table, into proposal tables) since the last reset (fnr_debug_scatter_records)."""
    n = (C.c_uint64 * 2)()
    check(load().fnr_debug_scatter_records(n, 1 if reset else 0), "debug_scatter_records")
    return int(n[0]), int(n[1])


def profile_collect(capacity: int = 1 << 20):
    """-> list of (op name, units, milliseconds) for every profiled entry-point call since enable()."""
    ops = (C.c_int32 * capacity)()
    units = (C.c_int64 * capacity)()
    ms = (C.c_float * capacity)()
    n = load().fnr_profile_collect(ops, units, ms, capacity)
    return [(PROFILE_OPS[ops[i]], units[i], ms[i]) for i in range(n)]
