"""CPU: the C-ABI library loads, exports every symbol include/fruitnerf_hip.h declares, and rejects bad
arguments / missing devices loudly (no compute without a GPU)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "fruitnerf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fnr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from fruitnerf_amd import _lib as L
    lib = L.load()
    declared = _declared_symbols()
    assert len(declared) >= 12
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/fruitnerf_hip.h but not exported"
        assert name in L.SIGNATURES, f"{name} has no ctypes signature in fruitnerf_amd/_lib.py"
    assert set(L.SIGNATURES) == set(declared)
    assert lib.fnr_abi_version() == L.ABI_VERSION == 13


def test_struct_layouts_match_header_sizes():
    from fruitnerf_amd import _lib as L
    assert C.sizeof(L.fnr_grid) == 4 + 4 + 16 * 4 + 8
    assert C.sizeof(L.fnr_rays) == 8 + 5 * 8
    assert C.sizeof(L.fnr_warp) == 4 + 6 * 4
    assert C.sizeof(L.fnr_lattice) == 16 + 3 * 8  # 3 ints padded to 16
    assert C.sizeof(L.fnr_prop_net) == C.sizeof(L.fnr_grid) + 8 + 4 * 8
    assert C.sizeof(L.fnr_field_net) == C.sizeof(L.fnr_grid) + 8 * 4 + (4 + 8 + 2 + 6 + 1) * 8 + 8  # + mlp_mode (padded)
    assert C.sizeof(L.fnr_table_adam) == 24 + 8 + 8 + 4 * 8   # int + 4 floats + slot, step, 2 floats, 4 pointers
    assert L.fnr_table_adam.slot.offset == 20 and L.fnr_table_adam.step.offset == 24
    assert C.sizeof(L.fnr_step_scalars) == 8 + 4 + 4 + L.FNR_PROGRAM_ADAM_SLOTS * 16 + 8
    assert L.fnr_step_scalars.losses.offset == 16 + L.FNR_PROGRAM_ADAM_SLOTS * 16
    assert C.sizeof(L.fnr_adam_span) == 3 * 8 + 4 + 4
    hdr = open(os.path.join(ROOT, "include", "fruitnerf_hip.h")).read()
    for name in ("FNR_MAX_ADAM_SPANS", "FNR_MAX_PROPOSAL_LEVELS", "FNR_LOSS_SLOTS", "FNR_MAX_POSITION_SOURCES",
                 "FNR_TRAIN_PROLOGUE_MAX_JITTER", "FNR_PROGRAM_ADAM_SLOTS"):
        assert int(re.search(r"#define %s (\d+)" % name, hdr).group(1)) == getattr(L, name)
    assert L.FNR_TRAIN_LOSSES_ACCUM_FLOATS == 4 * L.FNR_LOSS_SLOTS + 33 * 32
    assert "#define FNR_TRAIN_LOSSES_ACCUM_FLOATS (4 * FNR_LOSS_SLOTS + 33 * 32)" in hdr


def test_invalid_arguments_are_rejected_without_a_gpu():
    from fruitnerf_amd import _lib as L
    lib = L.load()
    rays = L.fnr_rays(0, None, None, None, None, None)
    rc = lib.fnr_sample_spaced(C.byref(rays), 1, 16, None, None, 0, None, None, None)
    assert rc == -1 and b"null" in lib.fnr_last_error()
    rc = lib.fnr_composite_fwd(C.byref(rays), 0, None, None, None, None, 0, None, None, None, None, None, None, None)
    assert rc == -1
    # the fused entry points of round 2 check their descriptors on the host, before any launch
    rc = lib.fnr_adam_step_spans(1, 1, 1, 1, 0, None, 0, 0.9, 0.999, 1e-8, 1.0, 0.0, 1, None)
    assert rc == -1 and b"adam_step_spans" in lib.fnr_last_error()
    span = (L.fnr_adam_span * 1)(L.fnr_adam_span(2, 8, 1, 1e-2, 0))          # offset not a multiple of 4
    rc = lib.fnr_adam_step_spans(1, 1, 1, 1, 1, span, 0, 0.9, 0.999, 1e-8, 1.0, 0.0, 1, None)
    assert rc == -1 and b"multiples of 4" in lib.fnr_last_error()
    rc = lib.fnr_train_losses(16, None, None, None, None, 1.0, None, None, 48, None, None, 0, None, None, None, None, None,
                              None, None, 1.0, 1, None, None, None)
    assert rc == -1 and b"train_losses" in lib.fnr_last_error()
    adam = L.table_adam(2, 1e-2, 0.9, 0.999, 1e-8, 1, 1.0, 0.0, 1, 1, 1, None)        # algorithm 2 does not exist
    grid = L.fnr_grid()
    grid.n_levels, grid.log2_hashmap_size = 16, 19
    rc = lib.fnr_hash_encode_bwd_adam(C.byref(grid), C.byref(L.fnr_warp()), C.byref(rays), 1, 4, 1, 1, 1 << 30, 0,
                                      C.byref(adam), None)
    assert rc == -1 and b"algorithm" in lib.fnr_last_error()
    net = L.fnr_field_net()
    net.grid.n_levels = 8  # not the built configuration
    rc = lib.fnr_field_mlp_fwd(C.byref(net), C.byref(rays), 4, 1, None, 1, 1, 1, 1, None, None, None, 1, 1 << 20, None)
    assert rc in (-1, -2)


def test_no_cpu_fallback():
    import torch
    from fruitnerf_amd import _lib as L
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no HIP device|no CPU"):
        L.device_check()
    from fruitnerf_amd import _kernels as K
    with pytest.raises(RuntimeError, match="no CPU path"):
        K.RaysArg(torch.zeros(4, 3), torch.zeros(4, 3), None, None)


def test_step_program_records_replays_and_refuses_without_a_gpu():
    """The program API of ABI 12 on operations that need no device: a stream dependency of a stream on itself is recorded
    and replayed (it enqueues nothing); an entry point without a recording hook poisons the recording — fnr_program_end
    fails, names it and leaves the program empty; replaying while recording is refused."""
    from fruitnerf_amd import _lib as L
    lib = L.load()
    h = C.c_void_p()
    assert lib.fnr_program_create(C.byref(h)) == 0 and h.value
    try:
        assert lib.fnr_program_size(h) == 0
        assert lib.fnr_program_begin(h) == 0
        assert lib.fnr_program_begin(h) == -1 and b"already recording" in lib.fnr_last_error()
        assert lib.fnr_stream_wait_stream(None, None) == 0
        assert lib.fnr_stream_wait_stream(None, None) == 0
        assert lib.fnr_program_replay(h, None) == -1           # not while it is being recorded
        assert lib.fnr_program_end(h) == 0
        assert lib.fnr_program_size(h) == 2
        assert lib.fnr_program_op_name(h, 0) == b"fnr_stream_wait_stream" and lib.fnr_program_op_name(h, 2) is None
        sc = L.fnr_step_scalars()
        assert lib.fnr_program_replay(h, C.byref(sc)) == 0 and lib.fnr_program_replay(h, None) == 0
        assert lib.fnr_stream_wait_stream(None, None) == 0     # outside a recording: not appended
        assert lib.fnr_program_size(h) == 2
        # an entry point that cannot be replayed runs while recording (its own arguments are bad too: nothing is launched)
        assert lib.fnr_program_begin(h) == 0
        rays = L.fnr_rays(0, None, None, None, None, None)
        assert lib.fnr_sample_spaced(C.byref(rays), 1, 16, None, None, 0, None, None, None) == -1
        assert lib.fnr_program_end(h) == -2 and b"fnr_sample_spaced" in lib.fnr_last_error()
        assert lib.fnr_program_size(h) == 0
        # abort leaves nothing behind and frees the thread for the next recording
        assert lib.fnr_program_begin(h) == 0 and lib.fnr_stream_wait_stream(None, None) == 0
        assert lib.fnr_program_abort(h) == 0 and lib.fnr_program_size(h) == 0
        assert lib.fnr_program_begin(h) == 0 and lib.fnr_program_end(h) == 0
    finally:
        assert lib.fnr_program_destroy(h) == 0


def test_call_log_names_what_python_asked_for():
    from fruitnerf_amd import _lib as L
    L.load()
    keep = []
    log = L.begin_call_log(keep=keep)
    try:
        import torch
        t = torch.zeros(4)
        assert L.ptr(t) == t.data_ptr() and keep == [t]
        assert L.load().fnr_abi_version() == L.ABI_VERSION
        L.load().fnr_stream_wait_stream(None, None)
    finally:
        names = L.end_call_log(log)
    assert names == ["fnr_abi_version", "fnr_stream_wait_stream"]
    assert not isinstance(L.load(), L._CallLog) and L._ptr_keep is None
