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
    assert lib.fnr_abi_version() == 5


def test_struct_layouts_match_header_sizes():
    from fruitnerf_amd import _lib as L
    assert C.sizeof(L.fnr_grid) == 4 + 4 + 16 * 4 + 8
    assert C.sizeof(L.fnr_rays) == 8 + 5 * 8
    assert C.sizeof(L.fnr_warp) == 4 + 6 * 4
    assert C.sizeof(L.fnr_lattice) == 16 + 3 * 8  # 3 ints padded to 16
    assert C.sizeof(L.fnr_prop_net) == C.sizeof(L.fnr_grid) + 8 + 4 * 8
    assert C.sizeof(L.fnr_field_net) == C.sizeof(L.fnr_grid) + 8 * 4 + (4 + 8 + 2 + 6 + 1) * 8 + 8  # + mlp_mode (padded)


def test_invalid_arguments_are_rejected_without_a_gpu():
    from fruitnerf_amd import _lib as L
    lib = L.load()
    rays = L.fnr_rays(0, None, None, None, None, None)
    rc = lib.fnr_sample_spaced(C.byref(rays), 1, 16, None, None, 0, None, None, None)
    assert rc == -1 and b"null" in lib.fnr_last_error()
    rc = lib.fnr_composite_fwd(C.byref(rays), 0, None, None, None, None, 0, None, None, None, None, None, None, None)
    assert rc == -1
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
