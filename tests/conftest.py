import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` through gpurun)")
    # the oracle side of the parity tests is PyTorch-CPU: size its thread pool by the container's CPU quota (16 CPUs of
    # 256 hardware threads on the MI355X boxes), or the CFS controller freezes the process every 100 ms
    from fruitnerf_amd.hostinfo import usable_cpus
    torch.set_num_threads(usable_cpus())


def pytest_sessionstart(session):
    """The suites bind the C ABI: (re)build libfruitnerf_hip.so when it is missing or older than its sources
    (`make` is a no-op otherwise; hipcc cross-compiles gfx950 without a GPU)."""
    import shutil
    import subprocess
    if shutil.which("make") and os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.run(["make", "-j8", "-C", os.path.join(ROOT, "fruitnerf_amd", "csrc")], stdout=subprocess.DEVNULL,
                       check=True)


# Collection order: kernel-parity suites first, anything that starts subprocesses / process groups / bench.py last, so
# that under `-x` a harness failure can never hide a parity test (round 4: a timing assertion in test_gpu_distributed.py
# stopped the driver's run at test 66 of 160).  Unlisted files keep their alphabetical place in the middle.
_FIRST = ("test_golden", "test_gpu_forward_parity", "test_gpu_reference_pins", "test_gpu_properties",
          "test_gpu_training_parity", "test_gpu_trained_state", "test_gpu_bf16", "test_gpu_modes", "test_gpu_cloud",
          "test_gpu_determinism")
_LAST = ("test_gpu_sharding", "test_gpu_distributed")


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if name in _FIRST:
            return (0, _FIRST.index(name))
        if name in _LAST:
            return (2, _LAST.index(name))
        return (1, 0)
    items.sort(key=rank)       # stable: the order inside a file is kept


@pytest.fixture(scope="session")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")
