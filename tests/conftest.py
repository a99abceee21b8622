import os
import sys

import pytest

# The parity tests run the UNet's stock convolutions in PyTorch's default MIOpen mode (immediate): MIOpen's find mode
# (pww_hip.enable_miopen_find, the drop-in API's default) picks solvers by timing them, so which convolution kernels -- and
# which roundings -- a test sees would depend on the box. The setting itself is covered by test_host_logic.py; its
# effect on throughput by bench.py.
os.environ.setdefault("PWW_MIOPEN_FIND", "0")

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "paint-with-words-sd_amd")
for p in (PKG, REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test, enabled with PWW_SLOW=1")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("PWW_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="set PWW_SLOW=1 to run")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def built_lib():
    """libpww_hip.so, built in-tree if missing (hipcc cross-compiles without a GPU)."""
    sys.path.insert(0, PKG)
    import build as pww_build
    return pww_build.build_lib()


@pytest.fixture(scope="session")
def experiments_lib(built_lib):
    """libpww_hip_experiments.so (+ tests/native/attn_check_experiments): the product library compiled with -DPWW_EXPERIMENTS=1 -- round 3's
    in-launch statistic, the attention + to_out launch, the A/B kernels behind PWW_DEBUG. Built in-tree when missing (about a minute of
    hipcc); only tests ask for it."""
    import build as pww_build
    lib = pww_build.build_experiments()
    pww_build.build_native_check_experiments()
    return lib


@pytest.fixture(scope="session")
def gpu_device(built_lib):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("test is marked gpu but no HIP device is visible")
    import pww_hip
    pww_hip.load_library()
    assert pww_hip.device_arch() == "gfx950"
    return torch.device("cuda:0")
