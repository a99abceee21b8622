"""Runs the native C++ harness (tests/native/attn_check.cpp) that drives the C ABI directly and
checks every entry point against an independent fp64 host reference: once over the product library
(libpww_hip.so) and once, with the cases of the moved entry points, over libpww_hip_experiments.so."""
import subprocess
import sys

import pytest


@pytest.mark.gpu
def test_native_c_abi_harness(gpu_device):
    import build as pww_build
    exe = pww_build.build_native_check()
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    sys.stdout.write(res.stdout[-6000:])
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert "NATIVE CHECK OK" in res.stdout


@pytest.mark.gpu
def test_native_c_abi_harness_experiments(gpu_device, experiments_lib):
    """The same harness built with -DPWW_EXPERIMENTS=1 against libpww_hip_experiments.so: + round 3's in-launch statistic (bit-identical to
    the two-launch path, hints, compact maps, hipGraph replays) and the attention + to_out launch."""
    import build as pww_build
    exe = pww_build.build_native_check_experiments()
    res = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    sys.stdout.write(res.stdout[-6000:])
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert "NATIVE CHECK OK" in res.stdout
