import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def host_math():
    """ctypes handle of the CPU harness around csrc/trace_math.h (tests/host_math)."""
    import ctypes
    d = os.path.join(ROOT, "tests", "host_math")
    out = os.path.join(d, "_build", "libhost_math.so")
    src = os.path.join(d, "host_math.cpp")
    hdr = os.path.join(ROOT, "tensorflowraytrace_amd", "csrc", "trace_math.h")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if (not os.path.exists(out)
            or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call([
            "g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off",
            "-I", os.path.dirname(hdr), src, "-o", out])
    return ctypes.CDLL(out)


@pytest.fixture
def cpu_backend(monkeypatch):
    """Oracle-backed stand-ins for ops.build_faces / ops.trace3d (tests/cpu_backend.py) so host
    logic can run without a GPU.  Test infrastructure only."""
    import cpu_backend as cb
    cb.install(monkeypatch)
    yield cb
    import tensorflowraytrace_amd.config as config
    config._device = None
