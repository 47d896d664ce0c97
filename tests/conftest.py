import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        from openmm_agbnp_plugin_amd import _lib
        return _lib.load().agbnp_hip_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_required():
    """GPU tests must not silently pass: without the extension or a device they FAIL (not skip) under -m gpu."""
    from openmm_agbnp_plugin_amd import _lib
    lib = _lib.load()  # raises ImportError if libagbnp_hip.so is missing
    assert lib.agbnp_hip_device_count() > 0, "no HIP device visible: the HIP path cannot run (no CPU fallback exists)"
    return lib


@pytest.fixture(scope="session")
def systems():
    import openmm_agbnp_plugin_amd as P
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = P.load_system(name)
        return cache[name]

    return get
