import os
import sys

import numpy as np
import pytest

# torch ships its own copy of the HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  Loaded first, it
# is the copy libadypt_hip.so binds to as well (same SONAME) and the process has ONE runtime; loaded second, the process
# ends up with two runtimes and whichever touches the GPU second reports that there is no device.  conftest.py is
# imported before any test module, so this import settles the order for the tests that use torch next to the library
# (bench.py imports torch first for the same reason; the library itself never needs torch).
try:
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The HIP library and the oracle are built in-tree before any test imports them."""
    import __graft_entry__ as g
    lib = os.path.join(ROOT, "adypt_amd", "libadypt_hip.so")
    orc = os.path.join(ROOT, "oracle", "liboracle.so")
    if not (os.path.exists(lib) and os.path.exists(orc)):
        g.build()
    # the library's test hooks (several shards / ranks on one device, planted faults) are ignored unless a process asks for them
    # explicitly (include/adypt_hip.h: adypt_enable_test_hooks): the tests do, the product never does
    from adypt_amd import api
    api.enable_test_hooks()


@pytest.fixture(scope="session")
def golden():
    return GOLDEN


@pytest.fixture(scope="session")
def sobol_matrices():
    return np.fromfile(os.path.join(GOLDEN, "sobol_matrices_64x32.u32"), dtype=np.uint32).reshape(64, 32)


@pytest.fixture(scope="session")
def scene_cache(tmp_path_factory):
    d = os.environ.get("ADYPT_CACHE")
    if d:
        os.makedirs(d, exist_ok=True)
        return d
    return str(tmp_path_factory.mktemp("adypt_scenes"))
