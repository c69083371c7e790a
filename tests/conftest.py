import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_runtime_first():
    """On a GPU box, let torch bring up its HIP runtime before libadypt_hip.so touches the device (the order bench.py
    uses): torch ships its own copy of the runtime and reports "No HIP GPUs are available" when it initialises second."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The HIP library and the oracle are built in-tree before any test imports them."""
    import __graft_entry__ as g
    lib = os.path.join(ROOT, "adypt_amd", "libadypt_hip.so")
    orc = os.path.join(ROOT, "oracle", "liboracle.so")
    if not (os.path.exists(lib) and os.path.exists(orc)):
        g.build()


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
