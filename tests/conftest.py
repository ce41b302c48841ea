import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Make sure the native pieces exist (no-op when they are up to date): the HIP library is needed even by the
    CPU suite (symbol / loud-failure checks), the oracle by almost every test.  hipcc cross-compiles without a GPU."""
    import shutil
    from colosseumrl_amd import _native
    if not os.path.exists(_native.LIB_PATH) and shutil.which("make") and os.path.exists("/opt/rocm/bin/hipcc"):
        _native.build()
    from oracle import oracle as O
    O.build()


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
            return {k: z[k] for k in z.files}
    return load
