import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(HERE, "golden", name + ".npz"))
    return load


@pytest.fixture(scope="session")
def engine():
    """One DiffuseTransfer context for the whole GPU session (fails loudly without the HIP library or a GPU)."""
    import radiativetransfer_amd as rt
    eng = rt.DiffuseTransfer()
    yield eng
    eng.close()
