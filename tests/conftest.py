import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(REPO, "tests", "golden", "tiny_golden.npz"))


@pytest.fixture(scope="session")
def tiny_sd(golden):
    from tests.procedural import procedural_state_dict
    keys = [str(k) for k in golden["keys"]]
    shapes = [tuple(int(x) for x in str(s).split(",")) for s in golden["shapes"]]
    return procedural_state_dict(zip(keys, shapes))
