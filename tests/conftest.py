import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def cref():
    """The plain-C CPU restatement (oracle/pasta_ref.c), built on demand."""
    from oracle import cref as c
    c.lib()
    return c


@pytest.fixture(scope="session")
def ctx():
    """One HIP context for the whole GPU session (one process, one GPU)."""
    import vdf_amd
    c = vdf_amd.Context(0)
    yield c
    c.close()
