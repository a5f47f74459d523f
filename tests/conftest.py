import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def keys(oracle):
    import oracle_lib
    return oracle_lib.Keys(oracle, seed=1)


@pytest.fixture(scope="session")
def engine(keys):
    """The HIP engine initialised with the session keys (GPU tests only)."""
    import cufhe_amd
    cufhe_amd.SetGPUNum(1)
    cufhe_amd.Initialize(keys.bk, keys.ksk)
    yield cufhe_amd
    cufhe_amd.CleanUp()
