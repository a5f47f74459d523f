import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_sessionstart(session):
    """Build the native pieces if a fresh checkout has none (hipcc cross-compiles without a GPU):
    cufhe_amd/libcufhe_amd.so and oracle/liboracle.so are git-ignored build products."""
    import importlib.util
    import subprocess
    so = os.path.join(ROOT, "cufhe_amd", "libcufhe_amd.so")
    if not os.path.exists(so):
        spec = importlib.util.spec_from_file_location("cufhe_amd_build", os.path.join(ROOT, "cufhe_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build()
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "all"], stdout=subprocess.DEVNULL)


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
