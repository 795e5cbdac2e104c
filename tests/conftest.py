import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding
    oracle_binding.lib()
    return oracle_binding


@pytest.fixture(scope="session")
def f360():
    import f360_amd
    if not os.path.exists(f360_amd.LIB_PATH):
        f360_amd.build_native()
    f360_amd.lib()
    return f360_amd


@pytest.fixture(scope="session")
def gpu_ctx(f360):
    if f360.device_count() < 1:
        pytest.fail("gpu-marked test started without a visible HIP device")
    ctx = f360.Context(0)
    yield ctx
    ctx.close()
