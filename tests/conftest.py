import sys
from pathlib import Path

import pytest

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return HERE / "golden"


def pytest_generate_tests(metafunc):
    # tests that name the `fir_mode` fixture run once per arithmetic of the advanced version's filter bank
    if "fir_mode" in metafunc.fixturenames:
        metafunc.parametrize("fir_mode", ["default", "f16x3"])


@pytest.fixture(autouse=True)
def _select_fir_mode(request):
    """tests/gpu_common.py: ctx() and tol() follow the test's fir_mode (default engine when it names none)"""
    import gpu_common
    m = request.getfixturevalue("fir_mode") if "fir_mode" in request.fixturenames else "default"
    if m == "f16x3" and "advanced" in getattr(request.node, "callspec", type("x", (), {"params": {}})).params \
            and not request.node.callspec.params["advanced"]:
        pytest.skip("the basic version does not pass through the filter bank: one mode is enough")
    gpu_common.set_mode(m)
    yield
    gpu_common.set_mode("default")
