import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402

ge.add_to_path()
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


def pytest_collection_modifyitems(config, items):
    """The statistical tests (end quality of whole fits: families of chaotic trajectories compared on their means) run
    LAST, after every deterministic kernel / net / notebook test: with `pytest -x` a red statistical test can then never
    hide a kernel test again (41 tests behind one)."""
    last = [it for it in items if "end_quality" in it.name]
    if last:
        first = [it for it in items if "end_quality" not in it.name]
        items[:] = first + last


@pytest.fixture(autouse=True, scope="session")
def _cpu_threads():
    """The oracle runs on the host: torch's CPU conv scaling collapses on many-core boxes (256
    threads on the MI355X host: 60x slower than 16, tools/cpu_sweep.py), so cap it."""
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    yield


@pytest.fixture(scope="session")
def built():
    """libdip_hip.so built (hipcc cross-compiles without a GPU) and loadable."""
    ge.build()
    import dip_native
    return dip_native.lib()


@pytest.fixture(scope="session")
def dev(built):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
