import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(autouse=True)
def _count_planned_operations():
    """backend.HITS counts the library calls made from Python; inside a launch plan (lidal_amd/network/plan.py) the
    operators are words of one lidal_plan_run call.  Tests that assert "the HIP kernel really ran" read HITS, so under
    test every plan also tallies its operations there (off in production: it walks the plan in Python)."""
    try:
        from lidal_amd.network import plan
    except Exception:           # noqa: BLE001
        yield
        return
    saved = plan.TALLY
    plan.TALLY = True
    yield
    plan.TALLY = saved
