import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


def pytest_sessionfinish(session, exitstatus):
    """write what the parity tests measured (tests/_parity.py) to gpurun_out/parity_errors.json"""
    try:
        import _parity
        _parity.dump()
    except Exception:           # never let bookkeeping fail a run
        pass
