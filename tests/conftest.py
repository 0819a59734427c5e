import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _clean_tuning_table():
    """The library's tuning knobs live in one process-wide table (sefd_amd.tuning, include/sefd.h sefd_tuning_*), not in the environment: tests set
    them through `util.knobs` and every test starts from - and leaves - an empty table."""
    yield
    try:
        import sefd_amd  # noqa: F401
        from sefd_amd import tuning
        tuning.clear()
    except Exception:
        pass
