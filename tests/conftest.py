import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # kill -USR1 <pid of this pytest>: Python stacks of all threads on the REAL stderr (bypasses the capture), for a run that
    # sits inside a library call
    import faulthandler
    import signal
    import sys

    try:
        faulthandler.register(signal.SIGUSR1, file=sys.__stderr__, all_threads=True)
    except (AttributeError, ValueError, OSError):
        pass


def pytest_collection_modifyitems(config, items):
    """A GPU test that stops making progress (a kernel that never returns leaves the host spinning on a mapped word)
    must end the run with a traceback, not sit until the box is reclaimed: every gpu test gets a generous wall limit
    (pytest-timeout, thread method: a C call cannot be interrupted by a signal handler)."""
    try:
        import pytest_timeout  # noqa: F401
    except ImportError:
        return
    for item in items:
        if item.get_closest_marker("gpu") and not item.get_closest_marker("timeout"):
            item.add_marker(pytest.mark.timeout(1500, method="thread"))


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name))

    return load
