import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def pytest_collection_modifyitems(config, items):
    """A wedged GPU test must fail (with every thread's stack) instead of eating the box's time limit."""
    for it in items:
        if it.get_closest_marker('gpu') is not None and it.get_closest_marker('timeout') is None:
            it.add_marker(pytest.mark.timeout(180, method="thread"))


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')
