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


def pytest_sessionfinish(session, exitstatus):
    """GPU runs: dump the measured parity errors (tests/gpu_checks.py REPORT) next to the test log, so that the numbers
    behind the bf16 gates can be committed under profiles/."""
    mod = sys.modules.get('tests.gpu_checks')
    rep = getattr(mod, 'REPORT', None) if mod is not None else None
    if rep:
        import json
        out = os.environ.get('OMP355_PARITY_REPORT', os.path.join(ROOT, 'gpurun_out', 'parity_report.json'))
        try:
            os.makedirs(os.path.dirname(out), exist_ok=True)
            with open(out, 'w') as f:
                json.dump(rep, f, indent=1)
        except OSError:
            pass
