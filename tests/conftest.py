import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
    """The checker library (C oracle) is built by `make` when it is missing or stale.  Do that now, before any test initialises the
    GPU: a process that has touched the GPU must never REPLACE itself with another program (exec*), and the build belongs outside
    the tests anyway.  Starting fresh CHILD processes from a GPU-initialised process is allowed on the GPU pool -- the multi-rank
    tests (tests/test_gpu_parallel.py) Popen their rank workers from this pytest process -- within the pool's limit of six
    processes on the card.  (The product library is never built from here: __graft_entry__.build() does that;
    catint_amd._capi fails loudly if it is missing.)"""
    try:
        from oracle import c_oracle
        c_oracle.load()
    except Exception as e:      # no compiler: the tests that need the C oracle will say so themselves
        sys.stderr.write('conftest: C oracle not available (%s)\n' % e)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
