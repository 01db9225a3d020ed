import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


# The one-launch epoch kernel (MultiModN._small_epoch) would take every small device-resident training loop of this suite,
# including the ones written to compare the step-by-step tiers with each other: it is switched on where it is the subject
# (tests/test_epoch_small.py sets MMN_EPOCH_KERNEL=1).
os.environ.setdefault("MMN_EPOCH_KERNEL", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
