import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: further variants of a full-size / multi-variant GPU check whose ONE representative stays in the default run "
                                       "(run them with -m 'gpu and slow', FAUCET_FULL_CHECKS=1, or scripts/full_checks_parallel.sh)")


def pytest_collection_modifyitems(config, items):
    """the driver-run GPU suite must stay well inside its 900 s.  Tests marked `slow` are deselected unless the marker
    expression names them or FAUCET_FULL_CHECKS=1 is set; every one of them has a sibling (another parameter of the same test) in the default run."""
    if "slow" in (config.getoption("-m") or "") or os.environ.get("FAUCET_FULL_CHECKS") == "1":
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("slow") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep
