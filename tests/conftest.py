import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (the real reference library, built where /root/reference exists)")
    config.addinivalue_line("markers", "slow: the long generations of the full-size oracle replays (round 4's lengths); run with PIPEINFER_SLOW_TESTS=1 (tools_dev/round_end.sh does), skipped otherwise so that `-m gpu` stays inside the driver's time limit")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("PIPEINFER_SLOW_TESTS") == "1":
        return
    skip = pytest.mark.skip(reason="long oracle replay: set PIPEINFER_SLOW_TESTS=1")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)
