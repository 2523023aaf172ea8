import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_sessionfinish(session, exitstatus):
    """the parity deltas the GPU tests observed (tests/helpers.py:observe) -> gpurun_out/parity_observed.json (merged back by gpurun)"""
    import json
    from tests import helpers
    if not helpers.OBSERVED:
        return
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_observed.json"), "w") as f:
            json.dump(helpers.OBSERVED, f, indent=1, sort_keys=True)
    except OSError:
        pass
