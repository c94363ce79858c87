import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cmk_lib():
    from centermask2_amd import _lib
    return _lib.load()


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Observed maximum errors of the parity checks (tests/helpers.py: close / close_abs), so the margin to the bar is on record."""
    try:
        from tests import helpers
    except Exception:
        return
    if not helpers.OBSERVED:
        return
    terminalreporter.write_line("observed max errors (what: err):")
    for k in sorted(helpers.OBSERVED):
        terminalreporter.write_line("  {}: {:.3e}".format(k, helpers.OBSERVED[k]))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        import json
        with open(os.path.join(out, "observed_errors.json"), "w") as f:
            json.dump(helpers.OBSERVED, f, indent=1, sort_keys=True)
