import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip: only auto-skip when the user
    # did not ask for gpu tests explicitly.
    if _gpu_available():
        return
    if "gpu" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    # the margin of every floating-point parity comparison of this run (tests/parity.py)
    from tests import parity
    if not parity.OBSERVED:
        return
    tr = terminalreporter
    tr.write_sep("-", "observed floating-point parity errors (max over the run; bar 1e-4)")
    for label in sorted(parity.OBSERVED):
        err, refmax, atol = parity.OBSERVED[label]
        tr.write_line(f"{label:70s} max|err| {err:9.3e}   max|ref| {refmax:9.3g}   tol {atol:8.1e}")
