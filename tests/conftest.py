import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


if os.environ.get("VFMSEG_NAN_FILL", "0") == "1":
    # Hunt for reads of uninitialised memory: every torch.empty() float buffer starts as NaN (torch's deterministic-debug
    # facility), so a kernel that consumes a byte it was supposed to have been given by an earlier kernel poisons the result
    # instead of silently reading whatever the caching allocator left there.  Run: VFMSEG_NAN_FILL=1 pytest -m gpu
    import torch
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = True


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: multi-second CPU oracle runs")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
