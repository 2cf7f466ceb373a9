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


if os.environ.get("VFMSEG_TEST_HALF", "") == "fp16":
    # Second pass of the kernel-level suites against the fp16 twin library (started by tests/test_fp16_twin_gpu.py in a child pytest):
    # libvfmseg_hip_f16.so is the SAME sources with the 16-bit type switched (csrc/common.h), so the same tests apply with every
    # 16-bit tensor fp16 instead of bf16.  The test bodies spell the type as `torch.bfloat16` / `.bfloat16()`; in this pass those
    # names mean fp16 (tensors, the tests' own rounding of the expected values, and precision mode "bf16" -> the fp16 library).
    # A kernel that still assumed bf16 bit patterns anywhere would compute garbage here and fail its comparison.
    import torch
    import vfmseg_amd.lib  # noqa: F401  (binds the real dtypes before the names are switched)
    import vfmseg_amd.precision as _P
    torch.bfloat16 = torch.float16
    torch.Tensor.bfloat16 = torch.Tensor.half
    _P.set_compute_dtype("bf16")
    assert _P.mode_name() == "fp16" and vfmseg_amd.lib.half_dtype() == torch.float16


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: multi-second CPU oracle runs")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
