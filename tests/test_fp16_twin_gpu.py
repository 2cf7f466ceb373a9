"""fp16 twin library (libvfmseg_hip_f16.so = the same sources with -DVFM_HALF_F16): what the reference's `--amp` computes in
(tools/train.py:87-102 -> mmengine AmpOptimWrapper -> torch.autocast(float16) + GradScaler).

1. The kernel-level suites run a second time against the twin (child pytest with VFMSEG_TEST_HALF=fp16, see tests/conftest.py):
   every tile configuration x epilogue of the GEMMs, both attention families, norms, elementwise, SAM flash.
2. fp16 really is fp16: a value bf16 cannot hold but fp16 can survives a cast, one beyond 65504 becomes inf.
3. The library refuses tensors of the other 16-bit type.
Model-level fp16 parity (train step / 3 steps / eval vs the reference goldens, loss-scale dynamics) lives next to the bf16 cases in
test_model_gpu.py, test_fulldepth_gpu.py, test_eval_gpu.py and test_amp_gpu.py."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def fp16_mode():
    from vfmseg_amd.precision import set_compute_dtype
    set_compute_dtype("fp16")
    yield
    set_compute_dtype("bf16")


def test_cast_is_ieee_fp16(fp16_mode):
    from vfmseg_amd import lib as L, ops
    assert L.load().vfm_half_kind() == 1 and L.half_dtype() == torch.float16
    x = torch.tensor([[1.0 + 2.0 ** -10, 65504.0, 70000.0, 2.0 ** -24, -3.14159, 1e-8, 0.0, 0.333]], device="cuda")
    y = torch.empty(1, 8, dtype=torch.float16, device="cuda")
    ops.cast(x, y)
    assert torch.equal(y.cpu(), x.cpu().half())            # RNE, overflow -> inf, subnormals kept: torch's .half()
    assert y[0, 0].item() == 1.0 + 2.0 ** -10 and torch.isinf(y[0, 2]) and y[0, 3].item() == 2.0 ** -24
    z = torch.empty(1, 8, device="cuda")
    ops.cast(y, z)
    assert torch.equal(z.cpu(), x.cpu().half().float())


def test_libraries_refuse_the_other_half_type(fp16_mode):
    from vfmseg_amd import lib as L, ops
    from vfmseg_amd.precision import set_compute_dtype
    x = torch.ones(4, 8, device="cuda")
    with pytest.raises(TypeError):
        ops.cast(x, torch.empty(4, 8, dtype=torch.bfloat16, device="cuda"))
    set_compute_dtype("bf16")
    assert L.load().vfm_half_kind() == 0
    with pytest.raises(TypeError):
        ops.cast(x, torch.empty(4, 8, dtype=torch.float16, device="cuda"))


def test_kernel_suites_against_the_fp16_twin():
    env = dict(os.environ, VFMSEG_TEST_HALF="fp16")
    # not re-run: the split-bf16 precision mode (bf16 by construction) and the off-by-default persistent-GEMM experiment
    cmd = [sys.executable, "-m", "pytest", "tests/test_kernels_gpu.py", "tests/test_sam_flash_gpu.py", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider",
           "-k", "not bf16x3 and not persistent and not experimental"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = "\n".join(r.stdout.strip().splitlines()[-15:])
    print("[fp16 twin] kernel suites:", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
    assert r.returncode == 0, tail + "\n" + r.stderr[-2000:]
