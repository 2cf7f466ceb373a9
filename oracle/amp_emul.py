"""TEST INFRASTRUCTURE ONLY - torch.autocast(device_type="cuda", dtype=float16) restated for CPU tensors, so that oracle/torch_ref.py
can be run the way the reference's `--amp` runs the model (tools/train.py:87-102 -> mmengine AmpOptimWrapper.optim_context ->
torch.autocast; CUDA default dtype float16) without a GPU.  torch's own CPU autocast has a different op policy (softmax / layer_norm
stay in the input dtype there), so the CUDA policy (aten/src/ATen/autocast_mode.cpp, torch 2.x) is applied by hand to the functions
the oracle calls:

  lower-precision list (inputs cast to fp16, fp16 result):  linear, conv2d, conv_transpose2d, matmul (@), einsum
  fp32 list (inputs cast to fp32, fp32 result):             layer_norm, group_norm, softmax, cross_entropy (= log_softmax + nll_loss)
  everything else runs in the dtype of its inputs with the usual type promotion (gelu / relu / interpolate / batch norm on fp16
  tensors give fp16; fp16 * fp32 LayerScale and fp32 residual + fp16 branch give fp32).

A product is computed as  round16(op(fp32(x16), fp32(w16)))  - fp32 accumulation and one rounding of the result, which is what both
cuBLAS/rocBLAS fp16 GEMMs and the MFMA do - with REAL fp16 tensors at the op boundaries, so autograd rounds the gradients to fp16 at
exactly the places torch does (the gradient of an fp16 tensor is fp16).  Kernels that CUDA runs "in fp16" but accumulate in fp32
internally (batch norm statistics) are upcast inside and rounded once at the end.

Parity status: a restatement of a published policy; pinned only by its own CPU test (tests/test_oracle_golden.py::test_amp_emulation_*:
rounding points, dtypes, gradient dtypes).  There is no fp16 output of the reference to compare with (no CUDA device in this
pipeline), so fp16 parity claims are stated as "as close to the reference's fp32 results as this emulation of its fp16 run is".
"""
import contextlib

import torch
import torch.nn.functional as F


def _lp(t, dt):
    return t if t is None or not t.is_floating_point() else t.to(dt)


def _f32(t):
    return t if t is None or not t.is_floating_point() else t.float()


@contextlib.contextmanager
def cuda_autocast(dtype=torch.float16):
    """Patch the functions oracle/torch_ref.py calls with their CUDA-autocast behaviour for the duration of the block."""
    o_linear, o_conv2d, o_convt2d = F.linear, F.conv2d, F.conv_transpose2d
    o_ln, o_gn, o_ce = F.layer_norm, F.group_norm, F.cross_entropy
    o_matmul, o_rmatmul, o_softmax, o_einsum = torch.Tensor.__matmul__, torch.Tensor.__rmatmul__, torch.Tensor.softmax, torch.einsum

    def linear(x, w, b=None):
        return o_linear(_lp(x, dtype).float(), _lp(w, dtype).float(), None if b is None else _lp(b, dtype).float()).to(dtype)

    def conv2d(x, w, b=None, *a, **k):
        return o_conv2d(_lp(x, dtype).float(), _lp(w, dtype).float(), None if b is None else _lp(b, dtype).float(), *a, **k).to(dtype)

    def conv_transpose2d(x, w, b=None, *a, **k):
        return o_convt2d(_lp(x, dtype).float(), _lp(w, dtype).float(), None if b is None else _lp(b, dtype).float(), *a, **k).to(dtype)

    def matmul(a, b):
        return o_matmul(_lp(a, dtype).float(), _lp(b, dtype).float()).to(dtype)

    def einsum(eq, *ops):
        return o_einsum(eq, *[_lp(t, dtype).float() for t in ops]).to(dtype)

    def layer_norm(x, shape, w=None, b=None, eps=1e-5):
        return o_ln(_f32(x), shape, _f32(w), _f32(b), eps)

    def group_norm(x, g, w=None, b=None, eps=1e-5):
        return o_gn(_f32(x), g, _f32(w), _f32(b), eps)

    def softmax(self, dim=-1, **k):
        return o_softmax(_f32(self), dim=dim, **k)

    def cross_entropy(x, *a, **k):
        return o_ce(_f32(x), *a, **k)

    F.linear, F.conv2d, F.conv_transpose2d = linear, conv2d, conv_transpose2d
    F.layer_norm, F.group_norm, F.cross_entropy = layer_norm, group_norm, cross_entropy
    torch.Tensor.__matmul__, torch.Tensor.softmax, torch.einsum = matmul, softmax, einsum
    torch.Tensor.__rmatmul__ = lambda self, other: matmul(other, self)
    try:
        yield
    finally:
        F.linear, F.conv2d, F.conv_transpose2d = o_linear, o_conv2d, o_convt2d
        F.layer_norm, F.group_norm, F.cross_entropy = o_ln, o_gn, o_ce
        torch.Tensor.__matmul__, torch.Tensor.__rmatmul__, torch.Tensor.softmax, torch.einsum = o_matmul, o_rmatmul, o_softmax, o_einsum
