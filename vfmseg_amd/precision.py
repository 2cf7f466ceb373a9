"""Compute-precision policy of the HIP path.

bf16  - MFMA bf16 GEMMs / flash attention with fp32 accumulation, fp32 residual stream, statistics and losses
        (the throughput configuration; BASELINE.json asks for bf16 MFMA).
f32   - exact-fp32 MFMA GEMMs and attention (the parity configuration: <=1e-3 rel logits, bit-exact argmax vs the
        reference's fp32 CPU path).
bf16x3 - the f32 configuration (fp32 tensors, exact-fp32 attention / norms / losses) with every fp32 GEMM computed as ONE bf16 MFMA
        GEMM over split operands ([hi | hi | lo] x [hi | lo | hi], K' = 3 K, fp32 accumulate; ops.gemm / vfm_split3): 16 significant bits
        per operand instead of 24 - products to ~2^-16 relative, three orders inside north_star's 1e-3 - at a third of the bf16 GEMM
        rate instead of the fp32 MFMA's 1/16.  The in-tolerance mode with a usable speed.
fp16  - the bf16 configuration with IEEE fp16 in place of bf16 everywhere (libvfmseg_hip_f16.so: fp16 storage, v_mfma_f32_32x32x16_f16,
        fp32 accumulation / residual stream / statistics / losses): what the reference's `--amp` computes in (tools/train.py:87-102 ->
        mmengine AmpOptimWrapper: torch.autocast(fp16) + GradScaler).  Needs the dynamic loss scale (optim.AmpOptimWrapper): fp16
        gradients under- and overflow where bf16 ones do not.
"""
import torch

from . import lib as _L

_compute = torch.bfloat16
_split3 = False


def set_compute_dtype(dt):
    global _compute, _split3
    _split3 = isinstance(dt, str) and dt in ("bf16x3", "split3")
    if _split3:
        dt = "f32"
    if isinstance(dt, str):
        dt = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "f32": torch.float32, "fp32": torch.float32,
              "float32": torch.float32, "fp16": torch.float16, "f16": torch.float16, "float16": torch.float16}[dt]
    assert dt in (torch.bfloat16, torch.float32, torch.float16)
    _compute = dt
    _L.set_half(torch.float16 if dt == torch.float16 else torch.bfloat16)


def compute_dtype():
    return _compute


class compute_as:
    """`with compute_as(torch.float32):` - the ops inside take fp32 as their compute dtype while the ACTIVE LIBRARY stays what it is
    (set_compute_dtype would switch libraries: a 16-bit tensor of the current mode could then not be consumed).  Used by the fp16 mode's
    predictions to run the small decode heads in fp32 on fp32 feature taps (see eval_heads_fp32)."""

    def __init__(self, dt, split3=False):
        self.dt, self.split3 = dt, split3

    def __enter__(self):
        global _compute, _split3
        self.old, _compute, _split3 = (_compute, _split3), self.dt, self.split3

    def __exit__(self, *a):
        global _compute, _split3
        _compute, _split3 = self.old


def eval_heads_fp32():
    """fp16 mode, predictions only: the feature taps leave the backbone in fp32 (they are cast from the fp32 residual stream anyway) and the
    decode heads (5 % of a prediction's FLOPs) run on the exact-fp32 MFMA.  Measured against the reference-made goldens
    (tests/golden/slide_modes.npz, ms_inference.npz): with fp16 heads the logits sit AT north_star's 1e-3 bound (1.0-1.2e-3: fp16's
    2^-11 rounding of the taps and of the heads' intermediate maps is not averaged over a deep stack as the backbone's roundings are);
    with fp32 heads every mode is inside it.  The train step under `--amp` keeps fp16 heads: that is what the reference's autocast
    computes in.  VFMSEG_FP16_EVAL_HEADS=fp16 turns it off."""
    import os
    return _compute == torch.float16 and os.environ.get("VFMSEG_FP16_EVAL_HEADS", "fp32") != "fp16"


def is_bf16():
    """True in the 16-bit MFMA configurations (bf16, or fp16 with the twin library)."""
    return _compute in (torch.bfloat16, torch.float16)


def is_half(dt):
    """dt is a 16-bit MFMA operand type (bf16 / fp16)."""
    return dt == torch.bfloat16 or dt == torch.float16


def split3():
    """True in the bf16x3 mode: fp32 GEMMs run as split-bf16 MFMA GEMMs."""
    return _split3


def mode_name():
    return "bf16x3" if _split3 else {torch.bfloat16: "bf16", torch.float16: "fp16", torch.float32: "f32"}[_compute]
