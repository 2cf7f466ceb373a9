"""Compute-precision policy of the HIP path.

bf16  - MFMA bf16 GEMMs / flash attention with fp32 accumulation, fp32 residual stream, statistics and losses
        (the throughput configuration; BASELINE.json asks for bf16 MFMA).
f32   - exact-fp32 MFMA GEMMs and attention (the parity configuration: <=1e-3 rel logits, bit-exact argmax vs the
        reference's fp32 CPU path).
"""
import torch

_compute = torch.bfloat16


def set_compute_dtype(dt):
    global _compute
    if isinstance(dt, str):
        dt = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "f32": torch.float32, "fp32": torch.float32,
              "float32": torch.float32}[dt]
    assert dt in (torch.bfloat16, torch.float32)
    _compute = dt


def compute_dtype():
    return _compute


def is_bf16():
    return _compute == torch.bfloat16
