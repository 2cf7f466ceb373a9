"""Coarse eval pass GEMMs (M = 2049 tokens): which tile configuration?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vfmseg_amd import ops
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
SHAPES = [(2049, 1024, 1024), (2049, 1024, 4096), (2049, 4096, 1024), (2049, 3072, 1024), (9225, 1024, 1024), (9225, 1024, 4096)]
if len(sys.argv) > 1 and sys.argv[1] == "sam":
    SHAPES = [(9216, 1280, 1280), (9216, 1280, 5120), (9216, 5120, 1280), (9216, 3840, 1280), (9225, 4096, 1024), (9225, 3072, 1024)]
for (M, N, K) in SHAPES:
    A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
    C = torch.empty(M, N, device="cuda"); R = torch.randn(M, N, device="cuda"); bias = torch.randn(N, device="cuda")
    out = []
    for cfg in (-1, 10, 18, 17, 34, 31, 33):
        try:
            ops.tune("gemm_cfg", cfg)
            us = t(lambda: ops.gemm(A, B, C, bias=bias, residual=R))
            out.append(f"c{cfg}: {us:6.1f} us {2.0*M*N*K/us/1e6:4.0f} TF")
        except Exception as e:
            out.append(f"c{cfg}: n/a")
    ops.tune("gemm_cfg", -1)
    print(f"M{M} N{N} K{K}: " + " | ".join(out))
