"""Times the bf16 attention forwards at the bs-2 backbone shape (4 images x 16 heads x 1024(+1) tokens); modes of
vfm_tune("attn_fwd64"): 0 = 32 queries per wave, 1 = 64 queries per wave, 3 / 5 / 7 = timing experiments (no LDS fragment reads / no exp / both)."""
import sys
import torch
from vfmseg_amd import ops, lib as L
B, H, d, n = 4, 16, 64, 1024
g = torch.Generator().manual_seed(0)
lib = L.load()
for ex in (1, 0):
    q, k, v = [(torch.randn(B * n + B * ex, H * d, generator=g)).to(torch.bfloat16).cuda() for _ in range(3)]
    o = torch.empty_like(q)
    lse = torch.empty(B, H, n + ex, device="cuda")
    for mode in (0, 1, 3, 9, 5, 7):
        lib.vfm_tune(b"attn_fwd64", mode)
        for _ in range(5):
            ops.attn_fwd(q, k, v, o, lse, B, H, d, n, ex, n, ex, d ** -0.5)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            ops.attn_fwd(q, k, v, o, lse, B, H, d, n, ex, n, ex, d ** -0.5)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 10
        print("extra=%d attn_fwd64=%d: %.1f us  (%.0f TFLOP/s)  checksum %.4f" % (ex, mode, us, 4.0 * B * H * (n + ex) ** 2 * d / us / 1e6, o.float().abs().mean().item()))
    lib.vfm_tune(b"attn_fwd64", 1)
