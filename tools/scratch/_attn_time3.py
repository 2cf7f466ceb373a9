import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vfmseg_amd import ops
def run(B, H, d, Np, ex):
    M = B * Np + B * ex
    qkv = torch.randn(M, 3 * H * d, device="cuda").bfloat16()
    o = torch.empty(M, H * d, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, H, Np + ex, device="cuda")
    do = torch.randn(M, H * d, device="cuda").bfloat16()
    dqkv = torch.empty_like(qkv)
    D = H * d
    def fwd(): ops.attn_fwd(qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:], o, lse, B, H, d, Np, ex, Np, ex, d ** -0.5)
    def bwd(): ops.attn_bwd(qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:], o, lse, do, dqkv[:, :D], dqkv[:, D:2*D], dqkv[:, 2*D:], B, H, d, Np, ex, Np, ex, d ** -0.5)
    def t(f, n=50):
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    fl = 4.0 * B * H * (Np + ex) ** 2 * d
    tf, tb = t(fwd), t(bwd)
    print(f"B{B} N{Np}+{ex}: fwd {tf:6.1f} us ({fl / tf / 1e6:6.0f} TF)  bwd {tb:6.1f} us ({2.5 * fl / tb / 1e6:6.0f} TF alg)")
for cfg in [(4, 16, 64, 1024, 1), (4, 16, 64, 1024, 0), (8, 16, 64, 1024, 0), (4, 16, 64, 2048, 0), (16, 16, 64, 1024, 0), (9, 16, 64, 1024, 1)]:
    run(*cfg)
