import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vfmseg_amd import ops
B, H, d, Np, ex = 4, 16, 64, 1024, 1
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
for pad in [int(x) for x in (sys.argv[1:] or ["0"])]:
    ops.tune("attn_lds_pad", pad)
    print(f"pad {pad:6d}: fwd {t(fwd):6.1f} us  bwd {t(bwd):6.1f} us")
