import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vfmseg_amd import ops
B, H, d, Np, ex = 4, 16, 64, 1024, 1
M = B * Np + B * ex
n = Np + ex
qkv = torch.randn(M, 3 * H * d, device="cuda").bfloat16()
o = torch.empty(M, H * d, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B, H, n, device="cuda")
do = torch.randn(M, H * d, device="cuda").bfloat16()
dqkv = torch.empty_like(qkv)
D = H * d
for _ in range(4):
    ops.attn_fwd(qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:], o, lse, B, H, d, Np, ex, Np, ex, d ** -0.5)
    ops.attn_bwd(qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:], o, lse, do, dqkv[:, :D], dqkv[:, D:2*D], dqkv[:, 2*D:], B, H, d, Np, ex, Np, ex, d ** -0.5)
torch.cuda.synchronize()
