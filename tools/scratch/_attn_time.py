import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vfmseg_amd import ops
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
B, H, d = 4, 16, 64
for Np, ex, eq, ek in [(1024, 0, 0, 0), (1024, 1, 1, 1), (1024, 1, 1, 0), (1024, 1, 0, 1)]:
    n = Np + ex
    M = B * Np + B * ex
    qkv = torch.randn(M, 3 * H * d, device="cuda").bfloat16()
    o = torch.empty(M, H * d, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, H, n, device="cuda")
    do = torch.randn(M, H * d, device="cuda").bfloat16()
    dqkv = torch.empty_like(qkv)
    D = H * d
    f = lambda: ops.attn_fwd(qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:], o, lse, B, H, d, Np, eq, Np, ek, d ** -0.5)
    b = lambda: ops.attn_bwd(qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:], o, lse, do, dqkv[:, :D], dqkv[:, D:2*D], dqkv[:, 2*D:], B, H, d, Np, eq, Np, ek, d ** -0.5)
    tf, tb = timeit(f), timeit(b)
    fl = 4.0 * B * H * n * n * d
    print(f"N={Np} eq={eq} ek={ek}: fwd {tf:6.1f} us ({fl/tf/1e6:5.0f} TF)  bwd {tb:6.1f} us ({2.5*fl/tb/1e6:5.0f} TF)", flush=True)
