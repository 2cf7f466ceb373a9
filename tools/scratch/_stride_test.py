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
    return e0.elapsed_time(e1) / iters
M, N = 4096, 1024
for K, pad in [(4096, 0), (4096, 64), (4096, 8), (4096, 128), (4096, 192), (1024, 0), (1024, 64), (3072, 0), (3072, 64)]:
    A = torch.randn(M, K + pad, device="cuda").bfloat16(); B = torch.randn(N, K + pad, device="cuda").bfloat16()
    a, b = A[:, :K], B[:, :K]
    c = torch.empty(M, N, dtype=torch.float32, device="cuda")
    res = []
    for cfg in (17, 31):
        ops.tune("gemm_cfg", cfg)
        t = timeit(lambda: ops.gemm(a, b, c))
        res.append(f"c{cfg}: {2.0*M*N*K/t/1e9:6.0f} TF ({t*1e3:5.1f} us)")
    ops.tune("gemm_cfg", -1)
    t = timeit(lambda: torch.matmul(a, b.t()))
    print(f"K={K} ld={K+pad}: " + "  ".join(res) + f"  torch {2.0*M*N*K/t/1e9:6.0f} TF", flush=True)
