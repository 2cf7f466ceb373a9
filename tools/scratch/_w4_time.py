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
M, N = 4096, 4096
for K in (1024, 4096):
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ops.tune("gemm_cfg", int(os.environ.get("CFG", "32")))
    for dbg in [int(x) for x in os.environ.get("DBGS", "0,1,3,5,7,9,15").split(",")]:
        ops.tune("pp_dbg", dbg)
        t = timeit(lambda: ops.gemm(a, b, c))
        print(f"K={K} dbg={dbg:2d}: {t:7.1f} us  {2.0*M*N*K/t/1e6:6.0f} TF", flush=True)
    ops.tune("pp_dbg", 0)
    t = timeit(lambda: torch.matmul(a, b.t()))
    print(f"K={K} torch: {t:7.1f} us  {2.0*M*N*K/t/1e6:6.0f} TF", flush=True)
