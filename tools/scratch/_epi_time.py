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
bias = torch.randn(N, device="cuda")
for K in (128, 256, 1024):
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"); c2 = torch.empty_like(c)
    cf = torch.empty(M, N, dtype=torch.float32, device="cuda"); res = torch.randn(M, N, device="cuda")
    for cfg in (17, 30, 32):
        ops.tune("gemm_cfg", cfg)
        t0 = timeit(lambda: ops.gemm(a, b, c))
        t1 = timeit(lambda: ops.gemm(a, b, c, bias=bias, ep_mode=ops.EP_GELU, c2=c2))
        t2 = timeit(lambda: ops.gemm(a, b, c, ep_mode=ops.EP_MUL_GELU_GRAD, aux=c2))
        t3 = timeit(lambda: ops.gemm(a, b, cf, bias=bias, residual=res))
        print(f"K={K:5d} cfg{cfg}: plain bf16 {t0:6.1f}  gelu+c2 {t1:6.1f}  mul_gelu_grad {t2:6.1f}  f32+res {t3:6.1f} us", flush=True)
    ops.tune("gemm_cfg", -1)
    t = timeit(lambda: torch.matmul(a, b.t(), out=c))
    print(f"K={K:5d} torch plain {t:6.1f} us", flush=True)
x = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
t = timeit(lambda: x.fill_(1.0)); print(f"fill 32 MB: {t:6.1f} us")
y = torch.empty(M, N, dtype=torch.float32, device="cuda")
t = timeit(lambda: y.fill_(1.0)); print(f"fill 64 MB: {t:6.1f} us")
t = timeit(lambda: x.copy_(c)); print(f"copy 32 MB: {t:6.1f} us")
