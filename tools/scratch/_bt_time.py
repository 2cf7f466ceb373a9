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
for (M, N, K) in [(2048, 256, 2048), (2048, 256, 1536), (2048, 256, 1024), (2048, 256, 512), (2048, 1024, 2048), (2048, 4096, 1024), (8192, 512, 1024), (2048, 512, 256)]:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(K, N, device="cuda").bfloat16()
    c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ref = a.float() @ w.float()
    line = []
    for v in (0, 1, 2):
        ops.tune("gemm_bt64", v)
        ops.gemm(a, w, c, trans_b=True)
        err = ((c.float() - ref).abs().max() / ref.abs().max()).item()
        t = timeit(lambda: ops.gemm(a, w, c, trans_b=True))
        line.append(f"bt64={v}: {t:6.1f} us" + ("" if err < 2e-2 else f" ERR {err:.1e}"))
    print(f"M={M} N={N} K={K} | " + "  ".join(line), flush=True)
