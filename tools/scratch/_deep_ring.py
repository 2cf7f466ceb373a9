import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vfmseg_amd import ops
def t(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
# cold-ish: rotate over several operand sets so that A / C are not MALL-resident
def bench(M, N, K, cfgs, sets=6, res=True):
    As = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(sets)]
    B = torch.randn(N, K, device="cuda").bfloat16()
    Cs = [torch.empty(M, N, device="cuda") for _ in range(sets)]
    Rs = [torch.randn(M, N, device="cuda") for _ in range(sets)]
    bias = torch.randn(N, device="cuda"); cs = torch.randn(N, device="cuda")
    out = []
    for cfg in cfgs:
        ops.tune("gemm_cfg", cfg)
        i = [0]
        def f():
            j = i[0] % sets; i[0] += 1
            ops.gemm(As[j], B, Cs[j], bias=bias, colscale=cs, residual=Rs[j] if res else None)
        us = t(f)
        out.append(f"c{cfg}: {us:6.1f} us {2.0*M*N*K/us/1e6:5.0f} TF")
    ops.tune("gemm_cfg", -1)
    print(f"M{M} N{N} K{K}: " + " | ".join(out))
for (M, N, K) in [(4096, 1024, 4096), (4096, 1024, 1024), (4100, 1024, 4096), (4100, 1024, 1024)]:
    bench(M, N, K, [34, 35, 36, -1])
