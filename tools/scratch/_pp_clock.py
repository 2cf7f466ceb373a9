import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vfmseg_amd import ops
M, N = 4096, 4096
for K in (1024, 4096):
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    dbgf = torch.zeros(1, 16, dtype=torch.float32, device="cuda"); dbgbuf = dbgf.view(torch.int64).view(-1)
    for dbg in (16,):
        ops.tune("gemm_cfg", 30); ops.tune("pp_dbg", dbg)
        for _ in range(5): ops.gemm(a, b, c, aux=dbgf)
        torch.cuda.synchronize()
        v = dbgbuf.tolist()
        nph = K // 32 * 2
        for blk in (0, 1):
            cyc, rt = v[2 * blk], v[2 * blk + 1]
            us = rt / 100.0
            print(f"K={K} dbg={dbg} blk{blk}: cycles={cyc} realtime={us:.2f}us clock={cyc/us/1e3:.2f}GHz cycles/phase={cyc/nph:.0f}", flush=True)
