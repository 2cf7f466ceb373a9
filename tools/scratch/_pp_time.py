import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vfmseg_amd import ops
M, N = 4096, 4096
for K in (1024, 4096):
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for cfg in (30,):
        for dbg in (0, 1, 5, 9, 13):
            ops.tune("gemm_cfg", cfg); ops.tune("pp_dbg", dbg)
            for _ in range(5): ops.gemm(a, b, c)
            torch.cuda.synchronize()
