"""Launches the backbone's four big GEMM shapes (default dispatch) a few times each: target of the rocprofv3 --pmc passes that
give HBM-side bytes per launch of the dominant kernel (tools/pmc_traffic.py parses the result)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vfmseg_amd import ops
cfg = int(os.environ.get("CFG", "-1"))
ops.tune("gemm_cfg", cfg)
shapes = [(4096, 3072, 1088), (4096, 1024, 1024), (4096, 4096, 1024), (4096, 1024, 4096)]
if os.environ.get("MNK"):
    shapes = [tuple(int(x) for x in os.environ["MNK"].split(","))]
for (M, N, K) in shapes:
    a = torch.randn(M, K, device='cuda').bfloat16(); b = torch.randn(N, K, device='cuda').bfloat16()
    c = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
    for _ in range(4): ops.gemm(a, b, c)
torch.cuda.synchronize()
