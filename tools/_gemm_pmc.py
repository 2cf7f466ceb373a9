import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vfmseg_amd import ops
cfg = int(os.environ.get("CFG", "16"))
M, N, K = [int(x) for x in os.environ.get("MNK", "4096,4096,1024").split(",")]
a = torch.randn(M, K, device='cuda').bfloat16(); b = torch.randn(N, K, device='cuda').bfloat16(); c = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
ops.tune("gemm_cfg", cfg)
for _ in range(5): ops.gemm(a, b, c)
torch.cuda.synchronize()
