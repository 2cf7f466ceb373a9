import sys, os
sys.path.insert(0, os.getcwd())
import torch
from vfmseg_amd import ops
for (M,N,K) in [(4,4096,1024),(4,1024,4096),(4,3072,1088),(4,1024,1024)]:
    a=torch.randn(M,K,device='cuda').bfloat16(); b=torch.randn(N,K,device='cuda').bfloat16(); c=torch.empty(M,N,device='cuda')
    for _ in range(5): ops.gemm(a,b,c)
torch.cuda.synchronize()
