import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vfmseg_amd import ops
DEV = "cuda"
rows, C = 133, 1024
g = torch.Generator().manual_seed(1)
x = (torch.randn(rows, C, generator=g) * 2 + 0.5).to(DEV)
w, b = (torch.randn(C, generator=g) * 0.1 + 1).to(DEV), (torch.randn(C, generator=g) * 0.1).to(DEV)
y0 = torch.empty(rows, C, dtype=torch.bfloat16, device=DEV); st0 = torch.empty(rows, 2, device=DEV)
ops.layernorm_fwd(x, w, b, 1e-6, y0, st0)
m0 = torch.empty(rows, C, dtype=torch.bfloat16, device=DEV); ops.dropout_mask(m0, 0.1, 1234, offset=5 * rows * C)
d0 = torch.empty_like(y0); ops.mul_mask(y0, m0, d0)
big = torch.zeros(rows, C + 64, dtype=torch.bfloat16, device=DEV); st1 = torch.empty(rows, 2, device=DEV)
m1, d1 = torch.empty_like(m0), torch.empty_like(d0)
ops.layernorm_dropout_fwd(x, w, b, 1e-6, big[:, :C], st1, d1, m1, 0.1, 1234, offset=5 * rows * C)
print("y", torch.equal(big[:, :C], y0), "st", torch.equal(st1, st0), "mask", torch.equal(m1, m0), "drop", torch.equal(d1.view(torch.int16), d0.view(torch.int16)))
print((m1 != m0).sum().item(), m0.unique(), m1.unique(), (big[:, :C] != y0).sum().item())
