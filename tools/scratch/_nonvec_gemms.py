"""Lists the GEMM calls of one train step whose epilogue cannot take the vector path (diagnostic)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from vfmseg_amd import ops, functional as Fh
from vfmseg_amd.precision import set_compute_dtype
set_compute_dtype("bf16")
dev = torch.device("cuda", 0)
model, ow = bench.build(dev, 2)
Fh.manual_seed(1)
data = bench.make_batch(2, 0, 0, dev)
orig = ops.gemm
seen = collections.Counter()
def hook(a, b, c, **kw):
    if a.dtype == torch.bfloat16:
        c2d = c[0] if c.dim() == 3 else c
        M, N = c2d.shape
        K = a.shape[-1] if not kw.get("trans_a") else a.shape[-2]
        al = lambda t: t is None or t.data_ptr() % 16 == 0
        bias, res, aux, c2 = kw.get("bias"), kw.get("residual"), kw.get("aux"), kw.get("c2")
        ok = N % 4 == 0 and c2d.stride(0) % 4 == 0 and al(c) and al(bias) and al(res) and al(aux) and al(c2) and al(kw.get("colscale"))
        if res is not None: ok = ok and (res[0] if res.dim() == 3 else res).stride(0) % 4 == 0
        if aux is not None: ok = ok and aux.stride(0) % 4 == 0
        if not ok:
            seen[(M, N, K, str(c.dtype), c2d.stride(0), kw.get("ep_mode", 0), bool(kw.get("trans_b")), res is not None, bias is not None)] += 1
    return orig(a, b, c, **kw)
ops.gemm = hook
import vfmseg_amd.functional, vfmseg_amd.heads, vfmseg_amd.backbones
for mod in (vfmseg_amd.functional, vfmseg_amd.heads, vfmseg_amd.backbones):
    if hasattr(mod, "ops"): pass
model.train_step(data, ow)
torch.cuda.synchronize()
for k, v in sorted(seen.items(), key=lambda x: -x[1]):
    print(v, k)
