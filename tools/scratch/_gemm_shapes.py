"""Per-signature timing of every bf16 GEMM of one train step under candidate tile configs (diagnostic)."""
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
for _ in range(2):
    model.train_step(data, ow)
orig = ops.gemm
seen = collections.OrderedDict()
def hook(a, b, c, **kw):
    if a.dtype == torch.bfloat16:
        c2d = c[0] if c.dim() == 3 else c
        M, N = c2d.shape
        K = a.shape[-1] if not kw.get("trans_a") else a.shape[-2]
        key = (M, N, K, a.dim(), str(c.dtype)[6:], bool(kw.get("trans_a")), bool(kw.get("trans_b")), kw.get("ep_mode", 0),
               kw.get("residual") is not None, kw.get("bias") is not None, kw.get("aux") is not None)
        if key not in seen:
            seen[key] = [0, (a, b, c, dict(kw))]
        seen[key][0] += 1
    return orig(a, b, c, **kw)
ops.gemm = hook
import vfmseg_amd.backbones as bb, vfmseg_amd.functional as fn
bb.ops.gemm = hook; fn.ops.gemm = hook
model.train_step(data, ow)
torch.cuda.synchronize()
ops.gemm = orig; bb.ops.gemm = orig; fn.ops.gemm = orig
def timeit(f, iters=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
cfgs = [int(x) for x in os.environ.get("CFGS", "-1,10,18,19,26,27,28,17").split(",")]
tot = 0.0
for key, (n, (a, b, c, kw)) in seen.items():
    line = []
    small = not key[5] and not key[6]
    for cfg in (cfgs if small else [-1]):
        ops.tune("gemm_cfg", cfg)
        try:
            t = timeit(lambda: orig(a, b, c, **kw))
            line.append(f"c{cfg}:{t:6.1f}")
        except Exception as ex:
            line.append(f"c{cfg}: fail")
        if cfg == -1: tot += t * n
    ops.tune("gemm_cfg", -1)
    print(f"{n:3d}x M={key[0]:6d} N={key[1]:5d} K={key[2]:5d} nd{key[3]} {key[4]:8s} tA{int(key[5])} tB{int(key[6])} ep{key[7]} r{int(key[8])} b{int(key[9])} x{int(key[10])} | " + " ".join(line), flush=True)
print("sum default us/step:", tot)
