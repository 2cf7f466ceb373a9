"""Which Python lines launch ATen fill / copy / elementwise kernels inside one train step: the tensor methods and factory functions that
launch such kernels are wrapped for one step and every call on a CUDA tensor is attributed to its innermost vfmseg_amd / bench frame.
Every such launch is glue that a pre-zeroed workspace, a slice or a fused kernel could remove.
    python tools/aten_sites.py"""
import collections
import os
import sys
import traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from vfmseg_amd import functional as Fh

dev = torch.device("cuda", 0)
EVAL = len(sys.argv) > 1 and sys.argv[1] == "eval"   # `python tools/aten_sites.py eval`: one 1024^2 ms_slide prediction instead of a train step
if EVAL:
    from vfmseg_amd import presets
    from vfmseg_amd.registry import MODELS
    from vfmseg_amd.synth import synth_image, synth_like
    cfg = presets.dinov2_ms_masked()
    cfg["test_cfg"]["conf"] = 2.0
    model = MODELS.build(cfg)
    model.load_state_dict(synth_like(model.state_dict()), strict=False)
    model = model.to(dev).eval()
    img = synth_image(1, 1024, seed=77).to(dev)

    def one():
        with torch.no_grad():
            model.predict(img)
else:
    model, ow = bench.build(dev, 2)
    Fh.manual_seed(1)
    data = bench.make_batch(2, 0, 0, dev)

    def one():
        model.train_step(data, ow)
for _ in range(3):
    one()
torch.cuda.synchronize()
sites = collections.Counter()


def site(name, t):
    if not (torch.is_tensor(t) and t.is_cuda):
        return
    fr = next((f for f in reversed(traceback.extract_stack()[:-2]) if "vfmseg_amd/" in f.filename or f.filename.endswith("bench.py")), None)
    where = f"{fr.filename.split('repo/')[-1]}:{fr.lineno} {fr.line[:70]}" if fr else "?"
    sites[(name, where, str(tuple(t.shape)), str(t.dtype).replace('torch.', ''))] += 1


def wrap_method(name):
    orig = getattr(torch.Tensor, name)

    def w(self, *a, **k):
        r = orig(self, *a, **k)
        site("Tensor." + name, r if torch.is_tensor(r) else self)
        return r
    setattr(torch.Tensor, name, w)
    return orig


def wrap_fn(name):
    orig = getattr(torch, name)

    def w(*a, **k):
        r = orig(*a, **k)
        site("torch." + name, r)
        return r
    setattr(torch, name, w)
    return orig


METHODS = ["zero_", "fill_", "copy_", "clone", "contiguous", "add_", "mul_", "add", "mul", "sub", "float", "to", "bfloat16", "sum", "mean", "__add__",
           "__mul__", "__gt__", "__sub__", "__truediv__", "div", "masked_fill_", "index_select", "__getitem__", "tolist", "item", "cpu", "cuda"]
FNS = ["zeros", "ones", "full", "cat", "stack", "zeros_like", "ones_like", "where", "empty_like", "tensor", "as_tensor", "from_numpy"]
saved = {m: wrap_method(m) for m in METHODS if hasattr(torch.Tensor, m)}
savedf = {f: wrap_fn(f) for f in FNS}
one()
torch.cuda.synchronize()
for m, o in saved.items():
    setattr(torch.Tensor, m, o)
for f, o in savedf.items():
    setattr(torch, f, o)
skip = ("Tensor.__getitem__", "Tensor.to", "torch.empty_like")   # views / no-ops mostly: listed last
for (name, where, shape, dt), n in sorted(sites.items(), key=lambda kv: (kv[0][0] in skip, -kv[1])):
    print(f"{n:4d}  {name:20s} {shape:24s} {dt:9s} {where}")
