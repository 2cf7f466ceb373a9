"""GPU wall time of the decoder heads inside a train step (no profiler): HIP events behind the backbone's forward and in front of its
backward, median over steps - against the kernel-time sum of the same region in the rocprofv3 trace (profiles/r04_final_step_profile.txt)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from vfmseg_amd import backbones, functional as Fh

dev = torch.device("cuda", 0)
model, ow = bench.build(dev, 2)
Fh.manual_seed(1)
data = bench.make_batch(2, 0, 0, dev)
ev = {}
orig_tokens = model._tokens


def tokens(jobs):
    out = orig_tokens(jobs)
    ev["fwd_done"] = torch.cuda.Event(enable_timing=True)
    ev["fwd_done"].record()
    return out


model._tokens = tokens
orig_bwd = backbones._BackboneFn.backward


def bwd(ctx, *g):
    ev["bwd_start"] = torch.cuda.Event(enable_timing=True)
    ev["bwd_start"].record()
    return orig_bwd(ctx, *g)


backbones._BackboneFn.backward = staticmethod(bwd)
for _ in range(5):
    model.train_step(data, ow)
torch.cuda.synchronize()
hs, st = [], []
for _ in range(15):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    model.train_step(data, ow)
    e1.record()
    torch.cuda.synchronize()
    hs.append(ev["fwd_done"].elapsed_time(ev["bwd_start"]))
    st.append(e0.elapsed_time(e1))
hs.sort(), st.sort()
print(f"heads (forward + loss + backward) GPU wall: median {hs[len(hs) // 2]:.3f} ms (min {hs[0]:.3f}); step median {st[len(st) // 2]:.3f} ms")
