"""Where the host time of the backbone forward / backward goes (cProfile by own time; bs-2 train shapes)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from vfmseg_amd import functional as Fh
dev = torch.device("cuda", 0)
model, ow = bench.build(dev, 2)
Fh.manual_seed(1)
data = bench.make_batch(2, 0, 0, dev)
for _ in range(3):
    model.train_step(data, ow)
torch.cuda.synchronize()
bb = model.backbone
vit = [m for m in bb.modules() if hasattr(m, "_engine")][0]
eng = vit.engine()
x = torch.randn(4, 3, 512, 512, device=dev)
for name in ("forward", "backward"):
    ts = []
    pr = cProfile.Profile()
    for it in range(4):
        xcat, hw, ctx = eng.forward([(x, None)], True, 7)
        dx = torch.randn_like(xcat)
        torch.cuda.synchronize()
        if name == "forward":
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if it == 3: pr.enable()
            eng.forward([(x, None)], True, 7)
            if it == 3: pr.disable()
            ts.append(time.perf_counter() - t0)
        else:
            t0 = time.perf_counter()
            if it == 3: pr.enable()
            eng.backward(ctx, dx)
            if it == 3: pr.disable()
            ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    print(f"== backbone {name}: host enqueue {1e3 * min(ts[:3]):.2f} ms (unprofiled)")
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
