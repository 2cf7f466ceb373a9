"""Host time of a full train step by Python function (own time), backward run on the calling thread so that cProfile sees it."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from vfmseg_amd import functional as Fh
dev = torch.device("cuda", 0)
model, ow = bench.build(dev, 2)
Fh.manual_seed(1)
data = bench.make_batch(2, 0, 0, dev)
torch.autograd.set_multithreading_enabled(False)
for _ in range(3):
    model.train_step(data, ow)
torch.cuda.synchronize()
ts = []
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.train_step(data, ow)
    ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
print("enqueue (single-threaded autograd): %.2f ms" % (1e3 * min(ts)))
pr = cProfile.Profile()
pr.enable()
model.train_step(data, ow)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(45)
