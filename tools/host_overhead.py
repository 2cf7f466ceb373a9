"""Host-side enqueue time of one train step vs its GPU time (are we launch-bound?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vfmseg_amd import functional as Fh

dev = torch.device("cuda", 0)
model, ow = bench.build(dev, 2)
Fh.manual_seed(1)
data = bench.make_batch(2, 0, 0, dev)
for _ in range(3):
    model.train_step(data, ow)
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(7):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.train_step(data, ow)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append(1e3 * (t1 - t0)), tot.append(1e3 * (t2 - t0))
print(f"VFMSEG_PLAN={os.environ.get('VFMSEG_PLAN', '1')}: host enqueue median {sorted(enq)[3]:.2f} ms (min {min(enq):.2f}), step from idle median {sorted(tot)[3]:.2f} ms", flush=True)
if os.environ.get("HOST_PROFILE", "0") != "1":
    sys.exit(0)
# phase split
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
model.train_step(data, ow)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr).sort_stats("cumulative")
st.print_stats(45)
