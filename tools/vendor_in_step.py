"""The vendor GEMM INSIDE the train step (yardstick only, never a dependency of the product).  The cold-buffer microbench
(tools/bench_gemm_step.py) hands every launch operands that nothing has touched; in the step the A operand was written by the previous
kernel.  Here every backbone GEMM of the step (qkv / proj / fc1 / fc2 forward and their input gradients: 8 per block) is DOUBLED by a
torch.matmul (hipBLASLt) of the same operands into a scratch output - plain product, no epilogue - issued right before ("pre") or right
after ("post") ours, and the step is timed with and without the doubles, interleaved in one process:
    vendor's in-step time per step ~= step(with doubles) - step(base)          vs ours: the kernel time of the same launches (step profile)
Launch plans are off here (VFMSEG_PLAN=0: the doubles are issued from Python, so ours must be too).
    python tools/vendor_in_step.py [--rounds 5 --steps 6]"""
import argparse
import os
import sys
import time

os.environ["VFMSEG_PLAN"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from vfmseg_amd import backbones as BB, functional as Fh  # noqa: E402

MODE = ["off"]
SCRATCH = {}
COUNT = [0]


def scratch(m, n, dev):
    key = (m, n)
    if key not in SCRATCH:
        SCRATCH[key] = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    return SCRATCH[key]


def wrap(orig, which):
    def f(self, a, out, **epi):
        w = self.w if which == "fwd" else self.wt
        big = a.shape[0] >= 4000 and w is not None and w.shape[0] >= 1000 and a.dtype == torch.bfloat16
        if MODE[0] == "pre" and big:
            torch.matmul(a, w.t(), out=scratch(a.shape[0], w.shape[0], a.device))
            COUNT[0] += 1
        r = orig(self, a, out, **epi)
        if MODE[0] == "post" and big:
            torch.matmul(a, w.t(), out=scratch(a.shape[0], w.shape[0], a.device))
            COUNT[0] += 1
        return r
    return f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=6)
    a = ap.parse_args()
    BB.Packed.fwd = wrap(BB.Packed.fwd, "fwd")
    BB.Packed.dgrad = wrap(BB.Packed.dgrad, "dgrad")
    dev = torch.device("cuda", 0)
    model, ow = bench.build(dev, 2)
    Fh.manual_seed(1)
    data = bench.make_batch(2, 0, 0, dev)
    for m in ("off", "pre", "post"):
        MODE[0] = m
        for _ in range(2):
            model.train_step(data, ow)
    torch.cuda.synchronize()
    times = {"off": [], "pre": [], "post": []}
    for r in range(a.rounds):
        for m in ("off", "pre", "post"):
            MODE[0] = m
            model.train_step(data, ow)
            torch.cuda.synchronize()
            COUNT[0] = 0
            t0 = time.perf_counter()
            for _ in range(a.steps):
                model.train_step(data, ow)
            torch.cuda.synchronize()
            times[m].append(1e3 * (time.perf_counter() - t0) / a.steps)
            n = COUNT[0] // a.steps
    med = {m: sorted(t)[len(t) // 2] for m, t in times.items()}
    print(f"vendor doubles per step: {n} (the wide backbone GEMMs: M >= 4000, N >= 1000)")
    print(f"step, launches from Python, no doubles : {med['off']:.3f} ms")
    for m in ("pre", "post"):
        print(f"step + vendor doubles issued {m:4s} ours : {med[m]:.3f} ms  ->  vendor GEMMs in the step ~ {med[m] - med['off']:.3f} ms per step")


if __name__ == "__main__":
    main()
