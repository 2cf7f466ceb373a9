"""In-step A/B of kernel tuning knobs: ONE process, one model, the settings interleaved in rounds (cdna_hip_programming.md 5.4 rule 24:
separate runs of bench.py differ by +-3 % on the same box).  Each setting is a comma-separated list of vfm_tune KEY=INT pairs
("base" = nothing changed).

    python tools/ab_step.py base gemm_use_ps=0 "gemm_use_ps=0,gemm_use_pp=56" [--rounds 6 --steps 8]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from vfmseg_amd import functional as Fh, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("settings", nargs="+")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--workload", default="ms1024")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    model, ow = bench.build(dev, 2, workload=a.workload)
    Fh.manual_seed(1)
    data = bench.make_batch(2, 0, 0, dev, size=1024 if a.workload == "ms1024" else 512)
    parsed, envs = [], {}
    for s in a.settings:
        # KEY=INT: a vfm_tune knob; env:NAME=VALUE: an environment variable the Python side reads per call (e.g. env:VFMSEG_WGRAD_STREAM=0)
        items = [] if s == "base" else s.split(",")
        kv = [(x.split("=")[0], int(x.split("=")[1])) for x in items if not x.startswith("env:")]
        envs[s] = [(x[4:].split("=")[0], x[4:].split("=")[1]) for x in items if x.startswith("env:")]
        parsed.append((s, kv))
    keys = {k for _, kv in parsed for k, _ in kv}
    env_names = {n for v in envs.values() for n, _ in v}
    env_base = {n: os.environ.get(n) for n in env_names}
    defaults = {"gemm_use_ps": 0, "gemm_use_pp": 184, "ps_burst": 0, "gemm_cfg": -1, "pp_dbg": 0, "attn_fwd64": 0, "attn_xcd": 1, "gemm_deep_tail_k": 0,
                "attn_v2": 1, "gemm_nt_mb": 0, "gemm_deep_sep_k": 0, "attn_short_grid": 256, "gemm_use_192": 3, "gemm_w44_k": 1024}
    for _ in range(4):
        model.train_step(data, ow)
    torch.cuda.synchronize()
    times = {s: [] for s, _ in parsed}
    for r in range(a.rounds):
        for s, kv in parsed:
            for k in keys:
                ops.tune(k, defaults[k])
            for k, v in kv:
                ops.tune(k, v)
            for n in env_names:
                if env_base[n] is None:
                    os.environ.pop(n, None)
                else:
                    os.environ[n] = env_base[n]
            for n, v in envs[s]:
                os.environ[n] = v
            model.train_step(data, ow)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                model.train_step(data, ow)
            torch.cuda.synchronize()
            times[s].append(1e3 * (time.perf_counter() - t0) / a.steps)
    for k in keys:
        ops.tune(k, defaults[k])
    base = sorted(times[parsed[0][0]])[len(times[parsed[0][0]]) // 2]
    for s, _ in parsed:
        t = sorted(times[s])
        med = t[len(t) // 2]
        print(f"{s:40s} median {med:7.3f} ms/step  min {t[0]:7.3f}  max {t[-1]:7.3f}  images/s {4e3 / med * 0.5:7.2f}  vs first {100 * (base / med - 1):+5.2f} %", flush=True)


if __name__ == "__main__":
    main()
