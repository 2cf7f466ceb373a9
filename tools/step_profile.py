#!/usr/bin/env python
"""Reads a rocprofv3 --kernel-trace CSV of `bench.py` and prints, for the LAST train step (between the last two k_adamw
launches): wall / busy time, per-kernel-family totals, and every (kernel, grid, workgroup) class of the MFMA kernels with its
launch count and mean duration - the per-shape in-step numbers DESIGN.md quotes.

    python tools/step_profile.py <kernel_trace.csv> [--seq]      (--seq also prints the launch sequence with gaps)"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "k_adamw" in r["Kernel_Name"]]
    if len(idx) >= 2:
        step = rows[idx[-2] + 1:idx[-1] + 1]
    else:
        step = rows
    t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
    dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])   # noqa: E731
    busy = sum(dur(r) for r in step)
    print(f"step wall {(t1 - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, launches {len(step)}")
    fam, cnt = collections.Counter(), collections.Counter()
    cls = collections.defaultdict(list)
    for r in step:
        n = r["Kernel_Name"]
        key = n.replace("(anonymous namespace)::", "").split("(")[0][:70]
        fam[key] += dur(r)
        cnt[key] += 1
        if "gemm" in n or "attn" in n:
            g = tuple(int(r.get(k, 0) or 0) for k in ("Grid_Size_X", "Grid_Size_Y", "Workgroup_Size_X"))
            cls[(key, g, int(r.get("LDS_Block_Size", 0) or 0))].append(dur(r))
    mf = sum(v for k, v in fam.items() if "gemm" in k or "attn" in k)
    print(f"MFMA kernels {mf / 1e6:.3f} ms, everything else {(busy - mf) / 1e6:.3f} ms")
    for k, v in fam.most_common(70):
        print(f"{v / 1e3:9.1f} us {cnt[k]:5d}  {k}")
    print("---- MFMA kernel classes (kernel, (grid_x [threads], grid_y, wg), lds): launches, mean us, total us")
    for (k, g, lds), v in sorted(cls.items(), key=lambda kv: -sum(kv[1])):
        print(f"{len(v):4d} x {sum(v) / len(v) / 1e3:8.1f} us = {sum(v) / 1e3:8.1f}  blocks {g[0] // max(g[2], 1):5d} x{g[1]:2d} wg {g[2]:4d} lds {lds:6d}  {k}")
    if "--seq" in sys.argv:
        prev = None
        for r in step:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            gap = (s - prev) / 1e3 if prev else 0
            print(f"{(s - t0) / 1e3:10.1f} +{gap:6.1f} {(e - s) / 1e3:8.1f}  {r['Kernel_Name'][:100]}")
            prev = e


if __name__ == "__main__":
    main()
