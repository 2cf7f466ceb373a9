"""Idle-gap analysis of a rocprofv3 kernel trace: busy time vs span, largest gaps and what follows them."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
# steps are delimited by k_adamw
idx = [i for i, e in enumerate(ev) if e[2].startswith("k_adamw")]
print("adamw launches:", len(idx))
for a, b in zip(idx[:-1], idx[1:]):
    seg = ev[a + 1:b + 1]
    span = seg[-1][1] - seg[0][0]
    busy = sum(e[1] - e[0] for e in seg)
    gaps = []
    cur_end = seg[0][1]
    for s, e, n in seg[1:]:
        if s > cur_end:
            gaps.append((s - cur_end, n))
        cur_end = max(cur_end, e)
    idle = sum(g for g, _ in gaps)
    print(f"step: kernels={len(seg)} span={span/1e6:.2f}ms busy={busy/1e6:.2f}ms idle={idle/1e6:.2f}ms gaps>20us={sum(1 for g,_ in gaps if g>20000)}")
    big = sorted(gaps, reverse=True)[:12]
    for g, n in big:
        print(f"    gap {g/1e3:8.1f} us before {n[:70]}")
    import collections
    hist = collections.Counter()
    for g, _ in gaps:
        hist[min(int(g / 1000), 20)] += 1
    print("    gap histogram (us -> count):", sorted(hist.items())[:21])
