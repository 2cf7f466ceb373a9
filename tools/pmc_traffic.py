"""Per-launch HBM-side traffic of the dominant kernel from a rocprofv3 --pmc FETCH_SIZE WRITE_SIZE pass over bench.py.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE -d OUT -o p --output-format csv -- python bench.py ...
    python tools/pmc_traffic.py OUT/p_counter_collection.csv profiles/r01_pmc_gemm_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide streaming read, so it is
doubled (MI355X_MICROARCH.md, HBM section).  Infinity-Cache hits are included in these counters (they sit on the fabric side
of L2), so this is L2-miss traffic, an upper bound of the HBM bytes."""
import collections
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
per = collections.defaultdict(lambda: collections.defaultdict(float))
name = {}
for r in rows:
    k = r["Dispatch_Id"]
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    name[k] = r["Kernel_Name"]
fam = collections.defaultdict(lambda: [0, 0.0, 0.0])
for k, c in per.items():
    n = name[k].split("(")[0]
    f = fam[n]
    f[0] += 1
    f[1] += c.get("FETCH_SIZE", 0.0) * 2 * 1024
    f[2] += c.get("WRITE_SIZE", 0.0) * 1024
top = sorted(fam.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))
out = {"source": "rocprofv3 --pmc FETCH_SIZE WRITE_SIZE over bench.py (separate pass); FETCH_SIZE doubled (gfx950), KiB -> bytes",
       "kernels": []}
out["per_dispatch"] = []
for k in sorted(per, key=int):
    if "gemm" in name[k]:
        out["per_dispatch"].append({"kernel": name[k].split("(")[0][:60], "fetch_bytes": per[k].get("FETCH_SIZE", 0.0) * 2048,
                                    "write_bytes": per[k].get("WRITE_SIZE", 0.0) * 1024})
for n, (cnt, fb, wb) in top[:12]:
    out["kernels"].append({"kernel": n, "launches": cnt, "fetch_bytes_per_launch": fb / cnt, "write_bytes_per_launch": wb / cnt,
                           "bytes_per_launch": (fb + wb) / cnt})
    print(f"{cnt:6d}  fetch {fb / cnt / 1e6:8.2f} MB  write {wb / cnt / 1e6:8.2f} MB  {n[:90]}")
dom = [k for k in out["kernels"] if "k_gemm_w4" in k["kernel"] or "k_gemm_bf16<128, 128, 2, 4, 2, true, false, false>" in k["kernel"]]
if dom:
    out["dominant"] = dom[0]
json.dump(out, open(sys.argv[2], "w"), indent=1)
