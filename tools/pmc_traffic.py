"""Per-launch HBM-side traffic of the GEMM kernels inside the train step, from two SEPARATE rocprofv3 --pmc passes over bench.py
(tools/pmc_gemm_traffic.sh):

    python tools/pmc_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv OUT.json

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide streaming read, so it is doubled
(MI355X_MICROARCH.md, HBM section).  The counters sit on the fabric side of L2, so Infinity-Cache hits are included: this is
L2-miss traffic, an upper bound of the HBM bytes.  `mean_bytes_per_launch` of the dominant kernel is what bench.py reports as
roofline.traffic; its algorithmic operand + result bytes are listed beside it."""
import collections
import csv
import json
import sys


def per_kernel(path, counter, scale):
    fam = collections.defaultdict(lambda: [0, 0.0])
    per = collections.defaultdict(float)
    name = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        per[r["Dispatch_Id"]] += float(r["Counter_Value"])
        name[r["Dispatch_Id"]] = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    for k, v in per.items():
        fam[name[k]][0] += 1
        fam[name[k]][1] += v * scale
    return fam


fetch = per_kernel(sys.argv[1], "FETCH_SIZE", 2 * 1024)
write = per_kernel(sys.argv[2], "WRITE_SIZE", 1024)
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) over bench.py --steps 2 --warmup 1; "
                 "FETCH_SIZE doubled (gfx950), KiB -> bytes; fabric side of L2 (Infinity-Cache hits included: upper bound of HBM bytes)",
       "kernels": []}
rows = []
for n in set(fetch) | set(write):
    cf, fb = fetch.get(n, [0, 0.0])
    cw, wb = write.get(n, [0, 0.0])
    cnt = max(cf, cw, 1)
    rows.append((fb + wb, n, cnt, fb / max(cf, 1), wb / max(cw, 1)))
rows.sort(reverse=True)
for tot, n, cnt, fpl, wpl in rows[:12]:
    out["kernels"].append({"kernel": n, "launches": cnt, "fetch_bytes_per_launch": fpl, "write_bytes_per_launch": wpl,
                           "bytes_per_launch": fpl + wpl})
    print(f"{cnt:6d}  fetch {fpl / 1e6:8.2f} MB  write {wpl / 1e6:8.2f} MB  {n[:90]}")
# the dominant family = the backbone's forward + input-gradient GEMM launches: the ring kernel in its 128 x 128 and 256 x 256 tile forms
# (and the persistent two-accumulator kernel when vfm_tune gemm_use_ps is on); launch-weighted mean
dom = [k for k in out["kernels"] if "k_gemm_w4<true, 2, 1, 2, 4, 2, 0" in k["kernel"] or "k_gemm_w4<true, 4, 2, 2, 4, 2, 0" in k["kernel"] or "k_gemm_ps<" in k["kernel"]]
if dom:
    nl = sum(k["launches"] for k in dom)
    out["dominant"] = dom
    out["mean_bytes_per_launch"] = sum(k["bytes_per_launch"] * k["launches"] for k in dom) / nl
    # algorithmic bytes of the same launches (bs 2: M = 4100 token rows, bf16 operands and results; per train step 24 layers x
    # {qkv+LoRA fwd [4100x3072x1088], qkv dgrad [4100x1088x3072], proj fwd+dgrad [4100x1024x1024] x2 (+fp32 residual read/write fwd),
    #  fc1 fwd [4100x4096x1024] (two bf16 results), fc2 dgrad [4100x4096x1024] (+bf16 aux read), fc2 fwd / fc1 dgrad [4100x1024x4096] x2})
    M = 4100

    def gemm_bytes(n, k, extra=0.0):
        return 2.0 * (M * k + n * k + M * n) + extra
    per_layer = [gemm_bytes(3072, 1088), gemm_bytes(1088, 3072), gemm_bytes(1024, 1024, 8.0 * M * 1024), gemm_bytes(1024, 1024),
                 gemm_bytes(4096, 1024, 2.0 * M * 4096), gemm_bytes(4096, 1024, 2.0 * M * 4096), gemm_bytes(1024, 4096, 8.0 * M * 1024),
                 gemm_bytes(1024, 4096)]
    out["algorithmic_mean_bytes_per_launch"] = sum(per_layer) / len(per_layer)
    out["ratio"] = out["mean_bytes_per_launch"] / out["algorithmic_mean_bytes_per_launch"]
json.dump(out, open(sys.argv[3], "w"), indent=1)
