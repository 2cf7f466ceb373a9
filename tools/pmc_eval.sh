#!/bin/bash
# SQ counters of the MFMA kernels of the eval paths (tools/bench_eval.py dinov2|sam): MFMA occupancy and issue mix per kernel class.
# Run on the GPU box from the repo root:  tools/pmc_eval.sh TAG dinov2|sam  ->  gpurun_out/TAG_pmc_eval.txt
tag=${1:-r03}; which=${2:-dinov2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${tag}_pmc_eval.txt
: > $out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "GRBM_COUNT GRBM_GUI_ACTIVE"; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmce_$tag -o p -- python3 tools/bench_eval.py $which > /dev/null 2>&1
  python3 - "$set" >> $out <<'PY'
import csv, glob, sys, collections
f = glob.glob("gpurun_out/pmce_*/**/*counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"].split("(")[0] + " grid " + r.get("Grid_Size", "?")
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== pass:", sys.argv[1])
for k, d in sorted(acc.items(), key=lambda kv: -len(next(iter(kv[1].values())))):
    if not any(s in k for s in ("k_gemm_w4", "k_attn_bf16", "k_sam_flash", "k_gemm_pp")):
        continue
    n = len(next(iter(d.values())))
    if n < 20:
        continue
    print(" ", k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "dispatches", n)
PY
  rm -rf gpurun_out/pmce_$tag
done
cat $out
