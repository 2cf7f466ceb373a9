#!/bin/bash
# SQ counters of the SAM flash kernels in one SAM-H train step (tools/bench_train_models.py, ONLY=sam): issue mix and MFMA occupancy.
# Run on the GPU box from the repo root:  tools/pmc_sam.sh TAG  ->  gpurun_out/TAG_pmc_sam.txt
tag=${1:-r02}
export ONLY=sam
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${tag}_pmc_sam.txt
: > $out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES"; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmcs_$tag -o p -- python3 tools/bench_train_models.py > /dev/null 2>&1
  python3 - "$set" >> $out <<'PY'
import csv, glob, sys, collections
f = glob.glob("gpurun_out/pmcs_*/**/*counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"].split("(")[0] + " grid " + r.get("Grid_Size", "?")
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== pass:", sys.argv[1])
for k, d in sorted(acc.items()):
    if "k_sam_flash" not in k:
        continue
    print(" ", k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "dispatches", len(next(iter(d.values()))))
PY
  rm -rf gpurun_out/pmcs_$tag
done
cat $out
