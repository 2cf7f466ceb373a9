#!/bin/bash
# PMC passes over the attention kernels (forward, dQ, dK/dV at the train-step shape: 4 images x 16 heads x 1025 tokens, d = 64).
# Separate passes (8 SQ slots per pass); run on the GPU box from the repo root:  tools/pmc_attention.sh TAG
tag=${1:-attn}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${tag}_pmc_attention.txt
: > $out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_$tag -o p -- python3 tools/scratch/_attn_pmc.py > /dev/null 2>&1
  python3 - "$set" >> $out <<'PY'
import csv, glob, sys, collections
f = glob.glob("gpurun_out/pmc_*/**/*counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== pass:", sys.argv[1])
for k, d in acc.items():
    if "attn" not in k: continue
    print(" ", k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "dispatches", len(next(iter(d.values()))))
PY
  rm -rf gpurun_out/pmc_$tag
done
cat $out
