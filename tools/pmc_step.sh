#!/bin/bash
# SQ / GRBM counters of the kernels of a short train-step run (bench.py, 2 steps), per kernel family, in separate PMC passes
# (8 SQ slots per pass).  Run on the GPU box from the repo root:  tools/pmc_step.sh TAG  ->  gpurun_out/TAG_pmc_step.txt
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${tag}_pmc_step.txt
: > $out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmcs_$tag -o p -- python3 bench.py --steps 2 --warmup 1 --no-eval --no-cpu-baseline --no-roofline --no-parity-mode > /dev/null 2>&1
  python3 - "$set" >> $out <<'PY'
import csv, glob, sys, collections
f = glob.glob("gpurun_out/pmcs_*/**/*counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
grid = {}
for r in rows:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0] + " grid " + r.get("Grid_Size", "?")
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== pass:", sys.argv[1])
for k, d in sorted(acc.items(), key=lambda kv: -sum(next(iter(kv[1].values())))):
    if not any(t in k for t in ("k_gemm_w4", "k_gemm_ps", "k_attn_bf16")):
        continue
    print(" ", k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "dispatches", len(next(iter(d.values()))))
PY
  rm -rf gpurun_out/pmcs_$tag
done
