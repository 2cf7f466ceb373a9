#!/bin/bash
# HBM-side traffic of the dominant GEMM kernel inside the train step: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)
# over a short bench.py run, as MI355X_MICROARCH.md prescribes; tools/pmc_traffic.py applies the gfx950 corrections.
# Run on the GPU box from the repo root:  tools/pmc_gemm_traffic.sh TAG   ->  gpurun_out/TAG_pmc_gemm_traffic.json
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_$c -o p -- python3 bench.py --steps 2 --warmup 1 --no-eval --no-cpu-baseline --no-roofline --no-parity-mode > gpurun_out/pmc_$c.log 2>&1
done
f1=$(find gpurun_out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
f2=$(find gpurun_out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py "$f1" "$f2" gpurun_out/${tag}_pmc_gemm_traffic.json
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
