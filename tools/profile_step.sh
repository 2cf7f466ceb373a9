#!/bin/bash
# rocprofv3 kernel trace of a short bench run -> per-step summary (tools/step_profile.py).  usage: tools/profile_step.sh TAG [bench args...]
# Run on the GPU box (gpurun) from the repo root.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 bench.py --steps 4 --warmup 2 --no-eval --no-cpu-baseline --no-roofline --no-parity-mode "$@" > gpurun_out/prof_$tag.log 2>&1
f=$(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1)
python tools/step_profile.py $f --seq > gpurun_out/${tag}_step_profile.txt
find gpurun_out/prof_$tag -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_kernel_stats.csv \;
rm -rf gpurun_out/prof_$tag
head -75 gpurun_out/${tag}_step_profile.txt
