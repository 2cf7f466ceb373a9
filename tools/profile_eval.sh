#!/bin/bash
# rocprofv3 kernel stats of the eval paths (tools/bench_eval.py).  usage: tools/profile_eval.sh TAG [dinov2|sam]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 tools/bench_eval.py "$@" > gpurun_out/prof_$tag.log 2>&1
find gpurun_out/prof_$tag -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_kernel_stats.csv \;
rm -rf gpurun_out/prof_$tag
tail -3 gpurun_out/prof_$tag.log
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${tag}_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print(f'{float(r["TotalDurationNs"])/1e3:10.1f} us {int(r["Calls"]):6d} {float(r["Percentage"]):5.1f}%  {r["Name"][:90]}')
print("total", tot/1e6, "ms over all iterations")
PY
