#!/bin/bash
# Kernel classes (name, grid) of one other-backbone train step: rocprofv3 kernel trace of tools/bench_train_models.py.  usage: tools/train_classes.sh TAG sam|clip|eva
tag=$1; export ONLY=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 tools/bench_train_models.py > gpurun_out/prof_$tag.log 2>&1
f=$(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1)
python3 - "$f" > gpurun_out/${tag}_train_classes.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"].split("(")[0][:70], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]))
    a = acc[k]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in acc.values())
print("total kernel time %.1f ms" % (tot / 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%6d x %8.1f us = %9.1f us (%4.1f%%)  blocks %5d x %-3d %s" % (v[0], v[1] / v[0], v[1], 100 * v[1] / tot, k[1], k[2], k[0]))
PY
rm -rf gpurun_out/prof_$tag
grep images gpurun_out/prof_$tag.log
