#!/bin/bash
# One GPU-box session (run through gpurun from the repo root): each step under its own timeout, logs under gpurun_out/.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
step() {  # name, timeout, command...
  local name=$1 t=$2; shift 2
  echo "=== $name" | tee -a $out/session.log
  timeout -k 10 $t "$@" > $out/$name.log 2>&1
  local rc=$?
  echo "=== $name rc=$rc" | tee -a $out/session.log
  tail -5 $out/$name.log | cut -c1-400
  return $rc
}
"$@"
