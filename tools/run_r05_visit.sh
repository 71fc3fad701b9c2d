#!/bin/bash
# Round-5 GPU visit (through gpurun): a list of steps, each a shell command run with a timeout, logs under gpurun_out/<tag>_<n>.log.
# usage: bash tools/run_r05_visit.sh <tag> "<cmd 1>" "<cmd 2>" ...      (a step that fails does not stop the visit)
set -u
T=$1; shift
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out
mkdir -p $O
i=0
for cmd in "$@"; do
  echo "=== [$i] $cmd"
  t0=$(date +%s)
  timeout 900 bash -c "$cmd" > $O/${T}_$i.log 2>&1
  echo "    rc=$? $(( $(date +%s) - t0 )) s"
  tail -n 12 $O/${T}_$i.log | cut -c1-400
  i=$((i+1))
done
