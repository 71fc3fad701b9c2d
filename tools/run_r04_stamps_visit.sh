#!/bin/bash
# Round-4 visit: per-tile stamps of the persistent ping-pong GEMM under its switches, then the mask band's new tests.
# usage: bash tools/run_r04_stamps_visit.sh <tag> "<pytest args>" "ENV.." ["ENV.." ...]
set -u
T=$1; shift
PYT=$1; shift
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out
mkdir -p $O
: > $O/${T}_gemm8_phase_cycles.log
for cfg in "$@"; do
  env $cfg timeout 300 python tools/gemm_stamps.py >> $O/${T}_gemm8_phase_cycles.log 2>&1
done
cat $O/${T}_gemm8_phase_cycles.log
if [ -n "$PYT" ]; then
  timeout 1500 python -m pytest $PYT -x -q -m gpu > $O/${T}_pytest.log 2>&1
  tail -15 $O/${T}_pytest.log
fi
