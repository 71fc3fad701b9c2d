#!/bin/bash
# Final visit of a round (through gpurun): bash tools/run_r06_final_visit.sh <tag>
#   1. tools/run_final_profiles.sh <tag>: the default bench line, rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE / GRBM_GUI_ACTIVE passes of the
#      sequential command, the bench line with every leg
#   2. enqueue layouts with the socket power beside (tools/overlap_bench.py), power / clock of the dominant kernel shapes (tools/power_clock.py),
#      per-tile stamps of the ping-pong GEMM, the batch-1 latency breakdown
#   3. the whole GPU suite with -s (the parity figures the documents quote) and the smoke entry
set -u
T=${1:-r06z}
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out
mkdir -p $O
bash tools/run_final_profiles.sh $T
timeout 400 python tools/overlap_bench.py --steps 6 --layouts seq,2way,flow1,flow2,depth1,depth2,3way,4way > $O/${T}_overlap_layouts_power.txt 2>&1
timeout 400 python tools/power_clock.py > $O/${T}_power_clock.txt 2>&1
timeout 300 python tools/gemm_stamps.py > $O/${T}_gemm8_phase_cycles.log 2>&1
timeout 120 python tools/latency_breakdown.py 1 > $O/${T}_latency_batch1.txt 2>&1
timeout 1700 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/${T}_pytest_gpu.log 2>&1
grep -E "passed|failed" $O/${T}_pytest_gpu.log | tail -2
grep -E "relmax|passed|failed" $O/${T}_pytest_gpu.log > $O/${T}_pytest_gpu_parity.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
