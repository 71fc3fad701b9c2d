#!/bin/bash
# Final visit of round 5 (through gpurun): bash tools/run_r05_final_visit.sh <tag>
#   1. tools/run_final_profiles.sh <tag>: rocprofv3 kernel stats of the default bench command (both precision modes), FETCH_SIZE / WRITE_SIZE
#      passes, the bench line with every leg
#   2. per-tile stamps of the ping-pong GEMM and of the 384 x 128 kernel, the L2 -> LDS transport probe, the batch-1 latency breakdown
#   3. the whole GPU suite with -s (the parity figures the documents quote) and the smoke entry
set -u
T=${1:-r05z}
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out
mkdir -p $O
bash tools/run_final_profiles.sh $T
timeout 300 python tools/gemm_stamps.py > $O/${T}_gemm8_phase_cycles.log 2>&1
timeout 300 python tools/n128_bench.py > $O/${T}_n128_tile_stamps.txt 2>&1
timeout 120 python tools/latency_breakdown.py 1 > $O/${T}_latency_batch1.txt 2>&1
timeout 1700 python -m pytest tests -m gpu -q -x -s -p no:cacheprovider > $O/${T}_pytest_gpu.log 2>&1
grep -E "passed|failed" $O/${T}_pytest_gpu.log | tail -2
grep -E "relmax" $O/${T}_pytest_gpu.log > $O/${T}_pytest_gpu_parity.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
