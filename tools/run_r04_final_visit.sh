#!/bin/bash
# Final visit of round 4 (through gpurun): bash tools/run_r04_final_visit.sh <tag>
#   1. tools/run_final_profiles.sh <tag>: rocprofv3 kernel stats of the default bench command (both precision modes), FETCH_SIZE / WRITE_SIZE
#      passes, the bench line with every leg
#   2. the flow_gmflow leg under rocprofv3 --stats and with FETCH / WRITE counters, per-tile stamps of the ping-pong GEMM
#   3. the whole GPU suite with -s (the parity figures the documents quote) and the smoke entry
set -u
T=${1:-r04z}
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out
mkdir -p $O
bash tools/run_final_profiles.sh $T
timeout 300 rocprofv3 --kernel-trace --stats -d $O/${T}_gm --output-format csv -- python tools/ab_gmflow.py > $O/${T}_gmflow_leg.log 2>&1
for f in $(ls $O/${T}_gm/*/*kernel_stats.csv 2>/dev/null | head -1); do cp $f $O/${T}_gmflow_leg_kernel_stats.csv; done
AB_PAIRS=15 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${T}_gm_f --output-format csv -- python tools/ab_gmflow.py > /dev/null 2>&1
AB_PAIRS=15 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${T}_gm_w --output-format csv -- python tools/ab_gmflow.py > /dev/null 2>&1
python tools/pmc_summary.py $O/${T}_gm_f $O/${T}_gm_w $O/${T}_gmflow_leg_pmc_traffic.json > $O/${T}_gmflow_leg_pmc_traffic.txt 2>&1
rm -rf $O/${T}_gm $O/${T}_gm_f $O/${T}_gm_w
tail -1 $O/${T}_gmflow_leg.log | cut -c1-600
timeout 300 python tools/gemm_stamps.py > $O/${T}_gemm8_phase_cycles.log 2>&1
timeout 1700 python -m pytest tests -m gpu -q -x -s -p no:cacheprovider > $O/${T}_pytest_gpu.log 2>&1
grep -E "passed|failed" $O/${T}_pytest_gpu.log | tail -2
grep -E "relmax" $O/${T}_pytest_gpu.log > $O/${T}_pytest_gpu_parity.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
