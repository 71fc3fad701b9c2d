#!/bin/bash
# round 6, fourth visit: the whole GPU suite on the tree with the host pipelines of the mask band / flow masks, the error-exit drain, split-K for
# max_batch = 1 contexts; then the bench and the one-frame latency breakdown
set -x
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q -x -s > gpurun_out/r06d_pytest_gpu.log 2>&1
tail -5 gpurun_out/r06d_pytest_gpu.log
grep -c "relmax" gpurun_out/r06d_pytest_gpu.log
python bench.py --steps 10 --warmup 3 --one-precision --no-cpu-baseline > gpurun_out/r06d_bench.json 2> gpurun_out/r06d_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06d_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['sequential']['value'], d['avg_power_w'], d['joules_per_frame'], d['pcie_inclusive_fps'], d['latency_720p_batch1_ms'], d['latency_720p_batch1_note'][-12:])
PY
python tools/latency_breakdown.py 1 > gpurun_out/r06d_latency_batch1.txt 2>&1
head -16 gpurun_out/r06d_latency_batch1.txt
