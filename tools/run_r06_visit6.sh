#!/bin/bash
# round 6, sixth visit: the tests touched by the plan-key fix, the power / clock arms with one op call per arm
set -x
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_raft.py tests/test_gpu_edges.py tests/test_gpu_gmflow.py -m gpu -q > gpurun_out/r06f_pytest_gpu.log 2>&1
tail -4 gpurun_out/r06f_pytest_gpu.log
timeout 600 python tools/power_clock.py > gpurun_out/r06f_power_clock.txt 2>&1
cat gpurun_out/r06f_power_clock.txt
