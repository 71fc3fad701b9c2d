#!/bin/bash
# round 6, fifth visit: the whole GPU suite (no -x), the power / clock arms of the dominant kernel shapes
set -x
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 2000 python -m pytest tests -m gpu -q -s > gpurun_out/r06e_pytest_gpu.log 2>&1
tail -8 gpurun_out/r06e_pytest_gpu.log
timeout 600 python tools/power_clock.py > gpurun_out/r06e_power_clock.txt 2>&1
cat gpurun_out/r06e_power_clock.txt
