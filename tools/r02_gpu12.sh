#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
PB_TAPIN=0 timeout 300 python tools/ab_depth.py 2>&1 | grep precision
timeout 300 python tools/ab_depth.py 2>&1 | grep precision
timeout 900 python -m pytest tests/test_gpu_depth.py tests/test_gpu_zoe.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep -E "golden|frame |metric|passed|failed|FAILED|Error" | tail -30
