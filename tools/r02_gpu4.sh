#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/precision_modes.py 16 short 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_depth.py tests/test_gpu_ops.py tests/test_gpu_zoe.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep -E "relmax|passed|failed|FAILED|Error" | tail -40
