#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_raft.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep -E "golden|pair |1080p|passed|failed|FAILED|Error|differing" | tail -30
PB_TAPIN=0 timeout 300 python tools/ab_flow.py 2>&1 | grep precision
timeout 300 python tools/ab_flow.py 2>&1 | grep precision
PB_TAPIN=2 timeout 300 python tools/ab_flow.py 2>&1 | grep precision
PB_MX=10 AB_PREC=1 timeout 300 python tools/ab_flow.py 2>&1 | grep precision
PB_MX=10 PB_TAPIN=0 AB_PREC=1 timeout 300 python tools/ab_flow.py 2>&1 | grep precision
