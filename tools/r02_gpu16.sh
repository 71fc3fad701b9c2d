#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_raft.py -m gpu -q -x -s -p no:cacheprovider 2>&1 | grep -E "golden|1080p|passed|failed|Error|rror" | tail -14
PB_VOLUME=0 AB_PREC=1 timeout 300 python tools/ab_flow.py 2>&1 | grep precision
AB_PREC=1 timeout 300 python tools/ab_flow.py 2>&1 | grep precision
