#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_raft.py -m gpu -q -x -p no:cacheprovider -k "pair_against or odd_feature or 1080p" 2>&1 | tail -2
AB_PREC=1 timeout 300 python tools/ab_flow.py 2>&1 | grep -o "total [0-9.]* ms\|\"corr_volume_kernel\": [0-9.]*"
