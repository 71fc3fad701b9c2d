#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_raft.py -m gpu -q -s -x -p no:cacheprovider -k "pair_against or odd_feature or fast_mode" 2>&1 | grep -E "golden|flow_lo|fmap|passed|failed|FAILED|Error|assert" | tail -30
