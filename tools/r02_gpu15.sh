#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_raft.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
timeout 300 python tools/ab_flow.py 2>&1 | grep precision
