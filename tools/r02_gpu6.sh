#!/bin/bash
export TMPDIR=/tmp
PRISMA_BANDS_LIB=$PWD/prisma_amd/libprisma_bands_premx.so timeout 600 python tools/ab_fams.py 2>&1 | grep precision
timeout 600 python tools/ab_fams.py 2>&1 | grep precision
timeout 900 python -m pytest tests/test_gpu_raft.py tests/test_gpu_depth.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
