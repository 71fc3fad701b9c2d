#!/bin/bash
# first GPU visit of round 2: parity tests in both precision modes, the default bench line, the precision-variant table
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -s -p no:cacheprovider 2>&1 | tail -250 > gpurun_out/r02_gpu1_pytest.log
tail -40 gpurun_out/r02_gpu1_pytest.log
timeout 600 python tools/precision_modes.py 16 > gpurun_out/r02_gpu1_precision_modes.log 2>&1
cat gpurun_out/r02_gpu1_precision_modes.log
timeout 900 python bench.py > gpurun_out/r02_gpu1_bench.json 2> gpurun_out/r02_gpu1_bench.err
cat gpurun_out/r02_gpu1_bench.json | head -c 6000
tail -5 gpurun_out/r02_gpu1_bench.err
