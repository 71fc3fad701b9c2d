#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
PB_TAPIN=0 timeout 300 python tools/ab_flow.py 2>&1 | grep precision
timeout 300 python tools/ab_flow.py 2>&1 | grep precision
PB_MX_UPD=1 AB_PREC=1 timeout 300 python tools/ab_flow.py 2>&1 | grep precision
PB_MX_UPD=1 timeout 600 python -m pytest tests/test_gpu_raft.py -m gpu -q -s -p no:cacheprovider -k "720p or 1080p" 2>&1 | grep -E "p1|passed|failed" | tail -12
timeout 600 python bench.py --steps 3 --one-precision --no-cpu-baseline 2>gpurun_out/r02_gpu10_bench.err | tee gpurun_out/r02_gpu10_bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['dtype'], d['this_precision']); print(d['kernel_ms_per_step']); print(d['roofline'])"
PB_MX_UPD=1 timeout 600 python bench.py --steps 3 --one-precision --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MX_UPD', d['value'], d['dtype'], d['this_precision']); print(d['kernel_ms_per_step'])"
