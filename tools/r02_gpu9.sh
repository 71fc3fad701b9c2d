#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_raft.py tests/test_gpu_edges.py tests/test_band_cli.py tests/test_band_multirank.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep -E "golden|pair |1080p|passed|failed|FAILED|Error|differing" | tail -40
timeout 600 python bench.py --steps 3 2>gpurun_out/r02_gpu9_bench.err | tee gpurun_out/r02_gpu9_bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['dtype'], d['this_precision'], d.get('other_precision')); print(d['kernel_ms_per_step']); print(d['roofline']); print(d['cpu_baseline'])"
