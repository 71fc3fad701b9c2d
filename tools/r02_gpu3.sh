#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_raft.py tests/test_gpu_ops.py tests/test_gpu_depth.py tests/test_band_cli.py -m gpu -q -s -p no:cacheprovider 2>&1 | tail -200 > gpurun_out/r02_gpu3_pytest.log
grep -E "relmax|passed|failed|FAILED|Error|differing" gpurun_out/r02_gpu3_pytest.log | tail -60
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python bench.py --precision 1 --one-precision --no-cpu-baseline --steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['dtype'], d['this_precision']); print(d['kernel_ms_per_step'])"
