#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/ab_fams.py 2>&1 | grep precision
timeout 2400 python -m pytest tests -m gpu -q -s -p no:cacheprovider 2>&1 | tail -300 > gpurun_out/r02_gpu7_pytest.log
grep -E "passed|failed|FAILED|Error" gpurun_out/r02_gpu7_pytest.log | tail -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/run_final_profiles.sh r02b
