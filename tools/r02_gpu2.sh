#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -s -p no:cacheprovider 2>&1 | tail -300 > gpurun_out/r02_gpu2_pytest.log
grep -E "passed|failed|FAILED|Error" gpurun_out/r02_gpu2_pytest.log | tail -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_gpu2_smoke.log 2>&1; tail -3 gpurun_out/r02_gpu2_smoke.log
bash tools/run_final_profiles.sh r02a
