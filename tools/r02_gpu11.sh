#!/bin/bash
# final visit of the round: the profile set, then the whole GPU suite and the smoke entry
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/run_final_profiles.sh r02c
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r02c_pytest_gpu.log 2>&1
tail -3 gpurun_out/r02c_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
