#!/bin/bash
# usage (through gpurun): bash tools/run_final_visit.sh <tag>
# final visit of a round: the profile set, then the whole GPU suite (-s: the parity figures DESIGN.md quotes) and the smoke entry
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r02e}
bash tools/run_final_profiles.sh $T
timeout 1500 python -m pytest tests -m gpu -q -x -s -p no:cacheprovider > gpurun_out/${T}_pytest_gpu.log 2>&1
grep -E "passed|failed" gpurun_out/${T}_pytest_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
