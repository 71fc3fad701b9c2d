#!/bin/bash
# Round-4 A/B visit (through gpurun): parity of the GEMM-bearing tests on the new build, then same-box bench legs under the
# environment switches named on the command line.  usage: bash tools/run_r04_ab_visit.sh <tag> "<pytest args>" "ENV=.. ENV=.." ["ENV.." ...]
set -u
T=$1; shift
PYT=$1; shift
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out
mkdir -p $O
if [ -n "$PYT" ]; then
  timeout 1500 python -m pytest $PYT -x -q -m gpu > $O/${T}_pytest.log 2>&1
  tail -15 $O/${T}_pytest.log
fi
i=0
for cfg in "$@"; do
  echo "=== [$i] $cfg" | tee -a $O/${T}_ab.log
  env $cfg timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --one-precision > $O/${T}_ab_$i.log 2> $O/${T}_ab_$i.err
  tail -1 $O/${T}_ab_$i.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
t=d['this_precision']
print('fps %.2f  step %.1f ms  depth %.2f  flow %.2f' % (d['value'], d['ms_per_step'], t['depth_ms_per_step'], t['flow_ms_per_step']))
k=d.get('kernel_ms_per_step',{})
print('  '+'  '.join('%s %.2f' % (n.replace('gemm8_kernel','g8').replace('gemm_kernel','g'),v) for n,v in sorted(k.items()) if v>2.0))
" 2>&1 | tee -a $O/${T}_ab.log
  i=$((i+1))
done
