#!/bin/bash
# Round-4 visit: pytest subset, then per-tile stamps and bench legs under switches.
# (STAMP_SHAPES="conv3x3 dense-k2304 ..." picks the shapes of tools/gemm_stamps.py)
# usage: bash tools/run_r04_mix_visit.sh <tag> "<pytest args>" "<stamps cfgs separated by ;>" "<bench cfgs separated by ;>"
set -u
T=$1; PYT=$2; SC=$3; BC=$4
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out
mkdir -p $O
if [ -n "$PYT" ]; then
  timeout 1500 python -m pytest $PYT -x -q -m gpu > $O/${T}_pytest.log 2>&1
  tail -12 $O/${T}_pytest.log
fi
: > $O/${T}_gemm8_phase_cycles.log
IFS=';' read -ra SCS <<< "$SC"
for cfg in "${SCS[@]}"; do
  [ -z "$cfg" ] && continue
  env $cfg timeout 300 python tools/gemm_stamps.py ${STAMP_SHAPES:-fc1 fc1-gelu proj fc2} >> $O/${T}_gemm8_phase_cycles.log 2>&1
done
cat $O/${T}_gemm8_phase_cycles.log
IFS=';' read -ra BCS <<< "$BC"
i=0
for cfg in "${BCS[@]}"; do
  [ -z "$cfg" ] && continue
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
