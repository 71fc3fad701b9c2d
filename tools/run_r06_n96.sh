#!/bin/bash
# A/B of the 128 x 96 tile (gemm.h TILE_128x96) on one box: its tests, then the bench with PB_TILE_N96=0 / 1 / 2 alternating
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export PYTHONUNBUFFERED=1
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_raft.py tests/test_gpu_gmflow.py -m gpu -x -q 2>&1 | tail -8 > $O/r06r_n96_tests.txt
cat $O/r06r_n96_tests.txt
D="--steps 10 --warmup 3 --one-precision --no-cpu-baseline --host-clips 0 --no-latency --no-clock"
for rep in 1 2; do
  for v in 0 1 2; do
    PB_TILE_N96=$v timeout 600 python bench.py $D > $O/r06r_n96_${v}_${rep}.log 2> $O/r06r_n96_${v}_${rep}.err
    tail -1 $O/r06r_n96_${v}_${rep}.log > $O/r06r_n96_${v}_${rep}.json
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06r_n96_?_?.json')):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, 'unreadable', e); continue
    s = d.get('sequential') or {}
    k = d.get('kernel_ms_per_step', {})
    print(f, d['value'], d['ms_per_step'], 'seq', s.get('ms_per_step'), 'flow seq', s.get('flow_ms_per_step'), {n: v for n, v in k.items() if n.startswith('flow/gemm_kernel<128')})
PY
