#!/bin/bash
# A/B of the DPT head tail at low resolution (engine.h wz_, elementwise.hip dpt_tail_kernel; PB_HEAD_TAIL=0 = the old formulation) on one box
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export PYTHONUNBUFFERED=1
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_depth.py tests/test_gpu_outliers.py tests/test_gpu_edges.py -m gpu -x -q -s 2>&1 | grep -E "relmax|passed|failed|Error|assert" > $O/r06u_head_tail_tests.txt
tail -25 $O/r06u_head_tail_tests.txt
D="--steps 10 --warmup 3 --one-precision --no-cpu-baseline --host-clips 0 --no-clock"
for rep in 1 2; do
  for v in 0 1; do
    PB_HEAD_TAIL=$v timeout 600 python bench.py $D > $O/r06u_head_tail_${v}_${rep}.log 2> $O/r06u_head_tail_${v}_${rep}.err
    tail -1 $O/r06u_head_tail_${v}_${rep}.log > $O/r06u_head_tail_${v}_${rep}.json
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06u_head_tail_?_?.json')):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, 'unreadable', e); continue
    s = d.get('sequential') or {}
    k = d.get('kernel_ms_per_step', {})
    print(f, d['value'], d['ms_per_step'], 'seq', s.get('ms_per_step'), 'depth seq', s.get('depth_ms_per_step'), 'lat', d.get('latency_720p_batch1_ms'),
          {n: v for n, v in k.items() if n in ('depth/elementwise', 'depth/gemm_kernel<256, 32, 4, 1, 1, 5, false, 2, true>', 'depth/gemm_kernel<128, 128, 2, 2, 0, 0, true, 2, true>')})
PY
