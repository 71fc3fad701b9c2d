#!/bin/bash
# round 6, third visit: new bench timed region (both bands at once, attribution from the sequential pass), cnet on F - 1 frames,
# PB_TILE_N128 under overlap once more, the flow parity tests
set -x
mkdir -p gpurun_out
cat /sys/bus/pci/devices/0000:*/hwmon/hwmon*/power1_cap 2>/dev/null | head -8 > gpurun_out/r06c_power_cap.txt
python bench.py --steps 10 --warmup 3 --one-precision --no-cpu-baseline > gpurun_out/r06c_bench.json 2> gpurun_out/r06c_bench.err
PB_TILE_N128=1 python bench.py --steps 10 --warmup 3 --one-precision --no-cpu-baseline --host-clips 0 --no-latency --no-clock > gpurun_out/r06c_bench_n128.json 2> gpurun_out/r06c_bench_n128.err
python bench.py --steps 10 --warmup 3 --one-precision --no-cpu-baseline --sequential-only --host-clips 0 --no-latency --no-clock > gpurun_out/r06c_bench_seq.json 2> gpurun_out/r06c_bench_seq.err
python - <<'PY'
import json
for f in ('r06c_bench','r06c_bench_n128','r06c_bench_seq'):
    try:
        d=json.loads(open(f'gpurun_out/{f}.json').read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['sequential'], d['avg_power_w'], d['joules_per_frame'], d['pcie_inclusive_fps'], d['latency_720p_batch1_ms'])
    except Exception as e: print(f, 'ERR', e)
PY
python -m pytest tests/test_gpu_raft.py -m gpu -x -q 2>&1 | tail -5
