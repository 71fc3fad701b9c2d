#!/bin/bash
# round 6, first visit: the tree as round 5 left it + tools/overlap_bench.py (what stream-level overlap can buy, with socket power beside)
set -x
mkdir -p gpurun_out
ls /sys/class/drm/card*/device/hwmon/hwmon*/ > gpurun_out/r06a_hwmon_ls.txt 2>&1
/opt/rocm/bin/amd-smi metric --power --clock --json > gpurun_out/r06a_smi.json 2>&1
python tools/overlap_bench.py --steps 6 > gpurun_out/r06a_overlap.txt 2>&1
tail -3 gpurun_out/r06a_overlap.txt
python bench.py --steps 10 --warmup 3 --one-precision --no-cpu-baseline --host-clips 0 > gpurun_out/r06a_bench.json 2> gpurun_out/r06a_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06a_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['this_precision'])
PY
