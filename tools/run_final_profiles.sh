#!/bin/bash
# Final profile set of a round (run on the GPU box through gpurun).
#   1. the default bench command as the driver runs it (both bands of a step at once)            -> <tag>_default_bench_line.json
#   2. rocprofv3 kernel stats of `bench.py --sequential-only` (bands one after the other: a launch's duration is the kernel's own; parent
#      process = the timed precision mode, child = the other one: one stats file each)             -> <tag>_sequential_bench_kernel_stats.csv, ..._line.json
#   3. HBM traffic counters of the sequential command (separate FETCH_SIZE / WRITE_SIZE passes)   -> <tag>_pmc_traffic.json
#   4. per-symbol effective shader clock (GRBM_GUI_ACTIVE pass, tools/pmc_clock.py)               -> <tag>_clock_per_symbol.json
#   5. the bench line with every optional leg                                                      -> <tag>_all_legs_bench_line.json
# usage: bash tools/run_final_profiles.sh <tag>      -> gpurun_out/<tag>_*
set -u
T=${1:-r06}
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out
mkdir -p $O
timeout 900 python bench.py > $O/${T}_bench.log 2> $O/${T}_bench.err
tail -1 $O/${T}_bench.log > $O/${T}_default_bench_line.json
timeout 900 rocprofv3 --kernel-trace --stats -d $O/${T}_seq --output-format csv -- python bench.py --sequential-only --host-clips 0 --no-latency > $O/${T}_seq.log 2> $O/${T}_seq.err
tail -1 $O/${T}_seq.log > $O/${T}_sequential_bench_line.json
i=0
for f in $(ls $O/${T}_seq/*/*kernel_stats.csv | sort -V); do
  if [ $i -eq 0 ]; then cp $f $O/${T}_sequential_bench_kernel_stats.csv; else cp $f $O/${T}_sequential_bench_other_precision_kernel_stats.csv; fi
  i=$((i+1))
done
D="--sequential-only --steps 1 --warmup 1 --no-cpu-baseline --one-precision --host-clips 0 --no-latency --no-clock"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${T}_pmc_fetch --output-format csv -- python bench.py $D > $O/${T}_pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${T}_pmc_write --output-format csv -- python bench.py $D > $O/${T}_pmc_write.log 2>&1
python tools/pmc_summary.py $O/${T}_pmc_fetch $O/${T}_pmc_write $O/${T}_pmc_traffic.json > $O/${T}_pmc_traffic.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/${T}_pmc_clock --output-format csv -- python bench.py $D > $O/${T}_pmc_clock.log 2>&1
python tools/pmc_clock.py $O/${T}_pmc_clock $O/${T}_clock_per_symbol.json > $O/${T}_clock_per_symbol.txt 2>&1
rm -rf $O/${T}_pmc_fetch $O/${T}_pmc_write $O/${T}_pmc_clock $O/${T}_seq
timeout 900 python bench.py --all-legs --one-precision > $O/${T}_all_legs.log 2> $O/${T}_all_legs.err
tail -1 $O/${T}_all_legs.log > $O/${T}_all_legs_bench_line.json
cut -c1-400 $O/${T}_default_bench_line.json
