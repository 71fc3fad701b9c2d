#!/bin/bash
# Final profile set of a round (run on the GPU box through gpurun): rocprofv3 kernel stats of the default bench command
# (the timed precision mode in the parent process, the other mode in a child process: one stats file each), HBM traffic
# counters of the same command (separate FETCH_SIZE / WRITE_SIZE passes), and the bench line with every optional leg.
# usage: bash tools/run_final_profiles.sh <tag>      -> gpurun_out/<tag>_*
set -u
T=${1:-r02}
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out
mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats -d $O/${T}_bench --output-format csv -- python bench.py > $O/${T}_bench.log 2> $O/${T}_bench.err
tail -1 $O/${T}_bench.log > $O/${T}_default_bench_line.json
i=0
for f in $(ls $O/${T}_bench/*/*kernel_stats.csv | sort -V); do
  if [ $i -eq 0 ]; then cp $f $O/${T}_default_bench_kernel_stats.csv; else cp $f $O/${T}_default_bench_other_precision_kernel_stats.csv; fi
  i=$((i+1))
done
D="--steps 1 --warmup 1 --no-cpu-baseline --one-precision --host-clips 0 --no-latency --no-clock"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${T}_pmc_fetch --output-format csv -- python bench.py $D > $O/${T}_pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${T}_pmc_write --output-format csv -- python bench.py $D > $O/${T}_pmc_write.log 2>&1
python tools/pmc_summary.py $O/${T}_pmc_fetch $O/${T}_pmc_write $O/${T}_pmc_traffic.json > $O/${T}_pmc_traffic.txt 2>&1
rm -rf $O/${T}_pmc_fetch $O/${T}_pmc_write $O/${T}_bench
timeout 900 python bench.py --all-legs --one-precision > $O/${T}_all_legs.log 2> $O/${T}_all_legs.err
tail -1 $O/${T}_all_legs.log > $O/${T}_all_legs_bench_line.json
cut -c1-400 $O/${T}_default_bench_line.json
