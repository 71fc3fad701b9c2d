#!/bin/bash
# Final profile set of a round (run on the GPU box through gpurun): rocprofv3 kernel stats of the default bench command,
# HBM traffic counters of the same command (separate FETCH_SIZE / WRITE_SIZE passes), SQ counters of the attention kernel,
# and the bench line with every optional leg (--all-legs: depth alone, flow 720p, mask, three-band pipeline, PCIe, latency).
# usage: bash tools/run_final_profiles.sh <tag>      -> gpurun_out/<tag>_*
set -u
T=${1:-r01h}
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${T}_bench --output-format csv -- python bench.py > $O/${T}_bench.log 2> $O/${T}_bench.err
tail -1 $O/${T}_bench.log > $O/${T}_default_bench_line.json
cp $O/${T}_bench/*/*kernel_stats.csv $O/${T}_default_bench_kernel_stats.csv
D="--steps 1 --warmup 1 --no-cpu-baseline"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${T}_pmc_fetch --output-format csv -- python bench.py $D > $O/${T}_pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${T}_pmc_write --output-format csv -- python bench.py $D > $O/${T}_pmc_write.log 2>&1
python tools/pmc_summary.py $O/${T}_pmc_fetch $O/${T}_pmc_write $O/${T}_pmc_traffic.json > $O/${T}_pmc_traffic.txt 2>&1
ATT_ITERS=2 timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d $O/${T}_attn_sq_a --output-format csv -- python tools/attn_bench.py 0 > $O/${T}_attn_sq_a.log 2>&1
ATT_ITERS=2 timeout 150 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $O/${T}_attn_sq_b --output-format csv -- python tools/attn_bench.py 0 > $O/${T}_attn_sq_b.log 2>&1
{ echo "# attnq_kernel<1,2,0,false,8>, B = 32, 16 heads, 2443 tokens: rocprofv3 --pmc (two passes), mean per launch"; tail -1 $O/${T}_attn_sq_a.log; python tools/pmc_sq.py $O/${T}_attn_sq_a; python tools/pmc_sq.py $O/${T}_attn_sq_b; echo "# timing-only ablations (tools/attn_bench.py 2 11 12 13 14 15 16)"; ATT_ITERS=30 python tools/attn_bench.py 2 2 11 12 13 14 15 16; } > $O/${T}_attention_sq_counters.txt 2>&1
timeout 600 python bench.py --all-legs > $O/${T}_all_legs.log 2> $O/${T}_all_legs.err
tail -1 $O/${T}_all_legs.log > $O/${T}_all_legs_bench_line.json
cut -c1-300 $O/${T}_default_bench_line.json
