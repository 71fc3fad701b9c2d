export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
timeout 120 python tools/attn_bench.py 2 3 4 2>&1 | tail -4
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > gpurun_out/sq_counters.txt
for v in 4 2; do
  ATT_ITERS=2 timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d gpurun_out/attn_pmc_a$v --output-format csv -- python tools/attn_bench.py $v > gpurun_out/attn_pmc_a$v.log 2>&1
  python tools/pmc_sq.py gpurun_out/attn_pmc_a$v
  ATT_ITERS=2 timeout 150 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM -d gpurun_out/attn_pmc_b$v --output-format csv -- python tools/attn_bench.py $v > gpurun_out/attn_pmc_b$v.log 2>&1
  python tools/pmc_sq.py gpurun_out/attn_pmc_b$v
done
