#!/bin/bash
# Round 5: SQ / LDS counters of the three GEMM tile geometries on one convolution shape (the RAFT encoders' 3 x 3, 128 -> 128, M = 569 k) and
# of the 256 x 256 ping-pong kernel on the ViT's fc1 - the counter side of DESIGN.md section 5's "what bounds the GEMM kernels: the LDS".
# usage (through gpurun): bash tools/run_r05_lds_pmc.sh <tag>
set -u
T=${1:-r05}
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out
mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_]*LDS[A-Z_]*\|SQ_INSTS_LDS\|SQ_ACTIVE_INST_LDS" | sort -u | tr '\n' ' ' > $O/${T}_lds_counters_available.txt
cat > /tmp/lds_run.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from prisma_amd import engine
ops = engine.Ops(0)
P = 31 * 18360
for t in (1, 11, 2):
    ops.gemm_bench(P, 128, 1152, tile=t, epi=10, iters=4)
ops.gemm_bench(78336, 4096, 1024, tile=2, epi=0, iters=4)
PY
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"
B="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM"
timeout 300 rocprofv3 --kernel-trace --pmc $A -d $O/${T}_lds_a --output-format csv -- python /tmp/lds_run.py > $O/${T}_lds_a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $B -d $O/${T}_lds_b --output-format csv -- python /tmp/lds_run.py > $O/${T}_lds_b.log 2>&1
: > $O/${T}_gemm_lds_sq_counters.txt
for k in "gemm_kernel<128, 128, 2, 2, 1, 0" "gemm8n_kernel<1, 0" "gemm8_kernel<1, 0, 0" "gemm8_kernel<0, 0, 0"; do
  echo "== $k" >> $O/${T}_gemm_lds_sq_counters.txt
  python tools/pmc_sq.py $O/${T}_lds_a "$k" >> $O/${T}_gemm_lds_sq_counters.txt
  python tools/pmc_sq.py $O/${T}_lds_b "$k" >> $O/${T}_gemm_lds_sq_counters.txt
done
tail -3 $O/${T}_lds_b.log
rm -rf $O/${T}_lds_a $O/${T}_lds_b
cat $O/${T}_lds_counters_available.txt; echo; cat $O/${T}_gemm_lds_sq_counters.txt
