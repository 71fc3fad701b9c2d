#!/bin/bash
# SQ counters of the RAFT update block's convolution kernels (the roofline kernel gemm8_kernel<1, 0, 0, true, false> and its 128 x 128
# sibling) in the split mode, with fp16 residual passes (default) and with the e4m3 / MX residual (PB_MX_UPD=1): what bounds them if
# halving the residual pass's matrix time buys 1 %?   usage (through gpurun): bash tools/run_upd_pmc.sh <tag>
set -u
T=${1:-r03}
export PYTHONUNBUFFERED=1 TMPDIR=/tmp AB_PREC=1 AB_FRAMES=${AB_FRAMES:-9}
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out
mkdir -p $O
for v in 0 1; do
  export PB_MX_UPD=$v
  timeout 200 python tools/ab_flow.py 2>&1 | tail -1 > $O/${T}_upd_mx${v}_times.txt
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d $O/${T}_upd_pmc_a$v --output-format csv -- python tools/ab_flow.py > $O/${T}_upd_pmc_a$v.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU -d $O/${T}_upd_pmc_b$v --output-format csv -- python tools/ab_flow.py > $O/${T}_upd_pmc_b$v.log 2>&1
  for k in "gemm8_kernel<1, 0, 0, true" "gemm_kernel<128, 128, 2, 2, 1, 0, true, 2"; do
    echo "== PB_MX_UPD=$v  $k" >> $O/${T}_upd_sq_counters.txt
    python tools/pmc_sq.py $O/${T}_upd_pmc_a$v "$k" >> $O/${T}_upd_sq_counters.txt
    python tools/pmc_sq.py $O/${T}_upd_pmc_b$v "$k" >> $O/${T}_upd_sq_counters.txt
  done
  rm -rf $O/${T}_upd_pmc_a$v $O/${T}_upd_pmc_b$v
done
cat $O/${T}_upd_mx0_times.txt $O/${T}_upd_mx1_times.txt
