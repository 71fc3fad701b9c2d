#!/bin/bash
# SQ / LDS counters of the halo-tiled convolution and of flow_gmflow's attention kernel (are they waiting on the LDS, and do their swizzles
# hold on the hardware's ds_read_b128 lane groups?)   usage (through gpurun): bash tools/run_halo_pmc.sh <tag>
set -u
T=${1:-r03}
export PYTHONUNBUFFERED=1 TMPDIR=/tmp AB_PREC=1 AB_FRAMES=${AB_FRAMES:-5} AB_PAIRS=${AB_PAIRS:-3}
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out
mkdir -p $O
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU"
for s in flow gmflow; do
  timeout 300 rocprofv3 --kernel-trace --pmc $A -d $O/${T}_${s}_pmc_a --output-format csv -- python tools/ab_$s.py > $O/${T}_${s}_pmc_a.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $B -d $O/${T}_${s}_pmc_b --output-format csv -- python tools/ab_$s.py > $O/${T}_${s}_pmc_b.log 2>&1
done
for k in "conv3x3_c64_mx2_kernel" "corr_volume_kernel"; do
  echo "== $k" >> $O/${T}_halo_attn_sq_counters.txt
  python tools/pmc_sq.py $O/${T}_flow_pmc_a "$k" >> $O/${T}_halo_attn_sq_counters.txt
  python tools/pmc_sq.py $O/${T}_flow_pmc_b "$k" >> $O/${T}_halo_attn_sq_counters.txt
done
for k in "attn128_kernelILb1ELb0ELi4ELi1" "attn128_kernelILb1ELb0ELi4ELi0" "attn128_kernelILb1ELb1ELi1ELi0"; do
  echo "== $k" >> $O/${T}_halo_attn_sq_counters.txt
  python tools/pmc_sq.py $O/${T}_gmflow_pmc_a "$k" >> $O/${T}_halo_attn_sq_counters.txt
  python tools/pmc_sq.py $O/${T}_gmflow_pmc_b "$k" >> $O/${T}_halo_attn_sq_counters.txt
done
rm -rf $O/${T}_flow_pmc_a $O/${T}_flow_pmc_b $O/${T}_gmflow_pmc_a $O/${T}_gmflow_pmc_b
cat $O/${T}_halo_attn_sq_counters.txt
