#!/bin/bash
# SQ counter passes over any micro-benchmark:  tools/pmc_kernel.sh OUTTAG MATCH -- python tools/bench_gemm.py ...
R=$PWD; TAG=$1; MATCH=$2; shift 3
export TMPDIR=/tmp
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES SQ_LDS_ADDR_CONFLICT" \
           "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  (cd $R && timeout 300 rocprofv3 --pmc $set -d $O -o $tag -- "$@" > $O/$tag.log 2>&1)
  python $R/tools/rocpd_pmc.py $O/${tag}_results.db --match "$MATCH" > $O/$tag.txt 2>&1
  rm -f $O/${tag}_results.db
  cat $O/$tag.txt
done
