#!/bin/bash
# PMC pass over the window-attention micro-benchmark (stage given as $1), SQ counters only.
R=$PWD; S=${1:-1}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_attn
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES SQ_LDS_ADDR_CONFLICT"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $set -d $R/gpurun_out/pmc_attn -o $tag -- python $R/tools/bench_attn.py --stages $S --iters 3 > $R/gpurun_out/pmc_attn/$tag.log 2>&1
  python $R/tools/rocpd_pmc.py $R/gpurun_out/pmc_attn/${tag}_results.db --match wattn > $R/gpurun_out/pmc_attn/$tag.txt 2>&1
  rm -f $R/gpurun_out/pmc_attn/${tag}_results.db
  cat $R/gpurun_out/pmc_attn/$tag.txt
done
