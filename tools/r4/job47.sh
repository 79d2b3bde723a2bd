#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job47; mkdir -p $O
for v in gw1 gw2; do
  export VTX_LIBVTX=$R/tools/probe/ablate/libvtx_$v.so
  echo "== $v"; VTX_CHECK_SKIP=1 timeout 200 python tools/r4/gemm_wide_check.py --modes 1 --case "vit f" 2>&1 | grep "us"
done | tee $O/ablate.log
unset VTX_LIBVTX
cd /tmp
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  (cd $R && timeout 300 rocprofv3 --pmc $set -d $O -o $tag -- python tools/r4/gemm_wide_check.py --modes 1 --case "vit fc1" > $O/$tag.log 2>&1)
  python $R/tools/rocpd_pmc.py $O/${tag}_results.db --match "gemm_wide" > $O/$tag.txt 2>&1
  rm -f $O/${tag}_results.db
  cat $O/$tag.txt
done
