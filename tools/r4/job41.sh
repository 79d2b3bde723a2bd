#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job41; mkdir -p $O
for v in ww1 ww4 ww3; do
  export VTX_LIBVTX=$R/tools/probe/ablate/libvtx_$v.so
  echo "== $v"; VTX_CHECK_SKIP=1 timeout 200 python tools/r4/wgrad_wide_check.py --wide 1 2>&1 | grep "us ("
done | tee $O/ablate.log
unset VTX_LIBVTX
cd /tmp
for set in "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" \
   "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
   "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  (cd $R && timeout 300 rocprofv3 --pmc $set -d $O -o $tag -- python tools/r4/wgrad_wide_check.py --case "vit-s/16 b256" > $O/$tag.log 2>&1)
  python $R/tools/rocpd_pmc.py $O/${tag}_results.db --match "wgrad_" > $O/$tag.txt 2>&1
  rm -f $O/${tag}_results.db
  cat $O/$tag.txt
done
