#!/bin/bash
# HBM traffic per kernel from PMC counters (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one pass).
#   tools/pmc_traffic.sh [swin_s|vit_s16]      -> gpurun_out/pmc_traffic_<model>.json / .md
R=$PWD; M=${1:-swin_s}
export TMPDIR=/tmp
O=$R/gpurun_out/pmc_traffic_$M
mkdir -p $O
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c -d $O -o $c -- python $R/bench.py --model $M --steps 3 --warmup 1 --no-cpu-baseline --no-secondary \
      --no-kernel-events > $O/$c.log 2>&1
done
python $R/tools/rocpd_traffic.py $O/FETCH_SIZE_results.db $O/WRITE_SIZE_results.db --model $M \
   --json $R/gpurun_out/pmc_traffic_$M.json > $R/gpurun_out/pmc_traffic_$M.md
rm -f $O/*_results.db
head -30 $R/gpurun_out/pmc_traffic_$M.md
