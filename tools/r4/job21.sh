#!/bin/bash
# SQ counters of the A-stationary kernel on the stage-3 shapes (M = 25 088)
R=$PWD; export TMPDIR=/tmp
tools/pmc_kernel.sh r4astat gemm_astat -- python tools/r4/astat_check.py --iters 6 > $R/gpurun_out/pmc_r4astat.txt 2>&1
tail -n 120 $R/gpurun_out/pmc_r4astat.txt
