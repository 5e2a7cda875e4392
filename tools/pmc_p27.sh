#!/bin/bash
# HBM-traffic and occupancy counters of the 27-point 256^3 SpMV (tools/quick_p27.py), one rocprofv3 pass per counter group
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_p27
mkdir -p $OUT
for pass in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  d=$OUT/$(echo $pass | tr ' ' '_' | cut -c1-40)
  (cd $R && timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $d -- python3 tools/quick_p27.py) > $d.log 2>&1 || { echo "pass $pass failed"; tail -5 $d.log; exit 1; }
done
cd $R && python3 tools/pmc_summary.py $OUT/* | grep -i "spmv"
