#!/bin/bash
# counters of the config-5 SpMV kernels, one rocprofv3 pass per counter group (gpurun_out/pmc_csrb/<pass>)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_csrb
mkdir -p $OUT
L=${1:-3}
shift || true
for pass in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_READ_sum"; do
  d=$OUT/L${L}_$(echo $pass | tr ' ' '_' | cut -c1-40)
  (cd $R && timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $d -- python3 tools/prof_csrb.py $L "$@") > $d.log 2>&1 || { echo "pass $pass failed"; tail -5 $d.log; exit 1; }
done
(cd $R && timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/L${L}_trace -- python3 tools/prof_csrb.py $L "$@") > $OUT/L${L}_trace.log 2>&1
cd $R && python3 tools/pmc_summary.py $OUT/L${L}_* | grep -i "spmv\|copy\|pb_\|read8"
