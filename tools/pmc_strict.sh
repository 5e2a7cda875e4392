#!/bin/bash
# SQ counters of the strict loop's kernels at 256^3 (tools/quick_strict.py), one rocprofv3 pass per counter group
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_strict
rm -rf $OUT && mkdir -p $OUT
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  d=$OUT/$(echo $pass | tr ' ' '_' | cut -c1-40)
  (cd $R && timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $d -- python3 tools/quick_strict.py 256 256 256 10) > $d.log 2>&1 || { echo "pass $pass failed"; tail -5 $d.log; }
done
cd $R && python3 tools/pmc_summary.py $OUT/* | grep -i "xs_fused\|cg_update\|spmv_ell_r2<6, 2"
