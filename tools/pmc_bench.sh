#!/bin/bash
# HBM bytes per SpMV launch of a bench.py workload from the PMC counters: one rocprofv3 pass per counter (FETCH_SIZE and
# WRITE_SIZE do not fit one pass), then tools/pmc_traffic.py turns the two CSVs into profiles/<round>_traffic_<name>.json
# (gfx950 correction calibrated in the same run on the 1 GiB read / copy kernels bench.py launches for its ceilings).
#   tools/pmc_bench.sh <workload> [name] [round]     (export CGH_* knobs beforehand, e.g. CGH_ELL_DICT=0)
set -e
W=$1
NAME=${2:-$W}
RND=${3:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$NAME
rm -rf $OUT && mkdir -p $OUT
for pass in FETCH_SIZE WRITE_SIZE; do
  (cd $R && timeout -s KILL 240 rocprofv3 --pmc $pass --output-format csv -d $OUT/$pass -- python3 bench.py --workload $W --steps 20 --warmup 2 --no-cpu-baseline) > $OUT/$pass.log 2>&1 || { echo "pass $pass failed"; tail -5 $OUT/$pass.log; exit 1; }
done
cd $R && python3 tools/pmc_summary.py $OUT/FETCH_SIZE $OUT/WRITE_SIZE > gpurun_out/${RND}_pmc_$NAME.txt
python3 tools/pmc_traffic.py $W $NAME $OUT $RND > gpurun_out/${RND}_traffic_$NAME.json
cat gpurun_out/${RND}_traffic_$NAME.json
