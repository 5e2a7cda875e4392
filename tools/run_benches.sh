set -e
for w in p7_256 pl_8m p27_256 foam_standin_128k; do
  timeout -k 10 400 python bench.py --workload $w > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
  echo "$w done"
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --no-cpu-baseline > gpurun_out/bench_under_rocprof.json 2> gpurun_out/bench_under_rocprof.err
ls gpurun_out/prof_bench/*/
