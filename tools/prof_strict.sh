#!/bin/bash
# per-kernel average durations of the strict (bit-identical) loop at one size: tools/prof_strict.sh "256 256 256" [iters]
# (run on the GPU box from the repo root; rocprofv3 --kernel-trace --stats)
sz="$1"; it="${2:-40}"; root="${GRAFT_REPO_ROOT:-$PWD}"; out="$root/gpurun_out/prof_strict_${sz// /_}"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$root/tools/quick_strict.py" $sz $it > "$out.log" 2>&1 || exit 1
f=$(ls "$out"/*/*kernel_stats.csv | tail -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    name = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    print("%-44s calls %5s  avg %9.1f us" % (name[:44], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
grep "strict" "$out.log"
