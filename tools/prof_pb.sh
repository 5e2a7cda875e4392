#!/bin/bash
# per-kernel average durations of one config-5 layout: tools/prof_pb.sh [layout] [tag]   (layout 5 = two-phase, 3 = column-blocked CSR)
# (run on the GPU box from the repo root; rocprofv3 --kernel-trace --stats; CGH_PB_* knobs pass through the environment)
lay="${1:-5}"; tag="${2:-pb}"; root="${GRAFT_REPO_ROOT:-$PWD}"; out="$root/gpurun_out/prof_$tag"
cd /tmp && export TMPDIR=/tmp
cd "$root" && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 tools/prof_csrb.py "$lay" > "$out.log" 2>&1 || exit 1
f=$(ls "$out"/*/*kernel_stats.csv | tail -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    name = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    print("%-60s calls %5s  avg %9.1f us" % (name[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
grep "layout" "$out.log"
