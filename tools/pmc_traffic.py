"""profiles/<round>_traffic_<name>.json from the two rocprofv3 --pmc passes of tools/pmc_bench.sh (FETCH_SIZE, WRITE_SIZE).
    python tools/pmc_traffic.py <workload> <name> <dir with FETCH_SIZE/ and WRITE_SIZE/> <round>
The SpMV of the timed loop is the spmv_ell_r2<K, 1, ...> / spmv_ell_staged<K, 1, ...> (per-workgroup p.Ap partials, no halo)
instantiation with the most launches; pl_8m: pb_expand + pb_accumulate.  Corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: units are KiB;
FETCH_SIZE on gfx950 counts 64 bytes per 128-byte request -- the factor is CALIBRATED here on the 1 GiB read16 / copy16 launches
of the same run (bench.py's ceilings), not assumed; WRITE_SIZE is calibrated on copy16's 1 GiB of stores."""
import collections
import csv
import glob
import json
import re
import sys

workload, name, root, rnd = sys.argv[1:5]


def averages(counter):
    agg = collections.defaultdict(list)
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (root, counter), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
                agg[k].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


F, W = averages("FETCH_SIZE"), averages("WRITE_SIZE")
GiB = float(1 << 30)
read_k = next(k for k in F if k.startswith("read16"))
copy_k = next(k for k in F if k.startswith("copy16"))
fcorr = GiB / (F[read_k][0] * 1024.0)
fcorr_copy = GiB / (F[copy_k][0] * 1024.0)
wcorr = GiB / (W[copy_k][0] * 1024.0)
if workload.startswith("pl_"):
    # one SpMV = pb_expand + pb_accumulate; of the latter the instantiation of the timed loop (with p.Ap: the most launches)
    ks = [k for k in F if k.startswith("pb_expand")] + [max((k for k in F if k.startswith("pb_accumulate")), key=lambda k: F[k][1])]
else:
    cand = [k for k in F if re.match(r"spmv_ell_r2<\d+, 1, \d+, \d+, 0>", k) or re.match(r"spmv_ell_staged<\d+, 1, \d+, \d+, \d+>", k)]
    ks = [max(cand, key=lambda k: F[k][1])]
by = {}
total = 0.0
for k in ks:
    rd = F[k][0] * 1024.0 * fcorr
    wr = W[k][0] * 1024.0 * wcorr
    by[k] = {"launches": F[k][1], "FETCH_SIZE_KiB_avg": F[k][0], "WRITE_SIZE_KiB_avg": W[k][0], "read_bytes": int(rd), "written_bytes": int(wr)}
    total += rd + wr
others = {k: {"launches": F[k][1], "read_bytes": int(F[k][0] * 1024.0 * fcorr), "written_bytes": int(W.get(k, (0.0, 0))[0] * 1024.0 * wcorr)}
          for k in sorted(F) if k not in ks and F[k][1] >= 5 and F[k][0] > 1000.0}
print(json.dumps({
    "workload": workload, "name": name, "round": rnd,
    "source": "tools/pmc_bench.sh %s: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes of `bench.py --workload %s --steps 20 --warmup 2 "
              "--no-cpu-baseline` on the %s tree; per-kernel averages in profiles/%s_pmc_%s.txt" % (workload, workload, rnd, rnd, name),
    "kernel": " + ".join(ks),
    "fetch_correction": fcorr,
    "fetch_correction_why": "gfx950: FETCH_SIZE (KiB) counts 64 bytes per 128-byte request; calibrated in the same run on %s, which reads 1 GiB and "
                            "reported %.1f KiB (on %s: %.4f)" % (read_k, F[read_k][0], copy_k, fcorr_copy),
    "write_correction": wcorr,
    "write_correction_why": "%s writes 1 GiB with 16-byte stores and reported %.1f KiB" % (copy_k, W[copy_k][0]),
    "hbm_bytes_per_launch": int(total),
    "by_kernel": by,
    "other_kernels_of_the_run": others,
}, indent=1))
