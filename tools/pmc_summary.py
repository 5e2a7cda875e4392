"""Summarise rocprofv3 --pmc counter_collection CSVs: average counter value per kernel name."""
import collections, csv, glob, re, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            print("%-28s %-42s %-14s n=%-4d avg=%.1f" % (d.split("/")[-1], k, c, len(v), sum(v) / len(v)))
