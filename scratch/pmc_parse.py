"""Sum rocprofv3 --pmc counters per kernel name substring."""
import csv, sys, glob, collections
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(float); calls = 0
for f in files:
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVES": calls += 1
for k, v in sorted(agg.items()): print(f"{k:28s} {v:.4g}")
