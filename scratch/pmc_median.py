"""Median per-launch value of one PMC counter for a kernel (rocprofv3 counter_collection.csv)."""
import csv, sys, glob, statistics, collections
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
vals = collections.defaultdict(list)
for f in files:
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, c), v in per.items(): vals[c].append(v)
for c, v in vals.items(): print(c, "median", statistics.median(v), "n", len(v))
