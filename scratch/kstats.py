"""Print the top rows of a rocprofv3 kernel_stats.csv (names contain commas: parse properly)."""
import csv, sys, glob
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(files[0])))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.1f}%")
