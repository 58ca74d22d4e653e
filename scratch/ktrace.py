"""Per-launch durations of one kernel from a rocprofv3 kernel_trace.csv, in launch order."""
import csv, sys, glob
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = [r for r in csv.DictReader(open(files[0])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print(sys.argv[2], len(d), "launches; last 12 (us):", [round(x, 1) for x in d[-12:]])
