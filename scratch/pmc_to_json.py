"""rocprofv3 --pmc counter_collection.csv files -> profiles/rNN_pmc_<kernel>.json: per-pass medians
of every collected counter for the dominant intersect kernel of the bench step.

Usage: pmc_to_json.py OUT.json KERNEL_SUBSTRING PASSES DIR [DIR ...]
Each DIR holds the output of one rocprofv3 --pmc run of `scratch/prof_step.py 1000000 fused N`
(the kernel is launched PASSES times per optimiser step, in pass order)."""
import collections, csv, glob, json, os, statistics, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench

out, kernel, passes = sys.argv[1], sys.argv[2], int(sys.argv[3])
per_pass = [collections.defaultdict(list) for _ in range(passes)]
launches = 0
for d in sys.argv[4:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                per[(int(r["Dispatch_Id"]), r["Counter_Name"])] += float(r["Counter_Value"])
        ids = sorted({k[0] for k in per})
        ids = ids[len(ids) % passes:]                      # whole steps
        ids = ids[(len(ids) // passes // 3) * passes:]     # drop the first third (warm-up steps)
        launches = max(launches, len(ids))
        for n, did in enumerate(ids):
            for (dd, c), v in per.items():
                if dd == did:
                    per_pass[n % passes][c].append(v)
doc = {
    "kernel": kernel, "collected": time.strftime("%Y-%m-%d"), "source_hash": bench._source_hash(),
    "workload": os.environ.get("PMC_WORKLOAD") or
    "bench.py default (cfg4: 1,000,000 rays x 10,574 faces, f32 state), eager fused step, "
    "rocprofv3 --kernel-trace --pmc, one counter set per run",
    "launches_per_pass": launches // passes,
    "units": "per launch, summed over XCDs/SEs; medians over the launches of a pass; FETCH_SIZE / "
             "WRITE_SIZE in KB as reported (FETCH_SIZE counts half of wide reads on gfx950)",
    "passes": [],
}
for p in range(passes):
    row = {"pass": p}
    for c, v in sorted(per_pass[p].items()):
        key = c + ("_KB" if c in ("FETCH_SIZE", "WRITE_SIZE") else "")
        row[key] = statistics.median(v)
    doc["passes"].append(row)
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc, indent=1))
