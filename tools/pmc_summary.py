"""Fold rocprofv3 counter_collection CSVs into per-kernel per-launch averages."""
import csv, glob, json, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:60]
        c = row["Counter_Name"]
        acc[k][c] += float(row["Counter_Value"])
        cnt[k][c] += 1
out = {}
for k in acc:
    if not any(t in k for t in ("residual", "update", "norm", "lusgs", "visc", "dplur", "implicit", "matrix", "bc_", "halo",
                                "rans", "sweep_records", "block_diag", "store_time")):
        continue
    out[k] = {c: acc[k][c] / cnt[k][c] for c in sorted(acc[k])}
    out[k]["launches_seen"] = max(cnt[k].values())
print(json.dumps(out, indent=1))
