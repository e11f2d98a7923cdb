"""Reduce a rocprofv3 counter_collection.csv to per-kernel means of every counter (row-AND kernels only)."""
import csv, json, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if "k_and_exact" in name: key = "k_and_exact"
    elif "k_and_count" in name: key = "k_and_count"
    else: continue
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in d.items()} for k, d in acc.items()}
print(json.dumps(out))
