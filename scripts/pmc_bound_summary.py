"""Per-kernel sums of the counter passes of scripts/pmc_bound_probe.sh (gpurun_out/bound/*/run_counter_collection.csv) -> JSON on stdout."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "bound")
out = collections.defaultdict(dict)
for path in sorted(glob.glob(os.path.join(root, "*", "run_counter_collection.csv"))):
    acc = collections.defaultdict(float)
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        key = "search_kernel" if "search_kernel" in k else ("reject_tiger_lds_kernel" if "reject_tiger" in k else None)
        if key:
            acc[(key, r["Counter_Name"])] += float(r["Counter_Value"])
            n[key].add(r["Dispatch_Id"])
    for (key, c), v in acc.items():
        out[key][c] = v
        out[key]["launches"] = len(n[key])
print(json.dumps(out, indent=1, sort_keys=True))
