"""Sum rocprofv3 --pmc counter_collection.csv per kernel name prefix: python scripts/pmc_summary.py <dir> [kernel-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "search_kernel"
tot = collections.defaultdict(float)
n = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if sub in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
            n[row["Counter_Name"]] += 1
for k in sorted(tot):
    print(f"{k:28s} {tot[k]:.4g}  (rows {n[k]})")
