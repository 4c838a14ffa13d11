"""Per-launch rocprofv3 --pmc counters of one kernel: python scripts/pmc_per_launch.py <dir> <kernel-substring>"""
import csv, glob, sys, collections
d, sub = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(dict)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if sub in row["Kernel_Name"]:
            rows[int(row["Dispatch_Id"])][row["Counter_Name"]] = float(row["Counter_Value"])
names = sorted({k for r in rows.values() for k in r})
print("dispatch " + " ".join(f"{n:>18s}" for n in names))
for did in sorted(rows):
    print(f"{did:8d} " + " ".join(f"{rows[did].get(n, 0):18.4g}" for n in names))
