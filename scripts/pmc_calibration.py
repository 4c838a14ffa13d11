"""profiles/r02_pmc_calibration.json from the two rocprofv3 passes over scripts/micro/pmc_calibrate:
python scripts/pmc_calibration.py <fetch-dir> <write-dir> <program-stdout.json> <out.json>"""
import csv, glob, json, sys

def per_kernel(d, counter):
    by = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            k = row["Kernel_Name"].split("(")[0].split("::")[-1].split(" ")[-1]
            by.setdefault(k, {}).setdefault(row["Dispatch_Id"], 0.0)
            by[k][row["Dispatch_Id"]] += float(row["Counter_Value"])
    return {k: [v * 1024.0 for _, v in sorted(d.items(), key=lambda kv: int(kv[0]))] for k, d in by.items()}

fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
known = json.load(open(sys.argv[3]))
out = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- scripts/micro/pmc_calibrate",
       "known": known, "counters_bytes_per_launch": {"FETCH_SIZE": fetch, "WRITE_SIZE": write}, "factors": {}}
for k in ("park_kernel", "gather_kernel", "stream_kernel"):
    f = max(fetch.get(k, [0.0])); w = max(write.get(k, [0.0]))
    e = {"FETCH_SIZE_bytes": f, "WRITE_SIZE_bytes": w}
    kn = known[k]
    if "fetch_bytes_known" in kn:
        e["true_over_reported_fetch"] = kn["fetch_bytes_known"] / f if f else None
    else:
        e["reported_over_all_reads"] = f / kn["fetch_bytes_if_every_read_misses"]
        e["reported_over_distinct_records"] = f / kn["fetch_bytes_distinct_records"]
    if kn.get("write_bytes_known"):
        e["true_over_reported_write"] = kn["write_bytes_known"] / w if w else None
    out["factors"][k] = e
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out["factors"], indent=1))
