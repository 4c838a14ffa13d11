"""Turn gpurun_out/refresh/ (scripts/refresh_profiles.sh) into the committed profiles/ files."""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "refresh")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"

shutil.copy(os.path.join(SRC, "bench_default.json"), os.path.join(DST, f"{tag}_bench_default.json.log"))
shutil.copy(os.path.join(SRC, "stats", "run_kernel_stats.csv"), os.path.join(DST, f"{tag}_bench_kernel_stats.csv"))


def per_kernel(path, counter):
    """KiB per launch, summed over the counter's instances (XCDs), per kernel name"""
    by = {}
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        key = (row["Kernel_Name"], row["Dispatch_Id"])
        by[key] = by.get(key, 0.0) + float(row["Counter_Value"])
    out = {}
    for (k, _), v in sorted(by.items(), key=lambda kv: int(kv[0][1])):
        out.setdefault(k, []).append(v)
    return out


fetch = per_kernel(os.path.join(SRC, "fetch", "run_counter_collection.csv"), "FETCH_SIZE")
write = per_kernel(os.path.join(SRC, "write", "run_counter_collection.csv"), "WRITE_SIZE")
rk = [k for k in fetch if "reject_tiger_lds_kernel" in k or "reject_kernel" in k][0]
# full launches only (the warm-up tick and the last, partly idle one move fewer bytes)
f_full = max(fetch[rk]) * 1024.0
w_full = max(write[rk]) * 1024.0
# FETCH_SIZE calibrated on this kernel's two access shapes (profiles/r02_pmc_calibration.json, scripts/micro/pmc_calibrate.hip):
# the parking pass touches every 64-byte record of the filter once -- known bytes, reported at 1 / park_factor of them --
# and what the counter saw beyond that is the gather's 64-byte record reads, tallied at 64 B per 128-byte line moved (x 2)
cal = json.load(open(os.path.join(DST, "r02_pmc_calibration.json")))
park_factor = cal["factors"]["park_kernel"]["true_over_reported_fetch"]
park_true = w_full  # slots x N x 64 B: the filter read once = the bytes written
f_cal = park_true + 2.0 * max(f_full - park_true / park_factor, 0.0)
doc = {
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline",
    "config": "default bench workload, 262144 slots, packed particles (64 B records)",
    "kernel": "reject_kernel",  # bench.py's name for the rejection update, whichever instantiation runs it
    "kernel_instantiation": rk,
    "fetch_bytes_per_launch_raw": f_full,
    "write_bytes_per_launch": w_full,
    "traffic_bytes_per_launch_raw": f_full + w_full,
    "traffic_bytes_per_launch_fetch_x2": 2 * f_full + w_full,
    "fetch_bytes_per_launch_calibrated": f_cal,
    "traffic_bytes_per_launch_calibrated": f_cal + w_full,
    "notes": [
        "counter unit KiB (MI355X_MICROARCH.md, HBM / rocprofv3): values below are KiB per launch summed over instances; the per-launch figure used is the largest (a launch in which every slot updates)",
        "WRITE_SIZE is exact on gfx950: 262144 slots x 4096 particles x 64 B = 68.72 GB is the minimum this kernel can write",
        "FETCH_SIZE calibrated with scripts/micro/pmc_calibrate (profiles/r02_pmc_calibration.json): the parking pass (4-byte words of every 64-byte record) reports 1 / %.2f of its bytes, the gather (16 B x 4 lanes, random 64-byte records) is tallied at 64 B per 128-byte line moved; calibrated fetch = slots x N x 64 + 2 x (raw - slots x N x 64 / %.2f)" % (park_factor, park_factor),
    ],
    "per_launch_KiB": {k: {"fetch_KiB_per_launch": fetch.get(k, []), "write_KiB_per_launch": write.get(k, [])}
                       for k in fetch if "search_kernel" in k or "reject_kernel" in k or "reject_tiger_lds_kernel" in k},
}
with open(os.path.join(DST, f"{tag}_pmc_fetch_write.json"), "w") as f:
    json.dump(doc, f, indent=1)
print(json.dumps({k: doc[k] for k in ("kernel_instantiation", "fetch_bytes_per_launch_raw", "write_bytes_per_launch", "traffic_bytes_per_launch_raw")}, indent=1))
print(open(os.path.join(DST, f"{tag}_bench_kernel_stats.csv")).read()[:900])
