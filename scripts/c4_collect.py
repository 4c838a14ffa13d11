"""Turn gpurun_out/final/ (scripts/c4_profiles.sh) into profiles/<tag>_c4_counters.json and <tag>_c4_kernel_stats.csv.
python scripts/c4_collect.py [tag]   (default r04)"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "final")
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
dst = lambda name: os.path.join(ROOT, "profiles", f"{tag}_{name}")


def last_json(path):
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def counters(sub):
    """{kernel short name: {counter: sum over launches and instances, 'launches': n}}"""
    out = {}
    for path in glob.glob(os.path.join(SRC, sub, "**", "*counter_collection.csv"), recursive=True):
        seen = {}
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"]
            short = next((n for n in ("search_hist2_kernel", "is_multi_step_kernel", "is_multi_resample_kernel") if n in k), None)
            if not short:
                continue
            d = out.setdefault(short, {})
            d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            seen.setdefault(short, set()).add(row["Dispatch_Id"])
        for k, ids in seen.items():
            out[k]["launches"] = len(ids)
    return out


kern = {}
for sub in ("c4_sq", "c4_sq2", "c4_fetch", "c4_write"):
    for k, d in counters(sub).items():
        kern.setdefault(k, {}).update(d)
line = last_json(os.path.join(SRC, "c4_sq.log"))
s = kern["search_hist2_kernel"]
# the SQ pass's own bench line covers its 2 timed ticks; the counters cover every launch of the run (warm-up included)
steps = line["search_kernel"]["steps_per_launch"] * s["launches"]
derived = {
    "launches": s["launches"],
    "simulated_steps_of_those_launches (steps_per_launch of the SQ pass's bench line x launches: an estimate, the line covers the 2 timed ticks)": steps,
    "valu_instructions_per_16_simulated_steps": s["SQ_INSTS_VALU"] / steps * 16,
    "salu_instructions_per_16_simulated_steps": s["SQ_INSTS_SALU"] / steps * 16,
    "active_inst_valu_over_wave_cycles": s["SQ_ACTIVE_INST_VALU"] / s["SQ_WAVE_CYCLES"],
    "simd_valu_utilisation_at_three_waves_per_simd": 3 * s["SQ_ACTIVE_INST_VALU"] / s["SQ_WAVE_CYCLES"],
    "wait_any_over_wave_cycles": s["SQ_WAIT_ANY"] / s["SQ_WAVE_CYCLES"],
    "wait_inst_any_over_wave_cycles": s["SQ_WAIT_INST_ANY"] / s["SQ_WAVE_CYCLES"],
    "fetched_bytes_per_step (FETCH_SIZE [KB] x 1024 x 2: a 128-byte line request is tallied at 64 B)": s["FETCH_SIZE"] * 1024 * 2 / steps,
    "written_bytes_per_step (WRITE_SIZE [KB] x 1024)": s["WRITE_SIZE"] * 1024 / steps,
}
upd = {}
for k in ("is_multi_step_kernel", "is_multi_resample_kernel"):
    if k in kern and "FETCH_SIZE" in kern[k]:
        upd[k] = {"launches": kern[k]["launches"],
                  "fetched_GB_per_launch (x 2)": kern[k]["FETCH_SIZE"] * 1024 * 2 / kern[k]["launches"] / 1e9,
                  "written_GB_per_launch": kern[k]["WRITE_SIZE"] * 1024 / kern[k]["launches"] / 1e9}
json.dump({
    "command": "scripts/c4_profiles.sh: rocprofv3 --kernel-trace --pmc <SQ_* | SQ_INSTS_VMEM_RD ... | FETCH_SIZE | WRITE_SIZE> --output-format csv -- python3 bench.py "
               "--workload c4 --steps 2 --warmup 8 --no-cpu-baseline  (one pass per counter group; sums over all launches of the run, warm-up included; SQ cycle "
               "counters in quad-cycles summed over waves; 49 152 slots, three waves per SIMD, two rollout steps per iteration, lock-step waves; collected by scripts/c4_collect.py)",
    "kernels": kern, "search_hist2_kernel_derived": derived, "belief_update_kernels_derived": upd, "bench_line_of_the_sq_pass": line,
}, open(dst("c4_counters.json"), "w"), indent=1)
stats = glob.glob(os.path.join(SRC, "c4_stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], dst("c4_kernel_stats.csv"))
print(json.dumps(derived, indent=1))
print(json.dumps(upd, indent=1))
