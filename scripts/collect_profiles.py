"""Turn gpurun_out/refresh/ (scripts/refresh_profiles.sh) into the committed profiles/<tag>_* files.
python scripts/collect_profiles.py [tag]      (default r04)"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "refresh")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"


def dst(name):
    return os.path.join(DST, f"{tag}_{name}")


def per_kernel(path, counter):
    """counter value per launch, summed over the counter's instances, per kernel name: {kernel: [v of launch 1, ...]}"""
    by = {}
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        key = (row["Kernel_Name"], int(row["Dispatch_Id"]))
        by[key] = by.get(key, 0.0) + float(row["Counter_Value"])
    out = {}
    for (k, _), v in sorted(by.items(), key=lambda kv: kv[0][1]):
        out.setdefault(k, []).append(v)
    return out


def last_json(path):
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def find(d, sub):
    ks = [k for k in d if sub in k]
    return ks[0] if ks else None


shutil.copy(os.path.join(SRC, "bench_default.json"), dst("bench_default.json.log"))
shutil.copy(os.path.join(SRC, "stats", "run_kernel_stats.csv"), dst("bench_kernel_stats.csv"))
if os.path.exists(os.path.join(SRC, "bench_importance.json")):
    shutil.copy(os.path.join(SRC, "bench_importance.json"), dst("bench_importance_sampling.json.log"))
with open(dst("other_configs.jsonl"), "w") as f:
    for wl in ("c1", "c3", "c4", "c5"):
        p = os.path.join(SRC, f"bench_{wl}.json")
        if os.path.exists(p) and os.path.getsize(p):
            f.write(open(p).read().strip() + "\n")

fetch = per_kernel(os.path.join(SRC, "fetch", "run_counter_collection.csv"), "FETCH_SIZE")     # KiB
write = per_kernel(os.path.join(SRC, "write", "run_counter_collection.csv"), "WRITE_SIZE")     # KiB
KIB = 1024.0

# ---- the random-sector micro-benchmark: what a counter reports for one random 64-byte record per lane, and the rate the part sustains
rl = {}
if os.path.exists(os.path.join(SRC, "randline.jsonl")):
    shutil.copy(os.path.join(SRC, "randline.jsonl"), dst("randline.jsonl"))
    rows = [json.loads(l) for l in open(os.path.join(SRC, "randline.jsonl")) if l.startswith("{")]
    shapes = [r for r in rows if "shape" in r]
    accesses = 4096 * 64 * 1024   # of the counter passes (iters 1024)
    cnt = {}
    for name, d, counter in (("fetch_B", "rl_fetch", "FETCH_SIZE"), ("write_B", "rl_write", "WRITE_SIZE"), ("rdreq", "rl_l2", "TCC_EA0_RDREQ_sum"),
                             ("rdreq_32B", "rl_l2", "TCC_EA0_RDREQ_32B_sum"), ("l2_miss", "rl_l2", "TCC_MISS_sum"), ("l2_hit", "rl_l2", "TCC_HIT_sum")):
        pk = per_kernel(os.path.join(SRC, d, "run_counter_collection.csv"), counter)
        for k, v in pk.items():
            # every shape runs a warm-up launch and a timed one: the timed one is the second
            cnt.setdefault(k, {})[name] = v[-1] * (KIB if name.endswith("_B") else 1.0) / accesses
    best = lambda name: max(s["G_records_per_s"] for s in shapes if s["shape"] == name)
    rl = {"shapes": shapes, "per_access_counters_of_one_XCD_instance_sum": cnt,
          "note": "rocprofv3 sums the TCC counters it collects; on this box the per-access figures come out at 1/8 of what a shape asks for "
                  "(one XCD's share): load16 reports 8.0 B of FETCH_SIZE per access = 64 B x 1/8 and TCC_EA0_RDREQ 0.125 per access.  pair64 (two lanes, "
                  "the two 64-byte halves of ONE random 128-byte line) makes HALF a request per lane access and runs at twice load16's rate; line128 (one "
                  "lane, both halves) one request for 128 bytes.  So a read request is a 128-byte line, FETCH_SIZE tallies it at 64 B, and fetched "
                  "bytes are FETCH_SIZE x 2 for every access shape (the guide's rule); a store into a line that is not resident fetches nothing.",
          "G_records_per_s": {k: best(k) for k in ("load16", "load64", "pair64", "line128", "far2x64") if any(s["shape"] == k for s in shapes)}}
    json.dump(rl, open(dst("randline_counters.json"), "w"), indent=1)
    load16 = max(s["G_records_per_s"] for s in shapes if s["shape"] == "load16")
else:
    raise SystemExit("gpurun_out/refresh/randline.jsonl is missing: the search kernel's ceiling (random lines per second) is measured, not assumed")

# ---- the search kernel: memory-side bytes, L2 requests and misses per simulated step
sk = find(fetch, "search_kernel")
line = last_json(os.path.join(SRC, "fetch.log"))
if sk and line:
    launches = len(fetch[sk])
    steps = line["search_kernel"]["steps_per_launch"] * launches
    l2 = {c: per_kernel(os.path.join(SRC, "l2", "run_counter_collection.csv"), c) for c in
          ("TCC_MISS_sum", "TCC_HIT_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum")}
    sq = {c: per_kernel(os.path.join(SRC, "sq", "run_counter_collection.csv"), c) for c in
          ("SQ_INSTS_VALU", "SQ_INSTS_VMEM", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_BUSY_CYCLES")}
    f_b, w_b = sum(fetch[sk]) * KIB, sum(write[find(write, "search_kernel")]) * KIB
    doc = {
        "command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE | TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum | SQ_*> "
                   "--output-format csv -- python3 bench.py --steps 4 --warmup 0 --no-cpu-baseline   (one pass per counter group)",
        "kernel": sk, "launches": launches, "simulated_steps_of_those_launches": steps,
        "fetch_bytes_per_step_counter": f_b / steps,
        "fetch_bytes_per_step": 2.0 * f_b / steps, "write_bytes_per_step": w_b / steps,    # (a read request is a 128-byte line tallied at 64 B)
        "fetch_sectors_64B_per_step": f_b / 64.0 / steps,                                   # = read requests (128-byte lines) per step
        "l2_misses_per_step": sum(l2["TCC_MISS_sum"][find(l2["TCC_MISS_sum"], "search_kernel")]) / steps,
        "l2_hits_per_step": sum(l2["TCC_HIT_sum"][find(l2["TCC_HIT_sum"], "search_kernel")]) / steps,
        "l1_read_requests_per_step": sum(l2["TCP_TCC_READ_REQ_sum"][find(l2["TCP_TCC_READ_REQ_sum"], "search_kernel")]) / steps,
        "l1_write_requests_per_step": sum(l2["TCP_TCC_WRITE_REQ_sum"][find(l2["TCP_TCC_WRITE_REQ_sum"], "search_kernel")]) / steps,
        "valu_wave_instructions_per_64_steps": sum(sq["SQ_INSTS_VALU"][find(sq["SQ_INSTS_VALU"], "search_kernel")]) * 64.0 / steps,
        "vmem_wave_instructions_per_64_steps": sum(sq["SQ_INSTS_VMEM"][find(sq["SQ_INSTS_VMEM"], "search_kernel")]) * 64.0 / steps,
        "lds_wave_instructions_per_64_steps": sum(sq["SQ_INSTS_LDS"][find(sq["SQ_INSTS_LDS"], "search_kernel")]) * 64.0 / steps,
        "sq_wait_any_over_wave_cycles": sum(sq["SQ_WAIT_ANY"][find(sq["SQ_WAIT_ANY"], "search_kernel")]) /
                                        max(sum(sq["SQ_WAVE_CYCLES"][find(sq["SQ_WAVE_CYCLES"], "search_kernel")]), 1.0),
        # MODEL estimate, not a measurement: per simulation one 64-byte particle record, per tree level below the two LDS-resident ones a
        # 64-byte node record read and 24 bytes of it written back, at 0.44 simulations and ~0.2 such levels per step
        "algorithmic_bytes_per_step": 0.44 * 64.0 + 0.2 * (64.0 + 24.0),
        "algorithmic_bytes_per_step_is": "a model estimate (simulations per step and HBM tree levels per step of the round-3 analysis), kept for the bench line's frac_algorithmic",
        "random_sector_ceiling_G_per_s": load16,   # random 128-byte LINES per second (scripts/micro/randline load16 of this refresh)
        "fetch_calibration": "fetched bytes = FETCH_SIZE x 2: a read request is one 128-byte line tallied at 64 B (profiles/%s_randline_counters.json)" % tag,
        "per_launch_KiB": {"fetch": fetch[sk], "write": write[find(write, "search_kernel")]},
    }
    json.dump(doc, open(dst("pmc_search.json"), "w"), indent=1)
    with open(dst("l2_counters.txt"), "w") as f:
        f.write(f"# {doc['command']}: search kernel, one row per launch\n")
        names = sorted(l2)
        f.write("launch " + " ".join(f"{n:>22s}" for n in names) + "\n")
        for i in range(launches):
            f.write(f"{i:6d} " + " ".join(f"{l2[n][find(l2[n], 'search_kernel')][i]:22.4g}" for n in names) + "\n")
    with open(dst("sq_counters.txt"), "w") as f:
        f.write(f"# {doc['command']}: sums over the run's launches\n")
        for kname in ("search_kernel", "reject_tiger_lds_kernel"):
            f.write(f"# {kname}\n")
            for n in sorted(sq):
                k = find(sq[n], kname)
                if k:
                    f.write(f"{n:28s} {sum(sq[n][k]):.4g}  (launches {len(sq[n][k])})\n")
    print("search:", {k: round(v, 3) for k, v in doc.items() if isinstance(v, float)})

# ---- the rejection update (the bench's roofline kernel): a launch in which every slot updates
rk = find(fetch, "reject_tiger_lds_kernel") or find(fetch, "reject_kernel")
if rk:
    f_full, w_full = max(fetch[rk]) * KIB, max(write[rk]) * KIB
    cal = json.load(open(os.path.join(DST, "r02_pmc_calibration.json")))
    park_factor = cal["factors"]["park_kernel"]["true_over_reported_fetch"]
    park_true = w_full                       # slots x N x 64 B: the filter read once = the bytes written
    gather_raw = max(f_full - park_true / park_factor, 0.0)
    doc = {
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py --steps 4 --warmup 0 --no-cpu-baseline",
        "config": "default bench workload, 262144 slots, packed particles (64 B records)",
        "kernel": "reject_kernel", "kernel_instantiation": rk,
        "fetch_bytes_per_launch_raw": f_full, "write_bytes_per_launch": w_full, "traffic_bytes_per_launch_raw": f_full + w_full,
        "fetch_bytes_per_launch_calibrated": park_true + 2.0 * gather_raw, "traffic_bytes_per_launch_calibrated": park_true + 2.0 * gather_raw + w_full,
        "traffic_bytes_per_launch_round3_undercount": park_true + gather_raw + w_full,
        "notes": [
            "counter unit KiB; per-launch figure = the largest launch (one in which every slot updates)",
            "WRITE_SIZE is exact on gfx950: slots x N x 64 B is the minimum this kernel can write",
            "FETCH_SIZE: the parking pass (4-byte words of every 64-byte record, records in order: a coalesced sweep) reports 1 / %.2f of its bytes "
            "(profiles/r02_pmc_calibration.json); the gather's random 64-byte record reads are one 128-byte line request each, tallied at 64 B: x 2 "
            "(profiles/%s_randline_counters.json: pair64 / line128).  Round 3 counted them once (kept as traffic_bytes_per_launch_round3_undercount)" % (park_factor, tag),
        ],
    }
    json.dump(doc, open(dst("pmc_fetch_write.json"), "w"), indent=1)
    print("reject:", {k: v for k, v in doc.items() if k.startswith("traffic") or k.startswith("fetch_bytes")})

# ---- the importance filter of the bench workload
if os.path.exists(os.path.join(SRC, "is_fetch", "run_counter_collection.csv")):
    fi = per_kernel(os.path.join(SRC, "is_fetch", "run_counter_collection.csv"), "FETCH_SIZE")
    wi = per_kernel(os.path.join(SRC, "is_write", "run_counter_collection.csv"), "WRITE_SIZE")
    ik = find(fi, "importance_kernel")
    if ik:
        f_full, w_full = max(fi[ik]) * KIB, max(wi[ik]) * KIB
        doc = {
            "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py --belief importance_sampling --steps 4 --warmup 0 --no-cpu-baseline",
            "kernel": "importance_kernel", "kernel_instantiation": ik,
            "fetch_bytes_per_launch_raw": f_full, "write_bytes_per_launch": w_full, "traffic_bytes_per_launch_raw": f_full + w_full,
            # the update pass sweeps the filter's records and weights in order (wide coalesced loads: tallied at half, the guide's x2); the
            # resample's gather reads random 64-byte records (counted in full).  The sweep is slots x N x (64 + 8) bytes.
            "fetch_bytes_per_launch_calibrated": 2.0 * f_full,
            "traffic_bytes_per_launch_calibrated": 2.0 * f_full + w_full,
            "notes": ["per-launch figure = the largest launch (every slot updates)",
                      "calibrated = FETCH_SIZE x 2 + WRITE_SIZE: every read request is a 128-byte line tallied at 64 B (profiles/%s_randline_counters.json)" % tag],
        }
        json.dump(doc, open(dst("pmc_importance.json"), "w"), indent=1)
        print("importance:", {k: v for k, v in doc.items() if k.startswith("traffic")})

# ---- C5 shape
if os.path.exists(os.path.join(SRC, "c5.json")) and os.path.getsize(os.path.join(SRC, "c5.json")):
    c5 = last_json(os.path.join(SRC, "c5.json"))
    fc = per_kernel(os.path.join(SRC, "c5_fetch", "run_counter_collection.csv"), "FETCH_SIZE")
    wc = per_kernel(os.path.join(SRC, "c5_write", "run_counter_collection.csv"), "WRITE_SIZE")
    upd = c5["updates"] + 1   # + the warm-up update
    f_b = sum(sum(v) for k, v in fc.items() if "is_multi" in k or "scan_" in k or "chunk_totals" in k) * KIB / upd
    w_b = sum(sum(v) for k, v in wc.items() if "is_multi" in k or "scan_" in k or "chunk_totals" in k) * KIB / upd
    c5["pmc_fetch_bytes_per_update_raw"] = f_b
    c5["pmc_write_bytes_per_update"] = w_b
    c5["pmc_note"] = ("FETCH_SIZE / WRITE_SIZE of the multi-workgroup filter's kernels (is_multi_*, scan_*, chunk_totals), summed and divided by the updates of the "
                      "run; the records are 3.5 KB and read as wide coalesced pieces, which FETCH_SIZE tallies at half (guide's x2)")
    c5["pmc_traffic_per_update_fetch_x2"] = 2 * f_b + w_b
    c5["frac_of_8TBs_on_pmc_traffic"] = (2 * f_b + w_b) / 1e9 / (c5["ms_per_update"] / 1e3) / 8000.0
    json.dump(c5, open(dst("c5_importance_update.json"), "w"), indent=1)
    shutil.copy(os.path.join(SRC, "c5_stats", "run_kernel_stats.csv"), dst("c5_kernel_stats.csv"))
    print("c5:", c5["ms_per_update"], c5["frac_of_8TBs"], c5["frac_of_8TBs_on_pmc_traffic"])
print(open(dst("bench_kernel_stats.csv")).read()[:700])

# ---- C3's rejection kernel (packed factored-tiger records): fabric traffic of a full launch
if os.path.exists(os.path.join(SRC, "c3_fetch", "run_counter_collection.csv")) and os.path.exists(os.path.join(SRC, "c3_write", "run_counter_collection.csv")):
    f3 = per_kernel(os.path.join(SRC, "c3_fetch", "run_counter_collection.csv"), "FETCH_SIZE")
    w3 = per_kernel(os.path.join(SRC, "c3_write", "run_counter_collection.csv"), "WRITE_SIZE")
    rk3, sk3 = find(f3, "reject_kernel"), find(f3, "search_kernel")
    ln = last_json(os.path.join(SRC, "c3_fetch.log"))
    if rk3 and ln:
        slots3, rec3 = ln["config"]["slots_per_gpu"], ln["roofline"]["particle_bytes_in_hbm"]
        full = max(range(len(w3[rk3])), key=lambda i: w3[rk3][i])   # the launch in which every slot updates writes the most
        doc = {
            "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py --workload c3 --steps 4 --warmup 0 --no-cpu-baseline",
            "config": ln["config"], "kernel": "reject_kernel", "kernel_instantiation": rk3,
            "per_launch_GB": {"fetch_counter": [x * KIB / 1e9 for x in f3[rk3]], "write": [x * KIB / 1e9 for x in w3[rk3]]},
            "full_launch": {"fetch_bytes_counter": f3[rk3][full] * KIB, "fetch_bytes_x2": 2 * f3[rk3][full] * KIB, "write_bytes": w3[rk3][full] * KIB,
                            "traffic_bytes": 2 * f3[rk3][full] * KIB + w3[rk3][full] * KIB, "min_traffic_bytes": 2.0 * slots3 * 4096 * rec3,
                            "note": "the launch in which every slot updates; FETCH_SIZE x 2: a read request is a 128-byte line tallied at 64 B "
                                    "(profiles/%s_randline_counters.json)" % tag},
            "search_kernel": {"kernel_instantiation": sk3, "per_launch_GB": {"fetch_counter": [x * KIB / 1e9 for x in f3[sk3]], "write": [x * KIB / 1e9 for x in w3[sk3]]}} if sk3 else None,
            "bench_line_of_the_fetch_pass": ln,
        }
        json.dump(doc, open(dst("pmc_c3.json"), "w"), indent=1)
        print("c3 reject:", doc["full_launch"])
# ---- C4 steady state: appended to the round's file of C4 lines
p4 = os.path.join(SRC, "bench_c4.json")
if os.path.exists(p4) and os.path.getsize(p4):
    with open(dst("c4_full_size.jsonl"), "a") as f:
        f.write(open(p4).read().strip().splitlines()[-1] + "\n")
