"""The literal drop-in: ONE (planner, belief) pair behind fba_select_action / fba_belief_update, as the
reference's planning / bapomdp executables would drive it -- milliseconds per call at slots = 1, beside
the CPU oracle's time for the same call (mt19937 mode = the reference's arithmetic, one core).
python scripts/latency_slots1.py [c1|c2 ...]  ->  one JSON line per config"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fba_pomdp_amd as fba
from fba_pomdp_amd import _native as N
from oracle import pyorc as orc

CONFIGS = {
    "c1": dict(domain="episodic-tiger", odomain=orc.DOM_TIGER_EPISODIC, model=N.MODEL_POMDP, sims=1024, particles=256),
    "c2": dict(domain="episodic-tiger", odomain=orc.DOM_TIGER_EPISODIC, model=N.MODEL_BA_TABLE, sims=4096, particles=4096),
}

for name in (sys.argv[1:] or ["c1", "c2"]):
    cfg = dict(CONFIGS[name])
    domain, odomain = cfg.pop("domain"), cfg.pop("odomain")
    eng = fba.Engine(domain, seed=3, slots=1, horizon=10, **cfg)
    eng.belief_init()
    if cfg["model"] != N.MODEL_POMDP:
        eng.belief_reset_domain_state()
    reps = 20
    sel, upd = [], []
    for k in range(reps + 2):
        eng.set_position(run=k, episode=0, t=0)
        t0 = time.perf_counter()
        eng.select_action(hist_len=0)
        t1 = time.perf_counter()
        eng.belief_update(2, 0)   # listen, hear left: never terminal
        t2 = time.perf_counter()
        if k >= 2:
            sel.append(t1 - t0)
            upd.append(t2 - t1)
    kt = eng.kernel_times()
    # the CPU oracle, same calls
    o = orc.Oracle(domain=odomain, model=cfg["model"], rng_mode=orc.RNG_MT, seed_str="3", horizon=10,
                   sims=cfg["sims"], particles=cfg["particles"])
    o.belief_initiate()
    if cfg["model"] != N.MODEL_POMDP:
        o.belief_reset_domain_state()
    osel, oupd = [], []
    for k in range(reps):
        t0 = time.perf_counter()
        o.select_action(0)
        t1 = time.perf_counter()
        o.belief_update(2, 0)
        t2 = time.perf_counter()
        osel.append(t1 - t0)
        oupd.append(t2 - t1)
    print(json.dumps({
        "config": name, "slots": 1, "sims": cfg["sims"], "particles": cfg["particles"],
        "gpu_ms_per_select_action": 1e3 * float(np.median(sel)), "gpu_ms_per_belief_update": 1e3 * float(np.median(upd)),
        "gpu_kernel_ms_per_search": kt["search_kernel"].ms / max(kt["search_kernel"].launches, 1),
        "cpu_oracle_ms_per_select_action": 1e3 * float(np.median(osel)), "cpu_oracle_ms_per_belief_update": 1e3 * float(np.median(oupd)),
        "note": "host wall time per C-ABI call (launch + sync + copies included); oracle = oracle/orc.c -O2, mt19937 mode, one core",
    }), flush=True)
    eng.close()
