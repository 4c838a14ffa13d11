"""What a Metropolis-Hastings re-draw of the filter costs (-B mh-within-gibbs / mh-nips): the belief-update time of whole experiments
with the threshold that triggers re-draws against the same experiments with a threshold that is never reached, on the engine
(all runs concurrent, one lane per chain) and on the CPU oracle (one run after the other).
python scripts/mh_timing.py [slots]   -> one JSON line per configuration"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fba_pomdp_amd as fba
from fba_pomdp_amd import _native as N
from oracle import pyorc as orc

slots = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
DOM = {"continuous-factored-tiger": orc.DOM_FTIGER_CONTINUOUS, "random-collision-avoidance": orc.DOM_COLLISION_AVOID, "gridworld": orc.DOM_GRIDWORLD}
CASES = [
    ("mh-within-gibbs", "continuous-factored-tiger", 0, 2, -1.0, dict(size=2, particles=48, sims=80, horizon=8)),
    ("mh-within-gibbs", "continuous-factored-tiger", 1, 2, -1.0, dict(size=2, particles=48, sims=80, horizon=8)),
    ("mh-nips", "continuous-factored-tiger", 0, 2, -1.0, dict(size=2, particles=48, sims=80, horizon=8)),
    ("mh-within-gibbs", "random-collision-avoidance", 0, 2, -2.0, dict(width=3, height=3, size=1, particles=24, sims=48, horizon=5)),
    ("mh-within-gibbs", "gridworld", 0, 2, -3.0, dict(size=3, particles=64, sims=48, horizon=6)),
]
for belief, domain, option, sp, thr, kw in CASES:
    row = {"belief": belief, "option": "rs" if option else "", "domain": domain, "slots": slots, **kw}
    for name, t in (("redraws", thr), ("never", -1e9)):
        eng = fba.Engine(domain, model=N.MODEL_BA_FACTORED, belief=belief, seed=33, slots=slots, runs=slots, episodes=3, structure_prior=sp,
                         threshold=t, belief_option=option, **kw)
        eng.reset_kernel_times()
        t0 = time.perf_counter()
        eng.run_bapomdp()
        row[f"gpu_{name}_wall_s"] = time.perf_counter() - t0
        kt = eng.kernel_times()["importance_kernel"]
        row[f"gpu_{name}_update_ms_total"] = kt.ms
        row["updates"] = int(kt.units // kw["particles"])
        eng.close()
    cpu_runs = 16
    for name, t in (("redraws", thr), ("never", -1e9)):
        o = orc.Oracle(domain=DOM[domain], model=orc.MODEL_BA_FACTORED, belief=N.BELIEF_NAMES[belief], rng_mode=orc.RNG_PHILOX, arith=orc.ARITH_DEV,
                       philox_seed=33, runs=cpu_runs, episodes=3, structure_prior=sp, threshold=t, belief_option=option, **kw)
        t0 = time.perf_counter()
        o.run_bapomdp()
        row[f"cpu_{name}_s_per_run"] = (time.perf_counter() - t0) / cpu_runs
    upd_per_run = row["updates"] / slots
    row["gpu_ms_per_redraw_update_all_slots"] = (row["gpu_redraws_update_ms_total"] - row["gpu_never_update_ms_total"]) / max(upd_per_run, 1e-9)
    row["gpu_us_per_update_and_chain_per_run"] = 1e3 * (row["gpu_redraws_update_ms_total"] - row["gpu_never_update_ms_total"]) / max(row["updates"], 1)
    row["cpu_ms_per_update_and_chain"] = 1e3 * (row["cpu_redraws_s_per_run"] - row["cpu_never_s_per_run"]) / max(upd_per_run, 1e-9)
    print(json.dumps(row), flush=True)
