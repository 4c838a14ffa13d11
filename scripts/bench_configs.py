"""Throughput of the BASELINE.json configs other than the bench one (which bench.py measures):
simulated env steps/s (planner + belief) over a few ticks, slots sized to what fits.
python scripts/bench_configs.py [c1|c3|c4|c4small ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fba_pomdp_amd as fba
from fba_pomdp_amd import _native as N

CONFIGS = {
    # planning -D episodic-tiger, POMCP 1024 sims, 256 particles (the reference's CPU-runnable case)
    "c1": dict(domain="episodic-tiger", model=N.MODEL_POMDP, belief="rejection_sampling", sims=1024, particles=256,
               horizon=10, slots=262144, ticks=8),
    # fbapomdp -D episodic-factored-tiger --size 3, 16384 sims, match-uniform structure prior
    "c3": dict(domain="episodic-factored-tiger", model=N.MODEL_BA_FACTORED, belief="rejection_sampling", size=3, sims=16384,
               particles=4096, structure_prior=2, horizon=10, slots=32768, ticks=3),
    # fbapomdp -D gridworld --size 7, 65536 sims, 16384 particles, importance sampling
    "c3x2": dict(domain="episodic-factored-tiger", model=N.MODEL_BA_FACTORED, belief="rejection_sampling", size=3, sims=16384,
                 particles=4096, structure_prior=2, horizon=10, slots=81920, ticks=2),   # 5 search waves per CU: what LDS holds (22.5 KB staged record per wave)
    "c3half": dict(domain="episodic-factored-tiger", model=N.MODEL_BA_FACTORED, belief="rejection_sampling", size=3, sims=16384,
                   particles=4096, structure_prior=2, horizon=10, slots=16384, ticks=2),
    # (two episodes per run: a history particle holds episodes * (horizon + 1) entries, fba_device.h)
    "c4": dict(domain="gridworld", model=N.MODEL_BA_FACTORED, belief="importance_sampling", size=7, sims=65536,
               particles=16384, structure_prior=2, horizon=20, episodes=2, slots=16384, ticks=2),
    # the parity-sized variant of c4 (--size 5), fewer simulations so that a tick is short
    "c4small": dict(domain="gridworld", model=N.MODEL_BA_FACTORED, belief="importance_sampling", size=5, sims=8192,
                    particles=4096, structure_prior=2, horizon=20, slots=512, ticks=2),
}

for name in (sys.argv[1:] or ["c1", "c3", "c4small"]):
    cfg = dict(CONFIGS[name])
    if os.environ.get("FBA_SLOTS"):   # try another number of concurrent runs
        cfg["slots"] = int(os.environ["FBA_SLOTS"])
    ticks = cfg.pop("ticks")
    if os.environ.get("FBA_TICKS"):
        ticks = int(os.environ["FBA_TICKS"])
    domain = cfg.pop("domain")
    cfg.setdefault("episodes", 1 if cfg['model'] == N.MODEL_POMDP else 64)
    t0 = time.perf_counter()
    eng = fba.Engine(domain, runs=1 << 30, seed=7, **cfg)
    eng.run_ticks(1)
    c0 = eng.counters()
    eng.reset_kernel_times()
    t1 = time.perf_counter()
    eng.run_ticks(ticks)
    dt = time.perf_counter() - t1
    c1 = eng.counters()
    kt = eng.kernel_times()
    steps = (c1.sim_steps - c0.sim_steps) + (c1.belief_steps - c0.belief_steps)
    out = {"config": name, "workload": {k: v for k, v in cfg.items()}, "domain": domain, "slots": eng.slots, "ticks": ticks,
           "particle_bytes": eng.particle_bytes,
           "steps_per_s": steps / dt, "ms_per_tick": 1e3 * dt / ticks, "setup_s": t1 - t0,
           "kernels_ms_per_tick": {k: v.ms / ticks for k, v in kt.items() if v.ms > 0}}
    bel = kt["reject_kernel"] if cfg["belief"] == "rejection_sampling" else kt["importance_kernel"]
    if bel.ms > 0:
        out["belief_kernel_algorithmic_GBs"] = bel.bytes / 1e9 / (bel.ms / 1e3)
    print(json.dumps(out), flush=True)
    eng.close()
