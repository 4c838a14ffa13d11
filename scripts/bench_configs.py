"""Throughput of the BASELINE.json configs other than the bench one (which bench.py measures):
simulated env steps/s (planner + belief) over a few ticks, slots sized to what fits.
python scripts/bench_configs.py [c1|c3|c4|c4small ...]
FBA_SLOTS / FBA_TICKS override a config's slots / ticks; FBA_CPU=1 also times the CPU oracle (16 processes,
mt19937 mode = the reference's arithmetic) on a bounded sample of the same shape."""
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
                 particles=4096, structure_prior=2, horizon=10, slots=163840, ticks=2),   # 10 search waves per CU: what LDS holds (packed records: 15 KB per wave)
    "c3half": dict(domain="episodic-factored-tiger", model=N.MODEL_BA_FACTORED, belief="rejection_sampling", size=3, sims=16384,
                   particles=4096, structure_prior=2, horizon=10, slots=16384, ticks=2),
    # (two episodes per run: a history particle holds episodes * (horizon + 1) entries, fba_device.h)
    "c4": dict(domain="gridworld", model=N.MODEL_BA_FACTORED, belief="importance_sampling", size=7, sims=65536,
               particles=16384, structure_prior=2, horizon=20, episodes=2, slots=49152, tree_buckets=32768, ticks=2),   # (5.5 MB per slot; lock-step ticks: bench.py --workload c4 is the budgeted form)
    # the parity-sized variant of c4 (--size 5), fewer simulations so that a tick is short
    "c4small": dict(domain="gridworld", model=N.MODEL_BA_FACTORED, belief="importance_sampling", size=5, sims=8192,
                    particles=4096, structure_prior=2, horizon=20, slots=512, ticks=2),
}

def cpu_worker(name, seed):
    """One process of the CPU comparison: the oracle on the config's shape, fewer simulations and particles per
    belief so that it finishes in seconds (its time per simulated step does not depend on either)."""
    from oracle import pyorc as orc
    dom = {"gridworld": orc.DOM_GRIDWORLD, "episodic-tiger": orc.DOM_TIGER_EPISODIC, "episodic-factored-tiger": orc.DOM_FTIGER_EPISODIC}
    cfg = dict(CONFIGS[name])
    kw = dict(domain=dom[cfg["domain"]], model=cfg["model"], belief=N.BELIEF_NAMES[cfg["belief"]], size=cfg.get("size", 0),
              structure_prior=cfg.get("structure_prior", 0), horizon=cfg["horizon"], sims=min(cfg["sims"], 2048),
              particles=min(cfg["particles"], 1024), runs=8, episodes=1, seed_str=f"cfg-{seed}")
    o = orc.Oracle(**kw)
    t0 = time.perf_counter()
    if cfg["model"] == N.MODEL_POMDP:
        _, res = o.run_planning()
    else:
        _, res = o.run_bapomdp()
    print(json.dumps({"steps": res.sim_steps + res.belief_steps, "seconds": time.perf_counter() - t0, "sims": kw["sims"],
                      "particles": kw["particles"]}), flush=True)


def cpu_compare(name, procs=16):
    import subprocess
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", name, str(k)], stdout=subprocess.PIPE,
                           stderr=subprocess.DEVNULL, text=True) for k in range(procs)]
    out = []
    for p in ps:
        try:
            so, _ = p.communicate(timeout=300)
            out.append(json.loads(so.strip().splitlines()[-1]))
        except Exception:
            p.kill()
    if not out:
        return None
    return {"value": sum(o["steps"] for o in out) / max(o["seconds"] for o in out), "unit": "simulated env steps/s", "cores": len(out),
            "kind": "port", "sample": f"{len(out)} processes x 8 runs x 1 episode, {out[0]['sims']} simulations, {out[0]['particles']} particles "
                                      f"per belief, same domain / model / horizon; longest process {max(o['seconds'] for o in out):.1f} s"}


if len(sys.argv) > 3 and sys.argv[1] == "--cpu-worker":
    cpu_worker(sys.argv[2], int(sys.argv[3]))
    sys.exit(0)

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
    eng.close()
    if os.environ.get("FBA_CPU"):
        out["cpu_baseline"] = cpu_compare(name)
        if out["cpu_baseline"]:
            out["gpu_over_cpu"] = out["steps_per_s"] / out["cpu_baseline"]["value"]
    print(json.dumps(out), flush=True)
