"""C4-shaped run over whole episodes, one JSON line per tick: how the cost of a simulated step grows with the
number of entries a history particle holds.  python scripts/c4_episode.py [sims] [slots] [ticks]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fba_pomdp_amd as fba
from fba_pomdp_amd import _native as N
sims = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 42
eng = fba.Engine("gridworld", model=N.MODEL_BA_FACTORED, belief="importance_sampling", size=7, sims=sims, particles=16384,
                 structure_prior=2, horizon=20, episodes=2, slots=slots, runs=1 << 30, seed=7)
tot_steps, tot_t = 0, 0.0
for k in range(ticks):
    c0 = eng.counters()
    t0 = time.perf_counter()
    eng.run_ticks(1)
    dt = time.perf_counter() - t0
    c1 = eng.counters()
    steps = (c1.sim_steps - c0.sim_steps) + (c1.belief_steps - c0.belief_steps)
    tot_steps += steps; tot_t += dt
    print(json.dumps({"tick": k, "steps": steps, "ms": 1e3 * dt, "steps_per_s": steps / dt, "us_per_step_per_tree": 1e6 * dt / (steps / slots)}), flush=True)
print(json.dumps({"ticks": ticks, "slots": slots, "sims": sims, "steps_per_s_overall": tot_steps / tot_t}))
