"""A short C4-shaped run for profiling: python scripts/c4_probe.py [sims] [slots] [ticks] [particles]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fba_pomdp_amd as fba
from fba_pomdp_amd import _native as N
sims = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 2
parts = int(sys.argv[4]) if len(sys.argv) > 4 else 16384
eng = fba.Engine("gridworld", model=N.MODEL_BA_FACTORED, belief="importance_sampling", size=7, sims=sims, particles=parts,
                 structure_prior=2, horizon=20, episodes=2, slots=slots, runs=1 << 30, seed=7)
eng.run_ticks(1)
c0 = eng.counters()
eng.reset_kernel_times()
t0 = time.perf_counter()
eng.run_ticks(ticks)
dt = time.perf_counter() - t0
c1 = eng.counters()
kt = eng.kernel_times()
steps = (c1.sim_steps - c0.sim_steps) + (c1.belief_steps - c0.belief_steps)
print(json.dumps({"sims": sims, "slots": slots, "ticks": ticks, "particles": parts, "steps": steps, "steps_per_s": steps / dt,
                  "ms_per_tick": 1e3 * dt / ticks, "search_ms": kt["search_kernel"].ms / ticks,
                  "importance_ms": kt["importance_kernel"].ms / ticks,
                  "us_per_step_per_lane": 1e6 * dt / (steps / slots)}))
