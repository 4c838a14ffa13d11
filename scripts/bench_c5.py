"""BASELINE configs[4] shape: collision avoidance (largest factored domain, W = H = 7, 2 obstacles,
correct-graph prior, Pb = 3532 B), 10^6 particles per belief, importance-weighted update + resample.
Prints the achieved algorithmic HBM rate of the update (all launches of the multi-workgroup filter)."""
import json
import sys
import time

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import fba_pomdp_amd as fba
from fba_pomdp_amd import _native as N

Np = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
eng = fba.Engine("random-collision-avoidance", model=N.MODEL_BA_FACTORED, belief="importance_sampling", size=2,
                 width=7, height=7, particles=Np, sims=4, slots=slots, seed=5)
eng.belief_init()
eng.belief_reset_domain_state()
obs = 3 * 7 + 3
eng.belief_update(1, obs)           # warm-up
eng.reset_kernel_times()
t0 = time.perf_counter()
for k in range(reps):
    eng.set_position(t=(k + 1) % 200)
    eng.belief_update(1 + (k % 2), obs)
dt = time.perf_counter() - t0
kt = eng.kernel_times()["importance_kernel"]
gbs = kt.bytes / 1e9 / (kt.ms / 1e3)
print(json.dumps({"workload": f"collision-avoidance 7x7x2, {Np} particles x {slots} beliefs, importance update+resample",
                  "updates": reps, "ms_per_update": kt.ms / reps, "wall_ms_per_update": 1e3 * dt / reps,
                  "algorithmic_GB_per_update": kt.bytes / reps / 1e9, "achieved_GBs": gbs, "frac_of_8TBs": gbs / 8000.0,
                  "particles_per_s": kt.units / (kt.ms / 1e3)}))
