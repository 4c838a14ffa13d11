"""Per tick of the bench workload: search ms, simulated steps, and how uneven the lanes of a search wave are --
a wave lasts as long as its slowest lane, so 64 x (max steps of a wave's lanes) summed over waves against the steps
actually made is the share of lane-iterations a lock-step tick leaves idle.  python scripts/tick_probe.py [slots] [ticks]"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fba_pomdp_amd as fba

slots = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
eng = fba.Engine("episodic-tiger", model=fba.MODEL_BA_TABLE, belief="rejection_sampling", sims=4096, particles=4096, horizon=10,
                 episodes=64, runs=1 << 30, slots=slots, seed=20261003)
eng.L.fba_debug_slot_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
prev = np.zeros(slots, np.uint64)
cur = np.zeros(slots, np.uint64)
t = np.zeros(slots, np.int32)
for k in range(ticks):
    eng.L.fba_debug_slot_counters(eng.h, None, t.ctypes.data)   # the time-step each slot searches at in this tick
    eng.reset_kernel_times()
    eng.run_ticks(1)
    kt = eng.kernel_times()
    eng.L.fba_debug_slot_counters(eng.h, cur.ctypes.data, None)
    d = (cur - prev).astype(np.int64)
    prev[:] = cur
    w = d.reshape(-1, 64)
    by_t = {int(v): float(d[t == v].mean()) for v in np.unique(t)}
    print(json.dumps({"tick": k, "search_ms": kt["search_kernel"].ms, "belief_ms": kt["reject_kernel"].ms, "steps": int(d.sum()),
                      "mean_steps_per_slot": float(d.mean()), "max_steps_of_a_slot": int(d.max()),
                      "lane_iterations_if_lockstep": int(64 * w.max(axis=1).sum()), "busy_fraction": float(d.sum() / (64.0 * w.max(axis=1).sum())),
                      "mean_of_wave_max": float(w.max(axis=1).mean()), "slots_by_t": {int(v): int((t == v).sum()) for v in np.unique(t)},
                      "mean_steps_by_t": by_t}), flush=True)
