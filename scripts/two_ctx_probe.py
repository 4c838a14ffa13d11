"""Do two engine contexts of half the slots each, driven from two host threads on their own HIP streams, finish the bench workload's
ticks sooner than one context of all the slots (the search of one overlapping the belief update of the other)?
python scripts/two_ctx_probe.py [groups] [ticks]  -> one JSON line"""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fba_pomdp_amd as fba
from fba_pomdp_amd import _native as N

groups = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
total = 262144
kw = dict(model=N.MODEL_BA_TABLE, belief="rejection_sampling", sims=4096, particles=4096, horizon=10, episodes=64)
engs = [fba.Engine("episodic-tiger", runs=1 << 30, slots=total // groups, run_offset=g * (total // groups), seed=20261003, **kw) for g in range(groups)]
stagger = os.environ.get("STAGGER", "1") == "1"


def steps_of(e):
    c = e.counters()
    return c.sim_steps + c.belief_steps


for i, e in enumerate(engs):
    e.run_ticks(2)
before = sum(steps_of(e) for e in engs)
t0 = time.perf_counter()
def work(i, e):
    if stagger and i:
        e.run_ticks(0) if False else None
        time.sleep(0.05 * i)      # half a tick behind the previous group: its search meets the other's belief update
    e.run_ticks(ticks)


ths = [threading.Thread(target=work, args=(i, e)) for i, e in enumerate(engs)]
for t in ths:
    t.start()
for t in ths:
    t.join()
dt = time.perf_counter() - t0
after = sum(steps_of(e) for e in engs)
print(json.dumps({"groups": groups, "ticks": ticks, "ms_per_tick": 1e3 * dt / ticks, "steps_per_s": (after - before) / dt}), flush=True)
