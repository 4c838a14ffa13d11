"""Probe: do two half-size engines on two HIP streams, started half a tick apart, overlap the
VALU-bound search of one with the HBM-bound belief update of the other?  Aggregate steps/s vs one
full-size engine."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fba_pomdp_amd as fba

def make(slots, off):
    return fba.Engine("episodic-tiger", model=fba.MODEL_BA_TABLE, belief="rejection_sampling", sims=4096, particles=4096,
                      horizon=10, episodes=64, runs=1 << 30, slots=slots, run_offset=off, seed=20261003)

def steps(e, c0):
    c = e.counters()
    return (c.sim_steps - c0.sim_steps) + (c.belief_steps - c0.belief_steps)

K = 8
for groups, delay in ((1, 0.0), (2, 0.0), (2, 0.045), (4, 0.03)):
    engs = [make(131072 // groups, g * (131072 // groups)) for g in range(groups)]
    for e in engs:
        e.run_ticks(2)
    c0 = [e.counters() for e in engs]
    def work(e, d):
        time.sleep(d)
        e.run_ticks(K)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(e, g * delay)) for g, e in enumerate(engs)]
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    tot = sum(steps(e, c) for e, c in zip(engs, c0))
    print(f"groups {groups} stagger {delay*1e3:.0f} ms: {tot/dt:.4g} steps/s, {1e3*dt/K:.1f} ms per tick-round", flush=True)
    for e in engs: e.close()
