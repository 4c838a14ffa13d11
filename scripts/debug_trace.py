import sys
import numpy as np
sys.path.insert(0, ".")
import fba_pomdp_amd as fba
from fba_pomdp_amd import _native as N
from oracle import pyorc as orc

kw = dict(particles=64, sims=200, runs=24)
eng = fba.Engine("episodic-tiger", model=N.MODEL_POMDP, belief="rejection_sampling", seed=1, slots=24, trace=1, **kw)
o = orc.Oracle(domain=orc.DOM_TIGER_EPISODIC, model=0, belief=0, rng_mode=orc.RNG_PHILOX, arith=orc.ARITH_DEV, philox_seed=1, trace=1, **kw)
eng.run_planning()
st, res = o.run_planning()
tr, otr = eng.trace(), o.trace(res.n_trace)
print(len(tr), len(otr))
names = ["run", "episode", "t", "action", "state", "obs", "terminal", "n_nodes", "tree_depth", "update_count", "reward", "belief_hash"]
for i in range(min(len(tr), len(otr), 16)):
    print("G", [tr[i][n] for n in names], tr[i]["root_n"][:3], tr[i]["root_q"][:3])
    print("O", [otr[i][n] for n in names], otr[i]["root_n"][:3], otr[i]["root_q"][:3])
