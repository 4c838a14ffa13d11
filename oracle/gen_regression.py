"""Regression fixtures of the ORACLE ITSELF (not reference vectors): short runs in mt19937 mode of the
paths whose orchestration cannot be pinned through oracle/_ref (their reference translation units need
Boost) -- the structure beliefs, the sysadmin / gridworld / collision-avoidance tabular priors.  They only
detect accidental changes of the restatement.  Writes tests/golden/oracle_regression.json."""
import hashlib
import numpy as np
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import pyorc as orc  # noqa: E402

CASES = {
    "reinvigoration_ftiger": dict(domain=orc.DOM_FTIGER_EPISODIC, size=3, model=orc.MODEL_BA_FACTORED, belief=orc.BELIEF_REINVIGORATION,
                                  resample_amount=6, structure_prior=orc.SP_MATCH_UNIFORM),
    "reinvigoration_collision_avoidance": dict(domain=orc.DOM_COLLISION_AVOID, width=4, height=3, size=2, model=orc.MODEL_BA_FACTORED,
                                               belief=orc.BELIEF_REINVIGORATION, resample_amount=4, structure_prior=orc.SP_UNIFORM),
    "reinvigoration_sysadmin": dict(domain=orc.DOM_SYSADMIN_LINEAR, size=3, model=orc.MODEL_BA_FACTORED, belief=orc.BELIEF_REINVIGORATION,
                                    resample_amount=5),
    "cheating_gridworld": dict(domain=orc.DOM_GRIDWORLD, size=3, model=orc.MODEL_BA_FACTORED, belief=orc.BELIEF_CHEATING,
                               resample_amount=3, threshold=-2.0, structure_prior=orc.SP_MATCH_UNIFORM),
    "mh_within_gibbs_ftiger_msg": dict(domain=orc.DOM_FTIGER_CONTINUOUS, size=2, model=orc.MODEL_BA_FACTORED, belief=orc.BELIEF_MH_GIBBS,
                                       threshold=-1.0, structure_prior=orc.SP_MATCH_UNIFORM),
    "mh_within_gibbs_ftiger_rs": dict(domain=orc.DOM_FTIGER_CONTINUOUS, size=2, model=orc.MODEL_BA_FACTORED, belief=orc.BELIEF_MH_GIBBS,
                                      threshold=-1.0, belief_option=1, structure_prior=orc.SP_UNIFORM),
    "mh_nips_ftiger": dict(domain=orc.DOM_FTIGER_CONTINUOUS, size=2, model=orc.MODEL_BA_FACTORED, belief=orc.BELIEF_MH_NIPS,
                           threshold=-1.0, structure_prior=orc.SP_MATCH_UNIFORM),
    "mh_within_gibbs_collision_avoidance": dict(domain=orc.DOM_COLLISION_AVOID, width=3, height=3, size=1, model=orc.MODEL_BA_FACTORED,
                                                belief=orc.BELIEF_MH_GIBBS, threshold=-2.0, structure_prior=orc.SP_MATCH_UNIFORM),
    "mh_nips_collision_avoidance": dict(domain=orc.DOM_COLLISION_AVOID, width=3, height=3, size=1, model=orc.MODEL_BA_FACTORED,
                                        belief=orc.BELIEF_MH_NIPS, threshold=-2.0, structure_prior=orc.SP_UNIFORM),
    "nested_bapomdp_tiger": dict(domain=orc.DOM_TIGER_EPISODIC, model=orc.MODEL_BA_TABLE, belief=orc.BELIEF_NESTED, particles=9),
    "nested_fbapomdp_collision_avoidance": dict(domain=orc.DOM_COLLISION_AVOID, width=3, height=3, size=1, model=orc.MODEL_BA_FACTORED,
                                                belief=orc.BELIEF_NESTED, particles=6, structure_prior=orc.SP_UNIFORM),
    "incubator_ftiger": dict(domain=orc.DOM_FTIGER_EPISODIC, size=2, model=orc.MODEL_BA_FACTORED, belief=orc.BELIEF_INCUBATOR,
                             resample_amount=6, threshold=0.5, structure_prior=orc.SP_MATCH_UNIFORM),
    "incubator_sysadmin": dict(domain=orc.DOM_SYSADMIN_LINEAR, size=3, model=orc.MODEL_BA_FACTORED, belief=orc.BELIEF_INCUBATOR,
                               resample_amount=5, threshold=0.5),
    "mh_within_gibbs_gridworld": dict(domain=orc.DOM_GRIDWORLD, size=3, model=orc.MODEL_BA_FACTORED, belief=orc.BELIEF_MH_GIBBS,
                                      threshold=-3.0, structure_prior=orc.SP_MATCH_UNIFORM, particles=64),
    "mh_within_gibbs_sysadmin": dict(domain=orc.DOM_SYSADMIN_LINEAR, size=3, model=orc.MODEL_BA_FACTORED, belief=orc.BELIEF_MH_GIBBS,
                                     threshold=-1.0, belief_option=1),
    "bapomdp_sysadmin": dict(domain=orc.DOM_SYSADMIN_INDEPENDENT, size=3, model=orc.MODEL_BA_TABLE),
    "bapomdp_gridworld": dict(domain=orc.DOM_GRIDWORLD, size=3, model=orc.MODEL_BA_TABLE, noise=0.1),
    "bapomdp_collision_avoidance": dict(domain=orc.DOM_COLLISION_AVOID, width=4, height=3, size=1, model=orc.MODEL_BA_TABLE, noise=0.1),
    "fbapomdp_collision_avoidance_match_uniform": dict(domain=orc.DOM_COLLISION_AVOID, width=4, height=3, size=2, model=orc.MODEL_BA_FACTORED,
                                                       belief=orc.BELIEF_IMPORTANCE, structure_prior=orc.SP_MATCH_UNIFORM),
}
COMMON = dict(particles=48, sims=64, horizon=6, max_depth=6, runs=3, episodes=3, rng_mode=orc.RNG_MT, arith=orc.ARITH_REF, trace=1)


def _legacy_layout(tr):
    """The hashes were taken when a trace record held 16 root entries; every case here has <= 16 actions, so
    cut the two root arrays back to 16 columns and the committed hashes keep their meaning."""
    fields = [(n, tr.dtype[n].base.str, (16,)) if tr.dtype[n].shape else (n, tr.dtype[n].str) for n in tr.dtype.names]
    old = np.zeros(len(tr), dtype=np.dtype(fields, align=tr.dtype.isalignedstruct))
    for n in tr.dtype.names:
        old[n] = tr[n][:, :16] if tr.dtype[n].shape else tr[n]
    return old


def run(name):
    o = orc.Oracle(seed_str="7", **{**COMMON, **CASES[name]})
    stats, res = o.run_bapomdp()
    tr = _legacy_layout(o.trace(res.n_trace))
    return {"means": [s.mean for s in stats], "sim_steps": res.sim_steps, "belief_steps": res.belief_steps,
            "trace_sha256": hashlib.sha256(tr.tobytes()).hexdigest(), "records": int(res.n_trace)}


if __name__ == "__main__":
    out = {k: run(k) for k in CASES}
    path = os.path.join(ROOT, "tests", "golden", "oracle_regression.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)
