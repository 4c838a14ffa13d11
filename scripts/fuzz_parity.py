"""Randomised differential test: engine (HIP) against the oracle (Philox, device arithmetic) over random
valid configurations -- domains x simulators x beliefs x sizes x modes.  python scripts/fuzz_parity.py [n] [seed] [belief-name filter]"""
import os, sys, random, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fba_pomdp_amd as fba
from fba_pomdp_amd import _native as N
from oracle import pyorc as orc

DOM = {"random-collision-avoidance": orc.DOM_COLLISION_AVOID, "centered-collision-avoidance": orc.DOM_COLLISION_AVOID,
       "gridworld": orc.DOM_GRIDWORLD, "episodic-tiger": orc.DOM_TIGER_EPISODIC, "continuous-tiger": orc.DOM_TIGER_CONTINUOUS,
       "episodic-factored-tiger": orc.DOM_FTIGER_EPISODIC, "continuous-factored-tiger": orc.DOM_FTIGER_CONTINUOUS,
       "independent-sysadmin": orc.DOM_SYSADMIN_INDEPENDENT, "linear-sysadmin": orc.DOM_SYSADMIN_LINEAR,
       "coffee": orc.DOM_COFFEE, "boutilier-coffee": orc.DOM_COFFEE_BOUTILIER, "agr": orc.DOM_AGR}


def draw(rng):
    domain = rng.choice(list(DOM))
    model = rng.choice([N.MODEL_POMDP, N.MODEL_BA_TABLE, N.MODEL_BA_FACTORED])
    if "coffee" in domain or domain == "agr":
        model = N.MODEL_POMDP   # planning only
    kw = dict(particles=rng.choice([1, 7, 33, 64, 130]), sims=rng.choice([1, 5, 40, 96]), horizon=rng.choice([1, 3, 7, 12]),
              runs=rng.choice([1, 3, 6]), discount=rng.choice([0.5, 0.95, 1.0]), exploration=rng.choice([0.0, 1.0, 100.0]))
    kw["max_depth"] = rng.choice([-1, 0, 1, 4, kw["horizon"]])
    slots = rng.choice([1, 2, kw["runs"]])
    if "tiger" in domain and "factored" not in domain or "coffee" in domain:
        pass
    elif domain == "agr":
        kw["particles"] = rng.choice([256, 400])   # 21 goals: a filter without the true one can never be updated
    elif "factored-tiger" in domain:
        kw["size"] = rng.choice([1, 2, 3])
    elif domain == "gridworld":
        kw["size"] = rng.choice([3, 3, 4, 4, 5])
        kw["particles"] = rng.choice([64, 130])   # a filter without the true goal can never be updated
    elif "sysadmin" in domain:
        kw["size"] = rng.choice([1, 2, 3, 4])
    else:
        kw["width"], kw["height"], kw["size"] = rng.choice([(3, 3, 1), (4, 3, 2), (3, 5, 1)])
    belief = rng.choice(["rejection_sampling", "importance_sampling"]) if domain != "agr" else "rejection_sampling"
    if rng.random() < 0.08 and ("tiger" in domain or "sysadmin" in domain or "coffee" in domain):
        # (elsewhere a single wrong state may never reproduce the observation: the reference spins, the engine faults)
        belief = "point_estimate"
    if model != N.MODEL_POMDP:
        kw["episodes"] = rng.choice([1, 2, 3])
        kw["counts_total"] = rng.choice([10.0, 777.0, 10000.0])
        if "tiger" in domain:
            kw["noise"] = rng.choice([0.0, 0.1, -0.1])
        elif domain == "gridworld" or "collision" in domain:
            kw["noise"] = rng.choice([0.0, 0.1])
    if model == N.MODEL_BA_FACTORED:
        if "factored-tiger" in domain:
            kw["structure_prior"] = rng.choice([0, 1, 2, 3])
        elif domain == "gridworld":
            kw["structure_prior"] = rng.choice([0, 2])
        elif "collision" in domain:
            kw["structure_prior"] = rng.choice([0, 1, 2, 3])
        b = rng.random()
        if b < 0.2 and domain != "gridworld" and not ("collision" in domain and kw.get("structure_prior") == 3):
            belief = "reinvigoration"
            kw["resample_amount"] = rng.choice([1, 4, 20])
        elif b < 0.28 and domain != "gridworld" and not ("collision" in domain and kw.get("structure_prior") == 3) and kw["particles"] > 1:
            belief = "incubator"
            kw["resample_amount"] = rng.choice([1, 4, 20])
            kw["threshold"] = rng.choice([0.05, 0.5, 1.0])
        elif b < 0.4:
            belief = "cheating-reinvigoration"
            kw["resample_amount"] = rng.choice([1, 5])
            kw["threshold"] = rng.choice([-0.3, -2.0, -50.0])
        elif b < 0.6 and ("factored-tiger" in domain or "collision" in domain or domain == "gridworld" or "sysadmin" in domain) and kw["particles"] <= 64:
            belief = rng.choice(["mh-within-gibbs", "mh-nips"]) if "sysadmin" not in domain else "mh-within-gibbs"
            kw["threshold"] = rng.choice([-0.3, -2.0, -50.0])
            if belief == "mh-within-gibbs":
                kw["belief_option"] = rng.choice([0, 1])
            if domain == "gridworld" and (belief == "mh-nips" or kw.get("belief_option") == 1):
                kw["horizon"] = min(kw["horizon"], 2)   # forward-sampled histories must reproduce every observation of an episode
                kw["max_depth"] = min(kw["max_depth"], 2)
    if model != N.MODEL_POMDP and belief in ("rejection_sampling", "importance_sampling") and rng.random() < 0.12:
        belief = "nested"            # NestedBelief: `particles` count particles, particles^2 domain states each
        kw["particles"] = rng.choice([1, 3, 8, 14])
        if domain == "gridworld":
            kw["particles"] = rng.choice([8, 14])
    if model != N.MODEL_POMDP and rng.random() < 0.2:
        longest = {"gridworld": 99, "random-collision-avoidance": kw.get("height", 0), "centered-collision-avoidance": kw.get("height", 0)}.get(domain, 2)
        if model == N.MODEL_BA_FACTORED and longest <= 16 or model == N.MODEL_BA_TABLE and "tiger" in domain and "factored" not in domain:
            kw["dirichlet_regular"] = 1
    if rng.random() < 0.15:
        kw["planner"] = rng.choice(["random", "ts"])
    if domain == "gridworld" and model == N.MODEL_BA_FACTORED and belief == "importance_sampling" and rng.random() < 0.6:
        kw["search_budget"] = rng.choice([1, 9, 60, 400])   # engine only: budgeted launches must give the results of whole searches
    if domain == "gridworld" and model == N.MODEL_BA_FACTORED and belief == "importance_sampling" and rng.random() < 0.5:
        kw["tree_buckets"] = max(8, kw["sims"] + rng.choice([0, 2, 40]))   # engine only: a bucket table that a dense little tree fills to the brim (lookups walk past their home line)
    return domain, model, belief, slots, kw


def one(domain, model, belief, slots, kw, seed):
    kw = dict(kw)
    planner = kw.pop("planner", "po-uct")
    eng = fba.Engine(domain, model=model, belief=belief, planner=planner, seed=seed, slots=slots, trace=1, **kw)
    okw = dict(kw)
    okw.pop("search_budget", None)     # (a schedule of the engine, not a parameter of the algorithm)
    okw.pop("tree_buckets", None)      # (the size of the engine's tree table, likewise)
    if domain == "centered-collision-avoidance":
        okw["ca_centered"] = 1
    o = orc.Oracle(domain=DOM[domain], model=model, belief=N.BELIEF_NAMES[belief], planner=N.PLANNER_NAMES[planner],
                   rng_mode=orc.RNG_PHILOX, arith=orc.ARITH_DEV, philox_seed=seed, trace=1, **okw)
    ba = model != N.MODEL_POMDP
    try:
        stats = eng.run_bapomdp() if ba else [eng.run_planning()]
    except fba.FbaError as e:
        if "accepted fewer than" in str(e) or "cannot reproduce the run's history" in str(e):   # the reference (and the oracle) would never return from this update
            print("   degenerate filter, skipped:", str(e)[:70], flush=True)
            eng.close()
            return
        raise
    ostats, res = o.run_bapomdp() if ba else (lambda r: ([r[0]], r[1]))(o.run_planning())
    tr, otr = eng.trace(), o.trace(res.n_trace)
    assert len(tr) == len(otr), (len(tr), len(otr))
    for name in tr.dtype.names:
        same = np.all((tr[name] == otr[name]).reshape(len(tr), -1), axis=1)
        assert same.all(), f"{name}: first mismatch at record {np.nonzero(~same)[0][0]}"
    for a, b in zip(stats, ostats):
        assert (a.count, a.mean, a.m2) == (b.count, b.mean, b.m2)
    c = eng.counters()
    assert (c.sim_steps, c.belief_steps, c.env_steps) == (res.sim_steps, res.belief_steps, res.env_steps)
    eng.close()


def run(n, seed, verbose=True, only=""):
    """(mismatches, refused) over n random configurations (only: those whose belief's name contains it)"""
    rng = random.Random(seed)
    bad = skipped = 0
    for i in range(n):
        cfg = draw(rng)
        while only not in cfg[2]:
            cfg = draw(rng)
        if verbose:
            print(i, cfg[0], cfg[1], cfg[2], cfg[3], cfg[4], flush=True)
        try:
            one(*cfg, seed=1000 + i)
        except ValueError as e:          # both sides must refuse the same configurations
            skipped += 1
            if verbose and skipped <= 8:
                print("refused:", cfg[0], cfg[1], cfg[2], str(e)[:90], flush=True)
        except Exception as e:
            bad += 1
            print("MISMATCH", i, cfg, "\n   ", str(e)[:300], flush=True)
            if not isinstance(e, AssertionError):
                traceback.print_exc()
    return bad, skipped


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    bad, skipped = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 1, only=sys.argv[3] if len(sys.argv) > 3 else "")
    print(f"{n} configurations: {bad} mismatches, {skipped} refused", flush=True)
    sys.exit(1 if bad else 0)
