"""A slice of the randomised differential test (scripts/fuzz_parity.py: random domain x simulator x belief x
sizes x modes, engine against oracle on the same Philox streams, every trace field bit-equal)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def _fuzz():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "fuzz_parity.py")
    spec = importlib.util.spec_from_file_location("fuzz_parity", path)
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    return fuzz


def test_random_configurations_match_the_oracle():
    fuzz = _fuzz()
    bad, refused = fuzz.run(250, seed=7, verbose=False)
    assert bad == 0
    assert refused < 60          # (model / domain pairs neither side supports)


def test_random_mh_belief_configurations_match_the_oracle():
    """The MH beliefs (mh-within-gibbs with both state-history samplers, mh-nips) over random factored-tiger and
    collision-avoidance configurations (sizes, structure priors, thresholds, particle counts from 1): engine == oracle
    on every trace field."""
    import random

    from fba_pomdp_amd import _native as N
    fuzz = _fuzz()
    rng = random.Random(77)
    ran = 0
    for i in range(40):
        domain = rng.choice(["episodic-factored-tiger", "continuous-factored-tiger", "random-collision-avoidance", "centered-collision-avoidance", "gridworld",
                             "linear-sysadmin"])
        belief = rng.choice(["mh-within-gibbs", "mh-nips"]) if "sysadmin" not in domain else "mh-within-gibbs"
        kw = dict(particles=rng.choice([1, 5, 24, 50]), sims=rng.choice([4, 30, 90]), horizon=rng.choice([2, 5, 9]),
                  runs=rng.choice([1, 3]), episodes=rng.choice([1, 2, 4]), structure_prior=rng.choice([0, 1, 2, 3]),
                  threshold=rng.choice([-0.2, -1.0, -6.0]), noise=rng.choice([0.0, 0.1]), discount=rng.choice([0.7, 0.95]))
        if "tiger" in domain:
            kw["size"] = rng.choice([1, 2, 3])
        elif "sysadmin" in domain:
            kw["size"] = rng.choice([2, 3])
            kw["structure_prior"] = 0
            kw.pop("noise")
        elif domain == "gridworld":
            kw["size"] = rng.choice([3, 4])
            kw["particles"] = 64                  # (a filter without the true goal can never be updated)
            kw["structure_prior"] = rng.choice([0, 2])
            kw["horizon"] = min(kw["horizon"], 5)
            kw["episodes"] = min(kw["episodes"], 2)
            if belief == "mh-nips":
                kw["horizon"] = 2                 # (forward-sampled histories: see test_fbapomdp_mh_beliefs)
        else:
            kw["width"], kw["height"], kw["size"] = rng.choice([(3, 3, 1), (4, 3, 2), (3, 5, 1)])
            if kw["structure_prior"] == 3:
                kw["structure_prior"] = 2   # (no fully connected prior for the MH beliefs: refused)
        if belief == "mh-within-gibbs":
            kw["belief_option"] = rng.choice([0, 1])
            if domain == "gridworld" and kw["belief_option"] == 1:
                kw["horizon"] = 2
        slots = rng.choice([1, kw["runs"]])
        try:
            fuzz.one(domain, N.MODEL_BA_FACTORED, belief, slots, kw, seed=4000 + i)
            ran += 1
        except ValueError:
            pass
    assert ran >= 36
