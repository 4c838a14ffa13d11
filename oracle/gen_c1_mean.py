"""Tighter estimate of the C1 mean episodic return: the oracle in mt19937 mode (bit-identical to the
reference binary on `--seed 1`, see tests/test_oracle_golden.py) over 16 seed strings x 10^4 runs.
Writes tests/golden/oracle_c1_mean.json.  TEST INFRASTRUCTURE ONLY."""
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(seed):
    from oracle import pyorc as orc
    o = orc.Oracle(domain=orc.DOM_TIGER_EPISODIC, sims=1024, particles=256, runs=10000, seed_str=seed)
    st, _ = o.run_planning()
    return seed, st.count, st.mean, st.m2


if __name__ == "__main__":
    seeds = [str(i) for i in range(1, 17)]
    with ProcessPoolExecutor(6) as ex:
        rows = list(ex.map(one, seeds))
    n = sum(r[1] for r in rows)
    mean = sum(r[1] * r[2] for r in rows) / n
    m2 = sum(r[3] + r[1] * (r[2] - mean) ** 2 for r in rows)
    var = m2 / (n - 1)
    out = {"config": "planning -D episodic-tiger -P po-uct -s 1024 --particle-amount 256 --runs 10000",
           "seeds": seeds, "per_seed_mean": [r[2] for r in rows], "count": n, "mean": mean, "var": var,
           "stder": (var / n) ** 0.5, "stder_at_1e4": (var / 1e4) ** 0.5}
    with open(os.path.join(ROOT, "tests", "golden", "oracle_c1_mean.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))
