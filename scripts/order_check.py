"""Device-order sums against reference-order sums ("device-order sums"), oracle against oracle on
the CPU: the same experiment, the same Philox draws, once with the importance filter's three sums and its
WeightedFilter::sample done as the reference does them (sequential sums, backward subtraction scan: ORC_ARITH_REF,
the arithmetic the golden vectors pin) and once in the engine's order (4 per lane + 64-lane Kogge-Stone + chained
chunks, search on prefix sums: ORC_ARITH_DEV, what the HIP engine matches bit for bit).  They can only part where a
threshold u * total lies within an ulp of a prefix sum.  Counts, over seeds x domains x filter sizes, the belief
updates after which the two filters are not the same particle set (position-keyed checksum over every particle's state,
weight and counts), and prints one JSON line per cell plus a total.
    python scripts/order_check.py [seeds] [processes]"""
import json
import os
import sys
from multiprocessing import Pool

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CELLS = [
    ("tiger", dict(domain=1, model=1), [500, 4096, 65536]),                                        # continuous tiger, tabular
    ("gridworld-5", dict(domain=4, model=2, size=5, structure_prior=2), [500, 4096]),              # factored, match-uniform
    ("gridworld-3", dict(domain=4, model=2, size=3, structure_prior=2), [65536]),                  # (a 65 536-particle size-5 filter is 5 GB)
    ("collision-avoidance 7x7x2", dict(domain=5, model=2, width=7, height=7, size=2), [500, 4096, 65536]),
]


def one(job):
    from oracle import pyorc as orc
    import numpy as np
    name, kw, n, seed, horizon = job
    common = dict(belief=orc.BELIEF_IMPORTANCE, rng_mode=orc.RNG_PHILOX, philox_seed=1000 + seed, particles=n, sims=8,
                  runs=1, episodes=1, horizon=horizon, trace=1, **kw)
    a = orc.Oracle(arith=orc.ARITH_DEV, **common)
    b = orc.Oracle(arith=orc.ARITH_REF, **common)
    _, ra = a.run_bapomdp()
    _, rb = b.run_bapomdp()
    ta, tb = a.trace(ra.n_trace), b.trace(rb.n_trace)
    m = min(len(ta), len(tb))
    same = (ta["belief_hash"][:m] == tb["belief_hash"][:m]) & (ta["action"][:m] == tb["action"][:m]) & (ta["obs"][:m] == tb["obs"][:m])
    first_bad = int(np.argmin(same)) if not same.all() else -1
    compared = m if first_bad < 0 else first_bad + 1   # updates are only comparable until the two experiments part
    return name, n, compared, 0 if first_bad < 0 else 1


if __name__ == "__main__":
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    jobs = []
    for name, kw, sizes in CELLS:
        for n in sizes:
            k = seeds if n <= 4096 else max(seeds // 5, 4)   # the reference-order scan is O(N^2) per update
            jobs += [(name, kw, n, s, 6 if n <= 4096 else 3) for s in range(k)]
    cells = {}
    with Pool(procs) as pool:
        for name, n, compared, bad in pool.imap_unordered(one, jobs, chunksize=1):
            c = cells.setdefault((name, n), [0, 0, 0])
            c[0] += 1; c[1] += compared; c[2] += bad
    tot = [0, 0, 0]
    for (name, n), c in sorted(cells.items()):
        print(json.dumps({"domain": name, "particles": n, "experiments": c[0], "updates_compared": c[1], "updates_that_differ": c[2],
                          "draws_compared": c[1] * n}), flush=True)
        tot = [tot[0] + c[0], tot[1] + c[1], tot[2] + c[2]]
    print(json.dumps({"total_experiments": tot[0], "total_updates_compared": tot[1], "total_updates_that_differ": tot[2]}))
