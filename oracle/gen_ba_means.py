"""Per-episode-index mean returns of the Bayes-adaptive configs from the oracle in mt19937 / reference-order mode -- the arithmetic and the
draw order of the reference binaries (`bapomdp` / `fbapomdp`), which report exactly this quantity: one Statistic per episode index over the runs
(src/experiments/BAPOMDPExperiment.cpp:20-30, 44-75).  The HIP engine draws from Philox streams in device order, so its tier of the parity
contract for these numbers is statistical (north_star: "mean episodic return within 1 sigma over 1e4 episodes"; SURVEY 8(c) tier 2):
tests/test_gpu_full_size.py::test_ba_per_episode_means_* compares the engine with the fixture this script writes,
tests/golden/oracle_ba_means.json.  A stream-addressing mistake that the oracle's Philox mode and the engine shared would pass every bit-exact
test; it cannot pass this one, because the mt19937 side has no streams.

  python oracle/gen_ba_means.py [workers]         (build container; minutes on 8 cores)

TEST INFRASTRUCTURE ONLY."""
import json
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# name -> (oracle keyword arguments, runs per seed string, seed strings).
# The test's tolerance is north_star's "1 sigma of a 1e4-episode mean"; that is only a meaningful bound when BOTH means are known much better
# than that, so the configs the bound is applied to run 4-8e4 runs here (and 1-2e5 on the engine).  The two full-size configs whose oracle
# runs are expensive keep 1e4 runs and are compared at 3 combined standard errors only.
CONFIGS = {
    # BASELINE configs[1] at its own size: bapomdp -D episodic-tiger -P po-uct -s 4096 --particle-amount 4096 -B rejection_sampling -C 10000 --noise 0
    "c2_full": (dict(domain="DOM_TIGER_EPISODIC", model=1, belief=0, sims=4096, particles=4096, horizon=10, episodes=5), 2500, 32),
    # the same workload with the importance filter (bench.py --belief importance_sampling): full size, 1e4 runs (every WeightedFilter::sample of the
    # reference order walks the filter: 0.25 s per run) ...
    "c2_importance": (dict(domain="DOM_TIGER_EPISODIC", model=1, belief=1, sims=4096, particles=4096, horizon=10, episodes=5), 1250, 8),
    # ... and at 1024 x 1024, 8e4 runs
    "c2_importance_1k": (dict(domain="DOM_TIGER_EPISODIC", model=1, belief=1, sims=1024, particles=1024, horizon=10, episodes=5), 2500, 32),
    # BASELINE configs[2] at its own size: fbapomdp -D episodic-factored-tiger --size 3 --structure-prior match-uniform -s 16384, 4096 particles: 1e4 runs ...
    "c3_full": (dict(domain="DOM_FTIGER_EPISODIC", model=2, belief=0, size=3, structure_prior=2, sims=16384, particles=4096, horizon=10, episodes=5), 625, 16),
    # ... and at 4096 simulations, 1024 particles: 8e4 runs
    "c3_reduced": (dict(domain="DOM_FTIGER_EPISODIC", model=2, belief=0, size=3, structure_prior=2, sims=4096, particles=1024, horizon=10, episodes=5), 2500, 32),
    # BASELINE configs[3]'s shape at reduced size (the oracle moves dense count tables: 191 KB per particle at --size 7):
    # fbapomdp -D gridworld --size 5 --structure-prior match-uniform -B importance_sampling, 1024 sims, 256 particles, horizon 20, 2 episodes: 4e4 runs
    "c4_size5_1k": (dict(domain="DOM_GRIDWORLD", model=2, belief=1, size=5, structure_prior=2, sims=1024, particles=256, horizon=20, episodes=2), 1250, 32),
    # the same at 2048 sims, 512 particles: 1e4 runs
    "c4_size5": (dict(domain="DOM_GRIDWORLD", model=2, belief=1, size=5, structure_prior=2, sims=2048, particles=512, horizon=20, episodes=2), 640, 16),
    # and at --size 3 with more simulations per particle: 8e4 runs
    "c4_size3": (dict(domain="DOM_GRIDWORLD", model=2, belief=1, size=3, structure_prior=2, sims=1024, particles=256, horizon=12, episodes=3), 2500, 32),
}


def one(job):
    name, seed = job
    from oracle import pyorc as orc
    kw, runs, _ = CONFIGS[name]
    kw = dict(kw)
    kw["domain"] = getattr(orc, kw["domain"])
    t0 = time.perf_counter()
    o = orc.Oracle(runs=runs, seed_str=seed, **kw)     # (rng_mode mt19937, reference-order arithmetic: the defaults)
    stats, res = o.run_bapomdp()
    return name, seed, [(s.count, s.mean, s.m2) for s in stats], res.sim_steps + res.belief_steps, time.perf_counter() - t0


def pooled(rows):
    """Chan's merge of (count, mean, M2) triples, the way analysis/preprocess/merge_result_files.py pools .res files."""
    n = sum(r[0] for r in rows)
    mean = sum(r[0] * r[1] for r in rows) / n
    m2 = sum(r[2] + r[0] * (r[1] - mean) ** 2 for r in rows)
    return n, mean, m2 / (n - 1)


if __name__ == "__main__":
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    only = sys.argv[2].split(",") if len(sys.argv) > 2 else list(CONFIGS)
    path = os.path.join(ROOT, "tests", "golden", "oracle_ba_means.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    jobs = [(name, f"ba-{name}-{k}") for name in only for k in range(CONFIGS[name][2])]
    t0 = time.perf_counter()
    with ProcessPoolExecutor(workers) as ex:
        rows = list(ex.map(one, jobs))
    for name in only:
        kw, runs, seeds = CONFIGS[name]
        mine = [r for r in rows if r[0] == name]
        eps = kw["episodes"]
        per_ep = [pooled([r[2][ep] for r in mine]) for ep in range(eps)]
        out[name] = {
            "oracle": dict(kw, rng="mt19937", arithmetic="reference order", runs_per_seed=runs, seed_strings=[r[1] for r in mine]),
            "count": [p[0] for p in per_ep], "mean": [p[1] for p in per_ep], "var": [p[2] for p in per_ep],
            "stder": [(p[2] / p[0]) ** 0.5 for p in per_ep], "stder_at_1e4": [(p[2] / 1e4) ** 0.5 for p in per_ep],
            "simulated_steps": sum(r[3] for r in mine), "cpu_seconds": sum(r[4] for r in mine),
        }
        print(name, [round(m, 4) for m in out[name]["mean"]], [round(s, 4) for s in out[name]["stder"]], f"{sum(r[4] for r in mine):.0f} cpu-s", flush=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(f"wrote {path} in {time.perf_counter() - t0:.0f} s")
