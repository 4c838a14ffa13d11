"""The N > 1 path on CPU (gloo, world_size 2): episode sharding by run_offset and the one
collective of the job, an all-reduce of {episodes, sum, sum of squares} (bench.py / DESIGN.md section 6).
The per-rank work is done by the oracle here (no GPU in this container); what is under test is the
partition (disjoint Philox run coordinates reproduce the single-process experiment exactly) and the merge."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KW = dict(domain=0, model=1, belief=0, rng_mode=1, arith=1, philox_seed=99, particles=64, sims=64, episodes=3, horizon=6)
RUNS = 10


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from oracle import pyorc as orc
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = RUNS // world
    o = orc.Oracle(runs=per, run_offset=rank * per, trace=1, **KW)
    stats, res = o.run_bapomdp()
    tr = o.trace(res.n_trace)
    rets = {}
    for r in tr:
        key = (int(r["run"]), int(r["episode"]))
        rets[key] = rets.get(key, 0.0) + float(r["reward"]) * 0.95 ** int(r["t"])
    v = np.array(list(rets.values()))
    sums = torch.tensor([len(v), v.sum(), (v * v).sum(), float(res.sim_steps + res.belief_steps)], dtype=torch.float64)
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    if rank == 0:
        q.put((sums.tolist(), sorted(rets.items())))
    else:
        q.put((None, sorted(rets.items())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_reproduce_single_process_experiment():
    from oracle import pyorc as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sums = next(g[0] for g in got if g[0] is not None)
    per_run = dict(kv for g in got for kv in g[1])

    o = orc.Oracle(runs=RUNS, trace=1, **KW)
    stats, res = o.run_bapomdp()
    tr = o.trace(res.n_trace)
    ref = {}
    for r in tr:
        key = (int(r["run"]), int(r["episode"]))
        ref[key] = ref.get(key, 0.0) + float(r["reward"]) * 0.95 ** int(r["t"])
    assert per_run == ref                      # sharded runs are the same runs, bit for bit
    assert sums[0] == RUNS * KW["episodes"]
    assert sums[3] == res.sim_steps + res.belief_steps
    pooled_mean = sums[1] / sums[0]
    assert abs(pooled_mean - np.mean(list(ref.values()))) < 1e-12
    # pooled variance from the all-reduced sums == variance over all episodes
    pooled_var = (sums[2] - sums[1] ** 2 / sums[0]) / (sums[0] - 1)
    assert abs(pooled_var - np.var(list(ref.values()), ddof=1)) < 1e-9
