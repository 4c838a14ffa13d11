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


def _collective_worker(rank, world, port, q, fail_rank):
    sys.path.insert(0, ROOT)
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))

    def probe(dist_, torch_, world_):     # stands in for the RCCL probe: it works on every rank but `fail_rank`
        if rank == fail_rank:
            raise RuntimeError("simulated RCCL failure on this rank only")
        return None, world_

    group, dev, how, seen = bench.open_collectives(dist, torch, rank, world, rank, False, probe=probe)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, group=group)        # the data collective goes wherever the ranks agreed it goes: it must complete
    # the out-of-memory step-down is per rank: the ranks then agree on the smallest count, and the line carries every rank's figures
    slots = bench.agree_on_slots(dist, torch, 1000 - 100 * rank)
    rows = bench.gather_per_rank(dist, torch, world, [10.0 * (rank + 1), 0.5 + rank, float(slots)])
    q.put((rank, dev, how, float(t[0]), seen, slots, rows))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_rank", [1, -1])
def test_ranks_agree_on_the_transport_before_the_data_collective(fail_rank):
    """bench.py's ranks decide RCCL-or-gloo together (an all-reduce(MIN) of every rank's probe result over the gloo group):
    when the probe fails on ONE rank, every rank reduces over gloo -- none is left waiting in an RCCL collective the other
    never enters (round 2's fallback was decided per rank and could hang); when it fails nowhere, every rank reports rccl."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 300) + (7 if fail_rank < 0 else 0)
    procs = [ctx.Process(target=_collective_worker, args=(r, 2, port, q, fail_rank)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [g[3] for g in got] == [3.0, 3.0]
    if fail_rank >= 0:
        assert all(g[1] == "cpu" and g[2].startswith("gloo (RCCL failed on at least one rank") and g[4] is None for g in got)
    else:
        assert all(g[2] == "rccl" and g[4] == 2 for g in got)     # returns.rccl_ranks: the count the probe's all-reduce returned
    assert all(g[5] == 900 for g in got)                           # rank 0 could hold 1000 slots, rank 1 900: both run 900
    assert all(g[6] == [[10.0, 0.5, 900.0], [20.0, 1.5, 900.0]] for g in got)


def test_the_launcher_ends_the_job_when_one_rank_dies_at_start_up(tmp_path):
    """`bench.py --gpus 8` without a launcher: rank 3 exits at once (as a rank whose fba_create fails would); its seven siblings sit in
    the gloo rendezvous, which waits 600 s for the missing rank.  spawn_ranks must notice the first non-zero exit, terminate the
    siblings and return that exit code -- within seconds, not after the rendezvous timeout (merge_result_files.py:60-78 pools
    per-process results; a job that lost a rank has nothing to pool)."""
    import time

    sys.path.insert(0, ROOT)
    import bench
    worker = tmp_path / "worker.py"
    worker.write_text(
        "import os, sys, datetime\n"
        "rank = int(os.environ['RANK'])\n"
        "if rank == 3:\n"
        "    sys.exit(7)\n"
        "import torch.distributed as dist\n"
        "dist.init_process_group('gloo', rank=rank, world_size=int(os.environ['WORLD_SIZE']), timeout=datetime.timedelta(seconds=600))\n"
        "dist.barrier()\n")
    t0 = time.monotonic()
    with open(tmp_path / "launcher.err", "w") as err:
        rc = bench.spawn_ranks(8, argv=[sys.executable, str(worker)], out=err)
    took = time.monotonic() - t0
    assert rc == 7
    assert took < 60, took
    assert "rank 3 exited with code 7" in (tmp_path / "launcher.err").read_text()


def test_the_launcher_returns_zero_when_every_rank_succeeds(tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    worker = tmp_path / "ok.py"
    worker.write_text("import os, sys\nsys.exit(0)\n")
    assert bench.spawn_ranks(4, argv=[sys.executable, str(worker)]) == 0


@pytest.mark.gpu
def test_bench_two_ranks_on_the_engine_equal_one_process():
    """`python bench.py --gpus 2` starts its own two worker processes (fresh processes, spawned before anything
    touches the GPU), each runs the HIP engine on its shard of the runs (run_offset = rank * slots) and the
    statistics are all-reduced: RCCL when every rank has its own GPU, gloo -- logged in the line -- when the box has
    one.  The two-rank totals must equal one process running the union of the runs: same simulated steps, same
    number of finished episodes, same return sum."""
    import json
    import subprocess

    import fba_pomdp_amd as fba

    slots, steps, warm = 1024, 3, 1
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", str(warm), "--slots", str(slots),
           "--sims", "256", "--particles", "256", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["slots_per_gpu"] == slots
    coll = line["returns"]["collective"]
    assert coll == "rccl" or coll.startswith("gloo")
    if coll != "rccl":
        assert "share" in coll or "RCCL failed" in coll         # the fallback names its reason
    # one process, the union of the two shards: runs 0 .. 2 * slots - 1
    eng = fba.Engine("episodic-tiger", model=fba.MODEL_BA_TABLE, belief="rejection_sampling", sims=256, particles=256, horizon=10,
                     episodes=64, runs=1 << 30, slots=2 * slots, run_offset=0, seed=20261003)
    eng.run_ticks(warm)
    c0 = eng.counters()
    eng.run_ticks(steps)
    c1 = eng.counters()
    one_steps = (c1.sim_steps - c0.sim_steps) + (c1.belief_steps - c0.belief_steps)
    n, s1, _ = eng.return_sums()
    assert round(line["value"] * line["ms_per_step"] * steps / 1e3) == one_steps
    assert line["returns"]["episodes"] == n
    assert abs(line["returns"]["mean"] - s1 / n) < 1e-9


@pytest.mark.gpu
def test_bench_two_ranks_scale_the_gridworld_workload():
    """`python bench.py --gpus 2 --workload c4` is the command a scaling run needs: BASELINE configs[3] (episode-sharded FBA-POMDP
    gridworld, history particles, importance sampling) on two ranks -- here with few slots, simulations and particles so that
    it finishes in seconds; the line names the workload and both ranks' steps are in it."""
    import json
    import subprocess

    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c4", "--steps", "2", "--warmup", "1", "--slots", "48",
           "--sims", "512", "--particles", "256", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["workload_key"] == "c4" and line["config"]["slots_per_gpu"] == 48
    assert "gridworld" in line["config"]["workload"] and line["roofline"]["kernel"] == "importance_kernel"
    assert line["value"] > 0 and line["search_kernel"]["steps_per_launch"] >= 48 * 512         # (rank 0's own launches: at least one step per simulation)
    assert line["returns"]["collective"] == "rccl" or line["returns"]["collective"].startswith("gloo")

