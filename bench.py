#!/usr/bin/env python3
"""bench.py -- simulated env steps / sec (belief + rollout) of the BA-POMCP hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--workload c1|c2|c3|c4|c5]   (N > 1: starts its own N worker processes, one per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one real time-step of every resident slot: POUCT / RBAPOUCT search (`sims` simulations),
true-environment step, and the particle-filter belief update.  The default workload (N = 1 and per GPU
for N > 1) is BASELINE.json configs[1] (`c2`): episodic-tiger BA-POMCP, 4096 sims/step, 4096 particles,
tabular BA-POMDP, expected-Dirichlet sampling, rejection-sampling belief (the reference default);
`--workload` selects another BASELINE config at its own size (WORKLOADS below) -- `c4` is the
episode-sharded FBA-POMDP gridworld north_star scales over GPUs.
Runs are independent, so N GPUs run N disjoint sets of runs (weak scaling); the only collective is
one all-reduce of {episodes, sum of returns, sum of squares} + the step counters at the end.

`value` counts simulator.step calls of the planner (tree + rollout) plus those of the belief
update, exactly as BASELINE.md defines the metric; inputs (priors, particles, trees) are resident
in HBM before the timed region.
"""
import argparse
import datetime
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

# BASELINE.json `configs`, each at its own size.  slots = concurrent runs per GPU (what fits / fills one MI355X);
# cpu_* = the bounded sample the CPU baseline runs (its time per simulated step depends on neither sims nor particles).
# model: 0 plain POMDP (planning), 1 tabular BA-POMDP (bapomdp), 2 factored (fbapomdp); structure_prior 2 = match-uniform.
WORKLOADS = {
    "c1": dict(name="configs[0]: planning -D episodic-tiger, POMCP, 1024 sims/step, 256 belief particles",
               domain="episodic-tiger", model=0, belief="rejection_sampling", sims=1024, particles=256, horizon=10, episodes=1,
               slots=262144, cpu=dict(runs=20000, episodes=1)),
    "c2": dict(name="configs[1]: episodic-tiger BA-POMCP (tabular BA-POMDP, expected Dirichlet), 4096 sims/step, 4096 particles",
               domain="episodic-tiger", model=1, belief="rejection_sampling", sims=4096, particles=4096, horizon=10, episodes=64,
               slots=None, cpu=dict(runs=4, episodes=500)),
    "c3": dict(name="configs[2]: episodic-factored-tiger FBA-POMDP (--size 3), factored Dirichlet prior (match-uniform), 16384 sims/step, 4096 particles",
               domain="episodic-factored-tiger", model=2, belief="rejection_sampling", size=3, structure_prior=2, sims=16384, particles=4096,
               horizon=10, episodes=64, slots=163840, cpu=dict(runs=4, episodes=60, sims=2048, particles=1024)),   # (ten search waves per CU: what LDS holds)
    "c4": dict(name="configs[3]: gridworld (--size 7) FBA-POMDP, 65536 sims/step, 16384 particles, importance sampling, episode-sharded",
               domain="gridworld", model=2, belief="importance_sampling", size=7, structure_prior=2, sims=65536, particles=16384,
               horizon=20, episodes=2, slots=49152, search_budget=16384, tree_buckets=32768, cpu=dict(runs=8, episodes=1, sims=2048, particles=1024)),
               # (49 152 slots = three search waves of 16 trees on each of the 1 024 SIMDs; 32 768 buckets per tree: 17 800 in use; 5.5 MB per slot, 270 GB)
    "c5": dict(name="configs[4]: collision avoidance 7x7, 2 obstacles (largest factored domain), 10^6 particles per belief, "
                    "importance-weighted update + resample (4 beliefs in flight: the search is 4 lanes of one wave and measures nothing -- "
                    "this workload times the filter)",
               domain="random-collision-avoidance", model=2, belief="importance_sampling", size=2, width=7, height=7, sims=16,
               particles=1_000_000, horizon=20, episodes=4, slots=4, cpu=dict(runs=2, episodes=1, sims=16, particles=20000)),
}
ORC_DOMAIN = {"episodic-tiger": "DOM_TIGER_EPISODIC", "episodic-factored-tiger": "DOM_FTIGER_EPISODIC", "gridworld": "DOM_GRIDWORLD",
              "random-collision-avoidance": "DOM_COLLISION_AVOID"}
ENGINE_KEYS = ("model", "belief", "size", "width", "height", "structure_prior", "sims", "particles", "horizon", "episodes", "search_budget", "tree_buckets")


def workload_of(args):
    """The workload dictionary with the command line's overrides (--sims / --particles / --horizon / --belief / --slots)."""
    w = dict(WORKLOADS[args.workload])
    for k in ("sims", "particles", "horizon", "belief", "slots", "search_budget", "tree_buckets"):
        if getattr(args, k) is not None:
            w[k] = getattr(args, k)
    return w


def cpu_worker(args):
    """One process of the CPU baseline (`bench.py --cpu-worker SEED`): the oracle in mt19937 mode (the
    reference's own arithmetic and draw order) on its own seed; prints one JSON line."""
    from oracle import pyorc as orc
    from fba_pomdp_amd import _native as N
    w = workload_of(args)
    cpu = w["cpu"]
    kw = dict(domain=getattr(orc, ORC_DOMAIN[w["domain"]]), model=w["model"], belief=N.BELIEF_NAMES[w["belief"]],
              size=w.get("size", 0), width=w.get("width", 0), height=w.get("height", 0), structure_prior=w.get("structure_prior", 0),
              sims=min(w["sims"], cpu.get("sims", w["sims"])), particles=min(w["particles"], cpu.get("particles", w["particles"])),
              horizon=w["horizon"], runs=args.cpu_runs or cpu["runs"], episodes=args.cpu_episodes or cpu["episodes"], seed_str=args.cpu_worker)
    o = orc.Oracle(**kw)
    t0 = time.perf_counter()
    _, res = o.run_planning() if w["model"] == 0 else o.run_bapomdp()
    print(json.dumps({"steps": res.sim_steps + res.belief_steps, "env_steps": res.env_steps, "seconds": time.perf_counter() - t0,
                      "sims": kw["sims"], "particles": kw["particles"], "runs": kw["runs"], "episodes": kw["episodes"]}), flush=True)


def cpu_baseline(args):
    """The oracle (CPU restatement of the reference path) timed on the host cores of this box on a
    bounded sample of the same workload: one PROCESS per available core (the reference algorithm is
    built on global RNG / static scratch state, SURVEY section 5, so no threads), disjoint seeds;
    aggregate = total simulated steps / longest process time.  Plain child processes with a hard
    timeout: they never touch the GPU."""
    import subprocess
    cores = max(1, min(len(os.sched_getaffinity(0)), args.cpu_cores or 10 ** 6))
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", args.workload]
    for k in ("sims", "particles", "horizon", "belief", "cpu_runs", "cpu_episodes"):
        if getattr(args, k) is not None:
            cmd += ["--" + k.replace("_", "-"), str(getattr(args, k))]
    t0 = time.perf_counter()
    procs = [subprocess.Popen(cmd + ["--cpu-worker", f"bench-{k}"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
             for k in range(cores)]
    out = []
    for p in procs:
        try:
            stdout, _ = p.communicate(timeout=max(5.0, 120.0 - (time.perf_counter() - t0)))
            out.append(json.loads(stdout.strip().splitlines()[-1]))
        except Exception:
            p.kill()
    wall = time.perf_counter() - t0
    if not out:
        return {"value": None, "unit": "simulated env steps/s", "cores": cores, "kind": "port", "sample": "CPU baseline workers failed"}
    steps = sum(o["steps"] for o in out)
    longest = max(o["seconds"] for o in out)
    per_core = sorted(o["steps"] / o["seconds"] for o in out)[len(out) // 2]
    o0 = out[0]
    return {
        "value": steps / longest, "unit": "simulated env steps/s", "cores": len(out), "kind": "port",
        "per_core": per_core,
        "sample": f"{len(out)} processes x {o0['runs']} runs x {o0['episodes']} episodes of the same domain / model / belief / horizon at "
                  f"{o0['sims']} sims and {o0['particles']} particles ({sum(o['env_steps'] for o in out)} real steps, {steps} simulated steps, "
                  f"longest process {longest:.1f} s, {wall:.1f} s wall, gcc -O2)",
    }


def committed_profile(name):
    """A counter summary committed under profiles/ (newest round first), or None."""
    for tag in ("r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", f"{tag}_{name}")
        if os.path.exists(path):
            with open(path) as f:
                return json.load(f), os.path.relpath(path, ROOT)
    return None, None


def measured_traffic(args, w, kname, slots):
    """Fabric bytes of a FULL launch (every slot updates) of the roofline kernel from the committed rocprofv3 PMC passes of this same command
    (FETCH_SIZE and WRITE_SIZE in separate runs; profiles/*.json says how they were collected).  FETCH_SIZE tallies a read request at 64 B
    while the memory side moves 128-byte lines (profiles/r04_randline_counters.json), so fetched bytes are the counter x 2, except where a
    micro-kernel of known traffic measured another factor for the access shape (the sequential parking sweep of reject_tiger_lds_kernel).
    None when the workload differs from a profiled one."""
    if args.workload == "c3" and kname == "reject_kernel" and slots == 163840 and w["sims"] == 16384 and w["particles"] == 4096:
        d, path = committed_profile("pmc_c3.json")
        return (d["full_launch"]["traffic_bytes"], path) if d else (None, None)
    if args.workload != "c2" or slots != 262144 or w["sims"] != 4096 or w["particles"] != 4096:
        return None, None
    d, path = committed_profile("pmc_fetch_write.json" if kname == "reject_kernel" else "pmc_importance.json")
    if not d or d.get("kernel") != kname:
        return None, None
    return d.get("traffic_bytes_per_launch_calibrated", d["traffic_bytes_per_launch_raw"]), path


def search_traffic(args, w, slots):
    """Memory-side bytes per simulated step of the search kernel from the committed PMC passes of this command (profiles/r03_pmc_search.json:
    FETCH_SIZE, WRITE_SIZE, the L2 request / miss counters, and the steps the profiled launches made)."""
    if args.workload != "c2" or slots != 262144 or w["sims"] != 4096 or w["particles"] != 4096 or w["belief"] != "rejection_sampling":
        return None, None
    return committed_profile("pmc_search.json")


def spawn_ranks(n, argv=None, poll=0.1, grace=5.0, out=sys.stderr):
    """`python bench.py --gpus N` without a launcher: this process starts N fresh workers (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, as torch.distributed.run would) BEFORE anything here touches a GPU
    and polls them.  The first worker that exits non-zero ends the job: its siblings -- which would otherwise sit in the
    gloo rendezvous or in a barrier until its 600-s timeout -- are terminated (killed after `grace` seconds) and the exit
    code is that worker's.  Rank 0 prints the JSON line on the inherited stdout.  Returns the job's exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    argv = argv if argv is not None else [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    live = {}
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        live[r] = subprocess.Popen(argv, env=env)
    rc, deadline = 0, None
    while live:
        for r, p in list(live.items()):
            code = p.poll()
            if code is None:
                continue
            del live[r]
            if code != 0 and rc == 0:
                rc = abs(code) if abs(code) < 256 else 1
                print(f"[bench] rank {r} exited with code {code}: terminating the other {len(live)} rank(s)", file=out, flush=True)
                for q in live.values():
                    q.terminate()
                deadline = time.monotonic() + grace
        if deadline is not None and live and time.monotonic() > deadline:
            for q in live.values():
                q.kill()
            deadline = time.monotonic() + 3600.0
        if live:
            time.sleep(poll)
    return rc


def rccl_probe(dist, torch, world):
    """An RCCL group over all ranks that has carried one all-reduce of ones (the communicator is built lazily: make it fail here if it
    will), and the sum that all-reduce returned = the number of ranks RCCL saw."""
    os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")     # a probe that cannot complete raises after the timeout instead of hanging
    group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120))   # nccl = RCCL on ROCm
    probe = torch.ones(1, dtype=torch.float64, device="cuda")
    dist.all_reduce(probe, group=group)
    torch.cuda.synchronize()
    if int(probe.item()) != world:
        raise RuntimeError(f"probe all-reduce returned {probe.item()} for {world} ranks")
    return group, int(probe.item())


def open_collectives(dist, torch, rank, world, local_rank, shared, probe=rccl_probe):
    """The job's process groups.  Every rank first joins a gloo group (MASTER_ADDR / MASTER_PORT of the launcher): it carries the barriers
    and is where the ranks AGREE on the transport of the one data collective -- RCCL when every rank has its own GPU and every rank's
    probe all-reduce came back, gloo otherwise.  A rank never decides that alone: if some ranks fell back while others sat in an RCCL
    collective, the job would hang.  Returns (group for the statistics reduce or None = the gloo default group, its device, description,
    the number of ranks the RCCL probe's all-reduce counted or None)."""
    dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))
    if shared:   # RCCL needs one GPU per rank ("Duplicate GPU detected"): the five doubles go over gloo, and the line says so
        print(f"[bench] rank {rank}: ranks share GPU {local_rank} -> statistics reduce over gloo", file=sys.stderr)
        return None, "cpu", "gloo (ranks share a GPU)", None
    ok, err, group, seen = 1, "", None, None
    try:
        group, seen = probe(dist, torch, world)
    except Exception as e:   # the statistics reduce is 5 doubles: never lose a scaling run to the transport
        ok, err = 0, str(e).splitlines()[0][:200] if str(e) else type(e).__name__
        print(f"[bench] rank {rank}: RCCL probe failed ({err})", file=sys.stderr)
    agreed = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(agreed, op=dist.ReduceOp.MIN)                  # over gloo: every rank learns whether EVERY rank's RCCL works
    if int(agreed.item()) == 1:
        return group, "cuda", "rccl", seen
    if rank == 0:
        print("[bench] not every rank's RCCL probe succeeded: all ranks reduce over gloo", file=sys.stderr)
    return None, "cpu", "gloo (RCCL failed on at least one rank" + (f": {err}" if err else "") + ")", None


def agree_on_slots(dist, torch, slots):
    """After the per-rank out-of-memory step-down the ranks may hold different slot counts; weak scaling means the SAME work per GPU, so every
    rank takes the smallest (all-reduce MIN over the gloo group)."""
    t = torch.tensor([slots], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


def gather_per_rank(dist, torch, world, values):
    """[values of rank 0, values of rank 1, ...] over the gloo group (a few doubles per rank, for the JSON line)."""
    mine = torch.tensor(values, dtype=torch.float64)
    if world == 1:
        return [mine.tolist()]
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    return [p.tolist() for p in parts]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS),
                    help="BASELINE.json config: c1 = configs[0] ... c5 = configs[4]; default c2 = configs[1], the config `metric` is quoted on")
    ap.add_argument("--slots", type=int, default=None,
                    help="concurrent runs per GPU; default: the workload's own (c2: 16 search waves per CU, what a CU holds at 112 VGPRs and "
                         "10.1 KB of LDS per wave = 262144 on the 256 CUs of an MI355X, 0.66 MB of HBM each)")
    ap.add_argument("--sims", type=int, default=None)
    ap.add_argument("--particles", type=int, default=None)
    ap.add_argument("--horizon", type=int, default=None)
    ap.add_argument("--belief", default=None, choices=["rejection_sampling", "importance_sampling"])
    ap.add_argument("--search-budget", type=int, default=None,
                    help="c4: iterations of the search loop per launch (slots advance on their own; 0 = lock-step ticks); default: the workload's")
    ap.add_argument("--tree-buckets", type=int, default=None,
                    help="c4: 64-byte node buckets per slot's search tree (fba_config.tree_buckets; 0 = 2 * (sims + 2), which no search can fill); default: the workload's")
    ap.add_argument("--cpu-cores", type=int, default=16,
                    help="processes of the CPU baseline (a 1-GPU box's CPU share is 16 cores, whatever the affinity mask says)")
    ap.add_argument("--cpu-runs", type=int, default=None)
    ap.add_argument("--cpu-episodes", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--all-ranks-on-device", type=int, default=None,
                    help="rehearsal only: put every rank on this one GPU (a 1-GPU box cannot give each rank its own)")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker is not None:
        cpu_worker(args)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    import torch
    import torch.distributed as dist
    ndev = torch.cuda.device_count()
    if ndev < 1 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: fba_pomdp_amd has no CPU path")
    # every rank its own GPU; with fewer GPUs than ranks (a 1-GPU rehearsal box) ranks share devices, round-robin
    shared = world > ndev or args.all_ranks_on_device is not None
    local_rank = args.all_ranks_on_device if args.all_ranks_on_device is not None else local_rank % ndev
    torch.cuda.set_device(local_rank)
    group, red_dev, collective, rccl_ranks = None, "cpu", None, None
    if world > 1:
        group, red_dev, collective, rccl_ranks = open_collectives(dist, torch, rank, world, local_rank, shared)

    import fba_pomdp_amd as fba
    w = workload_of(args)
    if w["slots"] is None:   # c2: one search wave (64 runs) per wave slot of the chip: 4 per SIMD, 16 per CU
        w["slots"] = 64 * 16 * torch.cuda.get_device_properties(local_rank).multi_processor_count
    per_rank = w["slots"]
    slots = per_rank

    def make_engine(n):
        return fba.Engine(w["domain"], runs=1 << 30, slots=n, run_offset=rank * per_rank, seed=20261003, device=local_rank,
                          **{k: w[k] for k in ENGINE_KEYS if k in w})
    while True:  # step down if this GPU cannot give the memory right now (c2: 0.66 MB per slot, 172 GB)
        try:
            eng = make_engine(slots)
            break
        except fba.FbaError as e:
            if "out of memory" not in str(e) or slots <= 1:
                raise
            ladder = (32768, 16384) if args.workload == "c4" else (245760, 196608, 163840, 131072)   # (whole waves per SIMD: a partial round of waves costs more than it adds)
            nxt = next((v for v in ladder if v < slots), (slots * 7) // 8 if slots > 8 else slots // 2)
            print(f"[bench] rank {rank}: {slots} slots do not fit ({e}); retrying with {nxt}", file=sys.stderr)
            slots = nxt
    if world > 1:   # weak scaling = the same work on every GPU: all ranks run the smallest count any of them could allocate
        agreed = agree_on_slots(dist, torch, slots)
        if agreed != slots:
            print(f"[bench] rank {rank}: {slots} slots here, {agreed} on the tightest rank -> {agreed}", file=sys.stderr)
            eng.close()
            slots = agreed
            eng = make_engine(slots)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()   # the gloo group
        torch.cuda.synchronize()

    eng.run_ticks(args.warmup)          # sets the slots up (init + reset) and warms the caches
    c0 = eng.counters()
    eng.reset_kernel_times()
    barrier()
    t0 = time.perf_counter()
    eng.run_ticks(args.steps)           # synchronises its HIP stream before returning
    barrier()
    dt = time.perf_counter() - t0
    c1 = eng.counters()
    kt = eng.kernel_times()
    steps = (c1.sim_steps - c0.sim_steps) + (c1.belief_steps - c0.belief_steps)
    rets = eng.return_sums()

    tot = torch.tensor([float(steps), float(c1.sim_steps - c0.sim_steps), rets[0], rets[1], rets[2]],
                       dtype=torch.float64, device=red_dev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=group)
    tot = tot.cpu().tolist()
    dt_max = float(tmax.cpu()[0])
    per_rank_rows = gather_per_rank(dist, torch, world, [float(steps), dt, float(eng.slots)])

    if rank == 0:
        kname = "reject_kernel" if w["belief"] == "rejection_sampling" else "importance_kernel"
        k = kt[kname]
        achieved = (k.bytes / 1e9) / (k.ms / 1e3) if k.ms > 0 else 0.0
        basis = "SURVEY.md 8(d): dense fp32 particle, Pb = 4 + 4 C bytes, Rt / Ro = the rows one step consults"
        dense_equiv = achieved
        tiger_packed = eng.particle_bytes == 64 and w["domain"] == "episodic-tiger" and w["model"] == 1
        if tiger_packed and kname == "reject_kernel" and w["particles"] <= 4096:
            # packed particles + LDS-resident attempts: the engine reports the alternative formula;
            # SURVEY's dense figure for the same launches, for comparison (attempts/particle read back from the counters)
            basis = ("alternative formula stated in DESIGN.md section 5 (SURVEY 8(d) allows it for non-dense particles): 3 x 64 B per "
                     "particle written = one sequential read of the filter into LDS, N accepted sources read, N records written; "
                     "the rejection attempts themselves run from LDS")
            att = (c1.belief_steps - c0.belief_steps) / max(k.units, 1)
            dense_equiv = k.units * (att * 116.0 + 100.0) / 1e9 / (k.ms / 1e3) if k.ms > 0 else 0.0
        if tiger_packed and kname == "importance_kernel":
            basis = ("SURVEY.md 8(d)'s formula on the bytes a packed particle has (DESIGN.md section 5): update 32 + Rt + Ro = 48 B, resample "
                     "8 + 2 x 64 B: 184 B per particle written")
            dense_equiv = achieved * 256.0 / 184.0
        if w["domain"] == "gridworld" and eng.particle_bytes < 4096:
            basis = ("alternative formula stated in DESIGN.md section 5 (history particles): 44 + 12 x len bytes per particle written, len = "
                     "entries in the record before the update; the Dirichlet rows come from the shared prior tables (L2)")
        search = kt["search_kernel"]
        n_ep = tot[2]
        # `traffic` = PMC bytes for the SAME launches `achieved` is computed over: the profiled figure is that of a launch
        # in which every slot updates, a timed launch updates `updated_fraction` of them (no update after a terminal step)
        traffic_full, traffic_src = measured_traffic(args, w, kname, eng.slots)
        launches = max(int(k.launches), 1)
        updated_fraction = k.units / float(launches * eng.slots * w["particles"])
        traffic = traffic_full * updated_fraction if traffic_full else None
        rec_b = eng.particle_bytes
        min_traffic = k.units * 2.0 * rec_b     # every written particle read once, written once
        s_launches = max(int(search.launches), 1)
        s_avg_ms = search.ms / s_launches
        s_line = {"avg_ms": s_avg_ms, "share_of_timed_region": search.ms / (1e3 * dt_max), "steps_per_s": search.units / (search.ms / 1e3) if search.ms > 0 else 0.0,
                  "steps_per_launch": search.units / s_launches}
        sp, sp_src = search_traffic(args, w, eng.slots)
        if sp:
            # the search is a gather over node and particle records that are NOT cache resident at 262 144 slots (trees + filters: 34 GB): SURVEY
            # 8(d)'s "KB-scale, cache-resident" does not hold here.  What bounds it is the rate at which the memory system serves random 64-byte
            # lines (scripts/micro/randline: 48.7 G/s with every lane of 4 096 waves asking for its own -- 128-byte lines, 6.2 TB/s), DESIGN.md section 5c.
            bps = sp["fetch_bytes_per_step"] + sp["write_bytes_per_step"]
            sectors = sp["fetch_sectors_64B_per_step"] * s_line["steps_per_s"]
            s_line.update({
                "traffic": bps * s_line["steps_per_launch"], "traffic_source": sp_src,
                "bytes_per_step": bps, "fetch_bytes_per_step": sp["fetch_bytes_per_step"], "write_bytes_per_step": sp["write_bytes_per_step"],
                "l2_misses_per_step": sp.get("l2_misses_per_step"), "algorithmic_bytes_per_step": sp["algorithmic_bytes_per_step"],
                "frac_traffic": bps * s_line["steps_per_launch"] / 1e9 / (s_avg_ms / 1e3) / HBM_PEAK_GBS if s_avg_ms > 0 else None,
                "frac_algorithmic": sp["algorithmic_bytes_per_step"] * s_line["steps_per_launch"] / 1e9 / (s_avg_ms / 1e3) / HBM_PEAK_GBS if s_avg_ms > 0 else None,
                "random_sectors_per_s": sectors, "random_sector_ceiling_per_s": sp["random_sector_ceiling_G_per_s"] * 1e9,
                "frac_sector_ceiling": sectors / (sp["random_sector_ceiling_G_per_s"] * 1e9),
                "traffic_note": "fabric bytes per simulated step from the committed PMC passes of this command (FETCH_SIZE x 2: a read request is one "
                                "128-byte line tallied at 64 B, profiles/r04_randline_counters.json; + WRITE_SIZE) x the steps of a timed launch; "
                                "algorithmic = a MODEL estimate of the record bytes a step needs (DESIGN.md section 5c), not a measurement; "
                                "random_sectors_per_s = read requests (lines) per step x this run's search steps/s, against the micro-benchmark's "
                                "ceiling of random lines per second",
            })
        out = {
            "metric": "simulated env steps/sec (belief+rollout)",
            "value": tot[0] / dt_max,
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt_max / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",  # Q values, returns, weights; counts are f32, indices i32
            "data": "synthetic",
            "config": {
                "workload": f"{w['name']}; {w['belief']}, H={w['horizon']}" +
                            (f" [overridden: {w['sims']} sims, {w['particles']} particles]" if (args.sims or args.particles) else ""),
                "workload_key": args.workload, "slots_per_gpu": eng.slots, "parallelism": f"episode-sharded x{world}",
                "search_budget": w.get("search_budget", 0), "tree_buckets": w.get("tree_buckets", 0),
            },
            "roofline": {
                "bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "traffic_note": "PMC bytes (FETCH_SIZE calibrated on this kernel's access shapes + WRITE_SIZE) of a launch in which every slot "
                                "updates, scaled by updated_fraction: per launch, like algorithmic_bytes_per_launch",
                "traffic_full_launch": traffic_full, "algorithmic_bytes_full_launch": k.bytes / max(k.units, 1) * eng.slots * w["particles"],
                "updated_fraction": updated_fraction,
                "frac_traffic": (traffic / 1e9) / (k.ms / launches / 1e3) / HBM_PEAK_GBS if traffic and k.ms > 0 else None,
                "frac_min_traffic": (min_traffic / 1e9) / (k.ms / 1e3) / HBM_PEAK_GBS if k.ms > 0 else None,
                "launches": int(k.launches), "avg_ms": k.ms / max(int(k.launches), 1),
                "share_of_timed_region": k.ms / (1e3 * dt_max),   # (this rank's launches over the timed wall time: what the kernel weighs in `value`)
                "algorithmic_bytes_per_launch": k.bytes / max(int(k.launches), 1),
                "algorithmic_basis": basis,
                "particle_bytes_in_hbm": eng.particle_bytes, "survey_dense_formula_GBs": dense_equiv,
            },
            "search_kernel": s_line,
            "returns": {"episodes": n_ep, "mean": tot[3] / n_ep if n_ep else None,
                        "var": (tot[4] - tot[3] ** 2 / n_ep) / (n_ep - 1) if n_ep > 1 else None,
                        "collective": collective, "rccl_ranks": rccl_ranks},
            # what every rank did in the timed region (gathered over gloo): simulated steps, its own wall seconds, its slots
            "per_rank": {"steps": [r[0] for r in per_rank_rows], "seconds": [r[1] for r in per_rank_rows], "slots": [int(r[2]) for r in per_rank_rows]},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
