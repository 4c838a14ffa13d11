#!/usr/bin/env python3
"""bench.py -- simulated env steps / sec (belief + rollout) of the BA-POMCP hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N worker processes, one per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one real time-step of every resident slot: RBAPOUCT search (`sims` simulations),
true-environment step, and the particle-filter belief update.  Workload (N = 1 and per GPU for
N > 1) is BASELINE.json configs[1]: episodic-tiger BA-POMCP, 4096 sims/step, 4096 particles,
tabular BA-POMDP, expected-Dirichlet sampling, rejection-sampling belief (the reference default).
Particles are stored packed (24 uint16 increment counts over the shared prior + state: 64 B).
Runs are independent, so N GPUs run N disjoint sets of runs (weak scaling); the only collective is
one all-reduce of {episodes, sum of returns, sum of squares} + the step counters at the end.

`value` counts simulator.step calls of the planner (tree + rollout) plus those of the belief
update, exactly as BASELINE.md defines the metric; inputs (priors, particles, trees) are resident
in HBM before the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_worker(args):
    """One process of the CPU baseline (`bench.py --cpu-worker SEED`): the oracle in mt19937 mode (the
    reference's own arithmetic and draw order) on its own seed; prints one JSON line."""
    from oracle import pyorc as orc
    o = orc.Oracle(domain=orc.DOM_TIGER_EPISODIC, model=orc.MODEL_BA_TABLE, belief=orc.BELIEF_REJECTION,
                   sims=args.sims, particles=args.particles, horizon=args.horizon,
                   runs=args.cpu_runs, episodes=args.cpu_episodes, seed_str=args.cpu_worker)
    t0 = time.perf_counter()
    _, res = o.run_bapomdp()
    print(json.dumps({"steps": res.sim_steps + res.belief_steps, "env_steps": res.env_steps,
                      "seconds": time.perf_counter() - t0}), flush=True)


def cpu_baseline(args):
    """The oracle (CPU restatement of the reference path) timed on the host cores of this box on a
    bounded sample of the same workload: one PROCESS per available core (the reference algorithm is
    built on global RNG / static scratch state, SURVEY section 5, so no threads), disjoint seeds;
    aggregate = total simulated steps / longest process time.  Plain child processes with a hard
    timeout: they never touch the GPU."""
    import subprocess
    cores = max(1, min(len(os.sched_getaffinity(0)), args.cpu_cores or 10 ** 6))
    cmd = [sys.executable, os.path.abspath(__file__), "--sims", str(args.sims), "--particles", str(args.particles),
           "--horizon", str(args.horizon), "--cpu-runs", str(args.cpu_runs), "--cpu-episodes", str(args.cpu_episodes)]
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
    return {
        "value": steps / longest, "unit": "simulated env steps/s", "cores": len(out), "kind": "port",
        "per_core": per_core,
        "sample": f"{len(out)} processes x {args.cpu_runs} runs x {args.cpu_episodes} episodes of the same config "
                  f"({sum(o['env_steps'] for o in out)} real steps, {steps} simulated steps, longest process {longest:.1f} s, "
                  f"{wall:.1f} s wall, gcc -O2)",
    }


def measured_traffic(args, kname, slots):
    """HBM bytes of a FULL launch (every slot updates) of the roofline kernel from the committed rocprofv3 PMC
    passes (FETCH_SIZE and WRITE_SIZE in separate runs of this same command; profiles/*.json says how they were
    collected; FETCH_SIZE is corrected by the factor scripts/micro/pmc_calibrate measured for this kernel's access
    shapes).  None when the workload differs from the profiled one."""
    for tag in ("r02", "r01"):
        path = os.path.join(ROOT, "profiles", f"{tag}_pmc_fetch_write.json")
        if os.path.exists(path):
            break
    else:
        return None, None
    if slots != 262144 or args.sims != 4096 or args.particles != 4096:
        return None, None
    with open(path) as f:
        d = json.load(f)
    if d.get("kernel") != kname:
        return None, None
    return d.get("traffic_bytes_per_launch_calibrated", d["traffic_bytes_per_launch_raw"]), os.path.relpath(path, ROOT)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: this process starts N fresh workers (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, as torch.distributed.run would) BEFORE anything here touches a GPU,
    waits for them and leaves with the worst exit code.  Rank 0 prints the JSON line on the inherited stdout."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--slots", type=int, default=None,
                    help="concurrent runs per GPU (0.66 MB of HBM each at the default workload, packed particles); default: 16 search "
                         "waves per CU, what a CU holds at 116 VGPRs and 10.1 KB of LDS per wave = 262144 on the 256 CUs of an MI355X")
    ap.add_argument("--sims", type=int, default=4096)
    ap.add_argument("--particles", type=int, default=4096)
    ap.add_argument("--horizon", type=int, default=10)
    ap.add_argument("--belief", default="rejection_sampling", choices=["rejection_sampling", "importance_sampling"])
    ap.add_argument("--cpu-cores", type=int, default=16,
                    help="processes of the CPU baseline (a 1-GPU box's CPU share is 16 cores, whatever the affinity mask says)")
    ap.add_argument("--cpu-runs", type=int, default=4)
    ap.add_argument("--cpu-episodes", type=int, default=500)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--all-ranks-on-device", type=int, default=None,
                    help="rehearsal only: put every rank on this one GPU (a 1-GPU box cannot give each rank its own)")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker is not None:
        cpu_worker(args)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)   # does not return
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
    collective = None
    if world > 1:
        if shared:   # RCCL needs one GPU per rank ("Duplicate GPU detected"): the five doubles go over gloo, and the line says so
            print(f"[bench] rank {rank}: {world} ranks on {ndev} GPU(s), device {local_rank} is shared -> statistics reduce over gloo",
                  file=sys.stderr)
            dist.init_process_group(backend="gloo")
            collective = "gloo (ranks share a GPU)"
        else:
            try:
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))  # nccl = RCCL on ROCm
                probe = torch.ones(1, dtype=torch.float64, device="cuda")
                dist.all_reduce(probe)                       # the communicator is built lazily: make it fail here if it will
                torch.cuda.synchronize()
                assert int(probe.item()) == world
                collective = "rccl"
            except Exception as e:  # the statistics reduce is 5 doubles: never lose a scaling run to the transport
                print(f"[bench] rank {rank}: RCCL process group failed ({e}); tearing it down, reducing over gloo instead", file=sys.stderr)
                try:
                    dist.destroy_process_group()
                except Exception:
                    pass
                dist.init_process_group(backend="gloo", init_method=f"tcp://127.0.0.1:{int(os.environ.get('MASTER_PORT', '29500')) + 1}",
                                        rank=rank, world_size=world)
                collective = "gloo (RCCL failed)"

    import fba_pomdp_amd as fba
    if args.slots is None:   # one search wave (64 runs) per wave slot of the chip: 4 per SIMD, 16 per CU
        args.slots = 64 * 16 * torch.cuda.get_device_properties(local_rank).multi_processor_count
    slots = args.slots
    while True:  # 0.66 MB of HBM per slot: step down if this GPU cannot give 172 GB right now
        try:
            eng = fba.Engine("episodic-tiger", model=fba.MODEL_BA_TABLE, belief=args.belief,
                             sims=args.sims, particles=args.particles, horizon=args.horizon,
                             episodes=64, runs=1 << 30, slots=slots, run_offset=rank * args.slots,
                             seed=20261003, device=local_rank)
            break
        except fba.FbaError as e:
            if "out of memory" not in str(e) or slots <= 1024:
                raise
            nxt = next((v for v in (245760, 196608, 163840, 131072) if v < slots), slots // 2)
            print(f"[bench] {slots} slots do not fit ({e}); retrying with {nxt}", file=sys.stderr)
            slots = nxt

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
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

    red_dev = "cuda" if collective == "rccl" else "cpu"
    tot = torch.tensor([float(steps), float(c1.sim_steps - c0.sim_steps), rets[0], rets[1], rets[2]],
                       dtype=torch.float64, device=red_dev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tot = tot.cpu().tolist()
    dt_max = float(tmax.cpu()[0])

    if rank == 0:
        kname = "reject_kernel" if args.belief == "rejection_sampling" else "importance_kernel"
        k = kt[kname]
        achieved = (k.bytes / 1e9) / (k.ms / 1e3) if k.ms > 0 else 0.0
        basis = "SURVEY.md 8(d): dense fp32 particle, Pb = 100 B, Rt = Ro = 8 B"
        dense_equiv = achieved
        if eng.particle_bytes == 64 and kname == "reject_kernel" and args.particles <= 4096:
            # packed particles + LDS-resident attempts: the engine reports the alternative formula;
            # SURVEY's dense figure for the same launches, for comparison (attempts/particle read back from the counters)
            basis = ("alternative formula stated in DESIGN.md section 5 (SURVEY 8(d) allows it for non-dense particles): 3 x 64 B per "
                     "particle written = one sequential read of the filter into LDS, N accepted sources read, N records written; "
                     "the rejection attempts themselves run from LDS")
            att = (c1.belief_steps - c0.belief_steps) / max(k.units, 1)
            dense_equiv = k.units * (att * 116.0 + 100.0) / 1e9 / (k.ms / 1e3) if k.ms > 0 else 0.0
        if eng.particle_bytes == 64 and kname == "importance_kernel":
            basis = ("SURVEY.md 8(d)'s formula on the bytes a packed particle has (DESIGN.md section 5): update 32 + Rt + Ro = 48 B, resample "
                     "8 + 2 x 64 B: 184 B per particle written")
            dense_equiv = achieved * 256.0 / 184.0
        search = kt["search_kernel"]
        n_ep = tot[2]
        # `traffic` = PMC bytes for the SAME launches `achieved` is computed over: the profiled figure is that of a launch
        # in which every slot updates, a timed launch updates `updated_fraction` of them (no update after a terminal step)
        traffic_full, traffic_src = measured_traffic(args, kname, eng.slots)
        launches = max(int(k.launches), 1)
        updated_fraction = k.units / float(launches * eng.slots * args.particles)
        traffic = traffic_full * updated_fraction if traffic_full else None
        rec_b = eng.particle_bytes
        min_traffic = k.units * 2.0 * rec_b     # every written particle read once, written once
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
                "workload": "episodic-tiger BA-POMCP (tabular BA-POMDP, expected Dirichlet), "
                            f"{args.sims} sims/step, {args.particles} particles, {args.belief}, H={args.horizon}",
                "slots_per_gpu": eng.slots, "parallelism": f"episode-sharded x{world}",
            },
            "roofline": {
                "bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "traffic_note": "PMC bytes (FETCH_SIZE calibrated on this kernel's access shapes + WRITE_SIZE) of a launch in which every slot "
                                "updates, scaled by updated_fraction: per launch, like algorithmic_bytes_per_launch",
                "traffic_full_launch": traffic_full, "algorithmic_bytes_full_launch": k.bytes / max(k.units, 1) * eng.slots * args.particles,
                "updated_fraction": updated_fraction,
                "frac_traffic": (traffic / 1e9) / (k.ms / launches / 1e3) / HBM_PEAK_GBS if traffic and k.ms > 0 else None,
                "frac_min_traffic": (min_traffic / 1e9) / (k.ms / 1e3) / HBM_PEAK_GBS if k.ms > 0 else None,
                "launches": int(k.launches), "avg_ms": k.ms / max(int(k.launches), 1),
                "algorithmic_bytes_per_launch": k.bytes / max(int(k.launches), 1),
                "algorithmic_basis": basis,
                "particle_bytes_in_hbm": eng.particle_bytes, "survey_dense_formula_GBs": dense_equiv,
            },
            "search_kernel": {"avg_ms": search.ms / max(int(search.launches), 1),
                              "steps_per_s": search.units / (search.ms / 1e3) if search.ms > 0 else 0.0},
            "returns": {"episodes": n_ep, "mean": tot[3] / n_ep if n_ep else None,
                        "var": (tot[4] - tot[3] ** 2 / n_ep) / (n_ep - 1) if n_ep > 1 else None,
                        "collective": collective},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
