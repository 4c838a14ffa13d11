"""Where a wave of search_kernel spends its cycles: an instrumented build of the same sources (-DFBA_PROFILE_SEARCH:
clock64() marks between the regions of the search loop, summed per wave) run on the bench workload.
python scripts/search_regions.py build   (here: hipcc, no GPU needed)  ->  fba_pomdp_amd/libfba_hip_prof.so
python scripts/search_regions.py [slots] (on the GPU box)              ->  one JSON line"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "fba_pomdp_amd", "libfba_hip_prof.so")
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "build":
    from fba_pomdp_amd import _native as N
    N.build(extra_flags=["-DFBA_PROFILE_SEARCH"], lib_path=PROF, obj_tag="_prof")
    print("built", PROF)
    sys.exit(0)

os.environ["FBA_LIB"] = PROF
import fba_pomdp_amd as fba  # noqa: E402
from fba_pomdp_amd import _native as N  # noqa: E402

cfg = os.environ.get("FBA_CFG", "c2")
slots = int(sys.argv[1]) if len(sys.argv) > 1 else (81920 if cfg == "c3" else 262144)
if cfg == "c4":      # BASELINE configs[3]: gridworld N = 7, history particles, four lanes per tree (search_hist_kernel)
    slots = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    kw = dict(size=7, sims=int(os.environ.get("FBA_SIMS", "65536")), particles=16384, structure_prior=2, horizon=20, episodes=2,
              search_budget=int(os.environ.get("FBA_BUDGET", "16384")), tree_buckets=int(os.environ.get("FBA_BUCKETS", "32768")))   # (bench.py's c4 shape)
    eng = fba.Engine("gridworld", model=fba.MODEL_BA_FACTORED, belief="importance_sampling", runs=1 << 30, slots=slots, seed=20261003, **kw)
elif cfg == "c3":      # BASELINE configs[2]: factored tiger, 16384 simulations, packed records
    kw = dict(size=3, sims=16384, particles=4096, structure_prior=2, horizon=10, episodes=64)
    eng = fba.Engine("episodic-factored-tiger", model=fba.MODEL_BA_FACTORED, belief="rejection_sampling", runs=1 << 30, slots=slots, seed=20261003, **kw)
else:
    kw = dict(sims=4096, particles=4096, horizon=10, episodes=64) if cfg == "c2" else dict(sims=1024, particles=1024, horizon=10, episodes=64)
    eng = fba.Engine("episodic-tiger", model=fba.MODEL_BA_TABLE, belief="rejection_sampling", runs=1 << 30, slots=slots, seed=20261003, **kw)
L = N.load()
out = (C.c_ulonglong * 8)()
eng.run_ticks(int(os.environ.get("FBA_WARM", "1")) if cfg in ("c3", "c4") else 2)
L.fba_debug_search_profile(out, 1)
c0 = eng.counters()
eng.run_ticks(2 if cfg in ("c3", "c4") else 3)
L.fba_debug_search_profile(out, 0)
c1 = eng.counters()
names = ["root sample + particle fetch", "action (UCB / rollout draw)", "simulator step", "child lookup / expand / rollout sums", "back-up", "iterations", "loop head"]
v = list(out)
total = sum(v[i] for i in (0, 1, 2, 3, 4, 6))
steps = c1.sim_steps - c0.sim_steps
print(json.dumps({"slots": slots, "config": kw, "waves": slots // (16 if cfg == "c4" else 64), "sim_steps": steps, "wave_iterations": v[5],
                  "steps_per_wave_iteration": steps / max(v[5], 1),
                  "cycles_per_wave_iteration": total / max(v[5], 1),
                  "share": {names[i]: round(v[i] / total, 4) for i in (0, 1, 2, 3, 4, 6)}}))
