"""Same-box A/B of library builds / environment switches (run ON the GPU box; boxes of the pool differ by +-4 %):
python scripts/ab_bench.py <rounds> name=lib.so[,ENV=VAL...] ...   [-- extra bench.py arguments]
Variants are alternated round by round; one line per run: name, ms per tick, search ms, belief-update ms."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
rounds = int(args[0])
variants = []
for spec in args[1:]:
    name, rest = spec.split("=", 1)
    parts = rest.split(",")
    env = dict(p.split("=", 1) for p in parts[1:])
    variants.append((name, parts[0], env))
for r in range(rounds):
    for name, lib, env in variants:
        e = dict(os.environ, FBA_LIB=os.path.join(ROOT, "fba_pomdp_amd", lib), **env)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + extra, env=e, capture_output=True, text=True, timeout=400)
        try:
            d = json.loads(p.stdout.strip().splitlines()[-1])
            print(f"{name:24s} tick {d['ms_per_step']:8.2f} ms  search {d['search_kernel']['avg_ms']:8.2f}  belief {d['roofline']['avg_ms']:7.2f}  steps/s {d['value']:.4g}  search steps/s {d['search_kernel']['steps_per_s']:.4g}", flush=True)
        except Exception:
            print(name, "FAILED", p.stderr[-800:], flush=True)
