"""The oracle's C restatement under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build; the GPU pool has no
sanitizer): every belief's whole-experiment path once in device arithmetic / Philox and once in reference arithmetic /
mt19937.  The checker has to be sound before it checks anything."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_runs_clean_under_asan_and_ubsan(tmp_path):
    lib = tmp_path / "liborc_san.so"
    obj = tmp_path / "orc_heap.o"
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
    subprocess.check_call(["g++", "-O1", "-g", "-fPIC", "-std=c++11"] + san + ["-c", os.path.join(ROOT, "oracle", "orc_heap.cpp"), "-o", str(obj)])
    subprocess.check_call(["gcc", "-O1", "-g", "-fPIC", "-std=gnu99", "-fno-fast-math", "-ffp-contract=off"] + san +
                          ["-shared", "-o", str(lib), os.path.join(ROOT, "oracle", "orc.c"), os.path.join(ROOT, "oracle", "orc_rng.c"), str(obj),
                           "-lm", "-lstdc++"])
    script = tmp_path / "run.py"
    script.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        from oracle import pyorc as orc
        orc._LIB = {str(lib)!r}
        B = dict(particles=12, sims=24, episodes=2, horizon=5, runs=2)
        cases = [
            dict(domain=0, model=0, belief=orc.BELIEF_REJECTION),
            dict(domain=1, model=1, belief=orc.BELIEF_IMPORTANCE),
            dict(domain=4, model=1, belief=orc.BELIEF_IMPORTANCE, size=3, particles=64),
            dict(domain=2, model=2, belief=orc.BELIEF_REINVIGORATION, size=2, structure_prior=2, resample_amount=4),
            dict(domain=4, model=2, belief=orc.BELIEF_CHEATING, size=3, structure_prior=2, resample_amount=3, threshold=-2.0),
            dict(domain=3, model=2, belief=orc.BELIEF_MH_GIBBS, size=2, structure_prior=2, threshold=-1.0),
            dict(domain=3, model=2, belief=orc.BELIEF_MH_GIBBS, belief_option=1, size=2, structure_prior=1, threshold=-1.0),
            dict(domain=5, model=2, belief=orc.BELIEF_MH_NIPS, width=3, height=3, size=1, structure_prior=1, threshold=-2.0),
            dict(domain=0, model=1, belief=orc.BELIEF_NESTED, particles=5),
            dict(domain=8, model=2, belief=orc.BELIEF_NESTED, size=3, particles=4),
            dict(domain=5, model=2, belief=orc.BELIEF_INCUBATOR, width=4, height=3, size=2, structure_prior=1, resample_amount=5, threshold=0.2),
            dict(domain=7, model=2, belief=orc.BELIEF_REJECTION, size=3, dirichlet_regular=1),
        ]
        for arith, rng in ((1, 1), (0, 0)):
            for kw in cases:
                kw = dict(B, **kw)
                if rng == 0:
                    kw["seed_str"] = "san"
                o = orc.Oracle(rng_mode=rng, arith=arith, philox_seed=3, trace=1, **kw)
                (o.run_planning if kw["model"] == 0 else o.run_bapomdp)()
                print(arith, kw["domain"], kw["belief"], flush=True)
        print("sanitized run complete")
        """))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    pre = [subprocess.check_output(["gcc", "-print-file-name=" + n], text=True).strip() for n in ("libasan.so", "libubsan.so")]
    env["LD_PRELOAD"] = ":".join(pre)
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "sanitized run complete" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
