"""include/fba_hip.h is a C header: examples/planning_from_c.c (plain C99, -pedantic) builds against it and the library, and
on a GPU prints the statistics the Python binding gets for the same configuration."""
import os
import re
import subprocess

import pytest

import fba_pomdp_amd as fba

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    fba.build()
    exe = str(tmp_path / "planning_from_c")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "planning_from_c.c"), "-L" + os.path.join(ROOT, "fba_pomdp_amd"), "-lfba_hip",
                        "-Wl,-rpath," + os.path.join(ROOT, "fba_pomdp_amd"), "-lm", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return exe


def test_the_header_is_plain_c99(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe, "4", "8", "8"], capture_output=True, text=True)     # no GPU here: the library must say so, not fall back
    if r.returncode != 0:
        assert "fba_create" in r.stderr


@pytest.mark.gpu
def test_c_example_matches_the_python_binding(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe, "200", "128", "64"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    m = re.match(r"runs (\S+)  mean return (\S+)  stder (\S+)", r.stdout)
    eng = fba.Engine("episodic-tiger", sims=128, particles=64, runs=200, seed=7)
    st = eng.run_planning()
    assert float(m.group(1)) == st.count == 200
    assert float(m.group(2)) == float("%.6g" % st.mean) and float(m.group(3)) == float("%.6g" % st.stder)
