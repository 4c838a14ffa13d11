"""The `.res` files fba_experiment writes are the reference's own result format: the reference's offline tooling
(/root/reference/analysis/preprocess/merge_result_files.py, which pools the files of independent processes: the merge a
multi-GPU job's all-reduce reproduces) reads them unchanged and pools them as this repo's Statistic merge does.
Build container only: the script is run where /root/reference exists, on fixtures the CLI wrote on a GPU
(tests/golden/res/, README there)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RES = os.path.join(ROOT, "tests", "golden", "res")
MERGE = "/root/reference/analysis/preprocess/merge_result_files.py"
EXAMPLE = "/root/reference/analysis/plotting/example/1.res"


def _rows(path):
    lines = open(path).read().splitlines()
    return [l for l in lines if l.startswith("#")], np.array([[float(x) for x in l.split(",")] for l in lines if l and not l.startswith("#")])


def _pooled(a, b):
    """{mean, unbiased variance, count} of two samples pooled (what Statistic would hold had it seen both)"""
    n = a[:, 2] + b[:, 2]
    mu = (a[:, 0] * a[:, 2] + b[:, 0] * b[:, 2]) / n
    m2 = a[:, 1] * (a[:, 2] - 1) + b[:, 1] * (b[:, 2] - 1) + a[:, 2] * (a[:, 0] - mu) ** 2 + b[:, 2] * (b[:, 0] - mu) ** 2
    return mu, m2 / (n - 1), n


def test_result_files_have_the_shape_of_the_reference_example():
    head, _ = _rows(os.path.join(RES, "planning_a.res"))
    assert head == ["# version 1:", "# return mean, return var, return count, return stder, step duration mean"]
    _, plan = _rows(os.path.join(RES, "planning_a.res"))
    _, ba = _rows(os.path.join(RES, "bapomdp_a.res"))
    assert plan.shape == (1, 5) and ba.shape == (5, 5)               # one line per experiment / per episode index, five columns
    for rows in (plan, ba):
        assert np.allclose(rows[:, 3], np.sqrt(rows[:, 1] / rows[:, 2]), rtol=1e-4)   # stder column = sqrt(var / n) (Statistic.cpp:41-46)
    if os.path.exists(EXAMPLE):
        ref_head, ref_rows = _rows(EXAMPLE)
        assert ref_head == head and ref_rows.shape[1] == 5           # the reference's own example file: same header, same columns


@pytest.mark.skipif(not os.path.exists(MERGE), reason="the reference's analysis scripts are only in the build container")
@pytest.mark.parametrize("kind", ["planning", "bapomdp"])
def test_reference_merge_script_pools_our_result_files(kind):
    a, b = (os.path.join(RES, f"{kind}_{x}.res") for x in "ab")
    r = subprocess.run([sys.executable, MERGE, a, b], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr[-2000:]
    out = [l for l in r.stdout.splitlines() if l and not l.startswith("#")]
    got = np.array([[float(x) for x in l.split(",")] for l in out])
    ra, rb = _rows(a)[1], _rows(b)[1]
    mu, var, n = _pooled(ra, rb)
    assert got.shape == ra.shape
    assert np.array_equal(got[:, 2], n)
    assert np.allclose(got[:, 0], mu, rtol=1e-5) and np.allclose(got[:, 1], var, rtol=1e-4)      # (the files hold six significant digits)
    assert np.allclose(got[:, 3], np.sqrt(var / n), rtol=1e-4)
