"""CPU-side checks of the product boundary: the C-ABI library builds for gfx950, loads, exports
every symbol include/fba_hip.h declares, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

import fba_pomdp_amd as fba
from fba_pomdp_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    fba.build()
    return fba.load()


def test_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "fba_hip.h")).read()
    declared = set(re.findall(r"\b(fba_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(N.EXPORTS), (declared ^ set(N.EXPORTS))
    for sym in declared:
        assert hasattr(lib, sym), f"libfba_hip.so does not export {sym}"
    assert lib.fba_abi_version() == 3   # 3: fba_config.tree_buckets; 2: fba_config.belief_option, fba_belief_get_particle


def test_default_config_matches_reference_cli_defaults(lib):
    cfg = N.Config()
    lib.fba_default_config(C.byref(cfg))
    # Conf.hpp:14-45, PlannerConf.hpp:16-18, BeliefConf.hpp:16-21, BAConf.hpp:17-22
    assert (cfg.runs, cfg.horizon, cfg.discount) == (1, 10, 0.95)
    assert (cfg.sims, cfg.max_depth, cfg.exploration) == (1000, -1, 100.0)
    assert cfg.particles == 100
    assert (cfg.episodes, cfg.noise, cfg.counts_total) == (1, 0.0, 10000.0)


def test_struct_layouts_match_between_engine_and_oracle():
    from oracle import pyorc as orc
    assert N.TRACE_DTYPE == orc.TRACE_DTYPE
    assert N.TRACE_DTYPE.itemsize == 352  # sizeof(fba_trace_rec) = sizeof(orc_trace_rec)


def test_statistic_matches_reference_known_answers(lib, golden):
    st = N.Stat()
    for x in golden["statistic"]["x"]:
        lib.fba_stat_add(C.byref(st), x)
    assert st.mean == golden["statistic"]["mean"]
    assert lib.fba_stat_var(C.byref(st)) == golden["statistic"]["var"]
    assert lib.fba_stat_stder(C.byref(st)) == golden["statistic"]["stder"]


def test_invalid_configurations_fail_like_the_reference(lib):
    def create(**kw):
        cfg = N.Config()
        lib.fba_default_config(C.byref(cfg))
        for k, v in kw.items():
            setattr(cfg, k, v)
        h = C.c_void_p()
        rc = lib.fba_create(C.byref(cfg), C.byref(h))
        return rc, lib.fba_last_error(None).decode()
    rc, msg = create(sims=0)
    assert rc == N.EINVAL and "cannot initiate POUCT with 0 simulations" in msg     # POUCT.cpp:34-38
    rc, msg = create(horizon=0)
    assert rc == N.EINVAL and "horizon" in msg                                        # POUCT.cpp:46-50
    rc, msg = create(particles=0)
    assert rc == N.EINVAL and "n = 0" in msg                                          # RejectionSampling.cpp:7-13


def test_no_gpu_means_loud_failure_not_a_cpu_path(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(fba.FbaError, match="no HIP device"):
        fba.Engine("episodic-tiger")


def test_product_never_touches_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "fba_pomdp_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("oracle/orc.c particle_hash", "").replace("oracle/orc.c dev_scan", ""), f


def test_the_documented_build_command_names_every_translation_unit():
    """INTEGRATION.md section 5 shows the hipcc command a maintainer can paste; it once fell a translation unit behind the build."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for src in N.SOURCES:
        assert os.path.relpath(src, ROOT) in text, src
