"""Build and load libfba_hip.so (the C-ABI of include/fba_hip.h) through ctypes.

There is no fallback: if the shared library is missing or has no gfx950 device to run on, the
calls raise.  torch is not involved here; it is only used by bench.py for torch.distributed.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.environ.get("FBA_LIB") or os.path.join(HERE, "libfba_hip.so")   # (FBA_LIB: an instrumented build of the same sources, scripts/search_regions.py)
SOURCES = [os.path.join(HERE, "csrc", f) for f in ("fba_search.hip", "fba_kernels.hip", "fba_engine.hip")]
HEADERS = [os.path.join(HERE, "csrc", f) for f in ("fba_device.h", "fba_state.h", "fba_kernels.h", "fba_kernels_common.h")] + [
    os.path.join(ROOT, "include", "fba_hip.h")]
OBJ_DIR = os.path.join(HERE, "build")   # per-source objects (git-ignored): a change to one translation unit recompiles that one

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    # the reference rounds a*b+c twice; contraction into FMA would break bit parity
    "-ffp-contract=off", "-fno-fast-math",
    "-Wall", "-Wno-unused-function",
]

MAX_ACTIONS = 24
TRACE_HIST_BINS = 64
K_SEARCH, K_ENV, K_BELIEF_RS, K_BELIEF_IS, K_BELIEF_RESET, K_BELIEF_INIT, K_COUNT = range(7)
KERNEL_NAMES = ["search_kernel", "env_kernel", "reject_kernel", "importance_kernel", "reset_kernel", "init_kernel"]

DOM_TIGER_EPISODIC, DOM_TIGER_CONTINUOUS, DOM_FTIGER_EPISODIC, DOM_FTIGER_CONTINUOUS, DOM_GRIDWORLD, DOM_COLLISION_AVOID, DOM_COLLISION_AVOID_CENTERED = range(7)
DOM_SYSADMIN_INDEPENDENT, DOM_SYSADMIN_LINEAR = 7, 8
DOM_COFFEE, DOM_COFFEE_BOUTILIER = 9, 10
DOM_AGR = 11
MODEL_POMDP, MODEL_BA_TABLE, MODEL_BA_FACTORED = range(3)
BELIEF_REJECTION, BELIEF_IMPORTANCE, BELIEF_REINVIGORATION, BELIEF_CHEATING, BELIEF_POINT, BELIEF_MH_GIBBS, BELIEF_MH_NIPS, BELIEF_NESTED, BELIEF_INCUBATOR = range(9)
PLANNER_POUCT, PLANNER_RANDOM, PLANNER_TS = range(3)
OK, EINVAL, EHIP, ENODEVICE, ESTATE = 0, -1, -2, -3, -4

DOMAIN_NAMES = {  # reference -D strings (DomainConf.cpp)
    "episodic-tiger": DOM_TIGER_EPISODIC, "continuous-tiger": DOM_TIGER_CONTINUOUS,
    "episodic-factored-tiger": DOM_FTIGER_EPISODIC, "continuous-factored-tiger": DOM_FTIGER_CONTINUOUS,
    "gridworld": DOM_GRIDWORLD,
    "random-collision-avoidance": DOM_COLLISION_AVOID, "centered-collision-avoidance": DOM_COLLISION_AVOID_CENTERED,
    "independent-sysadmin": DOM_SYSADMIN_INDEPENDENT, "linear-sysadmin": DOM_SYSADMIN_LINEAR,
    "coffee": DOM_COFFEE, "boutilier-coffee": DOM_COFFEE_BOUTILIER, "agr": DOM_AGR,
}
BELIEF_NAMES = {"rejection_sampling": BELIEF_REJECTION, "importance_sampling": BELIEF_IMPORTANCE,
                "reinvigoration": BELIEF_REINVIGORATION, "cheating-reinvigoration": BELIEF_CHEATING,
                "point_estimate": BELIEF_POINT, "mh-within-gibbs": BELIEF_MH_GIBBS, "mh-nips": BELIEF_MH_NIPS, "nested": BELIEF_NESTED, "incubator": BELIEF_INCUBATOR}
PLANNER_NAMES = {"po-uct": PLANNER_POUCT, "random": PLANNER_RANDOM, "ts": PLANNER_TS}


class Config(C.Structure):
    _fields_ = [
        ("domain", C.c_int32), ("size", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
        ("model", C.c_int32), ("belief", C.c_int32), ("planner", C.c_int32),
        ("particles", C.c_int32), ("sims", C.c_int32), ("max_depth", C.c_int32), ("horizon", C.c_int32),
        ("exploration", C.c_double), ("discount", C.c_double),
        ("runs", C.c_int32), ("episodes", C.c_int32),
        ("noise", C.c_float), ("counts_total", C.c_float), ("structure_prior", C.c_int32),
        ("seed", C.c_uint64), ("run_offset", C.c_int32), ("slots", C.c_int32),
        ("device", C.c_int32), ("trace", C.c_int32), ("dirichlet_regular", C.c_int32),
        ("resample_amount", C.c_int32), ("threshold", C.c_double), ("belief_option", C.c_int32), ("search_budget", C.c_int32),
        ("tree_buckets", C.c_int32),
    ]


TRACE_DTYPE = np.dtype([
    ("run", "<i4"), ("episode", "<i4"), ("t", "<i4"),
    ("action", "<i4"), ("state", "<i4"), ("obs", "<i4"), ("terminal", "<i4"),
    ("n_nodes", "<i4"), ("tree_depth", "<i4"), ("update_count", "<i4"),
    ("root_n", "<i4", (MAX_ACTIONS,)), ("root_q", "<f8", (MAX_ACTIONS,)),
    ("reward", "<f8"), ("weight_total", "<f8"), ("belief_hash", "<u8"),
], align=True)


class Stat(C.Structure):
    _fields_ = [("count", C.c_double), ("mean", C.c_double), ("m2", C.c_double)]

    @property
    def var(self):
        return 0.0 if self.count < 2 else self.m2 / (self.count - 1)

    @property
    def stder(self):
        return 0.0 if self.count < 2 else (self.var / self.count) ** 0.5


class Counters(C.Structure):
    _fields_ = [("sim_steps", C.c_uint64), ("belief_steps", C.c_uint64), ("env_steps", C.c_uint64)]


MAX_FEATURES, MAX_NODES = 8, 160


class FactoredNode(C.Structure):
    _fields_ = [("offset", C.c_int32), ("out", C.c_int32), ("n_candidates", C.c_int32), ("mask_word", C.c_int32),
                ("fixed_mask", C.c_uint32), ("candidate", C.c_uint8 * MAX_FEATURES), ("candidate_size", C.c_uint8 * MAX_FEATURES)]


class FactoredLayout(C.Structure):
    _fields_ = [("n_state_features", C.c_int32), ("n_obs_features", C.c_int32), ("n_nodes", C.c_int32), ("n_counts", C.c_int32),
                ("n_mask_words", C.c_int32), ("state_feature_size", C.c_int32 * MAX_FEATURES),
                ("obs_feature_size", C.c_int32 * MAX_FEATURES), ("node", FactoredNode * MAX_NODES)]


class KernelTime(C.Structure):
    _fields_ = [("ms", C.c_double), ("launches", C.c_uint64), ("units", C.c_uint64), ("bytes", C.c_uint64)]


# every symbol include/fba_hip.h declares
EXPORTS = [
    "fba_abi_version", "fba_default_config", "fba_create", "fba_destroy", "fba_last_error",
    "fba_domain_sizes", "fba_counts_len", "fba_slots", "fba_particle_bytes", "fba_set_model_tabular", "fba_set_model_factored", "fba_log_bd_score", "fba_selftest_lgamma", "fba_get_prior", "fba_get_factored_layout",
    "fba_set_position", "fba_belief_init", "fba_belief_reset_domain_state", "fba_select_action",
    "fba_belief_update", "fba_belief_get", "fba_belief_get_particle", "fba_belief_set", "fba_belief_get_fully_connected", "fba_belief_get_nested", "fba_belief_get_shadow", "fba_last_step_info",
    "fba_run_planning", "fba_run_bapomdp", "fba_run_ticks", "fba_get_returns", "fba_get_counters", "fba_get_return_sums",
    "fba_get_kernel_times", "fba_reset_kernel_times", "fba_trace_count", "fba_get_trace", "fba_get_trace_hist",
    "fba_selftest_ucb", "fba_stat_add", "fba_stat_var", "fba_stat_stder",
]


def build(force=False, verbose=False, extra_flags=(), lib_path=None, obj_tag=""):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU) -> fba_pomdp_amd/libfba_hip.so.
    One object per source, compiled side by side; an object is rebuilt when its source or any header is newer."""
    lib_path = lib_path or LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdr_time = max(os.path.getmtime(h) for h in HEADERS)
    jobs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(src))[0] + obj_tag + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            cmd = ["hipcc"] + HIPCC_FLAGS + list(extra_flags) + ["-I" + os.path.join(ROOT, "include"), "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in jobs:
        if p.wait() != 0:
            for _, q in jobs:
                q.wait()
            raise subprocess.CalledProcessError(p.returncode, cmd)
    if jobs or not os.path.exists(lib_path) or any(os.path.getmtime(o) > os.path.getmtime(lib_path) for o in objs):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib_path]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib_path


CLI_PATH = os.path.join(HERE, "fba_experiment")
CLI_SOURCE = os.path.join(HERE, "csrc", "host", "fba_cli.cpp")


def build_cli(force=False):
    """g++ the reference-compatible command line (planning / bapomdp / fbapomdp) against the C-ABI."""
    if (not force and os.path.exists(CLI_PATH)
            and max(os.path.getmtime(CLI_SOURCE), os.path.getmtime(os.path.join(HERE, "csrc", "host", "conf_bridge.hpp"))) <= os.path.getmtime(CLI_PATH)
            and os.path.getmtime(LIB_PATH) <= os.path.getmtime(CLI_PATH)):
        return CLI_PATH
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc", "host"), CLI_SOURCE,
                           "-L" + HERE, "-lfba_hip", "-Wl,-rpath,$ORIGIN", "-o", CLI_PATH])
    return CLI_PATH


_lib = None


def load():
    """Load the HIP engine.  Raises if the library is absent -- there is no other implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc) first; "
                           "fba_pomdp_amd has no non-HIP implementation")
    L = C.CDLL(LIB_PATH)
    P = C.POINTER
    vp = C.c_void_p
    L.fba_abi_version.restype = C.c_int
    L.fba_default_config.argtypes = [P(Config)]
    L.fba_create.argtypes = [P(Config), P(vp)]
    L.fba_destroy.argtypes = [vp]
    L.fba_last_error.restype = C.c_char_p
    L.fba_last_error.argtypes = [vp]
    L.fba_domain_sizes.argtypes = [vp, P(C.c_int32), P(C.c_int32), P(C.c_int32)]
    L.fba_counts_len.argtypes = [vp]
    L.fba_particle_bytes.argtypes = [vp]
    L.fba_slots.argtypes = [vp]
    L.fba_set_model_tabular.argtypes = [vp, vp, vp]
    L.fba_set_model_factored.argtypes = [vp, vp, vp]
    L.fba_log_bd_score.argtypes = [vp, vp, vp, vp]
    L.fba_selftest_lgamma.argtypes = [vp, vp, C.c_int32, vp]
    L.fba_get_prior.argtypes = [vp, vp]
    L.fba_get_factored_layout.argtypes = [vp, vp]
    L.fba_set_position.argtypes = [vp, vp, vp, vp]
    L.fba_belief_init.argtypes = [vp]
    L.fba_belief_reset_domain_state.argtypes = [vp]
    L.fba_select_action.argtypes = [vp, vp, vp, vp]
    L.fba_belief_update.argtypes = [vp, vp, vp, vp]
    L.fba_belief_get.argtypes = [vp, C.c_int32, vp, vp, vp]
    L.fba_belief_set.argtypes = [vp, C.c_int32, vp, vp, vp]
    if not os.environ.get("FBA_LIB") or hasattr(L, "fba_belief_get_particle"):
        L.fba_belief_get_particle.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp]
    L.fba_belief_get_fully_connected.argtypes = [vp, C.c_int32, vp, vp]
    if not os.environ.get("FBA_LIB") or hasattr(L, "fba_belief_get_nested"):   # (FBA_LIB may name an older build for an A/B run)
        L.fba_belief_get_nested.argtypes = [vp, C.c_int32, vp]
        L.fba_belief_get_shadow.argtypes = [vp, C.c_int32, vp, vp, vp]
    L.fba_last_step_info.argtypes = [vp, vp]
    L.fba_run_planning.argtypes = [vp, P(Stat)]
    L.fba_run_bapomdp.argtypes = [vp, P(Stat)]
    L.fba_run_ticks.argtypes = [vp, C.c_int32]
    L.fba_get_returns.argtypes = [vp, vp, vp]
    L.fba_get_counters.argtypes = [vp, P(Counters)]
    L.fba_get_return_sums.argtypes = [vp, vp]
    L.fba_get_kernel_times.argtypes = [vp, P(KernelTime)]
    L.fba_reset_kernel_times.argtypes = [vp]
    L.fba_trace_count.argtypes = [vp]
    L.fba_get_trace.argtypes = [vp, vp, C.c_int32]
    if not os.environ.get("FBA_LIB") or hasattr(L, "fba_get_trace_hist"):
        L.fba_get_trace_hist.argtypes = [vp, vp, C.c_int32]
    L.fba_selftest_ucb.argtypes = [vp, vp, vp, C.c_int32, C.c_double, vp]
    L.fba_stat_add.argtypes = [P(Stat), C.c_double]
    L.fba_stat_var.restype = C.c_double
    L.fba_stat_var.argtypes = [P(Stat)]
    L.fba_stat_stder.restype = C.c_double
    L.fba_stat_stder.argtypes = [P(Stat)]
    _lib = L
    return L
