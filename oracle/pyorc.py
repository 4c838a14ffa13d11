"""ctypes binding of oracle/liborc.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (fba_pomdp_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liborc.so")

# enums (oracle/orc.h)
DOM_TIGER_EPISODIC, DOM_TIGER_CONTINUOUS, DOM_FTIGER_EPISODIC, DOM_FTIGER_CONTINUOUS, DOM_GRIDWORLD, DOM_COLLISION_AVOID = range(6)
DOM_SYSADMIN_INDEPENDENT, DOM_SYSADMIN_LINEAR = 7, 8
DOM_COFFEE, DOM_COFFEE_BOUTILIER = 9, 10
DOM_AGR = 11
MODEL_POMDP, MODEL_BA_TABLE, MODEL_BA_FACTORED = range(3)
BELIEF_REJECTION, BELIEF_IMPORTANCE, BELIEF_REINVIGORATION, BELIEF_CHEATING, BELIEF_POINT, BELIEF_MH_GIBBS, BELIEF_MH_NIPS, BELIEF_NESTED, BELIEF_INCUBATOR = range(9)
ARITH_REF, ARITH_DEV = range(2)
RNG_MT, RNG_PHILOX = range(2)
PLANNER_POUCT, PLANNER_RANDOM, PLANNER_TS = range(3)
SP_NONE, SP_UNIFORM, SP_MATCH_UNIFORM, SP_FULLY_CONNECTED = range(4)
PH_INIT, PH_RESET, PH_START, PH_SEARCH, PH_ENV, PH_REJECT, PH_IS_UPDATE, PH_RESAMPLE = range(8)
MAX_ACTIONS = 24


class Config(C.Structure):
    _fields_ = [
        ("domain", C.c_int32), ("size", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
        ("model", C.c_int32), ("belief", C.c_int32), ("particles", C.c_int32),
        ("sims", C.c_int32), ("max_depth", C.c_int32), ("horizon", C.c_int32),
        ("exploration", C.c_double), ("discount", C.c_double),
        ("runs", C.c_int32), ("episodes", C.c_int32),
        ("noise", C.c_float), ("counts_total", C.c_float),
        ("structure_prior", C.c_int32), ("rng_mode", C.c_int32), ("arith", C.c_int32),
        ("philox_seed", C.c_uint64), ("seed_str", C.c_char * 64),
        ("run_offset", C.c_int32), ("trace", C.c_int32), ("planner", C.c_int32), ("ca_centered", C.c_int32), ("dirichlet_regular", C.c_int32),
        ("resample_amount", C.c_int32), ("threshold", C.c_double), ("belief_option", C.c_int32),
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


class Result(C.Structure):
    _fields_ = [("sim_steps", C.c_uint64), ("belief_steps", C.c_uint64), ("env_steps", C.c_uint64),
                ("seconds", C.c_double), ("n_trace", C.c_int32)]


class Rng(C.Structure):
    _fields_ = [("mode", C.c_int), ("mt", C.c_uint32 * 624), ("mti", C.c_int),
                ("key", C.c_uint32 * 2), ("ctr", C.c_uint32 * 4), ("blk", C.c_uint32 * 4),
                ("draw", C.c_uint32), ("blk_valid", C.c_int),
                ("run", C.c_uint32), ("episode", C.c_uint32), ("t", C.c_uint32),
                ("words", C.c_uint64)]


def build(force=False):
    """Compile liborc.so (building the checker is not using it)."""
    srcs = [os.path.join(_HERE, f) for f in ("orc.c", "orc_rng.c", "orc.h", "orc_rng.h", "Makefile")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liborc.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        P = C.POINTER
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [P(Config)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_error.restype = C.c_char_p
        L.orc_error.argtypes = [C.c_void_p]
        L.orc_run_planning.argtypes = [C.c_void_p, P(Stat), P(Result)]
        L.orc_run_bapomdp.argtypes = [C.c_void_p, P(Stat), P(Result)]
        L.orc_trace.restype = C.c_void_p
        L.orc_trace.argtypes = [C.c_void_p]
        L.orc_domain_sizes.argtypes = [C.c_void_p, P(C.c_int32), P(C.c_int32), P(C.c_int32)]
        L.orc_counts_len.argtypes = [C.c_void_p]
        L.orc_prior_counts.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_ctx_rng.restype = P(Rng)
        L.orc_ctx_rng.argtypes = [C.c_void_p]
        L.orc_env_step.argtypes = [C.c_void_p, P(C.c_int32), C.c_int32, P(C.c_int32), P(C.c_double)]
        L.orc_env_start.argtypes = [C.c_void_p]
        L.orc_chance_add_visit.argtypes = [P(C.c_int32), P(C.c_double), C.c_double]
        L.orc_chance_add_visit.restype = None
        L.orc_ext_terminal.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        L.orc_ext_reward.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        L.orc_ext_reward.restype = C.c_double
        L.orc_det_lgamma.argtypes = [C.c_double]
        L.orc_det_lgamma.restype = C.c_double
        L.orc_log_bd_score.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_log_bd_score.restype = C.c_double
        L.orc_random_action.argtypes = [C.c_void_p, C.c_int32]
        L.orc_belief_initiate.argtypes = [C.c_void_p]
        L.orc_belief_update.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.orc_is_update.restype = C.c_double
        L.orc_is_update.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.orc_is_resample.argtypes = [C.c_void_p]
        L.orc_belief_reset_domain_state.argtypes = [C.c_void_p]
        L.orc_select_action.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_belief_hash.restype = C.c_uint64
        L.orc_belief_hash.argtypes = [C.c_void_p]
        L.orc_last_update_count.argtypes = [C.c_void_p]
        L.orc_belief_get.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_belief_get_fc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_belief_get_nested.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_belief_get_shadow.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_least_likely.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_marginalize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_belief_set.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_model_step.argtypes = [C.c_void_p, C.c_void_p, P(C.c_int32), C.c_int32, P(C.c_int32), P(C.c_double), C.c_int]
        L.orc_model_obs_prob.restype = C.c_double
        L.orc_model_obs_prob.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        L.orc_ftiger_set_structure.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_gamma.restype = C.c_double
        L.orc_gamma.argtypes = [C.c_void_p, C.c_double]
        L.orc_sample_sampled_mult.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_sample_mult.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_det_log.restype = C.c_double
        L.orc_det_log.argtypes = [C.c_double]
        L.orc_det_exp.restype = C.c_double
        L.orc_det_exp.argtypes = [C.c_double]
        L.orc_dev_scan.restype = C.c_double
        L.orc_dev_scan.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        # rng
        L.orc_rng_init_mt_str.argtypes = [P(Rng), C.c_char_p, C.c_size_t]
        L.orc_rng_init_mt_u32.argtypes = [P(Rng), C.c_uint32]
        L.orc_rng_init_philox.argtypes = [P(Rng), C.c_uint64]
        L.orc_rng_episode.argtypes = [P(Rng), C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_rng_stream.argtypes = [P(Rng), C.c_uint32, C.c_uint32]
        L.orc_u01.restype = C.c_double
        L.orc_u01.argtypes = [P(Rng)]
        L.orc_bool.argtypes = [P(Rng)]
        L.orc_int.argtypes = [P(Rng), C.c_int]
        L.orc_slow_int.argtypes = [P(Rng), C.c_int, C.c_int]
        L.orc_mt_next.restype = C.c_uint32
        L.orc_mt_next.argtypes = [P(Rng)]
        L.orc_philox4x32_10.argtypes = [P(C.c_uint32 * 4), P(C.c_uint32 * 2), P(C.c_uint32 * 4)]
        L.orc_sample_expected_mult.argtypes = [P(Rng), C.c_void_p, C.c_int]
        L.orc_sample_from_mult_f.argtypes = [P(Rng), C.c_void_p, C.c_int, C.c_double]
        L.orc_expected_mult.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_stat_add.argtypes = [P(Stat), C.c_double]
        _lib = L
    return _lib


def least_likely(w, n):
    """WeightedFilter::leastLikely (oracle/orc_heap.cpp)"""
    L = lib()
    w = np.ascontiguousarray(w, np.float64)
    out = np.zeros(n, np.int32)
    L.orc_least_likely(w.ctypes.data, len(w), n, out.ctypes.data)
    return out


def make_config(**kw):
    """Defaults = the reference's CLI defaults (Conf.hpp:14-45, PlannerConf.hpp:16-18,
    BeliefConf.hpp:16-21, BAConf.hpp:17-22)."""
    c = Config()
    c.domain = DOM_TIGER_EPISODIC
    c.model = MODEL_POMDP
    c.belief = BELIEF_REJECTION
    c.particles = 100
    c.sims = 1000
    c.max_depth = -1
    c.horizon = 10
    c.exploration = 100.0
    c.discount = 0.95
    c.runs = 1
    c.episodes = 1
    c.noise = 0.0
    c.counts_total = 10000.0
    c.structure_prior = SP_NONE
    c.rng_mode = RNG_MT
    c.arith = ARITH_REF
    c.philox_seed = 0
    c.seed_str = b""
    for k, v in kw.items():
        if k == "seed_str" and isinstance(v, str):
            v = v.encode()
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    if c.belief == BELIEF_POINT:
        c.particles = 1   # what the library does with it; keeps the array sizes of this wrapper right
    return c


class Oracle:
    def __init__(self, **kw):
        self.cfg = kw["cfg"] if "cfg" in kw else make_config(**kw)
        self.L = lib()
        self.h = self.L.orc_create(C.byref(self.cfg))
        err = self.L.orc_error(self.h)
        if err:
            msg = err.decode()
            self.L.orc_destroy(self.h)
            self.h = None
            raise ValueError(msg)
        S, A, O = C.c_int32(), C.c_int32(), C.c_int32()
        self.L.orc_domain_sizes(self.h, C.byref(S), C.byref(A), C.byref(O))
        self.S, self.A, self.O = S.value, A.value, O.value
        self.ncnt = self.L.orc_counts_len(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def sizes(self):
        return self.S, self.A, self.O

    @property
    def rng(self):
        return self.L.orc_ctx_rng(self.h)

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self.L.orc_error(self.h).decode())

    def run_planning(self):
        st, res = Stat(), Result()
        self._check(self.L.orc_run_planning(self.h, C.byref(st), C.byref(res)))
        return st, res

    def run_bapomdp(self):
        st = (Stat * self.cfg.episodes)()
        res = Result()
        self._check(self.L.orc_run_bapomdp(self.h, st, C.byref(res)))
        return list(st), res

    def trace(self, n):
        if n == 0:
            return np.zeros(0, TRACE_DTYPE)
        ptr = self.L.orc_trace(self.h)
        buf = (C.c_char * (n * TRACE_DTYPE.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=TRACE_DTYPE, count=n).copy()

    def prior_counts(self):
        out = np.zeros(self.ncnt, np.float32)
        self._check(self.L.orc_prior_counts(self.h, out.ctypes.data))
        return out

    # ---- env
    def env_start(self):
        return self.L.orc_env_start(self.h)

    def random_action(self, s):
        return self.L.orc_random_action(self.h, s)

    def env_step(self, s, a):
        cs, o, r = C.c_int32(s), C.c_int32(), C.c_double()
        t = self.L.orc_env_step(self.h, C.byref(cs), a, C.byref(o), C.byref(r))
        return cs.value, o.value, r.value, t

    # ---- belief
    def belief_initiate(self):
        self.L.orc_belief_initiate(self.h)

    def belief_update(self, a, o):
        self.L.orc_belief_update(self.h, a, o)

    def is_update(self, a, o):
        return self.L.orc_is_update(self.h, a, o)

    def is_resample(self):
        self.L.orc_is_resample(self.h)

    def belief_reset_domain_state(self):
        self.L.orc_belief_reset_domain_state(self.h)

    def belief_get(self):
        n = self.cfg.particles
        s = np.zeros(n, np.int32)
        w = np.zeros(n, np.float64)
        cnt = np.zeros((n, self.ncnt), np.float32)
        self.L.orc_belief_get(self.h, s.ctypes.data, w.ctypes.data, cnt.ctypes.data if self.ncnt else None)
        return s, w, cnt

    def belief_get_shadow(self):
        """the incubator belief's weighted shadow filter"""
        n = self.cfg.particles
        s = np.zeros(n, np.int32)
        w = np.zeros(n, np.float64)
        cnt = np.zeros((n, self.ncnt), np.float32)
        self.L.orc_belief_get_shadow(self.h, s.ctypes.data, w.ctypes.data, cnt.ctypes.data)
        return s, w, cnt

    def belief_get_nested(self):
        """the flat filters of domain states of the nested belief: [particles][particles^2]"""
        n = self.cfg.particles
        st = np.zeros((n, n * n), np.int32)
        self.L.orc_belief_get_nested(self.h, st.ctypes.data)
        return st

    def belief_get_fc(self):
        n = self.cfg.particles
        s = np.zeros(n, np.int32)
        cnt = np.zeros((n, self.ncnt), np.float32)
        self.L.orc_belief_get_fc(self.h, s.ctypes.data, cnt.ctypes.data)
        return s, cnt

    def marginalize(self, cnt, masks):
        cnt = np.ascontiguousarray(cnt, np.float32)
        masks = np.ascontiguousarray(masks, np.uint32)
        out = np.zeros(self.ncnt, np.float32)
        if self.L.orc_marginalize(self.h, cnt.ctypes.data, masks.ctypes.data, out.ctypes.data):
            raise RuntimeError("marginalize: factored model only")
        return out

    def belief_set(self, s=None, w=None, cnt=None):
        s = None if s is None else np.ascontiguousarray(s, np.int32)
        w = None if w is None else np.ascontiguousarray(w, np.float64)
        cnt = None if cnt is None else np.ascontiguousarray(cnt, np.float32)
        self.L.orc_belief_set(self.h, None if s is None else s.ctypes.data,
                              None if w is None else w.ctypes.data,
                              None if cnt is None else cnt.ctypes.data)

    def belief_hash(self):
        return self.L.orc_belief_hash(self.h)

    def select_action(self, hist_len=0):
        rec = np.zeros(1, TRACE_DTYPE)
        a = self.L.orc_select_action(self.h, hist_len, rec.ctypes.data)
        return a, rec[0]

    def model_step(self, cnt, s, a, update):
        cs, o, r = C.c_int32(s), C.c_int32(), C.c_double()
        t = self.L.orc_model_step(self.h, cnt.ctypes.data, C.byref(cs), a, C.byref(o), C.byref(r), int(update))
        return cs.value, o.value, r.value, t

    def ftiger_set_structure(self, cnt, mask):
        assert self.L.orc_ftiger_set_structure(self.h, cnt.ctypes.data, mask) == 0

    def model_obs_prob(self, cnt, new_s, a, o):
        return self.L.orc_model_obs_prob(self.h, cnt.ctypes.data, new_s, a, o)


def dev_scan(w):
    w = np.ascontiguousarray(w, np.float64)
    incl = np.zeros_like(w)
    tot = lib().orc_dev_scan(w.ctypes.data, len(w), incl.ctypes.data)
    return tot, incl
