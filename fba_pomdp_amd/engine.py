"""Host-side mirror of the reference's experiment / planner / belief interface over the C-ABI.

Method names follow the reference: `select_action` = Planner::selectAction (Planner.hpp:23-24),
`belief_*` = Belief::initiate / updateEstimation / sample (Belief.hpp:25-40) and
BABelief::resetDomainStateDistribution (BABelief.hpp:34), `run_planning` / `run_bapomdp` =
experiment::planning::run / experiment::bapomdp::run.  Errors the reference throws as strings
surface as ValueError with the same wording.
"""
import ctypes as C

import numpy as np

from . import _native as N


class FbaError(RuntimeError):
    pass


def _enum(value, table, what):
    if isinstance(value, str):
        if value not in table:
            raise ValueError(f"{what} '{value}' is not supported")
        return table[value]
    return int(value)


class Engine:
    """One fba_ctx: `slots` independent (planner, belief) pairs resident on one MI355X."""

    def __init__(self, domain="episodic-tiger", model=N.MODEL_POMDP, belief="rejection_sampling",
                 planner="po-uct", **kw):
        self.L = N.load()
        cfg = N.Config()
        self.L.fba_default_config(C.byref(cfg))
        cfg.domain = _enum(domain, N.DOMAIN_NAMES, "domain")
        cfg.model = int(model)
        cfg.belief = _enum(belief, N.BELIEF_NAMES, "belief")
        cfg.planner = _enum(planner, N.PLANNER_NAMES, "planner")
        for k, v in kw.items():
            if not hasattr(cfg, k):
                raise AttributeError(f"fba_config has no field '{k}'")
            setattr(cfg, k, v)
        if cfg.belief == N.BELIEF_POINT:
            cfg.particles = 1   # what fba_create does with it; keeps the array sizes of this wrapper right
        self.cfg = cfg
        h = C.c_void_p()
        rc = self.L.fba_create(C.byref(cfg), C.byref(h))
        if rc != N.OK:
            msg = self.L.fba_last_error(None).decode()
            raise (ValueError if rc == N.EINVAL else FbaError)(msg)
        self.h = h
        S, A, O = C.c_int32(), C.c_int32(), C.c_int32()
        self.L.fba_domain_sizes(h, C.byref(S), C.byref(A), C.byref(O))
        self.S, self.A, self.O = S.value, A.value, O.value
        self.ncnt = self.L.fba_counts_len(h)
        self.slots = self.L.fba_slots(h)
        self.particle_bytes = self.L.fba_particle_bytes(h)

    def close(self):
        if getattr(self, "h", None):
            self.L.fba_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _chk(self, rc):
        if rc != N.OK:
            msg = self.L.fba_last_error(self.h).decode()
            raise (ValueError if rc == N.EINVAL else FbaError)(msg)

    # ---- experiments
    def run_planning(self):
        st = N.Stat()
        self._chk(self.L.fba_run_planning(self.h, C.byref(st)))
        return st

    def run_bapomdp(self):
        st = (N.Stat * self.cfg.episodes)()
        self._chk(self.L.fba_run_bapomdp(self.h, st))
        return list(st)

    def run_ticks(self, ticks):
        self._chk(self.L.fba_run_ticks(self.h, ticks))

    def returns(self):
        n = self.cfg.runs * self.cfg.episodes
        r = np.zeros(n, np.float64)
        ln = np.zeros(n, np.int32)
        self._chk(self.L.fba_get_returns(self.h, r.ctypes.data, ln.ctypes.data))
        return r.reshape(self.cfg.runs, self.cfg.episodes), ln.reshape(self.cfg.runs, self.cfg.episodes)

    def counters(self):
        c = N.Counters()
        self._chk(self.L.fba_get_counters(self.h, C.byref(c)))
        return c

    def return_sums(self):
        out = np.zeros(3, np.float64)
        self._chk(self.L.fba_get_return_sums(self.h, out.ctypes.data))
        return out

    def kernel_times(self):
        kt = (N.KernelTime * N.K_COUNT)()
        self._chk(self.L.fba_get_kernel_times(self.h, kt))
        return {N.KERNEL_NAMES[i]: kt[i] for i in range(N.K_COUNT)}

    def reset_kernel_times(self):
        self._chk(self.L.fba_reset_kernel_times(self.h))

    def trace(self):
        n = self.L.fba_trace_count(self.h)
        out = np.zeros(max(n, 1), N.TRACE_DTYPE)
        n = self.L.fba_get_trace(self.h, out.ctypes.data, len(out))
        if n < 0:
            self._chk(n)
        return out[:n]

    def trace_hist(self):
        """trace = 2: the filter's state histogram after the belief update of every trace record, in the order of trace()."""
        n = self.L.fba_trace_count(self.h)
        out = np.zeros((max(n, 1), N.TRACE_HIST_BINS), np.uint32)
        n = self.L.fba_get_trace_hist(self.h, out.ctypes.data, len(out))
        if n < 0:
            self._chk(n)
        return out[:n]

    def belief_get_particle(self, index, slot=0, weight=False):
        """One particle of the filter (Belief::sample() for a host planner that has drawn the index): state, weight, counts."""
        s = np.zeros(1, np.int32)
        w = np.zeros(1, np.float64)
        cnt = np.zeros(max(self.ncnt, 1), np.float32)
        self._chk(self.L.fba_belief_get_particle(self.h, slot, index, s.ctypes.data, w.ctypes.data if weight else None,
                                                 cnt.ctypes.data if self.ncnt else None))
        return int(s[0]), float(w[0]), cnt[:self.ncnt]

    # ---- per-step interface
    def prior(self):
        out = np.zeros(self.ncnt, np.float32)
        self._chk(self.L.fba_get_prior(self.h, out.ctypes.data))
        return out

    def factored_layout(self):
        """How a factored particle's count blob is laid out (include/fba_hip.h, fba_factored_layout)."""
        out = N.FactoredLayout()
        self._chk(self.L.fba_get_factored_layout(self.h, C.byref(out)))
        return out

    def set_model_tabular(self, phi, psi):
        phi = np.ascontiguousarray(phi, np.float32)
        psi = np.ascontiguousarray(psi, np.float32)
        self._chk(self.L.fba_set_model_tabular(self.h, phi.ctypes.data, psi.ctypes.data))

    def set_model_factored(self, counts, layout=None):
        """Replace the factored base prior: `counts` = a particle's whole blob (CPT counts, then the parent-set words) in
        the engine's layout (factored_layout())."""
        layout = layout or self.factored_layout()
        c = np.ascontiguousarray(counts, np.float32)
        assert c.size == self.ncnt
        self._chk(self.L.fba_set_model_factored(self.h, C.byref(layout), c.ctypes.data))

    def set_position(self, run=None, episode=None, t=None):
        def arr(x):
            if x is None:
                return None, None
            a = np.ascontiguousarray(np.broadcast_to(np.asarray(x, np.int32), (self.slots,)))
            return a, a.ctypes.data
        r, rp = arr(run)
        e, ep = arr(episode)
        tt, tp = arr(t)
        self._chk(self.L.fba_set_position(self.h, rp, ep, tp))

    def belief_init(self):
        self._chk(self.L.fba_belief_init(self.h))

    def belief_reset_domain_state(self):
        self._chk(self.L.fba_belief_reset_domain_state(self.h))

    def select_action(self, hist_len=0, active=None):
        hl = np.ascontiguousarray(np.broadcast_to(np.asarray(hist_len, np.int32), (self.slots,)))
        act = None if active is None else np.ascontiguousarray(active, np.uint8)
        out = np.zeros(self.slots, np.int32)
        self._chk(self.L.fba_select_action(self.h, hl.ctypes.data, None if act is None else act.ctypes.data, out.ctypes.data))
        return out

    def belief_update(self, action, obs, active=None):
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(action, np.int32), (self.slots,)))
        o = np.ascontiguousarray(np.broadcast_to(np.asarray(obs, np.int32), (self.slots,)))
        act = None if active is None else np.ascontiguousarray(active, np.uint8)
        self._chk(self.L.fba_belief_update(self.h, a.ctypes.data, o.ctypes.data, None if act is None else act.ctypes.data))

    def belief_get(self, slot=0, weights=None, counts=True):
        """States, weights and (counts=True) every particle's count table -- N x fba_counts_len floats, whatever the
        storage on the device; ask for counts=False where that is gigabytes."""
        n = self.cfg.particles
        s = np.zeros(n, np.int32)
        want_w = self.cfg.belief in (N.BELIEF_IMPORTANCE, N.BELIEF_CHEATING, N.BELIEF_MH_GIBBS, N.BELIEF_MH_NIPS, N.BELIEF_NESTED) if weights is None else weights
        w = np.zeros(n, np.float64)
        cnt = np.zeros((n, self.ncnt), np.float32) if counts else None
        self._chk(self.L.fba_belief_get(self.h, slot, s.ctypes.data, w.ctypes.data if want_w else None,
                                        cnt.ctypes.data if counts and self.ncnt else None))
        return s, w, cnt

    def belief_get_fully_connected(self, slot=0):
        """The second filter of the reinvigoration (fully connected) / cheating (correct graph) belief."""
        n = self.cfg.particles
        s = np.zeros(n, np.int32)
        cnt = np.zeros((n, self.ncnt), np.float32)
        self._chk(self.L.fba_belief_get_fully_connected(self.h, slot, s.ctypes.data, cnt.ctypes.data))
        return s, cnt

    def belief_get_shadow(self, slot=0):
        """The incubator belief's weighted shadow filter: states, weights, counts."""
        n = self.cfg.particles
        s = np.zeros(n, np.int32)
        w = np.zeros(n, np.float64)
        cnt = np.zeros((n, self.ncnt), np.float32)
        self._chk(self.L.fba_belief_get_shadow(self.h, slot, s.ctypes.data, w.ctypes.data, cnt.ctypes.data))
        return s, w, cnt

    def belief_get_nested(self, slot=0):
        """The nested belief's flat filters of domain states, [particles][particles^2] (belief_get returns the count particles)."""
        n = self.cfg.particles
        st = np.zeros((n, n * n), np.int32)
        self._chk(self.L.fba_belief_get_nested(self.h, slot, st.ctypes.data))
        return st

    def belief_set(self, slot, state=None, weight=None, counts=None):
        s = None if state is None else np.ascontiguousarray(state, np.int32)
        w = None if weight is None else np.ascontiguousarray(weight, np.float64)
        c = None if counts is None else np.ascontiguousarray(counts, np.float32)
        self._chk(self.L.fba_belief_set(self.h, slot, None if s is None else s.ctypes.data,
                                        None if w is None else w.ctypes.data, None if c is None else c.ctypes.data))

    def last_step_info(self):
        out = np.zeros(self.slots, N.TRACE_DTYPE)
        self._chk(self.L.fba_last_step_info(self.h, out.ctypes.data))
        return out
