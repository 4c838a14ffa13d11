"""Parity of the HIP engine (through the C-ABI) with the CPU oracle on identical Philox streams.
Integer outputs (actions, states, observations, node counts, rejection counts) and the belief
checksum must be bit-exact; fp64 outputs (root Q values, total weights, returns) must be
bit-identical too, because both sides perform the same IEEE operations in the same order."""
import ctypes as C

import numpy as np
import pytest

import fba_pomdp_amd as fba
from fba_pomdp_amd import _native as N
from oracle import pyorc as orc

pytestmark = pytest.mark.gpu

DOM = {"random-collision-avoidance": orc.DOM_COLLISION_AVOID, "centered-collision-avoidance": orc.DOM_COLLISION_AVOID,
       "gridworld": orc.DOM_GRIDWORLD, "episodic-tiger": orc.DOM_TIGER_EPISODIC, "continuous-tiger": orc.DOM_TIGER_CONTINUOUS,
       "episodic-factored-tiger": orc.DOM_FTIGER_EPISODIC, "continuous-factored-tiger": orc.DOM_FTIGER_CONTINUOUS,
       "independent-sysadmin": orc.DOM_SYSADMIN_INDEPENDENT, "linear-sysadmin": orc.DOM_SYSADMIN_LINEAR,
       "coffee": orc.DOM_COFFEE, "boutilier-coffee": orc.DOM_COFFEE_BOUTILIER, "agr": orc.DOM_AGR}


def _pair(domain, model, belief, seed, slots=None, size=0, **kw):
    runs = kw.get("runs", 1)
    eng = fba.Engine(domain, model=model, belief=belief, seed=seed, slots=slots or runs, trace=1, size=size, **kw)
    okw = dict(kw)
    if domain == "centered-collision-avoidance":
        okw["ca_centered"] = 1
    if isinstance(okw.get("planner"), str):
        okw["planner"] = N.PLANNER_NAMES[okw["planner"]]
    o = orc.Oracle(domain=DOM[domain], model=model, belief=N.BELIEF_NAMES[belief], rng_mode=orc.RNG_PHILOX,
                   arith=orc.ARITH_DEV, philox_seed=seed, trace=1, size=size, **okw)
    return eng, o


def _assert_same_experiment(eng, o, ba):
    stats = eng.run_bapomdp() if ba else [eng.run_planning()]
    if ba:
        ostats, res = o.run_bapomdp()
    else:
        st, res = o.run_planning()
        ostats = [st]
    tr, otr = eng.trace(), o.trace(res.n_trace)
    assert len(tr) == len(otr)
    for name in tr.dtype.names:
        bad = np.nonzero(~np.all((tr[name] == otr[name]).reshape(len(tr), -1), axis=1))[0]
        assert bad.size == 0, f"{name}: first mismatch at record {bad[0]}: {tr[bad[0]]} vs {otr[bad[0]]}"
    for a, b in zip(stats, ostats):
        assert (a.count, a.mean, a.m2) == (b.count, b.mean, b.m2)
    c = eng.counters()
    assert (c.sim_steps, c.belief_steps, c.env_steps) == (res.sim_steps, res.belief_steps, res.env_steps)


def test_device_fp64_divide_and_sqrt_round_like_the_host():
    eng = fba.Engine("episodic-tiger", particles=4, sims=4, slots=1)
    rng = np.random.default_rng(0)
    m = rng.integers(0, 70000, 200000)
    n = rng.integers(1, 70000, 200000).astype(np.int32)
    L = np.log1p(m.astype(np.float64))
    out = np.zeros_like(L)
    eng._chk(eng.L.fba_selftest_ucb(eng.h, L.ctypes.data, n.ctypes.data, len(L), 100.0, out.ctypes.data))
    assert np.array_equal(out, 100.0 * np.sqrt(L / n))


@pytest.mark.parametrize("seed", [1, 77])
def test_planning_tiger_rejection(seed):
    eng, o = _pair("episodic-tiger", N.MODEL_POMDP, "rejection_sampling", seed, particles=64, sims=200, runs=24)
    _assert_same_experiment(eng, o, ba=False)


def test_planning_c1_shape():
    # BASELINE configs[0]: 1024 sims, 256 particles (fewer runs than the 10^4 of the statistical test)
    eng, o = _pair("episodic-tiger", N.MODEL_POMDP, "rejection_sampling", 5, particles=256, sims=1024, runs=16)
    _assert_same_experiment(eng, o, ba=False)


def test_planning_continuous_tiger_importance():
    eng, o = _pair("continuous-tiger", N.MODEL_POMDP, "importance_sampling", 3, particles=300, sims=150, runs=12, horizon=8)
    _assert_same_experiment(eng, o, ba=False)


def test_planning_slots_fewer_than_runs_reuses_slots():
    eng, o = _pair("episodic-tiger", N.MODEL_POMDP, "rejection_sampling", 9, slots=5, particles=32, sims=64, runs=23)
    _assert_same_experiment(eng, o, ba=False)


def test_planning_factored_tiger_true_dynamics():
    eng, o = _pair("episodic-factored-tiger", N.MODEL_POMDP, "rejection_sampling", 11, size=3, particles=128, sims=100, runs=10)
    _assert_same_experiment(eng, o, ba=False)


@pytest.mark.parametrize("belief", ["rejection_sampling", "importance_sampling"])
def test_bapomdp_tiger(belief):
    eng, o = _pair("episodic-tiger", N.MODEL_BA_TABLE, belief, 21, particles=200, sims=256, runs=10, episodes=4)
    _assert_same_experiment(eng, o, ba=True)


def test_bapomdp_tiger_noisy_prior_and_short_horizon():
    eng, o = _pair("continuous-tiger", N.MODEL_BA_TABLE, "rejection_sampling", 22, particles=100, sims=128, runs=6,
                   episodes=3, horizon=5, noise=0.1, counts_total=100.0)
    _assert_same_experiment(eng, o, ba=True)


def test_bapomdp_factored_tiger_flat_prior():
    eng, o = _pair("episodic-factored-tiger", N.MODEL_BA_TABLE, "importance_sampling", 23, size=2, particles=96,
                   sims=128, runs=6, episodes=3)
    _assert_same_experiment(eng, o, ba=True)


@pytest.mark.parametrize("sp,belief,size", [
    (0, "rejection_sampling", 3),      # "" = correct structure
    (2, "rejection_sampling", 3),      # match-uniform (BASELINE configs[2])
    (2, "importance_sampling", 2),
    (1, "importance_sampling", 3),     # uniform
    (3, "rejection_sampling", 2),      # fully-connected
])
def test_fbapomdp_factored_tiger(sp, belief, size):
    """fbapomdp -D episodic-factored-tiger: BABNModel / DBNNode sampling, per-particle structure
    drawn from the structure prior, observation-CPT increment quirk."""
    eng, o = _pair("episodic-factored-tiger", N.MODEL_BA_FACTORED, belief, 61 + sp, size=size, particles=150,
                   sims=200, runs=8, episodes=4, structure_prior=sp)
    _assert_same_experiment(eng, o, ba=True)


@pytest.mark.parametrize("sp,size,amount,domain", [
    (2, 3, 8, "episodic-factored-tiger"),       # match-uniform structure prior, the C3 domain
    (1, 2, 40, "episodic-factored-tiger"),      # uniform prior, many bred particles per update (victims collide)
    (0, 3, 1, "continuous-factored-tiger"),     # correct-structure prior, one bred particle
])
def test_fbapomdp_reinvigoration_belief(sp, size, amount, domain):
    """-B reinvigoration (ReinvigoratingRejectionSampling): breed = flip one edge of a sampled structure,
    marginalise a fully connected particle's counts onto it, replace a random particle; then rejection
    sampling on both filters."""
    eng, o = _pair(domain, N.MODEL_BA_FACTORED, "reinvigoration", 131 + sp, size=size, particles=120,
                   sims=150, runs=7, episodes=4, structure_prior=sp, resample_amount=amount, slots=4)
    _assert_same_experiment(eng, o, ba=True)


@pytest.mark.parametrize("sp,W,H,n,amount", [(0, 4, 3, 1, 5), (2, 5, 3, 2, 9), (1, 4, 3, 2, 3)])
def test_fbapomdp_reinvigoration_belief_collision_avoidance(sp, W, H, n, amount):
    """-B reinvigoration on collision avoidance: CollisionAvoidanceFactoredPrior::mutate picks an action and
    an obstacle and flips one of that node's edges; the fully connected filter has every state feature as
    a parent of every obstacle node."""
    eng, o = _pair("random-collision-avoidance", N.MODEL_BA_FACTORED, "reinvigoration", 231 + sp, size=n, width=W, height=H,
                   particles=60, sims=64, runs=5, episodes=3, structure_prior=sp, resample_amount=amount, slots=3)
    _assert_same_experiment(eng, o, ba=True)
    s, _, cnt = eng.belief_get(0)
    fs, fcnt = eng.belief_get_fully_connected(0)
    nvar = 3 * n
    assert np.all(fcnt.view(np.uint32)[:, -nvar:] == (1 << (2 + n)) - 1)
    assert len({tuple(m) for m in cnt.view(np.uint32)[:, -nvar:].tolist()}) > 1


@pytest.mark.parametrize("domain,size,amount", [("linear-sysadmin", 3, 6), ("independent-sysadmin", 4, 2), ("linear-sysadmin", 5, 20)])
def test_fbapomdp_reinvigoration_belief_sysadmin(domain, size, amount):
    """-B reinvigoration on sysadmin: every transition node may take any set of computers as parents
    (SysAdminFactoredPrior::mutate flips an edge of T[action][computer]); the fully connected filter
    starts from SysAdmin::failProbability with a total count of one per row."""
    eng, o = _pair(domain, N.MODEL_BA_FACTORED, "reinvigoration", 271 + size, size=size, particles=50, sims=64, runs=5,
                   episodes=3, horizon=6, resample_amount=amount, slots=3)
    _assert_same_experiment(eng, o, ba=True)
    fs, fcnt = eng.belief_get_fully_connected(0)
    nvar = 2 * size * size
    assert np.all(fcnt.view(np.uint32)[:, -nvar:] == (1 << size) - 1)
    s, _, cnt = eng.belief_get(0)
    assert len({tuple(m) for m in cnt.view(np.uint32)[:, -nvar:].tolist()}) > 1


def test_reinvigoration_belief_both_filters_equal_oracle_step_by_step():
    kw = dict(size=3, particles=96, sims=32, structure_prior=2, resample_amount=12)
    eng = fba.Engine("episodic-factored-tiger", model=N.MODEL_BA_FACTORED, belief="reinvigoration", seed=19, slots=1, **kw)
    o = orc.Oracle(domain=orc.DOM_FTIGER_EPISODIC, model=orc.MODEL_BA_FACTORED, belief=orc.BELIEF_REINVIGORATION,
                   rng_mode=orc.RNG_PHILOX, arith=orc.ARITH_DEV, philox_seed=19, **kw)
    L = orc.lib()

    def same():
        s, _, cnt = eng.belief_get(0)
        os_, _, ocnt = o.belief_get()
        assert np.array_equal(s, os_) and np.array_equal(cnt.view(np.uint32), ocnt.view(np.uint32))
        fs, fcnt = eng.belief_get_fully_connected(0)
        ofs, ofcnt = o.belief_get_fc()
        assert np.array_equal(fs, ofs) and np.array_equal(fcnt.view(np.uint32), ofcnt.view(np.uint32))
        return cnt.view(np.uint32)[:, -1], fcnt.view(np.uint32)[:, -1]

    L.orc_rng_episode(o.rng, 0, 0, 0)
    o.belief_initiate()
    eng.belief_init()
    o.belief_reset_domain_state()
    eng.belief_reset_domain_state()
    masks, fmasks = same()
    assert np.all(fmasks == 15) and np.all(masks & 1 == 1)
    seen_unforced = False
    for t, ob in enumerate([0, 1, 0, 0, 1]):
        L.orc_rng_episode(o.rng, 0, 0, t)
        eng.set_position(run=0, episode=0, t=t)
        o.belief_update(2, ob)
        eng.belief_update(2, ob)
        masks, fmasks = same()
        assert np.all(fmasks == 15)
        seen_unforced |= bool(np.any(masks & 1 == 0))     # mutate() may drop the tiger-location parent
        assert eng.last_step_info()[0]["update_count"] == L.orc_last_update_count(o.h)
    assert seen_unforced


@pytest.mark.parametrize("domain,kw", [
    ("episodic-factored-tiger", dict(size=3, structure_prior=1, threshold=-1.5, resample_amount=7)),
    ("gridworld", dict(size=3, structure_prior=2, threshold=-3.0, resample_amount=4, horizon=8)),
    ("random-collision-avoidance", dict(size=1, width=4, height=3, structure_prior=1, threshold=-2.0, resample_amount=3)),
    ("linear-sysadmin", dict(size=3, threshold=-1.0, resample_amount=5, horizon=6)),
])
def test_fbapomdp_cheating_reinvigoration_belief(domain, kw):
    """-B cheating-reinvigoration (prototypes/CheatingReinvigoration.cpp): importance sampling on the
    belief, rejection sampling on a second filter of correct-graph particles; when the accumulated log
    likelihood drops below --threshold, --resample-amount particles are copied across."""
    eng, o = _pair(domain, N.MODEL_BA_FACTORED, "cheating-reinvigoration", 291, particles=48, sims=64, runs=5, episodes=3, slots=3, **kw)
    _assert_same_experiment(eng, o, ba=True)
    fs, fcnt = eng.belief_get_fully_connected(0)
    assert fcnt.shape == (48, eng.ncnt)


def test_cheating_belief_cheats_and_keeps_both_filters_equal_to_the_oracle():
    kw = dict(size=2, particles=64, sims=16, structure_prior=1, resample_amount=10, threshold=-0.5)
    eng = fba.Engine("episodic-factored-tiger", model=N.MODEL_BA_FACTORED, belief="cheating-reinvigoration", seed=37, slots=1, **kw)
    o = orc.Oracle(domain=orc.DOM_FTIGER_EPISODIC, model=orc.MODEL_BA_FACTORED, belief=orc.BELIEF_CHEATING,
                   rng_mode=orc.RNG_PHILOX, arith=orc.ARITH_DEV, philox_seed=37, **kw)
    L = orc.lib()
    L.orc_rng_episode(o.rng, 0, 0, 0)
    o.belief_initiate()
    eng.belief_init()
    o.belief_reset_domain_state()
    eng.belief_reset_domain_state()
    correct_mask_seen = False
    for t, ob in enumerate([0, 1, 1, 0, 1, 0]):
        L.orc_rng_episode(o.rng, 0, 0, t)
        eng.set_position(run=0, episode=0, t=t)
        o.belief_update(2, ob)
        eng.belief_update(2, ob)
        s, w, cnt = eng.belief_get(0)
        os_, ow, ocnt = o.belief_get()
        assert np.array_equal(s, os_) and np.array_equal(w, ow) and np.array_equal(cnt.view(np.uint32), ocnt.view(np.uint32))
        fs, fcnt = eng.belief_get_fully_connected(0)
        ofs, ofcnt = o.belief_get_fc()
        assert np.array_equal(fs, ofs) and np.array_equal(fcnt.view(np.uint32), ofcnt.view(np.uint32))
        assert np.all(fcnt.view(np.uint32)[:, -1] == 1)                  # the second filter keeps the correct graph {tiger location}
        assert eng.last_step_info()[0]["weight_total"] > 0
    with pytest.raises(ValueError, match="resample_threshold >= 0"):
        fba.Engine("episodic-factored-tiger", model=N.MODEL_BA_FACTORED, belief="cheating-reinvigoration", size=2, particles=8,
                   sims=4, resample_amount=2, threshold=0.0)
    with pytest.raises(ValueError, match="needs a factored model"):
        fba.Engine("episodic-tiger", model=N.MODEL_BA_TABLE, belief="cheating-reinvigoration", particles=8, sims=4,
                   resample_amount=2, threshold=-1.0)


def test_reinvigoration_belief_argument_checks():
    with pytest.raises(ValueError, match="needs a factored model"):
        fba.Engine("episodic-tiger", model=N.MODEL_BA_TABLE, belief="reinvigoration", resample_amount=4, particles=8, sims=4)
    with pytest.raises(ValueError, match="resample size of < 1"):
        fba.Engine("episodic-factored-tiger", model=N.MODEL_BA_FACTORED, belief="reinvigoration", size=2, particles=8, sims=4)
    eng = fba.Engine("episodic-factored-tiger", model=N.MODEL_BA_FACTORED, size=2, particles=8, sims=4, slots=1)
    eng.belief_init()
    with pytest.raises(ValueError, match="second filter"):
        eng.belief_get_fully_connected(0)


def test_fbapomdp_prior_particles_equal_oracle():
    kw = dict(size=3, particles=64, sims=16, structure_prior=2)
    eng = fba.Engine("episodic-factored-tiger", model=N.MODEL_BA_FACTORED, seed=71, slots=1, **kw)
    o = orc.Oracle(domain=orc.DOM_FTIGER_EPISODIC, model=orc.MODEL_BA_FACTORED, rng_mode=orc.RNG_PHILOX,
                   arith=orc.ARITH_DEV, philox_seed=71, **kw)
    orc.lib().orc_rng_episode(o.rng, 0, 0, 0)
    o.belief_initiate()
    eng.belief_init()
    s, _, cnt = eng.belief_get(0)
    os_, _, ocnt = o.belief_get()
    assert np.array_equal(s, os_)
    assert np.array_equal(cnt.view(np.uint32), ocnt.view(np.uint32))   # counts and structure masks
    masks = cnt.view(np.uint32)[:, -1]
    assert np.all(masks & 1 == 1) and len(set(masks.tolist())) > 1       # match-uniform: feature 0 forced, rest random


@pytest.mark.parametrize("size", [3, 5])
def test_planning_gridworld_importance(size):
    """planning -D gridworld -B importance_sampling: true GridWorld dynamics as simulator, O = N*N*G
    observations (hashed child table for N = 5), computeObservationProbability in the filter."""
    eng, o = _pair("gridworld", N.MODEL_POMDP, "importance_sampling", 81, size=size, particles=120, sims=150,
                   runs=8, horizon=12)
    _assert_same_experiment(eng, o, ba=False)


@pytest.mark.parametrize("size,sp", [(3, 0), (3, 2), (5, 2)])
def test_fbapomdp_gridworld(size, sp):
    """fbapomdp -D gridworld -B importance_sampling (BASELINE configs[3] at parity size): factored
    prior, per-particle structure (goal as extra parent of the x / y nodes), BABNModel step."""
    eng, o = _pair("gridworld", N.MODEL_BA_FACTORED, "importance_sampling", 83 + sp, size=size, particles=64,
                   sims=100, runs=4, episodes=3, horizon=10, structure_prior=sp)
    _assert_same_experiment(eng, o, ba=True)


@pytest.mark.parametrize("size,belief,noise", [(3, "rejection_sampling", 0.0), (3, "importance_sampling", 0.1), (4, "rejection_sampling", 0.05)])
def test_bapomdp_gridworld_flat_prior(size, belief, noise):
    """bapomdp -D gridworld: GridWorldFlatBAPrior (S*A*S + A*S*O counts per particle: 23 KB at --size 3,
    131 KB at --size 4), rows of S entries sampled with the float CDF."""
    eng, o = _pair("gridworld", N.MODEL_BA_TABLE, belief, 211 + size, size=size, particles=48, sims=80, runs=5, episodes=3,
                   horizon=8, slots=3, noise=noise)
    assert np.array_equal(eng.prior().view(np.uint32), o.prior_counts().view(np.uint32))
    _assert_same_experiment(eng, o, ba=True)
    with pytest.raises(ValueError, match="Gridworld expects noise"):
        fba.Engine("gridworld", model=N.MODEL_BA_TABLE, size=3, noise=-0.1, particles=4, sims=4)


def test_fbapomdp_gridworld_prior_particles_equal_oracle():
    kw = dict(size=4, particles=32, sims=8, structure_prior=2, belief=1)
    eng = fba.Engine("gridworld", model=N.MODEL_BA_FACTORED, seed=91, slots=1, **kw)
    o = orc.Oracle(domain=orc.DOM_GRIDWORLD, model=orc.MODEL_BA_FACTORED, rng_mode=orc.RNG_PHILOX,
                   arith=orc.ARITH_DEV, philox_seed=91, **kw)
    assert np.array_equal(eng.prior().view(np.uint32), o.L.orc_counts_len(o.h) and _oracle_base_prior(o).view(np.uint32))
    orc.lib().orc_rng_episode(o.rng, 0, 0, 0)
    o.belief_initiate()
    eng.belief_init()
    s, _, cnt = eng.belief_get(0)
    os_, _, ocnt = o.belief_get()
    assert np.array_equal(s, os_)
    assert np.array_equal(cnt.view(np.uint32), ocnt.view(np.uint32))
    masks = cnt.view(np.uint32)[:, -8:]
    assert set(np.unique(masks).tolist()) == {3, 7}


def _oracle_base_prior(o):
    """the correct-structure prior record: what a particle holds under structure prior '' """
    base = orc.Oracle(domain=o.cfg.domain, size=o.cfg.size, model=orc.MODEL_BA_FACTORED, structure_prior=0)
    return base.prior_counts()


@pytest.mark.parametrize("kw", [
    dict(particles=1, sims=1, horizon=1),            # smallest everything
    dict(particles=3, sims=7, horizon=4, max_depth=0),   # depth 0: every simulation returns at the root
    dict(particles=5, sims=9, horizon=3, max_depth=50),  # depth cap beyond the horizon
    dict(particles=2, sims=33, horizon=12, discount=1.0, exploration=0.0),
])
def test_edge_sizes(kw):
    eng, o = _pair("episodic-tiger", N.MODEL_BA_TABLE, "rejection_sampling", 101, runs=6, episodes=3, **kw)
    _assert_same_experiment(eng, o, ba=True)
    eng, o = _pair("continuous-tiger", N.MODEL_POMDP, "importance_sampling", 102, runs=6, **kw)
    _assert_same_experiment(eng, o, ba=False)


def test_c3_size_one_search_and_update():
    """BASELINE configs[2] at its own sizes: factored tiger (K = 3), 16384 simulations, 4096 particles, match-uniform
    structure prior; one selectAction + one updateEstimation per slot, eight slots, compared with the oracle."""
    kw = dict(size=3, particles=4096, sims=16384, structure_prior=2)
    eng = fba.Engine("episodic-factored-tiger", model=N.MODEL_BA_FACTORED, seed=111, slots=8, **kw)
    L = orc.lib()
    eng.belief_init()
    eng.belief_reset_domain_state()
    acts = eng.select_action(hist_len=0)
    info = eng.last_step_info()
    eng.belief_update(2, 1)
    for e in range(8):
        o = orc.Oracle(domain=orc.DOM_FTIGER_EPISODIC, model=orc.MODEL_BA_FACTORED, rng_mode=orc.RNG_PHILOX,
                       arith=orc.ARITH_DEV, philox_seed=111, **kw)
        L.orc_rng_episode(o.rng, e, 0, 0)
        o.belief_initiate()
        o.belief_reset_domain_state()
        a_ref, rec = o.select_action(0)
        assert acts[e] == a_ref
        assert np.array_equal(info[e]["root_n"], rec["root_n"]) and np.array_equal(info[e]["root_q"], rec["root_q"])
        assert info[e]["n_nodes"] == rec["n_nodes"]
        o.belief_update(2, 1)
        s, _, cnt = eng.belief_get(e)
        os_, _, ocnt = o.belief_get()
        assert np.array_equal(s, os_) and np.array_equal(cnt.view(np.uint32), ocnt.view(np.uint32))


@pytest.mark.parametrize("size,sp,noise", [(3, 2, 0.0), (5, 2, 0.1), (7, 2, 0.0), (4, 0, 0.05)])
def test_history_particles_equal_dense_ones(size, sp, noise, monkeypatch):
    """The gridworld FBA-POMDP particle stored as its own history over the shared prior (Problem::hist, 4 bytes per
    real step) against the same experiment on dense count tables (FBA_DENSE_PARTICLES=1): every trace field, the
    checksum over every particle's whole count table included, every statistic, every counter."""
    kw = dict(model=N.MODEL_BA_FACTORED, belief="importance_sampling", seed=131 + size, size=size, particles=96, sims=120, runs=6,
              slots=6, episodes=3, horizon=9, structure_prior=sp, noise=noise, trace=1)
    hist = fba.Engine("gridworld", **kw)
    assert hist.particle_bytes == 4 * ((2 + 3 * 9 + 3) // 4 * 4)           # state, structure bits, episodes * horizon entries
    monkeypatch.setenv("FBA_DENSE_PARTICLES", "1")
    dense = fba.Engine("gridworld", **kw)
    monkeypatch.delenv("FBA_DENSE_PARTICLES")
    assert dense.particle_bytes > 100 * hist.particle_bytes or size < 5
    sh, sd = hist.run_bapomdp(), dense.run_bapomdp()
    th, td = hist.trace(), dense.trace()
    assert len(th) == len(td) > 0
    for name in th.dtype.names:
        assert np.array_equal(th[name], td[name]), name
    assert [(s.count, s.mean, s.m2) for s in sh] == [(s.count, s.mean, s.m2) for s in sd]
    ch, cd = hist.counters(), dense.counters()
    assert (ch.sim_steps, ch.belief_steps, ch.env_steps) == (cd.sim_steps, cd.belief_steps, cd.env_steps)


def test_fbapomdp_gridworld7_thousand_particles():
    """BASELINE configs[3] domain (gridworld N = 7, match-uniform structure prior, importance sampling) with 1024
    particles per belief against the oracle's dense 191 KB count tables: whole experiment, every trace field."""
    eng, o = _pair("gridworld", N.MODEL_BA_FACTORED, "importance_sampling", 977, size=7, particles=1024, sims=400, runs=2,
                   episodes=2, horizon=12, structure_prior=2)
    assert eng.particle_bytes <= 128
    _assert_same_experiment(eng, o, ba=True)


def test_history_particles_refuse_more_steps_than_they_hold():
    eng = fba.Engine("gridworld", model=N.MODEL_BA_FACTORED, belief="importance_sampling", size=3, particles=16, sims=8, horizon=2,
                     structure_prior=2, slots=1, seed=5)
    eng.belief_init()
    eng.belief_reset_domain_state()
    eng.belief_update(0, 0)
    eng.belief_update(1, 0)                      # episodes * horizon = 2 entries: full
    with pytest.raises(fba.FbaError, match="FBA_DENSE_PARTICLES"):
        eng.belief_update(0, 0)
    with pytest.raises(ValueError, match="FBA_DENSE_PARTICLES"):
        eng.belief_set(0, state=np.zeros(16, np.int32))


def test_gridworld7_one_search_and_update():
    """BASELINE configs[3] domain size (N = 7: S = O = 490, hashed child table), parity-sized counts."""
    kw = dict(size=7, particles=48, sims=600, structure_prior=2, belief=1, horizon=20)
    eng = fba.Engine("gridworld", model=N.MODEL_BA_FACTORED, seed=113, slots=1, **kw)
    o = orc.Oracle(domain=orc.DOM_GRIDWORLD, model=orc.MODEL_BA_FACTORED, rng_mode=orc.RNG_PHILOX,
                   arith=orc.ARITH_DEV, philox_seed=113, **kw)
    L = orc.lib()
    L.orc_rng_episode(o.rng, 0, 0, 0)
    o.belief_initiate()
    eng.belief_init()
    o.belief_reset_domain_state()
    eng.belief_reset_domain_state()
    for t, (a_fix, ob) in enumerate([(1, 10 * 7 + 0), (0, 10 * 7 * 1 + 10 + 3)]):
        L.orc_rng_episode(o.rng, 0, 0, t)
        eng.set_position(run=0, episode=0, t=t)
        a_ref, rec = o.select_action(t)
        a = eng.select_action(hist_len=t)[0]
        info = eng.last_step_info()[0]
        assert a == a_ref and info["n_nodes"] == rec["n_nodes"]
        assert np.array_equal(info["root_n"], rec["root_n"]) and np.array_equal(info["root_q"], rec["root_q"])
        o.belief_update(a_fix, ob)
        eng.belief_update(a_fix, ob)
        s, w, cnt = eng.belief_get(0)
        os_, ow, ocnt = o.belief_get()
        assert np.array_equal(s, os_) and np.array_equal(w, ow)
        assert np.array_equal(cnt.view(np.uint32), ocnt.view(np.uint32))


@pytest.mark.parametrize("domain", ["random-collision-avoidance", "centered-collision-avoidance"])
def test_planning_collision_avoidance(domain):
    """planning -D *-collision-avoidance: true dynamics (obstacle moves, rounded-normal observation
    noise, crash / arrival termination), importance filter with computeObservationProbability."""
    eng, o = _pair(domain, N.MODEL_POMDP, "importance_sampling", 121, size=2, width=5, height=3, particles=100,
                   sims=200, runs=10, horizon=8)
    _assert_same_experiment(eng, o, ba=False)


@pytest.mark.parametrize("sp,W,H,n", [(0, 5, 3, 3), (0, 7, 7, 2), (3, 4, 3, 1), (1, 4, 3, 1), (2, 5, 3, 2), (1, 4, 5, 2)])
def test_fbapomdp_collision_avoidance(sp, W, H, n):
    """fbapomdp -D random-collision-avoidance (largest factored domain; BASELINE configs[4] at
    parity size): correct-graph and fully-connected priors, and the edge-noise priors uniform (1) /
    match-uniform (2), whose obstacle nodes draw their parent sets per particle."""
    eng, o = _pair("random-collision-avoidance", N.MODEL_BA_FACTORED, "importance_sampling", 123 + sp, size=n, width=W,
                   height=H, particles=80, sims=150, runs=6, episodes=3, structure_prior=sp)
    _assert_same_experiment(eng, o, ba=True)


@pytest.mark.parametrize("domain,W,H,n,belief", [("random-collision-avoidance", 4, 3, 1, "rejection_sampling"),
                                                 ("centered-collision-avoidance", 3, 3, 2, "importance_sampling")])
def test_bapomdp_collision_avoidance_table_prior(domain, W, H, n, belief):
    """bapomdp -D *-collision-avoidance: CollisionAvoidanceTablePrior (S*A*S + A*S*O counts per particle)."""
    eng, o = _pair(domain, N.MODEL_BA_TABLE, belief, 251 + n, size=n, width=W, height=H, particles=40, sims=64, runs=5,
                   episodes=3, slots=3, noise=0.1, counts_total=500.0)
    assert np.array_equal(eng.prior().view(np.uint32), o.prior_counts().view(np.uint32))
    _assert_same_experiment(eng, o, ba=True)


def test_collision_avoidance_edge_noise_prior_particles_equal_oracle():
    """CollisionAvoidanceFactoredPrior::sampleFBAPOMDPState / sampleBlockTModel on the device."""
    for sp in (1, 2):
        kw = dict(width=4, height=3, size=2, structure_prior=sp, noise=0.1, counts_total=90.0, particles=96, sims=8)
        eng = fba.Engine("random-collision-avoidance", model=N.MODEL_BA_FACTORED, seed=29, slots=1, **kw)
        o = orc.Oracle(domain=orc.DOM_COLLISION_AVOID, model=orc.MODEL_BA_FACTORED, rng_mode=orc.RNG_PHILOX,
                       arith=orc.ARITH_DEV, philox_seed=29, **kw)
        orc.lib().orc_rng_episode(o.rng, 0, 0, 0)
        o.belief_initiate()
        eng.belief_init()
        s, _, cnt = eng.belief_get(0)
        os_, _, ocnt = o.belief_get()
        assert np.array_equal(s, os_)
        assert np.array_equal(cnt.view(np.uint32), ocnt.view(np.uint32))
        masks = cnt.view(np.uint32)[:, -6:]                       # var (a, obstacle) = a * n + obstacle
        assert len({tuple(m) for m in masks.tolist()}) > 50       # per-particle structures
        if sp == 2:
            assert np.all(masks[:, 0::2] & 4) and np.all(masks[:, 1::2] & 8)   # own edge forced
        else:
            assert not np.all(masks[:, 0::2] & 4)


def test_collision_avoidance_prior_equals_oracle():
    for sp in (0, 3):
        kw = dict(width=5, height=5, size=2, structure_prior=sp, noise=0.1, counts_total=500.0)
        eng = fba.Engine("random-collision-avoidance", model=N.MODEL_BA_FACTORED, particles=4, sims=4, slots=1, belief=1, **kw)
        o = orc.Oracle(domain=orc.DOM_COLLISION_AVOID, model=orc.MODEL_BA_FACTORED, **kw)
        assert np.array_equal(eng.prior(), o.prior_counts())


@pytest.mark.parametrize("domain,size,belief", [
    ("independent-sysadmin", 3, "rejection_sampling"), ("linear-sysadmin", 5, "importance_sampling"),
    ("linear-sysadmin", 8, "rejection_sampling")])
def test_planning_sysadmin(domain, size, belief):
    """planning -D *-sysadmin: 2N actions (ucb_pick<16> at N = 8), never terminal."""
    eng, o = _pair(domain, N.MODEL_POMDP, belief, 151 + size, size=size, particles=100, sims=160, runs=9, horizon=7, slots=4)
    _assert_same_experiment(eng, o, ba=False)


@pytest.mark.parametrize("domain,belief,planner", [
    ("coffee", "rejection_sampling", "po-uct"), ("coffee", "importance_sampling", "po-uct"),
    ("boutilier-coffee", "rejection_sampling", "po-uct"), ("boutilier-coffee", "importance_sampling", "ts")])
def test_planning_coffee(domain, belief, planner):
    """planning -D coffee | boutilier-coffee (CoffeeProblem.cpp): 32 states, 2 actions, fractional rewards,
    never terminal; both versions spend the same draws per step."""
    eng, o = _pair(domain, N.MODEL_POMDP, belief, 171, particles=150, sims=200, runs=10, horizon=8, slots=5, planner=planner)
    _assert_same_experiment(eng, o, ba=False)


@pytest.mark.parametrize("domain,model,kw", [
    ("continuous-tiger", N.MODEL_POMDP, dict(runs=40, horizon=10)),
    ("boutilier-coffee", N.MODEL_POMDP, dict(runs=12, horizon=8)),
    ("episodic-tiger", N.MODEL_BA_TABLE, dict(runs=12, episodes=4)),
    ("episodic-factored-tiger", N.MODEL_BA_FACTORED, dict(runs=10, episodes=3, size=2, structure_prior=2)),
    ("linear-sysadmin", N.MODEL_BA_FACTORED, dict(runs=6, episodes=2, size=3, horizon=6, planner="ts")),
])
def test_point_estimate_belief(domain, model, kw):
    """-B point_estimate (PointEstimation.cpp / BAPointEstimation.cpp): one (BA) state per slot, updated by rejection,
    sampled without a draw; `particles` is ignored."""
    eng, o = _pair(domain, model, "point_estimate", 191, particles=50, sims=128, slots=4, **kw)
    _assert_same_experiment(eng, o, ba=model != N.MODEL_POMDP)
    s, _, cnt = eng.belief_get(0)
    assert s.shape == (1,) and cnt.shape[0] == 1


@pytest.mark.parametrize("domain,kw", [
    ("episodic-factored-tiger", dict(size=3, structure_prior=2)),
    ("gridworld", dict(size=3, structure_prior=2)),
    ("random-collision-avoidance", dict(width=4, height=3, size=2, structure_prior=1)),
    ("linear-sysadmin", dict(size=3)),
])
def test_factored_layout_decodes_particles(domain, kw):
    """fba_get_factored_layout: a host that only knows the published layout rule (include/fba_hip.h) reproduces the
    model's observation probabilities P(o | a, s') = prod_f row_f[o_f] / sum(row_f) from a particle's raw blob."""
    eng = fba.Engine(domain, model=N.MODEL_BA_FACTORED, belief="importance_sampling", particles=16, sims=4, slots=1, seed=5, **kw)
    o = orc.Oracle(domain=DOM[domain], model=N.MODEL_BA_FACTORED, belief=orc.BELIEF_IMPORTANCE, particles=16, sims=4,
                   rng_mode=orc.RNG_PHILOX, arith=orc.ARITH_DEV, philox_seed=5, **kw)
    eng.belief_init()
    lay = eng.factored_layout()
    FS, FO = lay.n_state_features, lay.n_obs_features
    assert lay.n_nodes == eng.A * (FS + FO) and lay.n_counts + lay.n_mask_words == eng.ncnt
    ssz = list(lay.state_feature_size[:FS]); osz = list(lay.obs_feature_size[:FO])
    assert int(np.prod(ssz)) == eng.S and int(np.prod(osz)) == eng.O
    # node tables tile the count part of the blob, in node order, each with room for every candidate parent
    off = 0
    for k in range(lay.n_nodes):
        nd = lay.node[k]
        assert nd.offset == off
        off += nd.out * int(np.prod([nd.candidate_size[j] for j in range(nd.n_candidates)] or [1]))
        assert -1 <= nd.mask_word < lay.n_mask_words
    assert off == lay.n_counts
    _, _, cnt = eng.belief_get(0)

    def features(idx, sizes):   # last feature fastest
        out = []
        for z in reversed(sizes):
            out.append(idx % z); idx //= z
        return out[::-1]

    rng = np.random.default_rng(1)
    for _ in range(40):
        blob = cnt[rng.integers(len(cnt))]
        a, ns, ob = int(rng.integers(eng.A)), int(rng.integers(eng.S)), int(rng.integers(eng.O))
        v, of = features(ns, ssz), features(ob, osz)
        p = 1.0
        for f in range(FO):
            nd = lay.node[eng.A * FS + a * FO + f]
            mask = int(blob[lay.n_counts + nd.mask_word:][:1].view(np.uint32)[0]) if nd.mask_word >= 0 else nd.fixed_mask
            idx = 0
            for j in range(nd.n_candidates):
                if (mask >> j) & 1:
                    idx = idx * nd.candidate_size[j] + v[nd.candidate[j]]
            row = blob[nd.offset + idx * nd.out:][:nd.out].astype(np.float64)
            p *= row[of[f]] / row.sum() if row.sum() > 0 else 0.0
        want = o.model_obs_prob(np.ascontiguousarray(blob), ns, a, ob)
        assert abs(p - want) <= 1e-6 * max(want, 1e-12), (a, ns, ob, p, want)
    with pytest.raises(ValueError, match="not a factored model"):
        fba.Engine("episodic-tiger", model=N.MODEL_BA_TABLE, particles=4, sims=4).factored_layout()


@pytest.mark.parametrize("belief", ["rejection_sampling", "importance_sampling", "point_estimate"])
def test_packed_tiger_particles_equal_dense_ones(belief, monkeypatch):
    """Tabular tiger particles are stored as uint16 increment counts over the prior (64 B) when that is exact;
    FBA_DENSE_PARTICLES=1 keeps fp32 counts (128 B).  Same experiment, same trace, same particles, and both equal
    the oracle (the packed one through every other tiger test here)."""
    kw = dict(particles=300, sims=256, runs=12, episodes=5, slots=4, seed=77, trace=1)
    out = []
    for dense in (False, True):
        if dense:
            monkeypatch.setenv("FBA_DENSE_PARTICLES", "1")
        eng = fba.Engine("episodic-tiger", model=N.MODEL_BA_TABLE, belief=belief, **kw)
        monkeypatch.delenv("FBA_DENSE_PARTICLES", raising=False)
        assert eng.particle_bytes == (128 if dense else 64) and eng.ncnt == 24
        stats = eng.run_bapomdp()
        out.append(([(s.count, s.mean, s.m2) for s in stats], eng.trace(), [eng.belief_get(k) for k in range(4)], eng))
    assert out[0][0] == out[1][0]
    assert out[0][1].tobytes() == out[1][1].tobytes()
    for a, b in zip(out[0][2], out[1][2]):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    # fba_belief_set speaks fp32 counts in both formats: move a filter from the dense engine into the packed one
    packed, dense = out[0][3], out[1][3]
    s, w, cnt = dense.belief_get(2)
    packed.belief_set(0, state=s, weight=w if belief == "importance_sampling" else None, counts=cnt)
    s2, _, cnt2 = packed.belief_get(0)
    assert np.array_equal(s, s2) and np.array_equal(cnt, cnt2)
    with pytest.raises(ValueError, match="plus 0..65535 increments"):
        packed.belief_set(0, counts=cnt + 0.5)
    with pytest.raises(ValueError, match="FBA_DENSE_PARTICLES"):
        packed.set_model_tabular(np.full(12, 0.3, np.float32), np.full(12, 0.7, np.float32))
    packed.set_model_tabular(np.full(12, 7.0, np.float32), np.full(12, 2.5, np.float32))   # exact under + 65535: accepted
    with pytest.raises(fba.FbaError, match="belief not initiated"):    # packed particles are relative to the table: initiate again
        packed.belief_update(2, 0)
    packed.belief_init()
    assert np.array_equal(packed.belief_get(0)[2][0], np.r_[np.full(12, 7.0, np.float32), np.full(12, 2.5, np.float32)])
    # priors that are not exact under "+ 65535" are never packed
    eng = fba.Engine("episodic-tiger", model=N.MODEL_BA_TABLE, counts_total=777.0, particles=8, sims=8)
    assert eng.particle_bytes == 128


def test_packed_rejection_update_carries_every_cell_of_its_particles(monkeypatch):
    """reject_tiger_lds_kernel runs its attempts against an LDS copy of the rows the update's action can read; everything else a
    particle holds must still travel with it: particles whose open-door cells carry increments (fba_belief_set) and an update
    after a door (fba_belief_update with an open action: only a host can ask for it) must come out as on fp32 records."""
    kw = dict(model=N.MODEL_BA_TABLE, particles=200, sims=8, slots=1, seed=5)
    engines = []
    for dense in (False, True):
        if dense:
            monkeypatch.setenv("FBA_DENSE_PARTICLES", "1")
        eng = fba.Engine("episodic-tiger", **kw)
        monkeypatch.delenv("FBA_DENSE_PARTICLES", raising=False)
        eng.belief_init()
        eng.belief_reset_domain_state()
        engines.append(eng)
    packed, dense = engines
    assert packed.particle_bytes == 64 and dense.particle_bytes == 128

    def same():
        (s, _, c), (s2, _, c2) = packed.belief_get(0), dense.belief_get(0)
        assert np.array_equal(s, s2) and np.array_equal(c, c2)
        return s, c
    for eng in engines:                    # a listen update on pristine particles
        eng.set_position(run=0, episode=0, t=0)
        eng.belief_update(2, 1)
    s, c = same()
    assert np.any(c[:, 20:24] != c[0, 20:24][None, :]) or np.any(c[:, 20:24] != dense.prior()[20:24][None, :])   # listen cells were bumped
    c = c.copy()
    c[::2, 0] += 3                          # T(0, open-left, 0) of every other particle
    c[1::3, 13] += 2                        # O(open-left, 0, 1)
    for eng in engines:
        eng.belief_set(0, state=s, counts=c)
        eng.set_position(run=0, episode=0, t=1)
        eng.belief_update(2, 0)             # listen again, on particles that differ outside the listen cells
    s, c2 = same()
    assert np.any(c2[:, 0] != c2[0, 0]) and np.any(c2[:, 13] != c2[0, 13])     # the door increments travelled with their particles
    for eng in engines:
        eng.set_position(run=0, episode=0, t=2)
        eng.belief_update(0, 1)             # an update after opening a door (only a host can ask for it)
    same()


@pytest.mark.parametrize("planner,sims", [("po-uct", 300), ("ts", 120)])
def test_planning_agr(planner, sims):
    """planning -D agr (AGR.cpp, AGR(10)): 441 states, 23 actions (search_kernel<.., 24, ..>), 22 observations,
    deterministic dynamics; the first 23 simulations each open a new root action."""
    eng, o = _pair("agr", N.MODEL_POMDP, "rejection_sampling", 181, particles=300, sims=sims, runs=8, horizon=9, slots=4, planner=planner)
    _assert_same_experiment(eng, o, ba=False)


def test_agr_refuses_weighted_beliefs():
    with pytest.raises(ValueError, match="computeObservationProbability nyi"):    # AGR.cpp:307-310
        fba.Engine("agr", belief="importance_sampling", particles=4, sims=4)


def test_coffee_has_no_bayes_adaptive_model():
    with pytest.raises(ValueError, match="planning only"):
        fba.Engine("coffee", model=N.MODEL_BA_TABLE, particles=4, sims=4)


@pytest.mark.parametrize("domain,size,model,belief", [
    ("independent-sysadmin", 3, N.MODEL_BA_TABLE, "rejection_sampling"),
    ("linear-sysadmin", 4, N.MODEL_BA_TABLE, "importance_sampling"),
    ("independent-sysadmin", 4, N.MODEL_BA_FACTORED, "importance_sampling"),
    ("linear-sysadmin", 5, N.MODEL_BA_FACTORED, "rejection_sampling"),
    ("linear-sysadmin", 8, N.MODEL_BA_FACTORED, "rejection_sampling"),     # FS + FO = 9 increments per step
])
def test_bapomdp_and_fbapomdp_sysadmin(domain, size, model, belief):
    eng, o = _pair(domain, model, belief, 171 + size, size=size, particles=80, sims=96, runs=6, episodes=3, horizon=6, slots=3)
    _assert_same_experiment(eng, o, ba=True)


def test_sysadmin_priors_equal_oracle():
    for dom, size in (("independent-sysadmin", 3), ("linear-sysadmin", 5)):
        for model in (N.MODEL_BA_TABLE, N.MODEL_BA_FACTORED):
            eng = fba.Engine(dom, model=model, size=size, particles=4, sims=4, slots=1)
            o = orc.Oracle(domain=DOM[dom], model=model, size=size)
            assert np.array_equal(eng.prior().view(np.uint32), o.prior_counts().view(np.uint32)), (dom, model)
    with pytest.raises(ValueError, match="Structure noise is not enabled"):
        fba.Engine("linear-sysadmin", model=N.MODEL_BA_FACTORED, size=3, structure_prior=2, particles=4, sims=4)
    with pytest.raises(ValueError, match="Sysadmin with n 0"):
        fba.Engine("linear-sysadmin", size=0, particles=4, sims=4)


def test_bapomdp_slots_fewer_than_runs():
    eng, o = _pair("episodic-tiger", N.MODEL_BA_TABLE, "rejection_sampling", 24, slots=3, particles=64, sims=64,
                   runs=8, episodes=2)
    _assert_same_experiment(eng, o, ba=True)


@pytest.mark.parametrize("domain,model,belief,kw", [
    ("episodic-tiger", N.MODEL_POMDP, "rejection_sampling", {}),
    ("episodic-tiger", N.MODEL_BA_TABLE, "importance_sampling", dict(episodes=3)),
    ("gridworld", N.MODEL_BA_FACTORED, "importance_sampling", dict(episodes=2, size=3, structure_prior=2, horizon=8)),
    ("linear-sysadmin", N.MODEL_BA_FACTORED, "rejection_sampling", dict(episodes=2, size=3, horizon=6)),
])
def test_thompson_sampling_planner(domain, model, belief, kw):
    """-P ts: TSPlanner / BATSPlanner -- one belief sample, then PO-UCT from that single particle."""
    kw = dict(kw)
    eng, o = _pair(domain, model, belief, 311, size=kw.pop("size", 0), planner="ts", particles=64, sims=96, runs=6, slots=3, **kw)
    _assert_same_experiment(eng, o, ba=model != N.MODEL_POMDP)


def test_random_planner():
    eng, o = _pair("episodic-tiger", N.MODEL_POMDP, "rejection_sampling", 31, particles=32, sims=10, runs=40,
                   planner=N.PLANNER_RANDOM)
    _assert_same_experiment(eng, o, ba=False)


def test_prior_tables_equal_oracle():
    for dom, size in [("episodic-tiger", 0), ("episodic-factored-tiger", 2)]:
        eng = fba.Engine(dom, model=N.MODEL_BA_TABLE, size=size, particles=4, sims=4, slots=1, noise=0.05)
        o = orc.Oracle(domain=DOM[dom], model=orc.MODEL_BA_TABLE, size=size, noise=0.05)
        assert np.array_equal(eng.prior(), o.prior_counts())


@pytest.mark.parametrize("dom", ["episodic-factored-tiger", "continuous-factored-tiger"])
def test_factored_tiger_priors_have_the_reference_tests_known_answers(dom):
    """/root/reference/test/domains/priors/TigerPriorTest.cpp:230-655 through the C-ABI: the flat prior (fba_get_prior), states
    sampled from the factored prior with and without structure noise (fba_belief_init + fba_belief_get), the fully connected
    and match-uniform structures -- the same checks tests/test_oracle_golden.py makes on the oracle."""
    import prior_known_answers as K
    for size in (1, 2, 3):
        for n in K.NOISES:
            eng = fba.Engine(dom, model=N.MODEL_BA_TABLE, size=size, noise=n, counts_total=K.TOTAL, particles=4, sims=4, slots=1)
            assert (eng.S, eng.A, eng.O) == (2 << size, 3, 2)
            K.check_flat(eng.prior(), size, n)
            eng.close()
            for sp in (0, 1):
                eng = fba.Engine(dom, model=N.MODEL_BA_FACTORED, size=size, noise=n, counts_total=K.TOTAL, structure_prior=sp, particles=12,
                                 sims=4, slots=1, seed=1000 + size)
                eng.belief_init()
                _, _, cnt = eng.belief_get(0)
                masks = {K.check_factored(cnt[i], size, n, sp != 0) for i in range(12)}
                assert (masks == {1}) if sp == 0 else (len(masks) > 1)
                if sp == 0:
                    K.check_factored(eng.prior(), size, n, False)
                eng.close()
        eng = fba.Engine(dom, model=N.MODEL_BA_FACTORED, size=size, structure_prior=2, particles=10, sims=4, slots=1, seed=5)
        eng.belief_init()
        assert all(K.factored_parts(c, size)[4] & 1 for c in eng.belief_get(0)[2])            # match-uniform: feature 0 is always a parent
        eng.close()
        eng = fba.Engine(dom, model=N.MODEL_BA_FACTORED, size=size, structure_prior=3, particles=3, sims=4, slots=1, seed=5)
        eng.belief_init()
        assert all(K.factored_parts(c, size)[4] == (2 << size) - 1 for c in eng.belief_get(0)[2])   # fully connected: every feature
        eng.close()


def test_collision_avoidance_priors_have_the_reference_tests_known_answers():
    """/root/reference/test/domains/priors/CollisionAvoidancePriorTests.cpp:15-131 (table prior) and :215-340 (factored prior) through
    the C-ABI (fba_get_prior) -- the checks tests/test_oracle_golden.py makes on the oracle."""
    import prior_known_answers as K
    for total in (3.0, 10.0, 19.0):
        eng = fba.Engine("random-collision-avoidance", model=N.MODEL_BA_TABLE, width=5, height=7, size=1, counts_total=total, noise=0.0,
                         particles=4, sims=4, slots=1)
        assert (eng.S, eng.A, eng.O) == (5 * 7 * 7, 3, 7)
        K.check_ca_flat(eng.prior(), 5, 7)
        eng.close()
    eng = fba.Engine("random-collision-avoidance", model=N.MODEL_BA_FACTORED, width=4, height=7, size=2, counts_total=1000.0, noise=0.0,
                     particles=4, sims=4, slots=1)
    K.check_ca_factored(eng.prior(), 4, 7, 2, pytest.approx)
    eng.close()


@pytest.mark.parametrize("belief", ["rejection_sampling", "importance_sampling"])
def test_trace_histograms_are_the_filters_states_after_each_update(belief):
    """cfg.trace = 2: the state histogram recorded with every trace record equals the histogram of the states fba_belief_get
    returns when the same run is replayed through the per-step interface (same streams, same filter)."""
    kw = dict(model=N.MODEL_BA_TABLE, belief=belief, sims=48, particles=56, horizon=6, episodes=2, runs=1)
    eng = fba.Engine("continuous-tiger", seed=77, slots=1, trace=2, **kw)
    eng.run_bapomdp()
    tr, hist = eng.trace(), eng.trace_hist()
    assert len(tr) == len(hist) == 12
    rep = fba.Engine("continuous-tiger", seed=77, slots=1, **kw)
    rep.set_position(run=0, episode=0, t=0)
    rep.belief_init()
    for rec, hrow in zip(tr, hist):
        if rec["t"] == 0:
            rep.set_position(run=0, episode=int(rec["episode"]), t=0)
            rep.belief_reset_domain_state()
        rep.set_position(run=0, episode=int(rec["episode"]), t=int(rec["t"]))
        assert rep.select_action(hist_len=int(rec["t"]))[0] == rec["action"]
        rep.belief_update(int(rec["action"]), int(rec["obs"]))
        s, _, _ = rep.belief_get(0)
        assert np.array_equal(np.bincount(s, minlength=N.TRACE_HIST_BINS)[:N.TRACE_HIST_BINS], hrow)


@pytest.mark.parametrize("size,budget", [(3, 1), (3, 37), (5, 250), (5, 1000000)])
def test_budgeted_searches_give_the_results_of_whole_searches(size, budget):
    """fba_config.search_budget: a launch stops every history-particle search at the first simulation boundary behind `budget`
    loop iterations, parks it in its tree and the next launch resumes it, while slots whose search is done take their real step
    and belief update -- slots advance on their own.  Runs never interact (BAPOMDPExperiment.cpp:44-75), so every trace record
    (actions, root statistics, tree sizes, weights, the checksum over every particle), every statistic and every counter must
    be those of lock-step ticks, whatever the budget (1: a launch per simulation)."""
    kw = dict(model=N.MODEL_BA_FACTORED, belief="importance_sampling", size=size, sims=96, particles=64, structure_prior=2, horizon=7,
              episodes=2, runs=7, slots=4, seed=606 + size, trace=1)
    out = []
    for b in (0, budget):
        eng = fba.Engine("gridworld", search_budget=b, **kw)
        assert eng.particle_bytes < 4096           # history particles: the records search_hist_kernel reads
        stats = eng.run_bapomdp()
        c = eng.counters()
        out.append((eng.trace(), [(s.count, s.mean, s.m2) for s in stats], (c.sim_steps, c.belief_steps, c.env_steps), eng.returns()))
        eng.close()
    (t0, s0, c0, r0), (t1, s1, c1, r1) = out
    assert len(t0) == len(t1) > 14 and s0 == s1 and c0 == c1
    for name in t0.dtype.names:
        assert np.array_equal(t0[name], t1[name]), name
    assert np.array_equal(r0[0], r1[0]) and np.array_equal(r0[1], r1[1])


@pytest.mark.parametrize("size,sims,buckets,budget", [(3, 128, 128, 0), (3, 200, 0, 0), (5, 160, 160, 0), (5, 300, 0, 37), (7, 256, 256, 0), (4, 128, 130, 11)])
def test_the_bucket_tree_gives_the_results_of_node_records(size, sims, buckets, budget, monkeypatch):
    """search_hist2_kernel keeps a slot's tree in one open-addressing table of 64-byte buckets -- a node IS its bucket, keyed by (parent
    bucket, action, observation); nodes never reached again are 4-byte keys in the buckets' spare words -- and asks for every line an
    iteration before it looks at it.  None of that may show in a result: every trace field (actions, root statistics, node counts, tree
    depths, the belief checksum), statistic and counter equals search_hist_kernel's on node records + hash table (FBA_HIST_TREE=records),
    with the default table (2 * (sims + 2) buckets, never full) and with one of `sims` buckets, which a dense little gridworld fills to the
    brim: most lookups then walk past their home line (the probe sequences of nodes and of keys end at different lines)."""
    kw = dict(model=N.MODEL_BA_FACTORED, belief="importance_sampling", size=size, sims=sims, particles=64, structure_prior=2, horizon=7,
              episodes=2, runs=9, slots=5, seed=1300 + size, trace=1, search_budget=budget)
    out = []
    for records in (True, False):
        if records:
            monkeypatch.setenv("FBA_HIST_TREE", "records")
        else:
            monkeypatch.delenv("FBA_HIST_TREE")
        eng = fba.Engine("gridworld", tree_buckets=0 if records else buckets, **kw)
        assert eng.particle_bytes < 4096
        stats = eng.run_bapomdp()
        c = eng.counters()
        out.append((eng.trace(), [(s.count, s.mean, s.m2) for s in stats], (c.sim_steps, c.belief_steps, c.env_steps), eng.returns()))
        eng.close()
    (t0, s0, c0, r0), (t1, s1, c1, r1) = out
    assert len(t0) == len(t1) > 14 and s0 == s1 and c0 == c1
    for name in t0.dtype.names:
        assert np.array_equal(t0[name], t1[name]), name
    assert np.array_equal(r0[0], r1[0]) and np.array_equal(r0[1], r1[1])


@pytest.mark.parametrize("horizon,episodes", [(35, 2), (60, 2)])
def test_long_histories_and_deep_horizons_on_the_bucket_tree(horizon, episodes, monkeypatch):
    """Records of more than 63 entries (episodes * horizon > 63): the quad's lanes cannot count in 6-bit fields, so every lane walks every entry for its
    own feature as search_hist_kernel does; a horizon of 60 also makes a wave's paths long enough that a workgroup holds one or two waves instead of
    four.  Same results as node records + hash table, and as the oracle."""
    kw = dict(model=N.MODEL_BA_FACTORED, belief="importance_sampling", size=3, sims=48, particles=64, structure_prior=2, horizon=horizon,
              episodes=episodes, runs=4, slots=4, seed=77 + horizon, trace=1)
    out = []
    for records in (True, False):
        if records:
            monkeypatch.setenv("FBA_HIST_TREE", "records")
        else:
            monkeypatch.delenv("FBA_HIST_TREE")
        eng = fba.Engine("gridworld", **kw)
        assert eng.particle_bytes < 4096
        stats = eng.run_bapomdp()
        out.append((eng.trace(), [(s.count, s.mean, s.m2) for s in stats]))
        eng.close()
    (t0, s0), (t1, s1) = out
    assert len(t0) == len(t1) and s0 == s1
    for name in t0.dtype.names:
        assert np.array_equal(t0[name], t1[name]), name
    okw = {k: v for k, v in kw.items() if k not in ("slots", "seed", "belief")}
    o = orc.Oracle(domain=orc.DOM_GRIDWORLD, belief=orc.BELIEF_IMPORTANCE, rng_mode=orc.RNG_PHILOX, arith=orc.ARITH_DEV, philox_seed=kw["seed"], **okw)
    ostats, res = o.run_bapomdp()
    otr = o.trace(res.n_trace)
    assert len(otr) == len(t1)
    for name in t1.dtype.names:
        assert np.array_equal(t1[name], otr[name]), name


@pytest.mark.parametrize("particles", [64, 4096])
def test_records_at_the_stride_of_their_length_hold_the_beliefs_of_full_stride_records(particles, monkeypatch):
    """History-particle records stand 64 bytes apart up to 14 entries, 128 up to 30, the whole record beyond (hist_stride): the per-step interface walks
    a slot through 36 updates -- across both thresholds, by one workgroup per slot (64 particles) and by several (4 096) -- and after every one the filter
    read back through fba_belief_get (states, weights, the materialised count tables) and single particles through fba_belief_get_particle are those of
    a context whose records are always the full stride apart (FBA_HIST_STRIDE=full); the episode boundary's reset in between included."""
    kw = dict(model=N.MODEL_BA_FACTORED, belief="importance_sampling", size=3, sims=16, particles=particles, structure_prior=2, horizon=18,
              episodes=2, slots=2, seed=91)
    monkeypatch.setenv("FBA_HIST_STRIDE", "full")
    full = fba.Engine("gridworld", **kw)
    monkeypatch.delenv("FBA_HIST_STRIDE")
    comp = fba.Engine("gridworld", **kw)
    assert comp.particle_bytes == full.particle_bytes
    rng = np.random.default_rng(5)
    for eng in (full, comp):
        eng.belief_init()
    for ep in range(2):
        for eng in (full, comp):
            eng.set_position(run=0, episode=ep, t=0)
            eng.belief_reset_domain_state()
        for t in range(18):
            a = int(rng.integers(0, 4))
            for eng in (full, comp):
                eng.set_position(run=0, episode=ep, t=t)
            # an observation the filter gives weight to: the first particle's state seen exactly (observations are indexed like states)
            sf, _, _ = full.belief_get(0, counts=False)
            ob = int(sf[0])
            for eng in (full, comp):
                eng.belief_update(a, ob)
            s0, w0, c0 = full.belief_get(0, weights=True)
            s1, w1, c1 = comp.belief_get(0, weights=True)
            assert np.array_equal(s0, s1) and np.array_equal(w0, w1), (ep, t)
            assert np.array_equal(c0.view(np.uint32), c1.view(np.uint32)), (ep, t)
            i = int(rng.integers(0, particles))
            p0, p1 = full.belief_get_particle(i, 0), comp.belief_get_particle(i, 0)
            assert p0[0] == p1[0] and np.array_equal(p0[2].view(np.uint32), p1[2].view(np.uint32)), (ep, t, i)
    full.close()
    comp.close()


@pytest.mark.parametrize("particles", [96, 4096])
def test_every_switch_of_the_history_particle_path_gives_the_default_results(particles, monkeypatch):
    """The A/B switches the profiles quote (read when a context is created): the prior's rows from L2 instead of LDS, in the search and in the update
    pass (FBA_HIST_ROWS=hbm); the update by one workgroup per slot or by several, forced either way (FBA_HIST_MULTI=0 / 1); records always the full
    stride apart (FBA_HIST_STRIDE=full); node records + hash table instead of the bucket table (FBA_HIST_TREE=records); double-buffered records
    (FBA_DOUBLE_BUFFER=1); every tree of a wave on its own instead of in lock-step (FBA_HIST_LOCKSTEP=0).  Every trace field -- the per-step checksum over every particle's state, weight and counts among them -- and every
    statistic equal the default build's."""
    kw = dict(model=N.MODEL_BA_FACTORED, belief="importance_sampling", size=3, sims=24, particles=particles, structure_prior=2, horizon=9,
              episodes=2, runs=3, slots=3, seed=4242, trace=1)

    def run():
        eng = fba.Engine("gridworld", **kw)
        stats = eng.run_bapomdp()
        out = (eng.trace(), [(s.count, s.mean, s.m2) for s in stats])
        eng.close()
        return out

    t0, s0 = run()
    assert len(t0) > 0
    for name, value in (("FBA_HIST_ROWS", "hbm"), ("FBA_HIST_MULTI", "0"), ("FBA_HIST_MULTI", "1"), ("FBA_HIST_STRIDE", "full"),
                        ("FBA_HIST_TREE", "records"), ("FBA_DOUBLE_BUFFER", "1"), ("FBA_HIST_LOCKSTEP", "0")):
        monkeypatch.setenv(name, value)
        t1, s1 = run()
        monkeypatch.delenv(name)
        assert len(t1) == len(t0) and s1 == s0, (name, value)
        for field in t0.dtype.names:
            assert np.array_equal(t0[field], t1[field]), (name, value, field)


def test_a_bucket_tree_that_is_too_small_stops_the_experiment_loudly():
    """fba_config.tree_buckets below what a search needs: FBA_ESTATE with the knob's name, never a wrong action."""
    eng = fba.Engine("gridworld", model=N.MODEL_BA_FACTORED, belief="importance_sampling", size=3, sims=512, particles=64, structure_prior=2,
                     horizon=7, episodes=1, runs=4, slots=4, seed=5, tree_buckets=8)
    with pytest.raises(fba.FbaError, match="tree_buckets"):
        eng.run_bapomdp()
    with pytest.raises(ValueError, match="tree_buckets"):
        fba.Engine("gridworld", model=N.MODEL_BA_FACTORED, belief="importance_sampling", size=3, sims=512, particles=64, structure_prior=2,
                   horizon=7, episodes=1, runs=4, slots=4, seed=5, tree_buckets=-3)


def test_budgeted_throughput_driver_makes_the_steps_it_is_asked_for():
    """fba_run_ticks with a search budget: launches until the slots have together made ticks x slots real steps; every record it
    traces is a record the lock-step driver traces too."""
    kw = dict(model=N.MODEL_BA_FACTORED, belief="importance_sampling", size=3, sims=128, particles=48, structure_prior=2, horizon=8,
              episodes=2, runs=1 << 20, slots=24, seed=909, trace=1)
    lock = fba.Engine("gridworld", **kw)
    lock.run_ticks(8)
    ref = {(int(r["run"]), int(r["episode"]), int(r["t"])): r for r in lock.trace()}
    eng = fba.Engine("gridworld", search_budget=200, **kw)
    eng.run_ticks(3)
    c = eng.counters()
    assert c.env_steps >= 3 * 24
    tr = eng.trace()
    assert len(tr) == c.env_steps
    seen = 0
    for r in tr:
        key = (int(r["run"]), int(r["episode"]), int(r["t"]))
        if key in ref:     # (a slot of the budgeted engine may be a step or two ahead of eight lock-step ticks: those are not compared)
            seen += 1
            for name in tr.dtype.names:
                assert np.array_equal(r[name], ref[key][name]), (key, name)
    assert seen >= 3 * 24 - 24


def test_an_experiment_after_run_ticks_on_a_budgeted_context_starts_fresh_searches():
    """A budgeted context that was driven by fba_run_ticks holds parked searches (s_sim > 0 for most slots).  An experiment started
    on it afterwards re-positions every slot, so its first launch must START searches; resuming the parked ones would apply the old
    run's root statistics, node count and hash epoch to the new run.  Same for the per-step interface: fba_select_action returns whole
    searches and leaves nothing parked.  Compared with a context that never ran anything else."""
    kw = dict(model=N.MODEL_BA_FACTORED, belief="importance_sampling", size=3, sims=96, particles=64, structure_prior=2, horizon=7,
              episodes=2, runs=9, slots=6, seed=4711, trace=1, search_budget=53)
    fresh = fba.Engine("gridworld", **kw)
    want_stats = [(s.count, s.mean, s.m2) for s in fresh.run_bapomdp()]
    want = fresh.trace()
    fresh.close()
    eng = fba.Engine("gridworld", **kw)
    eng.run_ticks(2)                       # leaves searches parked in the middle of their simulations
    got_stats = [(s.count, s.mean, s.m2) for s in eng.run_bapomdp()]
    got = eng.trace()
    assert got_stats == want_stats and len(got) == len(want)
    for name in want.dtype.names:
        assert np.array_equal(got[name], want[name]), name
    # the per-step interface after parked searches: select_action on re-initiated beliefs equals a fresh context's
    eng.run_ticks(1)
    outs = []
    for e in (eng, fba.Engine("gridworld", **kw)):
        e.set_position(run=3, episode=0, t=0)
        e.belief_init()
        e.belief_reset_domain_state()
        a = e.select_action(hist_len=0)
        outs.append((np.array(a), e.last_step_info()))
    assert np.array_equal(outs[0][0], outs[1][0])
    for name in ("n_nodes", "tree_depth", "root_n", "root_q"):
        assert np.array_equal(outs[0][1][name], outs[1][1][name]), name


def test_per_step_interface_matches_oracle_calls():
    """Planner::selectAction / Belief::updateEstimation one call at a time (slots = 1)."""
    kw = dict(particles=128, sims=300)
    eng = fba.Engine("episodic-tiger", model=N.MODEL_BA_TABLE, seed=41, slots=1, **kw)
    o = orc.Oracle(domain=orc.DOM_TIGER_EPISODIC, model=orc.MODEL_BA_TABLE, rng_mode=orc.RNG_PHILOX,
                   arith=orc.ARITH_DEV, philox_seed=41, **kw)
    L = orc.lib()
    L.orc_rng_episode(o.rng, 0, 0, 0)
    o.belief_initiate()
    eng.belief_init()
    o.belief_reset_domain_state()
    eng.belief_reset_domain_state()
    s, _, cnt = eng.belief_get(0)
    os_, _, ocnt = o.belief_get()
    assert np.array_equal(s, os_) and np.array_equal(cnt, ocnt)
    for t, ob in enumerate([0, 0, 1, 0]):
        L.orc_rng_episode(o.rng, 0, 0, t)
        eng.set_position(run=0, episode=0, t=t)
        a_ref, rec = o.select_action(t)
        a = eng.select_action(hist_len=t)[0]
        info = eng.last_step_info()[0]
        assert a == a_ref
        assert np.array_equal(info["root_n"], rec["root_n"]) and np.array_equal(info["root_q"], rec["root_q"])
        assert info["n_nodes"] == rec["n_nodes"] and info["tree_depth"] == rec["tree_depth"]
        o.belief_update(2, ob)           # listen, hear `ob`
        eng.belief_update(2, ob)
        s, _, cnt = eng.belief_get(0)
        os_, _, ocnt = o.belief_get()
        assert np.array_equal(s, os_) and np.array_equal(cnt, ocnt)
        assert eng.last_step_info()[0]["update_count"] == L.orc_last_update_count(o.h)


def test_rejection_update_with_an_impossible_observation_fails_instead_of_spinning():
    """beliefs::rejectSample loops until N particles reproduce the observation (RejectionSampling.hpp:26-72);
    when none can, the reference never returns.  The device gives up after 2^28 attempts and the call fails."""
    eng = fba.Engine("episodic-tiger", model=N.MODEL_BA_TABLE, particles=4, sims=4, slots=2, seed=3)
    prior = eng.prior()
    prior[12 + 2 * 4 + 0 * 2 + 1] = 0      # psi(listen, left, hear right) = 0
    prior[12 + 2 * 4 + 1 * 2 + 1] = 0      # psi(listen, right, hear right) = 0: "hear right" cannot happen
    eng.set_model_tabular(prior[:12], prior[12:])
    eng.belief_init()
    eng.belief_reset_domain_state()
    with pytest.raises(fba.FbaError, match="accepted fewer than 4 particles"):
        eng.belief_update(2, 1, active=[0, 1])
    eng.belief_update(2, 0, active=[1, 1])   # the ctx stays usable


def test_packed_tiger_rejection_beyond_the_lds_filter_size():
    """Packed tiger filters of more than 4096 particles do not fit reject_tiger_lds_kernel's LDS copy and take
    reject_kernel<packed>; the two agree with the oracle on either side of the limit."""
    for particles in (4096, 4100):
        eng, o = _pair("continuous-tiger", N.MODEL_BA_TABLE, "rejection_sampling", 23, particles=particles, sims=48, runs=2,
                       episodes=2, horizon=4, slots=2)
        assert eng.particle_bytes == 64
        _assert_same_experiment(eng, o, ba=True)


def test_search_tree_node_bound_and_its_guard(monkeypatch):
    """Episodic tiger trees are at most the complete binary tree of the search depth (only `listen` continues, two
    observations), so a slot owns 2^(depth+1) node records instead of sims + 2.  Results are unchanged (every other
    tiger test runs with the bound); a bound that is too small is caught by the kernel, not written past."""
    eng, o = _pair("episodic-tiger", N.MODEL_POMDP, "rejection_sampling", 41, particles=64, sims=3000, runs=4, horizon=6, max_depth=3, slots=4)
    _assert_same_experiment(eng, o, ba=False)          # 3000 simulations in 17 node records
    assert int(eng.trace()["n_nodes"].max()) <= 15
    monkeypatch.setenv("FBA_NODE_BOUND", "5")
    small = fba.Engine("episodic-tiger", model=N.MODEL_BA_TABLE, particles=64, sims=400, runs=4, slots=4, seed=2)
    monkeypatch.delenv("FBA_NODE_BOUND")
    with pytest.raises(fba.FbaError, match="more than the 5 node records"):
        small.run_bapomdp()


def test_a_failed_allocation_leaves_no_error_behind():
    """fba_create that runs out of HBM reports it and cleans up; the next context must not trip over the stale
    HIP error (bench.py steps down to fewer slots exactly this way)."""
    with pytest.raises(fba.FbaError, match="out of memory"):
        fba.Engine("episodic-tiger", model=N.MODEL_BA_TABLE, particles=4096, sims=4096, slots=600000)
    eng, o = _pair("episodic-tiger", N.MODEL_BA_TABLE, "rejection_sampling", 5, particles=64, sims=64, runs=4, episodes=2)
    _assert_same_experiment(eng, o, ba=True)


def test_belief_update_rejects_out_of_range_arguments():
    eng = fba.Engine("episodic-tiger", particles=8, sims=4, slots=2)
    eng.belief_init()
    with pytest.raises(ValueError, match="action"):
        eng.belief_update([0, 3], [0, 0])
    with pytest.raises(ValueError, match="observation"):
        eng.belief_update([0, 1], [0, 2])
    eng2 = fba.Engine("episodic-tiger", particles=8, sims=4, slots=1)
    with pytest.raises(fba.FbaError, match="not initiated"):
        eng2.select_action()


def test_device_order_sums_equal_reference_order_on_ancestors():
    """The importance filter sums weights in a fixed parallel order.  The reference
    sums sequentially; the two may differ in the last bit of a total, which can only change an
    ancestor index when a threshold falls within one ulp of a prefix sum.  On these seeded runs
    the ancestors (hence every later state) must be identical."""
    kw = dict(domain=orc.DOM_TIGER_CONTINUOUS, model=orc.MODEL_BA_TABLE, belief=orc.BELIEF_IMPORTANCE,
              rng_mode=orc.RNG_PHILOX, philox_seed=55, particles=500, sims=64, runs=4, episodes=3, horizon=6, trace=1)
    a = orc.Oracle(arith=orc.ARITH_DEV, **kw)
    b = orc.Oracle(arith=orc.ARITH_REF, **kw)
    _, ra = a.run_bapomdp()
    _, rb = b.run_bapomdp()
    ta, tb = a.trace(ra.n_trace), b.trace(rb.n_trace)
    for name in ("action", "state", "obs", "n_nodes", "root_n"):
        assert np.array_equal(ta[name], tb[name])
    assert np.allclose(ta["weight_total"], tb["weight_total"], rtol=1e-14, atol=0)


def test_multi_workgroup_importance_filter_equals_single_workgroup(monkeypatch):
    """The large-filter path (N > 65536: update, scan and resample as separate launches over the
    whole chip) computes the same device-order sums as the one-workgroup kernel: forced on at
    N = 700 it must reproduce the oracle bit for bit."""
    monkeypatch.setenv("FBA_IS_MULTI_MIN", "1")
    eng, o = _pair("continuous-tiger", N.MODEL_BA_TABLE, "importance_sampling", 131, particles=700, sims=64, runs=6,
                   episodes=3, horizon=6)
    _assert_same_experiment(eng, o, ba=True)
    eng, o = _pair("random-collision-avoidance", N.MODEL_BA_FACTORED, "importance_sampling", 132, size=2, width=5,
                   height=5, particles=333, sims=40, runs=4, episodes=2)
    _assert_same_experiment(eng, o, ba=True)


def test_large_importance_filter_properties(monkeypatch):
    """200 000 particles in ONE belief (collision avoidance, correct-graph prior): the resampled set
    must consist of updated copies of the old particles (count sums grow by FS + FO per update), all
    weights 1/N, total weight in (0, 1]."""
    Np = 200000
    eng = fba.Engine("random-collision-avoidance", model=N.MODEL_BA_FACTORED, belief="importance_sampling", size=2,
                     width=7, height=7, particles=Np, sims=4, slots=1, seed=133)
    eng.belief_init()
    eng.belief_reset_domain_state()
    prior = eng.prior()
    for k, (a, ob) in enumerate([(1, 3 * 7 + 3), (2, 3 * 7 + 4)]):
        eng.set_position(t=k)
        eng.belief_update(a, ob)
        s, w, cnt = eng.belief_get(0)
        assert np.all(w == 1.0 / Np)
        assert np.all((s >= 0) & (s < eng.S))
        d = cnt - prior
        assert np.all(d >= 0) and np.all(d.sum(axis=1) == (k + 1) * 6)     # FS + FO = 4 + 2 increments per update
        tot = eng.last_step_info()[0]["weight_total"]
        assert 0 < tot <= 1.0


@pytest.mark.parametrize("domain,model,belief,kw", [
    ("episodic-tiger", N.MODEL_BA_TABLE, "rejection_sampling", dict(particles=60, sims=80, runs=5, episodes=3)),
    ("continuous-tiger", N.MODEL_BA_TABLE, "importance_sampling", dict(particles=60, sims=60, runs=4, episodes=2, horizon=5)),
    ("episodic-factored-tiger", N.MODEL_BA_FACTORED, "importance_sampling",
     dict(size=2, particles=50, sims=60, runs=4, episodes=2, structure_prior=2)),
])
def test_regular_dirichlet_mode(domain, model, belief, kw):
    """--dirichlet_sampling_method regular on the device: ziggurat half-normal, Marsaglia-Tsang gamma,
    sampleFromSampledMult / sampleMult with the deterministic log / exp, bit-equal to the oracle."""
    eng, o = _pair(domain, model, belief, 141, dirichlet_regular=1, **kw)
    _assert_same_experiment(eng, o, ba=True)


def test_regular_mode_rejects_long_rows():
    with pytest.raises(ValueError, match="rows of at most"):
        fba.Engine("episodic-factored-tiger", model=N.MODEL_BA_TABLE, size=4, particles=4, sims=4, slots=1, dirichlet_regular=1)


def test_set_model_factored_replaces_the_prior_every_particle_starts_from():
    """fba_set_model_factored: a reference-side FBAPOMDPPrior's CPTs injected through the C-ABI, in the
    layout fba_get_factored_layout describes.  Factored tiger, fixed structure: after the call a freshly initiated
    belief holds exactly the injected counts, and a search + update from there equals the oracle's on particles
    that were given the same counts."""
    kw = dict(size=2, particles=40, sims=60, structure_prior=0, horizon=6)
    eng = fba.Engine("episodic-factored-tiger", model=N.MODEL_BA_FACTORED, belief="importance_sampling", seed=57, slots=1, **kw)
    o = orc.Oracle(domain=orc.DOM_FTIGER_EPISODIC, model=orc.MODEL_BA_FACTORED, belief=orc.BELIEF_IMPORTANCE, rng_mode=orc.RNG_PHILOX,
                   arith=orc.ARITH_DEV, philox_seed=57, **kw)
    lay = eng.factored_layout()
    prior = eng.prior()
    mine = prior.copy()
    rng = np.random.default_rng(3)
    mine[:lay.n_counts] += rng.integers(0, 40, lay.n_counts).astype(np.float32) * (prior[:lay.n_counts] > 0)
    eng.set_model_factored(mine)
    with pytest.raises(fba.FbaError, match="not initiated"):
        eng.select_action(0)
    assert np.array_equal(eng.prior().view(np.uint32), mine.view(np.uint32))
    L = orc.lib()
    L.orc_rng_episode(o.rng, 0, 0, 0)
    o.belief_initiate()
    eng.belief_init()
    s0, _, _ = o.belief_get()
    o.belief_set(s=s0, cnt=np.tile(mine, (kw["particles"], 1)))
    s, _, cnt = eng.belief_get(0)
    assert np.array_equal(s, s0)
    assert np.array_equal(cnt.view(np.uint32), np.tile(mine, (kw["particles"], 1)).view(np.uint32))
    o.belief_reset_domain_state()
    eng.belief_reset_domain_state()
    for t, ob in enumerate([0, 1]):
        L.orc_rng_episode(o.rng, 0, 0, t)
        eng.set_position(run=0, episode=0, t=t)
        a_ref, rec = o.select_action(t)
        a = eng.select_action(hist_len=t)[0]
        info = eng.last_step_info()[0]
        assert a == a_ref and np.array_equal(info["root_q"], rec["root_q"]) and np.array_equal(info["root_n"], rec["root_n"])
        o.belief_update(2, ob)
        eng.belief_update(2, ob)
        s, _, cnt = eng.belief_get(0)
        os_, _, ocnt = o.belief_get()
        assert np.array_equal(s, os_) and np.array_equal(cnt.view(np.uint32), ocnt.view(np.uint32))
    # a layout that is not the engine's is refused
    lay.node[0].offset += 2
    with pytest.raises(ValueError, match="fba_get_factored_layout"):
        eng.set_model_factored(mine, layout=lay)


def test_log_bd_score_and_lgamma_equal_the_oracle():
    """DBNNode::LogBDScore / BABNModel::LogBDScore on the device (fba_log_bd_score, det_lgamma) against the oracle in
    device arithmetic, bit for bit -- the oracle's reference-arithmetic twin is pinned against the real
    DBNNode.cpp through oracle/_ref (test_fbapomdp_state_matches_reference).  The first building block of the
    MH structure beliefs."""
    eng = fba.Engine("episodic-factored-tiger", model=N.MODEL_BA_FACTORED, size=2, structure_prior=3, particles=8, sims=4, slots=1)
    o = orc.Oracle(domain=orc.DOM_FTIGER_EPISODIC, size=2, model=orc.MODEL_BA_FACTORED, structure_prior=orc.SP_FULLY_CONNECTED,
                   arith=orc.ARITH_DEV, rng_mode=orc.RNG_PHILOX, philox_seed=9)
    L = orc.lib()
    xs = np.concatenate([np.linspace(1, 40, 3000), np.logspace(0, 7, 3000)])
    out = np.zeros_like(xs)
    eng._chk(eng.L.fba_selftest_lgamma(eng.h, xs.ctypes.data, len(xs), out.ctypes.data))
    assert np.array_equal(out, np.array([L.orc_det_lgamma(float(x)) for x in xs]))
    prior = o.prior_counts()
    o.ftiger_set_structure(prior, 0b101)
    assert np.array_equal(eng.prior()[:8], prior[:8])
    rng = np.random.default_rng(5)
    for trial in range(6):
        cnt = prior.copy()
        s = 3
        L.orc_rng_episode(o.rng, trial, 0, 0)
        for i in range(40 * (trial + 1)):
            s, _, _, _ = o.model_step(cnt, s, int(rng.integers(0, 3)), update=True)
        got = C.c_double()
        eng._chk(eng.L.fba_log_bd_score(eng.h, cnt.ctypes.data, prior.ctypes.data, C.byref(got)))
        assert got.value == L.orc_log_bd_score(o.h, cnt.ctypes.data, prior.ctypes.data) != 0.0


@pytest.mark.parametrize("size,sp,noise", [(1, 0, 0.0), (2, 2, 0.1), (3, 1, 0.0), (3, 3, -0.1)])
def test_packed_factored_tiger_particles_equal_dense_ones(size, sp, noise, monkeypatch):
    """Factored-tiger particles stored as uint16 increments over a prior that is a function of the cell and the
    particle's structure bits (PackedFtigerView: 144 B instead of 288 B at --size 3) against fp32 count tables
    (FBA_DENSE_PARTICLES=1): same experiment, same trace (checksum over every count included), same particles through
    the API; both equal the oracle through every other factored-tiger test."""
    kw = dict(model=N.MODEL_BA_FACTORED, belief="rejection_sampling", size=size, structure_prior=sp, noise=noise, particles=200, sims=150,
              runs=8, episodes=4, slots=4, seed=311 + size, trace=1)
    out = []
    for dense in (False, True):
        if dense:
            monkeypatch.setenv("FBA_DENSE_PARTICLES", "1")
        eng = fba.Engine("episodic-factored-tiger", **kw)
        monkeypatch.delenv("FBA_DENSE_PARTICLES", raising=False)
        FS = size + 1
        nc = 8 * FS + 4 + (2 << FS)
        assert eng.ncnt == nc + 1
        if not dense and noise >= 0:   # (a prior value that is not exact under + 65535 keeps the records dense: -0.1 here)
            assert eng.particle_bytes == 4 * ((nc // 2 + 2 + 3) // 4 * 4)   # increments, structure word, state
        stats = eng.run_bapomdp()
        out.append(([(s.count, s.mean, s.m2) for s in stats], eng.trace(), [eng.belief_get(k) for k in range(4)], eng))
    assert out[0][3].particle_bytes <= out[1][3].particle_bytes
    assert out[0][0] == out[1][0]
    assert out[0][1].tobytes() == out[1][1].tobytes()
    for a, b in zip(out[0][2], out[1][2]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
    packed, dense = out[0][3], out[1][3]
    s, _, cnt = dense.belief_get(2)
    packed.belief_set(0, state=s, counts=cnt)
    s2, _, cnt2 = packed.belief_get(0)
    assert np.array_equal(s, s2) and np.array_equal(cnt.view(np.uint32), cnt2.view(np.uint32))
    if packed.particle_bytes < dense.particle_bytes:
        bad = cnt.copy()
        bad[:, 0] += 0.5
        with pytest.raises(ValueError, match="plus 0..65535 increments"):
            packed.belief_set(0, counts=bad)


@pytest.mark.parametrize("belief,domain,option,sp,thr", [
    ("mh-within-gibbs", "continuous-factored-tiger", 0, 2, -1.0), ("mh-within-gibbs", "continuous-factored-tiger", 1, 2, -1.0),
    ("mh-within-gibbs", "episodic-factored-tiger", 0, 1, -0.5), ("mh-within-gibbs", "continuous-factored-tiger", 0, 0, -3.0),
    ("mh-nips", "continuous-factored-tiger", 0, 2, -1.0), ("mh-nips", "episodic-factored-tiger", 0, 1, -0.5),
    ("mh-nips", "continuous-factored-tiger", 0, 0, -3.0),
    ("mh-within-gibbs", "random-collision-avoidance", 0, 2, -2.0), ("mh-within-gibbs", "centered-collision-avoidance", 1, 1, -2.0),
    ("mh-nips", "random-collision-avoidance", 0, 2, -2.0), ("mh-nips", "centered-collision-avoidance", 0, 0, -1.0),
    ("mh-within-gibbs", "gridworld", 0, 2, -3.0), ("mh-within-gibbs", "gridworld", 1, 0, -3.0), ("mh-nips", "gridworld", 0, 2, -3.0),
    ("mh-within-gibbs", "linear-sysadmin", 0, 0, -1.0), ("mh-within-gibbs", "independent-sysadmin", 1, 0, -1.0),
])
def test_fbapomdp_mh_beliefs(belief, domain, option, sp, thr):
    """-B mh-within-gibbs (MHwithinGibbs.cpp; --belief-option "" = message passing, "rs" = rejection-sampled state
    histories) and -B mh-nips (MHNIPS2018.cpp): importance filter + the run's history + a Metropolis-Hastings re-draw of
    the filter over structures, scored by LogBDScore, when the log likelihood falls below --threshold.  Factored tiger
    (one structure word: the listen node's parents) and collision avoidance (one word per action and obstacle).  Whole
    experiments, every trace field (the checksum over every particle's counts after every update included) against the oracle."""
    ca = "collision" in domain
    kw = dict(width=3, height=3, size=1, particles=24, sims=48, horizon=5) if ca else dict(size=2, particles=48, sims=80, horizon=8)
    if "sysadmin" in domain:
        kw = dict(size=3, particles=24, sims=48, horizon=6)
    if domain == "gridworld":
        # (histories replayed by forward sampling -- the "rs" option and mh-nips's computePosterior -- have to reproduce every
        # observation of an episode, one of N * N * G values per step: two-step episodes there, as in the reference's own use)
        kw = dict(size=3, particles=64, sims=48, horizon=6 if (belief == "mh-within-gibbs" and option == 0) else 2)
    eng, o = _pair(domain, N.MODEL_BA_FACTORED, belief, 601 + option, runs=4, episodes=3, structure_prior=sp, threshold=thr,
                   belief_option=option, **kw)
    _assert_same_experiment(eng, o, ba=True)
    s, w, cnt = eng.belief_get(0)
    assert np.all(w == 1.0 / kw["particles"])
    # the chain has really run: with a threshold that is never reached the same experiment leaves other filters behind
    never = fba.Engine(domain, model=N.MODEL_BA_FACTORED, belief=belief, seed=601 + option, slots=4, trace=1, runs=4, episodes=3,
                       structure_prior=sp, threshold=-1e9, belief_option=option, **kw)
    never.run_bapomdp()
    assert not np.array_equal(never.trace()["belief_hash"], eng.trace()["belief_hash"])


def test_mh_beliefs_refuse_what_they_are_not_built_for():
    with pytest.raises(ValueError, match="MHwithinGibbs::cannot initiate with threshold >= 0"):
        fba.Engine("episodic-factored-tiger", model=N.MODEL_BA_FACTORED, belief="mh-within-gibbs", size=1, particles=8, sims=8, threshold=0.5)
    with pytest.raises(ValueError, match="MHNIPS2018::cannot initiate with threshold >= 0"):
        fba.Engine("episodic-factored-tiger", model=N.MODEL_BA_FACTORED, belief="mh-nips", size=1, particles=8, sims=8, threshold=0.0)
    with pytest.raises(ValueError, match="needs a factored model"):
        fba.Engine("linear-sysadmin", model=N.MODEL_BA_TABLE, belief="mh-within-gibbs", size=3, particles=8, sims=8, threshold=-1.0)
    with pytest.raises(ValueError, match="mh-nips belief on sysadmin"):      # MHNIPS2018::MH never accepts a proposal there
        fba.Engine("independent-sysadmin", model=N.MODEL_BA_FACTORED, belief="mh-nips", size=3, particles=8, sims=8, threshold=-1.0)
    with pytest.raises(ValueError, match="expected Dirichlet mode"):
        fba.Engine("gridworld", model=N.MODEL_BA_FACTORED, belief="mh-nips", size=3, particles=8, sims=8, threshold=-1.0, dirichlet_regular=1)


@pytest.mark.parametrize("domain,model,kw", [
    ("episodic-tiger", N.MODEL_BA_TABLE, dict(particles=12, sims=96, horizon=8)),
    ("continuous-tiger", N.MODEL_BA_TABLE, dict(particles=5, sims=64, horizon=6, dirichlet_regular=1)),
    ("episodic-factored-tiger", N.MODEL_BA_FACTORED, dict(size=2, particles=9, sims=80, horizon=8, structure_prior=2)),
    ("gridworld", N.MODEL_BA_TABLE, dict(size=3, particles=7, sims=48, horizon=6)),
    ("gridworld", N.MODEL_BA_FACTORED, dict(size=3, particles=6, sims=48, horizon=6, structure_prior=2)),
    ("random-collision-avoidance", N.MODEL_BA_FACTORED, dict(width=3, height=3, size=1, particles=6, sims=48, horizon=5, structure_prior=1)),
    ("independent-sysadmin", N.MODEL_BA_TABLE, dict(size=2, particles=8, sims=48, horizon=6, planner="ts")),
    ("linear-sysadmin", N.MODEL_BA_FACTORED, dict(size=3, particles=4, sims=32, horizon=5, planner="random")),
])
def test_nested_belief(domain, model, kw):
    """-B nested (NestedBelief.cpp): `particles` count particles by weight, each with its own flat filter of particles^2
    domain states; the update is a rejection sampler per count particle whose accepted samples add 1 / particles^2 to the
    counts the next attempt samples from, the weight is multiplied by 1 / attempts, the top filter is never resampled.
    Whole experiments: every trace field (attempt totals, the weight total before normalisation, the checksum over every
    count particle's counts and weight and every domain state of every flat filter) against the oracle, then the filters
    themselves."""
    eng, o = _pair(domain, model, "nested", 811, runs=3, episodes=3, **kw)
    _assert_same_experiment(eng, o, ba=True)
    s, w, cnt = eng.belief_get(0)
    os_, ow, ocnt = o.belief_get()       # (the oracle holds the last run: slot = run here, so compare with the last slot)
    s, w, cnt = eng.belief_get(2)
    assert np.array_equal(w, ow) and np.array_equal(cnt.view(np.uint32), ocnt.view(np.uint32)) and not s.any()
    assert np.array_equal(eng.belief_get_nested(2), o.belief_get_nested())
    assert abs(w.sum() - 1) < 1e-12 and len(set(w.tolist())) > 1      # normalised, and no longer uniform


def test_nested_belief_refusals():
    with pytest.raises(ValueError, match="Bayes-adaptive"):
        fba.Engine("episodic-tiger", model=N.MODEL_POMDP, belief="nested", particles=4, sims=8)
    with pytest.raises(ValueError, match="at most 256"):
        fba.Engine("episodic-tiger", model=N.MODEL_BA_TABLE, belief="nested", particles=300, sims=8)
    eng = fba.Engine("episodic-tiger", model=N.MODEL_BA_TABLE, belief="nested", particles=4, sims=8)
    eng.belief_init()
    with pytest.raises(ValueError, match="cannot be set from the host"):
        eng.belief_set(0, state=np.zeros(4, np.int32))


@pytest.mark.parametrize("domain,kw", [
    ("episodic-factored-tiger", dict(size=2, structure_prior=2, particles=48, sims=80, horizon=8, resample_amount=6, threshold=0.5)),
    ("continuous-factored-tiger", dict(size=3, structure_prior=1, particles=33, sims=64, horizon=8, resample_amount=1, threshold=1.0)),
    ("random-collision-avoidance", dict(width=4, height=3, size=2, structure_prior=1, particles=24, sims=48, horizon=5, resample_amount=5, threshold=0.2)),
    ("linear-sysadmin", dict(size=3, particles=20, sims=40, horizon=6, resample_amount=19, threshold=0.06)),
])
def test_fbapomdp_incubator_belief(domain, kw):
    """-B incubator (StructureIncubatorSampling.cpp): the reinvigoration belief's two rejection filters plus a weighted
    shadow filter of bred particles -- the --resample-amount least likely (WeightedFilter::leastLikely, a
    std::priority_queue) are bred anew before every update, the filter is importance-sampled and resampled.  Whole
    experiments against the oracle (the trace carries the main filter's checksum and rejection count and the shadow
    filter's weight total), then the three filters themselves."""
    eng, o = _pair(domain, N.MODEL_BA_FACTORED, "incubator", 907, runs=3, episodes=3, **kw)
    _assert_same_experiment(eng, o, ba=True)
    assert np.any(eng.trace()["weight_total"] > 0)
    s, w, cnt = eng.belief_get_shadow(2)
    os_, ow, ocnt = o.belief_get_shadow()
    assert np.array_equal(s, os_) and np.array_equal(w, ow) and np.array_equal(cnt.view(np.uint32), ocnt.view(np.uint32))
    fs, fcnt = eng.belief_get_fully_connected(2)
    ofs, ofcnt = o.belief_get_fc()
    assert np.array_equal(fs, ofs) and np.array_equal(fcnt.view(np.uint32), ofcnt.view(np.uint32))
    ms, _, mcnt = eng.belief_get(2)
    oms, _, omcnt = o.belief_get()
    assert np.array_equal(ms, oms) and np.array_equal(mcnt.view(np.uint32), omcnt.view(np.uint32))


def test_incubator_belief_refusals():
    kw = dict(model=N.MODEL_BA_FACTORED, belief="incubator", size=2, sims=8)
    with pytest.raises(ValueError, match="must initiate with 1 < threshold <= 0"):
        fba.Engine("episodic-factored-tiger", particles=8, resample_amount=2, threshold=0.0, **kw)
    with pytest.raises(ValueError, match="is promoted at once"):       # 1/8 > 0.1: the reference would divide by a zero total weight
        fba.Engine("episodic-factored-tiger", particles=8, resample_amount=2, threshold=0.1, **kw)
    with pytest.raises(ValueError, match="leastLikely"):
        fba.Engine("episodic-factored-tiger", particles=8, resample_amount=8, threshold=0.5, **kw)
    with pytest.raises(ValueError, match="factored tiger, collision avoidance or sysadmin"):
        fba.Engine("gridworld", model=N.MODEL_BA_FACTORED, belief="incubator", size=3, sims=8, particles=8, resample_amount=2, threshold=0.5)


@pytest.mark.parametrize("env", [{"FBA_NODE_VISITS": "1"}, {"FBA_NODE_WORDS": "16"}, {"FBA_DOUBLE_BUFFER": "1"}, {"FBA_SCRATCH_SLOTS": "2"}])
def test_layout_knobs_do_not_change_results(env, monkeypatch):
    """The memory-layout choices C4 runs on -- tree nodes without a visits word (48 bytes), one record buffer per slot for
    history particles -- against their alternatives (the knobs the same-box A/B runs of DESIGN section 5a used): the same
    experiment, every trace field equal to the oracle's either way."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    # (five slots: with a scratch pool of two, the resample and the reset work through three chunks, the last one partial)
    eng, o = _pair("gridworld", N.MODEL_BA_FACTORED, "importance_sampling", 977, size=4, particles=130, sims=96, runs=5, episodes=2,
                   horizon=7, structure_prior=2)
    _assert_same_experiment(eng, o, ba=True)
