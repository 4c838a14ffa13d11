"""The drop-in boundary on a real GPU: the reference's own episode loop (episode::run), its own Tiger environment and
mt19937 drive the HIP engine through the C++ adapters (fba_pomdp_amd/csrc/host/adapters.hpp) -- the binary
oracle/_ref/adapters_drive_gpu, linked from the reference's objects where /root/reference exists (`make -C oracle ref`,
part of __graft_entry__.build()) and shipped to the GPU box as a built artefact.  The (action, observation) stream it
logs is replayed through the oracle at the stream positions the adapters must have used -- run per Belief::initiate,
episode per resetDomainStateDistribution, t = History::length() -- and every action has to come out the same."""
import os
import subprocess

import pytest

from oracle import pyorc as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "adapters_drive_gpu")

pytestmark = pytest.mark.gpu


def _parse(text):
    runs = []
    for line in text.splitlines():
        w = line.split()
        if w[0] == "run":
            runs.append([])
        elif w[0] == "episode":
            runs[-1].append({"steps": []})
        elif w[0] == "step":
            runs[-1][-1]["steps"].append(tuple(int(float(x.split("=")[1])) for x in w[1:]))
        elif w[0] == "return":
            runs[-1][-1]["ret"], runs[-1][-1]["len"] = float(w[1]), int(w[3])
    return runs


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/adapters_drive_gpu is built only where /root/reference exists")
@pytest.mark.parametrize("mode,sims,particles,runs,episodes,horizon", [("planning", 256, 64, 5, 1, 8), ("bapomdp", 200, 48, 3, 4, 6)])
def test_reference_episode_loop_drives_the_engine_through_the_adapters(mode, sims, particles, runs, episodes, horizon):
    seed = 424242
    r = subprocess.run([EXE, mode, str(sims), str(particles), str(runs), str(episodes), str(horizon), str(seed)], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    log = _parse(r.stdout)
    assert len(log) == runs and all(len(eps) == episodes for eps in log)
    ba = mode == "bapomdp"
    o = orc.Oracle(domain=orc.DOM_TIGER_EPISODIC, model=orc.MODEL_BA_TABLE if ba else orc.MODEL_POMDP, belief=orc.BELIEF_REJECTION,
                   rng_mode=orc.RNG_PHILOX, arith=orc.ARITH_DEV, philox_seed=seed, sims=sims, particles=particles, horizon=horizon,
                   episodes=episodes, runs=runs)
    L = orc.lib()
    n_steps = 0
    for run, eps in enumerate(log):
        L.orc_rng_episode(o.rng, run, 0, 0)
        o.belief_initiate()
        for ep, rec in enumerate(eps):
            if ba:
                L.orc_rng_episode(o.rng, run, ep, 0)
                o.belief_reset_domain_state()
            assert rec["len"] == len(rec["steps"]) <= horizon
            ret = 0.0
            for t, (a, ob, rew, term) in enumerate(rec["steps"]):
                L.orc_rng_episode(o.rng, run, ep, t)
                a_ref, _ = o.select_action(t)
                assert a_ref == a, (mode, run, ep, t)
                if not term:
                    o.belief_update(a, ob)
                ret += rew * 0.95 ** t
                n_steps += 1
            assert abs(ret - rec["ret"]) < 1e-9
    assert n_steps >= runs * episodes        # at least one real step per episode went through the GPU


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/adapters_drive_gpu is built only where /root/reference exists")
@pytest.mark.parametrize("mode,particles", [("sample-planning", 64), ("sample-bapomdp", 48), ("sample-bapomdp-is", 48), ("sample-fbapomdp", 32)])
def test_belief_sample_hands_a_host_planner_real_particles(mode, particles):
    """Belief::sample() of the hip beliefs (Belief.hpp:33) under the reference's own RandomPlanner: every sample is a particle
    of the filter the engine holds at that moment -- its state and, Bayes-adaptive, its learned counts read back through the
    reference's BAPOMDPState / FBAPOMDPState accessors -- drawn with the filter's own distribution, and a planner may swap
    the sample's domain state and put it back (RBAPOUCT.cpp:92-106)."""
    draws = 400
    r = subprocess.run([EXE, mode, str(draws), str(particles), "2", "3", "6", "99"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("samples ")]
    assert len(lines) >= 2 * (1 + (1 if mode == "sample-planning" else 3))
    seen_states = set()
    for l in lines:
        kv = dict(w.split("=") for w in l.split()[1:])
        assert int(kv["matched"]) == draws, l          # every sample IS a particle of the device filter (state and counts)
        if mode != "sample-planning":
            assert int(kv["swapped_back"]) == draws, l
        filt = [float(x) for x in kv["filter"].split(",")]
        smp = [int(x) for x in kv["sampled"].split(",")]
        assert sum(smp) == draws
        for p, k in zip(filt, smp):                    # drawn with the filter's distribution (5 sigma of a binomial)
            assert abs(k / draws - p) <= 5 * (p * (1 - p) / draws) ** 0.5 + 1e-12, l
            if k:
                seen_states.add(p)
    assert len(seen_states) > 1                        # not a constant
