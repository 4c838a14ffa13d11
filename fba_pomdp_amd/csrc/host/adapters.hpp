// adapters.hpp -- C++ classes that plug libfba_hip.so in behind the reference's own interfaces.
//
// Compiled INSIDE the reference tree (it includes the reference's headers); see INTEGRATION.md for
// the factory patch.  One fba::HipSession wraps one fba_ctx with slots = 1, i.e. exactly one
// reference (Planner, Belief) pair; the planner and belief adapters share it:
//
//   HipPOUCT            : Planner            replaces planners::POUCT     (src/planners/mcts/POUCT.hpp)
//   HipRBAPOUCT         : planners::BAPlanner replaces planners::RBAPOUCT (src/planners/bayes-adaptive/RBAPOUCT.hpp)
//   HipParticleBelief   : Belief             replaces beliefs::RejectionSampling / ImportanceSampler
//   HipBAParticleBelief : beliefs::BABelief  replaces beliefs::BARejectionSampling / BAImportanceSampling
//
// Where the experiment is.  The engine addresses its Philox streams by (run, episode, t); the reference's
// loops never tell a planner or a belief those numbers, so the session counts them from the calls it sees,
// in the order the experiments make them (PlanningExperiment.cpp:39-52, BAPOMDPExperiment.cpp:44-75,
// Episode.cpp:39-55):
//   Belief::initiate                          -> next run, no episode yet
//   BABelief::resetDomainStateDistribution    -> next episode of the run (planning has none: episode 0)
//   Planner::selectAction(..., history)       -> t = history.length()
//   Belief::updateEstimation                  -> the step selectAction just planned (same t)
// and sets the position before every call into the library (tests/test_adapters_run.py drives the real
// episode::run over a recording stub of the C-ABI and checks the sequence).
//
// Errors from the C-ABI are re-thrown as std::string, which the reference's main()s catch
// (src/planning.cpp:46-54).  Actions are obtained from the simulator (copyAction of one of its legal
// actions) so the reference's ownership rules (Environment.hpp:16-46, POMDP.hpp:79-80) hold.
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "fba_hip.h"

#include "bayes-adaptive/models/table/BAPOMDP.hpp"
#include "beliefs/Belief.hpp"
#include "beliefs/bayes-adaptive/BABelief.hpp"
#include "domains/POMDP.hpp"
#include "environment/Action.hpp"
#include "environment/History.hpp"
#include "environment/Observation.hpp"
#include "environment/State.hpp"
#include "planners/Planner.hpp"
#include "planners/bayes-adaptive/BAPlanner.hpp"

namespace fba {

class HipSession
{
public:
    explicit HipSession(fba_config cfg)
    {
        cfg.slots = 1;
        if (fba_create(&cfg, &_ctx) != FBA_OK) throw std::string(fba_last_error(nullptr));
        _particles = cfg.particles;
    }
    ~HipSession() { fba_destroy(_ctx); }
    HipSession(HipSession const&) = delete;
    HipSession& operator=(HipSession const&) = delete;

    fba_ctx* ctx() const { return _ctx; }
    int particles() const { return _particles; }
    void check(int rc) const
    {
        if (rc != FBA_OK) throw std::string(fba_last_error(_ctx));
    }
    // ---- the experiment's position, counted from the calls (see the header comment)
    void begin_run()        // Belief::initiate
    {
        ++_run;
        _episode = -1;
        _t       = 0;
        position(_run, 0, 0);
    }
    void begin_episode()    // BABelief::resetDomainStateDistribution
    {
        ++_episode;
        _t = 0;
        position(_run, _episode, 0);
    }
    void at_step(int t)     // Planner::selectAction with a history of length t
    {
        if (_run < 0) _run = 0;  // a planner used without a belief adapter
        _t = t;
        position(_run, _episode < 0 ? 0 : _episode, _t);
    }
    void at_update() { position(_run < 0 ? 0 : _run, _episode < 0 ? 0 : _episode, _t); }  // Belief::updateEstimation
    int run() const { return _run; }
    int episode() const { return _episode < 0 ? 0 : _episode; }
    int t() const { return _t; }

private:
    void position(int run, int episode, int t) const { check(fba_set_position(_ctx, &run, &episode, &t)); }
    fba_ctx* _ctx  = nullptr;
    int _particles = 0;
    int _run = -1, _episode = -1, _t = 0;
};

// Planner::selectAction for both planners: the search runs where the particles live; the action handed out is
// owned by the simulator, as POUCT does (POUCT.cpp:88-90, RBAPOUCT.cpp:112-114)
inline Action const* hip_select_action(HipSession& s, POMDP const& simulator, Belief const& belief, History const& h)
{
    int32_t const hist_len = static_cast<int32_t>(h.length());
    int32_t action         = -1;
    s.at_step(hist_len);
    s.check(fba_select_action(s.ctx(), &hist_len, nullptr, &action));
    std::vector<Action const*> legal;
    simulator.addLegalActions(belief.sample(), &legal);
    Action const* chosen = simulator.copyAction(legal.at(static_cast<size_t>(action)));
    for (auto a : legal) simulator.releaseAction(a);
    return chosen;
}

// Planner::selectAction (src/planners/Planner.hpp:23-24).  The belief argument must be the
// adapter that shares this session: the search reads the particles where they live, in HBM.
class HipPOUCT : public Planner
{
public:
    explicit HipPOUCT(std::shared_ptr<HipSession> s) : _s(std::move(s)) {}

    Action const* selectAction(POMDP const& simulator, Belief const& belief, History const& h) const override
    {
        return hip_select_action(*_s, simulator, belief, h);
    }

private:
    std::shared_ptr<HipSession> _s;
};

class HipRBAPOUCT : public planners::BAPlanner
{
public:
    explicit HipRBAPOUCT(std::shared_ptr<HipSession> s) : _s(std::move(s)) {}

    Action const* selectAction(BAPOMDP const& bapomdp, beliefs::BABelief const& belief, History const& h) const override
    {
        return hip_select_action(*_s, bapomdp, belief, h);
    }

private:
    std::shared_ptr<HipSession> _s;
};

// Belief over plain domain states (planning executable).
// Belief::initiate / free / sample / updateEstimation (src/beliefs/Belief.hpp:25-40).
class HipParticleBelief : public Belief
{
public:
    explicit HipParticleBelief(std::shared_ptr<HipSession> s) : _s(std::move(s)), _host_state(0) {}

    void initiate(POMDP const& /*domain*/) override
    {
        _s->begin_run();
        _s->check(fba_belief_init(_s->ctx()));
        _fresh = false;
    }
    void free(POMDP const& /*domain*/) override {}  // particles live in the ctx
    // A host-side caller asks for one particle (the adapters' own selectAction only needs it to enumerate the legal
    // actions): particle 0 of the (exchangeable) device set, downloaded once per belief state, not per call.
    State const* sample() const override
    {
        if (!_fresh) {
            _states.resize(static_cast<size_t>(_s->particles()));
            _s->check(fba_belief_get(_s->ctx(), 0, _states.data(), nullptr, nullptr));
            _fresh = true;
        }
        _host_state.index(_states[0]);
        return &_host_state;
    }
    void updateEstimation(Action const* a, Observation const* o, POMDP const& /*domain*/) override
    {
        int32_t const ai = a->index(), oi = o->index();
        _s->at_update();
        _s->check(fba_belief_update(_s->ctx(), &ai, &oi, nullptr));
        _fresh = false;
    }

protected:
    std::shared_ptr<HipSession> _s;
    mutable IndexState _host_state;
    mutable std::vector<int32_t> _states;
    mutable bool _fresh = false;
};

// Bayes-adaptive belief (bapomdp / fbapomdp executables): adds
// BABelief::resetDomainStateDistribution (src/beliefs/bayes-adaptive/BABelief.hpp:34).
class HipBAParticleBelief : public beliefs::BABelief
{
public:
    explicit HipBAParticleBelief(std::shared_ptr<HipSession> s) : _s(std::move(s)) {}
    ~HipBAParticleBelief() override
    {
        if (_host && _owner) _owner->releaseState(_host);
    }

    void initiate(POMDP const& domain) override
    {
        _owner = &domain;
        _s->begin_run();
        _s->check(fba_belief_init(_s->ctx()));
    }
    void free(POMDP const& domain) override
    {
        if (_host) domain.releaseState(_host);
        _host = nullptr;
    }
    // Host mirror for code that inspects a particle (e.g. -v 3 logCounts,
    // BAPOMDPExperiment.cpp:48-52): a prior sample whose domain state is particle 0's.
    State const* sample() const override
    {
        if (!_host && _owner) _host = _owner->sampleStartState();
        return _host;
    }
    void updateEstimation(Action const* a, Observation const* o, POMDP const& /*domain*/) override
    {
        int32_t const ai = a->index(), oi = o->index();
        _s->at_update();
        _s->check(fba_belief_update(_s->ctx(), &ai, &oi, nullptr));
    }
    void resetDomainStateDistribution(BAPOMDP const& /*domain*/) override
    {
        _s->begin_episode();
        _s->check(fba_belief_reset_domain_state(_s->ctx()));
    }

private:
    std::shared_ptr<HipSession> _s;
    POMDP const* _owner       = nullptr;
    mutable State const* _host = nullptr;
};

}  // namespace fba
