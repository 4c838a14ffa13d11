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
//   Belief::updateEstimation                  -> the step selectAction just planned (same t); the next step is t + 1 (so a
//                                                reference planner that never tells the session where it is -- RandomPlanner
//                                                over a hip belief -- still advances the streams)
// and sets the position before every call into the library (tests/test_adapters_run.py drives the real
// episode::run over a recording stub of the C-ABI and checks the sequence).
//
// Belief::sample() (Belief.hpp:33).  The adapters' own planners never need it beyond the legal-action probe -- the search
// reads the particles where they live -- but a HOST planner paired with a hip belief does (POUCT.cpp:67-85,
// RBAPOUCT.cpp:74-107), so sample() honours the contract: the host draws the index the way the reference's filter
// would (FlatFilter::sample: uniform; WeightedFilter::sample: by weight, scanning from the last particle), with the
// reference's own rnd:: generator, downloads THAT particle (fba_belief_get_particle) and, for the Bayes-adaptive belief,
// builds the BAPOMDPState / FBAPOMDPState the planner borrows -- its counts are the particle's learned counts, its
// _domain_state is a domain state the planner may swap and must put back (RBAPOUCT.cpp:92-106).  The pointer stays valid
// until the next mutating call (initiate / updateEstimation / resetDomainStateDistribution / free).
//
// Errors from the C-ABI are re-thrown as std::string, which the reference's main()s catch
// (src/planning.cpp:46-54).  Actions are obtained from the simulator (copyAction of one of its legal
// actions) so the reference's ownership rules (Environment.hpp:16-46, POMDP.hpp:79-80) hold.
#pragma once

#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "fba_hip.h"

#include "bayes-adaptive/models/Domain_Size.hpp"
#include "bayes-adaptive/models/factored/Domain_Feature_Size.hpp"
#include "bayes-adaptive/models/table/BAPOMDP.hpp"
#include "bayes-adaptive/states/factored/BABNModel.hpp"
#include "bayes-adaptive/states/factored/FBAPOMDPState.hpp"
#include "bayes-adaptive/states/table/BAFlatModel.hpp"
#include "bayes-adaptive/states/table/BAPOMDPState.hpp"
#include "beliefs/Belief.hpp"
#include "beliefs/bayes-adaptive/BABelief.hpp"
#include "domains/POMDP.hpp"
#include "environment/Action.hpp"
#include "environment/History.hpp"
#include "environment/Observation.hpp"
#include "environment/State.hpp"
#include "planners/Planner.hpp"
#include "planners/bayes-adaptive/BAPlanner.hpp"
#include "utils/index.hpp"
#include "utils/random.hpp"

namespace fba {

class HipSession
{
public:
    explicit HipSession(fba_config cfg)
    {
        if (fba_abi_version() != FBA_ABI_VERSION)   // a library built from another fba_hip.h: fba_config may not have this layout
            throw std::string("libfba_hip.so has ABI version " + std::to_string(fba_abi_version()) + ", the adapters were compiled against " +
                              std::to_string(FBA_ABI_VERSION));
        cfg.slots = 1;
        if (fba_create(&cfg, &_ctx) != FBA_OK) throw std::string(fba_last_error(nullptr));
        _particles = cfg.belief == FBA_BELIEF_POINT ? 1 : cfg.particles;
        _weighted  = cfg.belief != FBA_BELIEF_REJECTION && cfg.belief != FBA_BELIEF_POINT && cfg.belief != FBA_BELIEF_REINVIGORATION &&
                    cfg.belief != FBA_BELIEF_INCUBATOR;
        _model = cfg.model;
    }
    ~HipSession() { fba_destroy(_ctx); }
    HipSession(HipSession const&) = delete;
    HipSession& operator=(HipSession const&) = delete;

    fba_ctx* ctx() const { return _ctx; }
    int particles() const { return _particles; }
    bool weighted() const { return _weighted; }   // the main filter is a WeightedFilter (importance sampling and the beliefs built on it)
    int model() const { return _model; }
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
    void step_done() { ++_t; }
    int run() const { return _run; }
    int episode() const { return _episode < 0 ? 0 : _episode; }
    int t() const { return _t; }

private:
    void position(int run, int episode, int t) const { check(fba_set_position(_ctx, &run, &episode, &t)); }
    fba_ctx* _ctx  = nullptr;
    int _particles = 0, _model = 0;
    bool _weighted = false;
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

// The index Belief::sample() returns the particle of, drawn on the host with the reference's generator:
// FlatFilter<T>::sample (FlatFilter.cpp:97-102) is uniform over the particles; WeightedFilter<T>::sample
// (WeightedFilter.cpp:163-191) draws a threshold in [0, total) and walks down from the last particle.
inline int hip_draw_particle(int n, std::vector<double> const* weights)
{
    if (!weights) return n <= 1 ? 0 : rnd::slowRandomInt(0, n);
    double total = 0;
    for (double w : *weights) total += w;
    double const threshold = rnd::uniform_rand01() * total;
    double remaining       = total;
    int i                  = n - 1;
    for (; i > 0; --i) {
        remaining -= (*weights)[static_cast<size_t>(i)];
        if (threshold > remaining) break;
    }
    return i;
}

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
    // One particle of the device filter, drawn as the reference's filter draws it; the states (4 B each, and the weights
    // of a weighted filter) are downloaded once per belief state, not per call.
    State const* sample() const override
    {
        if (!_fresh) {
            _states.resize(static_cast<size_t>(_s->particles()));
            _weights.resize(_s->weighted() ? _states.size() : 0);
            _s->check(fba_belief_get(_s->ctx(), 0, _states.data(), _s->weighted() ? _weights.data() : nullptr, nullptr));
            _fresh = true;
        }
        _host_state.index(_states[static_cast<size_t>(hip_draw_particle(_s->particles(), _s->weighted() ? &_weights : nullptr))]);
        return &_host_state;
    }
    void updateEstimation(Action const* a, Observation const* o, POMDP const& /*domain*/) override
    {
        int32_t const ai = a->index(), oi = o->index();
        _s->at_update();
        _s->check(fba_belief_update(_s->ctx(), &ai, &oi, nullptr));
        _s->step_done();
        _fresh = false;
    }

protected:
    std::shared_ptr<HipSession> _s;
    mutable IndexState _host_state;
    mutable std::vector<int32_t> _states;
    mutable std::vector<double> _weights;
    mutable bool _fresh = false;
};

// Where the Bayes-adaptive mirror gets the domain State objects it hands out inside a BAState: by default the BAPOMDP
// the reference passes to initiate / resetDomainStateDistribution (copyDomainState(domainState(i)), released with
// releaseDomainState); a host without a BAPOMDP (the tests: BAPOMDP.cpp cannot be built without Boost) supplies its own
// pair and compiles with -DFBA_ADAPTERS_NO_BAPOMDP, which removes the only calls into BAPOMDP.cpp from this header.
struct HipDomainStates {
    std::function<State const*(int)> make;
    std::function<void(State const*)> release;
};

// Bayes-adaptive belief (bapomdp / fbapomdp executables): adds
// BABelief::resetDomainStateDistribution (src/beliefs/bayes-adaptive/BABelief.hpp:34).
class HipBAParticleBelief : public beliefs::BABelief
{
public:
    explicit HipBAParticleBelief(std::shared_ptr<HipSession> s, HipDomainStates states = HipDomainStates()) :
            _s(std::move(s)), _states_src(std::move(states)), _size(0, 0, 0), _fsize({}, {}), _steps({}, {})
    {
        int32_t S = 0, A = 0, O = 0;
        _s->check(fba_domain_sizes(_s->ctx(), &S, &A, &O));
        _size     = Domain_Size(S, A, O);
        _counts_n = fba_counts_len(_s->ctx());
        if (_s->model() == FBA_MODEL_BA_FACTORED) {
            _layout.reset(new fba_factored_layout);
            _s->check(fba_get_factored_layout(_s->ctx(), _layout.get()));
            _fsize = Domain_Feature_Size(std::vector<int>(_layout->state_feature_size, _layout->state_feature_size + _layout->n_state_features),
                                         std::vector<int>(_layout->obs_feature_size, _layout->obs_feature_size + _layout->n_obs_features));
            _steps = bayes_adaptive::factored::BABNModel::Indexing_Steps(indexing::stepSize(_fsize._S), indexing::stepSize(_fsize._O));
        } else if (_s->model() != FBA_MODEL_BA_TABLE) {
            throw std::string("HipBAParticleBelief needs a Bayes-adaptive session (FBA_MODEL_BA_TABLE or FBA_MODEL_BA_FACTORED)");
        }
    }
    ~HipBAParticleBelief() override { drop_mirrors(); }

    // (experiment::bapomdp::run hands the BAPOMDP itself to initiate, BAPOMDPExperiment.cpp:44)
    void initiate(POMDP const& domain) override
    {
#ifndef FBA_ADAPTERS_NO_BAPOMDP
        if (!_states_src.make) _bapomdp = static_cast<BAPOMDP const*>(&domain);
#else
        (void)domain;
#endif
        drop_mirrors();
        _s->begin_run();
        _s->check(fba_belief_init(_s->ctx()));
    }
    void free(POMDP const& /*domain*/) override { drop_mirrors(); }
    // A BAState whose counts are the drawn particle's and whose _domain_state the caller may swap and put back
    // (RBAPOUCT.cpp:92-106; BAPOMDPExperiment.cpp:48-52 logs its counts).  One mirror per distinct particle drawn since the
    // last mutating call; all of them are released at that call.
    State const* sample() const override
    {
        if (_s->weighted() && _weights.empty()) {
            _weights.resize(static_cast<size_t>(_s->particles()));
            _s->check(fba_belief_get(_s->ctx(), 0, nullptr, _weights.data(), nullptr));
        }
        int const i = hip_draw_particle(_s->particles(), _s->weighted() ? &_weights : nullptr);
        auto it     = _mirrors.find(i);
        if (it == _mirrors.end()) it = _mirrors.emplace(i, mirror(i)).first;
        return it->second;
    }
    void updateEstimation(Action const* a, Observation const* o, POMDP const& /*domain*/) override
    {
        int32_t const ai = a->index(), oi = o->index();
        drop_mirrors();
        _s->at_update();
        _s->check(fba_belief_update(_s->ctx(), &ai, &oi, nullptr));
        _s->step_done();
    }
    void resetDomainStateDistribution(BAPOMDP const& bapomdp) override
    {
        if (!_states_src.make) _bapomdp = &bapomdp;
        drop_mirrors();
        _s->begin_episode();
        _s->check(fba_belief_reset_domain_state(_s->ctx()));
    }

private:
    State const* domain_state(int i) const
    {
        if (_states_src.make) return _states_src.make(i);
#ifndef FBA_ADAPTERS_NO_BAPOMDP
        if (_bapomdp) return _bapomdp->copyDomainState(_bapomdp->domainState(i));
#endif
        throw std::string("HipBAParticleBelief::sample: no source of domain states (before initiate, or built without BAPOMDP and without HipDomainStates)");
    }
    void release_domain_state(State const* s) const
    {
        if (_states_src.release) _states_src.release(s);
#ifndef FBA_ADAPTERS_NO_BAPOMDP
        else if (!_states_src.make && _bapomdp) _bapomdp->releaseDomainState(s);
#endif
    }
    void drop_mirrors() const
    {
        for (auto& m : _mirrors) {
            release_domain_state(m.second->_domain_state);
            delete m.second;
        }
        _mirrors.clear();
        _weights.clear();
    }
    // particle i as the reference's own state class
    BAState* mirror(int i) const
    {
        int32_t state = 0;
        std::vector<float> counts(static_cast<size_t>(_counts_n));
        _s->check(fba_belief_get_particle(_s->ctx(), 0, i, &state, nullptr, counts.data()));
        if (_s->model() == FBA_MODEL_BA_TABLE) {   // BAFlatModel layout: phi[s][a][s'] then psi[a][s'][o] (BAFlatModel.hpp:19-124)
            size_t const nphi = static_cast<size_t>(_size._S) * _size._A * _size._S;
            auto phi = std::make_shared<std::vector<float> const>(counts.begin(), counts.begin() + static_cast<std::ptrdiff_t>(nphi));
            auto psi = std::make_shared<std::vector<float> const>(counts.begin() + static_cast<std::ptrdiff_t>(nphi), counts.end());
            return new BAPOMDPState(domain_state(state), bayes_adaptive::table::BAFlatModel(phi, psi, &_size));
        }
        // factored: every node with the parents the particle's mask word names, rows in DBNNode::cptIndex order
        // (include/fba_hip.h, fba_get_factored_layout)
        using bayes_adaptive::factored::BABNModel;
        fba_factored_layout const& L = *_layout;
        int const FS = L.n_state_features, FO = L.n_obs_features;
        std::vector<DBNNode> T, O;
        for (int k = 0; k < L.n_nodes; ++k) {
            fba_factored_node const& nd = L.node[k];
            uint32_t mask = nd.fixed_mask;
            if (nd.mask_word >= 0) {   // parent-set words are uint32 bit patterns stored in float slots behind the counts
                float const w = counts[static_cast<size_t>(L.n_counts + nd.mask_word)];
                std::memcpy(&mask, &w, 4);
            }
            std::vector<int> parents, sizes;
            for (int j = 0; j < nd.n_candidates; ++j)
                if ((mask >> j) & 1u) {
                    parents.push_back(nd.candidate[j]);
                    sizes.push_back(nd.candidate_size[j]);
                }
            DBNNode node(&_fsize._S, parents, nd.out);
            int rows = 1;
            for (int sz : sizes) rows *= sz;
            std::vector<int> pv(parents.size(), 0);
            for (int r = 0; r < rows; ++r) {
                int rem = r;
                for (int j = static_cast<int>(parents.size()) - 1; j >= 0; --j) {   // last parent fastest
                    pv[static_cast<size_t>(j)] = rem % sizes[static_cast<size_t>(j)];
                    rem /= sizes[static_cast<size_t>(j)];
                }
                for (int v = 0; v < nd.out; ++v) node.count(pv, v) = counts[static_cast<size_t>(nd.offset + r * nd.out + v)];
            }
            (k < _size._A * FS ? T : O).push_back(std::move(node));
        }
        (void)FO;
        return new FBAPOMDPState(domain_state(state), BABNModel(&_size, &_fsize, &_steps, std::move(T), std::move(O)));
    }

    std::shared_ptr<HipSession> _s;
    HipDomainStates _states_src;
    BAPOMDP const* _bapomdp = nullptr;
    Domain_Size _size;
    Domain_Feature_Size _fsize;
    bayes_adaptive::factored::BABNModel::Indexing_Steps _steps;
    std::unique_ptr<fba_factored_layout> _layout;
    int _counts_n = 0;
    mutable std::map<int, BAState*> _mirrors;
    mutable std::vector<double> _weights;
};

}  // namespace fba
