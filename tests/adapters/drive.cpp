// TEST INFRASTRUCTURE: drives the adapters through the REAL reference loops (episode::run, Episode.cpp:16-64; the
// run / episode loops of PlanningExperiment.cpp:39-52 and BAPOMDPExperiment.cpp:44-75 restated around it, because
// those two translation units need Boost) over the recording stub of the C-ABI.
#include <cstdio>
#include <memory>

#include "easylogging++.h"

#include "adapters.hpp"
#include "environment/Action.hpp"
#include "environment/Observation.hpp"
#include "domains/tiger/Tiger.hpp"
#include "environment/Discount.hpp"
#include "environment/Horizon.hpp"
#include "experiments/Episode.hpp"
#include "utils/random.hpp"

INITIALIZE_EASYLOGGINGPP

int main(int argc, char** argv)
{
    START_EASYLOGGINGPP(argc, argv);
    el::Configurations conf;
    conf.setToDefault();
    conf.setGlobally(el::ConfigurationType::Enabled, "false");
    el::Loggers::reconfigureAllLoggers(conf);
    rnd::initiate();
    { std::string seed_str("3"); rnd::seed(seed_str); }

    fba_config cfg;
    fba_default_config(&cfg);
    domains::Tiger env(domains::Tiger::EPISODIC), sim(domains::Tiger::EPISODIC);
    {
        // planning::run: for run { belief->initiate; episode::run; belief->free }
        std::printf("# planning\n");
        auto session = std::make_shared<fba::HipSession>(cfg);
        fba::HipPOUCT planner(session);
        fba::HipParticleBelief belief(session);
        for (int run = 0; run < 3; ++run) {
            belief.initiate(sim);
            auto const res = episode::run(planner, belief, env, sim, Horizon(4), Discount(.95));
            std::printf("episode length=%d\n", res.length);
            belief.free(sim);
        }
    }
    {
        // bapomdp::run: for run { belief->initiate; for episode { belief->resetDomainStateDistribution; episode::run } }
        // (BAPOMDP itself cannot be built without Boost: the Bayes-adaptive belief adapter is driven through the same
        // episode::run with the tiger POMDP as simulator; its resetDomainStateDistribution ignores its argument, so a
        // reference that is never used stands in for the BAPOMDP)
        std::printf("# bapomdp\n");
        cfg.model    = FBA_MODEL_BA_TABLE;
        auto session = std::make_shared<fba::HipSession>(cfg);
        fba::HipPOUCT planner(session);   // HipRBAPOUCT::selectAction is the same hip_select_action
        fba::HipDomainStates states;      // (no BAPOMDP here: the mirror's domain states are plain index states)
        states.make    = [](int i) -> State const* { return new IndexState(i); };
        states.release = [](State const* st) { delete st; };
        fba::HipBAParticleBelief belief(session, states);
        alignas(16) static char never_used[64];
        BAPOMDP const& no_bapomdp = *reinterpret_cast<BAPOMDP const*>(never_used);
        for (int run = 0; run < 2; ++run) {
            belief.initiate(sim);
            for (int ep = 0; ep < 3; ++ep) {
                belief.resetDomainStateDistribution(no_bapomdp);
                auto const res = episode::run(planner, belief, env, sim, Horizon(4), Discount(.95));
                std::printf("episode length=%d\n", res.length);
            }
            // Belief::sample() on the host: a BAPOMDPState with the drawn particle's counts (the stub's particle i holds
            // 100 i + k in cell k and state i & 1), whose domain state a planner may swap and put back (RBAPOUCT.cpp:92-106)
            for (int k = 0; k < 4; ++k) {
                auto particle = static_cast<BAState const*>(belief.sample());
                auto const* typed = dynamic_cast<BAPOMDPState const*>(particle);
                IndexState s0(0), s1(1);
                IndexAction a2(2);
                IndexObservation o1(1);
                int const idx = (int)(const_cast<BAPOMDPState*>(typed)->model()->count(&s0, &a2, &s0)) / 100;   // cell 4 = 100 i + 4
                std::printf("sample particle=%d state=%d phi(1,2,1)=%g psi(2,1,1)=%g\n", idx, particle->_domain_state->index(),
                            (double)const_cast<BAPOMDPState*>(typed)->model()->count(&s1, &a2, &s1),
                            (double)const_cast<BAPOMDPState*>(typed)->model()->count(&a2, &s1, &o1));
                auto const old = particle->_domain_state;
                IndexState swapped(1 - old->index());
                const_cast<BAState*>(particle)->_domain_state = &swapped;
                const_cast<BAState*>(particle)->_domain_state = old;
            }
            belief.free(sim);
        }
    }
    return 0;
}
