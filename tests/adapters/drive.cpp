// TEST INFRASTRUCTURE: drives the adapters through the REAL reference loops (episode::run, Episode.cpp:16-64; the
// run / episode loops of PlanningExperiment.cpp:39-52 and BAPOMDPExperiment.cpp:44-75 restated around it, because
// those two translation units need Boost) over the recording stub of the C-ABI.
#include <cstdio>
#include <memory>

#include "easylogging++.h"

#include "adapters.hpp"
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
        auto session = std::make_shared<fba::HipSession>(cfg);
        fba::HipPOUCT planner(session);   // HipRBAPOUCT::selectAction is the same hip_select_action
        fba::HipBAParticleBelief belief(session);
        alignas(16) static char never_used[64];
        BAPOMDP const& no_bapomdp = *reinterpret_cast<BAPOMDP const*>(never_used);
        for (int run = 0; run < 2; ++run) {
            belief.initiate(sim);
            for (int ep = 0; ep < 3; ++ep) {
                belief.resetDomainStateDistribution(no_bapomdp);
                auto const res = episode::run(planner, belief, env, sim, Horizon(4), Discount(.95));
                std::printf("episode length=%d\n", res.length);
            }
            belief.free(sim);
        }
    }
    return 0;
}
