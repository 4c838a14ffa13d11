// TEST INFRASTRUCTURE: the reference's own episode loop (episode::run, Episode.cpp:16-64), its own Tiger environment
// and mt19937, driving the HIP engine on a real GPU through the adapters of fba_pomdp_amd/csrc/host/adapters.hpp --
// what a maintainer gets after the patch of INTEGRATION.md section 2.  Built where /root/reference exists (`make -C
// oracle ref`, into oracle/_ref/), run on the GPU box by tests/test_adapters_gpu.py, which replays the logged
// (action, observation) stream through the oracle and asks for the same actions.
//
//   drive_gpu <planning|bapomdp> <sims> <particles> <runs> <episodes> <horizon> <seed>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

#include "easylogging++.h"

#include "adapters.hpp"
#include "domains/tiger/Tiger.hpp"
#include "environment/Discount.hpp"
#include "environment/Environment.hpp"
#include "environment/Horizon.hpp"
#include "environment/Reward.hpp"
#include "experiments/Episode.hpp"
#include "utils/random.hpp"

INITIALIZE_EASYLOGGINGPP

namespace {
// the real environment, every step written down
class LoggedEnvironment : public Environment
{
public:
    explicit LoggedEnvironment(Environment const& e) : _e(e) {}
    State const* sampleStartState() const override { return _e.sampleStartState(); }
    Terminal step(State const** s, Action const* a, Observation const** o, Reward* r) const override
    {
        auto const t = _e.step(s, a, o, r);
        std::printf("step a=%d o=%d r=%g terminal=%d\n", a->index(), (*o)->index(), r->toDouble(), t.terminated() ? 1 : 0);
        return t;
    }
    void releaseObservation(Observation const* o) const override { _e.releaseObservation(o); }
    void releaseState(State const* s) const override { _e.releaseState(s); }
    Observation const* copyObservation(Observation const* o) const override { return _e.copyObservation(o); }
    State const* copyState(State const* s) const override { return _e.copyState(s); }

private:
    Environment const& _e;
};
}  // namespace

int main(int argc, char** argv)
{
    if (argc < 8) {
        std::fprintf(stderr, "usage: drive_gpu <planning|bapomdp> <sims> <particles> <runs> <episodes> <horizon> <seed>\n");
        return 2;
    }
    el::Configurations conf;
    conf.setToDefault();
    conf.setGlobally(el::ConfigurationType::Enabled, "false");
    el::Loggers::reconfigureAllLoggers(conf);
    rnd::initiate();
    { std::string seed_str("3"); rnd::seed(seed_str); }

    bool const ba = std::strcmp(argv[1], "bapomdp") == 0;
    int const runs = std::atoi(argv[4]), episodes = std::atoi(argv[5]), horizon = std::atoi(argv[6]);
    fba_config cfg;
    fba_default_config(&cfg);
    cfg.domain    = FBA_DOM_TIGER_EPISODIC;
    cfg.model     = ba ? FBA_MODEL_BA_TABLE : FBA_MODEL_POMDP;
    cfg.belief    = FBA_BELIEF_REJECTION;
    cfg.sims      = std::atoi(argv[2]);
    cfg.particles = std::atoi(argv[3]);
    cfg.horizon   = horizon;
    cfg.episodes  = episodes;
    cfg.runs      = runs;
    cfg.seed      = std::strtoull(argv[7], nullptr, 10);

    domains::Tiger tiger(domains::Tiger::EPISODIC), sim(domains::Tiger::EPISODIC);
    LoggedEnvironment env(tiger);
    try {
        auto session = std::make_shared<fba::HipSession>(cfg);
        fba::HipPOUCT planner(session);
        if (!ba) {   // experiment::planning::run (PlanningExperiment.cpp:39-52)
            fba::HipParticleBelief belief(session);
            for (int run = 0; run < runs; ++run) {
                std::printf("run %d\n", run);
                belief.initiate(sim);
                std::printf("episode 0\n");
                auto const res = episode::run(planner, belief, env, sim, Horizon(horizon), Discount(.95));
                std::printf("return %.17g length %d\n", res.ret.toDouble(), res.length);
                belief.free(sim);
            }
        } else {     // experiment::bapomdp::run (BAPOMDPExperiment.cpp:44-75); BAPOMDP itself needs Boost: the adapter ignores it
            fba::HipBAParticleBelief belief(session);
            alignas(16) static char never_used[64];
            BAPOMDP const& no_bapomdp = *reinterpret_cast<BAPOMDP const*>(never_used);
            for (int run = 0; run < runs; ++run) {
                std::printf("run %d\n", run);
                belief.initiate(sim);
                for (int ep = 0; ep < episodes; ++ep) {
                    std::printf("episode %d\n", ep);
                    belief.resetDomainStateDistribution(no_bapomdp);
                    auto const res = episode::run(planner, belief, env, sim, Horizon(horizon), Discount(.95));
                    std::printf("return %.17g length %d\n", res.ret.toDouble(), res.length);
                }
                belief.free(sim);
            }
        }
    } catch (std::string const& e) {
        std::fprintf(stderr, "error: %s\n", e.c_str());
        return 1;
    }
    return 0;
}
