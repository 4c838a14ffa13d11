// conf_bridge.hpp -- the reference's parsed command line (configurations::Conf / BAConf / FBAConf) as an fba_config.
//
// What a maintainer calls inside the factories.  A template over the configuration type,
// because every header under src/configurations includes <boost/program_options.hpp> (Conf.hpp:4) and this repo must be
// checkable without Boost: with the reference's own types it instantiates to exactly the member accesses below
// (Conf.hpp:18-34, DomainConf.hpp:14-19, PlannerConf.hpp:16-18, BeliefConf.hpp:16-21, BAConf.hpp:17-22, FBAConf.hpp:19);
// tests/test_adapters_compile.py instantiates it with a struct of the same members.
//
//   fba::to_fba_config(conf, FBA_MODEL_POMDP)          from factory::makePlanner / makeBelief          (planning)
//   fba::to_fba_config_ba(conf, FBA_MODEL_BA_TABLE)    from makeBAPlanner / makeBABelief               (bapomdp)
//   fba::to_fba_config_fba(conf)                       the same with FBAConf                            (fbapomdp)
// Unknown strings throw std::string with the reference's wording (DomainConf.cpp:52-58, Conf.cpp:81-103), which its
// main()s catch (planning.cpp:46-54).  A "hip-" prefix on the planner / belief key (the factory keys of the patch) is accepted.
#pragma once

#include <cstdint>
#include <ctime>
#include <string>

#include "fba_hip.h"

namespace fba {

// --seed: the reference seeds mt19937 with the characters of the string (random.cpp:76-83); the engine keys Philox with
// their FNV-1a hash -- the same function fba_experiment uses, so a seed names the same GPU run through either door
inline uint64_t seed_from_string(std::string const& s)
{
    if (s.empty()) return static_cast<uint64_t>(std::time(nullptr));  // rnd::initiate(): time(nullptr)
    uint64_t h = 1469598103934665603ull;
    for (unsigned char ch : s) { h ^= ch; h *= 1099511628211ull; }
    return h;
}

inline std::string strip_hip(std::string const& key) { return key.compare(0, 4, "hip-") == 0 ? key.substr(4) : key; }

inline int domain_id(std::string const& d)
{
    static const struct { char const* name; int id; } map[] = {
        {"episodic-tiger", FBA_DOM_TIGER_EPISODIC}, {"continuous-tiger", FBA_DOM_TIGER_CONTINUOUS},
        {"episodic-factored-tiger", FBA_DOM_FTIGER_EPISODIC}, {"continuous-factored-tiger", FBA_DOM_FTIGER_CONTINUOUS},
        {"gridworld", FBA_DOM_GRIDWORLD}, {"random-collision-avoidance", FBA_DOM_COLLISION_AVOID},
        {"centered-collision-avoidance", FBA_DOM_COLLISION_AVOID_CENTERED},
        {"independent-sysadmin", FBA_DOM_SYSADMIN_INDEPENDENT}, {"linear-sysadmin", FBA_DOM_SYSADMIN_LINEAR},
        {"coffee", FBA_DOM_COFFEE}, {"boutilier-coffee", FBA_DOM_COFFEE_BOUTILIER}, {"agr", FBA_DOM_AGR}};
    for (auto const& m : map)
        if (d == m.name) return m.id;
    throw std::string("please enter a legit domain, provided: " + d);  // DomainConf.cpp:52-58 (dummy, linear_dummy, factored-dummy: test scaffolding, not built)
}

// configurations::Conf: everything the planning executable knows
template <class Conf>
fba_config to_fba_config(Conf const& c, int model = FBA_MODEL_POMDP)
{
    fba_config f;
    fba_default_config(&f);
    f.model  = model;
    f.domain = domain_id(c.domain_conf.domain);
    f.size   = static_cast<int32_t>(c.domain_conf.size);
    f.width  = static_cast<int32_t>(c.domain_conf.width);
    f.height = static_cast<int32_t>(c.domain_conf.height);
    std::string const planner = strip_hip(c.planner), belief = strip_hip(c.belief);
    if (planner == "po-uct") f.planner = FBA_PLANNER_POUCT;
    else if (planner == "random") f.planner = FBA_PLANNER_RANDOM;
    else if (planner == "ts") f.planner = FBA_PLANNER_TS;
    else throw std::string("please enter a legit planner: random, ts or po-uct, provided: " + c.planner);
    bool const fba = model == FBA_MODEL_BA_FACTORED, ba = model != FBA_MODEL_POMDP;
    if (belief == "rejection_sampling") f.belief = FBA_BELIEF_REJECTION;
    else if (belief == "importance_sampling") f.belief = FBA_BELIEF_IMPORTANCE;
    else if (belief == "point_estimate") f.belief = FBA_BELIEF_POINT;
    else if (belief == "reinvigoration" && fba) f.belief = FBA_BELIEF_REINVIGORATION;
    else if (belief == "cheating-reinvigoration" && fba) f.belief = FBA_BELIEF_CHEATING;
    else if (belief == "mh-within-gibbs" && fba) f.belief = FBA_BELIEF_MH_GIBBS;
    else if (belief == "mh-nips" && fba) f.belief = FBA_BELIEF_MH_NIPS;
    else if (belief == "incubator" && fba) f.belief = FBA_BELIEF_INCUBATOR;
    else if (belief == "nested" && ba) f.belief = FBA_BELIEF_NESTED;
    else throw std::string("please enter a legit state stimator, provided: " + c.belief);
    f.particles       = static_cast<int32_t>(c.belief_conf.particle_amount);
    f.resample_amount = static_cast<int32_t>(c.belief_conf.resample_amount);
    f.threshold       = c.belief_conf.threshold;
    if (!c.belief_conf.option.empty() && !(f.belief == FBA_BELIEF_MH_GIBBS && c.belief_conf.option == "rs"))   // BeliefConf.cpp:51-56
        throw std::string("You have set the illegal belief_option '" + c.belief_conf.option + "' with belief " + c.belief + ".");
    f.belief_option = c.belief_conf.option == "rs" ? 1 : 0;
    f.sims        = c.planner_conf.mcts_simulation_amount;
    f.max_depth   = c.planner_conf.mcts_max_depth;      // -1 => the horizon, as ArgumentParser.cpp:37-40 rewrites it
    f.exploration = c.planner_conf.mcts_exploration_const;
    f.horizon     = c.horizon;
    f.discount    = c.discount;
    f.runs        = c.num_runs;
    f.episodes    = 1;
    f.seed        = seed_from_string(c.seed);
    f.trace       = c.verbose >= 3 ? 2 : (c.verbose >= 2 ? 1 : 0);
    return f;
}

// configurations::BAConf: + episodes, prior noise / total counts, the Dirichlet sampling method
template <class BAConf>
fba_config to_fba_config_ba(BAConf const& c, int model = FBA_MODEL_BA_TABLE)
{
    fba_config f        = to_fba_config(c, model);
    f.episodes          = c.num_episodes;
    f.noise             = c.noise;
    f.counts_total      = c.counts_total;
    f.dirichlet_regular = static_cast<int>(c.bayes_sample_method) == 0 ? 1 : 0;   // rnd::sample::Dir::SAMPLETYPE { Regular, Expected } (random.hpp)
    return f;
}

// configurations::FBAConf: + the structure prior
template <class FBAConf>
fba_config to_fba_config_fba(FBAConf const& c)
{
    fba_config f = to_fba_config_ba(c, FBA_MODEL_BA_FACTORED);
    if (c.structure_prior.empty() || c.structure_prior == "match-counts") f.structure_prior = FBA_SP_NONE;
    else if (c.structure_prior == "uniform") f.structure_prior = FBA_SP_UNIFORM;
    else if (c.structure_prior == "match-uniform") f.structure_prior = FBA_SP_MATCH_UNIFORM;
    else if (c.structure_prior == "fully-connected") f.structure_prior = FBA_SP_FULLY_CONNECTED;
    else throw std::string("unknown structure prior '" + c.structure_prior + "'");
    return f;
}

}  // namespace fba
