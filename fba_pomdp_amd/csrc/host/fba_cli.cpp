// fba_cli.cpp -- `fba_experiment planning|bapomdp|fbapomdp <flags>`: the reference's three
// executables (src/planning.cpp, src/bapomdp.cpp, src/fbapomdp.cpp) over libfba_hip.so.
//
// Flag names and defaults are the reference's (src/configurations/Conf.cpp:7-60, PlannerConf.cpp:5-26,
// BeliefConf.cpp:5-35, DomainConf.cpp:5-31, BAConf.cpp:42-68, FBAConf.cpp:5-25); the result file has
// the reference's format (PlannerExperiment.cpp:19-25: one "mean, var, count, stder, step duration"
// line; BAPOMDPExperiment.cpp:20-30: one such line per episode index), so
// analysis/preprocess/merge_result_files.py and the plotting scripts read it unchanged.
// -v 2 prints one "T=.. a=.. s'=.. o=.. r=.." line per real step (Episode.cpp:44-45), -v 3 adds the
// root statistics of every search (POUCT.cpp:93-101) and the rejection-loop count
// (RejectionSampling.hpp:68).
//
// All --runs execute concurrently on the GPU (--slots at a time); --seed is hashed into the Philox
// key, so results for a seed differ from the reference's mt19937 stream but not in distribution.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <chrono>
#include <fstream>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "fba_hip.h"
#include "conf_bridge.hpp"

namespace {

struct Options {
    std::string mode;
    std::string domain, planner = "po-uct", belief = "rejection_sampling", seed, output_file = "results.txt", structure_prior,
                                dirichlet = "expected", id, belief_option;
    int verbose = 0, runs = 1, horizon = 10, sims = 1000, max_depth = -1, particles = 100, size = 0, height = 0, width = 0,
        episodes = 1, slots = 0, device = 0, resample_amount = 0;
    double discount = .95, exploration = 100, threshold = 0;
    float noise = 0, counts_total = 10000;
    bool help = false;
};

void usage()
{
    std::puts(
        "usage: fba_experiment planning|bapomdp|fbapomdp [options]\n"
        "  -h, --help                     Print help options\n"
        "  -v, --verbose N                1: track episodes, 2: environment interactions, 3: algorithm summaries\n"
        "  -f, --output-file FILE         result file (results.txt)\n"
        "      --runs N                   Number of runs (1)\n"
        "  -H, --horizon N                Horizon, number of steps per episode (10)\n"
        "  -d, --discount X               Discount for future rewards (0.95)\n"
        "  -P, --planner NAME             random, ts (Thompson sampling) or po-uct (po-uct)\n"
        "  -B, --belief NAME              point_estimate, rejection_sampling, importance_sampling or (fbapomdp) reinvigoration,\n"
        "                                 cheating-reinvigoration, mh-within-gibbs\n"
        "      --seed STR                 Global seed for all random samples\n"
        "      --id STR                   The id to give this process\n"
        "  -s, --simulation-amount N      simulations per search (1000)\n"
        "      --mcts-max-depth N         max search depth, horizon if negative (-1)\n"
        "  -u, --exploration-constant X   UCB exploration constant (100)\n"
        "      --particle-amount N        particles in the filter (100)\n"
        "      --resample-amount N        particles reinvigorated per belief update (reinvigoration beliefs)\n"
        "      --threshold X              cheating-reinvigoration / mh-within-gibbs / mh-nips: log likelihood below which the belief is repaired (< 0);\n"
        "                                 incubator: normalised weight above which a shadow particle is promoted (0 < X <= 1)\n"
        "      --belief-option rs         mh-within-gibbs: state histories by rejection sampling instead of message passing\n"
        "  -D, --domain NAME              episodic-tiger, continuous-tiger, episodic-factored-tiger,\n"
        "                                 continuous-factored-tiger, gridworld, random-collision-avoidance,\n"
        "                                 centered-collision-avoidance, independent-sysadmin, linear-sysadmin,\n"
        "                                 coffee, boutilier-coffee, agr (planning only)\n"
        "      --size N  --height N  --width N\n"
        "      --episodes N               (bapomdp, fbapomdp) episodes per run (1)\n"
        "      --dirichlet_sampling_method regular|expected (expected)\n"
        "      --noise X                  prior noise (0)\n"
        "  -C, --counts-total X           prior counts (10000)\n"
        "      --structure-prior NAME     (fbapomdp) '', uniform, match-counts, match-uniform, fully-connected\n"
        "      --slots N  --device N      runs resident at once on the GPU (auto), HIP device (0)");
}

bool parse(int argc, char** argv, Options& o, std::string& err)
{
    if (argc < 2) { err = "missing mode (planning, bapomdp or fbapomdp)"; return false; }
    o.mode = argv[1];
    if (o.mode == "-h" || o.mode == "--help") { o.help = true; return true; }
    if (o.mode != "planning" && o.mode != "bapomdp" && o.mode != "fbapomdp") { err = "unknown mode '" + o.mode + "'"; return false; }
    o.id = std::to_string(std::time(nullptr));
    static const std::map<std::string, std::string> shorts = {
        {"-h", "--help"}, {"-v", "--verbose"}, {"-f", "--output-file"}, {"-H", "--horizon"}, {"-d", "--discount"},
        {"-P", "--planner"}, {"-B", "--belief"}, {"-s", "--simulation-amount"}, {"-u", "--exploration-constant"},
        {"-D", "--domain"}, {"-C", "--counts-total"}};
    for (int i = 2; i < argc; ++i) {
        std::string k = argv[i], v;
        const size_t eq = k.find('=');
        bool has_v = false;
        if (k.rfind("--", 0) == 0 && eq != std::string::npos) { v = k.substr(eq + 1); k = k.substr(0, eq); has_v = true; }
        if (shorts.count(k)) k = shorts.at(k);
        if (k == "--help") { o.help = true; continue; }
        if (!has_v) {
            if (i + 1 >= argc) { err = "the required argument for option '" + k + "' is missing"; return false; }
            v = argv[++i];
        }
        try {
            if (k == "--verbose") o.verbose = std::stoi(v);
            else if (k == "--output-file") o.output_file = v;
            else if (k == "--runs") o.runs = std::stoi(v);
            else if (k == "--horizon") o.horizon = std::stoi(v);
            else if (k == "--discount") o.discount = std::stod(v);
            else if (k == "--planner") o.planner = v;
            else if (k == "--belief") o.belief = v;
            else if (k == "--seed") o.seed = v;
            else if (k == "--id") o.id = v;
            else if (k == "--simulation-amount") o.sims = std::stoi(v);
            else if (k == "--mcts-max-depth") o.max_depth = std::stoi(v);
            else if (k == "--exploration-constant") o.exploration = std::stod(v);
            else if (k == "--particle-amount") o.particles = std::stoi(v);
            else if (k == "--resample-amount") o.resample_amount = std::stoi(v);
            else if (k == "--threshold") o.threshold = std::stod(v);
            else if (k == "--belief-option") o.belief_option = v;
            else if (k == "--domain") o.domain = v;
            else if (k == "--size") o.size = std::stoi(v);
            else if (k == "--height") o.height = std::stoi(v);
            else if (k == "--width") o.width = std::stoi(v);
            else if (k == "--episodes") o.episodes = std::stoi(v);
            else if (k == "--dirichlet_sampling_method") o.dirichlet = v;
            else if (k == "--noise") o.noise = std::stof(v);
            else if (k == "--counts-total") o.counts_total = std::stof(v);
            else if (k == "--structure-prior") o.structure_prior = v;
            else if (k == "--slots") o.slots = std::stoi(v);
            else if (k == "--device") o.device = std::stoi(v);
            else { err = "unrecognised option '" + k + "'"; return false; }
        } catch (std::exception const&) {
            err = "the argument ('" + v + "') for option '" + k + "' is invalid";
            return false;
        }
    }
    return true;
}

// The reference seeds mt19937 from the characters of --seed; here they key Philox (FNV-1a: conf_bridge.hpp, the function the
// factory patch of INTEGRATION.md uses too).
uint64_t seed_from(std::string const& s) { return fba::seed_from_string(s); }

bool to_config(Options const& o, fba_config& c, std::string& err)
{
    fba_default_config(&c);
    static const std::map<std::string, int> domains = {
        {"episodic-tiger", FBA_DOM_TIGER_EPISODIC}, {"continuous-tiger", FBA_DOM_TIGER_CONTINUOUS},
        {"episodic-factored-tiger", FBA_DOM_FTIGER_EPISODIC}, {"continuous-factored-tiger", FBA_DOM_FTIGER_CONTINUOUS},
        {"gridworld", FBA_DOM_GRIDWORLD}, {"random-collision-avoidance", FBA_DOM_COLLISION_AVOID},
        {"centered-collision-avoidance", FBA_DOM_COLLISION_AVOID_CENTERED},
        {"independent-sysadmin", FBA_DOM_SYSADMIN_INDEPENDENT}, {"linear-sysadmin", FBA_DOM_SYSADMIN_LINEAR},
        {"coffee", FBA_DOM_COFFEE}, {"boutilier-coffee", FBA_DOM_COFFEE_BOUTILIER}, {"agr", FBA_DOM_AGR}};
    if (!domains.count(o.domain)) { err = "please enter a legit domain, provided: " + o.domain; return false; }  // DomainConf.cpp:52-58
    c.domain = domains.at(o.domain);
    if (o.planner == "po-uct" || o.planner == "hip-po-uct") c.planner = FBA_PLANNER_POUCT;
    else if (o.planner == "random") c.planner = FBA_PLANNER_RANDOM;
    else if (o.planner == "ts") c.planner = FBA_PLANNER_TS;
    else { err = "please enter a legit planner: random, ts or po-uct, provided: " + o.planner; return false; }
    if (o.belief == "rejection_sampling" || o.belief == "hip-rejection_sampling") c.belief = FBA_BELIEF_REJECTION;
    else if (o.belief == "importance_sampling" || o.belief == "hip-importance_sampling") c.belief = FBA_BELIEF_IMPORTANCE;
    else if (o.belief == "point_estimate") c.belief = FBA_BELIEF_POINT;  // Belief.cpp:13-14, BABelief.cpp:19-20
    else if (o.belief == "reinvigoration" && o.mode == "fbapomdp") c.belief = FBA_BELIEF_REINVIGORATION;  // BABelief.cpp:28-31
    else if (o.belief == "cheating-reinvigoration" && o.mode == "fbapomdp") c.belief = FBA_BELIEF_CHEATING;  // BABelief.cpp:60-65
    else if (o.belief == "mh-within-gibbs" && o.mode == "fbapomdp") {  // BABelief.cpp:33-47: "" = MSG, "rs" = RS
        c.belief = FBA_BELIEF_MH_GIBBS;
        if (!o.belief_option.empty() && o.belief_option != "rs") {  // BeliefConf.cpp:51-56
            err = "You have set the illegal belief_option '" + o.belief_option + "' with belief " + o.belief + ".";
            return false;
        }
        c.belief_option = o.belief_option == "rs" ? 1 : 0;
    }
    else if (o.belief == "mh-nips" && o.mode == "fbapomdp") c.belief = FBA_BELIEF_MH_NIPS;  // BABelief.cpp:33-36
    else if (o.belief == "nested" && o.mode != "planning") c.belief = FBA_BELIEF_NESTED;   // BABelief.cpp:67-70
    else if (o.belief == "incubator" && o.mode == "fbapomdp") c.belief = FBA_BELIEF_INCUBATOR;   // BABelief.cpp:53-58
    else { err = "please enter a legit state stimator: point_estimate, rejection_sampling, importance_sampling or (fbapomdp) reinvigoration, provided: " + o.belief; return false; }
    if ((o.resample_amount == 0) ^ (o.belief != "reinvigoration" && o.belief != "cheating-reinvigoration" && o.belief != "incubator")) {  // BeliefConf.cpp:40-49
        err = "You have set the resample amount (" + std::to_string(o.resample_amount) + "), but are not using one of the beliefs (" + o.belief +
              ") that use it: reinvigoration, cheating-reinvigoration and incubator";
        return false;
    }
    if (!o.belief_option.empty() && o.belief != "mh-within-gibbs") {  // BeliefConf.cpp:51-56
        err = "You have set the illegal belief_option '" + o.belief_option + "' with belief " + o.belief + ".";
        return false;
    }
    c.resample_amount = o.resample_amount;
    c.threshold = o.threshold;
    if (o.dirichlet == "expected" || o.dirichlet == "1") c.dirichlet_regular = 0;
    else if (o.dirichlet == "regular" || o.dirichlet == "0") c.dirichlet_regular = 1;
    else { err = "please enter either 'regular' or 'expected' for dirichlet_sampling_method, given: " + o.dirichlet; return false; }
    if (o.structure_prior.empty() || o.structure_prior == "match-counts") c.structure_prior = FBA_SP_NONE;
    else if (o.structure_prior == "uniform") c.structure_prior = FBA_SP_UNIFORM;
    else if (o.structure_prior == "match-uniform") c.structure_prior = FBA_SP_MATCH_UNIFORM;
    else if (o.structure_prior == "fully-connected") c.structure_prior = FBA_SP_FULLY_CONNECTED;
    else { err = "unknown structure prior '" + o.structure_prior + "'"; return false; }
    c.model = o.mode == "planning" ? FBA_MODEL_POMDP : (o.mode == "bapomdp" ? FBA_MODEL_BA_TABLE : FBA_MODEL_BA_FACTORED);
    if (o.mode != "planning" && o.episodes < 1) { err = "episodes must be >= 1"; return false; }
    if (o.mode == "planning" && c.dirichlet_regular) c.dirichlet_regular = 0;
    c.size = o.size; c.width = o.width; c.height = o.height;
    c.particles = o.particles; c.sims = o.sims; c.max_depth = o.max_depth; c.horizon = o.horizon;
    c.exploration = o.exploration; c.discount = o.discount;
    c.runs = o.runs; c.episodes = o.mode == "planning" ? 1 : o.episodes;
    c.noise = o.noise; c.counts_total = o.counts_total;
    c.seed = seed_from(o.seed);
    c.slots = o.slots; c.device = o.device;
    c.trace = o.verbose >= 3 ? 2 : (o.verbose >= 2 ? 1 : 0);   // (2: with the filter's state histogram after every update)
    return true;
}

// -v 1 / 2 / 3 in the reference's own format, "V%vlevel: %fbase\t%msg" (ArgumentParser.cpp:16-20), from the trace the engine kept:
//   V1  run / episode                                       BAPOMDPExperiment.cpp:56, PlanningExperiment.cpp:41
//   V2  the real step, the end of an episode                 Episode.cpp:44, :58
//   V3  po-uct's pick and root statistics                    POUCT.cpp:94-100, RBAPOUCT.cpp:118-124 (ChanceNode::toString, MCTSTreeNodes.cpp:24-28)
//       the rejection update's loops, the filter behind it   RejectionSampling.hpp:68; (BA)RejectionSampling.cpp:39 / :46 + FlatFilter::toString (FlatFilter.cpp:70-94)
//       the importance update's total weight                 ImportanceSampler.hpp:56
// States, actions and observations are printed as the index elements they are here, "(i)" (IndexedElements.hpp:25) -- the reference's
// gridworld / collision-avoidance types describe themselves in words.  The runs of an experiment execute side by side; their records
// are printed run by run, episode by episode.
void print_trace(fba_ctx* ctx, const Options& o, const fba_config& cfg, int A)
{
    const int n = fba_trace_count(ctx);
    if (n <= 0) return;
    std::vector<fba_trace_rec> tr((size_t)n);
    const int got = fba_get_trace(ctx, tr.data(), n);
    const int verbose = o.verbose;
    const bool planning = o.mode == "planning", pouct = cfg.planner == FBA_PLANNER_POUCT;
    const int particles = cfg.belief == FBA_BELIEF_POINT ? 1 : cfg.particles;
    const char* planner_file = planning ? "POUCT.cpp" : "RBAPOUCT.cpp";
    std::vector<uint32_t> hist;   // the filter's state histogram after every update (domains of at most FBA_TRACE_HIST_BINS states)
    if (verbose >= 3) {
        hist.resize((size_t)n * FBA_TRACE_HIST_BINS);
        if (fba_get_trace_hist(ctx, hist.data(), n) != got) hist.clear();
    }
    double ret = 0, disc = 1;
    for (int i = 0; i < got; ++i) {
        const fba_trace_rec& r = tr[(size_t)i];
        if (r.t == 0) {
            if (planning) std::printf("V1: PlanningExperiment.cpp\trun %d/%d\n", r.run + 1, cfg.runs);
            else std::printf("V1: BAPOMDPExperiment.cpp\trun %d/%d, episode %d/%d\n", r.run + 1, cfg.runs, r.episode + 1, cfg.episodes);
            ret = 0; disc = 1;
        }
        if (verbose >= 3 && pouct) {
            const int a = r.action;
            std::printf("V3: %s\tpo-uct picked node (a=(%d), q=%f, n=%d) at tree of depth=%d and %d action nodes\n", planner_file, a,
                        r.root_q[a], r.root_n[a], r.tree_depth, r.n_nodes);
            std::printf("V3: %s\tAction stats:\n", planner_file);
            for (int k = 0; k < A; ++k) std::printf("V3: %s\t\t(a=(%d), q=%f, n=%d)\n", planner_file, k, r.root_q[k], r.root_n[k]);
        }
        std::printf("V2: Episode.cpp\tT=%d\ta=(%d)\ts'=(%d)\to=(%d)\tr=%g\n", r.t, r.action, r.state, r.obs, r.reward);
        ret += r.reward * disc;
        disc *= cfg.discount;
        if (verbose >= 3 && !r.terminal) {
            if (r.update_count >= 0) {
                std::printf("V3: RejectionSampling.hpp\tperformed %d loops for rejection sampling for %d samples\n", r.update_count, particles);
                if (!hist.empty()) {   // one message of several lines, as the reference's
                    std::printf("V3: %s\tStatus of rejection sampling filter after update:Particle filter contains:\n",
                                planning ? "RejectionSampling.cpp" : "BARejectionSampling.cpp");
                    for (int st = 0; st < FBA_TRACE_HIST_BINS; ++st) {
                        const uint32_t k = hist[(size_t)i * FBA_TRACE_HIST_BINS + st];
                        if (k) std::printf("\t((%d): %f(%u))\n", st, k / static_cast<double>(particles), k);
                    }
                }
            } else if (cfg.belief == FBA_BELIEF_IMPORTANCE) {
                std::printf("V3: ImportanceSampler.hpp\tacquired total weight of %g after updating %d particles\n", r.weight_total, particles);
            }
        }
        const bool last = i + 1 == got || tr[(size_t)i + 1].t == 0;
        if (last) std::printf("V2: Episode.cpp\tEnd of episode at s=(%d) with return=%g\n", r.state, ret);
    }
}

}  // namespace

int main(int argc, char** argv)
{
    Options o;
    std::string err;
    if (!parse(argc, argv, o, err)) {
        std::fprintf(stderr, "ERROR: %s\n", err.c_str());
        return 1;
    }
    if (o.help) {
        usage();
        return 0;
    }
    fba_config cfg;
    if (!to_config(o, cfg, err)) {
        std::fprintf(stderr, "ERROR: %s\n", err.c_str());
        return 1;
    }
    fba_ctx* ctx = nullptr;
    if (fba_create(&cfg, &ctx) != FBA_OK) {
        std::fprintf(stderr, "ERROR: %s\n", fba_last_error(nullptr));
        return 1;
    }
    std::fprintf(stderr, "INFO: (%s): Starting %s experiment\n", o.id.c_str(), o.mode == "planning" ? "planning" : "BAPOMDP");
    std::vector<fba_stat> stats((size_t)cfg.episodes);
    const auto t0 = std::chrono::steady_clock::now();
    const int rc  = o.mode == "planning" ? fba_run_planning(ctx, stats.data()) : fba_run_bapomdp(ctx, stats.data());
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rc != FBA_OK) {
        std::fprintf(stderr, "ERROR: %s\n", fba_last_error(ctx));
        fba_destroy(ctx);
        return 1;
    }
    fba_counters cnt;
    fba_get_counters(ctx, &cnt);
    // "step duration mean": every run advances one real step per tick, so the time one run's step
    // takes is the wall time over the ticks = real steps of the longest chain of runs on one slot
    const int slots = fba_slots(ctx);
    const double chains = std::ceil((double)cfg.runs / slots);
    const double step_duration = cnt.env_steps ? wall / ((double)cnt.env_steps / ((double)cfg.runs / chains)) : 0.0;
    if (o.verbose >= 2) {
        int32_t S, A, O;
        fba_domain_sizes(ctx, &S, &A, &O);
        print_trace(ctx, o, cfg, A);
    }
    {
        std::ofstream f(o.output_file);
        f << "# version 1:\n# return mean, return var, return count, return stder, step duration mean\n";
        if (o.mode == "planning") {
            f << stats[0].mean << ", " << fba_stat_var(&stats[0]) << ", " << stats[0].count << ", " << fba_stat_stder(&stats[0]) << ", "
              << step_duration;
        } else {
            for (auto const& s : stats)
                f << s.mean << ", " << fba_stat_var(&s) << ", " << s.count << ", " << fba_stat_stder(&s) << ", " << step_duration << "\n";
        }
        f << std::endl;
    }
    std::fprintf(stderr, "INFO: (%s): Succesfully ran experiment: %llu simulated steps in %.3f s\n", o.id.c_str(),
                 (unsigned long long)(cnt.sim_steps + cnt.belief_steps), wall);
    fba_destroy(ctx);
    return 0;
}
