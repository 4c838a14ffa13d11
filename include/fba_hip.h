/*
 * include/fba_hip.h -- C-ABI of libfba_hip.so, the MI355X-native BA-POMCP engine.
 *
 * This is the drop-in boundary for the reference's hot path.  samkatt/fba-pomdp has no FFI of
 * its own: its boundary is a set of C++ abstract classes + string-keyed factories
 * (SURVEY.md section 8b).  Each entry point below names the reference interface it replaces;
 * INTEGRATION.md shows the C++ adapter classes (fba_pomdp_amd/csrc/host/adapters.hpp) a
 * maintainer registers in those factories.
 *
 * Conventions: every function returns 0 on success or a negative FBA_E* code;
 * fba_last_error() gives the message (the adapters re-throw it as std::string, which the
 * reference's main()s already catch: src/planning.cpp:46-54).  The caller owns every host
 * buffer; the ctx owns all device memory.  One ctx <-> one host thread <-> one HIP stream.
 *
 * A ctx holds `slots` independent (planner, belief) pairs that advance in lock-step on the
 * device; slots = 1 is exactly one reference Planner + Belief.  All per-slot array arguments
 * have length `slots`.
 */
#ifndef FBA_HIP_H
#define FBA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FBA_ABI_VERSION 3   /* 3: fba_config.tree_buckets; 2: fba_config.belief_option and .search_budget, fba_belief_get_particle, fba_get_trace_hist; hosts check fba_abi_version() == FBA_ABI_VERSION before fba_create */
#define FBA_MAX_ACTIONS 24

/* domains: reference src/domains, selected by -D (DomainConf.hpp) */
enum {
    FBA_DOM_TIGER_EPISODIC    = 0, /* episodic-tiger             src/domains/tiger/Tiger.cpp          */
    FBA_DOM_TIGER_CONTINUOUS  = 1, /* continuous-tiger                                                */
    FBA_DOM_FTIGER_EPISODIC   = 2, /* episodic-factored-tiger    src/domains/tiger/FactoredTiger.cpp  */
    FBA_DOM_FTIGER_CONTINUOUS = 3, /* continuous-factored-tiger                                       */
    FBA_DOM_GRIDWORLD         = 4, /* gridworld                  src/domains/gridworld/GridWorld.cpp  */
    FBA_DOM_COLLISION_AVOID   = 5, /* random-collision-avoidance src/domains/collision-avoidance/CollisionAvoidance.cpp */
    FBA_DOM_COLLISION_AVOID_CENTERED = 6, /* centered-collision-avoidance (VERSION INITIALIZE_CENTRE) */
    FBA_DOM_SYSADMIN_INDEPENDENT = 7, /* independent-sysadmin --size N  src/domains/sysadmin/SysAdmin.cpp */
    FBA_DOM_SYSADMIN_LINEAR      = 8, /* linear-sysadmin --size N                                         */
    FBA_DOM_COFFEE               = 9, /* coffee            src/domains/coffee/CoffeeProblem.cpp (planning only) */
    FBA_DOM_COFFEE_BOUTILIER     = 10, /* boutilier-coffee  (the version with acquiring / losing coffee switched off) */
    FBA_DOM_AGR                  = 11  /* agr               src/domains/agr/AGR.cpp, AGR(10): planning with rejection sampling only */
};
/* simulator: plain POMDP (planning), tabular BA-POMDP (bapomdp), factored (fbapomdp) */
enum { FBA_MODEL_POMDP = 0, FBA_MODEL_BA_TABLE = 1, FBA_MODEL_BA_FACTORED = 2 };
/* -B rejection_sampling | importance_sampling (BeliefConf.hpp, Belief.cpp:13-24) | reinvigoration
 * (BABelief.cpp:28-31: ReinvigoratingRejectionSampling, factored models only) */
enum { FBA_BELIEF_REJECTION = 0, FBA_BELIEF_IMPORTANCE = 1, FBA_BELIEF_REINVIGORATION = 2,
       FBA_BELIEF_CHEATING = 3, /* cheating-reinvigoration (BABelief.cpp:60-65: prototypes::CheatingReinvigoration) */
       FBA_BELIEF_POINT = 4, /* point_estimate (Belief.cpp:13-14 PointEstimation, BABelief.cpp:19-20 BAPointEstimation): one state,
                             * updated by rejection; `particles` is ignored (1) and sample() draws nothing */
       FBA_BELIEF_MH_GIBBS = 5, /* mh-within-gibbs (BABelief.cpp:37-47: factored::MHwithinGibbs; `belief_option` 1 = "rs"): importance
                                 * filter whose particles are re-drawn by a Metropolis-Hastings chain over structures when the log
                                 * likelihood falls below `threshold`; factored tiger, collision avoidance, gridworld, sysadmin */
       FBA_BELIEF_MH_NIPS = 6,  /* mh-nips (BABelief.cpp:33-36: factored::MHNIPS2018): the same filter and trigger; the re-draw makes
                                 * independent proposals (a particle's structure or a mutation, updated along a simulated history) */
       FBA_BELIEF_INCUBATOR = 8, /* incubator (BABelief.cpp:53-58: factored::StructureIncubatorSampling(particles, resample_amount, threshold)): the
                                  * reinvigoration belief's two rejection filters + a weighted shadow filter of bred particles
                                  * (fba_belief_get_shadow), importance-sampled; factored tiger, collision avoidance, sysadmin */
       FBA_BELIEF_NESTED = 7    /* nested (BABelief.cpp:67-70: NestedBelief(particles, particles^2)): a weighted filter of count particles,
                                 * each with its own flat filter of particles^2 domain states (fba_belief_get_nested); bapomdp / fbapomdp */ };
/* -P po-uct | random | ts (Planner.cpp:12-19, BAPlanner.cpp:13-20; ts = Thompson sampling: TSPlanner / BATSPlanner) */
enum { FBA_PLANNER_POUCT = 0, FBA_PLANNER_RANDOM = 1, FBA_PLANNER_TS = 2 };
/* --structure-prior (FBAConf.hpp) */
enum { FBA_SP_NONE = 0, FBA_SP_UNIFORM = 1, FBA_SP_MATCH_UNIFORM = 2, FBA_SP_FULLY_CONNECTED = 3 };

/* Philox stream phases: a draw is addressed by (seed, run, episode, t, phase, unit, draw#) */
enum {
    FBA_PHASE_INIT      = 0,
    FBA_PHASE_RESET     = 1,
    FBA_PHASE_START     = 2,
    FBA_PHASE_SEARCH    = 3,
    FBA_PHASE_ENV       = 4,
    FBA_PHASE_REJECT    = 5,
    FBA_PHASE_IS_UPDATE = 6,
    FBA_PHASE_RESAMPLE  = 7,
    FBA_PHASE_REINVIG   = 8,  /* reinvigoration: unit = index of the bred particle            */
    FBA_PHASE_INIT_FC   = 9,  /* fully connected filter of the reinvigoration belief: initiate */
    FBA_PHASE_RESET_FC  = 10, /*   ... resetDomainStateDistribution                            */
    FBA_PHASE_REJECT_FC = 11, /*   ... rejection sampling, unit = attempt index                */
    FBA_PHASE_RESET_SH  = 12, /* incubator belief, shadow filter: resetDomainStateDistribution     */
    FBA_PHASE_INIT_SH   = 13  /*   ... initiate, unit = index of the bred particle                 */
};

enum {
    FBA_OK        = 0,
    FBA_EINVAL    = -1, /* bad argument / unsupported configuration (reference: throw "...") */
    FBA_EHIP      = -2, /* HIP runtime error                                                  */
    FBA_ENODEVICE = -3, /* no gfx950 device visible: the engine never falls back to the CPU   */
    FBA_ESTATE    = -4  /* call order violated (e.g. select_action before belief_init)        */
};

/* Flag names and defaults follow the reference CLI (Conf.hpp:14-45, PlannerConf.hpp:16-18,
 * BeliefConf.hpp:16-21, BAConf.hpp:17-22, FBAConf.hpp).  fba_default_config() fills them. */
typedef struct fba_config {
    int32_t domain;          /* -D                                  */
    int32_t size;            /* --size                              */
    int32_t width;           /* --width                             */
    int32_t height;          /* --height                            */
    int32_t model;           /* which executable: planning / bapomdp / fbapomdp */
    int32_t belief;          /* -B                                  */
    int32_t planner;         /* -P                                  */
    int32_t particles;       /* --particle-amount        (100)      */
    int32_t sims;            /* -s / --num-sims          (1000)     */
    int32_t max_depth;       /* --mcts-max-depth   (-1 => horizon)  */
    int32_t horizon;         /* -H                       (10)       */
    double exploration;      /* -u                       (100)      */
    double discount;         /* -d                       (0.95)     */
    int32_t runs;            /* --runs                   (1)        */
    int32_t episodes;        /* --episodes               (1)        */
    float noise;             /* --noise                  (0)        */
    float counts_total;      /* -C                       (10000)    */
    int32_t structure_prior; /* --structure-prior                   */
    uint64_t seed;           /* --seed (Philox key)                 */
    int32_t run_offset;      /* global index of this ctx's first run (episode sharding)      */
    int32_t slots;           /* concurrent runs on the device; 0 => min(runs, auto)          */
    int32_t device;          /* HIP device ordinal                                           */
    int32_t trace;           /* 1 => record one fba_trace_rec per real time-step; 2 => also the filter's state histogram
                              * after every belief update (domains of at most FBA_TRACE_HIST_BINS states): fba_get_trace_hist */
    int32_t dirichlet_regular; /* --dirichlet_sampling_method regular (0 = expected, default);
                                * bug-compatible with the reference's biased sampler (BAConf.hpp:22,
                                * random.cpp:146-242) */
    int32_t resample_amount; /* --resample-amount: particles bred per update by the reinvigoration
                              * belief / copied per cheat by the cheating belief (BeliefConf.cpp:17-21) */
    double threshold;        /* --threshold: log likelihood below which the cheating belief cheats (< 0) */
    int32_t belief_option;   /* --belief-option: mh-within-gibbs 0 = state histories by message passing (default), 1 = "rs" */
    int32_t search_budget;   /* history-particle searches (gridworld FBA-POMDP, importance filter): iterations of the search loop per launch.
                              * 0 = a launch runs every slot's whole search (lock-step ticks: a tick lasts as long as its deepest tree);
                              * > 0 = a launch stops at the first simulation boundary behind that many iterations, unfinished searches
                              * are parked in their trees and resumed by the next launch, and slots whose search is done take their
                              * real step and belief update meanwhile -- slots advance on their own, results are the same
                              * (Episode.cpp:39-55 and BAPOMDPExperiment.cpp:44-75 never couple two runs) */
    int32_t tree_buckets;    /* history-particle searches: 64-byte node buckets per slot's tree (the reference's tree is heap-allocated and
                              * unbounded: MCTSTreeNodes.hpp:39-108).  0 = 2 * (sims + 2), which no search can fill -- 8.4 MB per slot at 65 536
                              * simulations.  A gridworld tree gives a bucket only to the nodes it reaches a second time (a quarter of them at
                              * the BASELINE size), so a throughput run that wants more slots per GB passes less here; a search that does outgrow
                              * its table stops with FBA_ESTATE and a message, never with a wrong result */
} fba_config;

/* One record per real time-step: the information the reference prints at -v 2 / -v 3
 * (Episode.cpp:44-45, POUCT.cpp:93-101, RejectionSampling.hpp:68) plus a belief checksum. */
typedef struct fba_trace_rec {
    int32_t run, episode, t;
    int32_t action, state, obs;
    int32_t terminal;
    int32_t n_nodes, tree_depth;
    int32_t update_count;
    int32_t root_n[FBA_MAX_ACTIONS];
    double root_q[FBA_MAX_ACTIONS];
    double reward;
    double weight_total;
    uint64_t belief_hash;
} fba_trace_rec;

/* utils::Statistic (src/utils/Statistic.cpp:5-46) */
typedef struct fba_stat {
    double count, mean, m2;
} fba_stat;

typedef struct fba_counters {
    uint64_t sim_steps;    /* simulator.step calls made by the planner (tree + rollout) */
    uint64_t belief_steps; /* simulator.step calls made by the belief update            */
    uint64_t env_steps;    /* true-environment steps                                     */
} fba_counters;

/* kernels the engine times with HIP events on its own stream (bench.py roofline) */
enum {
    FBA_K_SEARCH       = 0,
    FBA_K_ENV          = 1,
    FBA_K_BELIEF_RS    = 2, /* rejection update: sample, step, compact, gather */
    FBA_K_BELIEF_IS    = 3, /* importance update + scan + resample gather      */
    FBA_K_BELIEF_RESET = 4,
    FBA_K_BELIEF_INIT  = 5,
    FBA_K_COUNT        = 6
};
typedef struct fba_kernel_time {
    double ms;          /* sum of HIP-event durations */
    uint64_t launches;
    uint64_t units;     /* particles written (belief kernels) / simulated steps (search) */
    uint64_t bytes;     /* algorithmic bytes, SURVEY.md section 8(d) formulas            */
} fba_kernel_time;

typedef struct fba_ctx fba_ctx;

int fba_abi_version(void);
void fba_default_config(fba_config* cfg);

/* Construction.  Replaces the factory calls of experiment::planning::run /
 * experiment::bapomdp::run (PlanningExperiment.cpp:31-36, BAPOMDPExperiment.cpp:36-42):
 * makePlanner / makeBAPlanner, makeBelief / makeBABelief, makePOMDP / makeTBAPOMDP / makeFBAPOMDP. */
int fba_create(const fba_config* cfg, fba_ctx** out);
void fba_destroy(fba_ctx* ctx);
const char* fba_last_error(const fba_ctx* ctx); /* ctx may be NULL: last create error */

int fba_domain_sizes(const fba_ctx* ctx, int32_t* S, int32_t* A, int32_t* O);
int fba_counts_len(const fba_ctx* ctx); /* floats per particle count blob (0 for plain POMDP) */
int fba_particle_bytes(const fba_ctx* ctx); /* HBM bytes of one particle record.  Tabular tiger particles are stored packed
                                             * (uint16 increment counts over the shared prior, 64 B instead of 128 B) when the
                                             * prior allows it exactly; FBA_DENSE_PARTICLES=1 in the environment forces fp32 counts.
                                             * fba_belief_get / fba_belief_set always speak fp32 counts. */
int fba_slots(const fba_ctx* ctx);      /* slots actually resident (cfg.slots, or the library's choice) */

/* Prior count tables.  fba_create builds the domain's own prior (TigerPriors.cpp:14-43,
 * FactoredTigerPriors.cpp:18-88); this call overrides it with tables built by the caller
 * (BAFlatModel layout: phi[s*A*S + a*S + s'], psi[a*S*O + s'*O + o]).  Takes effect at the next
 * fba_belief_init (the prior is what initiate copies into the particles); where particles are stored packed
 * (fba_particle_bytes) that call is required before the belief is used again. */
int fba_set_model_tabular(fba_ctx* ctx, const float* phi, const float* psi);
int fba_get_prior(const fba_ctx* ctx, float* counts);

/* Layout of a factored particle's count blob (model = FBA_MODEL_BA_FACTORED), for hosts that read or
 * write particles with fba_belief_get / fba_belief_set -- what BABNModel + DBNNode hold per state
 * (BABNModel.hpp:30-200, DBNNode.hpp:20-120), flattened:
 *   blob = n_counts floats of CPTs, then n_mask_words words (uint32 bit patterns stored in float slots).
 *   node k: T(a, f) = node[a * n_state_features + f], O(a, f) = node[A * n_state_features + a * n_obs_features + f].
 *   A node owns room for EVERY candidate parent set ("max layout"); the parents in use are the candidates j
 *   whose bit j is set in the particle's mask word `mask_word` (or in `fixed_mask` when mask_word < 0).
 *   Row of parent values v: idx = 0; for j in candidate order, if bit j set: idx = idx * candidate_size[j] + v[candidate[j]];
 *   the row starts at blob[offset + idx * out] and has `out` counts (DBNNode::cptIndex, last parent fastest).
 *   Parent values are features of the PREVIOUS state for T nodes and of the NEW state for O nodes; a state index
 *   is its features in mixed radix, last feature fastest (indexing::project, utils/index.cpp:51-83). */
#define FBA_MAX_FEATURES 8
#define FBA_MAX_NODES 160
typedef struct fba_factored_node {
    int32_t offset, out, n_candidates, mask_word;
    uint32_t fixed_mask;
    uint8_t candidate[FBA_MAX_FEATURES];      /* state-feature ids, in order */
    uint8_t candidate_size[FBA_MAX_FEATURES]; /* number of values of each candidate */
} fba_factored_node;
typedef struct fba_factored_layout {
    int32_t n_state_features, n_obs_features, n_nodes, n_counts, n_mask_words;
    int32_t state_feature_size[FBA_MAX_FEATURES], obs_feature_size[FBA_MAX_FEATURES];
    fba_factored_node node[FBA_MAX_NODES];
} fba_factored_layout;
int fba_get_factored_layout(const fba_ctx* ctx, fba_factored_layout* out);
/* The factored counterpart of fba_set_model_tabular: replaces the base prior every particle starts from with CPTs built
 * by the caller -- what FBAPOMDPPrior::sample copies into a new FBAPOMDPState (FBAPOMDPPrior.cpp:27-37: the prior's
 * BABNModel, node by node, DBNNode::count order).  `layout` must be the engine's own (fba_get_factored_layout: the host
 * walks it to know where node (a, f)'s rows go); `counts` = n_counts floats followed by n_mask_words parent-set words,
 * exactly a particle's blob.  Per-particle structure draws of the configured structure prior still apply on top.  Takes
 * effect at the next fba_belief_init, which is required before the belief is used again. */
int fba_set_model_factored(fba_ctx* ctx, const fba_factored_layout* layout, const float* counts);

/* ---- per-step interface: one call per reference virtual call -------------------------- */

/* Where each slot is in its experiment; addresses the Philox streams of the calls below. */
int fba_set_position(fba_ctx* ctx, const int32_t* run, const int32_t* episode, const int32_t* t);

/* Belief::initiate(POMDP const&)                       src/beliefs/Belief.hpp:25          */
int fba_belief_init(fba_ctx* ctx);
/* BABelief::resetDomainStateDistribution(BAPOMDP const&) src/beliefs/bayes-adaptive/BABelief.hpp:34 */
int fba_belief_reset_domain_state(fba_ctx* ctx);
/* Planner::selectAction(POMDP const&, Belief const&, History const&)  src/planners/Planner.hpp:23-24
 * hist_len[slot] = History::length(); action[slot] <- chosen action index.
 * active may be NULL (all slots) */
int fba_select_action(fba_ctx* ctx, const int32_t* hist_len, const uint8_t* active, int32_t* action);
/* Belief::updateEstimation(Action const*, Observation const*, POMDP const&)  Belief.hpp:40 */
int fba_belief_update(fba_ctx* ctx, const int32_t* action, const int32_t* obs, const uint8_t* active);
/* Belief::sample() for a host-side planner, and bulk download for tests:
 * state[particles], weight[particles] (may be NULL), counts[particles * counts_len] (may be NULL) */
int fba_belief_get(fba_ctx* ctx, int32_t slot, int32_t* state, double* weight, float* counts);
int fba_belief_set(fba_ctx* ctx, int32_t slot, const int32_t* state, const double* weight, const float* counts);
/* One particle of the filter -- what Belief::sample() (Belief.hpp:33) hands a HOST-side planner once the host has drawn the
 * index (FlatFilter::sample FlatFilter.cpp:97-102: uniform; WeightedFilter::sample WeightedFilter.cpp:163-191: by weight):
 * state[1], weight[1] (may be NULL; importance filters only), counts[counts_len] (may be NULL).  The adapters build the
 * BAPOMDPState / FBAPOMDPState a reference planner borrows from it (RBAPOUCT.cpp:89-107). */
int fba_belief_get_particle(fba_ctx* ctx, int32_t slot, int32_t index, int32_t* state, double* weight, float* counts);
/* the second filter of the reinvigoration belief (ReinvigoratingRejectionSampling.hpp:
 * _fully_connected_belief) or of the cheating belief (CheatingReinvigoration.hpp:
 * _correct_structured_belief), for tests */
int fba_belief_get_fully_connected(fba_ctx* ctx, int32_t slot, int32_t* state, float* counts);
/* the nested belief's flat filters of domain states (NestedBelief.hpp: the FlatFilter<State const*> of every top
 * particle): states[particles][particles^2]; fba_belief_get returns the count particles and their weights */
int fba_belief_get_nested(fba_ctx* ctx, int32_t slot, int32_t* states);
/* the incubator belief's shadow filter (StructureIncubatorSampling.hpp: _shadow_belief), for tests */
int fba_belief_get_shadow(fba_ctx* ctx, int32_t slot, int32_t* state, double* weight, float* counts);
/* per-slot record of the last select_action / belief_update (root statistics, rejection count,
 * belief checksum) */
int fba_last_step_info(fba_ctx* ctx, fba_trace_rec* recs /* [slots] */);

/* ---- whole-experiment interface ------------------------------------------------------- */

/* experiment::planning::run(conf)            src/experiments/PlanningExperiment.cpp:27-55
 * stats[1]; all `runs` runs execute concurrently, `slots` at a time. */
int fba_run_planning(fba_ctx* ctx, fba_stat* stats);
/* experiment::bapomdp::run(bapomdp, conf)    src/experiments/BAPOMDPExperiment.cpp:32-78
 * stats[episodes]: statistic e accumulates the return of episode e over runs, in run order. */
int fba_run_bapomdp(fba_ctx* ctx, fba_stat* stats);
/* throughput driver used by bench.py: advance every slot by `ticks` real time-steps
 * (search + env step + belief update), restarting episodes/runs as they finish. */
int fba_run_ticks(fba_ctx* ctx, int32_t ticks);

/* per-(run, episode) discounted returns of the last fba_run_*: returns[runs * episodes] */
int fba_get_returns(const fba_ctx* ctx, double* returns, int32_t* lengths);
int fba_get_counters(fba_ctx* ctx, fba_counters* out);
/* {episodes finished, sum of returns, sum of squared returns} over all slots since fba_create:
 * what a multi-GPU job all-reduces (the pooled merge of analysis/preprocess/merge_result_files.py:60-78) */
int fba_get_return_sums(fba_ctx* ctx, double* out /* [3] */);
int fba_get_kernel_times(fba_ctx* ctx, fba_kernel_time* out /* [FBA_K_COUNT] */);
int fba_reset_kernel_times(fba_ctx* ctx);
int fba_trace_count(const fba_ctx* ctx);
int fba_get_trace(const fba_ctx* ctx, fba_trace_rec* out, int32_t cap);
/* cfg.trace = 2: hist[i][s] = how many particles of the filter held domain state s after the belief update of trace record i
 * (the order of fba_get_trace) -- what FlatFilter::toString (src/beliefs/particle_filters/FlatFilter.cpp:70-94) prints at -v 3
 * ("Status of rejection sampling filter after update", RejectionSampling.cpp:39, BARejectionSampling.cpp:46); all zero for a
 * record without an update (terminal step).  out[cap][FBA_TRACE_HIST_BINS]. */
#define FBA_TRACE_HIST_BINS 64
int fba_get_trace_hist(const fba_ctx* ctx, uint32_t* out, int32_t cap);

/* diagnostic: out[i] = u * sqrt(L[i] / n[i]) evaluated on the device, to check that the
 * engine's fp64 divide and square root round like the host's (they must, for UCB parity) */
int fba_selftest_ucb(fba_ctx* ctx, const double* L, const int32_t* n, int32_t count, double u, double* out);

/* BABNModel::LogBDScore(prior) (src/bayes-adaptive/states/factored/BABNModel.cpp:451-478 over DBNNode::LogBDScore,
 * DBNNode.cpp:82-117): the log Bayesian-Dirichlet score of one particle's counts against a prior blob of the same
 * structure -- what the structure-learning beliefs (MHwithinGibbs.cpp:334-395, MHNIPS2018.cpp) accept and reject models by.
 * Evaluated on the device with the engine's deterministic lgamma; fba_selftest_lgamma exposes that function. */
int fba_log_bd_score(fba_ctx* ctx, const float* counts, const float* prior, double* out);
int fba_selftest_lgamma(fba_ctx* ctx, const double* x, int32_t count, double* out);

/* utils::Statistic::add / var / stder, exported so hosts merge returns exactly as the
 * reference does */
void fba_stat_add(fba_stat* s, double v);
double fba_stat_var(const fba_stat* s);
double fba_stat_stder(const fba_stat* s);

#ifdef __cplusplus
}
#endif
#endif
