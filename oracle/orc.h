/*
 * oracle/orc.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the BA-POMCP hot path of samkatt/fba-pomdp:
 * domains, Bayes-adaptive count models, POUCT / RBAPOUCT tree search, rejection- and
 * importance-sampling particle filters, and the episode / experiment loops that call them.
 * Every function in orc.c cites the reference file:line it restates.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library,
 * and only as the checker / reported baseline.  The product path (fba_pomdp_amd/) must never
 * link, import or call it.
 *
 * Pinning status: see oracle/README.md.
 */
#ifndef ORC_H
#define ORC_H

#include <stdint.h>

#include "orc_rng.h"

#ifdef __cplusplus
extern "C" {
#endif

/* domains (reference src/domains) */
enum {
    ORC_DOM_TIGER_EPISODIC    = 0,
    ORC_DOM_TIGER_CONTINUOUS  = 1,
    ORC_DOM_FTIGER_EPISODIC   = 2,
    ORC_DOM_FTIGER_CONTINUOUS = 3,
    ORC_DOM_GRIDWORLD         = 4,
    ORC_DOM_COLLISION_AVOID   = 5,
    /* 6 is the HIP engine's centered-collision-avoidance (here: ca_centered) */
    ORC_DOM_SYSADMIN_INDEPENDENT = 7, /* -D independent-sysadmin --size N */
    ORC_DOM_SYSADMIN_LINEAR      = 8, /* -D linear-sysadmin --size N      */
    ORC_DOM_COFFEE               = 9, /* -D coffee            (planning only) src/domains/coffee/CoffeeProblem.cpp */
    ORC_DOM_COFFEE_BOUTILIER     = 10, /* -D boutilier-coffee                                                     */
    ORC_DOM_AGR                  = 11  /* -D agr (AGR(10), planning with rejection sampling only) src/domains/agr/AGR.cpp */
};
/* simulator model */
enum { ORC_MODEL_POMDP = 0, ORC_MODEL_BA_TABLE = 1, ORC_MODEL_BA_FACTORED = 2 };
/* belief */
/* REINVIGORATION = beliefs::bayes_adaptive::factored::ReinvigoratingRejectionSampling (-B reinvigoration) */
/* CHEATING = beliefs::bayes_adaptive::prototypes::CheatingReinvigoration (-B cheating-reinvigoration) */
enum { ORC_BELIEF_REJECTION = 0, ORC_BELIEF_IMPORTANCE = 1, ORC_BELIEF_REINVIGORATION = 2, ORC_BELIEF_CHEATING = 3,
       ORC_BELIEF_POINT = 4, /* -B point_estimate: src/beliefs/point_estimation/PointEstimation.cpp, bayes-adaptive/BAPointEstimation.cpp */
       ORC_BELIEF_MH_GIBBS = 5, /* -B mh-within-gibbs: src/beliefs/bayes-adaptive/factored/MHwithinGibbs.cpp (belief_option 1 = "rs") */
       ORC_BELIEF_MH_NIPS = 6,  /* -B mh-nips: src/beliefs/bayes-adaptive/factored/MHNIPS2018.cpp */
       ORC_BELIEF_INCUBATOR = 8, /* -B incubator: src/beliefs/bayes-adaptive/factored/StructureIncubatorSampling.cpp (--resample-amount, --threshold) */
       ORC_BELIEF_NESTED = 7    /* -B nested: src/beliefs/bayes-adaptive/NestedBelief.cpp (`particles` count particles, each with particles^2 domain states) */ };
/* floating-point summation order of the importance-sampling filter:
 * REF = the reference's sequential loops; DEV = the HIP engine's fixed reduction tree */
enum { ORC_ARITH_REF = 0, ORC_ARITH_DEV = 1 };
/* planner (reference -P) */
enum { ORC_PLANNER_POUCT = 0, ORC_PLANNER_RANDOM = 1, ORC_PLANNER_TS = 2 /* -P ts: TSPlanner / BATSPlanner */ };
/* FBA structure prior (reference FBAConf::structure_prior) */
enum { ORC_SP_NONE = 0, ORC_SP_UNIFORM = 1, ORC_SP_MATCH_UNIFORM = 2, ORC_SP_FULLY_CONNECTED = 3 };

#define ORC_MAX_ACTIONS 24

typedef struct orc_config {
    int32_t domain;
    int32_t size;   /* --size   */
    int32_t width;  /* --width  */
    int32_t height; /* --height */
    int32_t model;
    int32_t belief;
    int32_t particles;     /* --particle-amount */
    int32_t sims;          /* -s   */
    int32_t max_depth;     /* --mcts-max-depth; < 0 => horizon (ArgumentParser.cpp:37-40) */
    int32_t horizon;       /* -H   */
    double exploration;    /* -u   */
    double discount;       /* -d   */
    int32_t runs;          /* --runs */
    int32_t episodes;      /* --episodes (BA only) */
    float noise;           /* --noise */
    float counts_total;    /* -C */
    int32_t structure_prior;
    int32_t rng_mode;      /* ORC_RNG_MT | ORC_RNG_PHILOX */
    int32_t arith;         /* ORC_ARITH_REF | ORC_ARITH_DEV */
    uint64_t philox_seed;
    char seed_str[64];     /* --seed (MT mode) */
    int32_t run_offset;    /* first run index (Philox streams; episode sharding) */
    int32_t trace;         /* 1 = record orc_trace_rec per real step */
    int32_t planner;       /* ORC_PLANNER_* */
    int32_t ca_centered;   /* collision avoidance: 1 = centered-collision-avoidance, 0 = random-collision-avoidance */
    int32_t dirichlet_regular; /* --dirichlet_sampling_method regular (0 = expected, the default) */
    int32_t resample_amount;   /* --resample-amount: particles bred per update (reinvigoration belief) / copied per cheat */
    double threshold;          /* --threshold: log-likelihood below which the cheating belief cheats (< 0) */
    int32_t belief_option;     /* --belief-option: mh-within-gibbs 0 = message passing (MSG), 1 = "rs" (rejection-sampled state history) */
} orc_config;

/* One record per real time-step; the HIP engine emits the same layout (fba_trace_rec). */
typedef struct orc_trace_rec {
    int32_t run, episode, t;
    int32_t action, state, obs; /* env s' and o after the step */
    int32_t terminal;
    int32_t n_nodes, tree_depth; /* POUCT.cpp:95-100 */
    int32_t update_count;        /* rejection attempts (RejectionSampling.hpp:68); -1 if no update */
    int32_t root_n[ORC_MAX_ACTIONS];
    double root_q[ORC_MAX_ACTIONS];
    double reward;
    double weight_total;   /* IS: total weight before normalisation; 0 otherwise */
    uint64_t belief_hash;  /* position-sensitive hash of the particle set after the update */
} orc_trace_rec;

typedef struct orc_stat { /* utils::Statistic, src/utils/Statistic.cpp:5-46 */
    double count, mean, m2;
} orc_stat;

typedef struct orc_result {
    uint64_t sim_steps;    /* simulator.step calls made by the planner (tree + rollout) */
    uint64_t belief_steps; /* simulator.step calls made by the belief update */
    uint64_t env_steps;
    double seconds;
    int32_t n_trace;
} orc_result;

typedef struct orc_ctx orc_ctx;

orc_ctx* orc_create(const orc_config* cfg);
void orc_destroy(orc_ctx* c);
const char* orc_error(const orc_ctx* c);

/* experiment::planning::run (PlanningExperiment.cpp:27-55): stats[1] */
int orc_run_planning(orc_ctx* c, orc_stat* stats, orc_result* res);
/* experiment::bapomdp::run (BAPOMDPExperiment.cpp:32-78): stats[episodes] */
int orc_run_bapomdp(orc_ctx* c, orc_stat* stats, orc_result* res);

const orc_trace_rec* orc_trace(const orc_ctx* c);

/* model introspection (prior tables), for fixtures and for feeding fba_set_model_* */
int orc_domain_sizes(const orc_ctx* c, int32_t* S, int32_t* A, int32_t* O);
int orc_counts_len(const orc_ctx* c);
/* draws one prior particle's count blob with the ctx RNG positioned by the caller */
int orc_prior_counts(orc_ctx* c, float* out);

/* stand-alone pieces for unit tests */
double orc_stat_var(const orc_stat* s);
double orc_stat_stder(const orc_stat* s);
void orc_stat_add(orc_stat* s, double v);
int orc_sample_from_mult_f(orc_rng* g, const float* row, int n, double total);
int orc_sample_expected_mult(orc_rng* g, const float* row, int n);
void orc_expected_mult(const float* row, int n, float* out);
orc_rng* orc_ctx_rng(orc_ctx* c);

/* tiger true-environment step exposed for golden-vector tests; returns terminal */
int orc_env_step(orc_ctx* c, int32_t* s, int32_t a, int32_t* o, double* r);
int orc_env_start(orc_ctx* c);
int orc_random_action(orc_ctx* c, int32_t s);

/* filter / planner level entry points (golden-vector and step-level parity tests) */
void orc_belief_initiate(orc_ctx* c);
void orc_belief_update(orc_ctx* c, int32_t a, int32_t o);
double orc_is_update(orc_ctx* c, int32_t a, int32_t o);
void orc_is_resample(orc_ctx* c);
void orc_belief_reset_domain_state(orc_ctx* c);
int orc_select_action(orc_ctx* c, int hist_len, orc_trace_rec* rec);
uint64_t orc_belief_hash(orc_ctx* c);
int orc_last_update_count(const orc_ctx* c);
void orc_belief_get(const orc_ctx* c, int32_t* s, double* w, float* cnt);
void orc_belief_get_fc(const orc_ctx* c, int32_t* s, float* cnt);
void orc_belief_get_nested(const orc_ctx* c, int32_t* states); /* [particles][particles^2] */
void orc_belief_get_shadow(const orc_ctx* c, int32_t* s, double* w, float* cnt); /* the incubator's weighted shadow filter */
void orc_least_likely(const double* w, int size, int n, int* out); /* WeightedFilter::leastLikely (orc_heap.cpp) */
int orc_marginalize(orc_ctx* c, const float* cnt, const uint32_t* new_masks, float* out);
void orc_belief_set(orc_ctx* c, const int32_t* s, const double* w, const float* cnt);
int orc_model_step(orc_ctx* c, float* cnt, int32_t* s, int32_t a, int32_t* o, double* r, int update);
double orc_model_obs_prob(orc_ctx* c, const float* cnt, int32_t new_s, int32_t a, int32_t o);
double orc_dev_scan(const double* w, int n, double* incl);
int orc_ftiger_set_structure(orc_ctx* c, float* cnt, uint32_t mask);
/* regular-Dirichlet building blocks (reference src/utils/random.cpp:146-304) for golden tests */
double orc_gamma(orc_ctx* c, double shape);
int orc_sample_sampled_mult(orc_ctx* c, const float* dir, int n);
void orc_sample_mult(orc_ctx* c, const float* dir, int n, float* out);
void orc_chance_add_visit(int32_t* n, double* q, double ret);
int orc_ext_terminal(orc_ctx* c, int32_t s, int32_t a, int32_t ns);
double orc_ext_reward(orc_ctx* c, int32_t s, int32_t a, int32_t ns);
double orc_det_lgamma(double x);
double orc_log_bd_score(orc_ctx* c, const float* cnt, const float* prior);
double orc_det_log(double x);
double orc_det_exp(double x);

#ifdef __cplusplus
}
#endif
#endif
