/*
 * oracle/orc.c -- TEST INFRASTRUCTURE ONLY (see orc.h).
 *
 * Restates, function by function, the reference's hot path.  "ref:" comments give the
 * reference file:line each block follows (paths relative to /root/reference).
 */
#include "orc.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ types */

typedef struct particle {
    int32_t s;    /* domain state index                                     */
    float* cnt;   /* Dirichlet counts of this particle (NULL for plain POMDP) */
    double w;     /* importance weight (WeightedFilter only)                */
} particle;

typedef struct simstate { /* what the planner / filters hand to simulator.step */
    int32_t s;
    float* cnt; /* read-only in KeepCounts mode */
} simstate;

typedef struct tree {
    int32_t* visits; /* ActionNode::_visit_count           [max_nodes]     */
    int32_t* cn;     /* ChanceNode::_visit_count           [max_nodes * A] */
    double* cq;      /* ChanceNode::_q                     [max_nodes * A] */
    int32_t* child;  /* dense child table [max_nodes*A*O] or NULL when hashed */
    uint64_t* hkey;  /* open-addressing hash: key = (node*A+a)*O+o+1 */
    int32_t* hval;
    uint64_t hmask;
    int32_t n_nodes, max_nodes;
    int32_t max_tree_depth, tree_depth;
} tree;

/* Factored Bayes-adaptive model (BABNModel + DBNNode), stored in a fixed "max layout":
 * node (a, f) owns a region of the count blob big enough for its largest allowed parent set;
 * a particle's actual parents are a bit mask over the node's `maxp` list (fixed for the ctx, or
 * per particle in mask word `var` stored after the counts); rows are indexed compactly by the
 * actual parents, exactly as DBNNode::cptIndex (DBNNode.cpp:171-205) does. */
#define ORC_MAXF 8
typedef struct fnode {
    int32_t off, out, nmax, var;
    int32_t maxp[ORC_MAXF];
    uint32_t fixed_mask;
} fnode;
typedef struct fdesc {
    int32_t FS, FO, nvar, ncounts;
    int32_t Ssz[ORC_MAXF], Osz[ORC_MAXF], Sstep[ORC_MAXF], Ostep[ORC_MAXF];
    fnode* T; /* [A * FS] */
    fnode* O; /* [A * FO] */
} fdesc;

struct orc_ctx {
    orc_config cfg;
    fdesc fd;
    orc_rng rng;
    char err[256];
    int32_t S, A, O;
    int32_t tiger_K;   /* factored tiger: number of irrelevant features */
    /* gridworld (src/domains/gridworld/GridWorld.cpp) */
    int32_t gw_N, gw_G, gw_nslow;
    int32_t gw_goal[16][2], gw_slow[8][2];
    float gw_disp[32]; /* _obs_displacement_probs */
    /* sysadmin (src/domains/sysadmin/SysAdmin.cpp); bit c of the state = computer c operational */
    int32_t sys_N;
    double sys_keep[3]; /* (1 - fail_prob) * pow(1 - fail_neighbour_factor, #failing neighbours) */
    /* collision avoidance (src/domains/collision-avoidance/CollisionAvoidance.cpp) */
    int32_t ca_W, ca_H, ca_n, ca_random_start, ca_Hn; /* Hn = H^n */
    double ca_err[16];        /* _observation_error_probability[d] */
    double ca_phi[20];        /* Phi(k + .5), k = -9..9: rounded-normal thresholds (Philox mode) */
    float ca_start_v;         /* probability of each start state (float) */
    double ca_start_total;    /* categoricalDistr::_total */
    int32_t ca_start_i0, ca_start_cnt;
    /* ziggurat tables of rnd::initiate() (random.cpp:47-74) */
    unsigned long zig_ul[128];
    double zig_wn[128], zig_fn[128];
    float nd_saved;           /* std::normal_distribution<float>::_M_saved */
    int nd_saved_available;
    int32_t ncnt;      /* floats per particle count blob */
    int32_t phi_len;   /* tabular: S*A*S */
    float* prior;      /* tabular prior blob (phi then psi) */
    /* MH-within-Gibbs belief: the run's (action, observation) history by episode, the log-likelihood, scratch */
    int16_t *mh_a, *mh_o;
    int32_t* mh_ep_len;
    int mh_n_ep;
    double log_lik;
    float *mh_prior, *mh_model, *mh_new, *mh_T, *mh_O;
    double *mh_msg, *mh_probs;
    int32_t* mh_seq;
    /* nested belief: per count particle a flat filter of nest_m domain states; one update's accepted states */
    int32_t *nest_s, *nest_new;
    int nest_m;
    int32_t ts_state;
    double* log1p_tab; /* log1p(m), m < sims (POUCT.cpp:330-338 factorised) */
    double gamma;
    tree tr;
    /* belief */
    particle* P;
    particle* Pnew;
    float* pool;
    float* pool_new;
    /* ReinvigoratingRejectionSampling::_fully_connected_belief (second FlatFilter) */
    particle* F;
    particle* Fnew;
    float* fpool;
    float* fpool_new;
    float* breed_tmp;
    /* StructureIncubatorSampling::_shadow_belief (a WeightedFilter of bred particles) */
    particle* Sh;
    particle* Shnew;
    float* shpool;
    float* shpool_new;
    double sh_total;
    double likelihood; /* CheatingReinvigoration::_likelihood */
    double total_w; /* WeightedFilter::_total_weight */
    double* wscratch;
    double* wscan;
    /* counters */
    uint64_t sim_steps, belief_steps, env_steps;
    uint64_t* step_counter; /* which counter sim_step bumps */
    /* trace */
    orc_trace_rec* trace;
    int32_t n_trace, cap_trace;
    /* per-selectAction outputs */
    int32_t last_update_count;
    int32_t point; /* belief = point estimate (run as a rejection filter of one particle) */
    double last_weight_total;
};

/* ------------------------------------------------------------------ utils */

/* ref: src/utils/Statistic.cpp:5-46 */
void orc_stat_add(orc_stat* s, double v)
{
    double delta, delta2;
    s->count += 1;
    delta = v - s->mean;
    s->mean += delta / s->count;
    delta2 = v - s->mean;
    s->m2 += delta * delta2;
}
double orc_stat_var(const orc_stat* s) { return s->count < 2 ? 0 : s->m2 / (s->count - 1); }
double orc_stat_stder(const orc_stat* s)
{
    return s->count < 2 ? 0 : sqrt(orc_stat_var(s) / s->count);
}

/* ref: src/utils/random.hpp:93-115  sampleFromMult<float const>: CDF accumulated in float,
 * compared against a double threshold; falls through to n-1 */
int orc_sample_from_mult_f(orc_rng* g, const float* mult, int n, double total)
{
    double p  = orc_u01(g) * total;
    float sum = mult[0];
    int i;
    for (i = 1; i < n; ++i) {
        if (p < sum) return i - 1;
        sum += mult[i];
    }
    return n - 1;
}

/* ref: src/utils/random.cpp:244-255  sampleFromExpectedMult: total accumulated in double */
int orc_sample_expected_mult(orc_rng* g, const float* dir, int n)
{
    double total = dir[0];
    int i;
    for (i = 1; i < n; ++i) total += dir[i];
    return orc_sample_from_mult_f(g, dir, n, total);
}

/* ref: src/utils/random.cpp:257-279  expectedMult: sum and division in float */
void orc_expected_mult(const float* dir, int n, float* out)
{
    float sum = dir[0];
    int i;
    for (i = 1; i < n; ++i) sum += dir[i];
    if (sum <= 1e-300) {
        for (i = 0; i < n; ++i) out[i] = 0;
        return;
    }
    for (i = 0; i < n; ++i) out[i] = dir[i] / sum;
}

static uint64_t mix64(uint64_t x)
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

static uint64_t particle_hash(uint64_t i, int32_t s, double w, const float* cnt, int ncnt)
{
    uint64_t h, wb;
    int k;
    memcpy(&wb, &w, 8);
    h = mix64(i * 0x9E3779B97F4A7C15ull + (uint64_t)(uint32_t)s);
    h = mix64(h ^ wb);
    for (k = 0; k < ncnt; ++k) {
        uint32_t b;
        memcpy(&b, &cnt[k], 4);
        h = mix64(h ^ ((uint64_t)b + ((uint64_t)k << 32)));
    }
    return h;
}

/* ------------------------------------------------------------------ domains */

static int is_tiger(int d) { return d == ORC_DOM_TIGER_EPISODIC || d == ORC_DOM_TIGER_CONTINUOUS; }
static int is_ftiger(int d)
{
    return d == ORC_DOM_FTIGER_EPISODIC || d == ORC_DOM_FTIGER_CONTINUOUS;
}
static int is_episodic(int d)
{
    return d == ORC_DOM_TIGER_EPISODIC || d == ORC_DOM_FTIGER_EPISODIC;
}
static int is_grid(int d) { return d == ORC_DOM_GRIDWORLD; }
static int is_ca(int d) { return d == ORC_DOM_COLLISION_AVOID; }
static int is_weighted(const orc_ctx* c);
static int is_coffee(int d) { return d == ORC_DOM_COFFEE || d == ORC_DOM_COFFEE_BOUTILIER; }
static int is_agr(int d) { return d == ORC_DOM_AGR; }
#define AGR_N 10 /* factory::makeEnvironment: new domains::AGR(10), Environment.cpp:32-33 */
static int is_sys(int d) { return d == ORC_DOM_SYSADMIN_INDEPENDENT || d == ORC_DOM_SYSADMIN_LINEAR; }

/* ---- collision avoidance.  ref: src/domains/collision-avoidance/CollisionAvoidance.cpp
 * state index = (x*H + y)*H^n + project(obstacles) (ctor :77-92); observation = project(observed
 * obstacle rows); actions MOVE_DOWN, STAY, MOVE_UP (hpp:91) */
static double normal_cdf(double x) { return .5 + .5 * erf(x / (1 * sqrt(2))); } /* rnd::normal::cdf random.cpp:119-124 */
static int ca_keep(const orc_ctx* c, int y) { return y < 0 ? 0 : (y > c->ca_H - 1 ? c->ca_H - 1 : y); }
static void ca_setup(orc_ctx* c, int W, int H, int n, int random_start)
{
    int d, k;
    c->ca_W = W; c->ca_H = H; c->ca_n = n; c->ca_random_start = random_start;
    c->ca_Hn = 1;
    for (k = 0; k < n; ++k) c->ca_Hn *= H;
    for (d = 0; d < H; ++d) c->ca_err[d] = normal_cdf(d + .5) - normal_cdf(d - .5); /* ctor :100-104 */
    for (k = 0; k < 19; ++k) c->ca_phi[k] = normal_cdf((k - 9) + .5);
    if (random_start) { /* ctor :121-133: every state with x = W-1 */
        c->ca_start_v   = (float)(1.f / pow(H, n + 1));
        c->ca_start_i0  = (W - 1) * H * c->ca_Hn;
        c->ca_start_cnt = H * c->ca_Hn;
        c->ca_start_total = 0;
        for (k = 0; k < c->ca_start_cnt; ++k) c->ca_start_total += c->ca_start_v; /* setRawValue: _total += v - 0 */
    } else {            /* :134-142: agent and obstacles in the middle row */
        int obs = 0;
        for (k = 0; k < n; ++k) obs = obs * H + H / 2;
        c->ca_start_v = 1; c->ca_start_total = 1; c->ca_start_cnt = 1;
        c->ca_start_i0 = ((W - 1) * H + H / 2) * c->ca_Hn + obs;
    }
}
/* std::normal_distribution<float>(0,1) as GNU libstdc++ 11 implements it (bits/random.tcc
 * normal_distribution::operator(): Marsaglia polar method with one saved variate;
 * generate_canonical<float,24> takes one 32-bit engine word), then static_cast<int>(std::round()).
 * Philox mode: the same discrete distribution by inverse CDF on Phi(k + .5). */
static float mt_canonical_f(orc_rng* g)
{
    float ret = (float)orc_mt_next(g) / 4294967296.0f;
    if (ret >= 1.0f) ret = nextafterf(1.0f, 0.0f);
    return ret;
}
static int ca_rounded_normal(orc_ctx* c)
{
    if (c->rng.mode == ORC_RNG_MT) {
        float ret;
        if (c->nd_saved_available) {
            c->nd_saved_available = 0;
            ret = c->nd_saved;
        } else {
            float x, y, r2, mult;
            do {
                x  = 2.0f * mt_canonical_f(&c->rng) - 1.0f;
                y  = 2.0f * mt_canonical_f(&c->rng) - 1.0f;
                r2 = x * x + y * y;
            } while (r2 > 1.0f || r2 == 0.0f);
            mult = sqrtf(-2 * logf(r2) / r2);
            c->nd_saved = x * mult;
            c->nd_saved_available = 1;
            ret = y * mult;
        }
        ret = ret * 1.0f + 0.0f;
        return (int)roundf(ret);
    } else {
        double u = orc_u01(&c->rng);
        int k;
        for (k = 0; k < 19; ++k)
            if (u < c->ca_phi[k]) return k - 9;
        return 10;
    }
}

/* ---- gridworld.  ref: src/domains/gridworld/GridWorld.cpp
 * state index = x*N*G + y*G + g (positionsToIndex :329-340); actions UP, RIGHT, DOWN, LEFT */
static void gw_setup(orc_ctx* c, int N)
{
    int edge = N - 1, start, i, G = 0, ns = 0;
    double prob;
    c->gw_N = N;
    /* goalLocations :124-152 */
    start = (N < 5) ? N - 2 : (N < 7) ? N - 3 : N - 4;
    for (i = start; i < N - 1; ++i) {
        c->gw_goal[G][0] = i; c->gw_goal[G][1] = edge; G++;
        c->gw_goal[G][0] = edge; c->gw_goal[G][1] = i; G++;
    }
    c->gw_goal[G][0] = edge; c->gw_goal[G][1] = edge; G++;
    if (N > 3) { c->gw_goal[G][0] = edge - 1; c->gw_goal[G][1] = edge - 1; G++; }
    if (N > 6) {
        c->gw_goal[G][0] = edge - 2; c->gw_goal[G][1] = edge - 1; G++;
        c->gw_goal[G][0] = edge - 1; c->gw_goal[G][1] = edge - 2; G++;
    }
    c->gw_G = G;
    /* generateSlowLocations :78-103 */
    if (N > 5) { c->gw_slow[ns][0] = 1; c->gw_slow[ns][1] = 1; ns++; }
    if (N == 3) { c->gw_slow[ns][0] = 1; c->gw_slow[ns][1] = 1; ns++; }
    else if (N < 7) {
        c->gw_slow[ns][0] = edge - 1; c->gw_slow[ns][1] = edge - 2; ns++;
        c->gw_slow[ns][0] = edge - 2; c->gw_slow[ns][1] = edge - 1; ns++;
    } else {
        c->gw_slow[ns][0] = edge - 1; c->gw_slow[ns][1] = edge - 3; ns++;
        c->gw_slow[ns][0] = edge - 3; c->gw_slow[ns][1] = edge - 1; ns++;
        c->gw_slow[ns][0] = edge - 2; c->gw_slow[ns][1] = edge - 2; ns++;
    }
    c->gw_nslow = ns;
    /* _obs_displacement_probs, ctor :60-70: {.8, .1, .05, ..., last repeated}, N entries */
    c->gw_disp[0] = (float)(1 - .2);
    prob = .2;
    for (i = 1; i < N - 1; ++i) { prob *= .5; c->gw_disp[i] = (float)prob; }
    c->gw_disp[N - 1] = (float)prob;
}
static int gw_slow_at(const orc_ctx* c, int x, int y)
{
    int i;
    for (i = 0; i < c->gw_nslow; ++i)
        if (c->gw_slow[i][0] == x && c->gw_slow[i][1] == y) return 1;
    return 0;
}
/* applyMove :342-364 */
static void gw_move(const orc_ctx* c, int a, int* x, int* y)
{
    int N = c->gw_N;
    switch (a) {
        case 0: if (*y != N - 1) (*y)++; break;
        case 2: if (*y != 0) (*y)--; break;
        case 1: if (*x != N - 1) (*x)++; break;
        case 3: if (*x != 0) (*x)--; break;
    }
}
/* obsDisplProb :164-185 (float result, double intermediate products) */
static float gw_obs_displ_prob(const orc_ctx* c, int loc, int observed)
{
    int disp = abs(loc - observed), i;
    float res = (disp == 0) ? (float)(1 - .2) : (float)(c->gw_disp[disp] * .5);
    if (observed == c->gw_N - 1 || observed == 0)
        for (i = disp + 1; i < c->gw_N; ++i) res = (float)(res + c->gw_disp[i] * .5);
    return res;
}

/* SysAdmin parameters (SysAdmin.cpp:12): fail .025f, observe .95f, reboot success .95f, reboot
 * cost 1.0f, neighbour factor .075f -- floats, promoted where the reference promotes them */
#define SYS_FAIL_PROB .025f
#define SYS_OBSERVE_PROB .95f
#define SYS_REBOOT_RATE .95f
#define SYS_NEIGHBOUR_FACTOR .075f
static void sys_setup(orc_ctx* c, int n)
{
    int k;
    c->sys_N = n;
    for (k = 0; k < 3; ++k) /* float * pow(double(float), n): SysAdmin.cpp:116-118 */
        c->sys_keep[k] = (double)(1 - SYS_FAIL_PROB) * pow((double)(1 - SYS_NEIGHBOUR_FACTOR), (double)k);
}
/* SysAdmin::numFailingNeighbours (SysAdmin.cpp:221-246).  isOperational(n) reads bit n of the index
 * whatever n is, which matters where the flat prior passes an action index for `comp` */
static int sys_failing_neighbours(const orc_ctx* c, int comp, int32_t s)
{
    int n = 0;
    if (c->cfg.domain == ORC_DOM_SYSADMIN_INDEPENDENT) return 0;
    if (comp > 0 && !((s >> (comp - 1)) & 1)) n++;
    if (comp < c->sys_N - 1 && !((s >> (comp + 1)) & 1)) n++;
    return n;
}

/* ref: Tiger::sampleStartState src/domains/tiger/Tiger.cpp:16-19 (LEFT=0 iff boolean()),
 *      FactoredTiger::sampleStartState src/domains/tiger/FactoredTiger.cpp (uniform_int{0,S-1}) */
static int32_t domain_start(orc_ctx* c)
{
    if (is_tiger(c->cfg.domain)) return orc_bool(&c->rng) ? 0 : 1;
    if (is_ftiger(c->cfg.domain)) return orc_int(&c->rng, c->S);
    if (is_sys(c->cfg.domain)) return c->S - 1; /* SysAdmin::sampleStartState :102-105: all computers on, no draw */
    if (is_agr(c->cfg.domain)) /* AGR::sampleStartState :236-239: the j-th state with target_pos == 0 is goal -n + j */
        return (2 * AGR_N + 1) * orc_int(&c->rng, 2 * AGR_N + 1) + AGR_N;
    if (is_coffee(c->cfg.domain)) return orc_int(&c->rng, 32); /* CoffeeProblem::sampleStartState :62-65: integerDistribution(0, 32) */
    if (is_ca(c->cfg.domain)) { /* sampleStartState :270-273 -> categoricalDistr::sample -> sampleFromMult<float>(values, S, _total) */
        double p  = orc_u01(&c->rng) * c->ca_start_total;
        float sum = 0;
        int k;
        /* zero entries before the block leave the float sum at 0; inside the block it grows by v */
        for (k = 0; k < c->ca_start_cnt; ++k) {
            sum += c->ca_start_v;
            if (p < sum) return c->ca_start_i0 + k;
        }
        return (c->ca_start_i0 + c->ca_start_cnt == c->S) ? c->S - 1 : c->S - 1;
    }
    if (is_grid(c->cfg.domain)) { /* GridWorld::sampleStartState :260-266: start_locations = {{0,0}} */
        int agent = orc_slow_int(&c->rng, 0, 1);
        int goal  = orc_slow_int(&c->rng, 0, c->gw_G);
        (void)agent;
        return 0 * c->gw_N * c->gw_G + 0 * c->gw_G + goal;
    }
    return 0;
}

/* ref: Tiger::generateRandomAction Tiger.cpp:21-25 (uniform_int_distribution<int>{0,2}),
 *      FactoredTiger::generateRandomAction (uniform_int_distribution<int>{0,A-1}) */
static int32_t domain_random_action(orc_ctx* c, int32_t s)
{
    (void)s;
    if (is_tiger(c->cfg.domain) || is_ftiger(c->cfg.domain)) return orc_int(&c->rng, 3);
    if (is_grid(c->cfg.domain)) return orc_slow_int(&c->rng, 0, 4); /* GridWorld::generateRandomAction :220-226 */
    if (is_ca(c->cfg.domain)) return orc_int(&c->rng, 3); /* integerDistribution(0, NUM_ACTIONS) */
    if (is_sys(c->cfg.domain)) return orc_int(&c->rng, c->A); /* SysAdmin.cpp:167-170: integerDistribution(0, A) */
    if (is_agr(c->cfg.domain)) return orc_int(&c->rng, c->A); /* AGR.cpp:241-245: integerDistribution(0, 2n + 3) */
    if (is_coffee(c->cfg.domain)) return orc_bool(&c->rng) ? 1 : 0; /* CoffeeProblem.cpp:27-34: _actions.get((int)boolean()) */
    return 0;
}

/* true dynamics.  ref: Tiger::step Tiger.cpp:40-82, FactoredTiger::step FactoredTiger.cpp:77-122 */
static int domain_step(orc_ctx* c, int32_t* s, int32_t a, int32_t* o, double* r)
{
    int d = c->cfg.domain;
    if (is_tiger(d)) {
        int tiger_left = (*s == 0);
        if (a == 2) { /* OBSERVE */
            int correct = orc_u01(&c->rng) < .85;
            *r          = -1;
            *o          = ((correct ^ tiger_left) != 0) ? 1 : 0;
        } else {
            *r = (a == *s) ? 10 : -100;
            *o = orc_bool(&c->rng) ? 1 : 0; /* _observations.get((int)boolean()) */
            *s = orc_bool(&c->rng) ? 1 : 0; /* _states.get((int)boolean())       */
        }
        return is_episodic(d) && a != 2;
    }
    if (is_ftiger(d)) {
        int loc = (*s < c->S / 2) ? 0 : 1; /* tigerLocation: LEFT iff idx < S/2 */
        if (a == 2) {
            int correct = orc_u01(&c->rng) < .85;
            *r          = -1;
            *o          = ((correct ^ (loc == 0)) != 0) ? 1 : 0;
        } else {
            *r = (a == loc) ? 10 : -100;
            *o = orc_bool(&c->rng) ? 0 : 1; /* boolean() ? &_observations[0] : &_observations[1] */
            *s = orc_int(&c->rng, c->S);    /* sampleStartState() */
        }
        return is_episodic(d) && a != 2;
    }
    if (is_agr(d)) { /* AGR::step :247-305: no draws.  state = (2n+1)(goal+n) + pos+n; actions help(-n..n) = 0..2n, work 2n+1, observe 2n+2 */
        int n = AGR_N, goal = *s / (2 * n + 1) - n, pos = *s % (2 * n + 1) - n, helped = 0, diff, stp;
        if (a <= 2 * n) {
            helped = (a - n == goal && pos == goal);
            *r     = helped ? 100 : -100;
        } else
            *r = (a == 2 * n + 1) ? -5 : -10;
        diff = goal - pos;
        stp  = diff > 1 ? 1 : diff;
        stp  = stp < -1 ? -1 : stp;
        pos += stp;
        *s = (2 * n + 1) * (goal + n) + pos + n;
        *o = (a == 2 * n + 2) ? pos + n : 2 * n + 1; /* observe sees the NEW position, everything else sees "none" */
        return helped;
    }
    if (is_coffee(d)) { /* CoffeeProblem::step :67-148; masks CoffeeProblemIndices.hpp: rains 1, umbrella 2, wet 4, has coffee 8, wants coffee 16 */
        int boutilier = d == ORC_DOM_COFFEE_BOUTILIER;
        int32_t st = *s, ns = st;
        double reward = -.5;
        if (st & 4) reward = -1;
        if (st & 16) reward += (st & 8) ? 2 : -2;
        if (a == 0) { /* GetCoffee */
            if ((st & 1) && !(st & 2)) ns |= 4;
            if (orc_u01(&c->rng) < .9) ns |= 8;
            if (orc_u01(&c->rng) < (boutilier ? 0 : .9)) ns &= ~16;
            *o = 0; /* Want_Coffee */
        } else { /* CheckCoffee */
            if (orc_u01(&c->rng) < (boutilier ? 0 : .3)) ns &= ~8;
            if (orc_u01(&c->rng) < (boutilier ? 0 : .3)) ns |= 16;
            if (ns & 16) *o = (orc_u01(&c->rng) < .8) ? 0 : 1;
            else *o = (orc_u01(&c->rng) < .9) ? 1 : 0;
        }
        *s = ns;
        *r = reward;
        return 0;
    }
    if (is_sys(d)) { /* SysAdmin::step :107-152 */
        int N = c->sys_N, rebooting = a >= N, op = rebooting ? a - N : a, k;
        int32_t index = *s;
        for (k = 0; k < N; ++k) /* one draw per computer, failing ones included; neighbours of the OLD state */
            if (orc_u01(&c->rng) > c->sys_keep[sys_failing_neighbours(c, k, *s)]) index &= ~(1 << k);
        if (rebooting && orc_u01(&c->rng) < (double)SYS_REBOOT_RATE) index |= 1 << op;
        *s = index;
        *o = ((orc_u01(&c->rng) < (double)SYS_OBSERVE_PROB) == (((index >> op) & 1) != 0)) ? 1 : 0; /* OPERATIONAL = 1 */
        *r = (double)((float)__builtin_popcount((unsigned)index) - 1.0f * (float)rebooting);
        return 0;
    }
    if (is_ca(d)) { /* CollisionAvoidance::step :236-270, moveObstacle :330-338, reward :196-208 */
        int H = c->ca_H, n = c->ca_n, Hn = c->ca_Hn, k;
        int x = *s / (H * Hn), y = (*s / Hn) % H, obs = *s % Hn;
        int b[8], ob[8], nx = x - 1, ny = ca_keep(c, y + a - 1), nobs = 0, oobs = 0, crashed;
        for (k = n - 1; k >= 0; --k) { b[k] = obs % H; obs /= H; }
        for (k = 0; k < n; ++k) {
            double prob = orc_u01(&c->rng);
            int m = (prob < .5) ? 1 : (prob > .5 * (1 + .5)) ? 2 : 0;
            b[k] = ca_keep(c, b[k] + m - 1);
        }
        for (k = 0; k < n; ++k) nobs = nobs * H + b[k];
        *s = (nx * H + ny) * Hn + nobs;
        for (k = 0; k < n; ++k) ob[k] = ca_keep(c, b[k] + ca_rounded_normal(c));
        for (k = 0; k < n; ++k) oobs = oobs * H + ob[k];
        *o = oobs;
        crashed = nx < n && nx >= 0 && ny == b[nx];
        *r = crashed ? -1000 : (a == 1 ? 0 : -1);
        return crashed || nx == 0;
    }
    if (is_grid(d)) { /* GridWorld::step :272-304, generateObservation :366-394 */
        int N = c->gw_N, G = c->gw_G;
        int x = *s / (N * G), y = (*s / G) % N, g = *s % G;
        int slow = gw_slow_at(c, x, y);
        int ok   = orc_u01(&c->rng) < (slow ? .15 : .95);
        int nx = x, ny = y, ng = g, found, dx, dy, ox, oy;
        if (ok) gw_move(c, a, &nx, &ny);
        found = (c->gw_goal[g][0] == x && c->gw_goal[g][1] == y);
        if (found) ng = orc_slow_int(&c->rng, 0, G);
        *s = nx * N * G + ny * G + ng;
        *r = found ? 1 : 0;
        dx = orc_sample_from_mult_f(&c->rng, c->gw_disp, N, 1);
        dy = orc_sample_from_mult_f(&c->rng, c->gw_disp, N, 1);
        ox = orc_bool(&c->rng) ? (nx - dx > 0 ? nx - dx : 0) : (nx + dx < N - 1 ? nx + dx : N - 1);
        oy = orc_bool(&c->rng) ? (ny - dy > 0 ? ny - dy : 0) : (ny + dy < N - 1 ? ny + dy : N - 1);
        *o = ox * N * G + oy * G + ng;
        return found;
    }
    return 1;
}

/* ref: Tiger::computeObservationProbability Tiger.cpp:27-38; FactoredTiger ditto */
static double domain_obs_prob(orc_ctx* c, int32_t o, int32_t a, int32_t new_s)
{
    int d = c->cfg.domain;
    if (is_tiger(d)) {
        if (a != 2) return .5;
        return (new_s == o) ? .85 : .15;
    }
    if (is_ftiger(d)) {
        int loc = (new_s < c->S / 2) ? 0 : 1;
        if (a != 2) return .5;
        return (loc == o) ? .85 : .15;
    }
    if (is_coffee(d)) { /* CoffeeProblem::computeObservationProbability :44-60 */
        if (a == 0) return o == 0 ? 1 : 0;
        if (new_s & 16) return o == 0 ? .8 : 1 - .8;
        return o == 1 ? .9 : 1 - .9;
    }
    if (is_sys(d)) { /* SysAdmin::computeObservationProbability :154-165: float results */
        int op = a >= c->sys_N ? a - c->sys_N : a;
        return (o == ((new_s >> op) & 1)) ? (double)SYS_OBSERVE_PROB : (double)(1 - SYS_OBSERVE_PROB);
    }
    if (is_ca(d)) { /* computeObservationProbability :216-229 */
        int H = c->ca_H, n = c->ca_n, k, pos = new_s % c->ca_Hn, obs = o;
        double p = 1;
        int pb[8], po[8];
        (void)a;
        for (k = n - 1; k >= 0; --k) { pb[k] = pos % H; pos /= H; po[k] = obs % H; obs /= H; }
        for (k = 0; k < n; ++k) p *= c->ca_err[abs(pb[k] - po[k])];
        return p;
    }
    if (is_grid(d)) { /* GridWorld::computeObservationProbability :236-250: float * float, goal ignored */
        int N = c->gw_N, G = c->gw_G;
        int x = new_s / (N * G), y = (new_s / G) % N, ox = o / (N * G), oy = (o / G) % N;
        float p = gw_obs_displ_prob(c, x, ox) * gw_obs_displ_prob(c, y, oy);
        (void)a;
        return p;
    }
    return 0;
}

/* BADomainExtension::terminal / reward.
 * ref: TigerBAExtension.cpp:21-44, FactoredTigerBAExtension.cpp (reward uses the PRE-state s) */
static int gw_on_goal(const orc_ctx* c, int32_t s)
{
    int N = c->gw_N, G = c->gw_G, g = s % G;
    return c->gw_goal[g][0] == s / (N * G) && c->gw_goal[g][1] == (s / G) % N;
}
static int ext_terminal(orc_ctx* c, int32_t s, int32_t a, int32_t ns)
{
    if (is_ca(c->cfg.domain)) { /* CollisionAvoidanceBAExtension.cpp:53-63: the NEW state */
        int H = c->ca_H, Hn = c->ca_Hn, x = ns / (H * Hn), y = (ns / Hn) % H, obs = ns % Hn, k;
        int b = 0;
        for (k = c->ca_n - 1; k >= 0; --k) { if (k == x) b = obs % H; obs /= H; }
        return (x < c->ca_n && y == b) || x == 0;
    }
    if (is_grid(c->cfg.domain)) return gw_on_goal(c, s); /* GridWorldBAExtension.cpp:74-83: the PRE-state */
    if (is_sys(c->cfg.domain)) return 0; /* SysAdminBAExtension.cpp:31-37 */
    return is_episodic(c->cfg.domain) && a != 2;
}
static double ext_reward(orc_ctx* c, int32_t s, int32_t a, int32_t ns)
{
    int d = c->cfg.domain;
    if (is_ca(d)) { /* CollisionAvoidanceBAExtension.cpp:65-82 */
        int H = c->ca_H, Hn = c->ca_Hn, x = ns / (H * Hn), y = (ns / Hn) % H, obs = ns % Hn, k, b = 0;
        for (k = c->ca_n - 1; k >= 0; --k) { if (k == x) b = obs % H; obs /= H; }
        if (x < c->ca_n && y == b) return -1000;
        return a == 1 ? 0 : -1;
    }
    if (is_grid(d)) return gw_on_goal(c, s) ? 1 : 0; /* GridWorldBAExtension.cpp:85-99 */
    if (is_sys(d)) /* SysAdminBAExtension.cpp:39-50: operational computers of the NEW state, minus the reboot cost */
        return (double)((float)__builtin_popcount((unsigned)ns) - 1.0f * (float)(a >= c->sys_N));
    if (a == 2) return -1;
    if (is_tiger(d)) return (a == s) ? 10 : -100;
    return (a == ((s < c->S / 2) ? 0 : 1)) ? 10 : -100;
}

/* ------------------------------------------------------------------ regular Dirichlet mode
 * ref: src/utils/random.cpp:40-44 (randomLong), :47-74 (tables), :146-213 (ziggurat normal,
 * Marsaglia-Tsang gamma), :217-242 (sampleFromSampledMult), :281-304 (sampleMult).
 * Bug-compatible: randomLong32 only yields 31-bit values, so the "normal" is
 * half-normal.  log / exp / pow come from libm in REF arithmetic (what the reference calls) and
 * from the deterministic det_log / det_exp below in DEV arithmetic (what the HIP engine evaluates:
 * plain IEEE +,-,*,/ sequences, bit-identical on host and device). */

/* log and exp in the classic argument-reduction + minimax-polynomial form, IEEE double ops only */
double orc_det_log(double x)
{
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
        Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
        Lg7 = 1.479819860511658591e-01;
    uint64_t ix;
    int k;
    double f, s, z, w, t1, t2, R, hfsq, dk;
    if (!(x > 0)) return x == 0 ? -HUGE_VAL : NAN;
    memcpy(&ix, &x, 8);
    k = 0;
    if ((ix >> 52) == 0) { x *= 18014398509481984.0; memcpy(&ix, &x, 8); k = -54; } /* subnormal */
    k += (int)(ix >> 52) - 1023;
    ix = (ix & 0x000fffffffffffffull) | 0x3ff0000000000000ull; /* x in [1, 2) */
    memcpy(&x, &ix, 8);
    if (x > 1.4142135623730951) { x *= 0.5; k += 1; }              /* x in (sqrt(2)/2, sqrt(2)] */
    f    = x - 1.0;
    s    = f / (2.0 + f);
    z    = s * s;
    w    = z * z;
    t1   = w * (Lg2 + w * (Lg4 + w * Lg6));
    t2   = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    R    = t2 + t1;
    hfsq = 0.5 * f * f;
    dk   = (double)k;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

double orc_det_exp(double x)
{
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
        inv_ln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
        P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    double hi, lo, r, c, y, t;
    int k;
    uint64_t bits;
    if (x > 709.0) return HUGE_VAL;
    if (x < -745.0) return 0.0;
    k  = (int)(inv_ln2 * x + (x < 0 ? -0.5 : 0.5));
    t  = (double)k;
    hi = x - t * ln2_hi;
    lo = t * ln2_lo;
    r  = hi - lo;
    t  = r * r;
    c  = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    y  = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    if (k < -1021) { /* result may be subnormal: scale in two steps */
        bits = (uint64_t)(k + 1000 + 1023) << 52;
        memcpy(&t, &bits, 8);
        return y * t * 9.33263618503218878990e-302; /* 2^-1000 */
    }
    bits = (uint64_t)(k + 1023) << 52;
    memcpy(&t, &bits, 8);
    return y * t;
}

static double m_log(const orc_ctx* c, double x) { return c->cfg.arith == ORC_ARITH_DEV ? orc_det_log(x) : log(x); }
static double m_exp(const orc_ctx* c, double x) { return c->cfg.arith == ORC_ARITH_DEV ? orc_det_exp(x) : exp(x); }
static double m_pow(const orc_ctx* c, double x, double y)
{
    if (c->cfg.arith != ORC_ARITH_DEV) return pow(x, y);
    if (x == 0) return y > 0 ? 0.0 : HUGE_VAL;
    return orc_det_exp(y * orc_det_log(x));
}

/* rnd::initiate() tables (random.cpp:47-74); always libm: they are built once on the host */
static void zig_tables(orc_ctx* c)
{
    double tn = 3.442619855899;
    const double m1 = 2147483648.0, vn = 9.91256303526217e-3, q = vn / exp(-.5 * tn * tn);
    int i;
    c->zig_ul[0]   = (unsigned long)((tn / q) * m1);
    c->zig_ul[1]   = 0;
    c->zig_wn[0]   = q / m1;
    c->zig_wn[127] = tn / m1;
    c->zig_fn[0]   = 1.;
    c->zig_fn[127] = exp(-.5 * tn * tn);
    for (i = 126; i > 0; --i) {
        const double dn  = sqrt(-2 * log(vn / tn + exp(-.5 * tn * tn)));
        c->zig_ul[i + 1] = (unsigned long)((dn / tn) * m1);
        c->zig_fn[i]     = exp(-.5 * dn * dn);
        c->zig_wn[i]     = dn / m1;
        tn               = dn;
    }
}

/* randomLong() (random.cpp:40-44): uniform_int_distribution<unsigned long>(0, 2^31-1) = one engine
 * word >> 1 on a 32-bit URBG (Lemire with a power-of-two range never rejects) */
static long random_long(orc_ctx* c)
{
    if (c->rng.mode == ORC_RNG_MT) return (long)(orc_mt_next(&c->rng) >> 1);
    return (long)((uint64_t)(orc_u01(&c->rng) * 2147483648.0)); /* top 31 bits of the draw */
}

/* normalRejectFix (random.cpp:146-176) */
static double normal_reject_fix(orc_ctx* c, long h, unsigned long i)
{
    const double r = 3.442620, r_inverse = 0.2904764;
    double y;
    for (;;) {
        double x = (double)h * c->zig_wn[i];
        if (i == 0) {
            do {
                x = -m_log(c, orc_u01(&c->rng)) * r_inverse;
                y = -m_log(c, orc_u01(&c->rng));
            } while (y + y < x * x);
            return (h > 0) ? r + x : -r - x;
        }
        if (c->zig_fn[i] + orc_u01(&c->rng) * (c->zig_fn[i - 1] - c->zig_fn[i]) < m_exp(c, -.5 * x * x)) return x;
        h = random_long(c);
        i = (unsigned long)(h & 127);
        if ((unsigned long)labs(h) < c->zig_ul[i]) return (double)h * c->zig_wn[i];
    }
}

/* randomNormal (random.cpp:181-187) */
static double random_normal(orc_ctx* c)
{
    const long h = random_long(c), i = h & 127;
    return ((unsigned long)labs(h) < c->zig_ul[i]) ? (double)h * c->zig_wn[i] : normal_reject_fix(c, h, (unsigned long)i);
}

/* rnd::sample::gamma (random.cpp:189-213) */
double orc_gamma(orc_ctx* c, double shape)
{
    double x, v, d, cc;
    if (shape < 1.) {
        double g = orc_gamma(c, shape + 1);
        return g * m_pow(c, orc_u01(&c->rng), 1 / shape);
    }
    d  = shape - 1. / 3.;
    cc = 1. / sqrt(9. * d);
    for (;;) {
        double u, x2;
        do {
            x = random_normal(c);
            v = 1.0 + cc * x;
        } while (v <= 0.0);
        v  = v * v * v;
        u  = orc_u01(&c->rng);
        x2 = x * x;
        if (u < 1.0 - 0.0331 * x2 * x2) return d * v;
        if (m_log(c, u) < .5 * x2 + d * (1. - v + m_log(c, v))) return d * v;
    }
}

/* sampleFromSampledMult (random.cpp:217-242): gamma per count, then sampleFromMult<double> */
int orc_sample_sampled_mult(orc_ctx* c, const float* dir, int n)
{
    double probs[64], sum, p, acc;
    int i;
    probs[0] = orc_gamma(c, dir[0]);
    sum      = probs[0];
    for (i = 1; i < n; ++i) {
        probs[i] = orc_gamma(c, dir[i]);
        sum += probs[i];
    }
    p   = orc_u01(&c->rng) * sum;
    acc = probs[0];
    for (i = 1; i < n; ++i) {
        if (p < acc) return i - 1;
        acc += probs[i];
    }
    return n - 1;
}

/* sampleMult (random.cpp:281-304): gammas stored as float, float sum, float division */
void orc_sample_mult(orc_ctx* c, const float* dir, int n, float* out)
{
    float sum = 0;
    int i;
    for (i = 0; i < n; ++i) {
        out[i] = (float)orc_gamma(c, dir[i]);
        sum += out[i];
    }
    for (i = 0; i < n; ++i) out[i] = out[i] / sum;
}

static int sample_dirichlet_row(orc_ctx* c, const float* row, int n)
{
    return c->cfg.dirichlet_regular ? orc_sample_sampled_mult(c, row, n) : orc_sample_expected_mult(&c->rng, row, n);
}

/* ------------------------------------------------------------------ priors (tabular) */

/* ref: TigerBAPrior src/domains/tiger/TigerPriors.cpp:14-43;
 *      FactoredTigerFlatPrior src/domains/tiger/FactoredTigerPriors.cpp:18-88;
 *      layout BAFlatModel: phi[s*A*S + a*S + s'], psi[a*S*O + s'*O + o] (utils/index.cpp:13-16) */

/* SysAdminFlatPrior (SysAdminFlatPrior.cpp:24-247): the true transition probabilities times 10000,
 * enumerated by recursion over the computers from N-1 down to 0.  Restated with the reference's
 * evaluation order, because later leaves overwrite earlier ones, and with its arithmetic types
 * (double accumulated probability, float parameters, float counts).  Quirk kept: setTrueTCounts
 * (:168-172) asks numFailingNeighbours about "computer" a = the ACTION index. */
static void sys_flat_leaf_reboot(orc_ctx* c, int32_t s, int32_t ns, double prob, int reb) /* :236-245 */
{
    c->prior[s * c->A * c->S + (c->sys_N + reb) * c->S + ns] = (float)prob * 10000.0f;
}
static void sys_flat_recur_reboot(orc_ctx* c, int32_t s, int32_t ns, int comp, double acc, int reb) /* :188-234 */
{
    int32_t ns_fail;
    if (comp == -1) { sys_flat_leaf_reboot(c, s, ns, acc, reb); return; }
    ns_fail = ns & ~(1 << comp);
    if (!((s >> comp) & 1)) {
        sys_flat_recur_reboot(c, s, ns_fail, comp - 1, acc, reb);
    } else {
        double fail_prob = 1 - c->sys_keep[sys_failing_neighbours(c, comp, s)];
        sys_flat_recur_reboot(c, s, ns, comp - 1, acc * (1 - fail_prob), reb);
        sys_flat_recur_reboot(c, s, ns_fail, comp - 1, acc * fail_prob, reb);
    }
}
static void sys_flat_leaf(orc_ctx* c, int32_t s, int32_t ns, double prob) /* setTrueTCounts :150-186 */
{
    int N = c->sys_N, a;
    float* phi = c->prior + (size_t)s * c->A * c->S;
    for (a = 0; a < N; ++a) phi[a * c->S + ns] = (float)prob * 10000.0f;
    for (a = N; a < 2 * N; ++a) {
        if ((ns >> (a - N)) & 1) {
            double fail_prob = 1 - c->sys_keep[sys_failing_neighbours(c, a, s)];
            phi[a * c->S + ns] = 10000.0f * (float)(prob + (prob * fail_prob / (1 - fail_prob) * SYS_REBOOT_RATE));
        } else {
            phi[a * c->S + ns] = 10000.0f * (float)(prob * (1 - SYS_REBOOT_RATE));
        }
    }
}
static void sys_flat_recur(orc_ctx* c, int32_t s, int32_t ns, int comp, double acc) /* :93-148 */
{
    int32_t ns_fail;
    if (comp == -1) { sys_flat_leaf(c, s, ns, acc); return; }
    ns_fail = ns & ~(1 << comp);
    if (!((s >> comp) & 1)) {
        sys_flat_recur(c, s, ns_fail, comp - 1, acc);
        sys_flat_recur_reboot(c, s, ns_fail, comp - 1, acc * (1 - SYS_REBOOT_RATE), comp);
        sys_flat_recur_reboot(c, s, ns, comp - 1, acc * SYS_REBOOT_RATE, comp);
    } else {
        double fail_prob = 1 - c->sys_keep[sys_failing_neighbours(c, comp, s)];
        sys_flat_recur(c, s, ns, comp - 1, acc * (1 - fail_prob));
        sys_flat_recur(c, s, ns_fail, comp - 1, acc * fail_prob);
    }
}
static int build_sysadmin_flat_prior(orc_ctx* c) /* precomputeFlatPrior :38-91 */
{
    int S = c->S, O = c->O, N = c->sys_N, s, ns, k;
    float* psi = c->prior + c->phi_len;
    float high = 10000.0f * SYS_OBSERVE_PROB, low = 10000.0f * (1 - SYS_OBSERVE_PROB);
    memset(c->prior, 0, sizeof(float) * (size_t)c->ncnt); /* BAFlatModel(&domain_size): zero counts */
    for (s = 0; s < S; ++s) sys_flat_recur(c, s, S - 1, N - 1, 1);
    for (ns = 0; ns < S; ++ns)
        for (k = 0; k < N; ++k) {
            int hi = (ns >> k) & 1;
            psi[k * S * O + ns * O + hi]           = high;
            psi[k * S * O + ns * O + (1 - hi)]     = low;
            psi[(k + N) * S * O + ns * O + hi]     = high;
            psi[(k + N) * S * O + ns * O + 1 - hi] = low;
        }
    return 0;
}

/* GridWorldFlatBAPrior (GridWorldBAPriors.cpp:21-156): tabular counts of the true gridworld.
 * Transition counts are ADDED (a blocked move and a failed move land in the same cell); observation
 * counts = P(o | s') * 100000 for the observations that show s' own goal, 0 elsewhere. */
static int build_gridworld_flat_prior(orc_ctx* c)
{
    int S = c->S, A = c->A, N = c->gw_N, G = c->gw_G, a, s, g2, x, y;
    float noise = c->cfg.noise, total = c->cfg.counts_total;
    float* phi = c->prior;
    float* psi = c->prior + c->phi_len;
    if (noise < 0 || noise > (1 - .15)) {
        snprintf(c->err, sizeof c->err, "Gridworld expects noise in between 0 and %f (received %f)", 1 - .15, noise);
        return -1;
    }
    memset(c->prior, 0, sizeof(float) * (size_t)c->ncnt);
    for (a = 0; a < A; ++a)
        for (s = 0; s < S; ++s) {
            int ax = s / (N * G), ay = (s / G) % N, gl = s % G, nx = ax, ny = ay;
            int on_goal = c->gw_goal[gl][0] == ax && c->gw_goal[gl][1] == ay;
            float success_prob = gw_slow_at(c, ax, ay) ? (float)(.15 + noise) : (float).95;
            float goal_prob    = (float)1 / (float)G;
            float* row         = phi + (size_t)s * A * S + (size_t)a * S;
            /* setPriorTransitionProbabilities :62-126 */
            if (on_goal) {
                float prob = (1 - success_prob) * goal_prob;
                for (g2 = 0; g2 < G; ++g2) row[ax * N * G + ay * G + g2] += prob * total;
            } else {
                float prob = 1 - success_prob;
                row[s] += prob * total;
            }
            gw_move(c, a, &nx, &ny);
            if (on_goal) {
                float prob = success_prob * goal_prob;
                for (g2 = 0; g2 < G; ++g2) row[nx * N * G + ny * G + g2] += prob * total;
            } else {
                row[nx * N * G + ny * G + gl] += success_prob * total;
            }
            /* setPriorObservationProbabilities :128-150 (s plays the role of new_s) */
            for (x = 0; x < N; ++x)
                for (y = 0; y < N; ++y) {
                    int o       = x * N * G + y * G + gl;
                    double prob = domain_obs_prob(c, o, a, s);
                    psi[(size_t)a * S * c->O + (size_t)s * c->O + o] = (float)(prob * 100000.0f);
                }
        }
    return 0;
}

/* CollisionAvoidanceTablePrior (CollisionAvoidancePriors.cpp:65-208): tabular counts.  For every
 * state s = (x, y, obstacles) and every obstacle configuration b': the transition count to
 * (x-1, y', b') is prod_i obstacleTransProb(b_i, b'_i) * -C (nothing from x = 0), and the
 * observation count of b' "in" s is prod_i observationDistr(H, b_i)[b'_i] * 10000. */
static double ca_obstacle_trans_prob(const orc_ctx* c, int y, int ny) /* :136-170 */
{
    int dist = abs(y - ny);
    if (dist > 1) return 0;
    if (y == 0 || y == c->ca_H - 1) return dist == 0 ? .75 + .5 * c->cfg.noise : .25 - .5 * c->cfg.noise;
    return dist == 0 ? .5 + c->cfg.noise : .25 - .5 * c->cfg.noise;
}
static float ca_observation_distr(const orc_ctx* c, int pos, int oy) /* observationDistr(height, obstacle_pos) :46-63 */
{
    int H = c->ca_H;
    if (oy == 0) return (float)normal_cdf(-pos + .5);
    if (oy == H - 1) return (float)normal_cdf(-(H - 1 - pos) + .5);
    { int dist = abs(oy - pos); return (float)(normal_cdf(dist + .5) - normal_cdf(dist - .5)); }
}
static int build_ca_flat_prior(orc_ctx* c)
{
    int S = c->S, A = c->A, O = c->O, W = c->ca_W, H = c->ca_H, n = c->ca_n, Hn = c->ca_Hn;
    int ob, nob, x, y, a, i;
    float* phi = c->prior;
    float* psi = c->prior + c->phi_len;
    if (!(c->cfg.noise < .5 && c->cfg.noise > -.5)) { /* assert(c.noise < .5 && c.noise > -.5) :80 */
        snprintf(c->err, sizeof c->err, "CollisionAvoidanceTablePrior needs -.5 < noise < .5 (is: %f)", c->cfg.noise);
        return -1;
    }
    memset(c->prior, 0, sizeof(float) * (size_t)c->ncnt);
    for (ob = 0; ob < Hn; ++ob)
        for (nob = 0; nob < Hn; ++nob) {
            int b[8], nb[8], r1 = ob, r2 = nob;
            double tprob = 1, oprob = 1;
            for (i = n - 1; i >= 0; --i) { b[i] = r1 % H; r1 /= H; nb[i] = r2 % H; r2 /= H; }
            for (i = 0; i < n; ++i) { /* stops at the first zero factor (:191-199) */
                tprob *= ca_obstacle_trans_prob(c, b[i], nb[i]);
                if (tprob == 0) break;
            }
            for (i = 0; i < n; ++i) oprob *= ca_observation_distr(c, b[i], nb[i]);
            for (x = 0; x < W; ++x)
                for (y = 0; y < H; ++y) {
                    int s = (x * H + y) * Hn + ob;
                    for (a = 0; a < A; ++a) {
                        psi[(size_t)a * S * O + (size_t)s * O + nob] = (float)(oprob * 10000);
                        if (x > 0 && tprob != 0) {
                            int ny = y + a - 1;
                            if (ny == -1 || ny == H) ny = y;
                            phi[(size_t)s * A * S + (size_t)a * S + ((x - 1) * H + ny) * Hn + nob] = (float)(tprob * c->cfg.counts_total);
                        }
                    }
                }
        }
    return 0;
}

static int build_tabular_prior(orc_ctx* c)
{
    int S = c->S, A = c->A, O = c->O, i, s, ns;
    float noise = c->cfg.noise, total = c->cfg.counts_total;
    float acc   = (.85f - noise) * total;
    float inacc = (.15f + noise) * total;
    float* phi;
    float* psi;
    c->phi_len = S * A * S;
    c->ncnt    = S * A * S + A * S * O;
    c->prior   = (float*)malloc(sizeof(float) * (size_t)c->ncnt);
    phi        = c->prior;
    psi        = c->prior + c->phi_len;
    if (is_sys(c->cfg.domain)) return build_sysadmin_flat_prior(c); /* ignores --noise / -C */
    if (is_grid(c->cfg.domain)) return build_gridworld_flat_prior(c);
    if (is_ca(c->cfg.domain)) return build_ca_flat_prior(c);
    if (noise <= -.15 || noise > .3) {
        snprintf(c->err, sizeof c->err, "noise has to be between -.15 and .3");
        return -1;
    }
    for (i = 0; i < c->ncnt; ++i) c->prior[i] = 5000;
    if (is_tiger(c->cfg.domain)) {
        phi[1 * A * S + 2 * S + 0] = 0; /* count(right, listen, left)  = 0 */
        phi[0 * A * S + 2 * S + 1] = 0; /* count(left,  listen, right) = 0 */
        psi[2 * S * O + 1 * O + 1] = acc;   /* listen, right, hear right */
        psi[2 * S * O + 1 * O + 0] = inacc; /* listen, right, hear left  */
        psi[2 * S * O + 0 * O + 1] = inacc; /* listen, left,  hear right */
        psi[2 * S * O + 0 * O + 0] = acc;   /* listen, left,  hear left  */
        return 0;
    }
    if (is_ftiger(c->cfg.domain)) {
        for (s = 0; s < S; ++s)
            for (ns = 0; ns < S; ++ns)
                if (s != ns) phi[s * A * S + 2 * S + ns] = 0;
        for (s = 0; s < S; ++s) {
            int left                       = s < S / 2;
            psi[2 * S * O + s * O + (left ? 0 : 1)] = acc;
            psi[2 * S * O + s * O + (left ? 1 : 0)] = inacc;
        }
        return 0;
    }
    snprintf(c->err, sizeof c->err, "domain %d has no tabular prior in the oracle", c->cfg.domain);
    return -1;
}

/* ------------------------------------------------------------------ simulator.step */

/* ref: BAPOMDP::step src/bayes-adaptive/models/table/BAPOMDP.cpp:111-143 with
 *      BAFlatModel::sampleStateIndex/sampleObservationIndex/incrementCountsOf
 *      src/bayes-adaptive/states/table/BAFlatModel.cpp:83-139 (expected-Dirichlet method) */
static int ba_table_step(orc_ctx* c, simstate* st, int32_t a, int32_t* o, double* r, int update)
{
    int S = c->S, A = c->A, O = c->O;
    float* phi = st->cnt;
    float* psi = st->cnt + c->phi_len;
    int32_t s  = st->s;
    int32_t ns = sample_dirichlet_row(c, &phi[s * A * S + a * S], S);
    int t;
    *o = sample_dirichlet_row(c, &psi[a * S * O + ns * O], O);
    t  = ext_terminal(c, s, a, ns);
    *r = ext_reward(c, s, a, ns);
    if (update) {
        phi[s * A * S + a * S + ns] += 1;
        psi[a * S * O + ns * O + *o] += 1;
    }
    st->s = ns;
    return t;
}

/* ------------------------------------------------------------------ factored BA model */

static int is_mh(const orc_ctx* c) { return c->cfg.belief == ORC_BELIEF_MH_GIBBS || c->cfg.belief == ORC_BELIEF_MH_NIPS; }
static int is_breeding(const orc_ctx* c) { return c->cfg.belief == ORC_BELIEF_REINVIGORATION || c->cfg.belief == ORC_BELIEF_INCUBATOR; }
static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* indexing::projectUsingStepSize (src/utils/index.cpp:98-118) */
static void features_of(int v, const int32_t* step, int n, int* out)
{
    int i;
    if (n == 1) { out[0] = v; return; }
    for (i = 0; i < n; ++i) { out[i] = v / step[i]; v = v % step[i]; }
}
/* indexing::project (src/utils/index.cpp:51-83): last dimension fastest */
static int project(const int* val, const int32_t* size, int n)
{
    int i, r = 0;
    for (i = 0; i < n; ++i) r = r * size[i] + val[i];
    return r;
}
static uint32_t node_mask(const orc_ctx* c, const fnode* nd, const float* cnt)
{
    return nd->var >= 0 ? f2u(cnt[c->fd.ncounts + nd->var]) : nd->fixed_mask;
}
/* DBNNode::cptIndex(graph input, 0) (DBNNode.cpp:171-205): mixed radix over the node's parents */
static int node_row(const orc_ctx* c, const fnode* nd, uint32_t mask, const int* fv)
{
    int j, idx = 0;
    for (j = 0; j < nd->nmax; ++j)
        if ((mask >> j) & 1u) idx = idx * c->fd.Ssz[nd->maxp[j]] + fv[nd->maxp[j]];
    return nd->off + idx * nd->out;
}

/* lgamma as one fixed sequence of IEEE operations (the HIP engine carries the same sequence, fba_device.h det_lgamma):
 * x >= 1 is shifted up to >= 16 with the recurrence lgamma(x) = lgamma(x + n) - log(x (x + 1) ... (x + n - 1)), then
 * Stirling's series to 1 / x^13 (truncation error below 2e-18 at 16).  Absolute error against libm below 1e-13 for
 * the counts a run can reach; relative error is large only next to the zeros at 1 and 2, where the value is tiny. */
double orc_det_lgamma(double x)
{
    double prod = 1.0, shift = 0.0, xi, x2, series;
    while (x < 16.0) {
        prod *= x;
        x += 1.0;
        if (prod > 1e250) { shift += orc_det_log(prod); prod = 1.0; }
    }
    shift += orc_det_log(prod);
    xi = 1.0 / x;
    x2 = xi * xi;
    series = xi * (1.0 / 12.0 + x2 * (-1.0 / 360.0 + x2 * (1.0 / 1260.0 + x2 * (-1.0 / 1680.0 + x2 * (1.0 / 1188.0 +
             x2 * (-691.0 / 360360.0 + x2 * (1.0 / 156.0)))))));
    return ((x - 0.5) * orc_det_log(x) - x + 0.91893853320467274178) + series - shift;
}
/* rnd::math::logGamma (random.cpp:127-135): 0 below 1, lgamma otherwise */
static double log_gamma(const orc_ctx* c, double x)
{
    if (x < 1) return 0;
    return c->cfg.arith == ORC_ARITH_DEV ? orc_det_lgamma(x) : lgamma(x);
}
/* BABNModel::LogBDScore (BABNModel.cpp:451-478) over DBNNode::LogBDScore (DBNNode.cpp:82-117): per action the
 * transition nodes, then the observation nodes; per node every Dirichlet row in CPT order: the cells' logGamma
 * differences as they come, then logGamma(prior total) - logGamma(total), one running double sum.
 * `cnt` and `prior` are particle blobs of the same structure (pinned by golden["log_bd_score"]). */
double orc_log_bd_score(orc_ctx* c, const float* cnt, const float* prior)
{
    const fdesc* d = &c->fd;
    double bd = 0;
    int a, k;
    for (a = 0; a < c->A; ++a)
        for (k = 0; k < d->FS + d->FO; ++k) {
            const fnode* nd = k < d->FS ? &d->T[a * d->FS + k] : &d->O[a * d->FO + (k - d->FS)];
            uint32_t mask   = node_mask(c, nd, cnt);
            int rows = 1, j, r, v;
            for (j = 0; j < nd->nmax; ++j)
                if ((mask >> j) & 1u) rows *= d->Ssz[nd->maxp[j]];
            for (r = 0; r < rows; ++r) {
                double tot = 0, ptot = 0;
                for (v = 0; v < nd->out; ++v) {
                    float x = cnt[nd->off + r * nd->out + v], y = prior[nd->off + r * nd->out + v];
                    tot += x;
                    ptot += y;
                    bd += log_gamma(c, x) - log_gamma(c, y);
                }
                bd += log_gamma(c, ptot) - log_gamma(c, tot);
            }
        }
    return bd;
}

/* BAPOMDP::step with BABNModel::sampleStateIndex / sampleObservationIndex / incrementCountsOf
 * ref: BABNModel.cpp:292-325, 354-382.  Quirk kept: the observation CPTs are
 * incremented at the row of the PREVIOUS state's parent values (parent_values = features(s)). */
static int ba_fact_step(orc_ctx* c, simstate* st, int32_t a, int32_t* o, double* r, int update)
{
    const fdesc* d = &c->fd;
    int fv[ORC_MAXF], nf[ORC_MAXF], of[ORC_MAXF], f, t;
    int32_t s = st->s, ns;
    features_of(s, d->Sstep, d->FS, fv);
    for (f = 0; f < d->FS; ++f) {
        const fnode* nd = &d->T[a * d->FS + f];
        nf[f] = sample_dirichlet_row(c, st->cnt + node_row(c, nd, node_mask(c, nd, st->cnt), fv), nd->out);
    }
    ns = project(nf, d->Ssz, d->FS);
    for (f = 0; f < d->FO; ++f) {
        const fnode* nd = &d->O[a * d->FO + f];
        of[f] = sample_dirichlet_row(c, st->cnt + node_row(c, nd, node_mask(c, nd, st->cnt), nf), nd->out);
    }
    *o = project(of, d->Osz, d->FO);
    t  = ext_terminal(c, s, a, ns);
    *r = ext_reward(c, s, a, ns);
    if (update) {
        for (f = 0; f < d->FS; ++f) {
            const fnode* nd = &d->T[a * d->FS + f];
            st->cnt[node_row(c, nd, node_mask(c, nd, st->cnt), fv) + nf[f]] += 1;
        }
        for (f = 0; f < d->FO; ++f) {
            const fnode* nd = &d->O[a * d->FO + f];
            st->cnt[node_row(c, nd, node_mask(c, nd, st->cnt), fv) + of[f]] += 1;
        }
    }
    st->s = ns;
    return t;
}

/* BABNModel::computeObservationProbability (BABNModel.cpp:328-352): product over observation
 * features of expectedMult(row(new_s))[o_f], accumulated in double */
static double ba_fact_obs_prob(orc_ctx* c, const simstate* st, int32_t a, int32_t o)
{
    const fdesc* d = &c->fd;
    int fv[ORC_MAXF], of[ORC_MAXF], f, i;
    double prob = 1;
    features_of(st->s, d->Sstep, d->FS, fv);
    features_of(o, d->Ostep, d->FO, of);
    for (f = 0; f < d->FO; ++f) {
        const fnode* nd  = &d->O[a * d->FO + f];
        const float* row = st->cnt + node_row(c, nd, node_mask(c, nd, st->cnt), fv);
        float sum        = row[0];
        if (c->cfg.dirichlet_regular) { /* sampleMultinominal = sampleMult: a fresh Dirichlet draw per feature */
            float tmp[64];
            orc_sample_mult(c, row, nd->out, tmp);
            prob *= tmp[of[f]];
            continue;
        }
        for (i = 1; i < nd->out; ++i) sum += row[i];
        prob *= (sum <= 1e-300) ? 0.0f : row[of[f]] / sum;
    }
    return prob;
}

static void fdesc_steps(const int32_t* size, int n, int32_t* step)
{
    int i;
    step[n - 1] = 1;
    for (i = n - 2; i >= 0; --i) step[i] = step[i + 1] * size[i + 1];
}

/* FactoredTigerFactoredPrior::setObservationModel (FactoredTigerPriors.cpp:221-263) into the
 * listen observation node's region, for the parent set `mask` (bit j = state feature j) */
static void ftiger_set_observation_model(orc_ctx* c, float* cnt, uint32_t mask)
{
    const fnode* nd = &c->fd.O[2 * c->fd.FO + 0];
    float acc = (.85f - c->cfg.noise) * c->cfg.counts_total;
    float inacc = (.15f + c->cfg.noise) * c->cfg.counts_total;
    float unif = .5f * c->cfg.counts_total;
    int np = __builtin_popcount(mask), rows = 1 << np, r;
    float* row = cnt + nd->off;
    memset(row, 0, sizeof(float) * (size_t)(2 << nd->nmax));
    for (r = 0; r < rows; ++r) {
        if (np > 0 && (mask & 1u)) { /* parents[0] == 0: informed by the tiger location */
            int loc    = r >> (np - 1); /* first parent is the most significant digit */
            row[2 * r + 0] = (loc == 0) ? acc : inacc;
            row[2 * r + 1] = (loc == 1) ? acc : inacc;
        } else {
            row[2 * r + 0] = unif;
            row[2 * r + 1] = unif;
        }
    }
    cnt[c->fd.ncounts + 0] = u2f(mask);
}

/* FactoredTigerFactoredPrior ctor (FactoredTigerPriors.cpp:95-195): everything but the listen
 * observation node, which is per particle */
static int build_ftiger_factored_prior(orc_ctx* c)
{
    fdesc* d = &c->fd;
    int A = c->A, a, f, off = 0;
    if (c->cfg.noise <= -.15 || c->cfg.noise > .3) {
        snprintf(c->err, sizeof c->err, "noise must be between -.15 and .3");
        return -1;
    }
    d->FS = c->tiger_K + 1;
    d->FO = 1;
    if (d->FS > ORC_MAXF) { snprintf(c->err, sizeof c->err, "too many state features"); return -1; }
    for (f = 0; f < d->FS; ++f) d->Ssz[f] = 2;
    d->Osz[0] = 2;
    fdesc_steps(d->Ssz, d->FS, d->Sstep);
    fdesc_steps(d->Osz, d->FO, d->Ostep);
    d->T = (fnode*)calloc((size_t)A * d->FS, sizeof(fnode));
    d->O = (fnode*)calloc((size_t)A * d->FO, sizeof(fnode));
    for (a = 0; a < A; ++a)
        for (f = 0; f < d->FS; ++f) {
            fnode* nd = &d->T[a * d->FS + f];
            nd->off = off; nd->out = 2; nd->var = -1;
            if (a == 2) { nd->nmax = 1; nd->maxp[0] = f; nd->fixed_mask = 1; off += 4; }
            else { nd->nmax = 0; nd->fixed_mask = 0; off += 2; }
        }
    for (a = 0; a < A; ++a) {
        fnode* nd = &d->O[a * d->FO];
        nd->off = off; nd->out = 2; nd->var = -1;
        if (a == 2) {
            nd->nmax = d->FS; nd->var = 0;
            for (f = 0; f < d->FS; ++f) nd->maxp[f] = f;
            off += 2 << d->FS;
        } else { nd->nmax = 0; off += 2; }
    }
    d->ncounts = off;
    d->nvar    = 1;
    c->ncnt    = off + d->nvar;
    c->prior   = (float*)calloc((size_t)c->ncnt, sizeof(float));
    for (a = 0; a < A; ++a)
        for (f = 0; f < d->FS; ++f) {
            const fnode* nd = &d->T[a * d->FS + f];
            if (a == 2) { c->prior[nd->off + 0] = 5000; c->prior[nd->off + 3] = 5000; } /* listen keeps feature values */
            else { c->prior[nd->off] = 5000; c->prior[nd->off + 1] = 5000; }
        }
    for (a = 0; a < 2; ++a) { c->prior[d->O[a].off] = 5000; c->prior[d->O[a].off + 1] = 5000; }
    /* default listen model = correct structure {0}; overwritten per particle when a structure prior is set */
    ftiger_set_observation_model(c, c->prior, 1u);
    return 0;
}

/* GridWorldFactBAPrior::setNoisyTransitionNode (GridWorldBAPriors.cpp:255-295) when `with_goal`,
 * else the x / y part of preComputePrior (:316-413): the agent-x (feature 0) or agent-y (feature 1)
 * transition node of action a, written into its region of `cnt` */
static void gw_fill_xy_node(orc_ctx* c, float* cnt, int a, int feature, int with_goal)
{
    const fnode* nd = &c->fd.T[a * c->fd.FS + feature];
    int N = c->gw_N, G = c->gw_G, x, y, g;
    float total = c->cfg.counts_total;
    memset(cnt + nd->off, 0, sizeof(float) * (size_t)N * N * G * N);
    for (x = 0; x < N; ++x)
        for (y = 0; y < N; ++y) {
            int nx = x, ny = y, loc, new_loc;
            float trans_prob = gw_slow_at(c, x, y) ? (float)(.15 + c->cfg.noise) : (float).95;
            gw_move(c, a, &nx, &ny);
            loc     = feature == 0 ? x : y;
            new_loc = feature == 0 ? nx : ny;
            if (with_goal) {
                for (g = 0; g < G; ++g) {
                    float* row = cnt + nd->off + ((x * N + y) * G + g) * N;
                    row[loc] += (1 - trans_prob) * total;
                    row[new_loc] += (trans_prob)*total;
                }
            } else {
                float* row = cnt + nd->off + (x * N + y) * N;
                row[loc] += (1 - trans_prob) * total;
                row[new_loc] += (trans_prob)*total;
            }
        }
}

/* GridWorldFactBAPrior ctor + preComputePrior (GridWorldBAPriors.cpp:158-198, 316-413): the
 * correct-structure prior.  Features {x, y, goal}; T parents x:{x,y} y:{x,y} goal:{x,y,goal};
 * O parents x_obs:{x} y_obs:{y} goal_obs:{goal}.  Under match-uniform each particle may add the
 * goal as a parent of the x / y nodes per action (8 per-particle mask words). */
static int build_gridworld_factored_prior(orc_ctx* c)
{
    fdesc* d = &c->fd;
    int A = c->A, N = c->gw_N, G = c->gw_G, a, f, off = 0, v, x, y, g, ng, nvar = 0;
    float static_total = 100000;
    if (c->cfg.noise < 0 || c->cfg.noise > (1 - .15)) {
        snprintf(c->err, sizeof c->err, "Gridworld expects noise in between 0 and %f (received %f)", 1 - .15, c->cfg.noise);
        return -1;
    }
    if (c->cfg.structure_prior != ORC_SP_NONE && c->cfg.structure_prior != ORC_SP_MATCH_UNIFORM) {
        snprintf(c->err, sizeof c->err, "Please enter a valid structure noise option for the GridWorld problem ('match-uniform' or 'match-counts')");
        return -1;
    }
    d->FS = 3; d->FO = 3;
    d->Ssz[0] = d->Ssz[1] = d->Osz[0] = d->Osz[1] = N;
    d->Ssz[2] = d->Osz[2] = G;
    fdesc_steps(d->Ssz, 3, d->Sstep);
    fdesc_steps(d->Osz, 3, d->Ostep);
    d->T = (fnode*)calloc((size_t)A * 3, sizeof(fnode));
    d->O = (fnode*)calloc((size_t)A * 3, sizeof(fnode));
    for (a = 0; a < A; ++a)
        for (f = 0; f < 3; ++f) {
            fnode* nd = &d->T[a * 3 + f];
            nd->off = off; nd->nmax = 3; nd->maxp[0] = 0; nd->maxp[1] = 1; nd->maxp[2] = 2;
            if (f < 2) { nd->out = N; nd->var = nvar++; nd->fixed_mask = 3; off += N * N * G * N; }
            else { nd->out = G; nd->var = -1; nd->fixed_mask = 7; off += N * N * G * G; }
        }
    for (a = 0; a < A; ++a)
        for (f = 0; f < 3; ++f) {
            fnode* nd = &d->O[a * 3 + f];
            nd->off = off; nd->nmax = 1; nd->maxp[0] = f; nd->var = -1; nd->fixed_mask = 1;
            nd->out = d->Osz[f];
            off += d->Ssz[f] * d->Osz[f];
        }
    d->ncounts = off;
    d->nvar    = nvar;
    c->ncnt    = off + nvar;
    c->prior   = (float*)calloc((size_t)c->ncnt, sizeof(float));
    for (a = 0; a < A; ++a) {
        /* O (known): x / y observation = displacement model, goal observed exactly */
        for (f = 0; f < 2; ++f)
            for (v = 0; v < N; ++v)
                for (x = 0; x < N; ++x) c->prior[d->O[a * 3 + f].off + v * N + x] = gw_obs_displ_prob(c, v, x) * static_total;
        for (v = 0; v < G; ++v) c->prior[d->O[a * 3 + 2].off + v * G + v] = static_total;
        /* T */
        gw_fill_xy_node(c, c->prior, a, 0, 0);
        gw_fill_xy_node(c, c->prior, a, 1, 0);
        for (x = 0; x < N; ++x)
            for (y = 0; y < N; ++y)
                for (g = 0; g < G; ++g) {
                    float* row = c->prior + d->T[a * 3 + 2].off + ((x * N + y) * G + g) * G;
                    if (c->gw_goal[g][0] != x || c->gw_goal[g][1] != y) row[g] = static_total;
                    else for (ng = 0; ng < G; ++ng) row[ng] = static_total;
                }
        c->prior[d->ncounts + d->T[a * 3 + 0].var] = u2f(3u);
        c->prior[d->ncounts + d->T[a * 3 + 1].var] = u2f(3u);
    }
    return 0;
}

/* CollisionAvoidanceFactoredPrior ctor (CollisionAvoidancePriors.cpp:210-347), structure priors ""
 * / "match-counts" (correct graph) and "fully-connected".  Features {x, y, obstacle_1..n};
 * T parents x:{x}, y:{y}, obstacle f:{f} (or all features when fully connected); O parents
 * observed obstacle f:{obstacle f}.  The per-particle edge noise ("uniform", "match-uniform")
 * needs variable-size CPTs and is not built. */
static void ca_obstacle_transition(const orc_ctx* c, int y, float* out) /* obstacleTransition :385-404 */
{
    int H = c->ca_H, k;
    float move_prob = (float)(.25 - .5 * c->cfg.noise);
    float stay_prob = (y == 0 || y == H - 1) ? (float)(3 * .25 + .5 * c->cfg.noise) : (float)(2 * .25 + c->cfg.noise);
    for (k = 0; k < H; ++k) out[k] = 0;
    if (y != 0) out[y - 1] = move_prob * c->cfg.counts_total;
    if (y != H - 1) out[y + 1] = move_prob * c->cfg.counts_total;
    out[y] = stay_prob * c->cfg.counts_total;
}

/* CollisionAvoidanceFactoredPrior::sampleBlockTModel (CollisionAvoidancePriors.cpp:528-590) for one
 * (action, obstacle feature): the node's CPT under parent set `mask` (bit k = state feature k).  With
 * the obstacle itself among the parents every row is obstacleTransition(its own value), otherwise
 * every row is uniform, counts_total / H. */
static void ca_fill_obstacle_node(orc_ctx* c, float* cnt, int a, int f, uint32_t mask)
{
    const fdesc* d  = &c->fd;
    const fnode* nd = &d->T[a * d->FS + f];
    int H = c->ca_H, rows = 1, rows_max = 1, k, r, y;
    for (k = 0; k < d->FS; ++k) {
        rows_max *= d->Ssz[k];
        if ((mask >> k) & 1u) rows *= d->Ssz[k];
    }
    memset(cnt + nd->off, 0, sizeof(float) * (size_t)rows_max * H);
    for (r = 0; r < rows; ++r) {
        float* row = cnt + nd->off + r * H;
        if ((mask >> f) & 1u) {
            int rem = r, own = 0;
            for (k = d->FS - 1; k >= 0; --k) /* last parent is the fastest digit */
                if ((mask >> k) & 1u) {
                    if (k == f) own = rem % d->Ssz[k];
                    rem /= d->Ssz[k];
                }
            ca_obstacle_transition(c, own, row);
        } else {
            for (y = 0; y < H; ++y) row[y] = c->cfg.counts_total / (float)H;
        }
    }
    if (nd->var >= 0) cnt[d->ncounts + nd->var] = u2f(mask);
}
static int build_ca_factored_prior(orc_ctx* c)
{
    fdesc* d = &c->fd;
    int A = c->A, W = c->ca_W, H = c->ca_H, n = c->ca_n, FS = 2 + n, a, f, off = 0, y, k;
    int full = c->cfg.structure_prior == ORC_SP_FULLY_CONNECTED;
    /* edge noise "uniform" / "match-uniform" (:349-383): every obstacle node's parents are drawn per particle */
    int noisy = c->cfg.structure_prior == ORC_SP_UNIFORM || c->cfg.structure_prior == ORC_SP_MATCH_UNIFORM ||
                ((is_breeding(c) || is_mh(c)) && !full); /* bred / re-drawn particles carry their own structures */
    if (c->cfg.noise > .5 || c->cfg.noise < -.5) {
        snprintf(c->err, sizeof c->err, "CollisionAvoidanceFactoredPrior must be intiiated with -.5 < noise < .5 (is: %f)", c->cfg.noise);
        return -1;
    }
    if (FS > ORC_MAXF) { snprintf(c->err, sizeof c->err, "too many state features"); return -1; }
    d->FS = FS; d->FO = n;
    d->Ssz[0] = W; d->Ssz[1] = H;
    for (f = 0; f < n; ++f) { d->Ssz[2 + f] = H; d->Osz[f] = H; }
    fdesc_steps(d->Ssz, FS, d->Sstep);
    fdesc_steps(d->Osz, n, d->Ostep);
    d->T = (fnode*)calloc((size_t)A * FS, sizeof(fnode));
    d->O = (fnode*)calloc((size_t)A * n, sizeof(fnode));
    for (a = 0; a < A; ++a)
        for (f = 0; f < FS; ++f) {
            fnode* nd = &d->T[a * FS + f];
            nd->off = off; nd->out = d->Ssz[f]; nd->var = -1;
            if (f >= 2 && (full || noisy)) {
                int rows = 1;
                nd->nmax = FS;
                for (k = 0; k < FS; ++k) { nd->maxp[k] = k; rows *= d->Ssz[k]; }
                nd->fixed_mask = (1u << FS) - 1u;
                if (noisy) nd->var = a * n + (f - 2);
                off += rows * H;
            } else {
                nd->nmax = 1; nd->maxp[0] = f; nd->fixed_mask = 1;
                off += d->Ssz[f] * d->Ssz[f];
            }
        }
    for (a = 0; a < A; ++a)
        for (f = 0; f < n; ++f) {
            fnode* nd = &d->O[a * n + f];
            nd->off = off; nd->out = H; nd->var = -1; nd->nmax = 1; nd->maxp[0] = 2 + f; nd->fixed_mask = 1;
            off += H * H;
        }
    d->ncounts = off;
    d->nvar    = noisy ? A * n : 0;
    c->ncnt    = off + d->nvar;
    c->prior   = (float*)calloc((size_t)c->ncnt, sizeof(float));
    for (a = 0; a < A; ++a) {
        int x;
        for (x = 1; x < W; ++x) c->prior[d->T[a * FS + 0].off + x * W + (x - 1)] = 1; /* agent always moves one column */
        for (y = 0; y < H; ++y) c->prior[d->T[a * FS + 1].off + y * H + ca_keep(c, y + a - 1)] += 1; /* setAgentYTransition */
        for (f = 2; f < FS; ++f) {
            const fnode* nd = &d->T[a * FS + f];
            if (noisy) { /* the base record carries the correct graph; every particle overwrites it */
                ca_fill_obstacle_node(c, c->prior, a, f, 1u << f);
            } else if (!full) {
                for (y = 0; y < H; ++y) ca_obstacle_transition(c, y, c->prior + nd->off + y * H);
            } else { /* every parent-value combination gets obstacleTransition(value of the obstacle itself) */
                int rows = 1, r;
                for (k = 0; k < FS; ++k) rows *= d->Ssz[k];
                for (r = 0; r < rows; ++r) {
                    int v = (r / d->Sstep[f]) % d->Ssz[f];
                    ca_obstacle_transition(c, v, c->prior + nd->off + r * H);
                }
            }
        }
        for (f = 0; f < n; ++f) /* observationDistr(height, y) * 10000 (:46-63, :289-300) */
            for (y = 0; y < H; ++y) {
                float* row = c->prior + d->O[a * n + f].off + y * H;
                int oy;
                row[0] = (float)normal_cdf(-y + .5);
                for (oy = 1; oy < H - 1; ++oy) {
                    int dist = abs(oy - y);
                    row[oy]  = (float)(normal_cdf(dist + .5) - normal_cdf(dist - .5));
                }
                row[H - 1] = (float)normal_cdf(-(H - 1 - y) + .5);
                for (oy = 0; oy < H; ++oy) row[oy] *= 10000;
            }
    }
    return 0;
}


/* SysAdminFactoredPrior (SysAdminFactoredPrior.cpp:17-45, 129-247, 279-333): one state feature per
 * computer (feature c is the model's computer c; in the state index it is bit N-1-c, last feature
 * fastest), transition node (a, c) with parents {c} (independent) or {c-1, c, c+1} (linear),
 * observation node (a) with parent {a mod N}.  No structure prior ("Structure noise is not
 * enabled for the Sysadmin problem"), so no per-particle draws. */
static float sys_failure_probability(const orc_ctx* c, int a, int comp, const int* parents, const int* pv, int np) /* :279-333 */
{
    int k, own = -1, nfn = 0, rebooting = (a == c->sys_N + comp);
    double fail_prob;
    for (k = 0; k < np; ++k)
        if (parents[k] == comp) own = k;
    if (own >= 0 && pv[own] == 0) return rebooting ? 1 - SYS_REBOOT_RATE : 1;
    if (c->cfg.domain == ORC_DOM_SYSADMIN_LINEAR)
        for (k = 0; k < np; ++k)
            if ((parents[k] == comp - 1 || parents[k] == comp + 1) && pv[k] == 0) nfn++;
    fail_prob = 1 - c->sys_keep[nfn];
    if (rebooting) fail_prob *= (1 - SYS_REBOOT_RATE);
    if (own < 0) { /* computer is not its own input: not reached by the two built structures */
        fail_prob += rebooting ? (1 - SYS_REBOOT_RATE) : 1;
        fail_prob *= .5;
    }
    return (float)fail_prob;
}

/* SysAdminFactoredPrior::fullyConnectedT (SysAdminFactoredPrior.cpp:249-277): every transition node
 * with all N computers as parents, counts {p, 1 - p} (a total of ONE per row) with
 * p = SysAdmin::failProbability(state of the parent values, a, c) (SysAdmin.cpp:84-100, float) */
static void sys_fill_fully_connected(orc_ctx* c, float* cnt)
{
    const fdesc* d = &c->fd;
    int A = c->A, N = c->sys_N, a, f, r, k;
    for (a = 0; a < A; ++a)
        for (f = 0; f < N; ++f) {
            const fnode* nd = &d->T[a * N + f];
            for (r = 0; r < (1 << N); ++r) {
                int32_t st = 0; /* SysAdmin::getState(parent values): bit k = value of feature k */
                double fail;
                float p;
                for (k = 0; k < N; ++k) st |= ((r >> (N - 1 - k)) & 1) << k;
                fail = ((st >> f) & 1) ? 1 - c->sys_keep[sys_failing_neighbours(c, f, st)] : 1;
                if (a == N + f) fail *= (1 - SYS_REBOOT_RATE);
                p = (float)fail;
                cnt[nd->off + 2 * r + 0] = p;
                cnt[nd->off + 2 * r + 1] = 1 - p;
            }
            cnt[d->ncounts + nd->var] = u2f((1u << N) - 1u);
        }
}
static int build_sysadmin_factored_prior(orc_ctx* c)
{
    fdesc* d = &c->fd;
    int A = c->A, N = c->sys_N, a, f, k, r, off = 0;
    int linear = c->cfg.domain == ORC_DOM_SYSADMIN_LINEAR;
    int reinvig = is_breeding(c) || is_mh(c);
    if (c->cfg.structure_prior != ORC_SP_NONE) {
        snprintf(c->err, sizeof c->err, "Structure noise is not enabled for the Sysadmin problem");
        return -1;
    }
    if (N > ORC_MAXF) { snprintf(c->err, sizeof c->err, "too many state features"); return -1; }
    d->FS = N;
    d->FO = 1;
    for (f = 0; f < N; ++f) d->Ssz[f] = 2;
    d->Osz[0] = 2;
    fdesc_steps(d->Ssz, d->FS, d->Sstep);
    fdesc_steps(d->Osz, d->FO, d->Ostep);
    d->T = (fnode*)calloc((size_t)A * N, sizeof(fnode));
    d->O = (fnode*)calloc((size_t)A, sizeof(fnode));
    for (a = 0; a < A; ++a)
        for (f = 0; f < N; ++f) {
            fnode* nd = &d->T[a * N + f];
            nd->off = off; nd->out = 2; nd->var = -1; nd->nmax = 0;
            if (reinvig) { /* bred particles choose any parent set: room for all N, the prior's set as a mask */
                uint32_t m = 1u << f;
                if (linear && f > 0) m |= 1u << (f - 1);
                if (linear && f < N - 1) m |= 1u << (f + 1);
                for (k = 0; k < N; ++k) nd->maxp[k] = k;
                nd->nmax = N; nd->var = a * N + f; nd->fixed_mask = m;
                off += 2 << N;
                continue;
            }
            if (linear && f > 0) nd->maxp[nd->nmax++] = f - 1;
            nd->maxp[nd->nmax++] = f;
            if (linear && f < N - 1) nd->maxp[nd->nmax++] = f + 1;
            nd->fixed_mask = (1u << nd->nmax) - 1u;
            off += 2 << nd->nmax;
        }
    for (a = 0; a < A; ++a) {
        fnode* nd = &d->O[a];
        nd->off = off; nd->out = 2; nd->var = -1; nd->nmax = 1; nd->maxp[0] = a % N; nd->fixed_mask = 1;
        off += 4;
    }
    d->ncounts = off;
    d->nvar    = reinvig ? A * N : 0;
    c->ncnt    = off + d->nvar;
    c->prior   = (float*)calloc((size_t)c->ncnt, sizeof(float));
    for (a = 0; a < A; ++a)
        for (f = 0; f < N; ++f) { /* disconnectedTransitions :186-216 / linearTransitions :218-257 */
            const fnode* nd = &d->T[a * N + f];
            int parents[3], pv[3], np = 0;
            for (k = 0; k < nd->nmax; ++k)
                if ((nd->fixed_mask >> k) & 1u) parents[np++] = nd->maxp[k];
            if (reinvig) c->prior[d->ncounts + nd->var] = u2f(nd->fixed_mask);
            for (r = 0; r < (1 << np); ++r) {
                float p;
                for (k = 0; k < np; ++k) pv[k] = (r >> (np - 1 - k)) & 1; /* last parent fastest */
                p = sys_failure_probability(c, a, f, parents, pv, np);
                c->prior[nd->off + 2 * r + 0] = p * 10000.0f;
                c->prior[nd->off + 2 * r + 1] = (1 - p) * 10000.0f;
            }
        }
    for (a = 0; a < A; ++a) { /* precomputeFactoredPrior :148-183: output 0 = FAILING */
        const fnode* nd = &d->O[a];
        c->prior[nd->off + 0] = 10000.0f * SYS_OBSERVE_PROB;       /* parent failing */
        c->prior[nd->off + 1] = 10000.0f * (1 - SYS_OBSERVE_PROB);
        c->prior[nd->off + 2] = 10000.0f * (1 - SYS_OBSERVE_PROB); /* parent working */
        c->prior[nd->off + 3] = 10000.0f * SYS_OBSERVE_PROB;
    }
    return 0;
}

/* one transition node of SysAdminFactoredPrior::computePriorModel (SysAdminFactoredPrior.cpp:98-127) for parent set `mask`:
 * every parent-value row gets {total * p, total * 1 - p} -- the reference's own precedence: total minus p */
static void sys_fill_node(orc_ctx* c, float* cnt, int a, int f, uint32_t mask)
{
    const fdesc* d  = &c->fd;
    const fnode* nd = &d->T[a * c->sys_N + f];
    int parents[ORC_MAXF], pv[ORC_MAXF], np = 0, k, r, rows_max = 1;
    float total = c->cfg.counts_total;
    for (k = 0; k < nd->nmax; ++k) {
        rows_max *= 2;
        if ((mask >> k) & 1u) parents[np++] = nd->maxp[k];
    }
    memset(cnt + nd->off, 0, sizeof(float) * (size_t)rows_max * 2);
    for (r = 0; r < (1 << np); ++r) {
        float p;
        for (k = 0; k < np; ++k) pv[k] = (r >> (np - 1 - k)) & 1; /* last parent fastest */
        p = sys_failure_probability(c, a, f, parents, pv, np);
        cnt[nd->off + 2 * r + 0] = total * p;
        cnt[nd->off + 2 * r + 1] = total * 1 - p;
    }
    if (nd->var >= 0) cnt[d->ncounts + nd->var] = u2f(mask);
}

static int build_factored_prior(orc_ctx* c)
{
    if (is_sys(c->cfg.domain)) return build_sysadmin_factored_prior(c);
    if (is_ca(c->cfg.domain)) return build_ca_factored_prior(c);
    if (is_ftiger(c->cfg.domain)) return build_ftiger_factored_prior(c);
    if (is_grid(c->cfg.domain)) return build_gridworld_factored_prior(c);
    snprintf(c->err, sizeof c->err, "domain %d has no factored prior in the oracle", c->cfg.domain);
    return -1;
}

/* FBAPOMDPPrior::sample -> FactoredTigerFactoredPrior::sampleFBAPOMDPState / sampleFullyConnectedState
 * (FBAPOMDPPrior.cpp:27-37, FactoredTigerPriors.cpp:197-219, 265-291): draws the listen node's
 * parent set (one boolean per state feature, feature 0 forced under match-uniform) */
static void factored_prior_sample(orc_ctx* c, float* cnt)
{
    memcpy(cnt, c->prior, sizeof(float) * (size_t)c->ncnt);
    if (is_sys(c->cfg.domain)) return; /* fixed structures only: no draws */
    if (is_ca(c->cfg.domain)) {
        /* CollisionAvoidanceFactoredPrior::sampleFBAPOMDPState (:349-383): per obstacle, per action, one
         * boolean per state feature (always drawn: it is the left operand of the ||); under
         * match-uniform the obstacle is always its own parent */
        int f, a, fp;
        if (c->cfg.structure_prior != ORC_SP_UNIFORM && c->cfg.structure_prior != ORC_SP_MATCH_UNIFORM) return;
        for (f = 2; f < c->fd.FS; ++f)
            for (a = 0; a < c->A; ++a) {
                uint32_t mask = 0;
                for (fp = 0; fp < c->fd.FS; ++fp) {
                    int b = orc_bool(&c->rng);
                    if (b || (fp == f && c->cfg.structure_prior == ORC_SP_MATCH_UNIFORM)) mask |= 1u << fp;
                }
                ca_fill_obstacle_node(c, cnt, a, f, mask);
            }
        return;
    }
    if (is_grid(c->cfg.domain)) {
        /* GridWorldFactBAPrior::sampleFBAPOMDPState (GridWorldBAPriors.cpp:415-441): per action,
         * one boolean for the x node and one for the y node: add the goal feature as a parent */
        int a, f;
        if (c->cfg.structure_prior != ORC_SP_MATCH_UNIFORM) return;
        for (a = 0; a < c->A; ++a)
            for (f = 0; f < 2; ++f)
                if (orc_bool(&c->rng)) {
                    gw_fill_xy_node(c, cnt, a, f, 1);
                    cnt[c->fd.ncounts + c->fd.T[a * 3 + f].var] = u2f(7u);
                }
        return;
    }
    if (c->cfg.structure_prior == ORC_SP_FULLY_CONNECTED) {
        ftiger_set_observation_model(c, cnt, (1u << c->fd.FS) - 1u);
    } else if (c->cfg.structure_prior == ORC_SP_UNIFORM || c->cfg.structure_prior == ORC_SP_MATCH_UNIFORM) {
        uint32_t mask = 0;
        int f;
        for (f = 0; f < c->fd.FS; ++f)
            if (orc_bool(&c->rng)) mask |= 1u << f;
        if (c->cfg.structure_prior == ORC_SP_MATCH_UNIFORM) mask |= 1u;
        ftiger_set_observation_model(c, cnt, mask);
    }
}

static int sim_step(orc_ctx* c, simstate* st, int32_t a, int32_t* o, double* r, int update)
{
    (*c->step_counter)++;
    switch (c->cfg.model) {
        case ORC_MODEL_POMDP: return domain_step(c, &st->s, a, o, r);
        case ORC_MODEL_BA_TABLE: return ba_table_step(c, st, a, o, r, update);
        case ORC_MODEL_BA_FACTORED: return ba_fact_step(c, st, a, o, r, update);
        default: return 1;
    }
}

/* POMDP::computeObservationProbability for the simulator in use.
 * ref: BAPOMDP.cpp:93-99 -> BAFlatModel::computeObservationProbability BAFlatModel.cpp:106-124
 *      (expectedMult()[o]: float sum, float division) */
static double sim_obs_prob(orc_ctx* c, const simstate* st, int32_t a, int32_t o)
{
    if (c->cfg.model == ORC_MODEL_POMDP) return domain_obs_prob(c, o, a, st->s);
    if (c->cfg.model == ORC_MODEL_BA_FACTORED) return ba_fact_obs_prob(c, st, a, o);
    if (c->cfg.model == ORC_MODEL_BA_TABLE) {
        float tmp[64];
        const float* row = st->cnt + c->phi_len + a * c->S * c->O + st->s * c->O;
        if (c->O == 1) return 1;
        if (c->cfg.dirichlet_regular) {
            orc_sample_mult(c, row, c->O, tmp);
            return tmp[o];
        }
        if (c->O <= 64) {
            orc_expected_mult(row, c->O, tmp);
            return tmp[o];
        } else {
            float sum = row[0];
            int i;
            for (i = 1; i < c->O; ++i) sum += row[i];
            if (sum <= 1e-300) return 0;
            return row[o] / sum;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ tree */

static int tree_alloc(orc_ctx* c)
{
    tree* t       = &c->tr;
    size_t mn     = (size_t)c->cfg.sims + 2;
    size_t A      = (size_t)c->A;
    t->max_nodes  = (int32_t)mn;
    t->visits     = (int32_t*)malloc(mn * sizeof(int32_t));
    t->cn         = (int32_t*)malloc(mn * A * sizeof(int32_t));
    t->cq         = (double*)malloc(mn * A * sizeof(double));
    if ((size_t)c->A * c->O <= 64) {
        t->child = (int32_t*)malloc(mn * A * c->O * sizeof(int32_t));
        t->hkey  = NULL;
        t->hval  = NULL;
    } else {
        uint64_t cap = 1;
        while (cap < 4 * mn) cap <<= 1;
        t->child = NULL;
        t->hkey  = (uint64_t*)malloc(cap * sizeof(uint64_t));
        t->hval  = (int32_t*)malloc(cap * sizeof(int32_t));
        t->hmask = cap - 1;
    }
    return 0;
}

static void tree_reset(orc_ctx* c)
{
    tree* t   = &c->tr;
    t->n_nodes = 0;
    t->tree_depth = 0;
    if (t->hkey) memset(t->hkey, 0, (t->hmask + 1) * sizeof(uint64_t));
}

/* ref: POUCT::createActionNode POUCT.cpp:305-315 + ActionNode ctor MCTSTreeNodes.cpp:52-57 */
static int32_t tree_new_node(orc_ctx* c)
{
    tree* t   = &c->tr;
    int32_t n = t->n_nodes++;
    int a;
    t->visits[n] = 0;
    for (a = 0; a < c->A; ++a) {
        t->cn[n * c->A + a] = 0;
        t->cq[n * c->A + a] = 0;
    }
    if (t->child) {
        int k, m = c->A * c->O;
        for (k = 0; k < m; ++k) t->child[(size_t)n * m + k] = -1;
    }
    return n;
}

static int32_t tree_get_child(orc_ctx* c, int32_t node, int32_t a, int32_t o)
{
    tree* t = &c->tr;
    if (t->child) return t->child[((size_t)node * c->A + a) * c->O + o];
    {
        uint64_t key = ((uint64_t)node * c->A + a) * c->O + o + 1;
        uint64_t h   = mix64(key) & t->hmask;
        while (t->hkey[h]) {
            if (t->hkey[h] == key) return t->hval[h];
            h = (h + 1) & t->hmask;
        }
        return -1;
    }
}

static void tree_set_child(orc_ctx* c, int32_t node, int32_t a, int32_t o, int32_t ch)
{
    tree* t = &c->tr;
    if (t->child) {
        t->child[((size_t)node * c->A + a) * c->O + o] = ch;
        return;
    }
    {
        uint64_t key = ((uint64_t)node * c->A + a) * c->O + o + 1;
        uint64_t h   = mix64(key) & t->hmask;
        while (t->hkey[h]) h = (h + 1) & t->hmask;
        t->hkey[h] = key;
        t->hval[h] = ch;
    }
}

/* ref: POUCT::UCB/initiateUCBTable POUCT.cpp:131-136,330-338: table[m][n] = u*sqrt(log1p(m)/n),
 *      table[m][0] = DBL_MAX.  Evaluated on the fly (value-identical to the table entry). */
static double ucb(const orc_ctx* c, int m, int n)
{
    if (n == 0) return DBL_MAX;
    return c->cfg.exploration * sqrt(c->log1p_tab[m] / n);
}

/* ref: POUCT::selectChanceNodeUCB POUCT.cpp:138-181 (= RBAPOUCT.cpp:162-205).
 * Always draws one slowRandomInt for the tie-break, even with a single candidate. */
static int32_t select_chance_ucb(orc_ctx* c, int32_t node, int explore)
{
    int32_t best[ORC_MAX_ACTIONS];
    int nbest     = 0, a;
    double best_q = -DBL_MAX;
    int m         = c->tr.visits[node];
    for (a = 0; a < c->A; ++a) {
        double q = c->tr.cq[node * c->A + a];
        if (explore) q += ucb(c, m, c->tr.cn[node * c->A + a]);
        if (q >= best_q) {
            if (q > best_q) nbest = 0;
            best_q        = q;
            best[nbest++] = a;
        }
    }
    return best[orc_slow_int(&c->rng, 0, nbest)];
}

/* ref: POUCT::rollout POUCT.cpp:273-303 (= RBAPOUCT.cpp:295-323) */
static double rollout(orc_ctx* c, simstate* st, int depth_to_go)
{
    double ret = 0, disc = 1, r = 0;
    int term = 0;
    int32_t o;
    while (depth_to_go > 0 && !term) {
        int32_t a = domain_random_action(c, st->s);
        term      = sim_step(c, st, a, &o, &r, 0);
        ret += r * disc;
        disc *= c->gamma;
        depth_to_go--;
    }
    return ret;
}

static double traverse_action(orc_ctx* c, int32_t node, simstate* st, int depth_to_go);

/* ref: ChanceNode::addVisit MCTSTreeNodes.cpp:8-12: n++; q += (r - q) / n  (pinned by golden["mcts_nodes"]) */
void orc_chance_add_visit(int32_t* n, double* q, double ret)
{
    (*n)++;
    *q += (ret - *q) / *n;
}
/* BADomainExtension::terminal / reward of the ctx's domain (pinned by golden["ext_*"]) */
int orc_ext_terminal(orc_ctx* c, int32_t s, int32_t a, int32_t ns) { return ext_terminal(c, s, a, ns); }
double orc_ext_reward(orc_ctx* c, int32_t s, int32_t a, int32_t ns) { return ext_reward(c, s, a, ns); }

/* ref: POUCT::traverseChanceNode POUCT.cpp:210-257 (= RBAPOUCT.cpp:233-277) */
static double traverse_chance(orc_ctx* c, int32_t node, int32_t a, simstate* st, int depth_to_go)
{
    int32_t o;
    double r = 0, delayed = 0, ret;
    int term = sim_step(c, st, a, &o, &r, 0);
    if (!term) {
        int32_t ch = tree_get_child(c, node, a, o);
        if (ch >= 0) {
            delayed = traverse_action(c, ch, st, depth_to_go - 1);
        } else {
            ch = tree_new_node(c);
            tree_set_child(c, node, a, o, ch);
            delayed = rollout(c, st, depth_to_go - 1);
        }
    }
    ret = r + c->gamma * delayed;
    orc_chance_add_visit(&c->tr.cn[node * c->A + a], &c->tr.cq[node * c->A + a], ret);
    return ret;
}

/* ref: POUCT::traverseActionNode POUCT.cpp:183-208 (= RBAPOUCT.cpp:207-231) */
static double traverse_action(orc_ctx* c, int32_t node, simstate* st, int depth_to_go)
{
    int32_t a;
    double ret;
    int d = c->tr.max_tree_depth - depth_to_go;
    if (d > c->tr.tree_depth) c->tr.tree_depth = d;
    if (depth_to_go == 0) return 0;
    a   = select_chance_ucb(c, node, 1);
    ret = traverse_chance(c, node, a, st, depth_to_go);
    c->tr.visits[node]++; /* ActionNode::addVisit, after the recursion */
    return ret;
}

/* ------------------------------------------------------------------ filters */

/* Canonical summation order of the HIP engine (DESIGN.md "device-order sums"):
 * groups of 4 consecutive elements summed sequentially per lane, a 64-lane Kogge-Stone
 * inclusive scan across the lane sums of one 256-element chunk, chunks chained sequentially.
 * Writes the inclusive prefix sums; returns the total. */
static double dev_scan(const double* w, int n, double* incl)
{
    double carry = 0;
    int base;
    for (base = 0; base < n; base += 256) {
        double lane[64], pre[64];
        int l, k, d;
        for (l = 0; l < 64; ++l) {
            double s = 0;
            for (k = 0; k < 4; ++k) {
                int i = base + 4 * l + k;
                s     = (k == 0) ? ((i < n) ? w[i] : 0.0) : s + ((i < n) ? w[i] : 0.0);
            }
            lane[l] = s;
        }
        for (d = 1; d < 64; d <<= 1) { /* Kogge-Stone inclusive */
            for (l = 0; l < 64; ++l) pre[l] = (l >= d) ? lane[l] + lane[l - d] : lane[l];
            memcpy(lane, pre, sizeof lane);
        }
        for (l = 0; l < 64; ++l) {
            double excl = (l > 0) ? lane[l - 1] : 0.0;
            double run  = carry + excl;
            for (k = 0; k < 4; ++k) {
                int i = base + 4 * l + k;
                if (i < n) {
                    run += w[i];
                    if (incl) incl[i] = run;
                }
            }
        }
        carry = carry + lane[63];
    }
    return carry;
}

/* FlatFilter::sample.  ref: src/beliefs/particle_filters/FlatFilter.cpp:97-102 */
static int32_t flat_sample(orc_ctx* c) { return c->point ? 0 : orc_int(&c->rng, c->cfg.particles); }

/* WeightedFilter::sample.  ref: src/beliefs/particle_filters/WeightedFilter.cpp:163-191
 * (scan from the back, strict >, index 0 is the fall-through).
 * DEV order: largest i >= 1 whose exclusive device-order prefix sum is < threshold, else 0. */
static int32_t weighted_sample(orc_ctx* c)
{
    int n            = c->cfg.particles;
    double threshold = orc_u01(&c->rng) * c->total_w;
    if (c->cfg.arith == ORC_ARITH_REF) {
        int32_t sample   = n - 1;
        double remaining = c->total_w;
        for (; sample > 0; --sample) {
            remaining -= c->P[sample].w;
            if (threshold > remaining) break;
        }
        return sample;
    } else {
        /* wscan[i] = inclusive prefix; exclusive prefix of i is wscan[i-1] */
        int lo = 0, hi = n - 1; /* find largest i in [1,n-1] with wscan[i-1] < threshold */
        while (lo < hi) {
            int mid = (lo + hi + 1) >> 1;
            if (c->wscan[mid - 1] < threshold) lo = mid; else hi = mid - 1;
        }
        return lo;
    }
}

static void weighted_refresh_scan(orc_ctx* c)
{
    int i, n = c->cfg.particles;
    if (c->cfg.arith != ORC_ARITH_DEV) return;
    for (i = 0; i < n; ++i) c->wscratch[i] = c->P[i].w;
    c->total_w = dev_scan(c->wscratch, n, c->wscan);
}

/* the main filter is a WeightedFilter (importance sampling; the cheating belief's _belief) */
static int is_nested(const orc_ctx* c) { return c->cfg.belief == ORC_BELIEF_NESTED; }
static int is_weighted(const orc_ctx* c) { return c->cfg.belief == ORC_BELIEF_IMPORTANCE || c->cfg.belief == ORC_BELIEF_CHEATING || is_mh(c) || is_nested(c); }

static int32_t belief_sample(orc_ctx* c)
{
    return is_weighted(c) ? weighted_sample(c) : flat_sample(c);
}
/* the domain state that comes with belief sample `src`: the particle's own, or -- NestedBelief::sample
 * (NestedBelief.cpp:116-125) -- one of the count particle's flat filter of domain states */
static int32_t belief_sample_state(orc_ctx* c, int32_t src)
{
    if (!is_nested(c)) return c->P[src].s;
    return c->nest_s[(size_t)src * c->nest_m + orc_int(&c->rng, c->nest_m)];
}

static void swap_pools(orc_ctx* c)
{
    particle* p = c->P;
    float* q    = c->pool;
    c->P        = c->Pnew;
    c->Pnew     = p;
    c->pool     = c->pool_new;
    c->pool_new = q;
}

/* simulator.sampleStartState() into slot i.
 * ref: BAPOMDP::sampleStartState BAPOMDP.cpp:101-104 -> prior->sample(domain start state) */
static void sample_start_into(orc_ctx* c, particle* p)
{
    p->s = domain_start(c);
    if (c->cfg.model == ORC_MODEL_BA_TABLE) memcpy(p->cnt, c->prior, sizeof(float) * c->ncnt);
    if (c->cfg.model == ORC_MODEL_BA_FACTORED) factored_prior_sample(c, p->cnt);
}

/* Belief::initiate.  ref: RejectionSampling.cpp:15-20 / BARejectionSampling (FlatFilter(n, alloc));
 *                         ImportanceSampler.cpp:45-55 / BAImportanceSampling (add(start, 1/n)) */
static void incubator_initiate_shadow(orc_ctx* c);
static void belief_initiate(orc_ctx* c)
{
    int i, n = c->cfg.particles;
    double w = 1.0 / (double)n;
    c->total_w = 0;
    if (is_nested(c)) {
        /* NestedBelief::initiate (NestedBelief.cpp:61-87): WeightedFilter(n, alloc) -- weights 1/n, total weight 1 -- of
         * pairs {sampleStartState(), FlatFilter(n^2, sampleDomainState)}; g++ builds the pair's second member first.  The
         * count particle's own domain state is released at once: kept as 0 here. */
        int j;
        for (i = 0; i < n; ++i) {
            orc_rng_stream(&c->rng, ORC_PH_INIT_FC, (uint32_t)i);
            for (j = 0; j < c->nest_m; ++j) c->nest_s[(size_t)i * c->nest_m + j] = domain_start(c);
            orc_rng_stream(&c->rng, ORC_PH_INIT, (uint32_t)i);
            sample_start_into(c, &c->P[i]);
            c->P[i].s = 0;
            c->P[i].w = w;
        }
        c->total_w = 1;
        weighted_refresh_scan(c);
        return;
    }
    for (i = 0; i < n; ++i) {
        orc_rng_stream(&c->rng, ORC_PH_INIT, (uint32_t)i);
        sample_start_into(c, &c->P[i]);
        c->P[i].w = w;
        c->total_w += w; /* WeightedFilter::add(T, w) WeightedFilter.cpp:60-66 */
    }
    if (is_weighted(c)) weighted_refresh_scan(c);
    if (is_mh(c)) {  /* MHwithinGibbs::initiate :277-294 (free() :296-312 clears the history); MHNIPS2018.cpp:132-175 likewise */
        c->mh_n_ep      = 1;
        c->mh_ep_len[0] = 0;
        c->log_lik      = 0;
    }
    if (c->cfg.belief == ORC_BELIEF_CHEATING) {
        /* CheatingReinvigoration::initiate (CheatingReinvigoration.cpp:64-90): a second FlatFilter of
         * fbapomdp.sampleCorrectGraphState() particles (the prior's own structure: the base record) */
        for (i = 0; i < n; ++i) {
            orc_rng_stream(&c->rng, ORC_PH_INIT_FC, (uint32_t)i);
            c->F[i].s = domain_start(c);
            c->F[i].w = 0;
            memcpy(c->F[i].cnt, c->prior, sizeof(float) * (size_t)c->ncnt);
        }
        c->likelihood = 1;
    }
    if (is_breeding(c)) {
        /* ReinvigoratingRejectionSampling::initiate (ReinvigoratingRejectionSampling.cpp:55-76):
         * after the n start states, n x FBAPOMDP::sampleFullyConnectedState (FBAPOMDP.cpp:63-67 ->
         * FactoredTigerFactoredPrior::sampleFullyConnectedState FactoredTigerPriors.cpp:324-337) */
        for (i = 0; i < n; ++i) {
            orc_rng_stream(&c->rng, ORC_PH_INIT_FC, (uint32_t)i);
            c->F[i].s = domain_start(c);
            c->F[i].w = 0;
            memcpy(c->F[i].cnt, c->prior, sizeof(float) * (size_t)c->ncnt);
            if (is_sys(c->cfg.domain)) { /* SysAdminFactoredPrior::sampleFullyConnectedState :57-69 */
                sys_fill_fully_connected(c, c->F[i].cnt);
            } else if (is_ca(c->cfg.domain)) { /* CollisionAvoidanceFactoredPrior::sampleFullyConnectedState :429-440 */
                int a, f;
                for (a = 0; a < c->A; ++a)
                    for (f = 2; f < c->fd.FS; ++f) ca_fill_obstacle_node(c, c->F[i].cnt, a, f, (1u << c->fd.FS) - 1u);
            } else {
                ftiger_set_observation_model(c, c->F[i].cnt, (1u << c->fd.FS) - 1u);
            }
        }
        if (c->cfg.belief == ORC_BELIEF_INCUBATOR) incubator_initiate_shadow(c); /* StructureIncubatorSampling::initiate :64-90 */
    }
}

/* beliefs::rejectSample.  ref: src/beliefs/particle_filters/RejectionSampling.hpp:26-72 */
static void reject_sample(orc_ctx* c, int32_t a, int32_t o, int phase)
{
    int n = c->cfg.particles, acc = 0, count = 0;
    c->step_counter = &c->belief_steps;
    while (acc < n) {
        int32_t src, so;
        double r;
        simstate st;
        particle* dst = &c->Pnew[acc];
        orc_rng_stream(&c->rng, phase, (uint32_t)count);
        src  = flat_sample(c);
        st.s = c->P[src].s;
        st.cnt = dst->cnt;
        if (c->ncnt) memcpy(dst->cnt, c->P[src].cnt, sizeof(float) * c->ncnt); /* copyState */
        sim_step(c, &st, a, &so, &r, 1); /* BA: mode is UpdateCounts outside the planner */
        if (so == o) {
            dst->s = st.s;
            dst->w = 0;
            acc++;
        }
        count++;
    }
    c->last_update_count = count;
    swap_pools(c);
}

/* importance_sampling::update.
 * ref: src/beliefs/particle_filters/ImportanceSampler.hpp:31-62;
 *      WeightedFilter::normalize WeightedFilter.cpp:130-143 */
static double is_update(orc_ctx* c, int32_t a, int32_t o)
{
    int i, n = c->cfg.particles;
    double total = 0, accw = 0;
    c->step_counter = &c->belief_steps;
    for (i = 0; i < n; ++i) {
        int32_t so;
        double r;
        simstate st;
        orc_rng_stream(&c->rng, ORC_PH_IS_UPDATE, (uint32_t)i);
        st.s   = c->P[i].s;
        st.cnt = c->P[i].cnt;
        sim_step(c, &st, a, &so, &r, 1);
        c->P[i].s = st.s;
        c->P[i].w *= sim_obs_prob(c, &st, a, o);
        total += c->P[i].w;
    }
    if (c->cfg.arith == ORC_ARITH_DEV) {
        for (i = 0; i < n; ++i) c->wscratch[i] = c->P[i].w;
        total = dev_scan(c->wscratch, n, NULL);
    }
    c->last_weight_total = total;
    for (i = 0; i < n; ++i) {
        c->P[i].w /= total;
        accw += c->P[i].w;
    }
    c->total_w = accw;
    weighted_refresh_scan(c); /* DEV: total and prefix sums in device order */
    return total;
}

/* importance_sampling::resample.
 * ref: src/beliefs/particle_filters/ImportanceSampler.hpp:71-94 (n multinomial draws through
 *      WeightedFilter::sample, deep copies, weight 1/n; WeightedFilter::add accumulates total) */
static void is_resample(orc_ctx* c)
{
    int i, n = c->cfg.particles;
    double w1 = 1.0 / (double)n, new_total = 0;
    for (i = 0; i < n; ++i) {
        int32_t src;
        orc_rng_stream(&c->rng, ORC_PH_RESAMPLE, (uint32_t)i);
        src          = weighted_sample(c);
        c->Pnew[i].s = c->P[src].s;
        c->Pnew[i].w = w1;
        if (c->ncnt) memcpy(c->Pnew[i].cnt, c->P[src].cnt, sizeof(float) * c->ncnt);
        new_total += w1;
    }
    swap_pools(c);
    c->total_w = new_total;
    weighted_refresh_scan(c);
    c->last_update_count = -1;
}

/* the main filter <-> the fully connected filter */
static void swap_main_fc(orc_ctx* c)
{
    particle* p; float* q;
    p = c->P; c->P = c->F; c->F = p;
    p = c->Pnew; c->Pnew = c->Fnew; c->Fnew = p;
    q = c->pool; c->pool = c->fpool; c->fpool = q;
    q = c->pool_new; c->pool_new = c->fpool_new; c->fpool_new = q;
}

/* DBNNode::marginalizeOut (DBNNode.cpp:40-80) on a max-layout node: `src` holds the node's CPT
 * under parent mask `om`, `dst` receives the CPT under mask `nm`.  Rows of the source are visited
 * in ascending order and added (float) into the row the new node's cptIndex maps their parent
 * values to; the reference reads those parent values as graph input (DBNNode.cpp:62), so the
 * source must hold every parent of the target: nm is a subset of om. */
static void node_marginalize(const orc_ctx* c, const fnode* nd, const float* src, uint32_t om, float* dst, uint32_t nm)
{
    int j, v, r, rows_old = 1, rows_max = 1, fv[ORC_MAXF] = {0};
    for (j = 0; j < nd->nmax; ++j) {
        rows_max *= c->fd.Ssz[nd->maxp[j]];
        if ((om >> j) & 1u) rows_old *= c->fd.Ssz[nd->maxp[j]];
    }
    memset(dst + nd->off, 0, sizeof(float) * (size_t)rows_max * nd->out);
    for (r = 0; r < rows_old; ++r) {
        int rem = r, nr;
        for (j = nd->nmax - 1; j >= 0; --j) /* last parent is the fastest digit (index.cpp:51-83) */
            if ((om >> j) & 1u) {
                fv[nd->maxp[j]] = rem % c->fd.Ssz[nd->maxp[j]];
                rem /= c->fd.Ssz[nd->maxp[j]];
            }
        nr = node_row(c, nd, nm, fv);
        for (v = 0; v < nd->out; ++v) dst[nr + v] += src[nd->off + r * nd->out + v];
    }
}

/* breed (ReinvigoratingRejectionSampling.cpp:24-35): BABNModel::marginalizeOut (BABNModel.cpp:205-229)
 * of the counts particle onto the mutated structure.  Nodes whose parents are fixed keep the counts
 * particle's CPT ("exactly same parents need no marginalizing", DBNNode.cpp:45-48). */
static void breed_counts(const orc_ctx* c, const float* counts_cnt, const uint32_t* new_masks, float* out)
{
    const fdesc* d = &c->fd;
    int k, nn = c->A * d->FS + c->A * d->FO;
    memcpy(out, counts_cnt, sizeof(float) * (size_t)c->ncnt);
    for (k = 0; k < nn; ++k) {
        const fnode* nd = k < c->A * d->FS ? &d->T[k] : &d->O[k - c->A * d->FS];
        if (nd->var < 0) continue;
        node_marginalize(c, nd, counts_cnt, f2u(counts_cnt[d->ncounts + nd->var]), out, new_masks[nd->var]);
        out[d->ncounts + nd->var] = u2f(new_masks[nd->var]);
    }
}

/* ReinvigoratingRejectionSampling::reinvigorateParticles (ReinvigoratingRejectionSampling.cpp:121-131).
 * Draw order of one iteration: g++ evaluates breed's arguments right to left, so
 * _fully_connected_belief.sample() comes first, then _belief.sample(); then the mutation
 * (FactoredTigerFactoredPrior::mutate FactoredTigerPriors.cpp:353-381 ->
 * BABNModel::Structure::flip_random_edge BABNModel.cpp:16-31: one slowRandomInt over the state
 * features, flips that parent of the listen observation node); then FlatFilter::replace picks the
 * victim (FlatFilter.cpp:39-46).  Iterations are sequential: a bred particle can be sampled by the next. */
/* breed (:24-35) into c->breed_tmp; returns the structure particle's index in the main filter */
static int32_t breed_one(orc_ctx* c)
{
    int k, n = c->cfg.particles;
    const fdesc* d = &c->fd;
    uint32_t masks[128];
    int32_t fc, b, edge;
    fc = orc_int(&c->rng, n);
    b  = orc_int(&c->rng, n);
    for (k = 0; k < d->nvar; ++k) masks[k] = f2u(c->P[b].cnt[d->ncounts + k]);
    if (is_sys(c->cfg.domain)) {
        /* SysAdminFactoredPrior::mutate (:47-55): flip_random_edge(&T[action()][computer()], N).  Under the
         * reference's --std=c++11 g++ evaluates the second subscript first: computer, action, edge */
        int mc = orc_int(&c->rng, c->sys_N), ma = orc_int(&c->rng, c->A);
        edge = orc_slow_int(&c->rng, 0, d->FS);
        masks[ma * c->sys_N + mc] ^= 1u << edge;
    } else if (is_ca(c->cfg.domain)) { /* CollisionAvoidanceFactoredPrior::mutate :455-488: action, obstacle, then the edge */
        int ma = orc_int(&c->rng, c->A), mo = orc_int(&c->rng, c->ca_n);
        edge = orc_slow_int(&c->rng, 0, d->FS);
        masks[ma * c->ca_n + mo] ^= 1u << edge;
    } else {
        edge = orc_slow_int(&c->rng, 0, d->FS);
        masks[0] ^= 1u << edge;
    }
    breed_counts(c, c->F[fc].cnt, masks, c->breed_tmp);
    return b;
}
static void reinvigorate(orc_ctx* c)
{
    int i, n = c->cfg.particles;
    for (i = 0; i < c->cfg.resample_amount; ++i) {
        int32_t b, victim;
        orc_rng_stream(&c->rng, ORC_PH_REINVIG, (uint32_t)i);
        b      = breed_one(c);
        victim = orc_int(&c->rng, n);
        c->P[victim].s = c->P[b].s; /* copyDomainState(structure_state->_domain_state) */
        memcpy(c->P[victim].cnt, c->breed_tmp, sizeof(float) * (size_t)c->ncnt);
    }
}

/* ------------------------------------------------------------------ incubator belief
 * ref: src/beliefs/bayes-adaptive/factored/StructureIncubatorSampling.cpp (FBAPOMDP.hpp -> Boost: restated).
 * The two rejection filters of the reinvigoration belief plus a weighted "shadow" filter of bred particles, updated
 * by importance sampling + resampling.  Per update: shadow particles whose normalised weight exceeds --threshold
 * replace random particles of the main filter (reinvigorateBelief :155-188), the --resample-amount least likely shadow
 * particles are bred anew (reinvigorateShadowBelief :137-153), then the three filters are updated (:105-131).
 * As written, the shadow weights are uniform whenever they are tested (every update ends in a resample), so either
 * none is promoted or all are -- and then the shadow's total weight is zero and normalize() divides by it; contexts
 * whose threshold would do that are refused at create. */
static void swap_main_sh(orc_ctx* c)
{
    particle* p; float* q;
    p = c->P; c->P = c->Sh; c->Sh = p;
    p = c->Pnew; c->Pnew = c->Shnew; c->Shnew = p;
    q = c->pool; c->pool = c->shpool; c->shpool = q;
    q = c->pool_new; c->pool_new = c->shpool_new; c->shpool_new = q;
}
static void incubator_initiate_shadow(orc_ctx* c)
{
    int i, n = c->cfg.particles;
    double w = 1.0 / (double)n;
    c->sh_total = 0;
    for (i = 0; i < n; ++i) { /* :81-89: _shadow_belief.add(breed(...), 1 / size) */
        int32_t b;
        orc_rng_stream(&c->rng, ORC_PH_INIT_SH, (uint32_t)i);
        b = breed_one(c);
        c->Sh[i].s = c->P[b].s;
        c->Sh[i].w = w;
        memcpy(c->Sh[i].cnt, c->breed_tmp, sizeof(float) * (size_t)c->ncnt);
        c->sh_total += w;
    }
    swap_main_sh(c);
    c->total_w = c->sh_total;
    weighted_refresh_scan(c);   /* DEV: total in device order */
    c->sh_total = c->total_w;
    swap_main_sh(c);
}
static void incubator_update(orc_ctx* c, int32_t a, int32_t o)
{
    int n = c->cfg.particles, i, k, added = 0, count;
    int order[1024];
    /* reinvigorateBelief (:155-188) */
    orc_rng_stream(&c->rng, ORC_PH_REINVIG, 0xffffu);
    for (i = 0; i < n; ++i)
        if (c->Sh[i].w / c->sh_total > c->cfg.threshold) {
            int victim = orc_int(&c->rng, n);   /* FlatFilter::replace (FlatFilter.cpp:39-46) */
            added = 1;
            c->P[victim].s = c->Sh[i].s;
            memcpy(c->P[victim].cnt, c->Sh[i].cnt, sizeof(float) * (size_t)c->ncnt);
            c->Sh[i].w = 0;
        }
    if (added) { /* WeightedFilter::normalize (WeightedFilter.cpp:113-143) */
        double tot = 0, acc = 0;
        for (i = 0; i < n; ++i) tot += c->Sh[i].w;
        for (i = 0; i < n; ++i) { c->Sh[i].w /= tot; acc += c->Sh[i].w; }
        c->sh_total = acc;
    }
    /* reinvigorateShadowBelief (:137-153): WeightedFilter::replace(i, bred) gives the weight total / size */
    for (i = 0; i < n; ++i) c->wscratch[i] = c->Sh[i].w;
    orc_least_likely(c->wscratch, n, c->cfg.resample_amount, order);
    for (k = 0; k < c->cfg.resample_amount; ++k) {
        int32_t b;
        double w;
        i = order[k];
        orc_rng_stream(&c->rng, ORC_PH_REINVIG, (uint32_t)k);
        b = breed_one(c);
        c->Sh[i].s = c->P[b].s;
        memcpy(c->Sh[i].cnt, c->breed_tmp, sizeof(float) * (size_t)c->ncnt);
        w = c->sh_total / (double)n;
        c->sh_total += w - c->Sh[i].w;
        c->Sh[i].w = w;
    }
    /* the three filters (:119-127) */
    reject_sample(c, a, o, ORC_PH_REJECT);
    count = c->last_update_count;
    swap_main_fc(c);
    reject_sample(c, a, o, ORC_PH_REJECT_FC);
    swap_main_fc(c);
    swap_main_sh(c);
    c->total_w = c->sh_total;
    is_update(c, a, o);
    is_resample(c);
    c->sh_total = c->total_w;
    swap_main_sh(c);
    c->last_update_count = count;
}

/* ------------------------------------------------------------------ MH-within-Gibbs belief
 * ref: src/beliefs/bayes-adaptive/factored/MHwithinGibbs.cpp (needs FBAPOMDP.hpp -> Boost: restated, not built).
 * A weighted filter updated by importance sampling + resampling; the run's history of (a, o) by episode; when the
 * accumulated log-likelihood falls below --threshold the whole filter is re-drawn by a Metropolis-Hastings chain over
 * structures (scored by LogBDScore on the counts a sampled state history gives) inside a Gibbs loop over state
 * histories.  One Philox stream (REINVIG, 0) serves the chain: its draws are sequential by definition.
 * Built for the factored-tiger prior (computePriorModel FactoredTigerPriors.cpp:293-321, mutate :351-381). */

#define MH_MAXVAR 128
/* the structure of a particle: the parent-set words of its variable nodes */
static void mh_structure_of(const orc_ctx* c, const float* cnt, uint32_t* masks)
{
    int v;
    for (v = 0; v < c->fd.nvar; ++v) masks[v] = f2u(cnt[c->fd.ncounts + v]);
}
/* FBAPOMDPPrior::computePriorModel(structure): factored tiger (FactoredTigerPriors.cpp:293-321) = the prior with the
 * listen observation node set for its parent set; collision avoidance (CollisionAvoidancePriors.cpp:490-526) = the
 * prior with every obstacle's transition node, per action, set for its parent set (sampleBlockTModel) */
static void mh_compute_prior(orc_ctx* c, const uint32_t* masks, float* out)
{
    memcpy(out, c->prior, sizeof(float) * (size_t)c->ncnt);
    if (is_sys(c->cfg.domain)) { /* SysAdminFactoredPrior::computePriorModel (:98-127): every transition node anew, the observation nodes as they are */
        int a, f;
        for (a = 0; a < c->A; ++a)
            for (f = 0; f < c->sys_N; ++f) sys_fill_node(c, out, a, f, masks[a * c->sys_N + f]);
    } else if (is_grid(c->cfg.domain)) { /* GridWorldFactBAPrior::computePriorModel (GridWorldBAPriors.cpp:227-254): setNoisyTransitionNode where the goal is a parent */
        int a, f;
        for (a = 0; a < c->A; ++a)
            for (f = 0; f < 2; ++f) {
                int var = c->fd.T[a * 3 + f].var;
                if (masks[var] == 7u) {
                    gw_fill_xy_node(c, out, a, f, 1);
                    out[c->fd.ncounts + var] = u2f(7u);
                }
            }
    } else if (is_ca(c->cfg.domain)) {
        int a, f;
        for (f = 2; f < c->fd.FS; ++f)
            for (a = 0; a < c->A; ++a) ca_fill_obstacle_node(c, out, a, f, masks[a * c->ca_n + (f - 2)]);
    } else {
        ftiger_set_observation_model(c, out, masks[0]);
    }
}
/* FBAPOMDP::mutate -> FactoredTigerFactoredPrior::mutate (FactoredTigerPriors.cpp:351-381): flip_random_edge
 * (BABNModel.cpp:16-31) of O[listen][0]; CollisionAvoidanceFactoredPrior::mutate (CollisionAvoidancePriors.cpp:455-488):
 * an action, an obstacle, then flip_random_edge of that transition node */
static void mh_mutate(orc_ctx* c, uint32_t* masks)
{
    if (is_sys(c->cfg.domain)) { /* SysAdminFactoredPrior::mutate (:47-55); g++ evaluates the second subscript first: computer, action, then the edge */
        int mc = orc_int(&c->rng, c->sys_N), ma = orc_int(&c->rng, c->A);
        masks[ma * c->sys_N + mc] ^= 1u << orc_slow_int(&c->rng, 0, c->fd.FS);
    } else if (is_grid(c->cfg.domain)) { /* GridWorldFactBAPrior::mutate (GridWorldBAPriors.cpp:200-225): an action, the x or the y node, the goal edge toggled */
        int a = orc_slow_int(&c->rng, 0, c->A);
        int f = orc_slow_int(&c->rng, 0, 2);
        masks[c->fd.T[a * 3 + f].var] ^= 4u;
    } else if (is_ca(c->cfg.domain)) {
        int a  = orc_int(&c->rng, c->A);
        int ob = orc_int(&c->rng, c->ca_n);
        masks[a * c->ca_n + ob] ^= 1u << orc_slow_int(&c->rng, 0, c->fd.FS);
    } else {
        masks[0] ^= 1u << orc_slow_int(&c->rng, 0, c->fd.FS);
    }
}
/* BABNModel::incrementCountsOf (BABNModel.cpp:354-382): observation rows at the OLD state's parent values (App. A #6) */
static void fact_increment(orc_ctx* c, float* cnt, int32_t s, int32_t a, int32_t o, int32_t ns, float amount)
{
    const fdesc* d = &c->fd;
    int fv[ORC_MAXF], nf[ORC_MAXF], of[ORC_MAXF], f;
    features_of(s, d->Sstep, d->FS, fv);
    features_of(ns, d->Sstep, d->FS, nf);
    features_of(o, d->Ostep, d->FO, of);
    for (f = 0; f < d->FS; ++f) {
        const fnode* nd = &d->T[a * d->FS + f];
        cnt[node_row(c, nd, node_mask(c, nd, cnt), fv) + nf[f]] += amount;
    }
    for (f = 0; f < d->FO; ++f) {
        const fnode* nd = &d->O[a * d->FO + f];
        cnt[node_row(c, nd, node_mask(c, nd, cnt), fv) + of[f]] += amount;
    }
}
/* MHwithinGibbs::computePosteriorCounts (:397-436): the prior's counts + one incrementCountsOf per step of the history
 * along the state sequence (one more state than steps per episode) */
static void mh_posterior(orc_ctx* c, const float* prior, const int32_t* seq, float* out)
{
    int e, t, k = 0, h = 0;
    memcpy(out, prior, sizeof(float) * (size_t)c->ncnt);
    for (e = 0; e < c->mh_n_ep; ++e) {
        for (t = 0; t < c->mh_ep_len[e]; ++t, ++h, ++k) fact_increment(c, out, seq[k], c->mh_a[h], c->mh_o[h], seq[k + 1], 1);
        k++;
    }
}
/* BABNModel::flattenT / flattenO (BABNModel.cpp:89-181): T[s][a][s'] = prod_f expectedMult(row_f(s))[s'_f] in float,
 * features in order; O[a][s'][o] likewise over the observation features */
static void mh_flatten(orc_ctx* c, const float* model)
{
    const fdesc* d = &c->fd;
    int S = c->S, A = c->A, O = c->O, a, s, ns, o, f;
    float e[ORC_MAXF][16];
    int fv[ORC_MAXF], nf[ORC_MAXF];
    for (a = 0; a < A; ++a)
        for (s = 0; s < S; ++s) {
            features_of(s, d->Sstep, d->FS, fv);
            for (f = 0; f < d->FS; ++f) {
                const fnode* nd = &d->T[a * d->FS + f];
                orc_expected_mult(model + node_row(c, nd, node_mask(c, nd, model), fv), nd->out, e[f]);
            }
            for (ns = 0; ns < S; ++ns) {
                float p = 1;
                features_of(ns, d->Sstep, d->FS, nf);
                for (f = 0; f < d->FS; ++f) p *= e[f][nf[f]];
                c->mh_T[((size_t)s * A + a) * S + ns] = p;
            }
            for (f = 0; f < d->FO; ++f) {   /* (s plays the new state here) */
                const fnode* nd = &d->O[a * d->FO + f];
                orc_expected_mult(model + node_row(c, nd, node_mask(c, nd, model), fv), nd->out, e[f]);
            }
            for (o = 0; o < O; ++o) {
                float p = 1;
                int of[ORC_MAXF];
                features_of(o, d->Ostep, d->FO, of);
                for (f = 0; f < d->FO; ++f) p *= e[f][of[f]];
                c->mh_O[((size_t)a * S + s) * O + o] = p;
            }
        }
}
/* rnd::sample::Dir::sampleFromMult<double> (random.hpp:93-115) */
static int sample_from_mult_d(orc_ctx* c, const double* m, int n, double total)
{
    double p = orc_u01(&c->rng) * total, sum = m[0];
    int i;
    for (i = 1; i < n; ++i) {
        if (p < sum) return i - 1;
        sum += m[i];
    }
    return n - 1;
}
/* msgSampleStateHistory (:96-213): per episode a backward pass of messages p(o_t.. | s_t), each normalised, then a
 * forward pass sampling s_0, s_1, ... from T x message */
static void mh_sample_history_msg(orc_ctx* c, const float* model, int32_t* seq)
{
    int S = c->S, A = c->A, O = c->O, e, k = 0, h0 = 0, st, step, ns;
    float prior_p = (float)((double)(1.0f / (float)S) / (double)((1.0f / (float)S) * (float)S)); /* categoricalDistr(size, init)::prob */
    mh_flatten(c, model);
    for (e = 0; e < c->mh_n_ep; ++e) {
        int L = c->mh_ep_len[e];
        const int16_t *ea = c->mh_a + h0, *eo = c->mh_o + h0;
        double* msg = c->mh_msg;
        double tot;
        for (st = 0; st < S; ++st) msg[(size_t)L * S + st] = c->mh_O[((size_t)ea[L - 1] * S + st) * O + eo[L - 1]];
        for (step = L - 1; step >= 0; --step) {
            int a = ea[step];
            tot = 0;
            for (st = 0; st < S; ++st) {
                double m = 0.0;
                for (ns = 0; ns < S; ++ns) m = m + c->mh_T[((size_t)st * A + a) * S + ns] * msg[(size_t)(step + 1) * S + ns];
                if (step != 0) m *= c->mh_O[((size_t)ea[step - 1] * S + st) * O + eo[step - 1]];
                else m *= prior_p;
                msg[(size_t)step * S + st] = m;
                tot += m;
            }
            for (st = 0; st < S; ++st) msg[(size_t)step * S + st] = msg[(size_t)step * S + st] / tot;
        }
        st = sample_from_mult_d(c, msg, S, 1);
        seq[k++] = st;
        for (step = 0; step < L; ++step) {
            tot = 0;
            for (ns = 0; ns < S; ++ns) {
                c->mh_probs[ns] = c->mh_T[((size_t)st * A + ea[step]) * S + ns] * msg[(size_t)(step + 1) * S + ns];
                tot += c->mh_probs[ns];
            }
            st = sample_from_mult_d(c, c->mh_probs, S, tot);
            seq[k++] = st;
        }
        h0 += L;
    }
}
/* rejectionSampleStateHistory (:38-94): per episode, a start state and one sampled step per history step, starting the
 * episode over whenever a sampled observation differs from the recorded one.  Returns 0 if an episode cannot be
 * reproduced in 2^22 tries (the reference would never return). */
static int mh_sample_history_rs(orc_ctx* c, const float* model, int32_t* seq)
{
    int e, k = 0, h0 = 0, t;
    for (e = 0; e < c->mh_n_ep; ++e) {
        int L = c->mh_ep_len[e], tries = 0, ok = 0;
        while (!ok) {
            simstate st;
            if (++tries > (1 << 22)) return 0;
            st.s   = domain_start(c);
            st.cnt = (float*)model;
            seq[k] = st.s;
            ok = 1;
            for (t = 0; t < L; ++t) {
                int32_t so;
                double r;
                ba_fact_step(c, &st, c->mh_a[h0 + t], &so, &r, 0);   /* sampleStateIndex, sampleObservationIndex: expected method */
                if (so != c->mh_o[h0 + t]) { ok = 0; break; }
                seq[k + 1 + t] = st.s;
            }
        }
        k += L + 1;
        h0 += L;
    }
    return 1;
}
static int mh_sample_history(orc_ctx* c, const float* model, int32_t* seq)
{
    if (c->cfg.belief_option == 1) return mh_sample_history_rs(c, model, seq);
    mh_sample_history_msg(c, model, seq);
    return 1;
}
/* MHwithinGibbs::reinvigorate (:334-395) */
static int mh_reinvigorate(orc_ctx* c)
{
    int n = c->cfg.particles, made = 0, nseq = 0, e, iters = 0, nvar = c->fd.nvar;
    uint32_t masks[MH_MAXVAR], nmasks[MH_MAXVAR];
    double score, w1 = 1.0 / (double)n;
    for (e = 0; e < c->mh_n_ep; ++e) nseq += c->mh_ep_len[e] + 1;
    orc_rng_stream(&c->rng, ORC_PH_REINVIG, 0);
    memcpy(c->mh_model, c->P[weighted_sample(c)].cnt, sizeof(float) * (size_t)c->ncnt);  /* old_belief.sample()->model() */
    if (!mh_sample_history(c, c->mh_model, c->mh_seq)) return 0;
    mh_structure_of(c, c->mh_model, masks);
    mh_compute_prior(c, masks, c->mh_prior);
    mh_posterior(c, c->mh_prior, c->mh_seq, c->mh_model);
    score = orc_log_bd_score(c, c->mh_model, c->mh_prior);
    while (made < n) {
        double new_score;
        memcpy(nmasks, masks, sizeof(uint32_t) * (size_t)nvar);
        mh_mutate(c, nmasks);
        if (++iters > (1 << 24)) return 0;
        mh_compute_prior(c, nmasks, c->mh_prior);
        mh_posterior(c, c->mh_prior, c->mh_seq, c->mh_new);
        new_score = orc_log_bd_score(c, c->mh_new, c->mh_prior);
        if (m_log(c, orc_u01(&c->rng)) < (new_score - score)) {
            c->Pnew[made].s = c->mh_seq[nseq - 1];
            c->Pnew[made].w = w1;
            memcpy(c->Pnew[made].cnt, c->mh_new, sizeof(float) * (size_t)c->ncnt);
            made++;
            if (!mh_sample_history(c, c->mh_model, c->mh_seq)) return 0;   /* (from the model of the LAST accepted structure, as the reference does) */
            mh_posterior(c, c->mh_prior, c->mh_seq, c->mh_model);
            memcpy(masks, nmasks, sizeof(uint32_t) * (size_t)nvar);
            score = orc_log_bd_score(c, c->mh_model, c->mh_prior);
        }
    }
    swap_pools(c);
    c->total_w = 0;
    for (e = 0; e < n; ++e) c->total_w += w1;   /* WeightedFilter::add accumulates */
    weighted_refresh_scan(c);
    c->log_lik = 0;
    return 1;
}

/* ------------------------------------------------------------------ MH belief of the NIPS 2018 paper
 * ref: src/beliefs/bayes-adaptive/factored/MHNIPS2018.cpp.  The same filter, history and trigger; the re-draw (MH,
 * :208-255) makes independent proposals: a particle of the old filter, its structure or (half of the time) a mutation,
 * the prior of that structure updated along a freshly simulated history (computePosterior, :39-105), accepted on the
 * difference of LogBDScores.  Stream (REINVIG, 0), sequential. */

/* computePosterior (:39-105): per episode a start state and one sampled step per history step (expected Dirichlet
 * method), counts incremented as it goes; a wrong observation undoes the episode's increments and starts it over.
 * Returns the last state, or -1 if an episode cannot be reproduced in 2^22 tries (the reference would never return). */
static int32_t mhnips_posterior(orc_ctx* c, float* model)
{
    int e, h0 = 0, t;
    int32_t last = 0;
    for (e = 0; e < c->mh_n_ep; ++e) {
        int L = c->mh_ep_len[e], tries = 0, done = 0;
        while (!done) {
            simstate st;
            int32_t* from = c->mh_seq;       /* state_transitions: (s, s') per applied step */
            if (++tries > (1 << 22)) return -1;
            st.s   = domain_start(c);
            st.cnt = model;
            for (t = 0; t < L; ++t) {
                int32_t so, s = st.s;
                double r;
                ba_fact_step(c, &st, c->mh_a[h0 + t], &so, &r, 0);
                last = st.s;
                if (so != c->mh_o[h0 + t]) break;
                fact_increment(c, model, s, c->mh_a[h0 + t], so, st.s, 1);
                from[2 * t] = s; from[2 * t + 1] = st.s;
            }
            if (t == L) { done = 1; break; }
            { int u; for (u = 0; u < t; ++u) fact_increment(c, model, from[2 * u], c->mh_a[h0 + u], c->mh_o[h0 + u], from[2 * u + 1], -1); }
        }
        h0 += L;
    }
    return last;
}
/* MHNIPS2018::MH (:208-255) */
static int mhnips_redraw(orc_ctx* c)
{
    int n = c->cfg.particles, made = 0, e, iters = 0;
    uint32_t masks[MH_MAXVAR];
    double w1 = 1.0 / (double)n;
    orc_rng_stream(&c->rng, ORC_PH_REINVIG, 0);
    while (made < n) {
        const float* sampled = c->P[weighted_sample(c)].cnt;   /* old_belief.sample()->model() */
        double old_score, new_score;
        int32_t s_index;
        if (++iters > (1 << 24)) return 0;
        mh_structure_of(c, sampled, masks);
        mh_compute_prior(c, masks, c->mh_prior);               /* sampled_prior_model */
        if (!orc_bool(&c->rng)) mh_mutate(c, masks);           /* the same structure half of the time */
        mh_compute_prior(c, masks, c->mh_model);               /* new_prior_model */
        memcpy(c->mh_new, c->mh_model, sizeof(float) * (size_t)c->ncnt);
        s_index = mhnips_posterior(c, c->mh_new);
        if (s_index < 0) return 0;
        old_score = orc_log_bd_score(c, sampled, c->mh_prior);
        new_score = orc_log_bd_score(c, c->mh_new, c->mh_model);
        if (m_log(c, orc_u01(&c->rng)) < (new_score - old_score)) {
            c->Pnew[made].s = s_index;
            c->Pnew[made].w = w1;
            memcpy(c->Pnew[made].cnt, c->mh_new, sizeof(float) * (size_t)c->ncnt);
            made++;
        }
    }
    swap_pools(c);
    c->total_w = 0;
    for (e = 0; e < n; ++e) c->total_w += w1;
    weighted_refresh_scan(c);
    c->log_lik = 0;
    return 1;
}
/* MHwithinGibbs::updateEstimation (:316-332) */
static void mh_update(orc_ctx* c, int32_t a, int32_t o)
{
    int h = 0, e;
    double l = is_update(c, a, o);
    c->log_lik += m_log(c, l);
    is_resample(c);
    for (e = 0; e < c->mh_n_ep; ++e) h += c->mh_ep_len[e];
    c->mh_a[h] = (int16_t)a;
    c->mh_o[h] = (int16_t)o;
    c->mh_ep_len[c->mh_n_ep - 1]++;
    if (c->log_lik < c->cfg.threshold && !(c->cfg.belief == ORC_BELIEF_MH_NIPS ? mhnips_redraw(c) : mh_reinvigorate(c)))
        snprintf(c->err, sizeof c->err, "%s: the history cannot be reproduced by the sampled model", c->cfg.belief == ORC_BELIEF_MH_NIPS ? "mh-nips" : "mh-within-gibbs");
}

/* BAState::incrementCountsOf with an amount: BAFlatModel (BAFlatModel.cpp:130-139) or BABNModel (fact_increment) */
static void model_increment(orc_ctx* c, float* cnt, int32_t s, int32_t a, int32_t o, int32_t ns, float amount)
{
    if (c->cfg.model == ORC_MODEL_BA_FACTORED) { fact_increment(c, cnt, s, a, o, ns, amount); return; }
    cnt[s * c->A * c->S + a * c->S + ns] += amount;
    cnt[c->phi_len + a * c->S * c->O + ns * c->O + o] += amount;
}
/* NestedBelief::updateEstimation (NestedBelief.cpp:127-192): per count particle, rejection sampling of its domain-state
 * filter with KeepCounts steps on the particle's own counts, every accepted sample adding 1/n^2 to the counts the next
 * attempt samples from; weight *= 1 / attempts; normalise.  One stream per count particle (REJECT, i). */
static void nested_update(orc_ctx* c, int32_t a, int32_t o)
{
    int n = c->cfg.particles, M = c->nest_m, i;
    float step = (float)(1.0 / (float)M);
    double total = 0, accw = 0;
    long long all = 0;
    c->step_counter = &c->belief_steps;
    for (i = 0; i < n; ++i) {
        int32_t* filter = c->nest_s + (size_t)i * M;
        int acc = 0, count = 0;
        orc_rng_stream(&c->rng, ORC_PH_REJECT, (uint32_t)i);
        while (acc < M) {
            int32_t so, old = filter[orc_int(&c->rng, M)];
            double r;
            simstate st;
            st.s   = old;
            st.cnt = c->P[i].cnt;
            sim_step(c, &st, a, &so, &r, 0);
            if (so == o) {
                c->nest_new[acc++] = st.s;
                model_increment(c, c->P[i].cnt, old, a, o, st.s, step);
            }
            if (++count > (1 << 24)) {
                snprintf(c->err, sizeof c->err, "nested belief: count particle %d cannot produce the observation", i);
                return;
            }
        }
        memcpy(filter, c->nest_new, sizeof(int32_t) * (size_t)M);
        c->P[i].w *= 1.0 / (double)count;
        total += c->P[i].w;
        all += count;
    }
    if (c->cfg.arith == ORC_ARITH_DEV) {
        for (i = 0; i < n; ++i) c->wscratch[i] = c->P[i].w;
        total = dev_scan(c->wscratch, n, NULL);
    }
    c->last_weight_total = total;
    for (i = 0; i < n; ++i) {   /* WeightedFilter::normalize WeightedFilter.cpp:113-143 */
        c->P[i].w /= total;
        accw += c->P[i].w;
    }
    c->total_w = accw;
    weighted_refresh_scan(c);
    c->last_update_count = (int32_t)all;
}

static void belief_update(orc_ctx* c, int32_t a, int32_t o)
{
    c->last_weight_total = 0;
    if (is_nested(c)) { nested_update(c, a, o); return; }
    if (c->cfg.belief == ORC_BELIEF_INCUBATOR) { incubator_update(c, a, o); return; }
    if (c->cfg.belief == ORC_BELIEF_REJECTION) reject_sample(c, a, o, ORC_PH_REJECT);
    else if (c->cfg.belief == ORC_BELIEF_REINVIGORATION) {
        /* ReinvigoratingRejectionSampling::updateEstimation (ReinvigoratingRejectionSampling.cpp:89-107) */
        int count;
        reinvigorate(c);
        reject_sample(c, a, o, ORC_PH_REJECT);
        count = c->last_update_count;
        swap_main_fc(c);
        reject_sample(c, a, o, ORC_PH_REJECT_FC);
        swap_main_fc(c);
        c->last_update_count = count;
    } else if (c->cfg.belief == ORC_BELIEF_CHEATING) {
        /* CheatingReinvigoration::updateEstimation (CheatingReinvigoration.cpp:105-128) */
        double l;
        swap_main_fc(c);
        reject_sample(c, a, o, ORC_PH_REJECT_FC);
        swap_main_fc(c);
        l = is_update(c, a, o);
        is_resample(c);
        c->likelihood *= l;
        if (m_log(c, c->likelihood) < c->cfg.threshold) {
            /* cheat (:130-143): g++ evaluates replace's arguments right to left, so the correct filter is
             * sampled before slowRandomInt picks the victim; the victim keeps its weight */
            int n = c->cfg.particles, k;
            for (k = 0; k < c->cfg.resample_amount; ++k) {
                int32_t src, victim;
                orc_rng_stream(&c->rng, ORC_PH_REINVIG, (uint32_t)k);
                src    = orc_int(&c->rng, n);
                victim = orc_slow_int(&c->rng, 0, n);
                c->P[victim].s = c->F[src].s;
                memcpy(c->P[victim].cnt, c->F[src].cnt, sizeof(float) * (size_t)c->ncnt);
            }
            c->likelihood = 1;
        }
    } else if (is_mh(c)) {
        mh_update(c, a, o);
    } else {
        is_update(c, a, o);
        is_resample(c);
    }
}

/* BABelief::resetDomainStateDistribution.
 * ref: BARejectionSampling.cpp:49-60 (reset every particle in order);
 *      BAImportanceSampling.cpp:90-111 (resample n copies, reset each, weight 1/n) */
static void belief_reset_domain_state(orc_ctx* c)
{
    int i, n = c->cfg.particles;
    if (is_nested(c)) { /* NestedBelief::resetDomainStateDistribution (NestedBelief.cpp:35-58): every flat filter drawn anew */
        int j;
        for (i = 0; i < n; ++i) {
            orc_rng_stream(&c->rng, ORC_PH_RESET, (uint32_t)i);
            for (j = 0; j < c->nest_m; ++j) c->nest_s[(size_t)i * c->nest_m + j] = domain_start(c);
        }
        return;
    }
    if (c->cfg.belief != ORC_BELIEF_IMPORTANCE) { /* (the cheating belief resets its weighted filter in place too, CheatingReinvigoration.cpp:48-62) */
        for (i = 0; i < n; ++i) {
            orc_rng_stream(&c->rng, ORC_PH_RESET, (uint32_t)i);
            c->P[i].s = domain_start(c);
        }
        if (is_mh(c) && c->mh_ep_len[c->mh_n_ep - 1] != 0) /* MHwithinGibbs::resetDomainStateDistribution :259-275 (MHNIPS2018.cpp:114-130): a new episode unless the open one is empty */
            c->mh_ep_len[c->mh_n_ep++] = 0;
        if (is_breeding(c) || c->cfg.belief == ORC_BELIEF_CHEATING) /* ReinvigoratingRejectionSampling.cpp:109-119 */
            for (i = 0; i < n; ++i) {
                orc_rng_stream(&c->rng, ORC_PH_RESET_FC, (uint32_t)i);
                c->F[i].s = domain_start(c);
            }
        if (c->cfg.belief == ORC_BELIEF_INCUBATOR) /* StructureIncubatorSampling::resetDomainStateDistribution :46-61: in place, weights kept */
            for (i = 0; i < n; ++i) {
                orc_rng_stream(&c->rng, ORC_PH_RESET_SH, (uint32_t)i);
                c->Sh[i].s = domain_start(c);
            }
    } else {
        double w1 = 1.0 / (double)n, new_total = 0;
        for (i = 0; i < n; ++i) {
            int32_t src;
            orc_rng_stream(&c->rng, ORC_PH_RESET, (uint32_t)i);
            src = weighted_sample(c);
            if (c->ncnt) memcpy(c->Pnew[i].cnt, c->P[src].cnt, sizeof(float) * c->ncnt);
            c->Pnew[i].s = domain_start(c);
            c->Pnew[i].w = w1;
            new_total += w1;
        }
        swap_pools(c);
        c->total_w = new_total;
        weighted_refresh_scan(c);
    }
}

static uint64_t belief_hash(orc_ctx* c)
{
    uint64_t h = 0;
    int i, n = c->cfg.particles;
    int weighted = is_weighted(c);
    for (i = 0; i < n; ++i)
        h += particle_hash((uint64_t)i, c->P[i].s, weighted ? c->P[i].w : 0.0, c->P[i].cnt,
                           c->ncnt);
    if (is_nested(c)) { /* + every domain state of every flat filter, keyed by its position */
        size_t k, tot = (size_t)n * c->nest_m;
        for (k = 0; k < tot; ++k) h += mix64(((uint64_t)n + k) * 0x9E3779B97F4A7C15ull + (uint64_t)(uint32_t)c->nest_s[k]);
    }
    return h;
}

/* ------------------------------------------------------------------ planner */

/* ref: POUCT::selectAction POUCT.cpp:63-129; RBAPOUCT::selectAction RBAPOUCT.cpp:67-153
 * (root particle borrowed by pointer, counts read-only: StepType::KeepCounts) */
static int32_t select_action(orc_ctx* c, int hist_len, orc_trace_rec* rec)
{
    int i, n = c->cfg.sims, a;
    int32_t root, best, ts_src = -1;
    c->step_counter = &c->sim_steps;
    if (c->cfg.planner == ORC_PLANNER_RANDOM) {
        /* ref: RandomPlanner::selectAction src/planners/random/RandomPlanner.cpp:14-24 */
        orc_rng_stream(&c->rng, ORC_PH_SEARCH, (uint32_t)n);
        return domain_random_action(c, belief_sample_state(c, belief_sample(c)));
    }
    if (c->cfg.planner == ORC_PLANNER_TS) {
        /* TSPlanner::selectAction (src/planners/ts/TSPlanner.cpp:16-29) / BATSPlanner (BATSPlanner.cpp:19-34):
         * one belief.sample(), then PO-UCT on a point-estimate belief whose sample() draws nothing */
        orc_rng_stream(&c->rng, ORC_PH_SEARCH, (uint32_t)n + 2);
        ts_src = belief_sample(c);
        c->ts_state = belief_sample_state(c, ts_src);
    }
    tree_reset(c);
    /* simulator.addLegalActions(belief.sample(), ...) : one belief draw, result unused here
     * because every supported domain has state-independent legal actions */
    orc_rng_stream(&c->rng, ORC_PH_SEARCH, (uint32_t)n);
    if (ts_src < 0) (void)belief_sample_state(c, belief_sample(c));
    root = tree_new_node(c);
    {
        int d = c->cfg.horizon - hist_len;
        c->tr.max_tree_depth = d < c->cfg.max_depth ? d : c->cfg.max_depth;
    }
    for (i = 0; i < n; ++i) {
        simstate st;
        int32_t src;
        orc_rng_stream(&c->rng, ORC_PH_SEARCH, (uint32_t)i);
        src    = ts_src >= 0 ? ts_src : belief_sample(c);
        st.s   = ts_src >= 0 ? c->ts_state : belief_sample_state(c, src);
        st.cnt = c->P[src].cnt;
        traverse_action(c, root, &st, c->tr.max_tree_depth);
    }
    if (getenv("ORC_TREE_STATS")) { /* diagnostic: how many nodes were ever revisited, and how the visits spread (sizing of tree records) */
        int32_t k, v0 = 0, v1 = 0, v2_7 = 0, v8 = 0;
        long long sumv = 0;
        for (k = 1; k < c->tr.n_nodes; ++k) {
            int32_t v = c->tr.visits[k];
            sumv += v;
            if (v == 0) v0++; else if (v == 1) v1++; else if (v < 8) v2_7++; else v8++;
        }
        fprintf(stderr, "tree_stats t=%d nodes=%d never_revisited=%d once=%d 2to7=%d ge8=%d levels_below_root_per_sim=%.3f depth=%d\n", hist_len,
                c->tr.n_nodes, v0, v1, v2_7, v8, (double)sumv / n, c->tr.tree_depth);
    }
    orc_rng_stream(&c->rng, ORC_PH_SEARCH, (uint32_t)n + 1);
    best = select_chance_ucb(c, root, 0);
    if (rec) {
        rec->n_nodes    = c->tr.n_nodes;
        rec->tree_depth = c->tr.tree_depth;
        for (a = 0; a < c->A && a < ORC_MAX_ACTIONS; ++a) {
            rec->root_n[a] = c->tr.cn[root * c->A + a];
            rec->root_q[a] = c->tr.cq[root * c->A + a];
        }
    }
    return best;
}

/* ------------------------------------------------------------------ episode / experiments */

static orc_trace_rec* trace_push(orc_ctx* c)
{
    if (!c->cfg.trace) return NULL;
    if (c->n_trace == c->cap_trace) {
        c->cap_trace = c->cap_trace ? 2 * c->cap_trace : 1024;
        c->trace     = (orc_trace_rec*)realloc(c->trace, sizeof(orc_trace_rec) * c->cap_trace);
    }
    memset(&c->trace[c->n_trace], 0, sizeof(orc_trace_rec));
    return &c->trace[c->n_trace++];
}

/* ref: episode::run src/experiments/Episode.cpp:16-64 */
static double episode_run(orc_ctx* c, int run, int episode, int* length)
{
    double ret = 0, disc = 1, r = 0;
    int term = 0, t;
    int32_t s, o = 0;
    uint32_t R = (uint32_t)(run + c->cfg.run_offset);
    orc_rng_episode(&c->rng, R, (uint32_t)episode, 0);
    orc_rng_stream(&c->rng, ORC_PH_START, 0);
    s = domain_start(c);
    for (t = 0; t < c->cfg.horizon && !term; ++t) {
        orc_trace_rec* rec = trace_push(c);
        int32_t a;
        orc_rng_episode(&c->rng, R, (uint32_t)episode, (uint32_t)t);
        a = select_action(c, t, rec);
        orc_rng_stream(&c->rng, ORC_PH_ENV, 0);
        term = domain_step(c, &s, a, &o, &r);
        c->env_steps++;
        c->last_update_count = -1;
        c->last_weight_total = 0;
        if (!term) belief_update(c, a, o);
        if (rec) {
            rec->run = run + c->cfg.run_offset; rec->episode = episode; rec->t = t;
            rec->action = a; rec->state = s; rec->obs = o; rec->terminal = term;
            rec->reward = r;
            rec->update_count = term ? -1 : c->last_update_count;
            rec->weight_total = c->last_weight_total;
            rec->belief_hash  = belief_hash(c);
        }
        ret += r * disc;
        disc *= c->cfg.discount;
    }
    *length = t;
    return ret;
}

static double now_sec(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void fill_result(orc_ctx* c, orc_result* res, double t0)
{
    if (!res) return;
    res->sim_steps    = c->sim_steps;
    res->belief_steps = c->belief_steps;
    res->env_steps    = c->env_steps;
    res->seconds      = now_sec() - t0;
    res->n_trace      = c->n_trace;
}

/* ref: experiment::planning::run src/experiments/PlanningExperiment.cpp:27-55 */
int orc_run_planning(orc_ctx* c, orc_stat* stats, orc_result* res)
{
    int run, len;
    double t0 = now_sec();
    if (c->cfg.model != ORC_MODEL_POMDP) {
        snprintf(c->err, sizeof c->err, "planning::run needs model = POMDP");
        return -1;
    }
    memset(stats, 0, sizeof(*stats));
    for (run = 0; run < c->cfg.runs; ++run) {
        orc_rng_episode(&c->rng, (uint32_t)(run + c->cfg.run_offset), 0, 0);
        belief_initiate(c);
        orc_stat_add(stats, episode_run(c, run, 0, &len));
    }
    fill_result(c, res, t0);
    return 0;
}

/* ref: experiment::bapomdp::run src/experiments/BAPOMDPExperiment.cpp:32-78 */
int orc_run_bapomdp(orc_ctx* c, orc_stat* stats, orc_result* res)
{
    int run, ep, len;
    double t0 = now_sec();
    if (c->cfg.model == ORC_MODEL_POMDP) {
        snprintf(c->err, sizeof c->err, "bapomdp::run needs a Bayes-adaptive model");
        return -1;
    }
    memset(stats, 0, sizeof(*stats) * (size_t)c->cfg.episodes);
    for (run = 0; run < c->cfg.runs; ++run) {
        uint32_t R = (uint32_t)(run + c->cfg.run_offset);
        orc_rng_episode(&c->rng, R, 0, 0);
        belief_initiate(c);
        for (ep = 0; ep < c->cfg.episodes; ++ep) {
            orc_rng_episode(&c->rng, R, (uint32_t)ep, 0);
            belief_reset_domain_state(c);
            orc_stat_add(&stats[ep], episode_run(c, run, ep, &len));
        }
    }
    fill_result(c, res, t0);
    return 0;
}

/* ------------------------------------------------------------------ lifecycle */

orc_ctx* orc_create(const orc_config* cfg)
{
    orc_ctx* c = (orc_ctx*)calloc(1, sizeof(orc_ctx));
    int i, n;
    orc_config eff = *cfg;
    if (eff.belief == ORC_BELIEF_POINT) { /* PointEstimation.cpp / BAPointEstimation.cpp: one state, updated by rejection,
                                           * and sample() returns it without touching the RNG */
        eff.belief    = ORC_BELIEF_REJECTION;
        eff.particles = 1;
        c->point      = 1;
    }
    cfg    = &eff;
    c->cfg = *cfg;
    if (c->cfg.max_depth < 0) c->cfg.max_depth = c->cfg.horizon; /* ArgumentParser.cpp:37-40 */
    if (cfg->rng_mode == ORC_RNG_MT) {
        if (cfg->seed_str[0]) orc_rng_init_mt_str(&c->rng, cfg->seed_str, strlen(cfg->seed_str));
        else orc_rng_init_mt_u32(&c->rng, (uint32_t)time(NULL)); /* rnd::initiate() */
    } else {
        orc_rng_init_philox(&c->rng, cfg->philox_seed);
    }
    c->gamma = cfg->discount;
    zig_tables(c);
    switch (cfg->domain) {
        case ORC_DOM_TIGER_EPISODIC:
        case ORC_DOM_TIGER_CONTINUOUS: c->S = 2; c->A = 3; c->O = 2; break;
        case ORC_DOM_FTIGER_EPISODIC:
        case ORC_DOM_FTIGER_CONTINUOUS:
            if (cfg->size < 1) {
                snprintf(c->err, sizeof c->err, "cannot initiate FactoredTiger with %d irrelevant features", cfg->size);
                return c;
            }
            c->tiger_K = cfg->size; c->S = 2 << cfg->size; c->A = 3; c->O = 2;
            break;
        case ORC_DOM_COLLISION_AVOID: {
            int W = cfg->width, H = cfg->height, n = cfg->size, k, Hn = 1;
            if (W < 1) { snprintf(c->err, sizeof c->err, "Cannot initiate CollisionAvoidance with width %d", W); return c; }
            if (H < 1 || H % 2 != 1 || H > 15) { snprintf(c->err, sizeof c->err, "Cannot initiate CollisionAvoidance with height %d, must be uneven", H); return c; }
            if (n > W || n < 1 || n > 6) { snprintf(c->err, sizeof c->err, "cannot initiate collision avoidance with more obstacles (%d ) than columns (%d)!", n, W); return c; }
            for (k = 0; k < n; ++k) Hn *= H;
            ca_setup(c, W, H, n, cfg->structure_prior >= 0 && cfg->ca_centered == 0);
            c->S = W * H * Hn; c->A = 3; c->O = Hn;
            break;
        }
        case ORC_DOM_SYSADMIN_INDEPENDENT:
        case ORC_DOM_SYSADMIN_LINEAR:
            if (cfg->size < 1) { snprintf(c->err, sizeof c->err, "Cannot initiate Sysadmin with n %d", cfg->size); return c; }
            if (cfg->size > 8) { snprintf(c->err, sizeof c->err, "sysadmin: at most 8 computers (2N <= %d actions)", ORC_MAX_ACTIONS); return c; }
            sys_setup(c, cfg->size);
            c->S = 1 << cfg->size; c->A = 2 * cfg->size; c->O = 2;
            break;
        case ORC_DOM_AGR:
            if (cfg->model != ORC_MODEL_POMDP || cfg->belief != ORC_BELIEF_REJECTION) { /* AGR.cpp:307-310 throws on the first weighted update */
                snprintf(c->err, sizeof c->err, "AGR::computeObservationProbability nyi");
                return c;
            }
            c->S = (2 * AGR_N + 1) * (2 * AGR_N + 1); c->A = 2 * AGR_N + 3; c->O = 2 * AGR_N + 2;
            break;
        case ORC_DOM_COFFEE:
        case ORC_DOM_COFFEE_BOUTILIER: c->S = 32; c->A = 2; c->O = 2; break;
        case ORC_DOM_GRIDWORLD:
            if (cfg->size < 3 || cfg->size > 15) {
                snprintf(c->err, sizeof c->err, "please enter a size larger than 3 to be able to run gridworld (you entered %d)", cfg->size);
                return c;
            }
            gw_setup(c, cfg->size);
            c->S = c->O = cfg->size * cfg->size * c->gw_G; c->A = 4;
            break;
        default:
            snprintf(c->err, sizeof c->err, "domain %d not supported by the oracle", cfg->domain);
            return c;
    }
    if (cfg->sims < 1) { snprintf(c->err, sizeof c->err, "cannot initiate POUCT with %d simulations, must be greater than 0", cfg->sims); return c; }
    if (c->cfg.max_depth < 0) { snprintf(c->err, sizeof c->err, "max depth must be >= 0"); return c; }
    if (cfg->horizon <= 0) { snprintf(c->err, sizeof c->err, "cannot initiate POUCT with %d horizon, must be greater than 0", cfg->horizon); return c; }
    if (cfg->particles < 1) { snprintf(c->err, sizeof c->err, "cannot initiate belief with n = %d", cfg->particles); return c; }
    if (cfg->model == ORC_MODEL_BA_TABLE) {
        if (build_tabular_prior(c)) return c;
    } else if (cfg->model == ORC_MODEL_BA_FACTORED) {
        if (build_factored_prior(c)) return c;
    } else if (cfg->model != ORC_MODEL_POMDP) {
        snprintf(c->err, sizeof c->err, "model %d not supported by the oracle", cfg->model);
        return c;
    }
    c->log1p_tab = (double*)malloc(sizeof(double) * (size_t)(cfg->sims + 1));
    for (i = 0; i <= cfg->sims; ++i) c->log1p_tab[i] = log1p((double)i);
    tree_alloc(c);
    n        = cfg->particles;
    c->P     = (particle*)calloc((size_t)n, sizeof(particle));
    c->Pnew  = (particle*)calloc((size_t)n, sizeof(particle));
    c->wscratch = (double*)malloc(sizeof(double) * (size_t)n);
    c->wscan    = (double*)malloc(sizeof(double) * (size_t)n);
    if (c->ncnt) {
        c->pool     = (float*)malloc(sizeof(float) * (size_t)n * c->ncnt);
        c->pool_new = (float*)malloc(sizeof(float) * (size_t)n * c->ncnt);
        for (i = 0; i < n; ++i) {
            c->P[i].cnt    = c->pool + (size_t)i * c->ncnt;
            c->Pnew[i].cnt = c->pool_new + (size_t)i * c->ncnt;
        }
    }
    if (cfg->belief == ORC_BELIEF_CHEATING) {
        if (cfg->model != ORC_MODEL_BA_FACTORED) {
            snprintf(c->err, sizeof c->err, "cheating-reinvigoration belief: needs a factored model (fbapomdp)");
            return c;
        }
        if (cfg->resample_amount < 1) { /* CheatingReinvigoration.cpp:30-34 */
            snprintf(c->err, sizeof c->err, "CheatingReinvigoration::cannot initiate belief of size < 1 (%d), or resample size of < 1 (%d)", n, cfg->resample_amount);
            return c;
        }
        if (cfg->threshold >= 0) { /* :36-40 */
            snprintf(c->err, sizeof c->err, "CheatingReinvigoration::cannot initiate with resample_threshold >= 0 (is:%f)", cfg->threshold);
            return c;
        }
    }
    if (is_mh(c)) {
        int cap = cfg->episodes * cfg->horizon;
        const char* name = cfg->belief == ORC_BELIEF_MH_NIPS ? "MHNIPS2018" : "MHwithinGibbs";
        if (cfg->model != ORC_MODEL_BA_FACTORED || cfg->dirichlet_regular ||
            !(is_ftiger(cfg->domain) || is_ca(cfg->domain) || is_grid(cfg->domain) || is_sys(cfg->domain)) ||
            (is_ca(cfg->domain) && cfg->structure_prior == ORC_SP_FULLY_CONNECTED) || c->fd.nvar > MH_MAXVAR) {
            snprintf(c->err, sizeof c->err, "%s belief: needs a factored model (fbapomdp) in the expected Dirichlet mode, at most %d structure words per particle",
                     cfg->belief == ORC_BELIEF_MH_NIPS ? "mh-nips" : "mh-within-gibbs", MH_MAXVAR);
            return c;
        }
        if (cfg->belief == ORC_BELIEF_MH_NIPS && is_sys(cfg->domain)) {
            snprintf(c->err, sizeof c->err, "mh-nips belief on sysadmin: MHNIPS2018::MH scores particles against computePriorModel(structure), "
                     "not the prior they grew from (SysAdminFactoredPrior.cpp:98-127): no proposal is ever accepted");
            return c;
        }
        if (cfg->threshold >= 0) { /* MHwithinGibbs.cpp:248-252, MHNIPS2018.cpp:121-126 */
            snprintf(c->err, sizeof c->err, "%s::cannot initiate with threshold >= 0 (is:%f)", name, cfg->threshold);
            return c;
        }
        c->mh_a      = (int16_t*)calloc((size_t)cap + 1, sizeof(int16_t));
        c->mh_o      = (int16_t*)calloc((size_t)cap + 1, sizeof(int16_t));
        c->mh_ep_len = (int32_t*)calloc((size_t)cfg->episodes + 1, sizeof(int32_t));
        c->mh_prior  = (float*)malloc(sizeof(float) * (size_t)c->ncnt);
        c->mh_model  = (float*)malloc(sizeof(float) * (size_t)c->ncnt);
        c->mh_new    = (float*)malloc(sizeof(float) * (size_t)c->ncnt);
        c->mh_T      = (float*)malloc(sizeof(float) * (size_t)c->S * c->A * c->S);
        c->mh_O      = (float*)malloc(sizeof(float) * (size_t)c->A * c->S * c->O);
        c->mh_msg    = (double*)malloc(sizeof(double) * (size_t)(cfg->horizon + 1) * c->S);
        c->mh_probs  = (double*)malloc(sizeof(double) * (size_t)c->S);
        c->mh_seq    = (int32_t*)malloc(sizeof(int32_t) * (size_t)(cfg->episodes * (cfg->horizon + 1) + 2 * cfg->horizon + 1));
        c->mh_n_ep   = 1;
    }
    if (cfg->belief == ORC_BELIEF_NESTED) {
        if (cfg->model == ORC_MODEL_POMDP) {
            snprintf(c->err, sizeof c->err, "nested belief: needs a Bayes-adaptive model (bapomdp / fbapomdp)");
            return c;
        }
        if (n > 256) {
            snprintf(c->err, sizeof c->err, "nested belief: at most 256 count particles (each carries particles^2 domain states)");
            return c;
        }
        c->nest_m   = n * n;   /* BABelief.cpp:67-70: NestedBelief(particle_amount, particle_amount^2) */
        c->nest_s   = (int32_t*)calloc((size_t)n * c->nest_m, sizeof(int32_t));
        c->nest_new = (int32_t*)calloc((size_t)c->nest_m, sizeof(int32_t));
    }
    if (cfg->belief == ORC_BELIEF_INCUBATOR) {
        double w = 1.0 / (double)n, tot = 0;
        if (cfg->model != ORC_MODEL_BA_FACTORED || !(is_ftiger(cfg->domain) || is_ca(cfg->domain) || is_sys(cfg->domain)) ||
            (is_ca(cfg->domain) && cfg->structure_prior == ORC_SP_FULLY_CONNECTED)) {
            snprintf(c->err, sizeof c->err, "incubator belief: needs a factored model (fbapomdp) of factored tiger, collision avoidance or sysadmin");
            return c;
        }
        if (cfg->resample_amount < 1) { /* StructureIncubatorSampling.cpp:28-33 */
            snprintf(c->err, sizeof c->err, "StructureIncubatorSampling::Cannot initiate Incubator belief update with size < 1 (%d) or resample size < 1 (%d)", n, cfg->resample_amount);
            return c;
        }
        if (cfg->threshold <= 0 || cfg->threshold > 1) { /* :35-39 */
            snprintf(c->err, sizeof c->err, "StructureIncubatorSampling::must initiate with 1 < threshold <= 0 (is:%f)", cfg->threshold);
            return c;
        }
        if (cfg->resample_amount >= n || cfg->resample_amount > 1024) { /* WeightedFilter::leastLikely asserts n < size() (WeightedFilter.cpp:207) */
            snprintf(c->err, sizeof c->err, "incubator belief: the resample amount (%d) must be below the number of particles (%d): WeightedFilter::leastLikely", cfg->resample_amount, n);
            return c;
        }
        if (cfg->arith == ORC_ARITH_DEV) {
            double* tmp = (double*)malloc(sizeof(double) * (size_t)n);
            for (i = 0; i < n; ++i) tmp[i] = w;
            tot = dev_scan(tmp, n, NULL);
            free(tmp);
        } else
            for (i = 0; i < n; ++i) tot += w;
        if (w / tot > cfg->threshold) {
            snprintf(c->err, sizeof c->err, "incubator belief: with threshold %g every one of the %d shadow particles (normalised weight %g) is promoted at "
                     "once; the shadow filter's total weight is then zero and StructureIncubatorSampling.cpp:160-188 divides by it", cfg->threshold, n, w / tot);
            return c;
        }
        c->Sh     = (particle*)calloc((size_t)n, sizeof(particle));
        c->Shnew  = (particle*)calloc((size_t)n, sizeof(particle));
        c->shpool     = (float*)malloc(sizeof(float) * (size_t)n * c->ncnt);
        c->shpool_new = (float*)malloc(sizeof(float) * (size_t)n * c->ncnt);
        for (i = 0; i < n; ++i) {
            c->Sh[i].cnt    = c->shpool + (size_t)i * c->ncnt;
            c->Shnew[i].cnt = c->shpool_new + (size_t)i * c->ncnt;
        }
    }
    if (is_breeding(c) || cfg->belief == ORC_BELIEF_CHEATING) {
        if (cfg->belief == ORC_BELIEF_CHEATING || cfg->belief == ORC_BELIEF_INCUBATOR) { /* checked above */
        } else if (cfg->model != ORC_MODEL_BA_FACTORED || !(is_ftiger(cfg->domain) || is_ca(cfg->domain) || is_sys(cfg->domain)) ||
            (is_ca(cfg->domain) && cfg->structure_prior == ORC_SP_FULLY_CONNECTED)) {
            /* the reference has fully connected priors for factored tiger, collision avoidance and
             * sysadmin; GridWorldFactBAPrior::sampleFullyConnectedState throws "nyi" */
            snprintf(c->err, sizeof c->err, "reinvigoration belief: needs a factored model (fbapomdp) of factored tiger, collision avoidance or sysadmin");
            return c;
        }
        if (cfg->resample_amount < 1) { /* ReinvigoratingRejectionSampling.cpp:43-49 */
            snprintf(c->err, sizeof c->err, "ReinvigoratingRejectionSampling::cannot initiate belief of size < 1 (%d), or resample size of < 1 (%d)", n, cfg->resample_amount);
            return c;
        }
        c->F     = (particle*)calloc((size_t)n, sizeof(particle));
        c->Fnew  = (particle*)calloc((size_t)n, sizeof(particle));
        c->fpool     = (float*)malloc(sizeof(float) * (size_t)n * c->ncnt);
        c->fpool_new = (float*)malloc(sizeof(float) * (size_t)n * c->ncnt);
        c->breed_tmp = (float*)malloc(sizeof(float) * (size_t)c->ncnt);
        for (i = 0; i < n; ++i) {
            c->F[i].cnt    = c->fpool + (size_t)i * c->ncnt;
            c->Fnew[i].cnt = c->fpool_new + (size_t)i * c->ncnt;
        }
    }
    c->step_counter = &c->sim_steps;
    return c;
}

void orc_destroy(orc_ctx* c)
{
    if (!c) return;
    free(c->prior); free(c->log1p_tab); free(c->fd.T); free(c->fd.O);
    free(c->tr.visits); free(c->tr.cn); free(c->tr.cq); free(c->tr.child);
    free(c->mh_a); free(c->mh_o); free(c->mh_ep_len); free(c->mh_prior); free(c->mh_model); free(c->mh_new); free(c->mh_T); free(c->mh_O);
    free(c->mh_msg); free(c->mh_probs); free(c->mh_seq);
    free(c->nest_s); free(c->nest_new);
    free(c->Sh); free(c->Shnew); free(c->shpool); free(c->shpool_new);
    free(c->tr.hkey); free(c->tr.hval);
    free(c->P); free(c->Pnew); free(c->pool); free(c->pool_new);
    free(c->F); free(c->Fnew); free(c->fpool); free(c->fpool_new); free(c->breed_tmp);
    free(c->wscratch); free(c->wscan); free(c->trace);
    free(c);
}

const char* orc_error(const orc_ctx* c) { return c->err[0] ? c->err : NULL; }
const orc_trace_rec* orc_trace(const orc_ctx* c) { return c->trace; }
orc_rng* orc_ctx_rng(orc_ctx* c) { return &c->rng; }

int orc_domain_sizes(const orc_ctx* c, int32_t* S, int32_t* A, int32_t* O)
{
    *S = c->S; *A = c->A; *O = c->O;
    return 0;
}
int orc_counts_len(const orc_ctx* c) { return c->ncnt; }
int orc_prior_counts(orc_ctx* c, float* out)
{
    if (!c->prior) return -1;
    if (c->cfg.model == ORC_MODEL_BA_FACTORED) factored_prior_sample(c, out);
    else memcpy(out, c->prior, sizeof(float) * (size_t)c->ncnt);
    return 0;
}

int orc_env_step(orc_ctx* c, int32_t* s, int32_t a, int32_t* o, double* r)
{
    return domain_step(c, s, a, o, r);
}
int orc_env_start(orc_ctx* c) { return domain_start(c); }
int orc_random_action(orc_ctx* c, int32_t s) { return domain_random_action(c, s); }

/* ---- filter-level entry points for golden-vector tests ---- */
void orc_belief_initiate(orc_ctx* c) { belief_initiate(c); }
void orc_belief_update(orc_ctx* c, int32_t a, int32_t o) { belief_update(c, a, o); }
double orc_is_update(orc_ctx* c, int32_t a, int32_t o) { return is_update(c, a, o); }
void orc_is_resample(orc_ctx* c) { is_resample(c); }
void orc_belief_reset_domain_state(orc_ctx* c) { belief_reset_domain_state(c); }
int orc_select_action(orc_ctx* c, int hist_len, orc_trace_rec* rec) { return select_action(c, hist_len, rec); }
uint64_t orc_belief_hash(orc_ctx* c) { return belief_hash(c); }
int orc_last_update_count(const orc_ctx* c) { return c->last_update_count; }
void orc_belief_get(const orc_ctx* c, int32_t* s, double* w, float* cnt)
{
    int i;
    for (i = 0; i < c->cfg.particles; ++i) {
        if (s) s[i] = c->P[i].s;
        if (w) w[i] = c->P[i].w;
        if (cnt && c->ncnt) memcpy(cnt + (size_t)i * c->ncnt, c->P[i].cnt, sizeof(float) * c->ncnt);
    }
}
/* the fully connected filter of the reinvigoration belief */
void orc_belief_get_shadow(const orc_ctx* c, int32_t* s, double* w, float* cnt)
{
    int i;
    if (!c->Sh) return;
    for (i = 0; i < c->cfg.particles; ++i) {
        if (s) s[i] = c->Sh[i].s;
        if (w) w[i] = c->Sh[i].w;
        if (cnt) memcpy(cnt + (size_t)i * c->ncnt, c->Sh[i].cnt, sizeof(float) * (size_t)c->ncnt);
    }
}
void orc_belief_get_nested(const orc_ctx* c, int32_t* states)
{
    if (c->nest_s) memcpy(states, c->nest_s, sizeof(int32_t) * (size_t)c->cfg.particles * c->nest_m);
}
void orc_belief_get_fc(const orc_ctx* c, int32_t* s, float* cnt)
{
    int i;
    if (!c->F) return;
    for (i = 0; i < c->cfg.particles; ++i) {
        if (s) s[i] = c->F[i].s;
        if (cnt) memcpy(cnt + (size_t)i * c->ncnt, c->F[i].cnt, sizeof(float) * c->ncnt);
    }
}
/* BABNModel::marginalizeOut of a caller-owned count blob onto new parent masks (one per variable node) */
int orc_marginalize(orc_ctx* c, const float* cnt, const uint32_t* new_masks, float* out)
{
    if (c->cfg.model != ORC_MODEL_BA_FACTORED) return -1;
    breed_counts(c, cnt, new_masks, out);
    return 0;
}
void orc_belief_set(orc_ctx* c, const int32_t* s, const double* w, const float* cnt)
{
    int i;
    double tot = 0;
    for (i = 0; i < c->cfg.particles; ++i) {
        if (s) c->P[i].s = s[i];
        if (w) c->P[i].w = w[i];
        tot += c->P[i].w;
        if (cnt && c->ncnt) memcpy(c->P[i].cnt, cnt + (size_t)i * c->ncnt, sizeof(float) * c->ncnt);
    }
    c->total_w = tot;
    weighted_refresh_scan(c);
}
/* simulator.step / computeObservationProbability on a caller-owned count blob */
int orc_model_step(orc_ctx* c, float* cnt, int32_t* s, int32_t a, int32_t* o, double* r, int update)
{
    simstate st;
    int t;
    st.s = *s; st.cnt = cnt;
    t = sim_step(c, &st, a, o, r, update);
    *s = st.s;
    return t;
}
double orc_model_obs_prob(orc_ctx* c, const float* cnt, int32_t new_s, int32_t a, int32_t o)
{
    simstate st;
    st.s = new_s; st.cnt = (float*)cnt;
    return sim_obs_prob(c, &st, a, o);
}
double orc_dev_scan(const double* w, int n, double* incl) { return dev_scan(w, n, incl); }

/* test hook: write the factored-tiger listen observation model for parent set `mask` into a blob */
int orc_ftiger_set_structure(orc_ctx* c, float* cnt, uint32_t mask)
{
    if (c->cfg.model != ORC_MODEL_BA_FACTORED || !is_ftiger(c->cfg.domain)) return -1;
    ftiger_set_observation_model(c, cnt, mask);
    return 0;
}
