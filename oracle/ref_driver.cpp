/*
 * oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Drives the REAL reference (the subset of /root/reference/src that compiles with g++ alone:
 * no Boost, no stand-ins) and prints golden vectors as JSON.  Built by `make -C oracle ref`
 * into oracle/_ref/ref_driver, only where /root/reference exists; its output is committed as
 * tests/golden/ref_vectors.json by oracle/gen_golden.py.  Nothing here is reference source:
 * it only #includes the reference's headers where they lie and calls its functions.
 *
 * Not buildable here (they include <boost/program_options.hpp> through configurations/Conf.hpp):
 * planners/mcts/POUCT.cpp, planners/bayes-adaptive/RBAPOUCT.cpp,
 * bayes-adaptive/models/table/BAPOMDP.cpp, every *Priors.cpp and every factory.  Every other
 * translation unit on the hot path is driven here: the tree nodes (MCTSTreeNodes.cpp), the BA / FBA
 * domain extensions, the BAState wrappers.
 */
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "easylogging++.h"

#include "bayes-adaptive/models/Domain_Size.hpp"
#include "bayes-adaptive/models/factored/Domain_Feature_Size.hpp"
#include "bayes-adaptive/states/factored/BABNModel.hpp"
#include "bayes-adaptive/states/factored/FBAPOMDPState.hpp"
#include "bayes-adaptive/states/table/BAPOMDPState.hpp"
#include "domains/collision-avoidance/CollisionAvoidanceBAExtension.hpp"
#include "domains/collision-avoidance/CollisionAvoidanceFBAExtension.hpp"
#include "domains/gridworld/GridWorldBAExtension.hpp"
#include "domains/gridworld/GridWorldFBAExtension.hpp"
#include "domains/sysadmin/SysAdminFBAExtension.hpp"
#include "domains/tiger/FactoredTigerBAExtension.hpp"
#include "domains/tiger/FactoredTigerFBAExtension.hpp"
#include "domains/tiger/TigerBAExtension.hpp"
#include "planners/mcts/MCTSTreeNodes.hpp"
#include "bayes-adaptive/states/table/BAFlatModel.hpp"
#include "beliefs/particle_filters/FlatFilter.hpp"
#include "beliefs/particle_filters/ImportanceSampler.hpp"
#include "beliefs/particle_filters/RejectionSampling.hpp"
#include "beliefs/particle_filters/WeightedFilter.hpp"
#include "beliefs/point_estimation/PointEstimation.hpp"
#include "domains/collision-avoidance/CollisionAvoidance.hpp"
#include "domains/agr/AGR.hpp"
#include "domains/coffee/CoffeeProblem.hpp"
#include "domains/gridworld/GridWorld.hpp"
#include "domains/sysadmin/SysAdmin.hpp"
#include "domains/sysadmin/SysAdminBAExtension.hpp"
#include "domains/tiger/FactoredTiger.hpp"
#include "domains/tiger/Tiger.hpp"
#include "environment/Action.hpp"
#include "environment/Discount.hpp"
#include "environment/Horizon.hpp"
#include "environment/Observation.hpp"
#include "environment/Reward.hpp"
#include "environment/State.hpp"
#include "experiments/Episode.hpp"
#include "planners/random/RandomPlanner.hpp"
#include "utils/Statistic.hpp"
#include "utils/index.hpp"
#include "utils/random.hpp"

INITIALIZE_EASYLOGGINGPP

static bool first_key = true;
static void key(char const* k)
{
    printf("%s\n\"%s\": ", first_key ? "" : ",", k);
    first_key = false;
}
static void seed(char const* s)
{
    std::string str(s);
    rnd::seed(str);
}
template<typename T, typename F>
static void arr(std::vector<T> const& v, F fmt)
{
    printf("[");
    for (size_t i = 0; i < v.size(); ++i) {
        if (i) printf(",");
        fmt(v[i]);
    }
    printf("]");
}
static void pi(int x) { printf("%d", x); }
static void pd(double x) { printf("%.17g", x); }

static void rng_vectors(char const* s)
{
    std::vector<double> u;
    std::vector<int> b, si, i3, i256, i4096, i100;
    seed(s);
    for (int i = 0; i < 64; ++i) u.push_back(rnd::uniform_rand01());
    for (int i = 0; i < 64; ++i) b.push_back(rnd::boolean());
    for (int i = 0; i < 64; ++i) si.push_back(rnd::slowRandomInt(0, 5));
    auto d3 = rnd::integerDistribution(0, 3);
    for (int i = 0; i < 64; ++i) i3.push_back(d3(rnd::rng()));
    auto d256 = rnd::integerDistribution(0, 256);
    for (int i = 0; i < 64; ++i) i256.push_back(d256(rnd::rng()));
    auto d100 = rnd::integerDistribution(0, 100);
    for (int i = 0; i < 64; ++i) i100.push_back(d100(rnd::rng()));
    auto d4096 = rnd::integerDistribution(0, 4096);
    for (int i = 0; i < 64; ++i) i4096.push_back(d4096(rnd::rng()));
    printf("{\"seed\": \"%s\", \"u01\": ", s);
    arr(u, pd);
    printf(", \"bool\": ");
    arr(b, pi);
    printf(", \"slow_int_0_5\": ");
    arr(si, pi);
    printf(", \"int3\": ");
    arr(i3, pi);
    printf(", \"int256\": ");
    arr(i256, pi);
    printf(", \"int100\": ");
    arr(i100, pi);
    printf(", \"int4096\": ");
    arr(i4096, pi);
    printf("}");
}

/* random walk through a POMDP: start state, then (random action, step) n times;
 * after a terminal step a new start state is drawn. */
static void walk(POMDP const& d, char const* s, int n)
{
    std::vector<int> rec; /* a, s', o, r, term, flattened */
    std::vector<double> rew; /* the rewards again, not truncated */
    seed(s);
    State const* st = d.sampleStartState();
    int start       = st->index();
    for (int i = 0; i < n; ++i) {
        Observation const* o = nullptr;
        Reward r(0);
        auto a = d.generateRandomAction(st);
        auto t = d.step(&st, a, &o, &r);
        rec.push_back(a->index());
        rec.push_back(st->index());
        rec.push_back(o->index());
        rec.push_back((int)r.toDouble());
        rew.push_back(r.toDouble());
        rec.push_back(t.terminated() ? 1 : 0);
        d.releaseAction(a);
        d.releaseObservation(o);
        if (t.terminated()) {
            d.releaseState(st);
            st = d.sampleStartState();
            rec.push_back(st->index());
        }
    }
    printf("{\"seed\": \"%s\", \"start\": %d, \"rec\": ", s, start);
    arr(rec, pi);
    printf(", \"rew\": ");
    arr(rew, pd);
    printf("}");
}

/* P(o | a, s') table */
static void obs_table(POMDP const& d, int S, int A, int O)
{
    std::vector<double> p;
    for (int a = 0; a < A; ++a)
        for (int s = 0; s < S; ++s)
            for (int o = 0; o < O; ++o) {
                IndexAction ia(a);
                IndexState is(s);
                IndexObservation io(o);
                p.push_back(d.computeObservationProbability(&io, &ia, &is));
            }
    arr(p, pd);
}

/* SysAdmin static_casts its states: P(o | a, s') through SysAdminBAExtension's state objects, plus
 * the extension's reward (SysAdminBAExtension.cpp:39-50) for every (a, s') */
static void sysadmin_tables(int n, char const* version)
{
    domains::SysAdmin d(n, version);
    bayes_adaptive::domain_extensions::SysAdminBAExtension ext(n);
    std::vector<double> p, rw;
    std::vector<int> nfn;
    for (int a = 0; a < 2 * n; ++a)
        for (int s = 0; s < (1 << n); ++s) {
            IndexAction ia(a);
            for (int o = 0; o < 2; ++o) {
                IndexObservation io(o);
                p.push_back(d.computeObservationProbability(&io, &ia, ext.getState(s)));
            }
            rw.push_back(ext.reward(ext.getState(0), &ia, ext.getState(s)).toDouble());
        }
    for (int c = 0; c < n; ++c)
        for (int s = 0; s < (1 << n); ++s) nfn.push_back((int)d.numFailingNeighbours(c, ext.getState(s)));
    printf("{\"obs_prob\": ");
    arr(p, pd);
    printf(", \"ext_reward\": ");
    arr(rw, pd);
    printf(", \"failing_neighbours\": ");
    arr(nfn, pi);
    printf("}");
}

/* GridWorld needs its own state / observation objects (it static_casts them) */
static void gridworld_obs_table(int size)
{
    domains::GridWorld d(size);
    auto goals = domains::GridWorld::goalLocations(size);
    std::vector<double> p;
    IndexAction a(0);
    for (unsigned x = 0; x < (unsigned)size; ++x)
        for (unsigned y = 0; y < (unsigned)size; ++y)
            for (auto const& g : goals)
                for (unsigned ox = 0; ox < (unsigned)size; ++ox)
                    for (unsigned oy = 0; oy < (unsigned)size; ++oy)
                        for (auto const& og : goals)
                            p.push_back(d.computeObservationProbability(d.getObservation({ox, oy}, og), &a, d.getState({x, y}, g)));
    arr(p, pd);
}

static void ca_obs_table(domains::CollisionAvoidance const& d, int W, int H, int n)
{
    std::vector<double> p;
    IndexAction a(1);
    int O = 1;
    for (int k = 0; k < n; ++k) O *= H;
    std::vector<int> range(n, H);
    for (int x = 0; x < W; ++x)
        for (int y = 0; y < H; ++y) {
            std::vector<int> obst(n, 0);
            do {
                for (int o = 0; o < O; ++o)
                    p.push_back(d.computeObservationProbability(d.getObservation(o), &a, d.getState(x, y, obst)));
            } while (!indexing::increment(obst, range));
        }
    arr(p, pd);
}

static std::vector<int> indices(FlatFilter<State const*> const& f)
{
    std::vector<int> v;
    for (auto p : f.particles()) v.push_back(p->index());
    return v;
}

/* beliefs::rejectSample on the real Tiger simulator */
static void reject_tiger(char const* s, int n)
{
    domains::Tiger d(domains::Tiger::CONTINUOUS);
    seed(s);
    FlatFilter<State const*> f((size_t)n, [&d] { return d.sampleStartState(); });
    int const acts[] = {2, 2, 0, 2, 1, 2, 2, 2};
    int const obs[]  = {0, 1, 1, 0, 0, 0, 0, 1};
    printf("{\"seed\": \"%s\", \"n\": %d, \"init\": ", s, n);
    arr(indices(f), pi);
    printf(", \"a\": [2,2,0,2,1,2,2,2], \"o\": [0,1,1,0,0,0,0,1], \"after\": [");
    for (int k = 0; k < 8; ++k) {
        IndexAction a(acts[k]);
        IndexObservation o(obs[k]);
        beliefs::rejectSample<State const*>(&a, &o, d, (size_t)n, f);
        if (k) printf(",");
        arr(indices(f), pi);
    }
    /* one draw afterwards pins the number of engine words consumed */
    printf("], \"next_u01\": %.17g}", rnd::uniform_rand01());
}

/* importance_sampling::update / resample on the real Tiger simulator */
static void is_tiger(char const* s, int n)
{
    domains::Tiger d(domains::Tiger::CONTINUOUS);
    seed(s);
    WeightedFilter<State const*> f;
    for (int i = 0; i < n; ++i) f.add(d.sampleStartState(), 1.0 / (double)n);
    int const acts[] = {2, 2, 1, 2, 2, 0};
    int const obs[]  = {1, 1, 0, 0, 1, 1};
    printf("{\"seed\": \"%s\", \"n\": %d, \"a\": [2,2,1,2,2,0], \"o\": [1,1,0,0,1,1], \"steps\": [", s, n);
    for (int k = 0; k < 6; ++k) {
        IndexAction a(acts[k]);
        IndexObservation o(obs[k]);
        double tot = beliefs::importance_sampling::update(f, &a, &o, d);
        std::vector<int> idx;
        std::vector<double> w;
        for (size_t i = 0; i < f.size(); ++i) {
            idx.push_back(f.particle(i)->particle->index());
            w.push_back(f.particle(i)->w);
        }
        if (k) printf(",");
        printf("{\"total\": %.17g, \"idx\": ", tot);
        arr(idx, pi);
        printf(", \"w\": ");
        arr(w, pd);
        beliefs::importance_sampling::resample(f, d, (size_t)n);
        idx.clear();
        for (size_t i = 0; i < f.size(); ++i) idx.push_back(f.particle(i)->particle->index());
        printf(", \"resampled\": ");
        arr(idx, pi);
        printf("}");
    }
    printf("], \"next_u01\": %.17g}", rnd::uniform_rand01());
}


/* importance_sampling::update / resample with the real GridWorld as simulator and as environment:
 * pins GridWorld::step + computeObservationProbability inside the filter */
static void is_gridworld(char const* s, int size, int n, int steps)
{
    domains::GridWorld d(size);
    seed(s);
    WeightedFilter<State const*> f;
    for (int i = 0; i < n; ++i) f.add(d.sampleStartState(), 1.0 / (double)n);
    State const* st = d.sampleStartState();
    printf("{\"seed\": \"%s\", \"size\": %d, \"n\": %d, \"start\": %d, \"steps\": [", s, size, n, st->index());
    for (int k = 0; k < steps; ++k) {
        Observation const* o = nullptr;
        Reward r(0);
        auto a = d.generateRandomAction(st);
        d.step(&st, a, &o, &r);
        double tot = beliefs::importance_sampling::update(f, a, o, d);
        std::vector<int> idx;
        std::vector<double> w;
        for (size_t i = 0; i < f.size(); ++i) {
            idx.push_back(f.particle(i)->particle->index());
            w.push_back(f.particle(i)->w);
        }
        if (k) printf(",");
        printf("{\"a\": %d, \"s\": %d, \"o\": %d, \"total\": %.17g, \"idx\": ", a->index(), st->index(), o->index(), tot);
        arr(idx, pi);
        printf(", \"w\": ");
        arr(w, pd);
        beliefs::importance_sampling::resample(f, d, (size_t)n);
        idx.clear();
        for (size_t i = 0; i < f.size(); ++i) idx.push_back(f.particle(i)->particle->index());
        printf(", \"resampled\": ");
        arr(idx, pi);
        printf("}");
        d.releaseAction(a);
    }
    printf("], \"next_u01\": %.17g}", rnd::uniform_rand01());
}

/* WeightedFilter::leastLikely / replace / normalizedWeight / normalize (WeightedFilter.cpp:70-87, 113-143, 193-238): what the
 * incubator belief asks of its shadow filter.  Cases: uniform weights (the only ones that belief ever has when it
 * asks), weights with ties, distinct weights. */
static void weighted_filter_ops()
{
    domains::Tiger d(domains::Tiger::CONTINUOUS);
    seed("77");
    struct Case { int size, n, kind; };
    Case const cases[] = {{16, 5, 0}, {20, 19, 0}, {33, 1, 0}, {48, 6, 0}, {7, 3, 0}, {24, 20, 0}, {130, 20, 0},
                          {12, 4, 1}, {12, 11, 1}, {40, 9, 1}, {10, 3, 2}, {64, 17, 2}};
    printf("[");
    for (size_t ci = 0; ci < sizeof cases / sizeof cases[0]; ++ci) {
        Case const c = cases[ci];
        WeightedFilter<State const*> f;
        std::vector<double> w;
        for (int i = 0; i < c.size; ++i) {
            double wi = 1.0 / (double)c.size;
            if (c.kind == 1) wi = (double)(1 + (i * 7) % 3) / 8.0;        /* three distinct values: ties */
            if (c.kind == 2) wi = rnd::uniform_rand01();
            w.push_back(wi);
            f.add(d.sampleStartState(), wi);
        }
        auto ll = f.leastLikely((size_t)c.n);
        if (ci) printf(",");
        printf("{\"w\": ");
        arr(w, pd);
        printf(", \"n\": %d, \"least_likely\": ", c.n);
        arr(ll, pi);
        /* replace the first of them as the incubator does, then the weights and the normalised weight of particle 0 */
        f.replace(ll[0], d.sampleStartState(), [&d](State const* st) { d.releaseState(st); });
        std::vector<double> w2;
        for (size_t i = 0; i < f.size(); ++i) w2.push_back(f.particle(i)->w);
        printf(", \"after_replace\": ");
        arr(w2, pd);
        printf(", \"normalized_w0\": %.17g", f.normalizedWeight(f.particle(0)->w));
        f.normalize();
        w2.clear();
        for (size_t i = 0; i < f.size(); ++i) w2.push_back(f.particle(i)->w);
        printf(", \"after_normalize\": ");
        arr(w2, pd);
        printf(", \"normalized_w0_again\": %.17g}", f.normalizedWeight(f.particle(0)->w));
        f.free([&d](State const* st) { d.releaseState(st); });
    }
    printf("]");
}

/* BAFlatModel sampling / probability / increment with the expected-Dirichlet method */
static void flat_model(char const* s)
{
    Domain_Size sz(2, 3, 2);
    bayes_adaptive::table::BAFlatModel m(&sz);
    /* tiger-prior-like counts, deliberately asymmetric */
    for (int st = 0; st < 2; ++st)
        for (int a = 0; a < 3; ++a)
            for (int ns = 0; ns < 2; ++ns) {
                IndexState x(st), y(ns);
                IndexAction ia(a);
                m.count(&x, &ia, &y) = (a == 2) ? ((st == ns) ? 5000.f : 0.f) : 5000.f + 7.f * (float)(st + 2 * ns + a);
            }
    for (int a = 0; a < 3; ++a)
        for (int ns = 0; ns < 2; ++ns)
            for (int o = 0; o < 2; ++o) {
                IndexState y(ns);
                IndexAction ia(a);
                IndexObservation io(o);
                m.count(&ia, &y, &io) = (a == 2) ? ((ns == o) ? 8500.f : 1500.f) : 5000.f + 3.f * (float)(o + a);
            }
    seed(s);
    std::vector<int> rec;
    std::vector<double> probs;
    int st = 0;
    for (int i = 0; i < 300; ++i) {
        int a = i % 3;
        IndexState x(st);
        IndexAction ia(a);
        int ns = m.sampleStateIndex(&x, &ia, rnd::sample::Dir::sampleFromExpectedMult);
        IndexState y(ns);
        int o = m.sampleObservationIndex(&ia, &y, rnd::sample::Dir::sampleFromExpectedMult);
        IndexObservation io(o);
        probs.push_back(m.computeObservationProbability(&io, &ia, &y, rnd::sample::Dir::expectedMult));
        m.incrementCountsOf(&x, &ia, &io, &y);
        rec.push_back(a);
        rec.push_back(ns);
        rec.push_back(o);
        st = ns;
    }
    std::vector<double> phi, psi;
    for (int s0 = 0; s0 < 2; ++s0)
        for (int a = 0; a < 3; ++a)
            for (int ns = 0; ns < 2; ++ns) {
                IndexState x(s0), y(ns);
                IndexAction ia(a);
                phi.push_back(m.count(&x, &ia, &y));
            }
    for (int a = 0; a < 3; ++a)
        for (int ns = 0; ns < 2; ++ns)
            for (int o = 0; o < 2; ++o) {
                IndexState y(ns);
                IndexAction ia(a);
                IndexObservation io(o);
                psi.push_back(m.count(&ia, &y, &io));
            }
    printf("{\"seed\": \"%s\", \"rec\": ", s);
    arr(rec, pi);
    printf(", \"obs_prob\": ");
    arr(probs, pd);
    printf(", \"phi_after\": ");
    arr(phi, pd);
    printf(", \"psi_after\": ");
    arr(psi, pd);
    printf("}");
}


/* BABNModel / DBNNode: sampleStateIndex, sampleObservationIndex, incrementCountsOf,
 * computeObservationProbability on a factored-tiger shaped model (K = 2: three binary state
 * features, one binary observation feature) whose listen observation node has `parents`.
 * The counts are filled through the reference's own DBNNode::count API. */
static void babn_model(char const* s, std::vector<int> parents)
{
    using bayes_adaptive::factored::BABNModel;
    Domain_Size sz(8, 3, 2);
    Domain_Feature_Size fsz({2, 2, 2}, {2});
    BABNModel::Indexing_Steps steps(indexing::stepSize(fsz._S), indexing::stepSize(fsz._O));
    BABNModel m(&sz, &fsz, &steps);
    IndexAction listen(2);
    for (int f = 0; f < 3; ++f) {
        m.resetTransitionNode(&listen, f, std::vector<int>({f}));
        for (int v = 0; v < 2; ++v) m.transitionNode(&listen, f).count(std::vector<int>({v}), v) = 5000;
    }
    for (int a = 0; a < 2; ++a) {
        IndexAction act(a);
        for (int f = 0; f < 3; ++f)
            for (int v = 0; v < 2; ++v) m.transitionNode(&act, f).count({}, v) = 5000 + 11.f * (float)(a + f + 2 * v);
        m.observationNode(&act, 0).count({}, 0) = 5000;
        m.observationNode(&act, 0).count({}, 1) = 4000;
    }
    m.resetObservationNode(&listen, 0, parents);
    {
        std::vector<int> pv(parents.size(), 0), pr(parents.size(), 2);
        if (parents.empty()) {
            m.observationNode(&listen, 0).setDirichletDistribution(pv, std::vector<float>({5000.f, 5000.f}));
        } else {
            do {
                bool informed = parents[0] == 0;
                m.observationNode(&listen, 0).count(pv, 0) = informed ? (pv[0] == 0 ? 8500.f : 1500.f) : 5000.f;
                m.observationNode(&listen, 0).count(pv, 1) = informed ? (pv[0] == 1 ? 8500.f : 1500.f) : 5000.f;
            } while (!indexing::increment(pv, pr));
        }
    }
    seed(s);
    std::vector<int> rec;
    std::vector<double> probs;
    int st = 5;
    for (int i = 0; i < 240; ++i) {
        int a = (i % 4 == 3) ? (i / 4) % 2 : 2; /* mostly listen, sometimes open */
        IndexState x(st);
        IndexAction ia(a);
        int ns = m.sampleStateIndex(&x, &ia, rnd::sample::Dir::sampleFromExpectedMult);
        IndexState y(ns);
        int o = m.sampleObservationIndex(&ia, &y, rnd::sample::Dir::sampleFromExpectedMult);
        IndexObservation io(o);
        m.incrementCountsOf(&x, &ia, &io, &y);
        IndexObservation other(1 - o);
        probs.push_back(m.computeObservationProbability(&io, &ia, &y, rnd::sample::Dir::expectedMult));
        probs.push_back(m.computeObservationProbability(&other, &ia, &y, rnd::sample::Dir::expectedMult));
        rec.push_back(a);
        rec.push_back(ns);
        rec.push_back(o);
        st = ns;
    }
    /* dump the listen observation CPT rows in parent-value order */
    std::vector<double> cpt;
    {
        std::vector<int> pv(parents.size(), 0), pr(parents.size(), 2);
        if (parents.empty()) {
            cpt.push_back(m.observationNode(&listen, 0).count(pv, 0));
            cpt.push_back(m.observationNode(&listen, 0).count(pv, 1));
        } else {
            do {
                cpt.push_back(m.observationNode(&listen, 0).count(pv, 0));
                cpt.push_back(m.observationNode(&listen, 0).count(pv, 1));
            } while (!indexing::increment(pv, pr));
        }
    }
    std::vector<double> tl;
    for (int f = 0; f < 3; ++f)
        for (int v = 0; v < 2; ++v)
            for (int w = 0; w < 2; ++w) tl.push_back(m.transitionNode(&listen, f).count(std::vector<int>({v}), w));
    printf("{\"seed\": \"%s\", \"parents\": ", s);
    arr(parents, pi);
    printf(", \"rec\": ");
    arr(rec, pi);
    printf(", \"obs_prob\": ");
    arr(probs, pd);
    printf(", \"listen_O_after\": ");
    arr(cpt, pd);
    printf(", \"listen_T_after\": ");
    arr(tl, pd);
    printf("}");
}


/* BABNModel::marginalizeOut (BABNModel.cpp:205-229) / DBNNode::marginalizeOut (DBNNode.cpp:40-80) of a
 * fully connected listen observation node onto every parent subset, and
 * BABNModel::Structure::flip_random_edge (BABNModel.cpp:16-31) */
static void marginalize(char const* s)
{
    using bayes_adaptive::factored::BABNModel;
    Domain_Size sz(8, 3, 2);
    Domain_Feature_Size fsz({2, 2, 2}, {2});
    BABNModel::Indexing_Steps steps(indexing::stepSize(fsz._S), indexing::stepSize(fsz._O));
    BABNModel m(&sz, &fsz, &steps);
    IndexAction listen(2);
    for (int f = 0; f < 3; ++f) {
        m.resetTransitionNode(&listen, f, std::vector<int>({f}));
        for (int v = 0; v < 2; ++v) m.transitionNode(&listen, f).count(std::vector<int>({v}), v) = 5000.25f + (float)f;
    }
    for (int a = 0; a < 2; ++a) {
        IndexAction act(a);
        for (int f = 0; f < 3; ++f)
            for (int v = 0; v < 2; ++v) m.transitionNode(&act, f).count({}, v) = 5000 + 11.f * (float)(a + f + 2 * v);
        m.observationNode(&act, 0).count({}, 0) = 5000;
        m.observationNode(&act, 0).count({}, 1) = 4000;
    }
    m.resetObservationNode(&listen, 0, std::vector<int>({0, 1, 2}));
    std::vector<double> full;
    {
        std::vector<int> pv(3, 0), pr(3, 2);
        int row = 0;
        do { /* counts whose float sums depend on the order of addition */
            for (int v = 0; v < 2; ++v) {
                float c = 1000.1f * (float)(row + 1) + 0.37f * (float)v + (row == 5 ? 1.e-3f : 0.f);
                m.observationNode(&listen, 0).count(pv, v) = c;
                full.push_back(c);
            }
            ++row;
        } while (!indexing::increment(pv, pr));
    }
    printf("{\"full\": ");
    arr(full, pd);
    printf(", \"onto\": [");
    for (int mask = 0; mask < 8; ++mask) {
        auto st = m.structure();
        std::vector<int> parents;
        for (int f = 0; f < 3; ++f)
            if ((mask >> f) & 1) parents.push_back(f);
        st.O[2][0] = parents;
        auto mm = m.marginalizeOut(st);
        std::vector<double> cpt;
        std::vector<int> pv(parents.size(), 0), pr(parents.size(), 2);
        if (parents.empty()) {
            cpt.push_back(mm.observationNode(&listen, 0).count(pv, 0));
            cpt.push_back(mm.observationNode(&listen, 0).count(pv, 1));
        } else {
            do {
                cpt.push_back(mm.observationNode(&listen, 0).count(pv, 0));
                cpt.push_back(mm.observationNode(&listen, 0).count(pv, 1));
            } while (!indexing::increment(pv, pr));
        }
        std::vector<double> tl;
        for (int f = 0; f < 3; ++f)
            for (int v = 0; v < 2; ++v)
                for (int w = 0; w < 2; ++w) tl.push_back(mm.transitionNode(&listen, f).count(std::vector<int>({v}), w));
        IndexAction open0(0);
        tl.push_back(mm.observationNode(&open0, 0).count({}, 1));
        printf("%s{\"mask\": %d, \"listen_O\": ", mask ? "," : "", mask);
        arr(cpt, pd);
        printf(", \"rest\": ");
        arr(tl, pd);
        printf("}");
    }
    printf("], ");
    seed(s);
    std::vector<int> edges = {0}, masks;
    for (int i = 0; i < 64; ++i) {
        BABNModel::Structure::flip_random_edge(&edges, 3);
        int mk = 0;
        for (int e : edges) mk |= 1 << e;
        masks.push_back(mk);
    }
    printf("\"seed\": \"%s\", \"flip_masks\": ", s);
    arr(masks, pi);
    printf("}");
}


/* regular Dirichlet mode: rnd::sample::gamma, sampleFromSampledMult, sampleMult (random.cpp:189-304) */
static void regular_dirichlet(char const* s)
{
    seed(s);
    std::vector<double> g;
    double const shapes[] = {0.3, 1, 5, 5000, 0, 8500, 0.9999, 1500, 2.5, 1e-3};
    for (int rep = 0; rep < 40; ++rep)
        for (double sh : shapes) g.push_back(rnd::sample::gamma(sh));
    std::vector<int> picks;
    float const rows[4][4] = {{5000, 5000, 0, 0}, {8500, 1500, 3, 0.25f}, {1, 2, 3, 4}, {0, 0, 7, 0}};
    for (int rep = 0; rep < 60; ++rep)
        for (auto const& r : rows) picks.push_back(rnd::sample::Dir::sampleFromSampledMult(r, 4));
    std::vector<double> mults;
    for (int rep = 0; rep < 10; ++rep)
        for (auto const& r : rows)
            for (float v : rnd::sample::Dir::sampleMult(r, 4)) mults.push_back(v);
    printf("{\"seed\": \"%s\", \"gamma\": ", s);
    arr(g, pd);
    printf(", \"picks\": ");
    arr(picks, pi);
    printf(", \"mults\": ");
    arr(mults, pd);
    printf(", \"next_u01\": %.17g}", rnd::uniform_rand01());
}

/* episode::run with the reference's RandomPlanner and RejectionSampling belief (n > 0) or its
 * PointEstimation belief (n == 0, on the continuous tiger so that episodes last the whole horizon) */
static void random_planner_episodes(char const* s, int n, int episodes)
{
    auto const type = n ? domains::Tiger::EPISODIC : domains::Tiger::CONTINUOUS;
    domains::Tiger env(type), sim(type);
    planners::RandomPlanner planner;
    beliefs::RejectionSampling rs((size_t)(n ? n : 1));
    beliefs::PointEstimation point;
    Belief& belief = n ? static_cast<Belief&>(rs) : static_cast<Belief&>(point);
    seed(s);
    std::vector<double> rets;
    std::vector<int> lens;
    for (int e = 0; e < episodes; ++e) {
        belief.initiate(sim);
        auto r = episode::run(planner, belief, env, sim, Horizon(10), Discount(.95));
        rets.push_back(r.ret.toDouble());
        lens.push_back(r.length);
        belief.free(sim);
    }
    printf("{\"seed\": \"%s\", \"particles\": %d, \"returns\": ", s, n);
    arr(rets, pd);
    printf(", \"lengths\": ");
    arr(lens, pi);
    printf(", \"next_u01\": %.17g}", rnd::uniform_rand01());
}

static void statistic()
{
    utils::Statistic st;
    double const xs[] = {1.5, -2.25, 10, -100, 3.125, 0, 7, 7, -1e-3, 42};
    for (double x : xs) st.add(x);
    printf("{\"x\": [1.5,-2.25,10,-100,3.125,0,7,7,-0.001,42], \"mean\": %.17g, \"var\": %.17g, \"count\": %.17g, \"stder\": %.17g}",
           st.mean(), st.var(), st.count(), st.stder());
}

static void expected_mult()
{
    float const rows[3][4] = {{5000, 0, 0, 0}, {8500, 1500, 3, 0.25f}, {1, 2, 3, 4}};
    printf("[");
    for (int r = 0; r < 3; ++r) {
        auto v = rnd::sample::Dir::expectedMult(rows[r], 4);
        std::vector<double> d(v.begin(), v.end());
        if (r) printf(",");
        arr(d, pd);
    }
    printf("]");
}

/* BADomainExtension::domainSize / terminal(s, a, s') / reward(s, a, s') as full tables (TigerBAExtension.cpp:21-44,
 * FactoredTigerBAExtension.cpp, GridWorldBAExtension.cpp:74-100, CollisionAvoidanceBAExtension.cpp:53-82):
 * pins which of s and s' each extension looks at. */
static void ext_tables(BADomainExtension const& ext)
{
    Domain_Size const sz = ext.domainSize();
    std::vector<int> term;
    std::vector<double> rew;
    for (int s = 0; s < sz._S; ++s)
        for (int a = 0; a < sz._A; ++a)
            for (int ns = 0; ns < sz._S; ++ns) {
                IndexAction ia(a);
                State const* x = ext.getState(s);
                State const* y = ext.getState(ns);
                term.push_back(ext.terminal(x, &ia, y).terminated() ? 1 : 0);
                rew.push_back(ext.reward(x, &ia, y).toDouble());
            }
    printf("{\"S\": %d, \"A\": %d, \"O\": %d, \"terminal\": ", sz._S, sz._A, sz._O);
    arr(term, pi);
    printf(", \"reward\": ");
    arr(rew, pd);
    printf("}");
}
static void feature_sizes(FBADomainExtension const& ext)
{
    Domain_Feature_Size const f = ext.domainFeatureSize();
    printf("{\"S\": ");
    arr(f._S, pi);
    printf(", \"O\": ");
    arr(f._O, pi);
    printf("}");
}

/* ChanceNode::addVisit / ActionNode::addVisit / child bookkeeping (MCTSTreeNodes.cpp:8-62): a three-action root whose
 * chance nodes receive a seeded sequence of returns; visit counts and running-mean Q after every visit. */
static void mcts_nodes(char const* s)
{
    seed(s);
    IndexAction a0(0), a1(1), a2(2);
    std::vector<Action const*> legal({&a0, &a1, &a2});
    ActionNode root(legal);
    ActionNode leaf(legal);
    std::vector<int> which, visits, has;
    std::vector<double> rets, qs;
    for (int i = 0; i < 400; ++i) {
        int const a   = rnd::slowRandomInt(0, 3);
        double const r = (rnd::uniform_rand01() < .3 ? -100.0 : 10.0) * rnd::uniform_rand01() - (double)(i % 7);
        ChanceNode& c  = *(root.begin() + a);
        c.addVisit(r);
        root.addVisit();
        which.push_back(a);
        rets.push_back(r);
        visits.push_back(c.visited());
        visits.push_back(root.visited());
        qs.push_back(c.qValue());
        int const o = i % 5;
        has.push_back(c.hasChild(o) ? 1 : 0);
        if (!c.hasChild(o)) c.addChild(o, &leaf);
        has.push_back(c.child(o) == &leaf ? 1 : 0);
    }
    printf("{\"seed\": \"%s\", \"action\": ", s);
    arr(which, pi);
    printf(", \"ret\": ");
    arr(rets, pd);
    printf(", \"visits\": ");
    arr(visits, pi);
    printf(", \"q\": ");
    arr(qs, pd);
    printf(", \"child\": ");
    arr(has, pi);
    printf("}");
}

/* BAPOMDPState (BAPOMDPState.cpp) through the BAState interface: sampleStateIndex / sampleObservationIndex /
 * computeObservationProbability / incrementCountsOf on the counts of flat_model(), with a copy() taken half way
 * whose counts must not move afterwards (deep copy, BAPOMDPState.cpp copy()). */
static void bapomdp_state(char const* s)
{
    Domain_Size sz(2, 3, 2);
    bayes_adaptive::table::BAFlatModel m(&sz);
    for (int st = 0; st < 2; ++st)
        for (int a = 0; a < 3; ++a)
            for (int ns = 0; ns < 2; ++ns) {
                IndexState x(st), y(ns);
                IndexAction ia(a);
                m.count(&x, &ia, &y) = (a == 2) ? ((st == ns) ? 5000.f : 0.f) : 5000.f + 7.f * (float)(st + 2 * ns + a);
            }
    for (int a = 0; a < 3; ++a)
        for (int ns = 0; ns < 2; ++ns)
            for (int o = 0; o < 2; ++o) {
                IndexState y(ns);
                IndexAction ia(a);
                IndexObservation io(o);
                m.count(&ia, &y, &io) = (a == 2) ? ((ns == o) ? 8500.f : 1500.f) : 5000.f + 3.f * (float)(o + a);
            }
    IndexState dom(1);
    BAPOMDPState state(&dom, m);
    BAState* ba = &state;
    BAState* frozen = nullptr;
    seed(s);
    std::vector<int> rec;
    std::vector<double> probs;
    int st = 1;
    for (int i = 0; i < 200; ++i) {
        if (i == 100) frozen = ba->copy(&dom);
        int a = (i * 7) % 3;
        IndexState x(st);
        IndexAction ia(a);
        int ns = ba->sampleStateIndex(&x, &ia, rnd::sample::Dir::sampleFromExpectedMult);
        IndexState y(ns);
        int o = ba->sampleObservationIndex(&ia, &y, rnd::sample::Dir::sampleFromExpectedMult);
        IndexObservation io(o);
        ba->incrementCountsOf(&x, &ia, &io, &y);
        probs.push_back(ba->computeObservationProbability(&io, &ia, &y, rnd::sample::Dir::expectedMult));
        rec.push_back(a);
        rec.push_back(ns);
        rec.push_back(o);
        st = ns;
    }
    auto dump = [&](BAPOMDPState* b, char const* name) {
        std::vector<double> all;
        for (int s0 = 0; s0 < 2; ++s0)
            for (int a = 0; a < 3; ++a)
                for (int ns = 0; ns < 2; ++ns) {
                    IndexState x(s0), y(ns);
                    IndexAction ia(a);
                    all.push_back(b->model()->count(&x, &ia, &y));
                }
        for (int a = 0; a < 3; ++a)
            for (int ns = 0; ns < 2; ++ns)
                for (int o = 0; o < 2; ++o) {
                    IndexState y(ns);
                    IndexAction ia(a);
                    IndexObservation io(o);
                    all.push_back(b->model()->count(&ia, &y, &io));
                }
        printf(", \"%s\": ", name);
        arr(all, pd);
    };
    printf("{\"seed\": \"%s\", \"rec\": ", s);
    arr(rec, pi);
    printf(", \"obs_prob\": ");
    arr(probs, pd);
    dump(&state, "counts_after");
    dump(static_cast<BAPOMDPState*>(frozen), "counts_of_copy_at_100");
    printf(", \"domain_state_index\": %d}", ba->index());
    delete frozen;
}

/* FBAPOMDPState (FBAPOMDPState.cpp) through the BAState interface on the model of babn_model({0, 2}); a copy() at
 * step 60 must keep its counts. */
static void fbapomdp_state(char const* s)
{
    using bayes_adaptive::factored::BABNModel;
    Domain_Size sz(8, 3, 2);
    Domain_Feature_Size fsz({2, 2, 2}, {2});
    BABNModel::Indexing_Steps steps(indexing::stepSize(fsz._S), indexing::stepSize(fsz._O));
    BABNModel m(&sz, &fsz, &steps);
    IndexAction listen(2);
    for (int f = 0; f < 3; ++f) {
        m.resetTransitionNode(&listen, f, std::vector<int>({f}));
        for (int v = 0; v < 2; ++v) m.transitionNode(&listen, f).count(std::vector<int>({v}), v) = 5000;
    }
    for (int a = 0; a < 2; ++a) {
        IndexAction act(a);
        for (int f = 0; f < 3; ++f)
            for (int v = 0; v < 2; ++v) m.transitionNode(&act, f).count({}, v) = 5000 + 11.f * (float)(a + f + 2 * v);
        m.observationNode(&act, 0).count({}, 0) = 5000;
        m.observationNode(&act, 0).count({}, 1) = 4000;
    }
    std::vector<int> parents({0, 2});
    m.resetObservationNode(&listen, 0, parents);
    {
        std::vector<int> pv(2, 0), pr(2, 2);
        do {
            m.observationNode(&listen, 0).count(pv, 0) = pv[0] == 0 ? 8500.f : 1500.f;
            m.observationNode(&listen, 0).count(pv, 1) = pv[0] == 1 ? 8500.f : 1500.f;
        } while (!indexing::increment(pv, pr));
    }
    IndexState dom(5);
    FBAPOMDPState state(&dom, m);
    BAState* ba     = &state;
    BAState* frozen = nullptr;
    seed(s);
    std::vector<int> rec;
    std::vector<double> probs;
    int st = 5;
    for (int i = 0; i < 120; ++i) {
        if (i == 60) frozen = ba->copy(&dom);
        int a = (i % 4 == 3) ? (i / 4) % 2 : 2;
        IndexState x(st);
        IndexAction ia(a);
        int ns = ba->sampleStateIndex(&x, &ia, rnd::sample::Dir::sampleFromExpectedMult);
        IndexState y(ns);
        int o = ba->sampleObservationIndex(&ia, &y, rnd::sample::Dir::sampleFromExpectedMult);
        IndexObservation io(o);
        ba->incrementCountsOf(&x, &ia, &io, &y);
        probs.push_back(ba->computeObservationProbability(&io, &ia, &y, rnd::sample::Dir::expectedMult));
        rec.push_back(a);
        rec.push_back(ns);
        rec.push_back(o);
        st = ns;
    }
    auto dump = [&](FBAPOMDPState* b, char const* name) {
        std::vector<double> cpt;
        std::vector<int> pv(2, 0), pr(2, 2);
        do {
            cpt.push_back(b->model()->observationNode(&listen, 0).count(pv, 0));
            cpt.push_back(b->model()->observationNode(&listen, 0).count(pv, 1));
        } while (!indexing::increment(pv, pr));
        for (int f = 0; f < 3; ++f)
            for (int v = 0; v < 2; ++v)
                for (int w = 0; w < 2; ++w) cpt.push_back(b->model()->transitionNode(&listen, f).count(std::vector<int>({v}), w));
        printf(", \"%s\": ", name);
        arr(cpt, pd);
    };
    printf("{\"seed\": \"%s\", \"rec\": ", s);
    arr(rec, pi);
    printf(", \"obs_prob\": ");
    arr(probs, pd);
    dump(&state, "listen_O_T_after");
    dump(static_cast<FBAPOMDPState*>(frozen), "listen_O_T_of_copy_at_60");
    /* BABNModel::LogBDScore (BABNModel.cpp:451-478, DBNNode.cpp:82-117) of the model after the walk, and of the copy
     * taken at step 60, against the model the walk started from */
    printf(", \"log_bd_score_after\": ");
    pd(state.model()->LogBDScore(m));
    printf(", \"log_bd_score_copy_at_60\": ");
    pd(static_cast<FBAPOMDPState*>(frozen)->model()->LogBDScore(m));
    printf(", \"log_bd_score_of_prior\": ");
    pd(m.LogBDScore(m));
    printf("}");
    delete frozen;
}

int main(int argc, char** argv)
{
    START_EASYLOGGINGPP(argc, argv);
    el::Configurations conf;
    conf.setToDefault();
    conf.setGlobally(el::ConfigurationType::Enabled, "false");
    el::Loggers::reconfigureAllLoggers(conf);
    rnd::initiate();

    printf("{");
    key("rng");
    printf("[");
    rng_vectors("1");
    printf(",");
    rng_vectors("7");
    printf(",");
    rng_vectors("hello fba");
    printf("]");

    key("tiger_episodic");
    { domains::Tiger d(domains::Tiger::EPISODIC); walk(d, "3", 300); }
    key("tiger_continuous");
    { domains::Tiger d(domains::Tiger::CONTINUOUS); walk(d, "4", 300); }
    key("ftiger3_episodic");
    { domains::FactoredTiger d(domains::FactoredTiger::EPISODIC, 3); walk(d, "5", 300); }
    key("ftiger2_continuous");
    { domains::FactoredTiger d(domains::FactoredTiger::CONTINUOUS, 2); walk(d, "6", 300); }
    key("gridworld5");
    { domains::GridWorld d(5); walk(d, "8", 400); }
    key("gridworld7");
    { domains::GridWorld d(7); walk(d, "9", 400); }
    key("gridworld3");
    { domains::GridWorld d(3); walk(d, "10", 200); }

    key("ca_5_3_1_random");
    { domains::CollisionAvoidance d(5, 3, 1, domains::CollisionAvoidance::INIT_RANDOM_POSITION); walk(d, "30", 300); }
    key("ca_5_3_3_random");
    { domains::CollisionAvoidance d(5, 3, 3, domains::CollisionAvoidance::INIT_RANDOM_POSITION); walk(d, "31", 400); }
    key("ca_7_7_2_centered");
    { domains::CollisionAvoidance d(7, 7, 2, domains::CollisionAvoidance::INITIALIZE_CENTRE); walk(d, "32", 400); }
    key("ca_4_5_2_obs_prob");
    { domains::CollisionAvoidance d(4, 5, 2, domains::CollisionAvoidance::INIT_RANDOM_POSITION); ca_obs_table(d, 4, 5, 2); }

    key("sysadmin3_independent");
    { domains::SysAdmin d(3, "independent"); walk(d, "50", 300); }
    key("sysadmin5_linear");
    { domains::SysAdmin d(5, "linear"); walk(d, "51", 400); }
    key("sysadmin8_linear");
    { domains::SysAdmin d(8, "linear"); walk(d, "52", 400); }
    key("sysadmin4_linear_tables");
    sysadmin_tables(4, "linear");

    key("coffee");
    { domains::CoffeeProblem d(""); walk(d, "60", 400); }
    key("coffee_boutilier");
    { domains::CoffeeProblem d("boutilier"); walk(d, "61", 400); }

    key("agr");
    { domains::AGR d(10); walk(d, "62", 600); }

    key("tiger_obs_prob");
    { domains::Tiger d(domains::Tiger::EPISODIC); obs_table(d, 2, 3, 2); }
    key("ftiger2_obs_prob");
    { domains::FactoredTiger d(domains::FactoredTiger::EPISODIC, 2); obs_table(d, 8, 3, 2); }

    key("gridworld3_obs_prob");
    gridworld_obs_table(3);
    key("gridworld4_obs_prob");
    gridworld_obs_table(4);
    key("is_gridworld");
    printf("[");
    is_gridworld("22", 5, 40, 8);
    printf(",");
    is_gridworld("23", 3, 64, 6);
    printf("]");

    key("weighted_filter_ops");
    weighted_filter_ops();

    key("reject_tiger");
    printf("[");
    reject_tiger("11", 64);
    printf(",");
    reject_tiger("12", 200);
    printf("]");

    key("is_tiger");
    printf("[");
    is_tiger("13", 48);
    printf(",");
    is_tiger("14", 300);
    printf("]");

    key("flat_model");
    flat_model("15");

    key("babn_model");
    printf("[");
    babn_model("17", {0});
    printf(",");
    babn_model("18", {0, 2});
    printf(",");
    babn_model("19", {1});
    printf(",");
    babn_model("20", {});
    printf(",");
    babn_model("21", {0, 1, 2});
    printf("]");

    key("marginalize");
    marginalize("23");

    key("regular_dirichlet");
    regular_dirichlet("40");

    key("random_planner_episodes");
    random_planner_episodes("16", 32, 60);

    key("random_planner_point_estimate");
    random_planner_episodes("17", 0, 40);

    key("statistic");
    statistic();

    {
        using namespace bayes_adaptive::domain_extensions;
        key("ext_tiger_episodic");
        { TigerBAExtension e(domains::Tiger::EPISODIC); ext_tables(e); }
        key("ext_tiger_continuous");
        { TigerBAExtension e(domains::Tiger::CONTINUOUS); ext_tables(e); }
        key("ext_ftiger2_episodic");
        { FactoredTigerBAExtension e(domains::FactoredTiger::EPISODIC, 2); ext_tables(e); }
        key("ext_ftiger1_continuous");
        { FactoredTigerBAExtension e(domains::FactoredTiger::CONTINUOUS, 1); ext_tables(e); }
        key("ext_gridworld3");
        { GridWorldBAExtension e(3); ext_tables(e); }
        key("ext_gridworld4");
        { GridWorldBAExtension e(4); ext_tables(e); }
        key("ext_ca_5_3_1");
        { CollisionAvoidanceBAExtension e(5, 3, 1); ext_tables(e); }
        key("ext_ca_3_3_2");
        { CollisionAvoidanceBAExtension e(3, 3, 2); ext_tables(e); }
        key("fext_ftiger3");
        { FactoredTigerFBAExtension e(3); feature_sizes(e); }
        key("fext_gridworld7");
        { GridWorldFBAExtension e(7); feature_sizes(e); }
        key("fext_ca_5_3_2");
        { CollisionAvoidanceFBAExtension e(5, 3, 2, domains::CollisionAvoidance::INIT_RANDOM_POSITION); feature_sizes(e); }
        key("fext_sysadmin4");
        { SysAdminFBAExtension e(4); feature_sizes(e); }
    }
    key("mcts_nodes");
    mcts_nodes("70");
    key("bapomdp_state");
    bapomdp_state("71");
    key("fbapomdp_state");
    fbapomdp_state("72");

    key("expected_mult");
    expected_mult();

    printf("\n}\n");
    return 0;
}
