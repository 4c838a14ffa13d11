// fba_search.hip -- the tree-search kernels of the BA-POMCP engine (gfx950 / CDNA4, wave64).
//
//   search_kernel        POUCT / RBAPOUCT tree search, one lane per tree          (P1-P8)
//   search_hist_kernel   the same search on history particles, four lanes per tree (BASELINE configs[3])
//
// The search is latency bound and gets its throughput from running one independent tree per lane (or quad), all
// lanes executing the common "simulate one step" body together.  No dense contraction, no MFMA.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "fba_kernels_common.h"

namespace fba {

// ---------------------------------------------------------------------------------------------
// search_kernel: one lane = one slot = one tree; `sims` simulations, sequential semantics.
// POUCT::selectAction POUCT.cpp:63-129, RBAPOUCT::selectAction RBAPOUCT.cpp:67-153 (the root
// particle's counts are read and never written: StepType::KeepCounts).
// The recursion traverseActionNode / traverseChanceNode / rollout (POUCT.cpp:183-303) is unrolled
// into a state machine whose every iteration performs exactly one simulator.step, so the 64
// trees of a wave execute the expensive part (Philox + Dirichlet-row sampling) in lock-step.
// LDS per lane: the path needed for the bottom-up back-up, [depth][lane], and -- when a particle
// record fits SEARCH_STAGE_WORDS (STAGE) -- the root particle's whole count blob, [word][lane], fetched with
// one burst of 16-byte loads per simulation so that no step waits on HBM for its Dirichlet rows.
// ---------------------------------------------------------------------------------------------
// FTIGER > 0: the simulator is the factored-tiger FBA-POMDP with FTIGER binary state features (expected
// Dirichlet mode); its step is ftiger_step<FTIGER>, the layout restated as literals.
// TIGER_POMDP: planning on the tiger POMDP itself (BASELINE configs[0]); the sizes and the domain are literals.
__host__ __device__ __forceinline__ bool root_children_in_lds(const Problem& P, const DeviceState& D)
{
    return P.A * P.O <= ROOT_CHILDREN && D.max_nodes <= 32767;
}

// FTP: the factored-tiger records are packed (PackedFtigerView).
//
// ETIGER: the episodic tiger family (tiger and factored tiger, POMDP or Bayes-adaptive simulator): A = 3, O = 2, and only
// `listen` (action 2) continues -- opening a door ends the episode in the domain and in the BA extensions alike
// (Tiger.cpp:75-79, TigerBAExtension.cpp:32-36, FactoredTigerBAExtension.cpp) -- so a node has at most two children and the
// tree is a binary tree over the observations heard.  What a simulation costs this kernel is the 64-byte sectors it has
// the memory system fetch (one more random sector per simulation is +45 ms on the bench), and the two
// nodes below the root are on nearly every path; they live in LDS beside the path (18 words per lane: counts as uint16,
// child indices as uint16, Q as fp64), never in HBM.  The path itself shrinks to 16 bits per level -- node index (12 bits:
// the host bounds the tree at 2^(depth+1) + 1 nodes), action (2), and the reward as a code (-1 / 10 / -100: the only
// rewards of these domains) -- which is what pays for the room.  Same draws, same arithmetic, same results.
constexpr int ET_NODE_WORDS = 9;   // an LDS node: n0 | n1 << 16, n2, child0 | child1 << 16, q0, q1, q2 (two words each)
__host__ __device__ __forceinline__ bool etiger_ok(const Problem& P, const DeviceState& D)
{
    return (P.domain == FBA_DOM_TIGER_EPISODIC || P.domain == FBA_DOM_FTIGER_EPISODIC) && P.A == 3 && P.O == 2 && P.sims <= 65535 &&
           D.max_nodes <= 4093 && !D.hash && P.planner != FBA_PLANNER_RANDOM;
}
// 32-bit words of the dynamic LDS allocation of one search workgroup in front of the staged model description
__host__ __device__ __forceinline__ size_t search_lds_words(const Problem& P, const DeviceState& D, bool stage, bool etiger)
{
    const int depth_cap = P.max_depth > 0 ? P.max_depth : 1;
    if (etiger)
        return (size_t)((depth_cap + 1) / 2) * SEARCH_BLOCK + (stage ? (size_t)P.Cs * SEARCH_BLOCK : 0) + (size_t)2 * ET_NODE_WORDS * SEARCH_BLOCK;
    return (size_t)depth_cap * SEARCH_BLOCK * 2 + (stage ? (size_t)P.Cs * SEARCH_BLOCK : 0) +
           (root_children_in_lds(P, D) ? (size_t)P.A * P.O * SEARCH_BLOCK / 2 : 0);
}
__device__ __forceinline__ int et_reward_code(double r) { return r == -1.0 ? 0 : (r == 10.0 ? 1 : 2); }
__device__ __forceinline__ double et_reward(int code) { return code == 0 ? -1.0 : (code == 1 ? 10.0 : -100.0); }
#ifdef FBA_PROFILE_SEARCH
// profiling build only (scripts/search_regions.py): shader-clock cycles a wave spends in each region of the search loop
__device__ unsigned long long g_search_prof[8];
#define PROF_MARK(r) { const long long now_ = clock64(); prof_[r] += now_ - prev_; prev_ = now_; }
#else
#define PROF_MARK(r)
#endif
template <bool STAGE, int AMAX, bool REG, int TIGER_TABLE, int MODEL, int FTIGER = 0, bool TIGER_POMDP = false, bool FTP = false, bool ETIGER = false>
__global__ void __launch_bounds__(SEARCH_BLOCK) search_kernel(Problem P, DeviceState D)
{
    if (TIGER_POMDP) {
        P.S = 2; P.A = 3; P.O = 2; P.C = 0; P.Cs = 4; P.planner = FBA_PLANNER_POUCT;
        if (P.domain != FBA_DOM_TIGER_CONTINUOUS) P.domain = FBA_DOM_TIGER_EPISODIC;
        P.belief = P.belief == FBA_BELIEF_IMPORTANCE ? FBA_BELIEF_IMPORTANCE : FBA_BELIEF_REJECTION;
        D.cn_off = 1; D.cq_off = 4; D.child_off = 10; D.node_words = 16; D.hash = nullptr;  // node layout of A = 3, O = 2 (fba_engine.hip)
    }
    // one instantiation per simulator: the launcher passes the model it read from P, so restating it
    // here drops the other simulators' code (a plain-POMDP search carries every domain's step(),
    // the Bayes-adaptive ones none of them) from this instantiation
    P.model = MODEL;
    if (FTIGER > 0) {  // sizes of factored tiger with FTIGER - 1 irrelevant features
        P.S = 1 << FTIGER; P.A = 3; P.O = 2;
        if (P.domain != FBA_DOM_FTIGER_CONTINUOUS) P.domain = FBA_DOM_FTIGER_EPISODIC;
        if (ETIGER) P.domain = FBA_DOM_FTIGER_EPISODIC;
    }
    if (ETIGER) { P.A = 3; P.O = 2; }
    // TIGER_TABLE: the launcher has checked that this is the tabular BA-POMDP over (episodic or
    // continuous) tiger; restating its sizes as literals lets the compiler unroll the two-entry
    // Dirichlet rows and fold every model / domain branch.  Same code, same results.
    if (TIGER_TABLE) {
        P.model = FBA_MODEL_BA_TABLE; P.planner = FBA_PLANNER_POUCT;
        P.S = 2; P.A = 3; P.O = 2; P.phi_len = 12; P.C = 24; P.Cs = 32;
        if (P.domain != FBA_DOM_TIGER_CONTINUOUS) P.domain = FBA_DOM_TIGER_EPISODIC;
        P.belief = P.belief == FBA_BELIEF_IMPORTANCE ? FBA_BELIEF_IMPORTANCE : FBA_BELIEF_REJECTION;
        D.cn_off = 1; D.cq_off = 4; D.child_off = 10; D.node_words = 16; D.hash = nullptr;  // node layout of A = 3, O = 2 (fba_engine.hip)
        if (TIGER_TABLE == 2) { P.C = 12; P.Cs = 16; }  // packed particles (PackedView): 24 uint16 + state in 64 bytes
    }
    if (ETIGER && (TIGER_TABLE || TIGER_POMDP)) P.domain = FBA_DOM_TIGER_EPISODIC;
    extern __shared__ double lds[];
    __shared__ __attribute__((aligned(8))) float s_prior[TIGER_TABLE == 2 ? 24 : 2];
    const int lane = threadIdx.x;
    const int e    = blockIdx.x * SEARCH_BLOCK + lane;
    if (TIGER_TABLE == 2) {
        if (lane < 24) s_prior[lane] = D.prior_dense[lane];
        __syncthreads();
    }
    if (MODEL == FBA_MODEL_BA_FACTORED && FTIGER == 0) {  // (the factored-tiger instantiations carry their layout as constants)
        // the factored model's description (which parents, how many values, where the rows start) is
        // consulted several times per sampled feature: keep the part in use in LDS, at the end of
        // this workgroup's allocation, instead of chasing it through global memory
        size_t words = search_lds_words(P, D, STAGE, ETIGER);
        words = (words + 3) & ~(size_t)3;  // 16-byte aligned
        uint4* dst       = reinterpret_cast<uint4*>(reinterpret_cast<uint32_t*>(lds) + words);
        const uint4* src = reinterpret_cast<const uint4*>(P.fd);
        for (int k = lane; k < (P.fd_bytes + 15) / 16; k += SEARCH_BLOCK) dst[k] = src[k];
        __syncthreads();
        P.fd = reinterpret_cast<const FDesc*>(dst);
    }
    if (e >= P.E || !D.active[e]) return;

    const int depth_cap = P.max_depth > 0 ? P.max_depth : 1;
    // Rewards on the path are kept as fp32: every reward of every domain here (and of the BA extensions) is a
    // small integer or a multiple of 1/2, so the round trip through float is exact and the back-up still
    // computes in fp64 -- and 4 bytes x depth x 64 lanes of LDS per wave buy one more resident wave per CU.
    float* path_r       = reinterpret_cast<float*>(lds) + lane;                         // [depth][block]
    int32_t* path_na    = reinterpret_cast<int32_t*>(path_r - lane + (size_t)depth_cap * SEARCH_BLOCK) + lane;
    // ETIGER: the path is [depth][block] of uint16 {node (12 bits; at level 1: which of the two LDS nodes), action << 12, reward code << 14}
    uint16_t* path16    = reinterpret_cast<uint16_t*>(lds) + lane;
    float* stage        = ETIGER ? reinterpret_cast<float*>(lds) + (size_t)((depth_cap + 1) / 2) * SEARCH_BLOCK + lane
                                 : reinterpret_cast<float*>(path_na - lane + (size_t)depth_cap * SEARCH_BLOCK) + lane;  // [Cs][block]
    // children of the root, [a*O + o][block], when there are at most ROOT_CHILDREN of them
    // (node indices fit 16 bits up to 32 766 simulations; beyond that the root's children stay in its record)
    const bool root_lds = !ETIGER && root_children_in_lds(P, D);
    int16_t* rootch     = reinterpret_cast<int16_t*>(stage - lane + (STAGE ? (size_t)P.Cs * SEARCH_BLOCK : 0)) + lane;
    // ETIGER: the two nodes below the root, [2][ET_NODE_WORDS][block]
    uint32_t* l1        = reinterpret_cast<uint32_t*>(stage - lane + (STAGE ? (size_t)P.Cs * SEARCH_BLOCK : 0)) + lane;

    Rng g               = slot_rng(P, D, e);
    const int hist_len  = D.t[e];
    const int max_tree_depth = min(P.horizon - hist_len, P.max_depth);
    const int W         = D.node_words;
    int32_t* tree       = D.nodes + (size_t)e * D.max_nodes * W;
    const float* prec   = D.p_rec + pbase(P, e, D.bufsel[e]) * (size_t)P.Cs;

    // (the nested belief stores dense records only: never with the packed instantiations)
    const bool nested = TIGER_TABLE != 2 && !TIGER_POMDP && !FTP && MODEL != FBA_MODEL_POMDP && P.nested != 0;
    if (P.planner == FBA_PLANNER_RANDOM) {  // RandomPlanner::selectAction RandomPlanner.cpp:14-24
        g.stream(FBA_PHASE_SEARCH, (uint32_t)P.sims);
        int ns = 0;
        const int src = nested ? nested_sample(P, D, e, g, ns) : belief_sample_uniform(P, D, g);
        D.action[e]   = domain_random_action(P, g, nested ? ns : slot_lazy(D, e) ? lazy_state(P, D, e, src) : rec_state(prec + (size_t)src * P.Cs, P.C));
        return;
    }

    // addLegalActions(belief.sample(), ...): the probe draw lives in its own stream (unit = sims)
    // and its result is not needed -- legal actions do not depend on the state in these domains.
    node_init(D, tree, P.A, P.O);
    int n_nodes = 1, tree_depth = 0;
    unsigned long long steps = 0;
    // The root is on the path of every simulation: its visit counts and Q values live in registers
    // for the whole search and its child pointers sit in LDS.
    int r_vis = 0, r_cn[AMAX];
    double r_cq[AMAX];
#pragma unroll
    for (int a = 0; a < AMAX; ++a) { r_cn[a] = 0; r_cq[a] = 0.0; }
    if (root_lds)
        for (int k = 0; k < P.A * P.O; ++k) rootch[k * SEARCH_BLOCK] = -1;
    uint32_t l1_exists = 0;   // ETIGER: bit o = the root's child after (listen, o) has been created
    // ETIGER: header and chosen Q of the path's nodes at levels 2 and 3 as this simulation's descent loaded them -- nothing
    // else writes this tree, so the back-up need not fetch their sectors again (by then they have left the L2)
    int4 cy_h2 = make_int4(0, 0, 0, 0), cy_h3 = make_int4(0, 0, 0, 0);
    double cy_q2 = 0, cy_q3 = 0;
    int4* tab      = hash_table(D, e);
    const uint32_t epoch = D.hash ? hash_begin_search(D, e, tab, 0, 1) : 0;

    const bool lazy = slot_lazy(D, e);  // particle states are still the episode's start-state draws (lazy_state)
    // -P ts: TSPlanner / BATSPlanner (src/planners/ts/TSPlanner.cpp:16-29, bayes-adaptive/BATSPlanner.cpp:19-34) sample
    // the belief once and plan on that point estimate, whose sample() draws nothing: every simulation starts
    // from the same particle and its stream begins with the UCB tie-break.
    int ts_src = -1, ts_state = 0;
    if (P.planner == FBA_PLANNER_TS) {
        g.stream(FBA_PHASE_SEARCH, (uint32_t)P.sims + 2u);
        ts_src = nested ? nested_sample(P, D, e, g, ts_state) : belief_sample_uniform(P, D, g);
    }
    int sim = 0, mode = 0;  // 0 = start a simulation, 1 = in the tree, 2 = rollout
    int s = 0, node = 0, dtg = 0, plen = 0, rdepth = 0;
    const float* cnt = prec;
    double rret = 0, rdisc = 1;
#ifdef FBA_PROFILE_SEARCH
    long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prev_ = clock64();
#endif
    while (true) {
        PROF_MARK(6)
        if (mode == 0) {
            if (sim >= P.sims) break;
            g.stream(FBA_PHASE_SEARCH, (uint32_t)sim);
            int nest_state = ts_state;
            const int src = ts_src >= 0 ? ts_src : (nested ? nested_sample(P, D, e, g, nest_state) : belief_sample_uniform(P, D, g));
            cnt = prec + (size_t)src * P.Cs;
            if (STAGE) {
                const float4* rp = reinterpret_cast<const float4*>(cnt);
                const int n4 = (P.C + 4) >> 2;  // counts and the state word; the padding behind them is not needed
                // up to twelve 16-byte loads in flight before the first of them is waited for: with a trip count the compiler does
                // not know, a load-then-store loop is one trip to memory per 16 bytes (C3: nine in a row, 42 % of the kernel)
                constexpr int NB = TIGER_TABLE == 2 ? 4 : 12;   // (same-box A/B on C3, nine 16-byte pieces: 4 -> 385.9 ms per tick, 8 -> 374, 12 -> 365.5)
                for (int k0 = 0; k0 < n4; k0 += NB) {
                    float4 v[NB];
#pragma unroll
                    for (int q = 0; q < NB; ++q) v[q] = rp[min(k0 + q, n4 - 1)];
#pragma unroll
                    for (int q = 0; q < NB; ++q) {
                        const int k = k0 + q;
                        if (k < n4) {
                            stage[(4 * k + 0) * SEARCH_BLOCK] = v[q].x;
                            stage[(4 * k + 1) * SEARCH_BLOCK] = v[q].y;
                            stage[(4 * k + 2) * SEARCH_BLOCK] = v[q].z;
                            stage[(4 * k + 3) * SEARCH_BLOCK] = v[q].w;
                        }
                    }
                }
                s = __float_as_int(stage[P.C * SEARCH_BLOCK]);
            } else {
                s = rec_state(cnt, P.C);
            }
            if (lazy) s = lazy_state(P, D, e, src);
            if (nested) s = nest_state;
            node = 0; dtg = max_tree_depth; plen = 0; mode = 1;
        }
        PROF_MARK(0)
        bool finish = false, do_step = true;
        double delayed = 0;
        int a = 0;
        if (mode == 1) {  // traverseActionNode
            tree_depth = max(tree_depth, max_tree_depth - dtg);
            if (dtg == 0) { finish = true; do_step = false; }
            else if (ETIGER) {
                // the node's statistics from where they live -- registers (root), LDS (the two nodes below it, node = -1 - o),
                // HBM (the rest) -- then ONE selection for all lanes
                int cn[AMAX], vis = r_vis;
                double cq[AMAX];
#pragma unroll
                for (int a2 = 0; a2 < AMAX; ++a2) { cn[a2] = r_cn[a2]; cq[a2] = r_cq[a2]; }
                if (node < 0) {
                    const uint32_t* nd = l1 + (size_t)(-1 - node) * ET_NODE_WORDS * SEARCH_BLOCK;
                    const uint32_t w0 = nd[0], w1 = nd[1 * SEARCH_BLOCK];
                    cn[0] = (int)(w0 & 0xffffu); cn[1] = (int)(w0 >> 16); cn[2] = (int)(w1 & 0xffffu);
#pragma unroll
                    for (int a2 = 0; a2 < 3; ++a2)
                        cq[a2] = __hiloint2double((int)nd[(4 + 2 * a2) * SEARCH_BLOCK], (int)nd[(3 + 2 * a2) * SEARCH_BLOCK]);
                    vis = (cn[0] + cn[1]) + cn[2];
                } else if (node > 0) {
                    const int32_t* rec = tree + (size_t)node * W;
                    const int4 h      = *reinterpret_cast<const int4*>(rec);
                    const double2 q01 = *reinterpret_cast<const double2*>(rec + 4);
                    const double q2   = *reinterpret_cast<const double*>(rec + 8);
                    vis = h.x; cn[0] = h.y; cn[1] = h.z; cn[2] = h.w;
                    cq[0] = q01.x; cq[1] = q01.y; cq[2] = q2;
                }
                a = ucb_pick<AMAX>(P, g, D.log1p_tab, vis, cn, cq, true);
                if (plen == 2) { cy_h2 = make_int4(vis, cn[0], cn[1], cn[2]); cy_q2 = a == 0 ? cq[0] : (a == 1 ? cq[1] : cq[2]); }
                if (plen == 3) { cy_h3 = make_int4(vis, cn[0], cn[1], cn[2]); cy_q3 = a == 0 ? cq[0] : (a == 1 ? cq[1] : cq[2]); }
            }
            else if (node == 0) a = ucb_pick<AMAX>(P, g, D.log1p_tab, r_vis, r_cn, r_cq, true);
            else a = ucb_select<AMAX>(P, D, g, tree + (size_t)node * W, true);
        } else {          // rollout: uniformly random action
            a = domain_random_action(P, g, s);
        }
        PROF_MARK(1)
        int o;
        double r;
        bool term;
        if (do_step) {
            if (FTIGER > 0 && STAGE && FTP)
                term = ftiger_step_packed<(FTIGER > 0 ? FTIGER : 1)>(P, g, LdsView<SEARCH_BLOCK>{stage}, s, a, o, r, NoInc{});
            else if (FTIGER > 0 && STAGE) term = ftiger_step<(FTIGER > 0 ? FTIGER : 1)>(P, g, LdsView<SEARCH_BLOCK>{stage}, s, a, o, r, NoInc{});
            else if (FTIGER > 0) term = ftiger_step<(FTIGER > 0 ? FTIGER : 1)>(P, g, GlobalSearchView{cnt}, s, a, o, r, NoInc{});
            else if (TIGER_TABLE == 2)
                term = tiger_step_packed(P, g, [&](int w) { return __float_as_uint(stage[w * SEARCH_BLOCK]); }, s_prior, s, a, o, r, NoInc{});
            else if (STAGE) term = sim_step<REG>(P, g, LdsView<SEARCH_BLOCK>{stage}, s, a, o, r, NoInc{});
            else term = sim_step<REG>(P, g, GlobalSearchView{cnt}, s, a, o, r, NoInc{});
            ++steps;
#ifdef FBA_PROFILE_SEARCH
        }
        PROF_MARK(2)
        if (do_step) {
#endif
            if (ETIGER && mode == 1) {  // traverseChanceNode on the episodic tiger tree
                path16[(size_t)plen * SEARCH_BLOCK] = (uint16_t)((node < 0 ? -1 - node : node) | (a << 12) | (et_reward_code(r) << 14));
                ++plen;
                if (term) finish = true;
                else {   // (a continuing step is a `listen`: the child after observation o)
                    int c = 0;   // 0 = none yet (the root is nobody's child)
                    if (node == 0) c = ((l1_exists >> o) & 1u) ? -1 - o : 0;
                    else if (node < 0) {
                        const uint32_t w2 = l1[((size_t)(-1 - node) * ET_NODE_WORDS + 2) * SEARCH_BLOCK];
                        c = (int)(o ? (w2 >> 16) : (w2 & 0xffffu));
                    } else {
                        c = tree[(size_t)node * W + D.child_off + 2 * 2 + o];
                        c = c < 0 ? 0 : c;
                    }
                    if (c != 0) { node = c; --dtg; }
                    else {  // expand: new leaf, then rollout(depth_to_go - 1)
                        const int nn = min(n_nodes, D.max_nodes - 1);
                        ++n_nodes;
                        if (node == 0) {
                            uint32_t* nd = l1 + (size_t)o * ET_NODE_WORDS * SEARCH_BLOCK;
#pragma unroll
                            for (int k = 0; k < ET_NODE_WORDS; ++k) nd[k * SEARCH_BLOCK] = 0;
                            l1_exists |= 1u << o;
                        } else {
                            node_init(D, tree + (size_t)nn * W, P.A, P.O);
                            if (node < 0) {
                                uint32_t* w2 = l1 + ((size_t)(-1 - node) * ET_NODE_WORDS + 2) * SEARCH_BLOCK;
                                *w2 = o ? ((*w2 & 0xffffu) | ((uint32_t)nn << 16)) : ((*w2 & 0xffff0000u) | (uint32_t)nn);
                            } else tree[(size_t)node * W + D.child_off + 2 * 2 + o] = nn;
                        }
                        mode = 2; rdepth = dtg - 1; rret = 0; rdisc = 1;
                        if (rdepth == 0) finish = true;
                    }
                }
            } else if (mode == 1) {  // traverseChanceNode
                path_r[(size_t)plen * SEARCH_BLOCK]  = (float)r;
                path_na[(size_t)plen * SEARCH_BLOCK] = (node << 5) | a  /* a < FBA_MAX_ACTIONS <= 32 */;
                ++plen;
                if (term) finish = true;
                else {
                    const bool at_root_lds = root_lds && node == 0;
                    const int c = at_root_lds ? rootch[(a * P.O + o) * SEARCH_BLOCK] : child_get(P, D, tree, tab, epoch, node, a, o);
                    if (c >= 0) { node = c; --dtg; }
                    else {  // expand: new leaf, then rollout(depth_to_go - 1)
                        // never write past the slot's records: beyond the host's node bound (fba_engine.hip; cannot happen
                        // while that bound holds) the last record is reused and the overflow reported after the loop.  (A
                        // `break` here instead cost the whole kernel 45 %: the loop lost its shape.)
                        const int nn = min(n_nodes, D.max_nodes - 1);
                        ++n_nodes;
                        node_init(D, tree + (size_t)nn * W, P.A, P.O);
                        if (at_root_lds) rootch[(a * P.O + o) * SEARCH_BLOCK] = (int16_t)nn;
                        else child_set(P, D, tree, tab, epoch, node, a, o, nn);
                        mode = 2; rdepth = dtg - 1; rret = 0; rdisc = 1;
                        if (rdepth == 0) finish = true;
                    }
                }
            } else {
                rret += r * rdisc;
                rdisc *= P.gamma;
                --rdepth;
                if (rdepth == 0 || term) { delayed = rret; finish = true; }
            }
        }
        PROF_MARK(3)
        if (finish) {
            // back-up, leaf to root: ret = r + gamma * delayed; ChanceNode::addVisit(ret)
            // (MCTSTreeNodes.cpp:8-12); ActionNode::addVisit() (:59-62)
            // (the root is entry 0 of every path and no other entry: the loop runs over the nodes below it, the root's
            // register copy is updated once behind it -- one code path per level instead of two)
            double del = delayed;
            for (int k = plen - 1; k >= (ETIGER ? 4 : 1); --k) {
                const int na     = ETIGER ? (int)path16[(size_t)k * SEARCH_BLOCK] : path_na[(size_t)k * SEARCH_BLOCK];
                const double ret = (ETIGER ? et_reward(na >> 14) : (double)path_r[(size_t)k * SEARCH_BLOCK]) + P.gamma * del;
                const int act    = ETIGER ? ((na >> 12) & 3) : (na & 31);
                int32_t* rec = tree + (size_t)(ETIGER ? (na & 0xfff) : (na >> 5)) * W;
                double* q    = reinterpret_cast<double*>(rec + D.cq_off) + act;
                int n;
                if (P.A == 3) {  // {visits, n0, n1, n2} is one 16-byte word: one load, one store
                    int4* hp = reinterpret_cast<int4*>(rec);
                    int4 h   = *hp;
                    n = act == 0 ? ++h.y : (act == 1 ? ++h.z : ++h.w);
                    ++h.x;
                    *hp = h;
                } else {
                    n = ++rec[D.cn_off + act];
                    if (D.cn_off) ++rec[0];
                }
                *q += (ret - *q) / (double)n;
                del = ret;
            }
            if (ETIGER) {   // levels 3 and 2: header and Q from the descent, two stores each, no load
#pragma unroll
                for (int lv = 3; lv >= 2; --lv)
                    if (plen > lv) {
                        const int na     = (int)path16[(size_t)lv * SEARCH_BLOCK];
                        const double ret = et_reward(na >> 14) + P.gamma * del;
                        const int act    = (na >> 12) & 3;
                        int32_t* rec     = tree + (size_t)(na & 0xfff) * W;
                        int4 h           = lv == 3 ? cy_h3 : cy_h2;
                        const double q0  = lv == 3 ? cy_q3 : cy_q2;
                        const int n      = act == 0 ? ++h.y : (act == 1 ? ++h.z : ++h.w);
                        ++h.x;
                        *reinterpret_cast<int4*>(rec) = h;
                        reinterpret_cast<double*>(rec + D.cq_off)[act] = q0 + (ret - q0) / (double)n;
                        del = ret;
                    }
            }
            if (ETIGER && plen > 1) {   // the node below the root: in LDS
                const int na     = (int)path16[1 * SEARCH_BLOCK];
                const double ret = et_reward(na >> 14) + P.gamma * del;
                const int act    = (na >> 12) & 3;
                uint32_t* nd     = l1 + (size_t)(na & 0xfff) * ET_NODE_WORDS * SEARCH_BLOCK;
                uint32_t* cw     = nd + (act >> 1) * SEARCH_BLOCK;   // n0 | n1 << 16, n2
                const uint32_t w = *cw;
                const int n      = (int)((act & 1) ? (w >> 16) : (w & 0xffffu)) + 1;
                *cw              = (act & 1) ? ((w & 0xffffu) | ((uint32_t)n << 16)) : ((w & 0xffff0000u) | (uint32_t)n);
                uint32_t* qw     = nd + (size_t)(3 + 2 * act) * SEARCH_BLOCK;
                double q         = __hiloint2double((int)qw[SEARCH_BLOCK], (int)qw[0]);
                q += (ret - q) / (double)n;
                qw[0]            = (uint32_t)__double2loint(q);
                qw[SEARCH_BLOCK] = (uint32_t)__double2hiint(q);
                del = ret;
            }
            if (plen > 0) {
                const double ret = (ETIGER ? et_reward((int)path16[0] >> 14) : (double)path_r[0]) + P.gamma * del;
                const int act    = ETIGER ? (((int)path16[0] >> 12) & 3) : (path_na[0] & 31);
                // register-array element `act`: select, ONE division, write back
                int n = 0;
                double q = 0.0;
#pragma unroll
                for (int a2 = 0; a2 < AMAX; ++a2)
                    if (a2 == act) { n = r_cn[a2]; q = r_cq[a2]; }
                ++n;
                q += (ret - q) / (double)n;
#pragma unroll
                for (int a2 = 0; a2 < AMAX; ++a2)
                    if (a2 == act) { r_cn[a2] = n; r_cq[a2] = q; }
                ++r_vis;
            }
            ++sim;
            mode = 0;
        }
        PROF_MARK(4)
#ifdef FBA_PROFILE_SEARCH
        prof_[5] += 1;
#endif
    }
#ifdef FBA_PROFILE_SEARCH
    if (lane == 0)
        for (int r = 0; r < 8; ++r) atomicAdd(&g_search_prof[r], (unsigned long long)prof_[r]);
#endif
    g.stream(FBA_PHASE_SEARCH, (uint32_t)P.sims + 1u);
    if (n_nodes > D.max_nodes) atomicCAS(D.fault, 0, -(1 + e));  // -> FBA_ESTATE on the host
    const int best = ucb_pick<AMAX>(P, g, D.log1p_tab, 0, r_cn, r_cq, false);
    D.action[e]    = best;
    D.sim_steps[e] += steps;
    fba_trace_rec& rec = D.cur[e];
    rec.n_nodes    = n_nodes;
    rec.tree_depth = tree_depth;
#pragma unroll
    for (int a = 0; a < FBA_MAX_ACTIONS; ++a) {
        rec.root_n[a] = a < AMAX && a < P.A ? r_cn[a < AMAX ? a : 0] : 0;
        rec.root_q[a] = a < AMAX && a < P.A ? r_cq[a < AMAX ? a : 0] : 0.0;
    }
}

// ---------------------------------------------------------------------------------------------
// search_hist_kernel: the same search (POUCT / RBAPOUCT::selectAction, state machine of search_kernel) for
// history particles (fba_device.h) -- the gridworld FBA-POMDP of BASELINE configs[3].  A tree of 65 536
// simulations costs megabytes of HBM, so a GPU holds tens of thousands of them, not the hundreds of thousands that
// "one lane = one tree" needs to keep its SIMDs busy.  Here FOUR lanes share a tree: every lane carries the whole
// search state and executes the tree logic redundantly (same addresses, same values: loads coalesce, stores
// agree), and the simulator step -- where the time goes -- is split: lane q generates Philox block q of the
// quad's eight draws, lane f < 3 scans the history for, fetches and samples the Dirichlet row of state /
// observation feature f (gridworld_hist_step_quad).  Sixteen trees per wave, four times the waves per tree count;
// same draws, same order, same results as search_kernel on dense records.
// LDS per wave: path [depth][16], the root particle's record [2 + entries][16].
// ---------------------------------------------------------------------------------------------
constexpr int HIST_TREES = SEARCH_BLOCK / HIST_QUAD;
template <int K>
__global__ void __launch_bounds__(SEARCH_BLOCK) search_hist_kernel(Problem P, DeviceState D)
{
    constexpr int AMAX = 4;
    P.model = FBA_MODEL_BA_FACTORED; P.domain = FBA_DOM_GRIDWORLD; P.A = 4; P.belief = FBA_BELIEF_IMPORTANCE;
    extern __shared__ double lds[];
    const int lane = threadIdx.x, tl = lane >> 2;
    const int e    = blockIdx.x * HIST_TREES + tl;
    if (e >= P.E || !D.active[e]) return;  // (a quad leaves together)

    const int depth_cap = P.max_depth > 0 ? P.max_depth : 1;
    // the chosen action's count and Q of every node on the path as the descent loaded them: the back-up needs no load (a trip to
    // memory per level, twenty in a row at t = 0), only its two stores -- nothing else writes this tree
    double* path_q   = lds + tl;                                                                                  // [depth][trees]
    int32_t* path_n  = reinterpret_cast<int32_t*>(path_q - tl + (size_t)depth_cap * HIST_TREES) + tl;
    float* path_r    = reinterpret_cast<float*>(path_n - tl + (size_t)depth_cap * HIST_TREES) + tl;
    int32_t* path_na = reinterpret_cast<int32_t*>(path_r - tl + (size_t)depth_cap * HIST_TREES) + tl;
    uint32_t* stage  = reinterpret_cast<uint32_t*>(path_na - tl + (size_t)depth_cap * HIST_TREES) + tl;           // [Cs][trees]
    const bool carry = D.cn_off == 0;   // (the 48-byte record of hashed trees with four actions: {n0..n3}, {q0..q3}; others reload)
    // Child speculation.  A tree level costs two dependent trips to memory: the node's statistics, then -- once the step has produced
    // the observation -- the hash probe for the child.  The upper 16 bits of a node's count words (a node below the root has fewer than
    // 65 536 visits) remember the child each action led to last time; its statistics are requested right after the action is chosen,
    // so they are on their way while the step is computed and the probe answers.  The probe still decides: a wrong guess costs a
    // wasted request, never a result.  (child index = hint + 2; hint + 2 == the node itself = none; the root's hints live in registers.)
    const bool spec = carry && D.max_nodes <= 65538 && P.sims <= 65536;
    int pf_node = -1;
    int4 pf_h = make_int4(0, 0, 0, 0);
    double2 pf_q01 = make_double2(0, 0), pf_q23 = make_double2(0, 0);
    int r_hint[AMAX] = {0, 0, 0, 0};   // the root's: child index, 0 = none

    QuadRng g;
    g.init(P.seed_lo, P.seed_hi, (uint32_t)D.run[e], (uint32_t)D.episode[e], (uint32_t)D.t[e], lane);
    const int hist_len  = D.t[e];
    const int max_tree_depth = min(P.horizon - hist_len, P.max_depth);
    const int W         = D.node_words;
    int32_t* tree       = D.nodes + (size_t)e * D.max_nodes * W;
    const float* prec   = D.p_rec + rec_base(P, D, e, D.bufsel[e]) * (size_t)P.Cs;
    const uint32_t hist_cnt = D.hist_cnt[e];  // entries of each action in every record of this slot, and where each group starts
    const int hist_n        = hist_total(hist_cnt);
    const uint32_t hist_off = (uint32_t)hist_offset(hist_cnt, 1) << 8 | (uint32_t)hist_offset(hist_cnt, 2) << 16 | (uint32_t)hist_offset(hist_cnt, 3) << 24;

    if (P.planner == FBA_PLANNER_RANDOM) {  // RandomPlanner::selectAction RandomPlanner.cpp:14-24
        g.stream(FBA_PHASE_SEARCH, (uint32_t)P.sims);
        g.ensure(2);
        (void)g.u01();                      // the belief sample: GridWorld::generateRandomAction does not look at the state
        D.action[e] = g.slow_int(0, 4);
        if (P.search_budget > 0) D.search_done[e] = 1;
        return;
    }
    // budgeted launches (Problem::search_budget): a search parked by an earlier launch is taken up where it stopped -- its tree and
    // hash table are where it left them, the root's statistics come back from the root's record
    const int budget  = P.search_budget;
    int sim           = budget > 0 ? D.s_sim[e] : 0;
    const bool resume = sim > 0;
    int n_nodes = 1, tree_depth = 0;
    unsigned long long steps = 0;
    int r_vis = 0, r_cn[AMAX];
    double r_cq[AMAX];
#pragma unroll
    for (int a = 0; a < AMAX; ++a) { r_cn[a] = 0; r_cq[a] = 0.0; }
    int4* tab = hash_table(D, e);
    uint32_t epoch;
    if (!resume) {
        node_init(D, tree, P.A, P.O);
        epoch = hash_begin_search(D, e, tab, g.q, HIST_QUAD);
    } else {
        n_nodes    = D.s_nodes[e];
        tree_depth = D.s_depth[e];
        epoch      = D.epoch[e];
        const double* rq = reinterpret_cast<const double*>(tree + D.cq_off);
#pragma unroll
        for (int a = 0; a < AMAX; ++a) { r_cn[a] = tree[D.cn_off + a]; r_cq[a] = rq[a]; r_vis += r_cn[a]; }
    }
    int ts_src = -1;
    if (P.planner == FBA_PLANNER_TS) {  // TSPlanner / BATSPlanner: one belief sample, then the search from that particle
        g.stream(FBA_PHASE_SEARCH, (uint32_t)P.sims + 2u);
        g.ensure(1);
        ts_src = uniform_weight_pick(D.uni_scan, P.N, g.u01() * D.uni_total, D.uni_total);
    }
    int mode = 0, iter = 0;  // 0 = start a simulation, 1 = in the tree, 2 = rollout
    int node = 0, dtg = 0, plen = 0, rdepth = 0;
    uint32_t sp = 0, hist_mask = 0;
    double rret = 0, rdisc = 1;
#ifdef FBA_PROFILE_SEARCH
    long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prev_ = clock64();
#endif
    while (true) {
        PROF_MARK(6)
        if (mode == 0) {
            if (sim >= P.sims) break;
            if (budget > 0 && iter >= budget) break;   // out of iterations at a simulation boundary: park the search (below)
            g.stream(FBA_PHASE_SEARCH, (uint32_t)sim);
            g.ensure(8);  // the root sample, the first action, six rows
            const int src = ts_src >= 0 ? ts_src : uniform_weight_pick(D.uni_scan, P.N, g.u01() * D.uni_total, D.uni_total);
            const uint4* rp = reinterpret_cast<const uint4*>(prec + (size_t)src * hist_stride(P, hist_n));
            const int n4 = (hist_n + 5) >> 2;  // state, structure bits, entries
            for (int k0 = g.q; k0 < n4; k0 += 4 * HIST_QUAD) {   // (four loads in flight per lane, as in search_kernel)
                uint4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = rp[min(k0 + q * HIST_QUAD, n4 - 1)];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k = k0 + q * HIST_QUAD;
                    if (k < n4) {
                        stage[(4 * k + 0) * HIST_TREES] = v[q].x;
                        stage[(4 * k + 1) * HIST_TREES] = v[q].y;
                        stage[(4 * k + 2) * HIST_TREES] = v[q].z;
                        stage[(4 * k + 3) * HIST_TREES] = v[q].w;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the other lanes' pieces (LDS operations of one wave complete in order)
            hist_mask = stage[1 * HIST_TREES];
            sp        = (hist_mask >> 16) & 0x3ffu;
            node = 0; dtg = max_tree_depth; plen = 0; mode = 1;
        }
        PROF_MARK(0)
        bool finish = false, do_step = true;
        double delayed = 0;
        int a = 0;
        if (mode == 1 && dtg == 0) { finish = true; do_step = false; }
        if (mode == 1) tree_depth = max(tree_depth, max_tree_depth - dtg);
        int o = 0;
        double r = 0;
        bool term = false;
        if (do_step) {
            g.ensure(7);  // the action, six rows
            if (mode == 1) {  // traverseActionNode
                int guess = 0;   // spec: the child this (node, action) led to last time, 0 = none
                if (node == 0) {
                    a = ucb_pick<AMAX>(P, g, D.log1p_tab, r_vis, r_cn, r_cq, true);
                    if (spec) {
                        guess = a == 0 ? r_hint[0] : (a == 1 ? r_hint[1] : (a == 2 ? r_hint[2] : r_hint[3]));
                        if (guess >= min(n_nodes, D.max_nodes)) guess = 0;
                    }
                } else if (carry) {
                    const int32_t* rec = tree + (size_t)node * W;
                    int4 h;
                    double2 q01, q23;
                    if (spec && node == pf_node) { h = pf_h; q01 = pf_q01; q23 = pf_q23; }   // requested a level ago
                    else {
                        h   = *reinterpret_cast<const int4*>(rec);
                        q01 = *reinterpret_cast<const double2*>(rec + 4);
                        q23 = *reinterpret_cast<const double2*>(rec + 8);
                    }
                    const int m = spec ? 0xffff : -1;
                    const int cn[AMAX]    = {h.x & m, h.y & m, h.z & m, h.w & m};
                    const double cq[AMAX] = {q01.x, q01.y, q23.x, q23.y};
                    a = ucb_pick<AMAX>(P, g, D.log1p_tab, ((cn[0] + cn[1]) + cn[2]) + cn[3], cn, cq, true);
                    const int word = a == 0 ? h.x : (a == 1 ? h.y : (a == 2 ? h.z : h.w));
                    path_n[(size_t)plen * HIST_TREES] = word;   // (spec: count | hint << 16)
                    path_q[(size_t)plen * HIST_TREES] = a == 0 ? q01.x : (a == 1 ? q01.y : (a == 2 ? q23.x : q23.y));
                    if (spec) {
                        guess = (int)((uint32_t)word >> 16) + 2;
                        if (guess == node || guess >= min(n_nodes, D.max_nodes)) guess = 0;   // none (node 1's "itself" does not fit the encoding), or not a node
                    }
                } else a = ucb_select<AMAX>(P, D, g, tree + (size_t)node * W, true);
                if (spec && guess > 0) {   // the likely child's statistics, on their way while the step is computed
                    const int32_t* crec = tree + (size_t)guess * W;
                    pf_h    = *reinterpret_cast<const int4*>(crec);
                    pf_q01  = *reinterpret_cast<const double2*>(crec + 4);
                    pf_q23  = *reinterpret_cast<const double2*>(crec + 8);
                    pf_node = guess;
                } else pf_node = -1;
            } else {
                a = g.slow_int(0, 4);  // GridWorld::generateRandomAction :220-226
            }
#ifdef FBA_PROFILE_SEARCH
        }
        PROF_MARK(1)
        if (do_step) {
#endif
            term = gridworld_hist_step_quad<K, HIST_TREES>(P, g, stage + (size_t)(2 + ((hist_off >> (8 * a)) & 0xffu)) * HIST_TREES,
                                                           hist_count(hist_cnt, a), hist_mask, sp, a, o, r,
                                                           HistRowsGlobal{P.hist_base, P.hist_alt, P.hist_base + HistLayout(P.gw_N, P.gw_G, 4).obase0, HistLayout(P.gw_N, P.gw_G, 4)});
            ++steps;
#ifdef FBA_PROFILE_SEARCH
        }
        PROF_MARK(2)
        if (do_step) {
#endif
            if (mode == 1) {  // traverseChanceNode
                path_r[(size_t)plen * HIST_TREES]  = (float)r;
                path_na[(size_t)plen * HIST_TREES] = (node << 5) | a;
                ++plen;
                if (term) finish = true;
                else {
                    const int c = child_get(P, D, tree, tab, epoch, node, a, o);
                    const int nn = min(n_nodes, D.max_nodes - 1);   // (the node an expansion creates)
                    if (spec) {   // remember where (node, a) led: the root's hint in its register, a node's in the word the back-up stores
                        const int to = c >= 0 ? c : nn;
                        if (node == 0) {
#pragma unroll
                            for (int a2 = 0; a2 < AMAX; ++a2)
                                if (a2 == a) r_hint[a2] = to;
                        } else {
                            int32_t* pw = path_n + (size_t)(plen - 1) * HIST_TREES;
                            *pw = (*pw & 0xffff) | (int)((uint32_t)((to - 2) & 0xffff) << 16);
                        }
                    }
                    if (c >= 0) { node = c; --dtg; }
                    else {  // expand: new leaf, then rollout(depth_to_go - 1)
                        ++n_nodes;
                        if (spec) {   // counts 0, no child remembered (hint + 2 == the node itself), Q 0
                            int32_t* nrec = tree + (size_t)nn * W;
                            const int none = (int)((uint32_t)((nn - 2) & 0xffff) << 16);
                            *reinterpret_cast<int4*>(nrec)        = make_int4(none, none, none, none);
                            *reinterpret_cast<double2*>(nrec + 4) = make_double2(0.0, 0.0);
                            *reinterpret_cast<double2*>(nrec + 8) = make_double2(0.0, 0.0);
                        } else node_init(D, tree + (size_t)nn * W, P.A, P.O);
                        child_set(P, D, tree, tab, epoch, node, a, o, nn);
                        mode = 2; rdepth = dtg - 1; rret = 0; rdisc = 1;
                        if (rdepth == 0) finish = true;
                    }
                }
            } else {
                rret += r * rdisc;
                rdisc *= P.gamma;
                --rdepth;
                if (rdepth == 0 || term) { delayed = rret; finish = true; }
            }
        }
        PROF_MARK(3)
        if (finish) {  // back-up, leaf to root (MCTSTreeNodes.cpp:8-12, 59-62)
            double del = delayed;
            for (int k = plen - 1; k >= 0; --k) {
                const int na     = path_na[(size_t)k * HIST_TREES];
                const double ret = (double)path_r[(size_t)k * HIST_TREES] + P.gamma * del;
                const int act    = na & 31;
                if ((na >> 5) == 0) {
                    int n = 0;
                    double q = 0.0;
#pragma unroll
                    for (int a2 = 0; a2 < AMAX; ++a2)
                        if (a2 == act) { n = r_cn[a2]; q = r_cq[a2]; }
                    ++n;
                    q += (ret - q) / (double)n;
#pragma unroll
                    for (int a2 = 0; a2 < AMAX; ++a2)
                        if (a2 == act) { r_cn[a2] = n; r_cq[a2] = q; }
                    ++r_vis;
                } else if (carry) {
                    int32_t* rec    = tree + (size_t)(na >> 5) * W;
                    const int word  = path_n[(size_t)k * HIST_TREES];
                    const int n     = (spec ? (word & 0xffff) : word) + 1;
                    const double q0 = path_q[(size_t)k * HIST_TREES];
                    rec[act] = spec ? ((word & (int)0xffff0000) | n) : n;
                    reinterpret_cast<double*>(rec + D.cq_off)[act] = q0 + (ret - q0) / (double)n;
                } else {
                    int32_t* rec = tree + (size_t)(na >> 5) * W;
                    double* q    = reinterpret_cast<double*>(rec + D.cq_off) + act;
                    const int n  = ++rec[D.cn_off + act];
                    if (D.cn_off) ++rec[0];
                    *q += (ret - *q) / (double)n;
                }
                del = ret;
            }
            ++sim;
            mode = 0;
        }
        PROF_MARK(4)
        ++iter;
#ifdef FBA_PROFILE_SEARCH
        prof_[5] += 1;
#endif
    }
#ifdef FBA_PROFILE_SEARCH
    if (lane == 0)
        for (int r2 = 0; r2 < 8; ++r2) atomicAdd(&g_search_prof[r2], (unsigned long long)prof_[r2]);
#endif
    if (sim < P.sims) {   // parked: the four lanes of the quad hold the same values and store them to the same places
        double* rq = reinterpret_cast<double*>(tree + D.cq_off);
#pragma unroll
        for (int a = 0; a < AMAX; ++a) { tree[D.cn_off + a] = r_cn[a]; rq[a] = r_cq[a]; }
        if (D.cn_off) tree[0] = r_vis;
        D.s_sim[e]   = sim;
        D.s_nodes[e] = n_nodes;
        D.s_depth[e] = tree_depth;
        if (g.q == 0) D.sim_steps[e] += steps;
        return;   // (search_done[e] stays 0: env_kernel leaves the slot alone)
    }
    if (D.s_sim) D.s_sim[e] = 0;   // (also after a whole search run on a budgeted context, fba_select_action: what was parked for this slot is stale)
    if (budget > 0) D.search_done[e] = 1;
    g.stream(FBA_PHASE_SEARCH, (uint32_t)P.sims + 1u);
    g.ensure(1);
    if (n_nodes > D.max_nodes) atomicCAS(D.fault, 0, -(1 + e));
    const int best = ucb_pick<AMAX>(P, g, D.log1p_tab, 0, r_cn, r_cq, false);
    D.action[e]    = best;
    if (g.q == 0) D.sim_steps[e] += steps;
    fba_trace_rec& rec = D.cur[e];
    rec.n_nodes    = n_nodes;
    rec.tree_depth = tree_depth;
#pragma unroll
    for (int a = 0; a < FBA_MAX_ACTIONS; ++a) {
        rec.root_n[a] = a < AMAX ? r_cn[a < AMAX ? a : 0] : 0;
        rec.root_q[a] = a < AMAX ? r_cq[a < AMAX ? a : 0] : 0.0;
    }
}

// ---------------------------------------------------------------------------------------------
// search_hist2_kernel: the history-particle search on a tree whose node IS its hash bucket, with every trip to HBM asked for one
// loop iteration before it is needed.
//
// What bounded search_hist_kernel was the chain of dependent trips a tree level makes -- the node's
// statistics, the two Dirichlet-row fetches of the step, then the hash probe for the child -- at two waves per SIMD, with the sixteen
// trees of a wave in different phases, so that nearly every iteration of the wave waited for all of them.  Here:
//   * DeviceState::bkt: one open-addressing table per slot, 64-byte buckets, two to a 128-byte line (the unit the memory side moves:
//     profiles/r04_randline_counters.json).  A bucket is a node -- key = (parent bucket, action, observation) | epoch << 28, the four
//     action counts as uint16, the four Q values -- so the probe for the child of (node, a, o) returns the child's statistics: one trip
//     per level where there were two.  Three quarters of a gridworld tree's nodes are created and never reached again (measured on the
//     CPU restatement: 48 000 of 65 537 at the BASELINE size); they have no statistics to keep and live as 4-byte keys in the last 16 bytes
//     of the buckets (eight per line), found by the same one-line probe.  A node gets a bucket when it is reached a second time.
//     Probing is linear over lines; a probe ends at the first line with a free bucket (nodes) / a free key slot (keys), nothing is ever
//     deleted, the epoch in the key makes the table empty for the next search.
//   * The line of the child is requested right after the step that produced the observation and consumed at the top of the NEXT
//     iteration: the wave executes a whole iteration of the other trees' work in between.  The four lanes of a quad each load a quarter
//     of the line (lane q: piece q of both buckets; lane 3 the eight keys) and exchange what is needed by DPP.  The next simulation's
//     root particle is requested when the current one finishes, the same way.
//   * The observation tables of the prior (3.7 KB) sit in LDS: the second of a step's two dependent row fetches never leaves the CU.
//   * A filter of 2^k uniformly weighted particles is sampled in closed form (its prefix sums are exact multiples of 2^-k).
// Same streams, same draws, same arithmetic as search_hist_kernel: every trace field is bit-equal.
// ---------------------------------------------------------------------------------------------
constexpr int H2_PF = 3;   // 16-byte pieces per lane the prefetch registers hold (a root particle of up to 12 pieces = 46 entries; more are fetched in place)

template <int LANE>
__device__ __forceinline__ uint32_t quad_get(uint32_t v)   // the value lane LANE of this quad holds (all four lanes of a quad are always active together)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, LANE * 0x55, 0xf, 0xf, true);
}
template <int LANE>
__device__ __forceinline__ double quad_get_f64(uint32_t lo, uint32_t hi)
{
    return __hiloint2double((int)quad_get<LANE>(hi), (int)quad_get<LANE>(lo));
}
__device__ __forceinline__ uint32_t h2_home_line(uint32_t code, uint32_t nlines) { return __umulhi(child_hash((uint64_t)code), nlines); }
// (24-bit multiplies for the row numbers and a cheaper 32-bit mix here were each about 1 % SLOWER on C4, same box: profiles/r04_search_experiments.txt 11)

#ifndef FBA_HIST2_WAVES
#define FBA_HIST2_WAVES 3   // waves per SIMD the register allocation aims at: 3 = at most 168 VGPRs, which the kernel meets with two spilled registers since
                            // the root's statistics moved to LDS (with them in registers the cap spilled 14-17 and cost 24 %); 2 = 175, for A/B builds
#endif
// LROWS: every Dirichlet row of the prior comes from LDS (Problem::hist_lds: row ids + the distinct rows, shared by the H2_WAVES waves of a
// workgroup); otherwise the transition rows come from the padded tables in HBM and only the observation tables sit in LDS.
constexpr int H2_WAVES = 4;                                   // waves per workgroup
constexpr int H2_BLOCK = H2_WAVES * 64;
__host__ __device__ __forceinline__ size_t h2_shared_bytes(const Problem& P, bool lrows)
{
    const int K = P.hist_row <= 8 ? 8 : (P.hist_row <= 10 ? 12 : 16);
    return lrows ? (size_t)P.hist_rid_bytes + (size_t)P.hist_distinct * K * sizeof(float) : (size_t)4 * HistLayout(P.gw_N, P.gw_G, 4).ostride * sizeof(float);
}
__host__ __device__ __forceinline__ size_t h2_wave_bytes(const Problem& P)
{
    const int depth_cap = P.max_depth > 0 ? P.max_depth : 1;
    return (size_t)depth_cap * HIST_TREES * (sizeof(double) + sizeof(int32_t) + sizeof(float) + sizeof(int32_t)) +
           (size_t)(P.Cs > 2 * depth_cap ? P.Cs : 2 * depth_cap) * HIST_TREES * sizeof(float) + (size_t)14 * HIST_TREES * sizeof(float);   // (+ the root's statistics, the record geometry)
}
// search_order (fba_state.h): a counting sort of the slots by the depth their searches have left (inactive slots first).  One workgroup; which of two
// slots of equal depth comes first is left to the atomics -- every tree is independent, so no result can depend on it.
__global__ void __launch_bounds__(1024) search_order_kernel(Problem P, DeviceState D)
{
    __shared__ int32_t s_cnt[258];
    const int tid = threadIdx.x;
    for (int k = tid; k < 258; k += 1024) s_cnt[k] = 0;
    __syncthreads();
    auto key_of = [&](int e) { return D.active[e] ? 1 + min(max(min(P.horizon - D.t[e], P.max_depth), 0), 255) : 0; };
    for (int e = tid; e < P.E; e += 1024) atomicAdd(&s_cnt[key_of(e) + 1], 1);
    __syncthreads();
    if (tid == 0)
        for (int k = 1; k < 258; ++k) s_cnt[k] += s_cnt[k - 1];   // s_cnt[key] = first place of the key
    __syncthreads();
    for (int e = tid; e < P.E; e += 1024) D.search_order[atomicAdd(&s_cnt[key_of(e)], 1)] = e;
}

template <int K, bool LROWS>
__global__ void __launch_bounds__(H2_BLOCK) __attribute__((amdgpu_waves_per_eu(FBA_HIST2_WAVES, FBA_HIST2_WAVES))) search_hist2_kernel(Problem P, DeviceState D)
{
    constexpr int AMAX = 4;
    P.model = FBA_MODEL_BA_FACTORED; P.domain = FBA_DOM_GRIDWORLD; P.A = 4; P.belief = FBA_BELIEF_IMPORTANCE;
    extern __shared__ double lds_all[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tl = lane >> 2;
    const int depth_cap = P.max_depth > 0 ? P.max_depth : 1;
    const size_t shared_bytes = h2_shared_bytes(P, LROWS);
    double* lds = reinterpret_cast<double*>(reinterpret_cast<char*>(lds_all) + shared_bytes + (size_t)wave * h2_wave_bytes(P));   // this wave's paths and staging area
    double* path_q   = lds + tl;                                                                                  // [depth][trees]: the chosen action's Q as the descent saw it
    int32_t* path_n  = reinterpret_cast<int32_t*>(path_q - tl + (size_t)depth_cap * HIST_TREES) + tl;            // ... and its count
    float* path_r    = reinterpret_cast<float*>(path_n - tl + (size_t)depth_cap * HIST_TREES) + tl;
    int32_t* path_na = reinterpret_cast<int32_t*>(path_r - tl + (size_t)depth_cap * HIST_TREES) + tl;            // bucket << 5 | action
    uint32_t* stage  = reinterpret_cast<uint32_t*>(path_na - tl + (size_t)depth_cap * HIST_TREES) + tl;           // [Cs][trees]
                                                                                                                  // (stage also holds the back-up's returns: two words per level)
    // the root's statistics, [4 counts, 4 Q's][trees]: read where the root is selected at and where it is backed up, i.e. twice per simulation -- in
    // registers they were thirteen of the kernel's live values for the whole search
    double* root_q  = reinterpret_cast<double*>(stage - tl + (size_t)max(P.Cs, 2 * depth_cap) * HIST_TREES) + tl;      // [4][trees]
    int32_t* root_n = reinterpret_cast<int32_t*>(root_q - tl + (size_t)4 * HIST_TREES) + tl;                           // [4][trees]
    int32_t* rec_geo = root_n - tl + (size_t)4 * HIST_TREES + tl;   // [trees]: pieces of a record | pieces between records << 8 (read twice per simulation)
    const HistLayout HL(P.gw_N, P.gw_G, 4);
    {   // the workgroup's shared tables: every thread, before any quad leaves
        const uint4* src = LROWS ? reinterpret_cast<const uint4*>(P.hist_lds) : reinterpret_cast<const uint4*>(P.hist_base + HL.obase0);
        uint4* dst       = reinterpret_cast<uint4*>(lds_all);
        for (int i = threadIdx.x; i < (int)(shared_bytes / 16); i += (int)blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    const uint8_t* s_rid = reinterpret_cast<const uint8_t*>(lds_all);
    const float* s_rows  = reinterpret_cast<const float*>(reinterpret_cast<const char*>(lds_all) + (LROWS ? P.hist_rid_bytes : 0));
    const int tree_k = (blockIdx.x * (int)(blockDim.x >> 6) + wave) * HIST_TREES + tl;   // (the launcher picks 4, 2 or 1 waves per workgroup: what 64 KB of LDS hold)
    if (tree_k >= P.E) return;
    const int e = D.ab_lockstep ? D.search_order[tree_k] : tree_k;   // (lock-step waves: slots dealt to waves by the depth their searches have left)
    if (!D.active[e]) return;  // (a quad leaves together)

    QuadRng g;
    g.init(P.seed_lo, P.seed_hi, (uint32_t)D.run[e], (uint32_t)D.episode[e], (uint32_t)D.t[e], lane);
    const int hist_len  = D.t[e];
    const int max_tree_depth = min(P.horizon - hist_len, P.max_depth);
    const float* prec   = D.p_rec + rec_base(P, D, e, D.bufsel[e]) * (size_t)P.Cs;
    const uint32_t hist_cnt = D.hist_cnt[e];
    // 16-byte pieces of a record (state, structure bits, entries) and, above them, the 16-byte pieces between this slot's records (hist_stride)
    *rec_geo = ((hist_total(hist_cnt) + 5) >> 2) | (hist_stride(P, hist_total(hist_cnt)) >> 2) << 8;
    const bool uni_exact    = (P.N & (P.N - 1)) == 0 && D.uni_total == 1.0;   // N = 2^k: the prefix sums of the weights 1/N are the exact values (i + 1) / N

    if (P.planner == FBA_PLANNER_RANDOM) {  // RandomPlanner::selectAction RandomPlanner.cpp:14-24
        g.stream(FBA_PHASE_SEARCH, (uint32_t)P.sims);
        g.ensure(2);
        (void)g.u01();                      // the belief sample: GridWorld::generateRandomAction does not look at the state
        D.action[e] = g.slow_int(0, 4);
        if (P.search_budget > 0) D.search_done[e] = 1;
        return;
    }
    const uint32_t nlines = (uint32_t)D.bkt_lines;
    uint4* tab            = D.bkt + (size_t)e * nlines * 8;
    uint32_t* tabw        = reinterpret_cast<uint32_t*>(tab);
    const int ROOT        = (int)(nlines * 2u);   // the root has no bucket: its statistics live in registers
    const int budget      = P.search_budget;
    int sim               = budget > 0 ? D.s_sim[e] : 0;
    const bool resume     = sim > 0;
    int n_nodes = 1, tree_depth = 0;
    uint32_t steps = 0;   // (of this launch: at most sims x horizon)
#pragma unroll
    for (int a = 0; a < AMAX; ++a) { root_n[a * HIST_TREES] = 0; root_q[a * HIST_TREES] = 0.0; }
    uint32_t epoch;
    if (!resume) {
        epoch = D.epoch[e] + 1;
        if (epoch > 15u) {   // the four bits of the key are used up: empty the table (once in fifteen searches)
            for (uint32_t k = (uint32_t)g.q; k < nlines * 8u; k += HIST_QUAD) tab[k] = make_uint4(0, 0, 0, 0);
            epoch = 1;
        }
        D.epoch[e] = epoch;
    } else {
        n_nodes    = D.s_nodes[e];
        tree_depth = D.s_depth[e];
        epoch      = D.epoch[e];
        const int32_t* rn = reinterpret_cast<const int32_t*>(D.s_root + (size_t)e * 6);
        const double* rq  = D.s_root + (size_t)e * 6 + 2;
#pragma unroll
        for (int a = 0; a < AMAX; ++a) { root_n[a * HIST_TREES] = rn[a]; root_q[a * HIST_TREES] = rq[a]; }
    }
    const uint32_t ekey = epoch << 28;
    int ts_src = -1;
    if (P.planner == FBA_PLANNER_TS) {  // TSPlanner / BATSPlanner: one belief sample, then the search from that particle
        g.stream(FBA_PHASE_SEARCH, (uint32_t)P.sims + 2u);
        g.ensure(1);
        ts_src = uniform_weight_pick(D.uni_scan, P.N, g.u01() * D.uni_total, D.uni_total);
    }
    int mode = 0, iter = 0;  // 0 = start a simulation, 1 = in the tree, 2 = rollout
    int node = ROOT, dtg = 0, plen = 0, rdepth = 0, cur_src = 0;
    uint32_t sp = 0, hist_mask = 0;
    double rret = 0, rdisc = 1;
    bool have_particle = false, pend = false, broken = false;
    uint32_t pk = 0, pline = 0;          // the child being looked up: its key, its home line
    uint4 pf[H2_PF];                     // this lane's share of what was requested an iteration ago: a line of the table, or a root particle
#pragma unroll
    for (int j = 0; j < H2_PF; ++j) pf[j] = make_uint4(0, 0, 0, 0);

    // the root particle of simulation `sim`: Belief::sample() on its stream, and this lane's pieces of the record on their way
    // (its stream is set and eight draws -- the root sample, the first action, six rows -- are ensured by the caller)
    auto request_particle = [&]() {
        if (ts_src >= 0) cur_src = ts_src;
        else {
            const double u = g.u01();
            if (uni_exact) cur_src = max((int)ceil(u * (double)P.N) - 1, 0);   // the largest i with i / N < u (WeightedFilter.cpp:163-191 on exact prefix sums)
            else cur_src = uniform_weight_pick(D.uni_scan, P.N, u * D.uni_total, D.uni_total);
        }
        const int n4s = *rec_geo;
        const uint4* rp = reinterpret_cast<const uint4*>(prec) + __umul24((uint32_t)cur_src, (uint32_t)n4s >> 8);   // (N * Cs / 4 < 2^24)
#pragma unroll
        for (int j = 0; j < H2_PF; ++j) pf[j] = rp[min(g.q + HIST_QUAD * j, (n4s & 0xff) - 1)];
    };

#ifdef FBA_PROFILE_SEARCH
    long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prev_ = clock64();
#endif
    while (true) {
        PROF_MARK(6)
        int cn[AMAX] = {0, 0, 0, 0};          // the current node's statistics (below the root), set where the node is entered
        double cq[AMAX] = {0.0, 0.0, 0.0, 0.0};
        bool finish = false, do_step = true;
        double delayed = 0;
        if (D.ab_lockstep) {
            // Lock-step waves (FBA_HIST_LOCKSTEP=0 turns them off): the wave's trees start their simulations together -- a tree whose simulation is over
            // waits in mode 3 until every tree of the wave that is still searching is there too.  Sixteen trees in sixteen phases made the wave
            // execute the selection, the particle's consumption and the back-up in nearly every iteration for one or two trees each; in step, the
            // selection runs in the iterations of the descent only, the other two once per simulation.  What the waiting costs is small because the
            // wave's slots have the same depth left (search_order: a rollout runs to the horizon), so their simulations have nearly the same length.
            // Every tree still runs the same simulations in the same order: no result changes.
            const bool waiting = mode == 3;
            if (__builtin_amdgcn_ballot_w64(waiting) == __builtin_amdgcn_ballot_w64(true)) mode = 0;
            else if (waiting) do_step = false;
        }
        if (mode == 0) {
            if (sim >= P.sims) break;
            if (budget > 0 && iter >= budget) break;   // out of iterations at a simulation boundary: park the search (below)
            if (!have_particle) {   // the launch's first simulation (later ones are asked for when their predecessor finishes)
                g.stream(FBA_PHASE_SEARCH, (uint32_t)sim);
                g.ensure(8);
                request_particle();
            }
            have_particle = false;
            const int n4s = *rec_geo;
#pragma unroll
            for (int j = 0; j < H2_PF; ++j) {
                const int k = g.q + HIST_QUAD * j;
                if (k < (n4s & 0xff)) {
                    stage[(4 * k + 0) * HIST_TREES] = pf[j].x;
                    stage[(4 * k + 1) * HIST_TREES] = pf[j].y;
                    stage[(4 * k + 2) * HIST_TREES] = pf[j].z;
                    stage[(4 * k + 3) * HIST_TREES] = pf[j].w;
                }
            }
            if ((n4s & 0xff) > HIST_QUAD * H2_PF) {   // (records of more than 46 entries: the rest in place)
                const uint4* rp = reinterpret_cast<const uint4*>(prec) + __umul24((uint32_t)cur_src, (uint32_t)n4s >> 8);   // (N * Cs / 4 < 2^24)
                for (int k = g.q + HIST_QUAD * H2_PF; k < (n4s & 0xff); k += HIST_QUAD) {
                    const uint4 v = rp[k];
                    stage[(4 * k + 0) * HIST_TREES] = v.x;
                    stage[(4 * k + 1) * HIST_TREES] = v.y;
                    stage[(4 * k + 2) * HIST_TREES] = v.z;
                    stage[(4 * k + 3) * HIST_TREES] = v.w;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the other lanes' pieces (LDS operations of one wave complete in order)
            hist_mask = stage[1 * HIST_TREES];
            sp        = (hist_mask >> 16) & 0x3ffu;
            node = ROOT; dtg = max_tree_depth; plen = 0; mode = 1; pend = false;
        } else if (mode == 1 && pend) {
            // traverseChanceNode's child lookup (POUCT.cpp:224-246), answered by the line requested an iteration ago
            pend = false;
            const uint32_t k0 = quad_get<0>(pf[0].x), k1 = quad_get<0>(pf[1].x);
            const bool v0 = (k0 >> 28) == epoch, v1 = (k1 >> 28) == epoch;
            uint32_t lm = 0, lf = 8;   // lane 3 holds the line's eight keys: is pk among them, and the first free place
            {
                const uint32_t w[8] = {pf[0].x, pf[0].y, pf[0].z, pf[0].w, pf[1].x, pf[1].y, pf[1].z, pf[1].w};
#pragma unroll
                for (int j = 7; j >= 0; --j) {
                    lm |= (w[j] == pk) ? 1u : 0u;
                    lf = ((w[j] >> 28) != epoch) ? (uint32_t)j : lf;
                }
            }
            lm = quad_get<3>(lm);
            lf = quad_get<3>(lf);
            int hit = (k0 == pk) ? 0 : ((k1 == pk) ? 1 : -1);          // the child has a bucket in its home line
            int bucket = (int)(pline * 2u) + max(hit, 0);
            bool is_leaf = false, have_stats = hit >= 0;
            int free_bucket = (!v0) ? (int)(pline * 2u) : ((!v1) ? (int)(pline * 2u) + 1 : -1);
            int free_key    = lf < 8u ? (int)((pline * 2u + (lf >> 2)) * 16u + 12u + (lf & 3u)) : -1;   // word index in the table
            if (hit < 0) {
                const bool nodes_done = free_bucket >= 0;              // a free bucket ends the probe for a node
                const bool keys_done  = lm != 0u || free_key >= 0;     // a match or a free place ends the probe for a key
                is_leaf = lm != 0u;
                if (!(nodes_done && keys_done)) {
                    // the home line is full of other nodes or other keys: walk on, line by line (a few per cent of the lookups at load 1/2)
                    bool need_n = !nodes_done, need_k = !keys_done;
                    uint32_t line = pline;
                    for (uint32_t n = 1; n < nlines && (need_n || need_k); ++n) {
                        line = line + 1u == nlines ? 0u : line + 1u;
                        const uint4* lp = tab + (size_t)line * 8;
                        const uint32_t c0 = lp[0].x, c1 = lp[4].x;
                        const uint4 ka = lp[3], kb = lp[7];
                        if (need_n) {
                            if (c0 == pk || c1 == pk) { hit = c0 == pk ? 0 : 1; bucket = (int)(line * 2u) + hit; need_n = false; need_k = false; is_leaf = false; }
                            else if ((c0 >> 28) != epoch || (c1 >> 28) != epoch) { free_bucket = (int)(line * 2u) + ((c0 >> 28) != epoch ? 0 : 1); need_n = false; }
                        }
                        if (need_k) {
                            const uint32_t w[8] = {ka.x, ka.y, ka.z, ka.w, kb.x, kb.y, kb.z, kb.w};
                            int fj = 8;
                            bool m = false;
#pragma unroll
                            for (int j = 7; j >= 0; --j) {
                                m  = m || w[j] == pk;
                                fj = ((w[j] >> 28) != epoch) ? j : fj;
                            }
                            if (m) { is_leaf = true; need_k = false; }
                            else if (fj < 8) { free_key = (int)((line * 2u + ((uint32_t)fj >> 2)) * 16u + 12u + ((uint32_t)fj & 3u)); need_k = false; }
                        }
                    }
                    if (need_n || need_k) { atomicCAS(D.fault, 0, -(1 + e)); broken = true; }   // the table is full (DeviceState::bkt_lines too small for this tree)
                }
            }
            if (hit >= 0) {          // traverseActionNode of an existing node with statistics
                if (have_stats) {
                    const uint4 mine = hit ? pf[1] : pf[0];   // this lane's piece of the bucket
                    const uint32_t c01 = quad_get<0>(mine.y), c23 = quad_get<0>(mine.z);
                    cn[0] = (int)(c01 & 0xffffu); cn[1] = (int)(c01 >> 16); cn[2] = (int)(c23 & 0xffffu); cn[3] = (int)(c23 >> 16);
                    cq[0] = quad_get_f64<1>(mine.x, mine.y); cq[1] = quad_get_f64<1>(mine.z, mine.w);
                    cq[2] = quad_get_f64<2>(mine.x, mine.y); cq[3] = quad_get_f64<2>(mine.z, mine.w);
                } else {             // found further down the probe sequence: fetch it now
                    const uint4* bp   = tab + (size_t)bucket * 4;
                    const uint4 h     = bp[0];
                    const double2 q01 = *reinterpret_cast<const double2*>(bp + 1);
                    const double2 q23 = *reinterpret_cast<const double2*>(bp + 2);
                    cn[0] = (int)(h.y & 0xffffu); cn[1] = (int)(h.y >> 16); cn[2] = (int)(h.z & 0xffffu); cn[3] = (int)(h.z >> 16);
                    cq[0] = q01.x; cq[1] = q01.y; cq[2] = q23.x; cq[3] = q23.y;
                }
                node = bucket; --dtg;
            } else if (broken) {
                finish = true; do_step = false; plen = 0; sim = P.sims;   // (the host reports the fault; leave the loop)
            } else if (is_leaf) {    // a node that exists and was never reached again: all its statistics are zero; it gets a bucket now
                if (free_bucket < 0) { atomicCAS(D.fault, 0, -(1 + e)); broken = true; finish = true; do_step = false; plen = 0; sim = P.sims; }
                else {
                    uint4* bp = tab + (size_t)free_bucket * 4;
                    if (g.q < 3) bp[g.q] = make_uint4(g.q == 0 ? pk : 0u, 0u, 0u, 0u);   // key + counts, Q's; the bucket's four keys stay
                    node = free_bucket; --dtg;
                }
            } else {                 // no such child: create it, then rollout(depth_to_go - 1)  (POUCT.cpp:236-244)
                if (free_key < 0) { atomicCAS(D.fault, 0, -(1 + e)); broken = true; finish = true; do_step = false; plen = 0; sim = P.sims; }
                else {
                    tabw[free_key] = pk;
                    ++n_nodes;
                    mode = 2; rdepth = dtg - 1; rret = 0; rdisc = 1;
                    if (rdepth == 0) { finish = true; do_step = false; }
                }
            }
        }
        PROF_MARK(0)
        if (mode == 1 && dtg == 0 && !finish) { finish = true; do_step = false; }
        if (mode == 1) tree_depth = max(tree_depth, max_tree_depth - dtg);
        int a = 0, o = 0;
        double r = 0;
        bool term = false;
        if (do_step) {
            // (the action's draw and the six rows': ensured at the end of the previous iteration)
            if (mode == 1) {  // traverseActionNode
                // one selectChanceNodeUCB for the wave, on the root's registers or the bucket's values (two inlined copies would run one after the other)
                const bool at_root = node == ROOT;
                int vis = 0;
#pragma unroll
                for (int a2 = 0; a2 < AMAX; ++a2) {
                    const int rn2 = root_n[a2 * HIST_TREES];
                    const double rq2 = root_q[a2 * HIST_TREES];
                    cn[a2] = at_root ? rn2 : cn[a2];
                    cq[a2] = at_root ? rq2 : cq[a2];
                    vis += cn[a2];    // (ActionNode::_visit_count is the sum of its chance nodes' counts: every back-up through the node adds one to exactly one of them)
                }
                a = ucb_pick<AMAX>(P, g, D.log1p_tab, vis, cn, cq, true);   // (the root's visits are the sum of its counts too: MCTSTreeNodes.cpp:59-62)
                path_n[(size_t)plen * HIST_TREES] = a == 0 ? cn[0] : (a == 1 ? cn[1] : (a == 2 ? cn[2] : cn[3]));   // (unused at the root: its back-up works on the registers)
                path_q[(size_t)plen * HIST_TREES] = a == 0 ? cq[0] : (a == 1 ? cq[1] : (a == 2 ? cq[2] : cq[3]));
            } else {
                a = g.slow_int4();  // GridWorld::generateRandomAction :220-226 (slowRandomInt(0, 4))
            }
#ifdef FBA_PROFILE_SEARCH
        }
        PROF_MARK(1)
        if (do_step) {
#endif
            // BAPOMDP::step over BABNModel (BAPOMDP.cpp:111-143, BABNModel.cpp:292-325) as two passes of hist_row_pass.  Pass A: the transition rows
            // of (state, a) for every lane.  Pass B: the observation rows of (a, s') for the trees that are in their tree -- and, for the trees that are
            // in a rollout, the transition rows of the NEXT step: a rollout never looks at an observation (POUCT.cpp:273-303 uses reward and terminal
            // only), so its three observation draws are skipped and the pass the wave executes anyway carries a second simulated step.
            const int f = min(g.q, 2), NW = P.gw_N, GW = P.gw_G, nrow = f == 2 ? GW : NW;
            const HistRowsLds<K> rl{s_rid, s_rows, HistRowIds(P.gw_N, P.gw_G, 4)};
            const HistRowsGlobal rg{P.hist_base, P.hist_alt, s_rows, HL};
            int x = hist_x(sp), y = hist_y(sp), gl = hist_g(sp), cell = x * NW + y;
            const uint32_t* listA = stage + (size_t)(2 + hist_offset(hist_cnt, a)) * HIST_TREES;
            const int nA = hist_count(hist_cnt, a);
            int nv;
            {
                const bool mx = (hist_mask >> (2 * a)) & 1u, my = (hist_mask >> (2 * a + 1)) & 1u;
                const bool with_goal = f == 2 || ((hist_mask >> (2 * a + f)) & 1u);
                const float* rowp = LROWS ? rl.t(a, f, with_goal, cell, gl) : rg.t(a, f, with_goal, cell, gl);
                nv = hist_row_pass<K, HIST_TREES, LROWS>(P, g, listA, nA, sp, false, mx, my, rowp, nrow, f, u01_of(g.at(g.draw + (uint32_t)f)));
            }
            const int nx = quad_bcast(g.addr0, 0, nv), ny = quad_bcast(g.addr0, 1, nv), ng = quad_bcast(g.addr0, 2, nv);
            const bool found = gridworld_on_goal(P, cell, gl);  // GridWorldBAExtension.cpp:74-99: terminal and reward from the OLD state
            const uint32_t spN = hist_pack(nx, ny, ng);
            // what pass B is for this tree
            bool second = false;
            int aB = a, nB = nA;   // (pass B walks the entries of this action: the step's own, or the rollout's next)
            const uint32_t* listB = listA;
            const float* rowB;
            double uB;
            if (mode == 1) {
                const int nvf = f == 0 ? nx : (f == 1 ? ny : ng);
                rowB = LROWS ? rl.o(a, f, nvf) : rg.o(a, f, nvf);
                uB   = u01_of(g.at(g.draw + 3u + (uint32_t)f));
            } else {
                // the rollout's step ends here (its observation would be sampled from draws 3..5 of the step: skipped, never used)
                rret += (found ? 1.0 : 0.0) * rdisc;
                rdisc *= P.gamma;
                --rdepth;
                ++steps;
                g.draw += 6;
                sp = spN;
                rowB = LROWS ? rl.o(a, f, 0) : rg.o(a, f, 0);   // (any row: nothing is counted into it and its draw is dropped)
                uB   = 0.0;
                nB   = 0;
                if (rdepth == 0 || found) { delayed = rret; finish = true; }
                else {   // the next step of the rollout, in this iteration's second pass
                    second = true;
                    g.ensure(4);   // its action and its three transition rows
                    aB = g.slow_int4();
                    x = nx; y = ny; gl = ng; cell = x * NW + y;
                    const bool with_goal = f == 2 || ((hist_mask >> (2 * aB + f)) & 1u);
                    listB = stage + (size_t)(2 + hist_offset(hist_cnt, aB)) * HIST_TREES;
                    nB    = hist_count(hist_cnt, aB);
                    rowB = LROWS ? rl.t(aB, f, with_goal, cell, gl) : rg.t(aB, f, with_goal, cell, gl);
                    uB   = u01_of(g.at(g.draw + (uint32_t)f));
                }
            }
            const int nvB = hist_row_pass<K, HIST_TREES, LROWS>(P, g, listB, nB, spN, mode == 1, (hist_mask >> (2 * aB)) & 1u, (hist_mask >> (2 * aB + 1)) & 1u, rowB, nrow, f, uB);
            const int v0 = quad_bcast(g.addr0, 0, nvB), v1 = quad_bcast(g.addr0, 1, nvB), v2 = quad_bcast(g.addr0, 2, nvB);
            if (mode == 1) {  // traverseChanceNode
                o = (v0 * NW + v1) * GW + v2;
                r = found ? 1 : 0;
                term = found;
                g.draw += 6;
                ++steps;
                sp = spN;
                path_r[(size_t)plen * HIST_TREES]  = (float)r;
                path_na[(size_t)plen * HIST_TREES] = (node << 5) | a;
                ++plen;
                if (term) finish = true;
                else {   // ask for the child's line; it is looked at when this loop comes round again
                    const uint32_t code = ((uint32_t)node * 4u + (uint32_t)a) * (uint32_t)P.O + (uint32_t)o;
                    pk    = ekey | code;
                    pline = h2_home_line(code, nlines);
                    const uint4* lp = tab + (size_t)pline * 8;
                    pf[0] = lp[g.q];
                    pf[1] = lp[4 + g.q];
                    pend  = true;
                }
            } else if (second) {
                const bool found2 = gridworld_on_goal(P, cell, gl);
                rret += (found2 ? 1.0 : 0.0) * rdisc;
                rdisc *= P.gamma;
                --rdepth;
                ++steps;
                g.draw += 6;
                sp = hist_pack(v0, v1, v2);
                if (rdepth == 0 || found2) { delayed = rret; finish = true; }
            }
        }
        PROF_MARK(2)
        PROF_MARK(3)
        if (finish) {
            // back-up, leaf to root (MCTSTreeNodes.cpp:8-12, 59-62).  The returns chain down the path (ret = r + gamma * delayed: two operations
            // per level, every lane); the Q updates -- a division each -- do not depend on one another, so lane j of the quad takes level j
            // (4 + j, ...) and the quad does four at a time.  Count and Q of the chosen action are the descent's (path_n, path_q: nothing else
            // writes this tree), the root's included; level 0 is the root, lane 0 hands its new statistics to the quad's registers.
            double del = delayed;
            for (int k = plen - 1; k >= 0; --k) {
                const double ret = (double)path_r[(size_t)k * HIST_TREES] + P.gamma * del;
                stage[(size_t)(2 * k) * HIST_TREES]     = (uint32_t)__double2loint(ret);   // (the simulation is over: its staged particle is no longer read)
                stage[(size_t)(2 * k + 1) * HIST_TREES] = (uint32_t)__double2hiint(ret);
                del = ret;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (int k0 = 0; k0 < plen; k0 += HIST_QUAD) {
                const int k      = min(k0 + g.q, plen - 1);
                const int na     = path_na[(size_t)k * HIST_TREES];
                const double ret = __hiloint2double((int)stage[(size_t)(2 * k + 1) * HIST_TREES], (int)stage[(size_t)(2 * k) * HIST_TREES]);
                const int act    = na & 31;
                const int n      = path_n[(size_t)k * HIST_TREES] + 1;
                const double q0  = path_q[(size_t)k * HIST_TREES];
                const double qn  = q0 + (ret - q0) / (double)n;
                if (k0 + g.q < plen && (na >> 5) != ROOT) {
                    uint32_t* bw = tabw + (size_t)(na >> 5) * 16;
                    reinterpret_cast<uint16_t*>(bw + 1)[act] = (uint16_t)n;
                    reinterpret_cast<double*>(bw + 4)[act]   = qn;
                }
                if (k0 == 0) {   // the root's level is lane 0's
                    const int ract   = (int)quad_get<0>((uint32_t)act), rn = (int)quad_get<0>((uint32_t)n);
                    const double rq  = quad_get_f64<0>((uint32_t)__double2loint(qn), (uint32_t)__double2hiint(qn));
                    root_n[ract * HIST_TREES] = rn;   // (the quad's four lanes store the same values)
                    root_q[ract * HIST_TREES] = rq;
                }
            }
            if (!broken) ++sim;
            mode = D.ab_lockstep ? 3 : 0; pend = false;
        }
        {
            // The one place of the loop where Philox blocks are made: what the next iteration draws -- a step's seven (the action, six rows), or
            // a new simulation's eight on its own stream (the root sample first).  ensure() only prepares blocks: the draws are the same ones.
            const bool new_sim = finish && sim < P.sims;
            if (new_sim) g.stream(FBA_PHASE_SEARCH, (uint32_t)sim);
            g.ensure(new_sim ? 8 : 7);
            if (new_sim) { request_particle(); have_particle = true; }
        }
        PROF_MARK(4)
        ++iter;
#ifdef FBA_PROFILE_SEARCH
        prof_[5] += 1;
#endif
    }
#ifdef FBA_PROFILE_SEARCH
    if (lane == 0)
        for (int r2 = 0; r2 < 8; ++r2) atomicAdd(&g_search_prof[r2], (unsigned long long)prof_[r2]);
#endif
    if (sim < P.sims) {   // parked: the four lanes of the quad hold the same values and store them to the same places
        int32_t* rn = reinterpret_cast<int32_t*>(D.s_root + (size_t)e * 6);
        double* rq  = D.s_root + (size_t)e * 6 + 2;
#pragma unroll
        for (int a = 0; a < AMAX; ++a) { rn[a] = root_n[a * HIST_TREES]; rq[a] = root_q[a * HIST_TREES]; }
        D.s_sim[e]   = sim;
        D.s_nodes[e] = n_nodes;
        D.s_depth[e] = tree_depth;
        if (g.q == 0) D.sim_steps[e] += steps;
        return;   // (search_done[e] stays 0: env_kernel leaves the slot alone)
    }
    if (D.s_sim) D.s_sim[e] = 0;
    if (budget > 0) D.search_done[e] = 1;
    g.stream(FBA_PHASE_SEARCH, (uint32_t)P.sims + 1u);
    g.ensure(1);
    int r_cn[AMAX];
    double r_cq[AMAX];
#pragma unroll
    for (int a = 0; a < AMAX; ++a) { r_cn[a] = root_n[a * HIST_TREES]; r_cq[a] = root_q[a * HIST_TREES]; }
    const int best = ucb_pick<AMAX>(P, g, D.log1p_tab, 0, r_cn, r_cq, false);
    D.action[e]    = best;
    if (g.q == 0) D.sim_steps[e] += steps;
    fba_trace_rec& rec = D.cur[e];
    rec.n_nodes    = n_nodes;
    rec.tree_depth = tree_depth;
#pragma unroll
    for (int a = 0; a < FBA_MAX_ACTIONS; ++a) {
        rec.root_n[a] = a < AMAX ? r_cn[a < AMAX ? a : 0] : 0;
        rec.root_q[a] = a < AMAX ? r_cq[a < AMAX ? a : 0] : 0.0;
    }
}

// ---------------------------------------------------------------------------------------------
// host launcher
// ---------------------------------------------------------------------------------------------
void launch_search(const Problem& P, const DeviceState& D, hipStream_t st)
{
    const bool stage = P.model != FBA_MODEL_POMDP && P.Cs <= SEARCH_STAGE_WORDS;
    // the episodic tiger family on its own tree layout (ETIGER); FBA_NO_ETIGER=1 keeps the general layout, for A/B runs
    const bool no_etiger = D.ab_no_etiger != 0;
    // (the instantiations that exist on that layout: the tabular tiger BA-POMDP, planning on tiger itself, packed factored tiger)
    const bool tiger_table = P.model == FBA_MODEL_BA_TABLE && P.planner == FBA_PLANNER_POUCT && !P.dirichlet_regular && stage &&
                             (P.domain == FBA_DOM_TIGER_EPISODIC || P.domain == FBA_DOM_TIGER_CONTINUOUS) && !D.hash;
    const bool tiger_pomdp = P.model == FBA_MODEL_POMDP && P.planner == FBA_PLANNER_POUCT && !D.hash &&
                             (P.domain == FBA_DOM_TIGER_EPISODIC || P.domain == FBA_DOM_TIGER_CONTINUOUS);
    const int ft_fs        = P.S > 0 ? 31 - __builtin_clz((unsigned)P.S) : 0;  // factored tiger: S = 2^FS
    const bool ftiger      = P.model == FBA_MODEL_BA_FACTORED && (P.domain == FBA_DOM_FTIGER_EPISODIC || P.domain == FBA_DOM_FTIGER_CONTINUOUS) &&
                             !P.dirichlet_regular && P.planner == FBA_PLANNER_POUCT && !D.hash;
    const bool et = !no_etiger && !P.hist && etiger_ok(P, D) &&
                    (tiger_table || tiger_pomdp || (ftiger && P.ft_packed && ft_fs >= 2 && ft_fs <= 4));
    size_t lds = search_lds_words(P, D, stage, et) * sizeof(uint32_t);
    if (P.model == FBA_MODEL_BA_FACTORED && !(ftiger && ft_fs >= 2 && ft_fs <= 4))
        lds = ((lds + 15) & ~(size_t)15) + (((size_t)P.fd_bytes + 15) & ~(size_t)15);  // + the model description (not the factored-tiger instantiations: 0.9 KB
                                                                                        //   that kept C3 at nine waves per CU where ten fit)
    static const size_t lds_pad = std::getenv("FBA_SEARCH_LDS_PAD") ? (size_t)std::atoi(std::getenv("FBA_SEARCH_LDS_PAD")) : 0;  // occupancy experiments
    lds += lds_pad;
    const int depth_cap = P.max_depth > 0 ? P.max_depth : 1;
    const dim3 grid(ceil_div(P.E, SEARCH_BLOCK)), block(SEARCH_BLOCK);
#define FBA_LAUNCH_SEARCH_M(STG, AM, MODEL)                                                                              \
    do {                                                                                                                 \
        if (P.dirichlet_regular && MODEL != FBA_MODEL_POMDP)                                                             \
            hipLaunchKernelGGL((search_kernel<STG, AM, (MODEL != FBA_MODEL_POMDP), 0, MODEL>), grid, block, lds, st, P, D); \
        else hipLaunchKernelGGL((search_kernel<STG, AM, false, 0, MODEL>), grid, block, lds, st, P, D);             \
    } while (0)
#define FBA_LAUNCH_SEARCH(STG, AM)                                                                  \
    do {                                                                                            \
        if (P.model == FBA_MODEL_BA_FACTORED) FBA_LAUNCH_SEARCH_M(STG, AM, FBA_MODEL_BA_FACTORED);  \
        else if (P.model == FBA_MODEL_BA_TABLE) FBA_LAUNCH_SEARCH_M(STG, AM, FBA_MODEL_BA_TABLE);   \
        else FBA_LAUNCH_SEARCH_M(false, AM, FBA_MODEL_POMDP);                                       \
    } while (0)
    if (P.hist) {  // history particles (gridworld FBA-POMDP): four lanes per tree
        lds = (size_t)depth_cap * HIST_TREES * (sizeof(double) + sizeof(int32_t) + sizeof(float) + sizeof(int32_t)) + (size_t)P.Cs * HIST_TREES * sizeof(float);
        const dim3 qgrid(ceil_div(P.E, HIST_TREES));
        if (D.bkt) {   // the tree as one table of buckets, trips to memory requested an iteration ahead
            const bool no_lrows = D.ab_rows_hbm != 0;   // A/B: transition rows from the padded tables
            const bool lrows = P.hist_lds != nullptr && !no_lrows;
            int nw = H2_WAVES;   // waves per workgroup: as many as 64 KB of LDS hold beside the shared tables (deep horizons have long paths)
            while (nw > 1 && h2_shared_bytes(P, lrows) + (size_t)nw * h2_wave_bytes(P) > 64 * 1024) nw >>= 1;
            const size_t lds2 = h2_shared_bytes(P, lrows) + (size_t)nw * h2_wave_bytes(P);
            const dim3 grid2(ceil_div(P.E, HIST_TREES * nw)), block2(64 * nw);
            if (D.ab_lockstep) hipLaunchKernelGGL(search_order_kernel, dim3(1), dim3(1024), 0, st, P, D);
#define FBA_LAUNCH_H2(KV)                                                                                              \
    do {                                                                                                               \
        if (lds2 > 64 * 1024) {   /* (one wave of a very deep horizon: past the default limit of a workgroup's dynamic LDS) */ \
            if (lrows) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&search_hist2_kernel<KV, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); \
            else (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&search_hist2_kernel<KV, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); \
        }                                                                                                              \
        if (lrows) hipLaunchKernelGGL((search_hist2_kernel<KV, true>), grid2, block2, lds2, st, P, D);                 \
        else hipLaunchKernelGGL((search_hist2_kernel<KV, false>), grid2, block2, lds2, st, P, D);                      \
    } while (0)
            if (P.hist_row <= 8) FBA_LAUNCH_H2(8);
            else if (P.hist_row <= 10) FBA_LAUNCH_H2(12);
            else FBA_LAUNCH_H2(16);
#undef FBA_LAUNCH_H2
            return;
        }
        if (P.hist_row <= 8) hipLaunchKernelGGL((search_hist_kernel<8>), qgrid, block, lds, st, P, D);
        else if (P.hist_row <= 10) hipLaunchKernelGGL((search_hist_kernel<12>), qgrid, block, lds, st, P, D);
        else hipLaunchKernelGGL((search_hist_kernel<16>), qgrid, block, lds, st, P, D);
        return;
    }
    if (tiger_table) {
        if (P.packed && et) hipLaunchKernelGGL((search_kernel<true, 4, false, 2, FBA_MODEL_BA_TABLE, 0, false, false, true>), grid, block, lds, st, P, D);
        else if (P.packed) hipLaunchKernelGGL((search_kernel<true, 4, false, 2, FBA_MODEL_BA_TABLE>), grid, block, lds, st, P, D);
        else if (et) hipLaunchKernelGGL((search_kernel<true, 4, false, 1, FBA_MODEL_BA_TABLE, 0, false, false, true>), grid, block, lds, st, P, D);
        else hipLaunchKernelGGL((search_kernel<true, 4, false, 1, FBA_MODEL_BA_TABLE>), grid, block, lds, st, P, D);
        return;
    }
    if (tiger_pomdp) {
        if (et) hipLaunchKernelGGL((search_kernel<false, 4, false, 0, FBA_MODEL_POMDP, 0, true, false, true>), grid, block, lds, st, P, D);
        else hipLaunchKernelGGL((search_kernel<false, 4, false, 0, FBA_MODEL_POMDP, 0, true>), grid, block, lds, st, P, D);
        return;
    }
    if (ftiger) {
        const int FS = ft_fs;
#define FBA_LAUNCH_FTIGER(FSV)                                                                                                    \
    do {                                                                                                                          \
        if (P.ft_packed && et) hipLaunchKernelGGL((search_kernel<true, 4, false, 0, FBA_MODEL_BA_FACTORED, FSV, false, true, true>), grid, block, lds, st, P, D); \
        else if (P.ft_packed) hipLaunchKernelGGL((search_kernel<true, 4, false, 0, FBA_MODEL_BA_FACTORED, FSV, false, true>), grid, block, lds, st, P, D); \
        else if (stage) hipLaunchKernelGGL((search_kernel<true, 4, false, 0, FBA_MODEL_BA_FACTORED, FSV>), grid, block, lds, st, P, D); \
        else hipLaunchKernelGGL((search_kernel<false, 4, false, 0, FBA_MODEL_BA_FACTORED, FSV>), grid, block, lds, st, P, D);      \
        return;                                                                                                                   \
    } while (0)
        if (FS == 2) FBA_LAUNCH_FTIGER(2);
        if (FS == 3) FBA_LAUNCH_FTIGER(3);
        if (FS == 4) FBA_LAUNCH_FTIGER(4);
#undef FBA_LAUNCH_FTIGER
    }
    if (P.A <= 4) { if (stage) FBA_LAUNCH_SEARCH(true, 4); else FBA_LAUNCH_SEARCH(false, 4); }
    else if (P.A <= 8) { if (stage) FBA_LAUNCH_SEARCH(true, 8); else FBA_LAUNCH_SEARCH(false, 8); }
    else if (P.A <= 16) { if (stage) FBA_LAUNCH_SEARCH(true, 16); else FBA_LAUNCH_SEARCH(false, 16); }
    else FBA_LAUNCH_SEARCH_M(false, 24, FBA_MODEL_POMDP);  // agr (23 actions) is a POMDP-only domain
#undef FBA_LAUNCH_SEARCH_M
#undef FBA_LAUNCH_SEARCH
}
#ifdef FBA_PROFILE_SEARCH
extern "C" int fba_debug_search_profile(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_search_prof), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) {
        const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_search_prof), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif

}  // namespace fba
