// fba_kernels_common.h -- device helpers shared by the search kernels (fba_search.hip) and the belief / episode kernels
// (fba_kernels.hip): stream addressing, particle record accessors, filter sampling, tree node records, UCB.
#pragma once

#include <float.h>

#include "fba_kernels.h"

namespace fba {


// ---------------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ Rng slot_rng(const Problem& P, const DeviceState& D, int e)
{
    Rng g;
    g.seed(P.seed_lo, P.seed_hi);
    g.position((uint32_t)D.run[e], (uint32_t)D.episode[e], (uint32_t)D.t[e]);
    g.draw = 0; g.c1 = 0; g.keep_lo = 0; g.keep_hi = 0;
    return g;
}

__device__ __forceinline__ size_t pbase(const Problem& P, int e, int buf) { return ((size_t)buf * P.E + e) * (size_t)P.N; }
// first record of slot e's current filter / of the filter a resample or reset is building (DeviceState::single_rec)
__device__ __forceinline__ size_t rec_base(const Problem& P, const DeviceState& D, int e, int buf) { return D.single_rec ? (size_t)D.rec_buf[e] * (size_t)P.N : pbase(P, e, buf); }
// the slot block `b` of a chunked launch works on: the b-th of the chunk, or of the compacted list (DeviceState::use_list)
__device__ __forceinline__ int chunk_slot(const DeviceState& D, int b) { return D.use_list ? D.slot_list[D.slot_base + b] : D.slot_base + b; }
__device__ __forceinline__ int scratch_place(const DeviceState& D, int e) { return D.use_list ? D.scratch_idx[e] : e - D.slot_base; }
__device__ __forceinline__ float* rec_dst(const Problem& P, const DeviceState& D, int e, int other)
{
    return D.single_rec ? D.p_rec + (size_t)D.rec_buf[P.E + scratch_place(D, e)] * (size_t)P.N * (size_t)P.Cs : D.p_rec + pbase(P, e, other) * (size_t)P.Cs;
}

// particle record accessors (layout: fba_state.h)
__device__ __forceinline__ int rec_state(const float* rec, int C) { return __float_as_int(rec[C]); }
__device__ __forceinline__ void rec_set_state(float* rec, int C, int s) { rec[C] = __int_as_float(s); }

// resetDomainStateDistribution of the plain rejection filter (BARejectionSampling.cpp:49-60) gives particle i
// the start state drawn from stream (run, episode, 0, RESET, i).  Writing 4 bytes into each of N records is a
// poor use of HBM, so the reset is only flagged (lazy_reset_kernel) and the state is derived where it is
// read -- the search's root sampling, the rejection attempts, the checksum -- until the first rejection
// update of the episode rewrites every record anyway.
__device__ __forceinline__ bool slot_lazy(const DeviceState& D, int e) { return D.lazy_reset[e] != 0; }
__device__ __forceinline__ int lazy_state(const Problem& P, const DeviceState& D, int e, int i)
{
    Rng g = slot_rng(P, D, e);
    g.position((uint32_t)D.run[e], (uint32_t)D.episode[e], 0);
    g.stream(FBA_PHASE_RESET, (uint32_t)i);
    return domain_start(P, g);
}

// WeightedFilter::sample (WeightedFilter.cpp:163-191) in device order: the largest i >= 1 whose
// exclusive prefix sum is below the threshold, else 0.  `incl` holds inclusive prefix sums.
__device__ __forceinline__ int weighted_pick(const double* __restrict__ incl, int n, double threshold)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (incl[mid - 1] < threshold) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// weighted_pick started where the threshold would sit if the weights were equal, galloping from there to bracket the
// answer before bisecting: the same index (incl is non-decreasing and the index is defined by incl alone), three to five
// dependent loads instead of log2(N) when the weights are of similar size -- the usual case, a filter is resampled
// after every update.
__device__ __forceinline__ int weighted_pick_guided(const double* __restrict__ incl, int n, double threshold, double total)
{
    if (n <= 1) return 0;
    int i = (int)(threshold / total * (double)n);
    i = min(max(i, 1), n - 1);
    int lo, hi;  // the answer is the largest i in [max(lo, 1), hi] with incl[i - 1] < threshold, or 0 if there is none
    if (incl[i - 1] < threshold) {   // i qualifies: gallop upwards until a candidate does not
        int step = 1;
        lo = i;
        while (lo + step <= n - 1 && incl[lo + step - 1] < threshold) { lo += step; step <<= 1; }
        hi = min(lo + step - 1, n - 1);
    } else {                          // i does not: gallop downwards until one does
        int cur = i, step = 1;
        while (true) {
            const int cand = cur - step;
            if (cand < 1) { lo = 0; hi = cur - 1; break; }
            if (incl[cand - 1] < threshold) { lo = cand; hi = cur - 1; break; }
            cur = cand;
            step <<= 1;
        }
    }
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (incl[mid - 1] < threshold) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// weighted_pick on the prefix sums of N equal weights: the answer is within a step or two of
// threshold / total * N, so start there and walk -- same result as the binary search (incl is
// non-decreasing), two or three loads instead of log2(N) dependent ones.
__device__ __forceinline__ int uniform_weight_pick(const double* __restrict__ incl, int n, double threshold, double total)
{
    int i = (int)(threshold / total * (double)n);
    i = min(max(i, 0), n - 1);
    while (i < n - 1 && incl[i] < threshold) ++i;          // i + 1 still qualifies
    while (i > 0 && !(incl[i - 1] < threshold)) --i;       // i itself does not
    return i;
}

// Belief::sample() of a freshly initiated / resampled filter (all weights 1/N)
template <class RNG>
__device__ __forceinline__ int belief_sample_uniform(const Problem& P, const DeviceState& D, RNG& g)
{
    if (P.belief == FBA_BELIEF_REJECTION) return P.point ? 0 : g.uniform_int(P.N);  // FlatFilter.cpp:97-102; PointEstimation::sample :30-34
    return uniform_weight_pick(D.uni_scan, P.N, g.u01() * D.uni_total, D.uni_total);
}

// Belief::sample() of the nested belief (NestedBelief::sample NestedBelief.cpp:116-125): a count particle by weight,
// then one of its domain states
__device__ __forceinline__ size_t nest_base(const Problem& P, int e, int buf) { return ((size_t)buf * P.E + e) * (size_t)P.N * (size_t)P.nested; }
template <class RNG>
__device__ __forceinline__ int nested_sample(const Problem& P, const DeviceState& D, int e, RNG& g, int& state)
{
    const int src = weighted_pick(D.nest_scan + (size_t)e * P.N, P.N, g.u01() * D.nest_total[e]);
    state = D.nest_s[nest_base(P, e, D.nest_sel[e]) + (size_t)src * P.nested + g.uniform_int(P.nested)];
    return src;
}

__device__ __forceinline__ void node_init(const DeviceState& D, int32_t* rec, int A, int O)
{
    if (D.cn_off) rec[0] = 0;   // (cn_off = 0: no visits word, ActionNode::_visits is the sum of its chance nodes' counts)
    for (int a = 0; a < A; ++a) rec[D.cn_off + a] = 0;
    double* q = reinterpret_cast<double*>(rec + D.cq_off);
    for (int a = 0; a < A; ++a) q[a] = 0.0;
    if (!D.hash)
        for (int k = 0; k < A * O; ++k) rec[D.child_off + k] = -1;
}

// ChanceNode::child / hasChild / addChild (MCTSTreeNodes.cpp:35-50): dense table in the node record,
// or the epoch-tagged hash table (fba_state.h)
__device__ __forceinline__ uint32_t child_hash(uint64_t code)
{
    code ^= code >> 33; code *= 0xff51afd7ed558ccdull; code ^= code >> 29;
    return (uint32_t)code;
}
// Compact form (DeviceState::hash_compact, whenever max_nodes * A * O < 2^27, e.g. gridworld N = 7 at 65 536
// simulations): 8-byte entries {epoch << 27 | code, child}, half the slots (load <= 1/2) -- 1 MB per tree instead of 4;
// a 5-bit epoch, so the slot's table is cleared once every 31 searches (hash_begin_search).
__device__ __forceinline__ uint32_t hash_begin_search(const DeviceState& D, int e, int4* tab, int part, int nparts)
{
    uint32_t epoch = D.epoch[e] + 1;
    if (D.hash_compact) {
        if (epoch > 31u) {
            uint4* t4 = reinterpret_cast<uint4*>(tab);
            for (uint32_t k = (uint32_t)part; k < (D.hmask + 1) / 2; k += (uint32_t)nparts) t4[k] = make_uint4(0, 0, 0, 0);
            epoch = 1;
        }
    } else {
        epoch &= 0x0fffffffu;
        if (epoch == 0) epoch = 1;
    }
    D.epoch[e] = epoch;
    return epoch;
}
__device__ __forceinline__ int4* hash_table(const DeviceState& D, int e)
{
    if (!D.hash) return nullptr;
    if (D.hash_compact) return reinterpret_cast<int4*>(reinterpret_cast<uint2*>(D.hash) + (size_t)e * (D.hmask + 1));
    return D.hash + (size_t)e * (D.hmask + 1);
}
__device__ __forceinline__ int child_get(const Problem& P, const DeviceState& D, const int32_t* tree, int4* tab, uint32_t epoch,
                                         int node, int a, int o)
{
    if (!D.hash) return tree[(size_t)node * D.node_words + D.child_off + a * P.O + o];
    const uint64_t code = ((uint64_t)node * P.A + a) * P.O + o;
    if (D.hash_compact) {
        const uint2* t8 = reinterpret_cast<const uint2*>(tab);
        const uint32_t key = (uint32_t)code | (epoch << 27);
        for (uint32_t h = child_hash(code) & D.hmask;; h = (h + 1) & D.hmask) {
            const uint2 en = t8[h];
            if ((en.x >> 27) != epoch) return -1;
            if (en.x == key) return (int)en.y;
        }
    }
    const uint32_t lo = (uint32_t)code, hi = (uint32_t)(code >> 32) | (epoch << 4);
    for (uint32_t h = child_hash(code) & D.hmask;; h = (h + 1) & D.hmask) {
        const int4 e = tab[h];
        if (((uint32_t)e.y >> 4) != epoch) return -1;  // empty or left over from an earlier search
        if ((uint32_t)e.x == lo && (uint32_t)e.y == hi) return e.z;
    }
}
__device__ __forceinline__ void child_set(const Problem& P, const DeviceState& D, int32_t* tree, int4* tab, uint32_t epoch, int node,
                                          int a, int o, int child)
{
    if (!D.hash) {
        tree[(size_t)node * D.node_words + D.child_off + a * P.O + o] = child;
        return;
    }
    const uint64_t code = ((uint64_t)node * P.A + a) * P.O + o;
    if (D.hash_compact) {
        uint2* t8 = reinterpret_cast<uint2*>(tab);
        for (uint32_t h = child_hash(code) & D.hmask;; h = (h + 1) & D.hmask) {
            if ((t8[h].x >> 27) != epoch) {
                t8[h] = make_uint2((uint32_t)code | (epoch << 27), (uint32_t)child);
                return;
            }
        }
    }
    const uint32_t lo = (uint32_t)code, hi = (uint32_t)(code >> 32) | (epoch << 4);
    for (uint32_t h = child_hash(code) & D.hmask;; h = (h + 1) & D.hmask) {
        if (((uint32_t)tab[h].y >> 4) != epoch) {
            tab[h] = make_int4((int)lo, (int)hi, child, 0);
            return;
        }
    }
}

// POUCT::selectChanceNodeUCB (POUCT.cpp:138-181 = RBAPOUCT.cpp:162-205).
// UCB(m, n) = u * sqrt(log1p(m) / n), DBL_MAX for n = 0 (POUCT.cpp:330-338); ties are collected in
// action order and one slowRandomInt is ALWAYS drawn, also for a single candidate.
// The node's statistics arrive in registers (cn / cq); log1p_tab[visits] = log1p(visits) as the host's libm gives it (the
// value the reference's table entry is built from) is only loaded where the exact fp64 comparison is needed.
template <int AMAX, class RNG>
__device__ __forceinline__ int ucb_pick(const Problem& P, RNG& g, const double* __restrict__ log1p_tab, int visits, const int (&cn)[AMAX],
                                        const double (&cq)[AMAX], bool explore)
{
    double best_q = -DBL_MAX;
    uint32_t mask = 0;
    bool decided = false;
    if (explore) {
        // Fast paths that give the fp64 arg-max set exactly, without the fp64 divisions and square roots:
        //  * unvisited actions: q + DBL_MAX rounds to DBL_MAX for every finite q here, so they tie exactly and
        //    beat every visited action -- the candidate set is the set of unvisited actions;
        //  * otherwise evaluate q + u sqrt(L / n) in fp32, with L = log(1 + visits) from the hardware's log2 (1 ulp) -- no
        //    trip to the table: each value lands within ~2^-20 of (|q| + bonus); if one action leads by more than 1e-5
        //    of the largest |q| + bonus -- ten times that error bound -- it is the unique fp64 maximum.  Anything closer
        //    falls through to fp64 and the table.  (A dependent table load per tree level was 6 % of the bench's search.)
        uint32_t unvisited = 0;
#pragma unroll
        for (int a = 0; a < AMAX; ++a)
            if (a < P.A && cn[a] == 0) unvisited |= 1u << a;
        if (unvisited) {
            mask    = unvisited;
            decided = true;
        } else {
            const float Lf = __log2f((float)(visits + 1)) * 0.69314718f, uf = (float)P.exploration;
            float w[AMAX], top = -FLT_MAX, scale = 0.f;
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                const float qa = (float)cq[a];
                const float b  = uf * __builtin_amdgcn_sqrtf(Lf * __builtin_amdgcn_rcpf((float)(a < P.A ? cn[a] : 1)));
                w[a]  = (a < P.A) ? qa + b : -FLT_MAX;
                top   = fmaxf(top, w[a]);
                scale = fmaxf(scale, (a < P.A) ? fabsf(qa) + b : 0.f);
            }
            const float cut = top - 1.0e-5f * scale;
            uint32_t near = 0;
#pragma unroll
            for (int a = 0; a < AMAX; ++a)
                if (w[a] >= cut) near |= 1u << a;
            if (__popc(near) == 1 && scale < 1.0e30f) {
                mask    = near;
                decided = true;
            } else if (AMAX <= 5 && scale < 1.0e30f) {   // (the instantiations of many actions would pay for the selects below with spilled registers)
                //  * several actions within the margin that carry the SAME count and the SAME Q (young nodes of a domain whose rewards are mostly
                //    zero: n = 1, Q = 0 each): their fp64 values are one and the same number, so they tie exactly, and every other action is below
                //    them by more than the error bound -- the candidate set is `near`, again without the divisions and square roots.
                const int f0 = __ffs(near) - 1;
                int n0 = cn[0];
                double q0 = cq[0];
#pragma unroll
                for (int a = 1; a < AMAX; ++a) { n0 = f0 == a ? cn[a] : n0; q0 = f0 == a ? cq[a] : q0; }
                bool same = true;
#pragma unroll
                for (int a = 0; a < AMAX; ++a)
                    if ((near >> a) & 1u) same = same && cn[a] == n0 && cq[a] == q0;
                if (same) {
                    mask    = near;
                    decided = true;
                }
            }
        }
    }
#ifdef FBA_TIMING_NO_FP64_UCB   // timing-only builds (scripts/): what the fp64 fall-through costs -- WRONG results
    if (!decided && explore) { mask = 1u; decided = true; }
#endif
    if (!decided) {
        const double L = explore ? log1p_tab[visits] : 0.0;
#pragma unroll
        for (int a = 0; a < AMAX; ++a)
            if (a < P.A) {
                double q = cq[a];
                if (explore) q += (cn[a] == 0) ? DBL_MAX : P.exploration * sqrt(L / (double)cn[a]);
                if (q >= best_q) {
                    if (q > best_q) mask = 0;
                    best_q = q;
                    mask |= 1u << a;
                }
            }
    }
    int k = g.slow_int(0, __popc(mask));
    while (k-- > 0) mask &= mask - 1;  // drop the k lowest candidates
    return __ffs(mask) - 1;
}

template <int AMAX, class RNG>
__device__ __forceinline__ int ucb_select(const Problem& P, const DeviceState& D, RNG& g, const int32_t* rec, bool explore)
{
    int cn[AMAX];
    double cq[AMAX];
    int visits;
    if (AMAX >= 3 && P.A == 3) {  // header {visits, n0, n1, n2} and the three Q values: one 16-byte, one 16-byte, one 8-byte load
        const int4 h = *reinterpret_cast<const int4*>(rec);
        const double2 q01 = *reinterpret_cast<const double2*>(rec + 4);
        const double q2   = *reinterpret_cast<const double*>(rec + 8);
        visits = h.x;
#pragma unroll
        for (int a = 0; a < AMAX; ++a) { cn[a] = 0; cq[a] = 0.0; }
        cn[0] = h.y; cn[1] = h.z; cn[2] = h.w;
        cq[0] = q01.x; cq[1] = q01.y; cq[2] = q2;
    } else if (AMAX == 4 && P.A == 4 && D.cn_off == 0) {  // {n0..n3}, {q0, q1}, {q2, q3}: three 16-byte loads of one 48-byte record
        const int4 h      = *reinterpret_cast<const int4*>(rec);
        const double2 q01 = *reinterpret_cast<const double2*>(rec + 4);
        const double2 q23 = *reinterpret_cast<const double2*>(rec + 8);
        cn[0] = h.x; cn[1] = h.y; cn[2] = h.z; cn[3] = h.w;
        cq[0] = q01.x; cq[1] = q01.y; cq[2] = q23.x; cq[3] = q23.y;
        visits = ((h.x + h.y) + h.z) + h.w;
    } else {
        const double* q = reinterpret_cast<const double*>(rec + D.cq_off);
        visits = D.cn_off ? rec[0] : 0;
#pragma unroll
        for (int a = 0; a < AMAX; ++a) {
            cn[a] = a < P.A ? rec[D.cn_off + a] : 0;
            cq[a] = a < P.A ? q[a] : 0.0;
            if (!D.cn_off) visits += cn[a];   // every back-up through the node adds one to exactly one of them (MCTSTreeNodes.cpp:8-12, 59-62)
        }
    }
    return ucb_pick<AMAX>(P, g, D.log1p_tab, visits, cn, cq, explore);
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace fba
