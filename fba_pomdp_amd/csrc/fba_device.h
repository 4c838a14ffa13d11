// fba_device.h -- device-side building blocks of the BA-POMCP engine (gfx950 only).
//
// Everything the kernels share: the Philox4x32-10 counter RNG (one stream per
// (run, episode, t, phase, unit)), the reference's sampling primitives with their exact
// float/double mix, the domain dynamics and the Bayes-adaptive count models.
// Built with -ffp-contract=off: the reference's a*b+c expressions round twice and so must we.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fba_hip.h"

namespace fba {

// ---------------------------------------------------------------------------------------------
// Philox4x32-10.  ctr = (block, unit, phase | t<<8 | episode<<16, run), key = seed.
// A draw is 64 bits; block b serves draws 2b and 2b+1.
// ---------------------------------------------------------------------------------------------
struct Rng {
    uint32_t k0, k1;   // key (seed)
    uint32_t c1, c2, c3;
    uint32_t draw;     // draws consumed in the current stream
    uint32_t keep_lo, keep_hi;

    __device__ __forceinline__ void seed(uint32_t lo, uint32_t hi) { k0 = lo; k1 = hi; }
    __device__ __forceinline__ void position(uint32_t run, uint32_t episode, uint32_t t)
    {
        c3 = run;
        c2 = ((t & 0xffu) << 8) | ((episode & 0xffffu) << 16);
    }
    __device__ __forceinline__ void stream(uint32_t phase, uint32_t unit)
    {
        c2   = (c2 & 0xffffff00u) | (phase & 0xffu);
        c1   = unit;
        draw = 0;
    }
    // Philox4x32-10 block `idx` of the current stream: draws 2 idx (words 0, 1) and 2 idx + 1 (words 2, 3)
    __device__ __forceinline__ void block(uint32_t idx, uint32_t& w0, uint32_t& w1, uint32_t& w2, uint32_t& w3) const
    {
        uint32_t x0 = idx, x1 = c1, x2 = c2, x3 = c3;
        uint32_t a = k0, b = k1;
#ifdef FBA_CHEAP_RNG   /* measurement builds only (scripts/search_regions.py): what the search would cost with a nearly free generator */
#define FBA_PHILOX_ROUNDS 5
#else
#define FBA_PHILOX_ROUNDS 10
#endif
#pragma unroll
        for (int i = 0; i < FBA_PHILOX_ROUNDS; ++i) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)x0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)x2;  // one 32x32->64 multiply each
            const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
            uint32_t n0 = hi1 ^ x1 ^ a, n2 = hi0 ^ x3 ^ b;
            x0 = n0; x1 = lo1; x2 = n2; x3 = lo0;
            a += 0x9E3779B9u;
            b += 0xBB67AE85u;
        }
        w0 = x0; w1 = x1; w2 = x2; w3 = x3;
    }
    __device__ __forceinline__ uint64_t next64()
    {
        uint64_t r;
        if ((draw & 1u) == 0) {
            uint32_t x0, x1, x2, x3;
            block(draw >> 1, x0, x1, x2, x3);
            r       = ((uint64_t)x1 << 32) | x0;
            keep_lo = x2;
            keep_hi = x3;
        } else {
            r = ((uint64_t)keep_hi << 32) | keep_lo;
        }
        ++draw;
        return r;
    }
    // rnd::uniform_rand01 (random.cpp:100) -- 53-bit mantissa from the top of the draw
    __device__ __forceinline__ double u01() { return (double)(next64() >> 11) * (1.0 / 9007199254740992.0); }
    // rnd::boolean (random.cpp:90): the event u01 < 0.5
    __device__ __forceinline__ bool boolean() { return (next64() >> 63) == 0; }
    // uniform_int_distribution<int>(0, n-1)
    __device__ __forceinline__ int uniform_int(int n) { return (int)__umul64hi(next64(), (uint64_t)(uint32_t)n); }
    // rnd::slowRandomInt (random.cpp:111)
    __device__ __forceinline__ int slow_int(int lo, int hi) { return lo + (int)floor(u01() * (double)(hi - lo)); }
};

// ---------------------------------------------------------------------------------------------
// Views of a particle's count blob: straight from HBM, or staged in LDS as [word][lane]
// (stride = workgroup size, so lane l only ever touches bank l mod 32: conflict-free).
// ---------------------------------------------------------------------------------------------
// `row_regs`: read a Dirichlet row into registers before using it (sample_expected_mult).  Pays off where
// few waves are in flight and every element would otherwise be a separate wait on HBM / L2 -- the
// search over records too big to stage; costs instructions where occupancy already hides the latency
// (belief kernels) or the row sits in LDS.
struct GlobalView {
    const float* p;
    static constexpr bool row_regs = false;
    __device__ __forceinline__ float at(int k) const { return p[k]; }
};
struct GlobalSearchView {
    const float* p;
    static constexpr bool row_regs = true;
    __device__ __forceinline__ float at(int k) const { return p[k]; }
};
// A record seen through increments that have not been written yet: at(k) = base.at(k) + 1.0f for every
// pending increment of cell k, added one at a time as incrementCountsOf would (importance update:
// the observation probability is read from the counts AFTER the step's increments).
template <class Base>
struct PendingIncView {
    Base base;
    const int32_t* col;  // pending cell indices col[q * stride], q < ninc
    int stride, ninc;
    static constexpr bool row_regs = Base::row_regs;
    __device__ __forceinline__ float at(int k) const
    {
        float v = base.at(k);
        for (int q = 0; q < ninc; ++q)
            if (col[q * stride] == k) v += 1.0f;
        return v;
    }
};
template <int STRIDE>
struct LdsView {
    const float* p;
    static constexpr bool row_regs = false;
    __device__ __forceinline__ float at(int k) const { return p[k * STRIDE]; }
};
// Packed tabular particle (Problem::packed): the record holds, per cell of the count table, how many "+1"s the
// cell has received (uint16, two per 32-bit word, even cell in the low half); count = prior[k] + that number.
// The engine only packs when prior[k] + 65535 is exactly representable in fp32 for every k, so the sum below
// is the very float the reference reaches by adding 1.0f that many times.  `words.at(w)` = word w of the record.
template <class Base>
struct PackedView {
    Base words;
    const float* prior;  // the dense prior table (LDS copy)
    static constexpr bool row_regs = false;
    __device__ __forceinline__ float at(int k) const
    {
        const uint32_t w = __float_as_uint(words.at(k >> 1));
        return prior[k] + (float)((k & 1) ? (w >> 16) : (w & 0xffffu));
    }
};

// Packed factored-tiger particle (Problem::ft_packed, FS = K + 1 binary state features): the record holds, per cell of
// the count table, how many "+1"s the cell has received (uint16, two per word, dense order), then the parent set of
// the listen observation node (one word), then the state.  A count is prior(cell, parent set) + that number: the prior
// of this model is a function of the cell and the structure bits alone -- FactoredTigerFactoredPrior's nodes
// (FactoredTigerPriors.cpp:95-195) and setObservationModel (:221-263), layout of build_ftiger_factored_prior:
//   T(open a, f) 2 counts at (a*FS + f)*2: 5000, 5000      T(listen, f) 4 counts at 4FS + 4f: 5000 where the value is kept
//   O(open a) 2 counts at 8FS + 2a: 5000, 5000             O(listen) rows of 2 at 8FS + 4: (acc, inacc) / (inacc, acc) by the
//   tiger's door when feature 0 is a parent, (unif, unif) otherwise, nothing beyond the 2^(#parents) rows in use.
// Used only when every such prior value p has p + 65535 exact in fp32 (checked on the host), so the sum is the
// very float the reference reaches by adding 1.0f that many times.  144 B instead of 288 B at K = 3.
template <int FS, class Base>
struct PackedFtigerView {
    Base words;
    uint32_t mask;       // the structure bits (word NC / 2 of the record), read once
    float acc, inacc, unif;
    static constexpr bool row_regs = false;
    static constexpr int NC = 8 * FS + 4 + (2 << FS);   // counts; the dense blob has the mask word behind them
    __device__ __forceinline__ float prior(int k) const
    {
        if (k < 4 * FS) return 5000.f;
        if (k < 8 * FS) { const int j = k - 4 * FS; return (((j >> 1) ^ j) & 1) ? 0.f : 5000.f; }
        if (k < 8 * FS + 4) return 5000.f;
        const int j = k - (8 * FS + 4), row = j >> 1, np = __popc(mask);
        if (row >= (1 << np)) return 0.f;
        if (!(mask & 1u)) return unif;
        return ((row >> (np - 1)) == (j & 1)) ? acc : inacc;
    }
    __device__ __forceinline__ float at(int k) const
    {
        if (k >= NC) return __uint_as_float(mask);
        const uint32_t w = __float_as_uint(words.at(k >> 1));
        return prior(k) + (float)((k & 1) ? (w >> 16) : (w & 0xffffu));
    }
};
// ---------------------------------------------------------------------------------------------
// Sampling primitives (reference src/utils/random.hpp:93-115, random.cpp:244-279)
// ---------------------------------------------------------------------------------------------

// sampleFromMult<float const>: CDF accumulated in float, compared with a double threshold.
template <class RNG, class View>
__device__ __forceinline__ int sample_from_mult_f(RNG& g, const View& row, int off, int n, double total)
{
    const double p = g.u01() * total;
    float sum      = row.at(off);
    for (int i = 1; i < n; ++i) {
        if (p < (double)sum) return i - 1;
        sum += row.at(off + i);
    }
    return n - 1;
}

// sampleFromExpectedMult: total accumulated in double.
// Views with `row_regs` read rows of up to ROWREG entries into registers first, all loads issued
// together: the two passes (total, CDF) then wait for memory once instead of once per element.
// Same operations, same order, same result.
constexpr int ROWREG = 16;
template <int K, class View>
__device__ __forceinline__ int sample_expected_mult_regs(Rng& g, const View& row, int off, int n)
{
    float r[K];
#pragma unroll
    for (int i = 0; i < K; ++i) r[i] = i < n ? row.at(off + i) : 0.f;
    double total = (double)r[0];
#pragma unroll
    for (int i = 1; i < K; ++i)
        if (i < n) total += (double)r[i];
    const double p = g.u01() * total;
    float sum = r[0];
    int pick  = n - 1;
    bool done = false;
#pragma unroll
    for (int i = 1; i < K; ++i)
        if (i < n && !done) {
            if (p < (double)sum) { pick = i - 1; done = true; }
            else sum += r[i];
        }
    return pick;
}
template <class View>
__device__ __forceinline__ int sample_expected_mult(Rng& g, const View& row, int off, int n)
{
    if (View::row_regs && n <= ROWREG) {  // (rows of one node have one length: the branch is as good as uniform)
        if (n <= 2) return sample_expected_mult_regs<2>(g, row, off, n);
        if (n <= 8) return sample_expected_mult_regs<8>(g, row, off, n);
        return sample_expected_mult_regs<ROWREG>(g, row, off, n);
    }
    double total = (double)row.at(off);
    for (int i = 1; i < n; ++i) total += (double)row.at(off + i);
    return sample_from_mult_f(g, row, off, n, total);
}

// expectedMult(row)[o]: float sum, float division (all-zero if the sum underflows)
template <class View>
__device__ __forceinline__ double expected_mult_at(const View& row, int off, int n, int o)
{
    float sum = row.at(off);
    for (int i = 1; i < n; ++i) sum += row.at(off + i);
    if ((double)sum <= 1e-300) return 0.0;
    return (double)(row.at(off + o) / sum);
}

// ---------------------------------------------------------------------------------------------
// Problem description shared by all kernels (filled on the host)
// ---------------------------------------------------------------------------------------------
// Factored Bayes-adaptive model (BABNModel + DBNNode, reference
// src/bayes-adaptive/states/factored/) in a fixed "max layout": node (a, f) owns a region of the
// count blob large enough for its biggest allowed parent set; a particle's actual parents are a
// bit mask over the node's `maxp` list -- fixed for the ctx, or per particle in mask word `var`
// stored right after the counts -- and rows are indexed compactly by the actual parents, exactly
// as DBNNode::cptIndex (DBNNode.cpp:171-205).
constexpr int MAXF     = 8;   // state / observation features
constexpr int MAXNODES = 160; // A * (FS + FO)
constexpr int MAXINC   = 9;   // count increments of one UpdateCounts step (FS + FO)
struct alignas(16) FNode {  // 48 bytes: three 16-byte loads bring a whole node description (load_node)
    int32_t off, out, nmax, var;
    uint32_t fixed_mask;
    uint8_t maxp[MAXF];  // candidate parents (state features), in order
    uint8_t psz[MAXF];   // psz[j] = number of values of parent maxp[j] (filled by the engine after the build)
    uint32_t pad[3];
};
static_assert(sizeof(FNode) == 48, "load_node reads an FNode as three 16-byte words");
struct FDesc {
    int32_t FS, FO, nvar, ncounts;
    int32_t Ssz[MAXF], Osz[MAXF], Sstep[MAXF], Ostep[MAXF];
    FNode nodes[MAXNODES];  // T(a, f) at a*FS + f, O(a, f) at A*FS + a*FO + f
};

// GridWorld geometry (reference src/domains/gridworld/GridWorld.cpp: goalLocations :124-152,
// generateSlowLocations :78-103, _obs_displacement_probs ctor :60-70), built on the host
struct GridDesc {
    int32_t N, G, nslow;
    int32_t goal[16][2], slow[8][2];
    float disp[32];
};

// Collision avoidance (reference src/domains/collision-avoidance/CollisionAvoidance.cpp), tables
// built on the host with libm: err[d] = Phi(d+.5) - Phi(d-.5) (_observation_error_probability),
// phi[k] = Phi(k - 9 + .5) (inverse-CDF thresholds of round(N(0,1)), the observation noise)
struct CADesc {
    int32_t W, H, n, Hn;
    int32_t start_i0, start_cnt;
    float start_v;
    double start_total;
    double err[16];
    double phi[20];
};

// SysAdmin (reference src/domains/sysadmin/SysAdmin.cpp): bit c of the state = computer c operational;
// keep[n] = (1 - fail_prob) * pow(1 - fail_neighbour_factor, n), n failing neighbours, built on the
// host with libm exactly as SysAdmin::step (:116-118) evaluates it
struct SysDesc {
    int32_t N, pad;
    double keep[3];
};

// Where the "+1"s of one UpdateCounts step go.  A step reports them as (slot k, blob index) pairs;
// slot k < ninc(model).  The search ignores them (KeepCounts); the belief kernels park them in one
// LDS column per thread, [k][thread], so nothing is indexed at run time in registers.
struct NoInc {
    __device__ __forceinline__ void add(int, int) const {}
};
template <int STRIDE>
struct LdsInc {
    int32_t* col;  // &s_inc[thread]
    __device__ __forceinline__ void add(int k, int idx) const { col[k * STRIDE] = idx; }
};

struct Problem {
    const FDesc* fd;  // device pointer; null unless model = BA_FACTORED
    const GridDesc* gw;  // device pointer; null unless domain = gridworld
    const CADesc* ca;    // device pointer; null unless domain = collision avoidance
    const struct SysDesc* sys;  // device pointer; null unless domain = sysadmin
    const struct ZigDesc* zig;  // ziggurat tables (regular Dirichlet mode only)
    int32_t dirichlet_regular;  // --dirichlet_sampling_method regular
    float noise, counts_total;
    int32_t structure_prior;
    int32_t domain, model, belief, planner;
    int32_t ca_plain;   // collision avoidance / sysadmin, factored model in the prior's own fixed graph (no masks): ca_fact_step / sysadmin_fact_step apply
    int32_t fd_bytes;   // bytes of *fd in use (header + A*(FS+FO) nodes): what a kernel stages in LDS
    int32_t incub;      // incubator belief (belief = REJECTION then): shadow particles bred per update, 0 = off
    int32_t nested;     // nested belief (belief = IMPORTANCE then): domain states per count particle (N^2), 0 = off
    int32_t mh;         // mh-within-gibbs belief (belief = IMPORTANCE then): 1 = state histories by message passing, 2 = by rejection sampling
    int32_t cheat;      // cheating belief: particles copied from the correct-graph filter per cheat (belief = IMPORTANCE then); 0 = off
    int32_t packed;     // tabular tiger particles stored as uint16 increment counts over the shared prior (PackedView); C, Cs are then in words of that record
    int32_t ft_packed;  // factored-tiger particles stored packed (PackedFtigerView): uint16 increments, then the structure bits, then the state
    float ft_acc, ft_inacc, ft_unif;   // ... and the three prior counts of its listen observation node (FactoredTigerPriors.cpp:221-263)
    int32_t point;      // point-estimate belief: N = 1 and Belief::sample() returns the state without a draw
    int32_t reinvig;    // reinvigoration belief: particles bred per update (belief = REJECTION then); 0 = off
    int32_t hist;       // history particles (gridworld FBA-POMDP, importance filter): a record holds the particle's increments as one
                        // 4-byte entry per real step over the shared prior tables (HistView below); C = 0 then, hist_cap entries per record
    int32_t hist_cap;
    int32_t hist_compact;   // records of short histories stand closer than Cs words (hist_stride below)
    int32_t gw_N, gw_G;          // gridworld: N, number of goals (copies of GridDesc's, as kernel arguments)
    uint32_t gw_goalcell0, gw_goalcell1, gw_goalcell2, gw_goalcell3;  // gridworld: x*N + y of goal g, 8 bits each, four goals per word (scalars, not an
                                                                      // array: an indexed member pins the whole by-value struct to scratch memory)
    int32_t search_budget;  // > 0: search_hist_kernel stops at the first simulation boundary behind this many loop iterations and parks the search (fba_hip.h)
    int32_t hist_row;   // the longest Dirichlet row of the model (max(N, G)): picks the row width the kernels are instantiated for
    const float* hist_base;  // the prior count table every particle starts from (max layout, x / y nodes without the goal parent)
    const float* hist_alt;   // [A][2][N*N*G*N]: the x / y transition nodes as a particle with the goal as their third parent starts them
    // The same rows once more, deduplicated (a gridworld prior has a few dozen distinct Dirichlet rows among its ten thousand): one byte per
    // row slot (HistRowIds numbering) + the distinct rows, hist_row rounded up to 4 floats each -- 13 KB at N = 7, which the search keeps in LDS
    // so that no simulated step waits for a row.  Null when the prior has more than 256 distinct rows.
    const uint8_t* hist_lds;    // [hist_rid_bytes] row ids, then [hist_distinct][K] floats
    int32_t hist_rid_bytes;     // multiple of 16
    int32_t hist_distinct;
    int32_t S, A, O;
    int32_t N;          // particles per slot
    int32_t C;          // floats per particle count blob
    int32_t Cs;         // record stride in floats: counts, then the state word at index C, then padding
    int32_t phi_len;    // tabular: S*A*S
    int32_t sims, max_depth, horizon, episodes;
    int32_t E;          // slots
    double exploration, gamma;
    uint32_t seed_lo, seed_hi;
};

template <int FS, class Base>
__device__ __forceinline__ PackedFtigerView<FS, Base> packed_ftiger_view(const Problem& P, const Base& words)
{
    return PackedFtigerView<FS, Base>{words, __float_as_uint(words.at(PackedFtigerView<FS, Base>::NC / 2)), P.ft_acc, P.ft_inacc, P.ft_unif};
}

__device__ __forceinline__ bool dom_is_tiger(int d) { return d == FBA_DOM_TIGER_EPISODIC || d == FBA_DOM_TIGER_CONTINUOUS; }
__device__ __forceinline__ bool dom_is_ftiger(int d) { return d == FBA_DOM_FTIGER_EPISODIC || d == FBA_DOM_FTIGER_CONTINUOUS; }
__device__ __forceinline__ bool dom_is_episodic(int d) { return d == FBA_DOM_TIGER_EPISODIC || d == FBA_DOM_FTIGER_EPISODIC; }
__device__ __forceinline__ bool dom_is_grid(int d) { return d == FBA_DOM_GRIDWORLD; }
__device__ __forceinline__ bool dom_is_ca(int d) { return d == FBA_DOM_COLLISION_AVOID || d == FBA_DOM_COLLISION_AVOID_CENTERED; }
__device__ __forceinline__ bool dom_is_agr(int d) { return d == FBA_DOM_AGR; }
constexpr int AGR_N = 10;  // factory::makeEnvironment builds AGR(10) (Environment.cpp:32-33)
__device__ __forceinline__ bool dom_is_coffee(int d) { return d == FBA_DOM_COFFEE || d == FBA_DOM_COFFEE_BOUTILIER; }
__device__ __forceinline__ bool dom_is_sys(int d) { return d == FBA_DOM_SYSADMIN_INDEPENDENT || d == FBA_DOM_SYSADMIN_LINEAR; }

// SysAdmin::numFailingNeighbours (SysAdmin.cpp:221-246): linear topology, neighbours c-1 and c+1
__device__ __forceinline__ int sys_failing_neighbours(const Problem& P, int comp, int s)
{
    if (P.domain == FBA_DOM_SYSADMIN_INDEPENDENT) return 0;
    int n = 0;
    if (comp > 0 && !((s >> (comp - 1)) & 1)) n++;
    if (comp < P.sys->N - 1 && !((s >> (comp + 1)) & 1)) n++;
    return n;
}

// ---- collision avoidance helpers: state = (x*H + y)*H^n + project(obstacle rows) ----
__device__ __forceinline__ int ca_keep(const CADesc* ca, int y) { return max(0, min(ca->H - 1, y)); }
// row of obstacle k (0-based, most significant digit first) in the packed obstacle index
__device__ __forceinline__ int ca_obstacle(const CADesc* ca, int packed, int k)
{
    int v = packed;
    for (int j = ca->n - 1; j > k; --j) v /= ca->H;
    return v % ca->H;
}
__device__ __forceinline__ bool ca_crashed(const CADesc* ca, int s)
{
    const int H = ca->H, Hn = ca->Hn, x = s / (H * Hn), y = (s / Hn) % H;
    return x < ca->n && y == ca_obstacle(ca, s % Hn, x);
}

// ---- gridworld helpers: state index = x*N*G + y*G + g (GridWorld.cpp:329-340) ----
__device__ __forceinline__ bool gw_slow_at(const GridDesc* gw, int x, int y)
{
    bool slow = false;
    for (int i = 0; i < gw->nslow; ++i) slow |= (gw->slow[i][0] == x && gw->slow[i][1] == y);
    return slow;
}
__device__ __forceinline__ void gw_move(const GridDesc* gw, int a, int& x, int& y)  // applyMove :342-364
{
    const int N = gw->N;
    if (a == 0) { if (y != N - 1) ++y; }
    else if (a == 2) { if (y != 0) --y; }
    else if (a == 1) { if (x != N - 1) ++x; }
    else { if (x != 0) --x; }
}
__device__ __forceinline__ bool gw_on_goal(const GridDesc* gw, int s)
{
    const int N = gw->N, G = gw->G, g = s % G;
    return gw->goal[g][0] == s / (N * G) && gw->goal[g][1] == (s / G) % N;
}
// obsDisplProb :164-185 (float result, double intermediate products)
__device__ __forceinline__ float gw_obs_displ_prob(const GridDesc* gw, int loc, int observed)
{
    const int disp = abs(loc - observed);
    float res = (disp == 0) ? (float)(1 - .2) : (float)((double)gw->disp[disp] * .5);
    if (observed == gw->N - 1 || observed == 0)
        for (int i = disp + 1; i < gw->N; ++i) res = (float)((double)res + (double)gw->disp[i] * .5);
    return res;
}

// Tiger::sampleStartState (Tiger.cpp:16-19), FactoredTiger::sampleStartState
__device__ __forceinline__ int domain_start(const Problem& P, Rng& g)
{
    if (dom_is_tiger(P.domain)) return g.boolean() ? 0 : 1;
    if (dom_is_agr(P.domain)) return (2 * AGR_N + 1) * g.uniform_int(2 * AGR_N + 1) + AGR_N;  // AGR::sampleStartState :236-239: target at 0, goal uniform
    if (dom_is_sys(P.domain)) return P.S - 1;  // SysAdmin::sampleStartState :102-105: all computers on, no draw
    if (dom_is_ca(P.domain)) {  // sampleStartState :270-273 -> categoricalDistr::sample -> sampleFromMult<float>
        const CADesc* ca = P.ca;
        const double p = g.u01() * ca->start_total;
        float sum = 0;
        for (int k = 0; k < ca->start_cnt; ++k) {
            sum += ca->start_v;
            if (p < (double)sum) return ca->start_i0 + k;
        }
        return P.S - 1;
    }
    if (dom_is_grid(P.domain)) {  // GridWorld::sampleStartState :260-266 (start_locations = {{0,0}})
        (void)g.slow_int(0, 1);
        return g.slow_int(0, P.gw->G);
    }
    return g.uniform_int(P.S);
}

// Tiger::generateRandomAction (Tiger.cpp:21-25), FactoredTiger::generateRandomAction
template <class RNG>
__device__ __forceinline__ int domain_random_action(const Problem& P, RNG& g, int /*s*/)
{
    if (dom_is_grid(P.domain)) return g.slow_int(0, 4);  // GridWorld::generateRandomAction :220-226
    if (dom_is_coffee(P.domain)) return g.boolean() ? 1 : 0;  // CoffeeProblem::generateRandomAction :27-34
    return g.uniform_int(P.A);
}

// True dynamics: Tiger::step (Tiger.cpp:40-82), FactoredTiger::step (FactoredTiger.cpp:77-122).
// Note the draw order when opening a door: observation coin first, then the next state.
template <class RNG>
__device__ __forceinline__ bool domain_step(const Problem& P, RNG& g, int& s, int a, int& o, double& r)
{
    const int d = P.domain;
    if (dom_is_agr(d)) {  // AGR::step :247-305, no draws: state = (2n+1)(goal+n) + pos+n; help(-n..n) = 0..2n, work, observe
        constexpr int n = AGR_N;
        const int goal = s / (2 * n + 1) - n;
        int pos = s % (2 * n + 1) - n;
        bool helped = false;
        if (a <= 2 * n) {
            helped = a - n == goal && pos == goal;
            r = helped ? 100 : -100;
        } else r = (a == 2 * n + 1) ? -5 : -10;
        pos += max(-1, min(1, goal - pos));
        s = (2 * n + 1) * (goal + n) + pos + n;
        o = (a == 2 * n + 2) ? pos + n : 2 * n + 1;  // observe sees the NEW position, the rest "none"
        return helped;
    }
    if (dom_is_coffee(d)) {  // CoffeeProblem::step :67-148; bits: rains 1, umbrella 2, wet 4, has coffee 8, wants coffee 16
        const bool boutilier = d == FBA_DOM_COFFEE_BOUTILIER;
        const int st = s;
        r = (st & 4) ? -1 : -.5;
        if (st & 16) r += (st & 8) ? 2 : -2;
        if (a == 0) {  // GetCoffee: two draws whatever the version; always hears Want_Coffee
            if ((st & 1) && !(st & 2)) s |= 4;
            if (g.u01() < .9) s |= 8;
            if (g.u01() < (boutilier ? 0 : .9)) s &= ~16;
            o = 0;
        } else {  // CheckCoffee
            if (g.u01() < (boutilier ? 0 : .3)) s &= ~8;
            if (g.u01() < (boutilier ? 0 : .3)) s |= 16;
            o = (s & 16) ? ((g.u01() < .8) ? 0 : 1) : ((g.u01() < .9) ? 1 : 0);
        }
        return false;
    }
    if (dom_is_sys(d)) {  // SysAdmin::step :107-152
        const int N = P.sys->N, op = a >= N ? a - N : a;
        int index = s;
        for (int k = 0; k < N; ++k)  // one draw per computer (failing ones too); neighbours of the OLD state
            if (g.u01() > P.sys->keep[sys_failing_neighbours(P, k, s)]) index &= ~(1 << k);
        if (a >= N && g.u01() < (double).95f) index |= 1 << op;  // _reboot_success_rate
        s = index;
        o = ((g.u01() < (double).95f) == (((index >> op) & 1) != 0)) ? 1 : 0;  // _observe_prob; OPERATIONAL = 1
        r = (double)((int)__popc((unsigned)index) - (a >= N ? 1 : 0));     // operational computers - reboot cost (1.0f)
        return false;
    }
    if (dom_is_ca(d)) {  // CollisionAvoidance::step :236-270, moveObstacle :330-338, reward :196-208
        const CADesc* ca = P.ca;
        const int H = ca->H, n = ca->n, Hn = ca->Hn;
        const int x = s / (H * Hn), y = (s / Hn) % H, packed = s % Hn;
        const int nx = x - 1, ny = ca_keep(ca, y + a - 1);
        int nobs = 0, hit = -1;
        for (int k = 0; k < n; ++k) {  // obstacles move first, in order ...
            const double prob = g.u01();
            const int m = (prob < .5) ? 1 : (prob > .5 * (1 + .5)) ? 2 : 0;
            const int b = ca_keep(ca, ca_obstacle(ca, packed, k) + m - 1);
            nobs = nobs * H + b;
            if (k == nx) hit = b;
        }
        int oobs = 0;
        for (int k = 0; k < n; ++k) {  // ... then each is observed with rounded N(0,1) noise
            const double u = g.u01();
            int noise = 10;
            for (int j = 18; j >= 0; --j)
                if (u < ca->phi[j]) noise = j - 9;
            oobs = oobs * H + ca_keep(ca, ca_obstacle(ca, nobs, k) + noise);
        }
        s = (nx * H + ny) * Hn + nobs;
        o = oobs;
        const bool crashed = nx < n && nx >= 0 && ny == hit;
        r = crashed ? -1000 : (a == 1 ? 0 : -1);
        return crashed || nx == 0;
    }
    if (dom_is_grid(d)) {  // GridWorld::step :272-304, generateObservation :366-394
        const GridDesc* gw = P.gw;
        const int N = gw->N, G = gw->G;
        const int x = s / (N * G), y = (s / G) % N, gl = s % G;
        const bool ok = g.u01() < (gw_slow_at(gw, x, y) ? .15 : .95);
        int nx = x, ny = y, ng = gl;
        if (ok) gw_move(gw, a, nx, ny);
        const bool found = gw->goal[gl][0] == x && gw->goal[gl][1] == y;
        if (found) ng = g.slow_int(0, G);
        s = nx * N * G + ny * G + ng;
        r = found ? 1 : 0;
        const GlobalView dv{gw->disp};
        const int dx = sample_from_mult_f(g, dv, 0, N, 1.0);
        const int dy = sample_from_mult_f(g, dv, 0, N, 1.0);
        const int ox = g.boolean() ? max(nx - dx, 0) : min(nx + dx, N - 1);
        const int oy = g.boolean() ? max(ny - dy, 0) : min(ny + dy, N - 1);
        o = ox * N * G + oy * G + ng;
        return found;
    }
    if (dom_is_tiger(d)) {
        if (a == 2) {
            const bool correct = g.u01() < .85;
            r                  = -1;
            o                  = (correct != (s == 0)) ? 1 : 0;
        } else {
            r = (a == s) ? 10 : -100;
            o = g.boolean() ? 1 : 0;
            s = g.boolean() ? 1 : 0;
        }
    } else {
        const int loc = (s < P.S / 2) ? 0 : 1;
        if (a == 2) {
            const bool correct = g.u01() < .85;
            r                  = -1;
            o                  = (correct != (loc == 0)) ? 1 : 0;
        } else {
            r = (a == loc) ? 10 : -100;
            o = g.boolean() ? 0 : 1;
            s = g.uniform_int(P.S);
        }
    }
    return dom_is_episodic(d) && a != 2;
}

// Tiger / FactoredTiger::computeObservationProbability (Tiger.cpp:27-38)
__device__ __forceinline__ double domain_obs_prob(const Problem& P, int o, int a, int new_s)
{
    if (dom_is_coffee(P.domain)) {  // CoffeeProblem::computeObservationProbability :44-60
        if (a == 0) return o == 0 ? 1 : 0;
        if (new_s & 16) return o == 0 ? .8 : 1 - .8;
        return o == 1 ? .9 : 1 - .9;
    }
    if (dom_is_sys(P.domain)) {  // SysAdmin::computeObservationProbability :154-165 (float results)
        const int op = a >= P.sys->N ? a - P.sys->N : a;
        return (o == ((new_s >> op) & 1)) ? (double).95f : (double)(1 - .95f);
    }
    if (dom_is_ca(P.domain)) {  // CollisionAvoidance::computeObservationProbability :216-229
        const CADesc* ca = P.ca;
        double p = 1;
        for (int k = 0; k < ca->n; ++k) p *= ca->err[abs(ca_obstacle(ca, new_s % ca->Hn, k) - ca_obstacle(ca, o, k))];
        return p;
    }
    if (dom_is_grid(P.domain)) {  // GridWorld::computeObservationProbability :236-250 (goal part ignored)
        const GridDesc* gw = P.gw;
        const int N = gw->N, G = gw->G;
        return (double)(gw_obs_displ_prob(gw, new_s / (N * G), o / (N * G)) * gw_obs_displ_prob(gw, (new_s / G) % N, (o / G) % N));
    }
    if (a != 2) return .5;
    const int loc = dom_is_tiger(P.domain) ? new_s : ((new_s < P.S / 2) ? 0 : 1);
    return (loc == o) ? .85 : .15;
}

// BADomainExtension::terminal / reward (TigerBAExtension.cpp:21-44, FactoredTigerBAExtension.cpp):
// the reward is looked up with the PRE-state s.
__device__ __forceinline__ bool ext_terminal(const Problem& P, int s, int a, int ns)
{
    if (dom_is_ca(P.domain)) return ca_crashed(P.ca, ns) || ns / (P.ca->H * P.ca->Hn) == 0;  // CollisionAvoidanceBAExtension.cpp:53-63 (NEW state)
    if (dom_is_grid(P.domain)) return gw_on_goal(P.gw, s);  // GridWorldBAExtension.cpp:74-83
    if (dom_is_sys(P.domain)) return false;                 // SysAdminBAExtension.cpp:31-37
    return dom_is_episodic(P.domain) && a != 2;
}
__device__ __forceinline__ double ext_reward(const Problem& P, int s, int a, int ns)
{
    if (dom_is_ca(P.domain)) return ca_crashed(P.ca, ns) ? -1000 : (a == 1 ? 0 : -1);  // CollisionAvoidanceBAExtension.cpp:65-82
    if (dom_is_grid(P.domain)) return gw_on_goal(P.gw, s) ? 1 : 0;  // GridWorldBAExtension.cpp:85-99
    if (dom_is_sys(P.domain)) return (double)((int)__popc((unsigned)ns) - (a >= P.sys->N ? 1 : 0));  // SysAdminBAExtension.cpp:39-50 (NEW state)
    if (a == 2) return -1;
    const int loc = dom_is_tiger(P.domain) ? s : ((s < P.S / 2) ? 0 : 1);
    return (a == loc) ? 10 : -100;
}

// ---------------------------------------------------------------------------------------------
// simulator.step for the three simulators.
// BAPOMDP::step (BAPOMDP.cpp:111-143) with BAFlatModel::sampleStateIndex /
// sampleObservationIndex (BAFlatModel.cpp:83-104), expected-Dirichlet method.
// The count increment of UpdateCounts mode is returned as two blob indices (inc0, inc1) so the
// caller decides where the +1 lands (in place for importance sampling, in the copy for
// rejection sampling).
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// `regular` Dirichlet mode (reference src/utils/random.cpp:40-44, 146-304), bug-compatible: the
// reference's randomLong32 yields 31-bit values only, so its ziggurat "normal" is a half-normal and
// its gammas are biased upward -- reproduced, not repaired.
// log / exp are fixed sequences of IEEE operations, det_log / det_exp (argument reduction +
// minimax polynomial; the CPU checker carries the same sequences): bit-identical on host and device, < 2 ulp from
// libm.  The ziggurat tables are built on the host with libm, as rnd::initiate() does.
// ---------------------------------------------------------------------------------------------
struct ZigDesc {
    uint32_t ul[128];
    double wn[128], fn[128];
};
constexpr int MAXROW = 16;  // longest Dirichlet row the regular mode samples in registers/scratch

__device__ __forceinline__ double det_log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    if (!(x > 0)) return x == 0 ? -HUGE_VAL : NAN;
    uint64_t ix = (uint64_t)__double_as_longlong(x);
    int k = 0;
    if ((ix >> 52) == 0) { x *= 18014398509481984.0; ix = (uint64_t)__double_as_longlong(x); k = -54; }
    k += (int)(ix >> 52) - 1023;
    ix = (ix & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    x  = __longlong_as_double((long long)ix);
    if (x > 1.4142135623730951) { x *= 0.5; k += 1; }
    const double f = x - 1.0, s = f / (2.0 + f), z = s * s, w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1, hfsq = 0.5 * f * f, dk = (double)k;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

__device__ __forceinline__ double det_exp(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 inv_ln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                 P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x > 709.0) return HUGE_VAL;
    if (x < -745.0) return 0.0;
    const int k = (int)(inv_ln2 * x + (x < 0 ? -0.5 : 0.5));
    double t = (double)k;
    const double hi = x - t * ln2_hi, lo = t * ln2_lo, r = hi - lo;
    t = r * r;
    const double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    if (k < -1021) return y * __longlong_as_double((long long)((uint64_t)(k + 1000 + 1023) << 52)) * 9.33263618503218878990e-302;
    return y * __longlong_as_double((long long)((uint64_t)(k + 1023) << 52));
}

__device__ __forceinline__ double det_pow(double x, double y)
{
    if (x == 0) return y > 0 ? 0.0 : HUGE_VAL;
    return det_exp(y * det_log(x));
}

// lgamma as one fixed sequence of IEEE operations (the CPU checker carries the same one): x >= 1 is shifted up to >= 16
// with lgamma(x) = lgamma(x + n) - log(x (x + 1) ... (x + n - 1)), then Stirling's series to 1 / x^13.  Within a few
// ulp of libm for the counts a run can reach.  For DBNNode::LogBDScore (DBNNode.cpp:82-117), the score the
// structure-learning beliefs accept and reject models by.
__device__ __forceinline__ double det_lgamma(double x)
{
    double prod = 1.0, shift = 0.0;
    while (x < 16.0) {
        prod *= x;
        x += 1.0;
        if (prod > 1e250) { shift += det_log(prod); prod = 1.0; }
    }
    shift += det_log(prod);
    const double xi = 1.0 / x, x2 = xi * xi;
    const double series = xi * (1.0 / 12.0 + x2 * (-1.0 / 360.0 + x2 * (1.0 / 1260.0 + x2 * (-1.0 / 1680.0 + x2 * (1.0 / 1188.0 +
                          x2 * (-691.0 / 360360.0 + x2 * (1.0 / 156.0)))))));
    return ((x - 0.5) * det_log(x) - x + 0.91893853320467274178) + series - shift;
}
// rnd::math::logGamma (random.cpp:127-135): 0 below 1
__device__ __forceinline__ double log_gamma(double x) { return x < 1 ? 0.0 : det_lgamma(x); }

// randomLong (random.cpp:40-44): 31 random bits, never negative
__device__ __forceinline__ long long random_long(Rng& g) { return (long long)(g.next64() >> 33); }

// randomNormal + normalRejectFix (random.cpp:146-187)
__device__ __noinline__ double random_normal(const ZigDesc* z, Rng& g)
{
    long long h = random_long(g);
    uint32_t i  = (uint32_t)(h & 127);
    if ((uint64_t)h < z->ul[i]) return (double)h * z->wn[i];
    const double r = 3.442620, r_inverse = 0.2904764;
    for (;;) {
        double x = (double)h * z->wn[i];
        if (i == 0) {
            double y;
            do {
                x = -det_log(g.u01()) * r_inverse;
                y = -det_log(g.u01());
            } while (y + y < x * x);
            return (h > 0) ? r + x : -r - x;
        }
        if (z->fn[i] + g.u01() * (z->fn[i - 1] - z->fn[i]) < det_exp(-.5 * x * x)) return x;
        h = random_long(g);
        i = (uint32_t)(h & 127);
        if ((uint64_t)h < z->ul[i]) return (double)h * z->wn[i];
    }
}

// rnd::sample::gamma (random.cpp:189-213): Marsaglia-Tsang; shape < 1 goes through shape + 1
__device__ __noinline__ double sample_gamma(const ZigDesc* z, Rng& g, double shape)
{
    const double sh = shape < 1. ? shape + 1 : shape;
    const double d = sh - 1. / 3., cc = 1. / sqrt(9. * d);
    double res;
    for (;;) {
        double x, v;
        do {
            x = random_normal(z, g);
            v = 1.0 + cc * x;
        } while (v <= 0.0);
        v = v * v * v;
        const double u = g.u01(), x2 = x * x;
        if (u < 1.0 - 0.0331 * x2 * x2) { res = d * v; break; }
        if (det_log(u) < .5 * x2 + d * (1. - v + det_log(v))) { res = d * v; break; }
    }
    if (shape < 1.) res = res * det_pow(g.u01(), 1 / shape);
    return res;
}

// sampleFromSampledMult (random.cpp:217-242)
template <class View>
__device__ __forceinline__ int sample_sampled_mult(const ZigDesc* z, Rng& g, const View& row, int off, int n)
{
    double probs[MAXROW];
    double sum = 0;
    for (int i = 0; i < n; ++i) {
        probs[i] = sample_gamma(z, g, (double)row.at(off + i));
        sum      = (i == 0) ? probs[0] : sum + probs[i];
    }
    const double p = g.u01() * sum;
    double acc     = probs[0];
    for (int i = 1; i < n; ++i) {
        if (p < acc) return i - 1;
        acc += probs[i];
    }
    return n - 1;
}

// sampleMult(row)[o] (random.cpp:281-304): gammas stored as float, float sum, float division
template <class View>
__device__ __forceinline__ double sample_mult_at(const ZigDesc* z, Rng& g, const View& row, int off, int n, int o)
{
    float sum = 0, mine = 0;
    for (int i = 0; i < n; ++i) {
        const float v = (float)sample_gamma(z, g, (double)row.at(off + i));
        sum += v;
        if (i == o) mine = v;
    }
    return (double)(mine / sum);
}

// one Dirichlet-row draw in the configured mode
template <bool REG, class View>
__device__ __forceinline__ int sample_row(const Problem& P, Rng& g, const View& row, int off, int n)
{
    if (REG) return sample_sampled_mult(P.zig, g, row, off, n);
    return sample_expected_mult(g, row, off, n);
}

// ---- factored model ---------------------------------------------------------------------------
// feature values travel packed, 8 bits each, so no runtime-indexed register arrays are needed
__device__ __forceinline__ uint64_t pack_features(int v, const int32_t* step, int n)
{
    if (n == 1) return (uint64_t)v;
    uint64_t p = 0;
#pragma unroll
    for (int i = 0; i < MAXF; ++i)
        if (i < n) { p |= (uint64_t)(v / step[i]) << (8 * i); v = v % step[i]; }  // indexing::projectUsingStepSize
    return p;
}
__device__ __forceinline__ int feat(uint64_t p, int i) { return (int)((p >> (8 * i)) & 0xffu); }

// One node description into registers with three 16-byte loads (LDS or global); its byte arrays are
// then read with compile-time shifts instead of one byte load per parent.
struct NodeRegs {
    int32_t off, out, nmax, var;
    uint32_t fixed_mask;
    uint32_t maxp_lo, maxp_hi, psz_lo, psz_hi;
    __device__ __forceinline__ int parent(int j) const { return (int)(((j < 4 ? maxp_lo : maxp_hi) >> (8 * (j & 3))) & 0xffu); }
    __device__ __forceinline__ int psize(int j) const { return (int)(((j < 4 ? psz_lo : psz_hi) >> (8 * (j & 3))) & 0xffu); }
};
__device__ __forceinline__ NodeRegs load_node(const FNode* p)
{
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 a = q[0], b = q[1], c = q[2];
    NodeRegs r;
    r.off = (int32_t)a.x; r.out = (int32_t)a.y; r.nmax = (int32_t)a.z; r.var = (int32_t)a.w;
    r.fixed_mask = b.x; r.maxp_lo = b.y; r.maxp_hi = b.z; r.psz_lo = b.w; r.psz_hi = c.x;
    return r;
}
template <class View>
__device__ __forceinline__ uint32_t node_mask(const FDesc* fd, const NodeRegs& nd, const View& cnt)
{
    return nd.var >= 0 ? __float_as_uint(cnt.at(fd->ncounts + nd.var)) : nd.fixed_mask;
}
__device__ __forceinline__ int node_row(const FDesc*, const NodeRegs& nd, uint32_t mask, uint64_t fv)
{
    int idx = 0;
#pragma unroll
    for (int j = 0; j < MAXF; ++j)
        if (j < nd.nmax && ((mask >> j) & 1u)) idx = idx * nd.psize(j) + feat(fv, nd.parent(j));
    return nd.off + idx * nd.out;
}
template <class View>
__device__ __forceinline__ uint32_t node_mask(const FDesc* fd, const FNode& nd, const View& cnt)
{
    return nd.var >= 0 ? __float_as_uint(cnt.at(fd->ncounts + nd.var)) : nd.fixed_mask;
}
// DBNNode::cptIndex(graph input, 0): mixed radix over the node's actual parents
__device__ __forceinline__ int node_row(const FDesc* fd, const FNode& nd, uint32_t mask, uint64_t fv)
{
    int idx = 0;
#pragma unroll
    for (int j = 0; j < MAXF; ++j)
        if (j < nd.nmax && ((mask >> j) & 1u)) idx = idx * nd.psz[j] + feat(fv, nd.maxp[j]);
    return nd.off + idx * nd.out;
}

// BAPOMDP::step over BABNModel (BABNModel.cpp:292-325) + the indices incrementCountsOf would
// touch (:354-382).  Quirk kept: observation CPTs are incremented at the row of
// the PREVIOUS state's parent values.
template <bool REG, class View, class Sink>
__device__ __forceinline__ bool fact_step(const Problem& P, Rng& g, const View& cnt, int& s, int a, int& o, double& r, const Sink& inc)
{
    const FDesc* fd   = P.fd;
    const int FS = fd->FS, FO = fd->FO;
    const uint64_t fv = pack_features(s, fd->Sstep, FS);
    uint64_t nf = 0;
    int ns = 0;
#pragma unroll
    for (int f = 0; f < MAXF; ++f)
        if (f < FS) {
            const NodeRegs nd = load_node(&fd->nodes[a * FS + f]);
            const int row   = node_row(fd, nd, node_mask(fd, nd, cnt), fv);
            const int v     = sample_row<REG>(P, g, cnt, row, nd.out);
            inc.add(f, row + v);
            nf |= (uint64_t)v << (8 * f);
            ns = ns * fd->Ssz[f] + v;  // indexing::project
        }
    int ob = 0;
#pragma unroll
    for (int f = 0; f < MAXF; ++f)
        if (f < FO) {
            const NodeRegs nd   = load_node(&fd->nodes[P.A * FS + a * FO + f]);
            const uint32_t mask = node_mask(fd, nd, cnt);
            const int v         = sample_row<REG>(P, g, cnt, node_row(fd, nd, mask, nf), nd.out);
            ob = ob * fd->Osz[f] + v;
            inc.add(FS + f, node_row(fd, nd, mask, fv) + v);
        }
    o            = ob;
    const bool t = ext_terminal(P, s, a, ns);
    r            = ext_reward(P, s, a, ns);
    s            = ns;
    return t;
}

// fact_step for the factored-tiger FBA-POMDP with FS = K + 1 binary state features, everything the
// generic code reads from the model description restated from build_ftiger_factored_prior's layout:
//   T(open a, f)   no parents   at (a*FS + f)*2          T(listen, f)  parent f   at 4*FS + 4*f + 2*value
//   O(open a)      no parents   at 8*FS + 2*a            O(listen)     parents = the particle's mask over the
//   features (feature 0 first), rows at 8*FS + 4 + 2*row;  mask word at 8*FS + 4 + (2 << FS).
// Feature f of state s is bit FS-1-f.  Same draws, same order, same increments as fact_step.
template <int FS, class View, class Sink>
__device__ __forceinline__ bool ftiger_step(const Problem& P, Rng& g, const View& cnt, int& s, int a, int& o, double& r, const Sink& inc)
{
    constexpr int OBS = 8 * FS, MASKW = 8 * FS + 4 + (2 << FS);
    const int loc = (s >> (FS - 1)) & 1;  // feature 0: the tiger's door
    int ns = 0;
#pragma unroll
    for (int f = 0; f < FS; ++f) {
        const int v   = (s >> (FS - 1 - f)) & 1;
        const int row = a == 2 ? 4 * FS + 4 * f + 2 * v : (a * FS + f) * 2;
        const int nv  = sample_expected_mult(g, cnt, row, 2);
        inc.add(f, row + nv);
        ns = ns * 2 + nv;
    }
    int row_new, row_old;  // the observation row of the new state (sampled) and of the old one (incremented: App. A #6)
    if (a == 2) {
        const uint32_t mask = __float_as_uint(cnt.at(MASKW));
        int in = 0, io = 0;
#pragma unroll
        for (int f = 0; f < FS; ++f)
            if ((mask >> f) & 1u) {
                in = in * 2 + ((ns >> (FS - 1 - f)) & 1);
                io = io * 2 + ((s >> (FS - 1 - f)) & 1);
            }
        row_new = OBS + 4 + 2 * in;
        row_old = OBS + 4 + 2 * io;
    } else {
        row_new = row_old = OBS + 2 * a;
    }
    o = sample_expected_mult(g, cnt, row_new, 2);
    inc.add(FS, row_old + o);
    const bool t = dom_is_episodic(P.domain) && a != 2;
    r = a == 2 ? -1.0 : (a == loc ? 10.0 : -100.0);
    s = ns;
    return t;
}

// ftiger_step on packed records (Problem::ft_packed, PackedFtigerView's format): every Dirichlet row of this model has two
// cells and starts at an even cell, so a row is ONE word of the record -- its two uint16 increment counts -- plus two
// prior values that the action, the parent value and the structure bits determine.  `words.at(w)` = word w of the
// record.  Same draws, same order, same sums as ftiger_step on the dense table.
template <int FS, class Words, class Sink>
__device__ __forceinline__ bool ftiger_step_packed(const Problem& P, Rng& g, const Words& words, int& s, int a, int& o, double& r, const Sink& inc)
{
    constexpr int OBS = 8 * FS, MASKW = (8 * FS + 4 + (2 << FS)) / 2;
    const int loc = (s >> (FS - 1)) & 1;
    const bool listen = a == 2;
    int ns = 0;
#pragma unroll
    for (int f = 0; f < FS; ++f) {
        const int v   = (s >> (FS - 1 - f)) & 1;
        const int row = listen ? 4 * FS + 4 * f + 2 * v : (a * FS + f) * 2;
        const uint32_t w = __float_as_uint(words.at(row >> 1));
        const float c0 = ((listen && v) ? 0.f : 5000.f) + (float)(w & 0xffffu), c1 = ((listen && !v) ? 0.f : 5000.f) + (float)(w >> 16);
        const double p = g.u01() * ((double)c0 + (double)c1);   // sampleFromExpectedMult on a row of two
        const int nv   = (p < (double)c0) ? 0 : 1;
        inc.add(f, row + nv);
        ns = ns * 2 + nv;
    }
    int row_new, row_old;
    float p0 = 5000.f, p1 = 5000.f;
    if (listen) {
        const uint32_t mask = __float_as_uint(words.at(MASKW));
        int in = 0, io = 0;
#pragma unroll
        for (int f = 0; f < FS; ++f)
            if ((mask >> f) & 1u) {
                in = in * 2 + ((ns >> (FS - 1 - f)) & 1);
                io = io * 2 + ((s >> (FS - 1 - f)) & 1);
            }
        row_new = OBS + 4 + 2 * in;
        row_old = OBS + 4 + 2 * io;
        if (mask & 1u) {  // informed by the tiger's door: feature 0 of the NEW state
            const bool left = ((ns >> (FS - 1)) & 1) == 0;
            p0 = left ? P.ft_acc : P.ft_inacc;
            p1 = left ? P.ft_inacc : P.ft_acc;
        } else p0 = p1 = P.ft_unif;
    } else {
        row_new = row_old = OBS + 2 * a;
    }
    {
        const uint32_t w = __float_as_uint(words.at(row_new >> 1));
        const float c0 = p0 + (float)(w & 0xffffu), c1 = p1 + (float)(w >> 16);
        const double p = g.u01() * ((double)c0 + (double)c1);
        o = (p < (double)c0) ? 0 : 1;
    }
    inc.add(FS, row_old + o);
    const bool t = dom_is_episodic(P.domain) && a != 2;
    r = a == 2 ? -1.0 : (a == loc ? 10.0 : -100.0);
    s = ns;
    return t;
}

// fact_step for the gridworld FBA-POMDP, with what the generic code reads from the model description
// restated from build_gridworld_factored_prior's layout (features x, y, goal; XY = N*N*G*N, GG = N*N*G*G):
//   T(a, x) at a*(2XY+GG), T(a, y) at +XY: parents {x, y} (mask 3: row (x*N+y)) or {x, y, goal} (mask 7: row
//   ((x*N+y)*G+goal)), N entries; T(a, goal) at +2XY: row ((x*N+y)*G+goal), G entries;
//   O(a, f) after all T nodes, per action N*N + N*N + G*G: row = value of feature f; masks at ncounts + 2a + f.
// Same draws, same order, same increments as fact_step (observation rows incremented at the OLD state's values).
template <class View, class Sink>
__device__ __forceinline__ bool gridworld_fact_step(const Problem& P, Rng& g, const View& cnt, int& s, int a, int& o, double& r, const Sink& inc)
{
    const GridDesc* gw = P.gw;
    const int N = gw->N, G = gw->G, A = P.A;
    const int XY = N * N * G * N, GG = N * N * G * G, NN = N * N;
    const int x = s / (N * G), y = (s / G) % N, gl = s % G;
    const int tbase = a * (2 * XY + GG), obase = A * (2 * XY + GG) + a * (2 * NN + G * G);
    const int ncounts = A * (2 * XY + GG) + A * (2 * NN + G * G);
    const int cell = x * N + y;
    const uint32_t mx = __float_as_uint(cnt.at(ncounts + 2 * a)), my = __float_as_uint(cnt.at(ncounts + 2 * a + 1));
    const int rx = tbase + ((mx & 4u) ? cell * G + gl : cell) * N;
    const int ry = tbase + XY + ((my & 4u) ? cell * G + gl : cell) * N;
    const int rg = tbase + 2 * XY + (cell * G + gl) * G;
    const int nx = sample_expected_mult(g, cnt, rx, N);
    inc.add(0, rx + nx);
    const int ny = sample_expected_mult(g, cnt, ry, N);
    inc.add(1, ry + ny);
    const int ng = sample_expected_mult(g, cnt, rg, G);
    inc.add(2, rg + ng);
    const int ns = (nx * N + ny) * G + ng;
    const int ox = sample_expected_mult(g, cnt, obase + nx * N, N);
    inc.add(3, obase + x * N + ox);
    const int oy = sample_expected_mult(g, cnt, obase + NN + ny * N, N);
    inc.add(4, obase + NN + y * N + oy);
    const int og = sample_expected_mult(g, cnt, obase + 2 * NN + ng * G, G);
    inc.add(5, obase + 2 * NN + gl * G + og);
    o = (ox * N + oy) * G + og;
    const bool t = ext_terminal(P, s, a, ns);
    r            = ext_reward(P, s, a, ns);
    s            = ns;
    return t;
}

// fact_step for the collision-avoidance FBA-POMDP with the correct-graph prior (no per-particle masks:
// fd->nvar == 0, every node has the one parent it should), layout of build_ca_factored_prior restated:
//   per action  T(x) W*W at +0, T(y) H*H at +W*W, T(obstacle k) H*H at +W*W + H*H*(1+k);  row = own value
//   O(a, k) H*H after all T nodes at a*n*H*H + k*H*H;  row = the obstacle's (new) position.
// Same draws, same order, same increments as fact_step.
template <class View, class Sink>
__device__ __forceinline__ bool ca_fact_step(const Problem& P, Rng& g, const View& cnt, int& s, int a, int& o, double& r, const Sink& inc)
{
    const CADesc* ca = P.ca;
    const int W = ca->W, H = ca->H, n = ca->n, Hn = ca->Hn, A = P.A;
    const int tsize = W * W + H * H * (1 + n), tbase = a * tsize, obase = A * tsize + a * n * H * H;
    const int x = s / (H * Hn), y = (s / Hn) % H, packed = s % Hn;
    const int rx = tbase + x * W;
    const int nx = sample_expected_mult(g, cnt, rx, W);
    inc.add(0, rx + nx);
    const int ry = tbase + W * W + y * H;
    const int ny = sample_expected_mult(g, cnt, ry, H);
    inc.add(1, ry + ny);
    int nobs = 0;
#pragma unroll
    for (int k = 0; k < MAXF - 2; ++k)
        if (k < n) {
            const int b   = ca_obstacle(ca, packed, k);
            const int row = tbase + W * W + H * H * (1 + k) + b * H;
            const int nb  = sample_expected_mult(g, cnt, row, H);
            inc.add(2 + k, row + nb);
            nobs = nobs * H + nb;
        }
    const int ns = (nx * H + ny) * Hn + nobs;
    int oobs = 0;
#pragma unroll
    for (int k = 0; k < MAXF - 2; ++k)
        if (k < n) {
            const int nb = ca_obstacle(ca, nobs, k), b = ca_obstacle(ca, packed, k);
            const int ob = sample_expected_mult(g, cnt, obase + k * H * H + nb * H, H);
            inc.add(2 + n + k, obase + k * H * H + b * H + ob);  // incremented at the OLD position (App. A #6)
            oobs = oobs * H + ob;
        }
    o = oobs;
    const bool t = ext_terminal(P, s, a, ns);
    r            = ext_reward(P, s, a, ns);
    s            = ns;
    return t;
}

// fact_step for the sysadmin FBA-POMDP with the prior's own graph (no per-particle masks: fd->nvar == 0),
// layout of build_sysadmin_factored_prior restated.  Feature c (the model's computer c) is bit N-1-c of s.
//   independent: T(a, c) parent {c}: 4 floats at a*4N + 4c, row = value of c
//   linear:      T(a, c) parents {c-1, c, c+1} clipped: c = 0 -> 8 floats at +0 (N = 1: 4), 0 < c < N-1 -> 16
//                floats at 8 + 16(c-1), c = N-1 -> 8 floats at 8 + 16(N-2); row = parent values, last fastest
//   O(a) parent {a mod N}: 4 floats after all T nodes at 4a
// Same draws, same order, same increments as fact_step.
template <class View, class Sink>
__device__ __forceinline__ bool sysadmin_fact_step(const Problem& P, Rng& g, const View& cnt, int& s, int a, int& o, double& r, const Sink& inc)
{
    const int N = P.sys->N, A = P.A;
    const bool linear = P.domain == FBA_DOM_SYSADMIN_LINEAR;
    const int tsize = !linear ? 4 * N : (N == 1 ? 4 : 16 * N - 16);
    const int tbase = a * tsize, obase = A * tsize + 4 * a;
    int ns = 0;
#pragma unroll
    for (int c = 0; c < MAXF; ++c)
        if (c < N) {
            const int v = (s >> (N - 1 - c)) & 1;
            int off, row;
            if (!linear || N == 1) { off = 4 * c; row = v; }
            else {
                const int vl = c > 0 ? (s >> (N - c)) & 1 : 0, vr = c < N - 1 ? (s >> (N - 2 - c)) & 1 : 0;
                if (c == 0) { off = 0; row = v * 2 + vr; }
                else if (c == N - 1) { off = 8 + 16 * (N - 2); row = vl * 2 + v; }
                else { off = 8 + 16 * (c - 1); row = (vl * 2 + v) * 2 + vr; }
            }
            const int cell = tbase + off + 2 * row;
            const int nv   = sample_expected_mult(g, cnt, cell, 2);
            inc.add(c, cell + nv);
            ns = ns * 2 + nv;
        }
    const int f = a % N;  // the operated computer's feature
    o = sample_expected_mult(g, cnt, obase + 2 * ((ns >> (N - 1 - f)) & 1), 2);
    inc.add(N, obase + 2 * ((s >> (N - 1 - f)) & 1) + o);  // incremented at the OLD state's value (App. A #6)
    const bool t = ext_terminal(P, s, a, ns);
    r            = ext_reward(P, s, a, ns);
    s            = ns;
    return t;
}

// ---------------------------------------------------------------------------------------------
// History particles (Problem::hist): the gridworld FBA-POMDP particle as "shared prior + what this particle
// has added to it".  A Bayes-adaptive particle never changes a count except by the "+1"s of
// BABNModel::incrementCountsOf (BABNModel.cpp:354-382) -- six per real step here, one per DBN node of the step's
// action, at cells that (s, a, s', o') determine -- so instead of the 191 KB count table (N = 7) a record holds
//   word 0   the domain state            word 1   bit 2a + f: node T(a, f), f in {x, y}, has the goal as third parent;
//                                                 bits 16..25: the state again, as hist_pack(x, y, goal)
//   word 2 + j   entry j = one real step of the run: s | s' << 10 | o' << 20, each as hist_pack (x 3 bits, y 3 bits,
//                goal 4 bits: N <= 8), the entries of action 0 first, then action 1, ...
// and a count is prior[cell] + (number of entries that incremented the cell).  All particles of a slot have taken
// the same actions, so where an action's entries start and how many there are is one word per SLOT
// (DeviceState::hist_cnt); a belief update inserts its entry at the end of its action's group.  The reference's
// own copy-on-write table (BAFlatModel.cpp:185-217) is the same idea at row granularity.  The engine only uses
// this form when prior[k] + j, j <= hist_cap + 1, is for every k the float that j additions of 1.0f reach (checked
// on the host), so a row read through the history is bit for bit the row of the dense table.
// One simulated step = two passes over the entries of its action (the transition rows of (s, a), then the
// observation rows of (a, s')); an entry costs a few compares and adds, no memory beyond its own 4 bytes.
// The prior tables sit in HBM once per context (L2-resident), rows padded to a multiple of four floats so that
// a row is two or three aligned 16-byte loads:
//   hist_base: per action  T(x) [N*N*G rows of NS]  T(y) [same]  T(goal) [N*N*G rows of GS], then per action
//              O(x) [N rows of NS]  O(y) [N rows of NS]  O(goal) [G rows of GS];   NS = N rounded up to 4, GS likewise
//              (an x / y node without the goal parent uses its first N*N rows, as in the dense max layout)
//   hist_alt:  per action and x / y node, the N*N*G rows of NS a particle with the goal parent starts from
// ---------------------------------------------------------------------------------------------
constexpr int HIST_MAX_CAP = 126;  // entries per record: 2 + cap words <= SEARCH_STAGE_WORDS, counts per cell and per action fit 8 bits
constexpr int HIST_MAX_N   = 8;    // x, y in 3 bits
constexpr int HIST_QUAD    = 4;    // lanes that share one tree in search_hist_kernel
__host__ __device__ __forceinline__ uint32_t hist_pack(int x, int y, int g) { return (uint32_t)x | ((uint32_t)y << 3) | ((uint32_t)g << 6); }
__host__ __device__ __forceinline__ int hist_x(uint32_t sp) { return (int)(sp & 7u); }
__host__ __device__ __forceinline__ int hist_y(uint32_t sp) { return (int)((sp >> 3) & 7u); }
__host__ __device__ __forceinline__ int hist_g(uint32_t sp) { return (int)((sp >> 6) & 15u); }
__host__ __device__ __forceinline__ uint32_t hist_entry(uint32_t s, uint32_t ns, uint32_t o) { return s | (ns << 10) | (o << 20); }
// feature f of a packed state: shift and mask
__host__ __device__ __forceinline__ int hist_field(uint32_t sp, int f) { return (int)((sp >> (3 * f)) & (f == 2 ? 15u : 7u)); }
// per-slot group bookkeeping (DeviceState::hist_cnt): entries of action a, 8 bits each
__host__ __device__ __forceinline__ int hist_count(uint32_t cnt, int a) { return (int)((cnt >> (8 * a)) & 0xffu); }
__host__ __device__ __forceinline__ int hist_offset(uint32_t cnt, int a)
{
    int off = 0;
    for (int k = 0; k < 4; ++k) off += k < a ? hist_count(cnt, k) : 0;
    return off;
}
__host__ __device__ __forceinline__ int hist_total(uint32_t cnt) { return hist_count(cnt, 0) + hist_count(cnt, 1) + hist_count(cnt, 2) + hist_count(cnt, 3); }

// Words between the records of a slot whose histories hold `total` entries: 64 bytes up to 14 entries, 128 up to 30, the whole record (Cs)
// beyond.  A belief update reads every record of a slot and its resampling gather reads N of them at random and writes N: at a fixed 176-byte
// stride those three passes move 176-byte records of which 8 + 4 * total bytes are alive, and a random record straddles two 128-byte lines.
// Every update rebuilds the slot's records in another buffer anyway, so it writes them at the stride of total + 1.
__host__ __device__ __forceinline__ int hist_stride(const Problem& P, int total)
{
    if (!P.hist_compact) return P.Cs;
    const int w = 2 + total;
    return (w <= 16 && P.Cs > 16) ? 16 : ((w <= 32 && P.Cs > 32) ? 32 : P.Cs);
}
// the same from the 16-byte pieces that hold a record of `total` entries, n4 = (total + 5) >> 2 (total <= 14 <=> n4 <= 4, total <= 30 <=> n4 <= 8)
__host__ __device__ __forceinline__ int hist_stride_of_pieces(const Problem& P, int n4)
{
    if (!P.hist_compact) return P.Cs;
    return (n4 <= 4 && P.Cs > 16) ? 16 : ((n4 <= 8 && P.Cs > 32) ? 32 : P.Cs);
}

// where the rows of the padded tables start (host and device)
struct HistLayout {
    int N, G, A, NS, GS, XY, GG, ON, OG, tstride, ostride, obase0, total, alt_total;
    __host__ __device__ HistLayout(int n, int g, int a) : N(n), G(g), A(a)
    {
        NS = (N + 3) & ~3; GS = (G + 3) & ~3;
        XY = N * N * G * NS; GG = N * N * G * GS; ON = N * NS; OG = G * GS;
        tstride = 2 * XY + GG; ostride = 2 * ON + OG;
        obase0 = A * tstride; total = obase0 + A * ostride; alt_total = A * 2 * XY;
    }
    // T(a, f) row of parent values (cell = x*N + y, goal); f = 0, 1: `with_goal` = the particle's structure bit
    __host__ __device__ int t_row(int a, int f, bool with_goal, int cell, int gl) const
    {
        if (f == 2) return a * tstride + 2 * XY + (cell * G + gl) * GS;
        return a * tstride + f * XY + (with_goal ? cell * G + gl : cell) * NS;
    }
    __host__ __device__ int alt_row(int a, int f, int cell, int gl) const { return (a * 2 + f) * XY + (cell * G + gl) * NS; }
    // O(a, f) row of the feature's value v
    __host__ __device__ int o_row(int a, int f, int v) const { return obase0 + a * ostride + (f == 2 ? 2 * ON + v * GS : f * ON + v * NS); }
};

// row slots of the deduplicated tables (Problem::hist_lds): per action the x, y and goal transition rows (N*N*G slots each; an x / y node
// without the goal parent uses the first N*N), then per action and x / y node the rows with the goal parent, then the observation rows
struct HistRowIds {
    int N, G, RR, ALT0, OB0, total;
    __host__ __device__ HistRowIds(int n, int g, int a) : N(n), G(g)
    {
        RR = N * N * G; ALT0 = a * 3 * RR; OB0 = ALT0 + a * 2 * RR; total = OB0 + a * (2 * N + G);
    }
    __host__ __device__ int t(int a, int f, bool with_goal, int cell, int gl) const
    {
        if (f < 2 && with_goal) return ALT0 + (a * 2 + f) * RR + cell * G + gl;
        return a * 3 * RR + f * RR + (f == 2 ? cell * G + gl : cell);
    }
    __host__ __device__ int o(int a, int f, int v) const { return OB0 + a * (2 * N + G) + (f == 2 ? 2 * N + v : f * N + v); }
};
// where a step's Dirichlet rows come from: the padded tables in HBM (L2-resident), the same with the observation tables in LDS, or the
// deduplicated rows in LDS
struct HistRowsGlobal {
    const float* base; const float* alt; const float* otab; HistLayout L;
    __device__ __forceinline__ const float* t(int a, int f, bool with_goal, int cell, int gl) const
    {
        return f < 2 && with_goal ? alt + L.alt_row(a, f, cell, gl) : base + L.t_row(a, f, with_goal, cell, gl);
    }
    __device__ __forceinline__ const float* o(int a, int f, int v) const { return otab + (L.o_row(a, f, v) - L.obase0); }
};
template <int K>
struct HistRowsLds {
    const uint8_t* rid; const float* rows; HistRowIds I;
    __device__ __forceinline__ const float* t(int a, int f, bool with_goal, int cell, int gl) const { return rows + (int)rid[I.t(a, f, with_goal, cell, gl)] * K; }
    __device__ __forceinline__ const float* o(int a, int f, int v) const { return rows + (int)rid[I.o(a, f, v)] * K; }
};

// increments per cell of one row (at most 16 cells), 8 bits each
struct RowCount {
    uint64_t lo, hi;
    __device__ __forceinline__ void add(bool hit, int cell)
    {
        const uint64_t one = hit ? 1ull : 0ull;
        lo += cell < 8 ? one << (8 * cell) : 0ull;
        hi += cell < 8 ? 0ull : one << (8 * (cell - 8));
    }
    __device__ __forceinline__ float at(int i) const { return (float)(uint32_t)(((i < 8 ? lo >> (8 * i) : hi >> (8 * (i - 8)))) & 0xffull); }
};

// the same with 6 bits per cell in one word: at most 10 cells, at most 63 increments per cell (records of at most 63 entries)
struct RowCount6 {
    uint64_t v;
    __device__ __forceinline__ void add(bool hit, int cell) { v += (uint64_t)(hit ? 1u : 0u) << (6 * cell); }
    __device__ __forceinline__ float at(int i) const { return (float)(uint32_t)((v >> ((6 * i) & 63)) & 63ull); }
};
// quad-wide sum by DPP (all four lanes of a quad are active together): every lane ends with the total.  No field of a RowCount6 can carry
// into its neighbour (the four lanes' counts of one cell add up to at most 63), so the 64-bit sum is a plain one.
template <int CTRL>
__device__ __forceinline__ uint64_t quad_perm_u64(uint64_t x)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)x, CTRL, 0xf, 0xf, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(x >> 32), CTRL, 0xf, 0xf, true);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t quad_sum_u64(uint64_t x)
{
    x += quad_perm_u64<0xB1>(x);   // lanes 1 0 3 2
    x += quad_perm_u64<0x4E>(x);   // lanes 2 3 0 1
    return x;
}

struct GlobalEntries {
    const uint32_t* p;
    __device__ __forceinline__ uint32_t at(int t) const { return p[t]; }
};

// a Dirichlet row in registers: prior row (K / 4 aligned 16-byte loads; the words past n belong to the next row
// or the table's padding and are not used) + this particle's increments
// K = floats fetched (16-byte pieces); the loops over a row run to KL: a 12-float row holds at most 10 values (G <= 10 goals, N <= 8), and the
// two padding places cost a sixth of every pass
template <int K>
struct HistRow {
#ifdef FBA_ROW_FULL   /* A/B builds: walk the padding too */
    static constexpr int KL = K;
#else
    static constexpr int KL = K == 12 ? 10 : K;
#endif
    float r[K];
    __device__ __forceinline__ void fetch(const float* __restrict__ prior)
    {
        const float4* p4 = reinterpret_cast<const float4*>(prior);
#pragma unroll
        for (int i = 0; i < K / 4; ++i) {
            const float4 v = p4[i];
            r[4 * i] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
        }
    }
    template <class COUNTS>
    __device__ __forceinline__ void add(int n, const COUNTS& c)
    {
#pragma unroll
        for (int i = 0; i < KL; ++i) r[i] = i < n ? r[i] + c.at(i) : 0.f;
    }
    // the same for a row whose places past n already hold 0.f (the deduplicated rows of Problem::hist_lds are padded with zeros by the host) and counts
    // that are zero there (no entry's feature value reaches past n): nothing to select
    template <class COUNTS>
    __device__ __forceinline__ void add_zero_padded(const COUNTS& c)
    {
#pragma unroll
        for (int i = 0; i < KL; ++i) r[i] = r[i] + c.at(i);
    }
    // sampleFromExpectedMult (random.cpp:244-255) for the uniform draw u: double total, float CDF against a double
    // threshold.  Counts are never negative, so the CDF does not decrease and "the first i with p < cdf(i), else
    // n - 1" is the number of i <= n - 2 with p >= cdf(i): no branches.
    __device__ __forceinline__ int sample(double u, int n) const
    {
        double total = (double)r[0];
#pragma unroll
        for (int i = 1; i < KL; ++i) total += (double)r[i];   // (add() left 0.f in the places past n: adding them changes nothing)
        const double p = u * total;
        // "p < (double)sum" for a float sum is "sum > pf" with pf the largest float <= p (p >= 0): the comparisons stay in fp32
        float pf = (float)p;
        if ((double)pf > p) pf = __uint_as_float(__float_as_uint(pf) - 1u);
        float sum = r[0];
        int pick  = 0;
#pragma unroll
        for (int i = 1; i < KL; ++i) {
            pick += (i < n && !(sum > pf)) ? 1 : 0;
            sum += r[i];
        }
        return pick;
    }
    // expectedMult(row)[o] as BABNModel::computeObservationProbability uses it: float sum, float division
    __device__ __forceinline__ float prob(int n, int o) const
    {
        float sum = r[0], mine = r[0];
#pragma unroll
        for (int i = 1; i < KL; ++i)
            if (i < n) { sum += r[i]; mine = (i == o) ? r[i] : mine; }
        return ((double)sum <= 1e-300) ? 0.0f : mine / sum;
    }
};
__device__ __forceinline__ double u01_of(uint64_t w) { return (double)(w >> 11) * (1.0 / 9007199254740992.0); }  // rnd::uniform_rand01 of one draw
__device__ __forceinline__ bool gridworld_on_goal(const Problem& P, int cell, int gl)
{
    // (every word is read, none chosen by address: a choice between two members' addresses keeps the whole by-value Problem in
    //  scratch memory, and each of its fields then costs a trip to memory where it is used)
    const int q      = gl >> 2;
    const uint32_t w = (q == 0 ? P.gw_goalcell0 : 0u) | (q == 1 ? P.gw_goalcell1 : 0u) | (q == 2 ? P.gw_goalcell2 : 0u) | (q == 3 ? P.gw_goalcell3 : 0u);
    return ((w >> (8 * (gl & 3))) & 0xffu) == (uint32_t)cell;
}
__device__ __forceinline__ uint32_t gridworld_pack_state(const Problem& P, int s)
{
    const int N = P.gw_N, G = P.gw_G;
    return hist_pack(s / (N * G), (s / G) % N, s % G);
}
__device__ __forceinline__ int gridworld_unpack_state(const Problem& P, uint32_t sp) { return (hist_x(sp) * P.gw_N + hist_y(sp)) * P.gw_G + hist_g(sp); }

// BAPOMDP::step (BAPOMDP.cpp:111-143) over BABNModel (BABNModel.cpp:292-325) for the gridworld FBA-POMDP on a
// history particle, one lane per particle (the belief update): the draws, their order and every row value are those
// of gridworld_fact_step on the dense table.  `ent` = the particle's n entries of action a; `sp` = the state as
// hist_pack(x, y, goal); returns the step's entry (what incrementCountsOf would add, :354-382, App. A #6:
// observation rows at the OLD state's values) and P(real_o | a, s') from the counts after the step's own
// increments (BABNModel.cpp:328-352).
template <int KN, int KG, class ROWS>
__device__ __forceinline__ bool gridworld_hist_step_k(const Problem& P, Rng& g, const ROWS& R, const uint32_t* __restrict__ ent, int n, uint32_t mask,
                                                      uint32_t& sp, int a, int& o, double& r, uint32_t& entry, int real_o, double& prob)
{
    const int N = P.gw_N, G = P.gw_G;
    const int x = hist_x(sp), y = hist_y(sp), gl = hist_g(sp);
    const int cell = x * N + y;
    const bool mx = (mask >> (2 * a)) & 1u, my = (mask >> (2 * a + 1)) & 1u;
    HistRow<KN> rx, ry;
    HistRow<KG> rg;
    rx.fetch(R.t(a, 0, mx, cell, gl));
    ry.fetch(R.t(a, 1, my, cell, gl));
    rg.fetch(R.t(a, 2, true, cell, gl));
    // pass 1: the increments this particle has made to the rows T(a, .)(x, y [, goal])
    RowCount cx{0, 0}, cy{0, 0}, cg{0, 0};
    for (int j = 0; j < n; ++j) {
        const uint32_t e = ent[j];
        const bool hit_xy = ((e ^ sp) & 0x3fu) == 0, hit_g = ((e ^ sp) & 0x3ffu) == 0;
        cx.add(mx ? hit_g : hit_xy, (int)((e >> 10) & 7u));
        cy.add(my ? hit_g : hit_xy, (int)((e >> 13) & 7u));
        cg.add(hit_g, (int)((e >> 16) & 15u));
    }
    rx.add(N, cx);
    const int nx = rx.sample(g.u01(), N);
    ry.add(N, cy);
    const int ny = ry.sample(g.u01(), N);
    rg.add(G, cg);
    const int ng = rg.sample(g.u01(), G);
    rx.fetch(R.o(a, 0, nx));
    ry.fetch(R.o(a, 1, ny));
    rg.fetch(R.o(a, 2, ng));
    // pass 2: the increments to the rows O(a, .)(value of the new state's feature); a step increments them at
    // the row of the state it STARTED from
    RowCount ox{0, 0}, oy{0, 0}, og{0, 0};
    for (int j = 0; j < n; ++j) {
        const uint32_t e = ent[j];
        ox.add((int)(e & 7u) == nx, (int)((e >> 20) & 7u));
        oy.add((int)((e >> 3) & 7u) == ny, (int)((e >> 23) & 7u));
        og.add((int)((e >> 6) & 15u) == ng, (int)((e >> 26) & 15u));
    }
    HistRow<KN> px = rx, py = ry;   // the prior rows, for the probability below
    HistRow<KG> pg = rg;
    rx.add(N, ox);
    const int vx = rx.sample(g.u01(), N);
    ry.add(N, oy);
    const int vy = ry.sample(g.u01(), N);
    rg.add(G, og);
    const int vg = rg.sample(g.u01(), G);
    o = (vx * N + vy) * G + vg;
    {
        // the step's own observation increments land at rows x, y, goal of the old state: they are part of the
        // rows read here only where the feature kept its value
        ox.add(x == nx, vx);
        oy.add(y == ny, vy);
        og.add(gl == ng, vg);
        px.add(N, ox);
        py.add(N, oy);
        pg.add(G, og);
        const int qg = real_o % G, qy = (real_o / G) % N, qx = real_o / (N * G);
        double pr = 1;
        pr *= px.prob(N, qx);
        pr *= py.prob(N, qy);
        pr *= pg.prob(G, qg);
        prob = pr;
    }
    const bool found = gridworld_on_goal(P, cell, gl);  // GridWorldBAExtension.cpp:74-99: terminal and reward from the OLD state
    r     = found ? 1 : 0;
    entry = hist_entry(sp, hist_pack(nx, ny, ng), hist_pack(vx, vy, vg));
    sp    = hist_pack(nx, ny, ng);
    return found;
}
// (R: where the rows come from -- HistRowsGlobal, or HistRowsLds<K> with K >= 8 for G <= 8 and K >= 12 beyond)
template <class ROWS>
__device__ __forceinline__ bool gridworld_hist_step(const Problem& P, Rng& g, const ROWS& R, const uint32_t* ent, int n, uint32_t mask, uint32_t& sp, int a,
                                                    int& o, double& r, uint32_t& entry, int real_o, double& prob)
{
    if (P.gw_G <= 8) return gridworld_hist_step_k<8, 8>(P, g, R, ent, n, mask, sp, a, o, r, entry, real_o, prob);
    return gridworld_hist_step_k<8, 12>(P, g, R, ent, n, mask, sp, a, o, r, entry, real_o, prob);   // (G <= 10)
}
__device__ __forceinline__ bool gridworld_hist_step(const Problem& P, Rng& g, const uint32_t* ent, int n, uint32_t mask, uint32_t& sp, int a, int& o,
                                                    double& r, uint32_t& entry, int real_o, double& prob)
{
    const HistLayout L(P.gw_N, P.gw_G, P.A);
    return gridworld_hist_step(P, g, HistRowsGlobal{P.hist_base, P.hist_alt, P.hist_base + L.obase0, L}, ent, n, mask, sp, a, o, r, entry, real_o, prob);
}

// ---- the same step shared by the four lanes of a quad (search_hist_kernel) ---------------------------------------
// Philox for a quad: the draw counter is the same in the four lanes; lane q generates block base + q of the
// stream, so the quad holds eight consecutive draws at the price of one block each, and a draw is fetched
// from the lane that made it (ds_bpermute within the quad).  Same streams, same draw numbers as Rng.
struct QuadRng {
    uint32_t k0, k1, c1, c2, c3;
    uint32_t draw, base;   // the same in the four lanes
    uint32_t w[4];         // this lane's block: draw 2 (base + q) = w[1]:w[0], the next one w[3]:w[2]
    int q, addr0;          // lane within the quad; ds_bpermute address of the quad's first lane
    __device__ __forceinline__ void init(uint32_t lo, uint32_t hi, uint32_t run, uint32_t episode, uint32_t t, int lane)
    {
        k0 = lo; k1 = hi;
        c3 = run;
        c2 = ((t & 0xffu) << 8) | ((episode & 0xffffu) << 16);
        c1 = 0; draw = 0; base = 0xfffffff0u;
        q = lane & 3; addr0 = (lane & ~3) * 4;
        w[0] = w[1] = w[2] = w[3] = 0;
    }
    __device__ __forceinline__ void stream(uint32_t phase, uint32_t unit)
    {
        c2   = (c2 & 0xffffff00u) | (phase & 0xffu);
        c1   = unit;
        draw = 0;
        base = 0xfffffff0u;
    }
    __device__ __forceinline__ void refill()
    {
        base = draw >> 1;
        uint32_t x0 = base + (uint32_t)q, x1 = c1, x2 = c2, x3 = c3;
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)x0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)x2;
            const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
            const uint32_t n0 = hi1 ^ x1 ^ a, n2 = hi0 ^ x3 ^ b;
            x0 = n0; x1 = lo1; x2 = n2; x3 = lo0;
            a += 0x9E3779B9u;
            b += 0xBB67AE85u;
        }
        w[0] = x0; w[1] = x1; w[2] = x2; w[3] = x3;
    }
    // the next n draws must all lie in the quad's eight
    __device__ __forceinline__ void ensure(int n)
    {
        const uint32_t lo = (draw >> 1) - base, hi = ((draw + (uint32_t)n - 1u) >> 1) - base;
        if (lo >= 4u || hi >= 4u) refill();
    }
    __device__ __forceinline__ uint64_t at(uint32_t d) const
    {
        const int addr = addr0 + 4 * (int)((d >> 1) - base);
        const uint32_t r0 = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)w[0]), r1 = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)w[1]);
        const uint32_t r2 = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)w[2]), r3 = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)w[3]);
        return (d & 1u) ? (((uint64_t)r3 << 32) | r2) : (((uint64_t)r1 << 32) | r0);
    }
    __device__ __forceinline__ uint64_t next64() { return at(draw++); }   // a draw all four lanes consume together
    __device__ __forceinline__ double u01() { return u01_of(next64()); }
    __device__ __forceinline__ int slow_int(int lo, int hi) { return lo + (int)floor(u01() * (double)(hi - lo)); }
    // slow_int(0, 4): floor(u * 4) with u = (w >> 11) * 2^-53 is the top two bits of the draw -- the scaling is exact, so this is the same integer
    __device__ __forceinline__ int slow_int4() { return (int)(next64() >> 62); }
};
__device__ __forceinline__ int quad_bcast(int addr0, int from, int v) { return __builtin_amdgcn_ds_bpermute(addr0 + 4 * from, v); }

// Lane f < 3 of the quad owns state / observation feature f (lane 3 repeats lane 2's work): one history pass,
// one row, one sampling chain per lane and phase instead of three.  Draws 0..2 of the phase go to features 0..2
// as in the one-lane form.  `list` = the root particle's n_list entries of action a, staged as [j * STRIDE].
// `rows` = where the Dirichlet rows come from (HistRowsGlobal / HistRowsLds).
// SPLIT: the passes over the record's entries are what a step costs once histories have grown (twenty vector instructions per entry
// and pass when every lane walks all of them for its own feature).  With SPLIT lane q walks entries q, q + 4, ... and counts for all three
// features -- 6-bit counters, one 64-bit word per feature -- and a DPP sum over the quad hands every lane the totals: a quarter of the
// entries per lane.  Same counts, so the same rows; needs records of at most 63 entries (else the lanes walk as before).
template <int K, int STRIDE, bool SPLIT = false, class ROWS>
__device__ __forceinline__ bool gridworld_hist_step_quad(const Problem& P, QuadRng& g, const uint32_t* list, int n_list, uint32_t mask, uint32_t& sp,
                                                         int a, int& o, double& r, const ROWS& rows)
{
    const HistLayout L(P.gw_N, P.gw_G, 4);
    const int N = L.N, G = L.G;
    const int f = min(g.q, 2);
    const int x = hist_x(sp), y = hist_y(sp), gl = hist_g(sp);
    const int cell = x * N + y, n = f == 2 ? G : N;
    const bool with_goal = f == 2 || ((mask >> (2 * a + f)) & 1u);
    const int shift = 3 * f;
    const uint32_t fmask = f == 2 ? 15u : 7u;
    const bool split = SPLIT && P.hist_cap <= 63;
    HistRow<K> row;
    row.fetch(rows.t(a, f, with_goal, cell, gl));
    if (split) {
        const bool mx = (mask >> (2 * a)) & 1u, my = (mask >> (2 * a + 1)) & 1u;
        RowCount6 cx{0}, cy{0}, cg{0};
        for (int j = g.q; j < n_list; j += HIST_QUAD) {
            const uint32_t e = list[j * STRIDE];
            const bool hit_xy = ((e ^ sp) & 0x3fu) == 0, hit_g = ((e ^ sp) & 0x3ffu) == 0;
            cx.add(mx ? hit_g : hit_xy, (int)((e >> 10) & 7u));
            cy.add(my ? hit_g : hit_xy, (int)((e >> 13) & 7u));
            cg.add(hit_g, (int)((e >> 16) & 15u));
        }
        const uint64_t sx = quad_sum_u64(cx.v), sy = quad_sum_u64(cy.v), sg = quad_sum_u64(cg.v);   // (every lane executes every DPP sum)
        row.add(n, RowCount6{f == 0 ? sx : (f == 1 ? sy : sg)});
    } else {
        RowCount cnt{0, 0};
        const uint32_t keep = with_goal ? 0x3ffu : 0x3fu;
        for (int j = 0; j < n_list; ++j) {
            const uint32_t e = list[j * STRIDE];
            cnt.add(((e ^ sp) & keep) == 0, (int)((e >> (10 + shift)) & fmask));
        }
        row.add(n, cnt);
    }
    const int nv = row.sample(u01_of(g.at(g.draw + (uint32_t)f)), n);
    const int nx = quad_bcast(g.addr0, 0, nv), ny = quad_bcast(g.addr0, 1, nv), ng = quad_bcast(g.addr0, 2, nv);
    row.fetch(rows.o(a, f, nv));
    if (split) {
        RowCount6 ox{0}, oy{0}, og{0};
        for (int j = g.q; j < n_list; j += HIST_QUAD) {
            const uint32_t e = list[j * STRIDE];
            ox.add((int)(e & 7u) == nx, (int)((e >> 20) & 7u));
            oy.add((int)((e >> 3) & 7u) == ny, (int)((e >> 23) & 7u));
            og.add((int)((e >> 6) & 15u) == ng, (int)((e >> 26) & 15u));
        }
        const uint64_t sx = quad_sum_u64(ox.v), sy = quad_sum_u64(oy.v), sg = quad_sum_u64(og.v);
        row.add(n, RowCount6{f == 0 ? sx : (f == 1 ? sy : sg)});
    } else {
        RowCount cnt{0, 0};
        for (int j = 0; j < n_list; ++j) {
            const uint32_t e = list[j * STRIDE];
            cnt.add(((e >> shift) & fmask) == (uint32_t)nv, (int)((e >> (20 + shift)) & fmask));
        }
        row.add(n, cnt);
    }
    const int ov = row.sample(u01_of(g.at(g.draw + 3u + (uint32_t)f)), n);
    g.draw += 6;
    const int vx = quad_bcast(g.addr0, 0, ov), vy = quad_bcast(g.addr0, 1, ov), vg = quad_bcast(g.addr0, 2, ov);
    o = (vx * N + vy) * G + vg;
    const bool found = gridworld_on_goal(P, cell, gl);  // GridWorldBAExtension.cpp:74-99: terminal and reward from the OLD state
    r  = found ? 1 : 0;
    sp = hist_pack(nx, ny, ng);
    return found;
}

// One pass of a simulated step over ONE Dirichlet row per lane (lane f < 3 of the quad owns feature f), in the form search_hist2_kernel runs twice per
// loop iteration: fetch the row, add the increments this particle's entries made to it, draw.  An entry counts where ((entry ^ pattern) & k) == 0,
// at the value its field at `cshift` (+ 3 per feature) holds:
//   transition pass  -- entries whose OLD state matches the current one (pattern = the state; kx / ky = 0x3f or 0x3ff: whether the x / y node has
//                       the goal as a parent; kg = 0x3ff), counted at their NEW state's values (cshift 10);
//   observation pass -- entries whose old state's feature equals the new state's (pattern = the new state; one field per mask: 7, 7 << 3, 15 << 6),
//                       counted at their OBSERVATION's values (cshift 20; BABNModel::incrementCountsOf files them under the old state, App. A #6).
// Records of at most 63 entries: lane q walks entries q, q + 4, ... for all three features (6-bit counters) and a DPP sum hands every lane the totals;
// longer records: every lane walks every entry for its own feature.  Same counts either way, so the same row and the same draw as gridworld_hist_step_quad.
// (`obs` selects the pass; mx / my = the x / y transition node of the step's action has the goal as a parent)
// ZPAD: the row's places past n hold 0.f (rows from Problem::hist_lds).
template <int K, int STRIDE, bool ZPAD = false>
__device__ __forceinline__ int hist_row_pass(const Problem& P, const QuadRng& g, const uint32_t* list, int n_list, uint32_t pattern, bool obs, bool mx, bool my,
                                             const float* rowp, int n, int f, double u)
{
    const uint32_t kx = obs ? 7u : (mx ? 0x3ffu : 0x3fu), ky = obs ? (7u << 3) : (my ? 0x3ffu : 0x3fu), kg = obs ? (15u << 6) : 0x3ffu;
    const int cshift = obs ? 20 : 10;
    HistRow<K> row;
    row.fetch(rowp);
    if (P.hist_cap <= 63) {
        RowCount6 cx{0}, cy{0}, cg{0};
        for (int j = g.q; j < n_list; j += HIST_QUAD) {
            const uint32_t e = list[j * STRIDE], d = e ^ pattern, c = e >> cshift;
            cx.add((d & kx) == 0, (int)(c & 7u));
            cy.add((d & ky) == 0, (int)((c >> 3) & 7u));
            cg.add((d & kg) == 0, (int)((c >> 6) & 15u));
        }
        const uint64_t sx = quad_sum_u64(cx.v), sy = quad_sum_u64(cy.v), sg = quad_sum_u64(cg.v);   // (every lane executes every DPP sum)
        if (ZPAD) row.add_zero_padded(RowCount6{f == 0 ? sx : (f == 1 ? sy : sg)});
        else row.add(n, RowCount6{f == 0 ? sx : (f == 1 ? sy : sg)});
    } else {
        const uint32_t k = f == 0 ? kx : (f == 1 ? ky : kg), fmask = f == 2 ? 15u : 7u;
        const int shift = cshift + 3 * f;
        RowCount cnt{0, 0};
        for (int j = 0; j < n_list; ++j) {
            const uint32_t e = list[j * STRIDE];
            cnt.add(((e ^ pattern) & k) == 0, (int)((e >> shift) & fmask));
        }
        row.add(n, cnt);
    }
    return row.sample(u, n);
}

// BABNModel::computeObservationProbability (BABNModel.cpp:328-352)
template <bool REG, class View>
__device__ __forceinline__ double fact_obs_prob(const Problem& P, Rng& g, const View& cnt, int new_s, int a, int o)
{
    const FDesc* fd   = P.fd;
    const uint64_t fv = pack_features(new_s, fd->Sstep, fd->FS);
    const uint64_t of = pack_features(o, fd->Ostep, fd->FO);
    double prob = 1;
#pragma unroll
    for (int f = 0; f < MAXF; ++f)
        if (f < fd->FO) {
            const NodeRegs nd = load_node(&fd->nodes[P.A * fd->FS + a * fd->FO + f]);
            const int row   = node_row(fd, nd, node_mask(fd, nd, cnt), fv);
            if (REG) {  // sampleMultinominal = sampleMult: a fresh Dirichlet draw per observation feature
                prob *= (float)sample_mult_at(P.zig, g, cnt, row, nd.out, feat(of, f));
            } else {
                float sum = cnt.at(row);
                for (int i = 1; i < nd.out; ++i) sum += cnt.at(row + i);
                prob *= ((double)sum <= 1e-300) ? 0.0f : cnt.at(row + feat(of, f)) / sum;
            }
        }
    return prob;
}

// BABNModel::LogBDScore (BABNModel.cpp:451-478) over DBNNode::LogBDScore (DBNNode.cpp:82-117): per action the transition
// nodes, then the observation nodes; per node every Dirichlet row in CPT order, one running double sum.  `cnt` and
// `prior` are particle blobs of the same structure.
template <class View>
__device__ __forceinline__ double log_bd_score(const Problem& P, const View& cnt, const View& prior)
{
    const FDesc* fd = P.fd;
    double bd = 0;
    for (int a = 0; a < P.A; ++a)
        for (int k = 0; k < fd->FS + fd->FO; ++k) {
            const FNode& nd = k < fd->FS ? fd->nodes[a * fd->FS + k] : fd->nodes[P.A * fd->FS + a * fd->FO + (k - fd->FS)];
            const uint32_t mask = node_mask(fd, nd, cnt);
            int rows = 1;
            for (int j = 0; j < nd.nmax; ++j)
                if ((mask >> j) & 1u) rows *= nd.psz[j];
            for (int r = 0; r < rows; ++r) {
                double tot = 0, ptot = 0;
                for (int v = 0; v < nd.out; ++v) {
                    const float x = cnt.at(nd.off + r * nd.out + v), y = prior.at(nd.off + r * nd.out + v);
                    tot += (double)x;
                    ptot += (double)y;
                    bd += log_gamma((double)x) - log_gamma((double)y);
                }
                bd += log_gamma(ptot) - log_gamma(tot);
            }
        }
    return bd;
}

// ---- simulator.step for the three simulators --------------------------------------------------
__device__ __forceinline__ int model_ninc(const Problem& P)
{
    return P.model == FBA_MODEL_POMDP ? 0 : (P.model == FBA_MODEL_BA_TABLE ? 2 : P.fd->FS + P.fd->FO);
}

template <bool REG, class View, class Sink>
__device__ __forceinline__ bool sim_step(const Problem& P, Rng& g, const View& cnt, int& s, int a, int& o, double& r, const Sink& inc)
{
    if (P.model == FBA_MODEL_POMDP) return domain_step(P, g, s, a, o, r);
    if (P.model == FBA_MODEL_BA_FACTORED) {
        if (!REG && dom_is_grid(P.domain)) return gridworld_fact_step(P, g, cnt, s, a, o, r, inc);
        if (!REG && dom_is_ca(P.domain) && P.ca_plain) return ca_fact_step(P, g, cnt, s, a, o, r, inc);
        if (!REG && dom_is_sys(P.domain) && P.ca_plain) return sysadmin_fact_step(P, g, cnt, s, a, o, r, inc);
        return fact_step<REG>(P, g, cnt, s, a, o, r, inc);
    }
    const int S = P.S, A = P.A, O = P.O;
    const int t_off = s * A * S + a * S;
    const int ns    = sample_row<REG>(P, g, cnt, t_off, S);
    const int o_off = P.phi_len + a * S * O + ns * O;
    o               = sample_row<REG>(P, g, cnt, o_off, O);
    const bool t    = ext_terminal(P, s, a, ns);
    r               = ext_reward(P, s, a, ns);
    inc.add(0, t_off + ns);
    inc.add(1, o_off + o);
    s               = ns;
    return t;
}

// BAPOMDP::step (sim_step's tabular branch) for packed tiger records (Problem::packed): both Dirichlet rows of a tiger
// step have two cells and start at an even cell -- T(s, a, .) at 6s + 2a, O(a, s', .) at 12 + 4a + 2s' -- so a row is ONE
// word of the record (its two uint16 increment counts) plus the two prior values beside it in the prior table (one
// 8-byte read).  `word(w)` = word w of the record.  Same draws, same sums, same results as sim_step through PackedView.
template <class RNG, class WordFn, class Sink>
__device__ __forceinline__ bool tiger_step_packed(const Problem& P, RNG& g, const WordFn& word, const float* prior, int& s, int a, int& o, double& r,
                                                  const Sink& inc)
{
    const int t_off = s * 6 + a * 2;
    int ns;
    {
        const uint32_t w = word(t_off >> 1);
        const float2 p2  = *reinterpret_cast<const float2*>(prior + t_off);
        const float c0 = p2.x + (float)(w & 0xffffu), c1 = p2.y + (float)(w >> 16);
        const double p = g.u01() * ((double)c0 + (double)c1);   // sampleFromExpectedMult on a row of two
        ns = (p < (double)c0) ? 0 : 1;
    }
    const int o_off = 12 + a * 4 + ns * 2;
    {
        const uint32_t w = word(o_off >> 1);
        const float2 p2  = *reinterpret_cast<const float2*>(prior + o_off);
        const float c0 = p2.x + (float)(w & 0xffffu), c1 = p2.y + (float)(w >> 16);
        const double p = g.u01() * ((double)c0 + (double)c1);
        o = (p < (double)c0) ? 0 : 1;
    }
    const bool t = ext_terminal(P, s, a, ns);
    r            = ext_reward(P, s, a, ns);
    inc.add(0, t_off + ns);
    inc.add(1, o_off + o);
    s = ns;
    return t;
}

// POMDP::computeObservationProbability of the simulator in use
// (BAPOMDP.cpp:93-99 -> BAFlatModel::computeObservationProbability BAFlatModel.cpp:106-124)
template <bool REG, class View>
__device__ __forceinline__ double sim_obs_prob(const Problem& P, Rng& g, const View& cnt, int new_s, int a, int o)
{
    if (P.model == FBA_MODEL_POMDP) return domain_obs_prob(P, o, a, new_s);
    if (P.model == FBA_MODEL_BA_FACTORED) return fact_obs_prob<REG>(P, g, cnt, new_s, a, o);
    if (P.O == 1) return 1.0;
    if (REG) return sample_mult_at(P.zig, g, cnt, P.phi_len + a * P.S * P.O + new_s * P.O, P.O, o);
    return expected_mult_at(cnt, P.phi_len + a * P.S * P.O + new_s * P.O, P.O, o);
}

// FactoredTigerFactoredPrior::setObservationModel (FactoredTigerPriors.cpp:221-263): the listen
// observation node of one particle for parent set `mask` (bit j = state feature j)
__device__ __forceinline__ void ftiger_set_observation_model(const Problem& P, float* rec, uint32_t mask)
{
    const FDesc* fd = P.fd;
    const FNode& nd = fd->nodes[P.A * fd->FS + 2 * fd->FO];
    const float acc = (.85f - P.noise) * P.counts_total, inacc = (.15f + P.noise) * P.counts_total;
    const float unif = .5f * P.counts_total;
    const int np = __popc(mask), rows = 1 << np, max_rows = 1 << nd.nmax;
    float* row = rec + nd.off;
    for (int r = 0; r < max_rows; ++r) {
        float l = 0.f, rr = 0.f;
        if (r < rows) {
            if (np > 0 && (mask & 1u)) {  // parents[0] == 0: informed by the tiger location
                const int loc = r >> (np - 1);
                l  = (loc == 0) ? acc : inacc;
                rr = (loc == 1) ? acc : inacc;
            } else {
                l = rr = unif;
            }
        }
        row[2 * r + 0] = l;
        row[2 * r + 1] = rr;
    }
    rec[fd->ncounts + 0] = __uint_as_float(mask);
}

// FBAPOMDPPrior::sample -> sampleFBAPOMDPState / sampleFullyConnectedState
// (FBAPOMDPPrior.cpp:27-37, FactoredTigerPriors.cpp:197-219, 265-291).  `rec` already holds the
// base prior record; the structure draws follow the domain start-state draw in the same stream.
// GridWorldFactBAPrior::setNoisyTransitionNode (GridWorldBAPriors.cpp:255-295): the x (feature 0)
// or y (feature 1) transition node of action a with the goal as third parent
__device__ __forceinline__ void gw_fill_xy_node_with_goal(const Problem& P, float* rec, int a, int feature)
{
    const GridDesc* gw = P.gw;
    const FNode& nd    = P.fd->nodes[a * 3 + feature];
    const int N = gw->N, G = gw->G;
    float* base = rec + nd.off;
    for (int k = 0; k < N * N * G * N; ++k) base[k] = 0.f;
    for (int x = 0; x < N; ++x)
        for (int y = 0; y < N; ++y) {
            int nx = x, ny = y;
            const float trans_prob = gw_slow_at(gw, x, y) ? (float)(.15 + (double)P.noise) : (float).95;
            gw_move(gw, a, nx, ny);
            const int loc = feature == 0 ? x : y, new_loc = feature == 0 ? nx : ny;
            for (int gl = 0; gl < G; ++gl) {
                float* row = base + ((x * N + y) * G + gl) * N;
                row[loc] += (1 - trans_prob) * P.counts_total;
                row[new_loc] += (trans_prob)*P.counts_total;
            }
        }
    rec[P.fd->ncounts + nd.var] = __uint_as_float(7u);
}

// CollisionAvoidanceFactoredPrior::obstacleTransition (CollisionAvoidancePriors.cpp:402-427): the H
// counts of an obstacle at row y
__device__ __forceinline__ void ca_obstacle_transition(const Problem& P, int y, float* out)
{
    const int H = P.ca->H;
    const float move_prob = (float)(.25 - .5 * (double)P.noise);
    const float stay_prob = (y == 0 || y == H - 1) ? (float)(3 * .25 + .5 * (double)P.noise) : (float)(2 * .25 + (double)P.noise);
    for (int k = 0; k < H; ++k) out[k] = 0.f;
    if (y != 0) out[y - 1] = move_prob * P.counts_total;
    if (y != H - 1) out[y + 1] = move_prob * P.counts_total;
    out[y] = stay_prob * P.counts_total;
}
// CollisionAvoidanceFactoredPrior::sampleBlockTModel (:528-590) for one (action, obstacle feature):
// the node's CPT under parent set `mask`.  With the obstacle among its own parents every row is
// obstacleTransition(its own value), otherwise every row is uniform, counts_total / H.
__device__ __forceinline__ void ca_fill_obstacle_node(const Problem& P, float* rec, int a, int f, uint32_t mask)
{
    const FDesc* fd = P.fd;
    const FNode& nd = fd->nodes[a * fd->FS + f];
    const int H = P.ca->H, FS = fd->FS;
    int rows = 1, rows_max = 1;
    for (int k = 0; k < FS; ++k) {
        rows_max *= fd->Ssz[k];
        if ((mask >> k) & 1u) rows *= fd->Ssz[k];
    }
    float* base = rec + nd.off;
    for (int r = 0; r < rows; ++r) {
        float* row = base + r * H;
        if ((mask >> f) & 1u) {
            int rem = r, own = 0;
            for (int k = FS - 1; k >= 0; --k)  // last parent is the fastest digit
                if ((mask >> k) & 1u) {
                    if (k == f) own = rem % fd->Ssz[k];
                    rem /= fd->Ssz[k];
                }
            ca_obstacle_transition(P, own, row);
        } else {
            const float u = P.counts_total / (float)H;
            for (int y = 0; y < H; ++y) row[y] = u;
        }
    }
    for (int k = rows * H; k < rows_max * H; ++k) base[k] = 0.f;
    rec[fd->ncounts + nd.var] = __uint_as_float(mask);
}

// SysAdminFactoredPrior::fullyConnectedT (SysAdminFactoredPrior.cpp:249-277): every transition node with
// all N computers as parents, counts {p, 1 - p} (a total of ONE per row), p = SysAdmin::failProbability
// of the state the parent values spell (SysAdmin.cpp:84-100, a float)
__device__ __forceinline__ void sys_fill_fully_connected(const Problem& P, float* rec)
{
    const FDesc* fd = P.fd;
    const int N = P.sys->N;
    for (int a = 0; a < P.A; ++a)
        for (int f = 0; f < N; ++f) {
            const FNode& nd = fd->nodes[a * N + f];
            for (int r = 0; r < (1 << N); ++r) {
                int st = 0;  // SysAdmin::getState(parent values): bit k = value of feature k
                for (int k = 0; k < N; ++k) st |= ((r >> (N - 1 - k)) & 1) << k;
                double fail = ((st >> f) & 1) ? 1 - P.sys->keep[sys_failing_neighbours(P, f, st)] : 1;
                if (a == N + f) fail *= (1 - .95f);
                const float p = (float)fail;
                rec[nd.off + 2 * r + 0] = p;
                rec[nd.off + 2 * r + 1] = 1 - p;
            }
            rec[fd->ncounts + nd.var] = __uint_as_float((1u << N) - 1u);
        }
}

// SysAdminFactoredPrior::computeFailureProbability (SysAdminFactoredPrior.cpp:279-353) for computer `comp` under action a,
// parents = the features in `mask` (ascending), their values = the bits of row r (last parent fastest)
__device__ __forceinline__ float sys_failure_probability(const Problem& P, int a, int comp, uint32_t mask, int r, int np)
{
    const int N = P.sys->N;
    const bool rebooting = a == N + comp;
    int own = -1, nfn = 0, j = 0;
    for (int k = 0; k < N; ++k)
        if ((mask >> k) & 1u) {
            const int v = (r >> (np - 1 - j)) & 1;
            if (k == comp) own = v;
            if (P.domain == FBA_DOM_SYSADMIN_LINEAR && (k == comp - 1 || k == comp + 1) && v == 0) ++nfn;
            ++j;
        }
    if (own == 0) return rebooting ? 1 - .95f : 1;
    double fail = 1 - P.sys->keep[nfn];
    if (rebooting) fail *= (1 - .95f);
    if (own < 0) {  // the computer is not its own input
        fail += rebooting ? (1 - .95f) : 1;
        fail *= .5;
    }
    return (float)fail;
}
// one transition node of SysAdminFactoredPrior::computePriorModel (:98-127) for parent set `mask`: every parent-value row gets
// {total * p, total * 1 - p} -- the reference's own precedence: total minus p
__device__ __forceinline__ void sys_fill_node(const Problem& P, float* rec, int a, int f, uint32_t mask)
{
    const FDesc* fd = P.fd;
    const int N = P.sys->N, np = __popc(mask);
    const FNode& nd = fd->nodes[a * N + f];
    const float total = P.counts_total;
    float* base = rec + nd.off;
    for (int r = 0; r < (1 << np); ++r) {
        const float p = sys_failure_probability(P, a, f, mask, r, np);
        base[2 * r + 0] = total * p;
        base[2 * r + 1] = total * 1 - p;
    }
    for (int k = 2 << np; k < (2 << N); ++k) base[k] = 0.f;
    rec[fd->ncounts + nd.var] = __uint_as_float(mask);
}

__device__ __forceinline__ void factored_prior_sample(const Problem& P, Rng& g, float* rec)
{
    if (dom_is_sys(P.domain)) return;  // fixed structures only
    if (dom_is_ca(P.domain)) {
        // CollisionAvoidanceFactoredPrior::sampleFBAPOMDPState (:349-383): per obstacle, per action, one
        // boolean per state feature (always drawn); match-uniform forces the obstacle's own edge
        if (P.structure_prior != FBA_SP_UNIFORM && P.structure_prior != FBA_SP_MATCH_UNIFORM) return;
        const int FS = P.fd->FS;
        for (int f = 2; f < FS; ++f)
            for (int a = 0; a < P.A; ++a) {
                uint32_t mask = 0;
                for (int fp = 0; fp < FS; ++fp) {
                    const bool b = g.boolean();
                    if (b || (fp == f && P.structure_prior == FBA_SP_MATCH_UNIFORM)) mask |= 1u << fp;
                }
                ca_fill_obstacle_node(P, rec, a, f, mask);
            }
        return;
    }
    if (dom_is_grid(P.domain)) {  // GridWorldFactBAPrior::sampleFBAPOMDPState :415-441
        if (P.structure_prior != FBA_SP_MATCH_UNIFORM) return;
        for (int a = 0; a < P.A; ++a)
            for (int f = 0; f < 2; ++f)
                if (g.boolean()) gw_fill_xy_node_with_goal(P, rec, a, f);
        return;
    }
    const int FS = P.fd->FS;
    if (P.structure_prior == FBA_SP_FULLY_CONNECTED) {
        ftiger_set_observation_model(P, rec, (1u << FS) - 1u);
    } else if (P.structure_prior == FBA_SP_UNIFORM || P.structure_prior == FBA_SP_MATCH_UNIFORM) {
        uint32_t mask = 0;
        for (int f = 0; f < FS; ++f)
            if (g.boolean()) mask |= 1u << f;
        if (P.structure_prior == FBA_SP_MATCH_UNIFORM) mask |= 1u;
        ftiger_set_observation_model(P, rec, mask);
    }
}

// position-sensitive particle checksum (same function as oracle/orc.c particle_hash)
__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

}  // namespace fba
