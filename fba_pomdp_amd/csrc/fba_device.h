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
    __device__ __forceinline__ uint64_t next64()
    {
        uint64_t r;
        if ((draw & 1u) == 0) {
            uint32_t x0 = draw >> 1, x1 = c1, x2 = c2, x3 = c3;
            uint32_t a = k0, b = k1;
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                uint32_t hi0 = __umulhi(0xD2511F53u, x0), lo0 = 0xD2511F53u * x0;
                uint32_t hi1 = __umulhi(0xCD9E8D57u, x2), lo1 = 0xCD9E8D57u * x2;
                uint32_t n0 = hi1 ^ x1 ^ a, n2 = hi0 ^ x3 ^ b;
                x0 = n0; x1 = lo1; x2 = n2; x3 = lo0;
                a += 0x9E3779B9u;
                b += 0xBB67AE85u;
            }
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
struct GlobalView {
    const float* p;
    __device__ __forceinline__ float at(int k) const { return p[k]; }
};
template <int STRIDE>
struct LdsView {
    const float* p;
    __device__ __forceinline__ float at(int k) const { return p[k * STRIDE]; }
};

// ---------------------------------------------------------------------------------------------
// Sampling primitives (reference src/utils/random.hpp:93-115, random.cpp:244-279)
// ---------------------------------------------------------------------------------------------

// sampleFromMult<float const>: CDF accumulated in float, compared with a double threshold.
template <class View>
__device__ __forceinline__ int sample_from_mult_f(Rng& g, const View& row, int off, int n, double total)
{
    const double p = g.u01() * total;
    float sum      = row.at(off);
    for (int i = 1; i < n; ++i) {
        if (p < (double)sum) return i - 1;
        sum += row.at(off + i);
    }
    return n - 1;
}

// sampleFromExpectedMult: total accumulated in double
template <class View>
__device__ __forceinline__ int sample_expected_mult(Rng& g, const View& row, int off, int n)
{
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
struct Problem {
    int32_t domain, model, belief, planner;
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

__device__ __forceinline__ bool dom_is_tiger(int d) { return d == FBA_DOM_TIGER_EPISODIC || d == FBA_DOM_TIGER_CONTINUOUS; }
__device__ __forceinline__ bool dom_is_ftiger(int d) { return d == FBA_DOM_FTIGER_EPISODIC || d == FBA_DOM_FTIGER_CONTINUOUS; }
__device__ __forceinline__ bool dom_is_episodic(int d) { return d == FBA_DOM_TIGER_EPISODIC || d == FBA_DOM_FTIGER_EPISODIC; }

// Tiger::sampleStartState (Tiger.cpp:16-19), FactoredTiger::sampleStartState
__device__ __forceinline__ int domain_start(const Problem& P, Rng& g)
{
    if (dom_is_tiger(P.domain)) return g.boolean() ? 0 : 1;
    return g.uniform_int(P.S);
}

// Tiger::generateRandomAction (Tiger.cpp:21-25), FactoredTiger::generateRandomAction
__device__ __forceinline__ int domain_random_action(const Problem& P, Rng& g, int /*s*/) { return g.uniform_int(P.A); }

// True dynamics: Tiger::step (Tiger.cpp:40-82), FactoredTiger::step (FactoredTiger.cpp:77-122).
// Note the draw order when opening a door: observation coin first, then the next state.
__device__ __forceinline__ bool domain_step(const Problem& P, Rng& g, int& s, int a, int& o, double& r)
{
    const int d = P.domain;
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
    if (a != 2) return .5;
    const int loc = dom_is_tiger(P.domain) ? new_s : ((new_s < P.S / 2) ? 0 : 1);
    return (loc == o) ? .85 : .15;
}

// BADomainExtension::terminal / reward (TigerBAExtension.cpp:21-44, FactoredTigerBAExtension.cpp):
// the reward is looked up with the PRE-state s.
__device__ __forceinline__ bool ext_terminal(const Problem& P, int /*s*/, int a, int /*ns*/) { return dom_is_episodic(P.domain) && a != 2; }
__device__ __forceinline__ double ext_reward(const Problem& P, int s, int a, int /*ns*/)
{
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
template <class View>
__device__ __forceinline__ bool sim_step(const Problem& P, Rng& g, const View& cnt, int& s, int a, int& o, double& r, int& inc0,
                                         int& inc1)
{
    if (P.model == FBA_MODEL_POMDP) {
        inc0 = inc1 = -1;
        return domain_step(P, g, s, a, o, r);
    }
    const int S = P.S, A = P.A, O = P.O;
    const int t_off = s * A * S + a * S;
    const int ns    = sample_expected_mult(g, cnt, t_off, S);
    const int o_off = P.phi_len + a * S * O + ns * O;
    o               = sample_expected_mult(g, cnt, o_off, O);
    const bool t    = ext_terminal(P, s, a, ns);
    r               = ext_reward(P, s, a, ns);
    inc0            = t_off + ns;
    inc1            = o_off + o;
    s               = ns;
    return t;
}

// POMDP::computeObservationProbability of the simulator in use
// (BAPOMDP.cpp:93-99 -> BAFlatModel::computeObservationProbability BAFlatModel.cpp:106-124)
template <class View>
__device__ __forceinline__ double sim_obs_prob(const Problem& P, const View& cnt, int new_s, int a, int o)
{
    if (P.model == FBA_MODEL_POMDP) return domain_obs_prob(P, o, a, new_s);
    if (P.O == 1) return 1.0;
    return expected_mult_at(cnt, P.phi_len + a * P.S * P.O + new_s * P.O, P.O, o);
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
