/*
 * oracle/orc_rng.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the random-number layer used by samkatt/fba-pomdp's hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or
 * call anything under oracle/; the product (fba_pomdp_amd/) never does.
 *
 * Two back-ends behind one interface:
 *
 *   ORC_RNG_MT      one global mt19937 stream with GNU libstdc++-11 distribution adaptors,
 *                   written out explicitly (no <random>) so results do not depend on the
 *                   libstdc++ of the machine the tests run on.  Mirrors
 *                   reference src/utils/random.cpp:11-16,76-115 and the adaptors it picks:
 *                     uniform_rand01  -> generate_canonical<double,53> (2 engine words)
 *                                        /usr/include/c++/11/bits/random.tcc:3348-3383
 *                     boolean         -> bernoulli_distribution(0.5)   (2 engine words)
 *                     I(n)            -> uniform_int_distribution<int>(0,n-1), Lemire
 *                                        /usr/include/c++/11/bits/uniform_int_dist.h:246-271
 *                     seed(str)       -> std::seed_seq(chars) + mt19937::seed(seq)
 *
 *   ORC_RNG_PHILOX  Philox4x32-10 counter streams, the generator the HIP engine uses.
 *                   A stream is addressed by (seed, run, episode, t, phase, unit); every draw
 *                   consumes 64 bits (2 words); block b of a stream is
 *                   philox(key=seed, ctr=(b, unit, phase | t<<8 | episode<<16, run)).
 *                   In this mode orc_rng_stream() re-positions the generator; in MT mode it
 *                   is a no-op, so the same algorithm code serves both modes.
 */
#ifndef ORC_RNG_H
#define ORC_RNG_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_RNG_MT = 0, ORC_RNG_PHILOX = 1 };

/* stream phases (shared numbering with include/fba_hip.h FBA_PHASE_*) */
enum {
    ORC_PH_INIT      = 0, /* belief initiate: unit = particle index                    */
    ORC_PH_RESET     = 1, /* resetDomainStateDistribution: unit = particle index       */
    ORC_PH_START     = 2, /* true-environment start state: unit = 0                    */
    ORC_PH_SEARCH    = 3, /* planner: unit = simulation index; n = probe; n+1 = final  */
    ORC_PH_ENV       = 4, /* true-environment step: unit = 0                           */
    ORC_PH_REJECT    = 5, /* rejection sampling: unit = attempt index                  */
    ORC_PH_IS_UPDATE = 6, /* importance update: unit = particle index                  */
    ORC_PH_RESAMPLE  = 7, /* importance resample: unit = output particle index         */
    ORC_PH_REINVIG   = 8, /* reinvigoration / cheating: unit = index of the bred (copied) particle */
    ORC_PH_INIT_FC   = 9, /* fully connected filter, initiate: unit = particle index    */
    ORC_PH_RESET_FC  = 10,/* fully connected filter, reset: unit = particle index       */
    ORC_PH_REJECT_FC = 11,/* fully connected filter, rejection: unit = attempt index    */
    ORC_PH_RESET_SH  = 12,/* incubator belief, shadow filter, reset: unit = particle index */
    ORC_PH_INIT_SH   = 13 /* incubator belief, shadow filter, initiate: unit = index of the bred particle */
};

typedef struct orc_rng {
    int mode;
    /* mt19937 state */
    uint32_t mt[624];
    int mti;
    /* philox state */
    uint32_t key[2];
    uint32_t ctr[4]; /* ctr[0] = block index of the cached block */
    uint32_t blk[4];
    uint32_t draw;   /* number of 64-bit draws consumed in the current stream */
    int blk_valid;
    /* stream address components kept so callers can set them piecewise */
    uint32_t run, episode, t;
    /* statistics */
    uint64_t words; /* engine words (MT) or draws (philox) consumed */
} orc_rng;

void orc_rng_init_mt_u32(orc_rng* g, uint32_t seed);                   /* mt19937::seed(value)  */
void orc_rng_init_mt_str(orc_rng* g, const char* seed, size_t len);    /* seed_seq(chars)       */
void orc_rng_init_philox(orc_rng* g, uint64_t seed);

/* stream addressing (no-ops in MT mode) */
void orc_rng_episode(orc_rng* g, uint32_t run, uint32_t episode, uint32_t t);
void orc_rng_stream(orc_rng* g, uint32_t phase, uint32_t unit);

/* raw engine output */
uint32_t orc_mt_next(orc_rng* g);
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* the three primitives the hot path uses */
double orc_u01(orc_rng* g);            /* rnd::uniform_rand01            random.cpp:100 */
int orc_bool(orc_rng* g);              /* rnd::boolean                   random.cpp:90  */
int orc_int(orc_rng* g, int n);        /* uniform_int_distribution<int>(0,n-1)          */
int orc_slow_int(orc_rng* g, int lo, int hi); /* rnd::slowRandomInt      random.cpp:111 */

/* std::seed_seq::generate restated (for tests) */
void orc_seed_seq_generate(const uint32_t* v, size_t s, uint32_t* out, size_t n);

#ifdef __cplusplus
}
#endif
#endif
