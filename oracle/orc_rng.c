/*
 * oracle/orc_rng.c -- TEST INFRASTRUCTURE ONLY (see orc_rng.h).
 */
#include "orc_rng.h"

#include <math.h>
#include <string.h>

/* ------------------------------------------------------------------ mt19937 */

static void mt_twist(orc_rng* g)
{
    uint32_t* mt = g->mt;
    int k;
    for (k = 0; k < 624 - 397; ++k) {
        uint32_t y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu);
        mt[k]      = mt[k + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; k < 623; ++k) {
        uint32_t y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu);
        mt[k]      = mt[k + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    {
        uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[623]    = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    g->mti = 0;
}

uint32_t orc_mt_next(orc_rng* g)
{
    uint32_t y;
    if (g->mti >= 624) mt_twist(g);
    y = g->mt[g->mti++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    g->words++;
    return y;
}

void orc_rng_init_mt_u32(orc_rng* g, uint32_t seed)
{
    int i;
    memset(g, 0, sizeof(*g));
    g->mode  = ORC_RNG_MT;
    g->mt[0] = seed;
    for (i = 1; i < 624; ++i)
        g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->mti = 624;
}

/* std::seed_seq::generate, ISO C++ [rand.util.seedseq]/8, all arithmetic mod 2^32 */
void orc_seed_seq_generate(const uint32_t* v, size_t s, uint32_t* b, size_t n)
{
    size_t k, t, p, q, m;
    if (n == 0) return;
    for (k = 0; k < n; ++k) b[k] = 0x8b8b8b8bu;
    t = (n >= 623) ? 11 : (n >= 68) ? 7 : (n >= 39) ? 5 : (n >= 7) ? 3 : (n - 1) / 2;
    p = (n - t) / 2;
    q = p + t;
    m = (s + 1 > n) ? s + 1 : n;
    for (k = 0; k < m; ++k) {
        uint32_t arg = b[k % n] ^ b[(k + p) % n] ^ b[(k + n - 1) % n];
        uint32_t r1  = 1664525u * (arg ^ (arg >> 27));
        uint32_t r2  = r1;
        if (k == 0)
            r2 += (uint32_t)s;
        else if (k <= s)
            r2 += (uint32_t)(k % n) + v[k - 1];
        else
            r2 += (uint32_t)(k % n);
        b[(k + p) % n] += r1;
        b[(k + q) % n] += r2;
        b[k % n] = r2;
    }
    for (k = m; k < m + n; ++k) {
        uint32_t arg = b[k % n] + b[(k + p) % n] + b[(k + n - 1) % n];
        uint32_t r3  = 1566083941u * (arg ^ (arg >> 27));
        uint32_t r4  = r3 - (uint32_t)(k % n);
        b[(k + p) % n] ^= r3;
        b[(k + q) % n] ^= r4;
        b[k % n] = r4;
    }
}

/* reference src/utils/random.cpp:76-83: std::seed_seq(seed_str.begin(), seed_str.end()) */
void orc_rng_init_mt_str(orc_rng* g, const char* seed, size_t len)
{
    uint32_t v[256];
    size_t i;
    int zero = 1;
    memset(g, 0, sizeof(*g));
    g->mode = ORC_RNG_MT;
    if (len > 256) len = 256;
    for (i = 0; i < len; ++i) v[i] = (uint32_t)(unsigned char)seed[i];
    orc_seed_seq_generate(v, len, g->mt, 624);
    /* mersenne_twister_engine::seed(Sseq&): all-zero guard */
    if ((g->mt[0] & 0x80000000u) != 0) zero = 0;
    for (i = 1; zero && i < 624; ++i)
        if (g->mt[i] != 0) zero = 0;
    if (zero) g->mt[0] = 0x80000000u;
    g->mti = 624;
}

/* ------------------------------------------------------------------ philox */

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    int r;
    for (r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void orc_rng_init_philox(orc_rng* g, uint64_t seed)
{
    memset(g, 0, sizeof(*g));
    g->mode   = ORC_RNG_PHILOX;
    g->key[0] = (uint32_t)seed;
    g->key[1] = (uint32_t)(seed >> 32);
}

void orc_rng_episode(orc_rng* g, uint32_t run, uint32_t episode, uint32_t t)
{
    g->run     = run;
    g->episode = episode;
    g->t       = t;
}

void orc_rng_stream(orc_rng* g, uint32_t phase, uint32_t unit)
{
    if (g->mode != ORC_RNG_PHILOX) return;
    g->ctr[1]    = unit;
    g->ctr[2]    = (phase & 0xffu) | ((g->t & 0xffu) << 8) | ((g->episode & 0xffffu) << 16);
    g->ctr[3]    = g->run;
    g->draw      = 0;
    g->blk_valid = 0;
}

static uint64_t philox_next64(orc_rng* g)
{
    uint32_t b = g->draw >> 1;
    uint32_t h = (g->draw & 1u) * 2u;
    if (!g->blk_valid || g->ctr[0] != b) {
        g->ctr[0] = b;
        orc_philox4x32_10(g->ctr, g->key, g->blk);
        g->blk_valid = 1;
    }
    g->draw++;
    g->words++;
    return ((uint64_t)g->blk[h + 1] << 32) | g->blk[h];
}

/* ------------------------------------------------------------------ primitives */

static double mt_canonical(orc_rng* g)
{
    /* generate_canonical<double,53> over a 32-bit engine: k = 2 words */
    double sum = (double)orc_mt_next(g);
    double ret;
    sum += (double)orc_mt_next(g) * 4294967296.0;
    ret = sum / 18446744073709551616.0;
    if (ret >= 1.0) ret = nextafter(1.0, 0.0);
    return ret;
}

double orc_u01(orc_rng* g)
{
    if (g->mode == ORC_RNG_MT) return mt_canonical(g);
    return (double)(philox_next64(g) >> 11) * (1.0 / 9007199254740992.0);
}

int orc_bool(orc_rng* g)
{
    if (g->mode == ORC_RNG_MT) return mt_canonical(g) < 0.5;
    return (philox_next64(g) >> 63) == 0; /* same event as u01 < 0.5 */
}

int orc_int(orc_rng* g, int n)
{
    if (g->mode == ORC_RNG_MT) {
        /* uniform_int_distribution<int>(0, n-1) on a 32-bit URBG: Lemire, 64-bit product */
        uint32_t range = (uint32_t)n;
        uint64_t prod;
        uint32_t low;
        if (range == 0) { /* urange == urngrange: full 32-bit range */
            return (int)orc_mt_next(g);
        }
        prod = (uint64_t)orc_mt_next(g) * (uint64_t)range;
        low  = (uint32_t)prod;
        if (low < range) {
            uint32_t threshold = (0u - range) % range;
            while (low < threshold) {
                prod = (uint64_t)orc_mt_next(g) * (uint64_t)range;
                low  = (uint32_t)prod;
            }
        }
        return (int)(prod >> 32);
    }
    {
        uint64_t x = philox_next64(g);
        return (int)(((unsigned __int128)x * (unsigned __int128)(uint32_t)n) >> 64);
    }
}

int orc_slow_int(orc_rng* g, int lo, int hi)
{
    return lo + (int)floor(orc_u01(g) * (double)(hi - lo));
}
