// TEST INFRASTRUCTURE: a recording stand-in for libfba_hip's C-ABI (include/fba_hip.h) -- NOT the product and not a
// CPU path: it computes nothing.  tests/test_adapters_run.py links the adapters (fba_pomdp_amd/csrc/host/adapters.hpp)
// and the reference's own Boost-free objects (episode::run, Tiger, ...) against it and checks the sequence of calls
// and stream positions the reference's loops produce.
#include <cstdio>
#include <cstring>

#include "fba_hip.h"

struct fba_ctx {
    fba_config cfg;
    int run, episode, t;
    int selects;
};
static fba_ctx g_ctx;

extern "C" {
int fba_abi_version(void) { return FBA_ABI_VERSION; }
int fba_domain_sizes(const fba_ctx*, int32_t* S, int32_t* A, int32_t* O) { *S = 2; *A = 3; *O = 2; return FBA_OK; }
int fba_counts_len(const fba_ctx*) { return 24; }
int fba_get_factored_layout(const fba_ctx*, fba_factored_layout*) { return FBA_EINVAL; }
int fba_belief_get_particle(fba_ctx*, int32_t, int32_t index, int32_t* state, double* weight, float* counts)
{
    if (state) *state = index & 1;
    if (weight) *weight = 1.0;
    if (counts) for (int k = 0; k < 24; ++k) counts[k] = 100.f * (float)index + (float)k;   // particle `index`, cell k
    std::printf("get_particle %d\n", index);
    return FBA_OK;
}
void fba_default_config(fba_config* cfg) { std::memset(cfg, 0, sizeof *cfg); cfg->particles = 8; cfg->horizon = 4; }
int fba_create(const fba_config* cfg, fba_ctx** out)
{
    g_ctx.cfg = *cfg;
    g_ctx.run = g_ctx.episode = g_ctx.t = -9;
    g_ctx.selects = 0;
    *out = &g_ctx;
    std::printf("create slots=%d\n", cfg->slots);
    return FBA_OK;
}
void fba_destroy(fba_ctx*) { std::printf("destroy\n"); }
const char* fba_last_error(const fba_ctx*) { return "stub"; }
int fba_set_position(fba_ctx* c, const int32_t* run, const int32_t* episode, const int32_t* t)
{
    if (run) c->run = *run;
    if (episode) c->episode = *episode;
    if (t) c->t = *t;
    return FBA_OK;
}
int fba_belief_init(fba_ctx* c) { std::printf("init run=%d episode=%d t=%d\n", c->run, c->episode, c->t); return FBA_OK; }
int fba_belief_reset_domain_state(fba_ctx* c) { std::printf("reset run=%d episode=%d t=%d\n", c->run, c->episode, c->t); return FBA_OK; }
int fba_select_action(fba_ctx* c, const int32_t* hist_len, const uint8_t*, int32_t* action)
{
    // listen twice, then open a door: episodes of three steps, the last one terminal (no belief update after it)
    *action = (*hist_len < 2) ? 2 : 0;
    std::printf("select run=%d episode=%d t=%d hist=%d\n", c->run, c->episode, c->t, *hist_len);
    ++c->selects;
    return FBA_OK;
}
int fba_belief_update(fba_ctx* c, const int32_t* action, const int32_t* obs, const uint8_t*)
{
    std::printf("update run=%d episode=%d t=%d a=%d o=%d\n", c->run, c->episode, c->t, *action, *obs);
    return FBA_OK;
}
int fba_belief_get(fba_ctx* c, int32_t, int32_t* state, double* weight, float*)
{
    for (int i = 0; i < c->cfg.particles; ++i) {
        if (state) state[i] = 1;
        if (weight) weight[i] = 1.0;
    }
    std::printf("get\n");
    return FBA_OK;
}
}
