// fba_engine.hip -- host side of libfba_hip.so: context, HBM allocation, prior tables,
// kernel orchestration (one HIP stream per ctx), HIP-event timing, and the C-ABI of
// include/fba_hip.h.  There is no CPU fallback anywhere in this file: without a gfx950 device
// fba_create fails with FBA_ENODEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <map>
#include <queue>
#include <vector>

#include "fba_kernels.h"

using namespace fba;

namespace {

thread_local std::string g_create_error;

struct EventPair {
    hipEvent_t a, b;
    int kind;
};

}  // namespace

struct fba_ctx {
    fba_config cfg{};
    Problem P{};
    DeviceState D{};
    hipStream_t stream = nullptr;
    std::string err;
    std::vector<void*> allocs;
    std::vector<float> prior;  // host copy, dense_C floats
    int dense_C          = 0;   // length of a particle's count table as the API sees it (= P.C unless P.packed)
    float* d_prior       = nullptr;
    float* d_prior_dense = nullptr;  // packed / history particles: the dense prior table on the device
    std::vector<float> prior_alt;    // history particles: the x / y transition nodes with the goal as third parent, [A][2][N*N*G*N]
    float* d_hist_base   = nullptr;  // ... and both tables on the device, rows padded to 16 bytes (HistLayout)
    float* d_hist_alt    = nullptr;
    uint8_t* d_hist_lds  = nullptr;  // ... and deduplicated: row ids + distinct rows (Problem::hist_lds)
    FDesc fdesc{};          // host copy of the factored model description
    FDesc* d_fdesc       = nullptr;
    GridDesc gdesc{};
    GridDesc* d_gdesc    = nullptr;
    CADesc cadesc{};
    CADesc* d_cadesc     = nullptr;
    SysDesc sysdesc{};
    SysDesc* d_sysdesc   = nullptr;
    ZigDesc* d_zig       = nullptr;
    double* d_uni_scan   = nullptr;
    double* d_log1p      = nullptr;
    int32_t* d_n_active  = nullptr;
    size_t returns_cap   = 0;
    bool started         = false;  // fba_run_ticks has set the slots up
    bool belief_ready    = false;
    std::vector<uint32_t> packed_updates;  // per slot: fba_belief_update calls since fba_belief_init (packed records hold uint16 increments)
    // timing
    bool timing = true;
    std::vector<EventPair> pending;
    std::vector<EventPair> free_events;
    double k_ms[FBA_K_COUNT]          = {0};
    uint64_t k_launches[FBA_K_COUNT]  = {0};
    uint64_t base_sim = 0, base_attempts = 0, base_particles = 0, base_entries = 0;
    std::vector<fba_trace_rec> trace_host;
};

namespace {

int fail(fba_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(c, call)                                                                             \
    do {                                                                                            \
        hipError_t _e = (call);                                                                     \
        if (_e != hipSuccess)                                                                       \
            return fail((c), FBA_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

template <typename T>
int dev_alloc(fba_ctx* c, T** p, size_t n, bool zero = true)
{
    if (n == 0) n = 1;
    void* q = nullptr;
    {
        const hipError_t e = hipMalloc(&q, n * sizeof(T));
        if (e != hipSuccess) {
            (void)hipGetLastError();  // the error is reported here; it must not stay behind for a later context's hipGetLastError()
            return fail(c, FBA_EHIP, "hipMalloc(&q, n * sizeof(T)) failed: %s (%s:%d)", hipGetErrorString(e), __FILE__, __LINE__);
        }
    }
    c->allocs.push_back(q);
    if (zero) HIPCHK(c, hipMemsetAsync(q, 0, n * sizeof(T), c->stream));
    *p = static_cast<T*>(q);
    return FBA_OK;
}

// gives a dev_alloc'ed buffer back before the ctx goes (buffers that are re-grown: the trace)
template <typename T>
void dev_free(fba_ctx* c, T*& p)
{
    if (!p) return;
    auto it = std::find(c->allocs.begin(), c->allocs.end(), static_cast<void*>(p));
    if (it != c->allocs.end()) c->allocs.erase(it);
    (void)hipFree(p);
    p = nullptr;
}

// room for `want` trace records (and, with trace = 2, their histograms: FBA_TRACE_HIST_BINS words each, so the cap is lower there --
// 2^18 records = 64 MB of histograms instead of 1 GiB; later records are counted, not kept)
int ensure_trace(fba_ctx* c, size_t want)
{
    const bool hist = c->cfg.trace >= 2 && c->P.S <= FBA_TRACE_HIST_BINS && !c->P.nested;
    const size_t cap = std::min<size_t>(want, hist ? (size_t)1 << 18 : (size_t)1 << 22);
    if ((size_t)c->D.trace_cap < cap) {
        int rc;
        HIPCHK(c, hipStreamSynchronize(c->stream));   // (nothing in flight reads the old buffers)
        dev_free(c, c->D.trace);
        dev_free(c, c->D.trace_hist);
        if ((rc = dev_alloc(c, &c->D.trace, cap))) return rc;
        if (hist)
            if ((rc = dev_alloc(c, &c->D.trace_hist, cap * FBA_TRACE_HIST_BINS))) return rc;
        c->D.trace_cap = (int32_t)cap;
    }
    HIPCHK(c, hipMemsetAsync(c->D.trace_count, 0, sizeof(int32_t), c->stream));
    return FBA_OK;
}

// a device buffer that is freed on every way out of the function (the selftests return early through HIPCHK)
template <typename T>
struct ScratchBuf {
    T* p = nullptr;
    ~ScratchBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n * sizeof(T)); }
};

bool is_tiger(int d) { return d == FBA_DOM_TIGER_EPISODIC || d == FBA_DOM_TIGER_CONTINUOUS; }
bool is_ftiger(int d) { return d == FBA_DOM_FTIGER_EPISODIC || d == FBA_DOM_FTIGER_CONTINUOUS; }

bool is_ca(int d) { return d == FBA_DOM_COLLISION_AVOID || d == FBA_DOM_COLLISION_AVOID_CENTERED; }
bool is_sys(int d) { return d == FBA_DOM_SYSADMIN_INDEPENDENT || d == FBA_DOM_SYSADMIN_LINEAR; }

// SysAdmin (reference src/domains/sysadmin/SysAdmin.cpp:12: parameters .025f, .95f, .95f, 1.0f, .075f).
// keep[n] = (1 - fail_prob) * pow(1 - fail_neighbour_factor, n) as SysAdmin::step (:116-118)
// evaluates it: float factor times a double pow, libm on the host so the device only looks it up.
constexpr float SYS_FAIL = .025f, SYS_OBSERVE = .95f, SYS_REBOOT = .95f, SYS_NEIGHBOUR = .075f;
void build_sysadmin(SysDesc& d, int n)
{
    d.N = n; d.pad = 0;
    for (int k = 0; k < 3; ++k) d.keep[k] = (double)(1 - SYS_FAIL) * std::pow((double)(1 - SYS_NEIGHBOUR), (double)k);
}
int sys_failing_neighbours(const SysDesc& d, bool linear, int comp, int s)  // SysAdmin::numFailingNeighbours :221-246
{
    if (!linear) return 0;
    int n = 0;
    if (comp > 0 && !((s >> (comp - 1)) & 1)) n++;
    if (comp < d.N - 1 && !((s >> (comp + 1)) & 1)) n++;
    return n;
}

// SysAdminFlatPrior::precomputeFlatPrior (SysAdminFlatPrior.cpp:38-247): every transition
// probability times 10000, enumerated by a recursion over the computers N-1 .. 0 that multiplies
// the per-computer outcomes in double and writes float counts at its leaves.  The order of the
// leaves is the reference's, because reboot-branch leaves overwrite counts written earlier; so is
// the quirk that setTrueTCounts (:168-172) hands numFailingNeighbours the ACTION index.
struct SysFlat {
    const Problem& P;
    const SysDesc& d;
    float* phi;
    double keep(int comp, int s) const { return d.keep[sys_failing_neighbours(d, P.domain == FBA_DOM_SYSADMIN_LINEAR, comp, s)]; }
    void leaf_reboot(int s, int ns, double prob, int reb) { phi[(s * P.A + d.N + reb) * P.S + ns] = (float)prob * 10000.f; }
    void recur_reboot(int s, int ns, int comp, double acc, int reb)  // :188-234
    {
        if (comp < 0) return leaf_reboot(s, ns, acc, reb);
        const int ns_fail = ns & ~(1 << comp);
        if (!((s >> comp) & 1)) return recur_reboot(s, ns_fail, comp - 1, acc, reb);
        const double fail = 1 - keep(comp, s);
        recur_reboot(s, ns, comp - 1, acc * (1 - fail), reb);
        recur_reboot(s, ns_fail, comp - 1, acc * fail, reb);
    }
    void leaf(int s, int ns, double prob)  // setTrueTCounts :150-186
    {
        const int N = d.N;
        for (int a = 0; a < N; ++a) phi[(s * P.A + a) * P.S + ns] = (float)prob * 10000.f;
        for (int a = N; a < 2 * N; ++a) {
            if ((ns >> (a - N)) & 1) {
                const double fail = 1 - keep(a, s);
                phi[(s * P.A + a) * P.S + ns] = 10000.f * (float)(prob + (prob * fail / (1 - fail) * SYS_REBOOT));
            } else {
                phi[(s * P.A + a) * P.S + ns] = 10000.f * (float)(prob * (1 - SYS_REBOOT));
            }
        }
    }
    void recur(int s, int ns, int comp, double acc)  // :93-148
    {
        if (comp < 0) return leaf(s, ns, acc);
        const int ns_fail = ns & ~(1 << comp);
        if (!((s >> comp) & 1)) {
            recur(s, ns_fail, comp - 1, acc);
            recur_reboot(s, ns_fail, comp - 1, acc * (1 - SYS_REBOOT), comp);
            recur_reboot(s, ns, comp - 1, acc * SYS_REBOOT, comp);
        } else {
            const double fail = 1 - keep(comp, s);
            recur(s, ns, comp - 1, acc * (1 - fail));
            recur(s, ns_fail, comp - 1, acc * fail);
        }
    }
};
double normal_cdf(double x) { return .5 + .5 * std::erf(x / (1 * std::sqrt(2.0))); }  // rnd::normal::cdf random.cpp:119-124

// Collision avoidance tables (reference CollisionAvoidance.cpp ctor :100-142)
void build_collision_avoidance(CADesc& c, int W, int H, int n, bool random_start)
{
    std::memset(&c, 0, sizeof c);
    c.W = W; c.H = H; c.n = n; c.Hn = 1;
    for (int k = 0; k < n; ++k) c.Hn *= H;
    for (int d = 0; d < H; ++d) c.err[d] = normal_cdf(d + .5) - normal_cdf(d - .5);
    for (int k = 0; k < 19; ++k) c.phi[k] = normal_cdf((k - 9) + .5);
    if (random_start) {  // every state with x = W-1, probability 1.f / pow(H, n + 1) each
        c.start_v   = (float)(1.f / std::pow(H, n + 1));
        c.start_i0  = (W - 1) * H * c.Hn;
        c.start_cnt = H * c.Hn;
        c.start_total = 0;
        for (int k = 0; k < c.start_cnt; ++k) c.start_total += c.start_v;  // categoricalDistr::setRawValue
    } else {             // agent and obstacles in the middle row
        int obs = 0;
        for (int k = 0; k < n; ++k) obs = obs * H + H / 2;
        c.start_v = 1; c.start_total = 1; c.start_cnt = 1;
        c.start_i0 = ((W - 1) * H + H / 2) * c.Hn + obs;
    }
}

// ziggurat tables of rnd::initiate() (reference src/utils/random.cpp:47-74), libm on the host
void build_ziggurat(ZigDesc& z)
{
    double tn = 3.442619855899;
    const double m1 = 2147483648.0, vn = 9.91256303526217e-3, q = vn / std::exp(-.5 * tn * tn);
    z.ul[0]   = (uint32_t)(unsigned long)((tn / q) * m1);
    z.ul[1]   = 0;
    z.wn[0]   = q / m1;
    z.wn[127] = tn / m1;
    z.fn[0]   = 1.;
    z.fn[127] = std::exp(-.5 * tn * tn);
    for (int i = 126; i > 0; --i) {
        const double dn = std::sqrt(-2 * std::log(vn / tn + std::exp(-.5 * tn * tn)));
        z.ul[i + 1]     = (uint32_t)(unsigned long)((dn / tn) * m1);
        z.fn[i]         = std::exp(-.5 * dn * dn);
        z.wn[i]         = dn / m1;
        tn              = dn;
    }
}

// GridWorld geometry (reference src/domains/gridworld/GridWorld.cpp)
void build_gridworld(GridDesc& g, int N)
{
    std::memset(&g, 0, sizeof g);
    const int edge = N - 1;
    g.N = N;
    int G = 0;
    // goalLocations :124-152
    const int start = (N < 5) ? N - 2 : (N < 7) ? N - 3 : N - 4;
    for (int i = start; i < N - 1; ++i) {
        g.goal[G][0] = i; g.goal[G][1] = edge; ++G;
        g.goal[G][0] = edge; g.goal[G][1] = i; ++G;
    }
    g.goal[G][0] = edge; g.goal[G][1] = edge; ++G;
    if (N > 3) { g.goal[G][0] = edge - 1; g.goal[G][1] = edge - 1; ++G; }
    if (N > 6) {
        g.goal[G][0] = edge - 2; g.goal[G][1] = edge - 1; ++G;
        g.goal[G][0] = edge - 1; g.goal[G][1] = edge - 2; ++G;
    }
    g.G = G;
    // generateSlowLocations :78-103
    int ns = 0;
    if (N > 5) { g.slow[ns][0] = 1; g.slow[ns][1] = 1; ++ns; }
    if (N == 3) { g.slow[ns][0] = 1; g.slow[ns][1] = 1; ++ns; }
    else if (N < 7) {
        g.slow[ns][0] = edge - 1; g.slow[ns][1] = edge - 2; ++ns;
        g.slow[ns][0] = edge - 2; g.slow[ns][1] = edge - 1; ++ns;
    } else {
        g.slow[ns][0] = edge - 1; g.slow[ns][1] = edge - 3; ++ns;
        g.slow[ns][0] = edge - 3; g.slow[ns][1] = edge - 1; ++ns;
        g.slow[ns][0] = edge - 2; g.slow[ns][1] = edge - 2; ++ns;
    }
    g.nslow = ns;
    // _obs_displacement_probs (ctor :60-70): {.8, .1, .05, ..., last one repeated}
    g.disp[0] = (float)(1 - .2);
    double prob = .2;
    for (int i = 1; i < N - 1; ++i) { prob *= .5; g.disp[i] = (float)prob; }
    g.disp[N - 1] = (float)prob;
}
bool gw_slow_at_h(const GridDesc& g, int x, int y)
{
    for (int i = 0; i < g.nslow; ++i)
        if (g.slow[i][0] == x && g.slow[i][1] == y) return true;
    return false;
}
void gw_move_h(const GridDesc& g, int a, int& x, int& y)
{
    const int N = g.N;
    if (a == 0) { if (y != N - 1) ++y; }
    else if (a == 2) { if (y != 0) --y; }
    else if (a == 1) { if (x != N - 1) ++x; }
    else { if (x != 0) --x; }
}
float gw_obs_displ_prob_h(const GridDesc& g, int loc, int observed)
{
    const int disp = std::abs(loc - observed);
    float res = (disp == 0) ? (float)(1 - .2) : (float)((double)g.disp[disp] * .5);
    if (observed == g.N - 1 || observed == 0)
        for (int i = disp + 1; i < g.N; ++i) res = (float)((double)res + (double)g.disp[i] * .5);
    return res;
}

// Tabular prior count tables, built on the host once per ctx (cold path).
// TigerBAPrior (reference src/domains/tiger/TigerPriors.cpp:14-43): every count 5000 except
// listen: T off-diagonal 0, O = (.85 - noise) * C / (.15 + noise) * C.
// FactoredTigerFlatPrior (src/domains/tiger/FactoredTigerPriors.cpp:18-88): same over S = 2^(K+1)
// states, tiger location = (s < S/2 ? LEFT : RIGHT).
int build_tabular_prior(fba_ctx* c)
{
    const Problem& P = c->P;
    const int S = P.S, A = P.A, O = P.O;
    const float noise = c->cfg.noise, total = c->cfg.counts_total;
    if (is_sys(P.domain)) {  // SysAdminFlatPrior: zero-initialised BAFlatModel, --noise / -C unused
        c->prior.assign((size_t)c->dense_C, 0.f);
        SysFlat flat{P, c->sysdesc, c->prior.data()};
        for (int s = 0; s < S; ++s) flat.recur(s, S - 1, c->sysdesc.N - 1, 1);
        float* psi = c->prior.data() + P.phi_len;  // :60-90: the observation tells the operated computer's bit
        const float high = 10000.f * SYS_OBSERVE, low = 10000.f * (1 - SYS_OBSERVE);
        for (int a = 0; a < A; ++a)
            for (int ns = 0; ns < S; ++ns) {
                const int up = (ns >> (a % c->sysdesc.N)) & 1;
                psi[(a * S + ns) * O + up]     = high;
                psi[(a * S + ns) * O + 1 - up] = low;
            }
        return FBA_OK;
    }
    if (is_ca(P.domain)) {
        // CollisionAvoidanceTablePrior (CollisionAvoidancePriors.cpp:65-208).  For state (x, y, b) and every
        // obstacle configuration b': transition count to (x-1, y', b') = prod_i obstacleTransProb(b_i, b'_i) * -C
        // (none from x = 0; the product stops at its first zero), observation count of b' in that state =
        // prod_i observationDistr(H, b_i)[b'_i] * 10000.
        const CADesc& ca = c->cadesc;
        const int W = ca.W, H = ca.H, n = ca.n, Hn = ca.Hn;
        if (!(noise < .5 && noise > -.5)) return fail(c, FBA_EINVAL, "CollisionAvoidanceTablePrior needs -.5 < noise < .5 (is: %f)", noise);
        c->prior.assign((size_t)c->dense_C, 0.f);
        float* phi = c->prior.data();
        float* psi = c->prior.data() + P.phi_len;
        auto trans = [&](int y, int ny) -> double {  // obstacleTransProb :136-170
            const int dist = std::abs(y - ny);
            if (dist > 1) return 0;
            if (y == 0 || y == H - 1) return dist == 0 ? .75 + .5 * noise : .25 - .5 * noise;
            return dist == 0 ? .5 + noise : .25 - .5 * noise;
        };
        auto seen = [&](int pos, int oy) -> float {  // observationDistr(height, obstacle_pos) :46-63
            if (oy == 0) return (float)normal_cdf(-pos + .5);
            if (oy == H - 1) return (float)normal_cdf(-(H - 1 - pos) + .5);
            const int dist = std::abs(oy - pos);
            return (float)(normal_cdf(dist + .5) - normal_cdf(dist - .5));
        };
        for (int ob = 0; ob < Hn; ++ob)
            for (int nob = 0; nob < Hn; ++nob) {
                int b[MAXF], nb[MAXF], r1 = ob, r2 = nob;
                for (int i = n - 1; i >= 0; --i) { b[i] = r1 % H; r1 /= H; nb[i] = r2 % H; r2 /= H; }
                double tprob = 1, oprob = 1;
                for (int i = 0; i < n && tprob != 0; ++i) tprob *= trans(b[i], nb[i]);
                for (int i = 0; i < n; ++i) oprob *= seen(b[i], nb[i]);
                for (int x = 0; x < W; ++x)
                    for (int y = 0; y < H; ++y) {
                        const int s = (x * H + y) * Hn + ob;
                        for (int a = 0; a < A; ++a) {
                            psi[((size_t)a * S + s) * O + nob] = (float)(oprob * 10000);
                            if (x == 0 || tprob == 0) continue;
                            int ny = y + a - 1;
                            if (ny == -1 || ny == H) ny = y;
                            phi[((size_t)s * A + a) * S + ((x - 1) * H + ny) * Hn + nob] = (float)(tprob * total);
                        }
                    }
            }
        return FBA_OK;
    }
    if (P.domain == FBA_DOM_GRIDWORLD) {
        // GridWorldFlatBAPrior (GridWorldBAPriors.cpp:21-156).  Transition counts accumulate (a move
        // into a wall and a failed move are the same cell; on a goal the next goal is uniform);
        // observation counts = P(o | s') * 100000 for the observations that carry s' own goal.
        const GridDesc& g = c->gdesc;
        const int N = g.N, G = g.G;
        if (noise < 0 || noise > (1 - .15))
            return fail(c, FBA_EINVAL, "Gridworld expects noise in between 0 and %f (received %f)", 1 - .15, noise);
        c->prior.assign((size_t)c->dense_C, 0.f);
        float* phi = c->prior.data();
        float* psi = c->prior.data() + P.phi_len;
        for (int a = 0; a < A; ++a)
            for (int s = 0; s < S; ++s) {
                const int ax = s / (N * G), ay = (s / G) % N, gl = s % G;
                const bool on_goal = g.goal[gl][0] == ax && g.goal[gl][1] == ay;
                const float success = gw_slow_at_h(g, ax, ay) ? (float)(.15 + noise) : (float).95;
                const float goal_prob = (float)1 / (float)G;
                float* row = phi + ((size_t)s * A + a) * S;
                int nx = ax, ny = ay;
                gw_move_h(g, a, nx, ny);
                if (on_goal) {  // the move fails / succeeds, then the goal is re-drawn uniformly (:76-88, :104-117)
                    const float stay = (1 - success) * goal_prob, go = success * goal_prob;
                    for (int g2 = 0; g2 < G; ++g2) row[(ax * N + ay) * G + g2] += stay * total;
                    for (int g2 = 0; g2 < G; ++g2) row[(nx * N + ny) * G + g2] += go * total;
                } else {
                    const float stay = 1 - success;
                    row[s] += stay * total;
                    row[(nx * N + ny) * G + gl] += success * total;
                }
                for (int x = 0; x < N; ++x)
                    for (int y = 0; y < N; ++y) {
                        const double prob = (double)(gw_obs_displ_prob_h(g, ax, x) * gw_obs_displ_prob_h(g, ay, y));
                        psi[((size_t)a * S + s) * O + (x * N + y) * G + gl] = (float)(prob * 100000.0f);
                    }
            }
        return FBA_OK;
    }
    if (noise <= -.15 || noise > .3) return fail(c, FBA_EINVAL, "noise has to be between -.15 and .3");
    const float acc = (.85f - noise) * total, inacc = (.15f + noise) * total;
    c->prior.assign((size_t)c->dense_C, 5000.f);
    float* phi = c->prior.data();
    float* psi = c->prior.data() + P.phi_len;
    const int listen = 2;
    if (is_tiger(P.domain)) {
        phi[1 * A * S + listen * S + 0] = 0;
        phi[0 * A * S + listen * S + 1] = 0;
        psi[listen * S * O + 1 * O + 1] = acc;
        psi[listen * S * O + 1 * O + 0] = inacc;
        psi[listen * S * O + 0 * O + 1] = inacc;
        psi[listen * S * O + 0 * O + 0] = acc;
    } else if (is_ftiger(P.domain)) {
        for (int s = 0; s < S; ++s)
            for (int ns = 0; ns < S; ++ns)
                if (s != ns) phi[s * A * S + listen * S + ns] = 0;
        for (int s = 0; s < S; ++s) {
            const bool left = s < S / 2;
            psi[listen * S * O + s * O + (left ? 0 : 1)] = acc;
            psi[listen * S * O + s * O + (left ? 1 : 0)] = inacc;
        }
    } else {
        return fail(c, FBA_EINVAL, "domain %d has no built-in tabular prior; use fba_set_model_tabular", P.domain);
    }
    return FBA_OK;
}

void fdesc_steps(const int32_t* size, int n, int32_t* step)
{
    step[n - 1] = 1;
    for (int i = n - 2; i >= 0; --i) step[i] = step[i + 1] * size[i + 1];
}

// Factored model description + base prior record for episodic/continuous factored tiger.
// FactoredTigerFactoredPrior ctor (reference src/domains/tiger/FactoredTigerPriors.cpp:95-195):
//   T listen (a = 2): feature f depends on feature f only, count 5000 on "keeps its value";
//   T open: no parents, {5000, 5000};  O open: no parents, {5000, 5000};
//   O listen: parents are per particle (structure prior), filled on the device by
//   ftiger_set_observation_model; the base record carries the correct structure {tiger location}.
// GridWorldFactBAPrior ctor + preComputePrior (reference
// src/domains/gridworld/GridWorldBAPriors.cpp:158-198, 316-413): correct-structure prior.
// Features {x, y, goal}; T parents x:{x,y}, y:{x,y}, goal:{x,y,goal}; O parents x_obs:{x},
// y_obs:{y}, goal_obs:{goal}.  The x / y transition nodes own room for the goal as a third parent
// (structure prior match-uniform, filled per particle on the device).
int build_gridworld_factored_prior(fba_ctx* c)
{
    Problem& P = c->P;
    const GridDesc& g = c->gdesc;
    const int N = g.N, G = g.G, A = P.A;
    const float noise = c->cfg.noise, total = c->cfg.counts_total, static_total = 100000;
    if (noise < 0 || noise > (1 - .15)) return fail(c, FBA_EINVAL, "Gridworld expects noise in between 0 and %f (received %f)", 1 - .15, noise);
    if (c->cfg.structure_prior != FBA_SP_NONE && c->cfg.structure_prior != FBA_SP_MATCH_UNIFORM)
        return fail(c, FBA_EINVAL, "Please enter a valid structure noise option for the GridWorld problem ('match-uniform' or 'match-counts')");
    FDesc& d = c->fdesc;
    std::memset(&d, 0, sizeof d);
    d.FS = d.FO = 3;
    d.Ssz[0] = d.Ssz[1] = d.Osz[0] = d.Osz[1] = N;
    d.Ssz[2] = d.Osz[2] = G;
    fdesc_steps(d.Ssz, 3, d.Sstep);
    fdesc_steps(d.Osz, 3, d.Ostep);
    int off = 0, nvar = 0;
    for (int a = 0; a < A; ++a)
        for (int f = 0; f < 3; ++f) {
            FNode& nd = d.nodes[a * 3 + f];
            nd.off = off; nd.nmax = 3; nd.maxp[0] = 0; nd.maxp[1] = 1; nd.maxp[2] = 2;
            if (f < 2) { nd.out = N; nd.var = nvar++; nd.fixed_mask = 3; off += N * N * G * N; }
            else { nd.out = G; nd.var = -1; nd.fixed_mask = 7; off += N * N * G * G; }
        }
    for (int a = 0; a < A; ++a)
        for (int f = 0; f < 3; ++f) {
            FNode& nd = d.nodes[A * 3 + a * 3 + f];
            nd.off = off; nd.nmax = 1; nd.maxp[0] = (uint8_t)f; nd.var = -1; nd.fixed_mask = 1;
            nd.out = d.Osz[f];
            off += d.Ssz[f] * d.Osz[f];
        }
    d.ncounts = off;
    d.nvar    = nvar;
    c->prior.assign((size_t)off + nvar, 0.f);
    float* pr = c->prior.data();
    for (int a = 0; a < A; ++a) {
        for (int f = 0; f < 2; ++f)
            for (int v = 0; v < N; ++v)
                for (int x = 0; x < N; ++x) pr[d.nodes[A * 3 + a * 3 + f].off + v * N + x] = gw_obs_displ_prob_h(g, v, x) * static_total;
        for (int v = 0; v < G; ++v) pr[d.nodes[A * 3 + a * 3 + 2].off + v * G + v] = static_total;
        for (int x = 0; x < N; ++x)
            for (int y = 0; y < N; ++y) {
                int nx = x, ny = y;
                const float trans_prob = gw_slow_at_h(g, x, y) ? (float)(.15 + (double)noise) : (float).95;
                gw_move_h(g, a, nx, ny);
                float* rx = pr + d.nodes[a * 3 + 0].off + (x * N + y) * N;
                float* ry = pr + d.nodes[a * 3 + 1].off + (x * N + y) * N;
                rx[x] += (1 - trans_prob) * total;
                ry[y] += (1 - trans_prob) * total;
                rx[nx] += (trans_prob)*total;
                ry[ny] += (trans_prob)*total;
                for (int gl = 0; gl < G; ++gl) {
                    float* row = pr + d.nodes[a * 3 + 2].off + ((x * N + y) * G + gl) * G;
                    if (g.goal[gl][0] != x || g.goal[gl][1] != y) row[gl] = static_total;
                    else for (int ng = 0; ng < G; ++ng) row[ng] = static_total;
                }
            }
        const uint32_t m3 = 3u;
        std::memcpy(&pr[off + d.nodes[a * 3 + 0].var], &m3, 4);
        std::memcpy(&pr[off + d.nodes[a * 3 + 1].var], &m3, 4);
    }
    return FBA_OK;
}

// CollisionAvoidanceFactoredPrior ctor (reference
// src/domains/collision-avoidance/CollisionAvoidancePriors.cpp:210-347, obstacleTransition :385-404,
// observationDistr :46-63) for the fixed structures: "" / match-counts (every obstacle depends on
// itself) and fully-connected (every obstacle depends on all features).  Features
// {x, y, obstacle_1..n}; observation features = the n observed obstacle rows.
int build_ca_factored_prior(fba_ctx* c)
{
    Problem& P = c->P;
    const CADesc& ca = c->cadesc;
    const int A = P.A, W = ca.W, H = ca.H, n = ca.n, FS = 2 + n;
    const float noise = c->cfg.noise, total = c->cfg.counts_total;
    const bool full = c->cfg.structure_prior == FBA_SP_FULLY_CONNECTED;
    if (noise > .5 || noise < -.5) return fail(c, FBA_EINVAL, "CollisionAvoidanceFactoredPrior must be intiiated with -.5 < noise < .5 (is: %f)", noise);
    // edge noise "uniform" / "match-uniform" (:349-383): every obstacle node's parents are drawn per particle
    // (on the device, factored_prior_sample); the node owns room for every state feature as a parent
    // (the reinvigoration belief breeds particles with structures of their own: same layout)
    const bool noisy = c->cfg.structure_prior == FBA_SP_UNIFORM || c->cfg.structure_prior == FBA_SP_MATCH_UNIFORM ||
                       ((c->cfg.belief == FBA_BELIEF_REINVIGORATION || c->cfg.belief == FBA_BELIEF_INCUBATOR || c->cfg.belief == FBA_BELIEF_MH_GIBBS ||
                         c->cfg.belief == FBA_BELIEF_MH_NIPS) && !full);
    if (FS > MAXF || A * (FS + n) > MAXNODES) return fail(c, FBA_EINVAL, "too many state features");
    FDesc& d = c->fdesc;
    std::memset(&d, 0, sizeof d);
    d.FS = FS; d.FO = n;
    d.Ssz[0] = W; d.Ssz[1] = H;
    for (int f = 0; f < n; ++f) { d.Ssz[2 + f] = H; d.Osz[f] = H; }
    fdesc_steps(d.Ssz, FS, d.Sstep);
    fdesc_steps(d.Osz, n, d.Ostep);
    int off = 0;
    for (int a = 0; a < A; ++a)
        for (int f = 0; f < FS; ++f) {
            FNode& nd = d.nodes[a * FS + f];
            nd.off = off; nd.out = d.Ssz[f]; nd.var = -1;
            if (f >= 2 && (full || noisy)) {
                int rows = 1;
                nd.nmax = FS;
                for (int k = 0; k < FS; ++k) { nd.maxp[k] = (uint8_t)k; rows *= d.Ssz[k]; }
                nd.fixed_mask = (1u << FS) - 1u;
                if (noisy) nd.var = a * n + (f - 2);
                off += rows * H;
            } else {
                nd.nmax = 1; nd.maxp[0] = (uint8_t)f; nd.fixed_mask = 1;
                off += d.Ssz[f] * d.Ssz[f];
            }
        }
    for (int a = 0; a < A; ++a)
        for (int f = 0; f < n; ++f) {
            FNode& nd = d.nodes[A * FS + a * n + f];
            nd.off = off; nd.out = H; nd.var = -1; nd.nmax = 1; nd.maxp[0] = (uint8_t)(2 + f); nd.fixed_mask = 1;
            off += H * H;
        }
    d.ncounts = off;
    d.nvar    = noisy ? A * n : 0;
    c->prior.assign((size_t)off + d.nvar, 0.f);
    float* pr = c->prior.data();
    auto obstacle_transition = [&](int y, float* out) {
        const float move_prob = (float)(.25 - .5 * noise);
        const float stay_prob = (y == 0 || y == H - 1) ? (float)(3 * .25 + .5 * noise) : (float)(2 * .25 + noise);
        for (int k = 0; k < H; ++k) out[k] = 0;
        if (y != 0) out[y - 1] = move_prob * total;
        if (y != H - 1) out[y + 1] = move_prob * total;
        out[y] = stay_prob * total;
    };
    for (int a = 0; a < A; ++a) {
        for (int x = 1; x < W; ++x) pr[d.nodes[a * FS + 0].off + x * W + (x - 1)] = 1;
        for (int y = 0; y < H; ++y) pr[d.nodes[a * FS + 1].off + y * H + std::max(0, std::min(H - 1, y + a - 1))] += 1;
        for (int f = 2; f < FS; ++f) {
            const FNode& nd = d.nodes[a * FS + f];
            if (noisy) {  // the base record carries the correct graph {f}; every particle overwrites it
                for (int y = 0; y < H; ++y) obstacle_transition(y, pr + nd.off + y * H);
                const uint32_t own = 1u << f;
                std::memcpy(&pr[off + nd.var], &own, 4);
            } else if (!full) {
                for (int y = 0; y < H; ++y) obstacle_transition(y, pr + nd.off + y * H);
            } else {
                int rows = 1;
                for (int k = 0; k < FS; ++k) rows *= d.Ssz[k];
                for (int r = 0; r < rows; ++r) obstacle_transition((r / d.Sstep[f]) % d.Ssz[f], pr + nd.off + r * H);
            }
        }
        for (int f = 0; f < n; ++f)
            for (int y = 0; y < H; ++y) {
                float* row = pr + d.nodes[A * FS + a * n + f].off + y * H;
                row[0] = (float)normal_cdf(-y + .5);
                for (int oy = 1; oy < H - 1; ++oy) {
                    const int dist = std::abs(oy - y);
                    row[oy] = (float)(normal_cdf(dist + .5) - normal_cdf(dist - .5));
                }
                row[H - 1] = (float)normal_cdf(-(H - 1 - y) + .5);
                for (int oy = 0; oy < H; ++oy) row[oy] *= 10000;
            }
    }
    return FBA_OK;
}

int build_ftiger_factored_prior(fba_ctx* c);

// SysAdminFactoredPrior (SysAdminFactoredPrior.cpp:17-45, 129-257, 279-333).  One binary feature per
// computer; transition node (a, c): parents {c} (independent) or {c-1, c, c+1} (linear), counts
// {p, 1-p} * 10000 with p = computeFailureProbability; observation node (a): parent {a mod N}.
// "Structure noise is not enabled for the Sysadmin problem": nothing is drawn per particle.
int build_sysadmin_factored_prior(fba_ctx* c)
{
    Problem& P = c->P;
    const int A = P.A, N = c->sysdesc.N;
    const bool linear = P.domain == FBA_DOM_SYSADMIN_LINEAR;
    const bool reinvig = c->cfg.belief == FBA_BELIEF_REINVIGORATION || c->cfg.belief == FBA_BELIEF_INCUBATOR ||
                         c->cfg.belief == FBA_BELIEF_MH_GIBBS || c->cfg.belief == FBA_BELIEF_MH_NIPS;   // particles with structures of their own
    if (c->cfg.structure_prior != FBA_SP_NONE) return fail(c, FBA_EINVAL, "Structure noise is not enabled for the Sysadmin problem");
    if (N > MAXF || A * (N + 1) > MAXNODES) return fail(c, FBA_EINVAL, "too many state features");
    FDesc& d = c->fdesc;
    std::memset(&d, 0, sizeof d);
    d.FS = N; d.FO = 1;
    for (int f = 0; f < N; ++f) d.Ssz[f] = 2;
    d.Osz[0] = 2;
    fdesc_steps(d.Ssz, N, d.Sstep);
    fdesc_steps(d.Osz, 1, d.Ostep);
    int off = 0;
    for (int a = 0; a < A; ++a)
        for (int f = 0; f < N; ++f) {
            FNode& nd = d.nodes[a * N + f];
            nd.off = off; nd.out = 2; nd.var = -1; nd.nmax = 0;
            if (reinvig) {  // bred particles choose any parent set: room for all N, the prior's set as the base mask
                uint32_t m = 1u << f;
                if (linear && f > 0) m |= 1u << (f - 1);
                if (linear && f < N - 1) m |= 1u << (f + 1);
                for (int k = 0; k < N; ++k) nd.maxp[k] = (uint8_t)k;
                nd.nmax = N; nd.var = a * N + f; nd.fixed_mask = m;
                off += 2 << N;
                continue;
            }
            if (linear && f > 0) nd.maxp[nd.nmax++] = (uint8_t)(f - 1);
            nd.maxp[nd.nmax++] = (uint8_t)f;
            if (linear && f < N - 1) nd.maxp[nd.nmax++] = (uint8_t)(f + 1);
            nd.fixed_mask = (1u << nd.nmax) - 1u;
            off += 2 << nd.nmax;
        }
    for (int a = 0; a < A; ++a) {
        FNode& nd = d.nodes[A * N + a];
        nd.off = off; nd.out = 2; nd.var = -1; nd.nmax = 1; nd.maxp[0] = (uint8_t)(a % N); nd.fixed_mask = 1;
        off += 4;
    }
    d.ncounts = off;
    d.nvar    = reinvig ? A * N : 0;
    c->prior.assign((size_t)off + d.nvar, 0.f);
    float* pr = c->prior.data();
    for (int a = 0; a < A; ++a)
        for (int f = 0; f < N; ++f) {
            const FNode& nd = d.nodes[a * N + f];
            const bool rebooting = a == N + f;
            int parents[MAXF], np = 0;
            for (int k = 0; k < nd.nmax; ++k)
                if ((nd.fixed_mask >> k) & 1u) parents[np++] = nd.maxp[k];
            if (reinvig) std::memcpy(&pr[off + nd.var], &nd.fixed_mask, 4);
            for (int r = 0; r < (1 << np); ++r) {  // row r: parent values, last parent the fastest digit
                int failing = 0;
                bool own_up = true;
                for (int k = 0; k < np; ++k) {
                    const int v = (r >> (np - 1 - k)) & 1;
                    if (parents[k] == f) own_up = v != 0;
                    else if (!v) failing++;              // parents other than f are its linear neighbours
                }
                float p;
                if (!own_up) p = rebooting ? 1 - SYS_REBOOT : 1;   // :295-298
                else {
                    double fail = 1 - c->sysdesc.keep[failing];          // :313-315
                    if (rebooting) fail *= (1 - SYS_REBOOT);
                    p = (float)fail;
                }
                pr[nd.off + 2 * r + 0] = p * 10000.f;
                pr[nd.off + 2 * r + 1] = (1 - p) * 10000.f;
            }
        }
    for (int a = 0; a < A; ++a) {  // precomputeFactoredPrior :148-183: output 0 = FAILING
        float* o = pr + d.nodes[A * N + a].off;
        o[0] = 10000.f * SYS_OBSERVE; o[1] = 10000.f * (1 - SYS_OBSERVE);   // operated computer failing
        o[2] = 10000.f * (1 - SYS_OBSERVE); o[3] = 10000.f * SYS_OBSERVE;   // operated computer working
    }
    return FBA_OK;
}

int build_factored_prior(fba_ctx* c)
{
    if (is_sys(c->P.domain)) return build_sysadmin_factored_prior(c);
    if (is_ca(c->P.domain)) return build_ca_factored_prior(c);
    if (c->P.domain == FBA_DOM_GRIDWORLD) return build_gridworld_factored_prior(c);
    return build_ftiger_factored_prior(c);
}

int build_ftiger_factored_prior(fba_ctx* c)
{
    Problem& P = c->P;
    if (!is_ftiger(P.domain)) return fail(c, FBA_EINVAL, "domain %d has no built-in factored prior", P.domain);
    const float noise = c->cfg.noise, total = c->cfg.counts_total;
    if (noise <= -.15 || noise > .3) return fail(c, FBA_EINVAL, "noise must be between -.15 and .3");
    FDesc& d = c->fdesc;
    std::memset(&d, 0, sizeof d);
    d.FS = c->cfg.size + 1;
    d.FO = 1;
    if (d.FS > MAXF) return fail(c, FBA_EINVAL, "factored tiger supports at most %d irrelevant features", MAXF - 1);
    if (P.A * (d.FS + d.FO) > MAXNODES) return fail(c, FBA_EINVAL, "too many DBN nodes");
    for (int f = 0; f < d.FS; ++f) d.Ssz[f] = 2;
    d.Osz[0] = 2;
    fdesc_steps(d.Ssz, d.FS, d.Sstep);
    fdesc_steps(d.Osz, d.FO, d.Ostep);
    int off = 0;
    for (int a = 0; a < P.A; ++a)
        for (int f = 0; f < d.FS; ++f) {
            FNode& nd = d.nodes[a * d.FS + f];
            nd.off = off; nd.out = 2; nd.var = -1;
            if (a == 2) { nd.nmax = 1; nd.maxp[0] = (uint8_t)f; nd.fixed_mask = 1; off += 4; }
            else { nd.nmax = 0; nd.fixed_mask = 0; off += 2; }
        }
    for (int a = 0; a < P.A; ++a) {
        FNode& nd = d.nodes[P.A * d.FS + a * d.FO];
        nd.off = off; nd.out = 2; nd.var = -1;
        if (a == 2) {
            nd.nmax = d.FS; nd.var = 0;
            for (int f = 0; f < d.FS; ++f) nd.maxp[f] = (uint8_t)f;
            off += 2 << d.FS;
        } else { nd.nmax = 0; off += 2; }
    }
    d.ncounts = off;
    d.nvar    = 1;
    c->prior.assign((size_t)off + d.nvar, 0.f);
    for (int a = 0; a < P.A; ++a)
        for (int f = 0; f < d.FS; ++f) {
            const FNode& nd = d.nodes[a * d.FS + f];
            if (a == 2) { c->prior[nd.off + 0] = 5000; c->prior[nd.off + 3] = 5000; }
            else { c->prior[nd.off] = 5000; c->prior[nd.off + 1] = 5000; }
        }
    for (int a = 0; a < 2; ++a) {
        const FNode& nd = d.nodes[P.A * d.FS + a * d.FO];
        c->prior[nd.off] = 5000; c->prior[nd.off + 1] = 5000;
    }
    {   // correct structure {0}: rows (LEFT, RIGHT) = (acc, inacc), (inacc, acc)
        const FNode& nd = d.nodes[P.A * d.FS + 2 * d.FO];
        const float acc = (.85f - noise) * total, inacc = (.15f + noise) * total;
        c->prior[nd.off + 0] = acc; c->prior[nd.off + 1] = inacc;
        c->prior[nd.off + 2] = inacc; c->prior[nd.off + 3] = acc;
        const uint32_t mask = 1u;
        std::memcpy(&c->prior[off + 0], &mask, 4);
    }
    return FBA_OK;
}

// PackedFtigerView::prior on the host (fba_device.h): the prior count of cell k of the factored-tiger model under the
// listen observation node's parent set `mask`
float ftiger_prior_host(int FS, int k, uint32_t mask, float acc, float inacc, float unif)
{
    if (k < 4 * FS) return 5000.f;
    if (k < 8 * FS) { const int j = k - 4 * FS; return (((j >> 1) ^ j) & 1) ? 0.f : 5000.f; }
    if (k < 8 * FS + 4) return 5000.f;
    const int j = k - (8 * FS + 4), row = j >> 1, np = __builtin_popcount(mask);
    if (row >= (1 << np)) return 0.f;
    if (!(mask & 1u)) return unif;
    return ((row >> (np - 1)) == (j & 1)) ? acc : inacc;
}

// prior[k] + j is the float the reference reaches by j additions of 1.0f for every j a uint16 holds
bool packable_prior(const std::vector<float>& prior)
{
    for (float p : prior) {
        const double top = (double)p + 65535.0;
        if (!(p >= 0.f) || (double)(float)top != top) return false;
    }
    return true;
}

// History particles (fba_device.h): GridWorldFactBAPrior::setNoisyTransitionNode (GridWorldBAPriors.cpp:255-295) for
// every action and the x / y node -- what gw_fill_xy_node_with_goal writes into a dense record on the device.
void build_gridworld_alt_prior(fba_ctx* c)
{
    const GridDesc& g = c->gdesc;
    const int N = g.N, G = g.G, A = c->P.A, XY = N * N * G * N;
    const float noise = c->cfg.noise, total = c->cfg.counts_total;
    c->prior_alt.assign((size_t)A * 2 * XY, 0.f);
    for (int a = 0; a < A; ++a)
        for (int f = 0; f < 2; ++f) {
            float* base = c->prior_alt.data() + (size_t)(a * 2 + f) * XY;
            for (int x = 0; x < N; ++x)
                for (int y = 0; y < N; ++y) {
                    int nx = x, ny = y;
                    const float trans_prob = gw_slow_at_h(g, x, y) ? (float)(.15 + (double)noise) : (float).95;
                    gw_move_h(g, a, nx, ny);
                    const int loc = f == 0 ? x : y, new_loc = f == 0 ? nx : ny;
                    for (int gl = 0; gl < G; ++gl) {
                        float* row = base + ((x * N + y) * G + gl) * N;
                        row[loc] += (1 - trans_prob) * total;
                        row[new_loc] += (trans_prob)*total;
                    }
                }
        }
}
// v + (float)j is the float that j additions of 1.0f to v reach, for every j up to `most`
bool increments_exact(const std::vector<float>& table, int most)
{
    for (float v : table) {
        if (!(v >= 0.f)) return false;
        float seq = v;
        for (int j = 1; j <= most; ++j) {
            seq += 1.0f;
            if (seq != v + (float)j) return false;
        }
    }
    return true;
}

int upload_prior(fba_ctx* c)
{
    std::vector<float> padded((size_t)c->P.Cs, 0.f);
    if (c->P.ft_packed) {  // the prior record in packed form: no increments, the correct structure's bits
        const uint32_t m = 1u;
        std::memcpy(&padded[(size_t)c->fdesc.ncounts / 2], &m, 4);
    } else if (c->P.hist) {  // records carry no counts: the tables sit beside them, rows padded to 16 bytes (HistLayout, fba_device.h)
        const HistLayout L(c->gdesc.N, c->gdesc.G, c->P.A);
        const int N = L.N, G = L.G, A = L.A, XYd = N * N * G * N, GGd = N * N * G * G, NNd = N * N;
        std::vector<float> base((size_t)L.total + 16, 0.f), alt((size_t)L.alt_total + 16, 0.f);
        const float* pr = c->prior.data();
        for (int a = 0; a < A; ++a) {
            const int td = a * (2 * XYd + GGd), od = A * (2 * XYd + GGd) + a * (2 * NNd + G * G);
            for (int f = 0; f < 2; ++f)
                for (int row = 0; row < N * N * G; ++row)
                    for (int i = 0; i < N; ++i) {
                        if (row < N * N) base[(size_t)a * L.tstride + f * L.XY + row * L.NS + i] = pr[td + f * XYd + row * N + i];
                        alt[(size_t)(a * 2 + f) * L.XY + row * L.NS + i] = c->prior_alt[(size_t)(a * 2 + f) * XYd + row * N + i];
                    }
            for (int row = 0; row < N * N * G; ++row)
                for (int i = 0; i < G; ++i) base[(size_t)a * L.tstride + 2 * L.XY + row * L.GS + i] = pr[td + 2 * XYd + row * G + i];
            for (int f = 0; f < 3; ++f) {
                const int n = f == 2 ? G : N;
                for (int v = 0; v < n; ++v)
                    for (int i = 0; i < n; ++i) base[(size_t)L.o_row(a, f, v) + i] = pr[od + (f == 2 ? 2 * NNd : f * NNd) + v * n + i];
            }
        }
        HIPCHK(c, hipMemcpyAsync(c->d_hist_base, base.data(), base.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->d_hist_alt, alt.data(), alt.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
        {
            // the same rows deduplicated (Problem::hist_lds): a row slot -> one byte, the distinct rows K floats each
            const HistRowIds I(N, G, A);
            const int K = c->P.hist_row <= 8 ? 8 : (c->P.hist_row <= 10 ? 12 : 16);   // (a 12-float row is walked to 10: HistRow::KL)
            const int rid_bytes = (I.total + 15) & ~15;
            std::vector<uint8_t> blob((size_t)rid_bytes, 0);
            std::vector<float> rows;
            std::map<std::vector<uint32_t>, int> ids;
            bool fits = true;
            auto put = [&](int slot, const float* src, int n) {
                std::vector<uint32_t> key((size_t)K, 0u);
                for (int i = 0; i < n; ++i) std::memcpy(&key[i], &src[i], 4);
                auto it = ids.find(key);
                if (it == ids.end()) {
                    if (ids.size() >= 256) { fits = false; return; }
                    it = ids.emplace(key, (int)ids.size()).first;
                    for (int i = 0; i < K; ++i) { float f; std::memcpy(&f, &key[i], 4); rows.push_back(f); }
                }
                blob[(size_t)slot] = (uint8_t)it->second;
            };
            for (int a = 0; a < A && fits; ++a) {
                for (int f = 0; f < 2; ++f)
                    for (int cell = 0; cell < N * N; ++cell) put(I.t(a, f, false, cell, 0), &base[(size_t)L.t_row(a, f, false, cell, 0)], N);
                for (int cell = 0; cell < N * N; ++cell)
                    for (int gl = 0; gl < G; ++gl) {
                        put(I.t(a, 2, true, cell, gl), &base[(size_t)L.t_row(a, 2, true, cell, gl)], G);
                        for (int f = 0; f < 2; ++f) put(I.t(a, f, true, cell, gl), &alt[(size_t)L.alt_row(a, f, cell, gl)], N);
                    }
                for (int f = 0; f < 3; ++f)
                    for (int v = 0; v < (f == 2 ? G : N); ++v) put(I.o(a, f, v), &base[(size_t)L.o_row(a, f, v)], f == 2 ? G : N);
            }
            if (fits) {
                const size_t nb = blob.size();
                blob.resize(nb + rows.size() * sizeof(float));
                std::memcpy(blob.data() + nb, rows.data(), rows.size() * sizeof(float));
                HIPCHK(c, hipMemcpyAsync(c->d_hist_lds, blob.data(), blob.size(), hipMemcpyHostToDevice, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));   // (blob is a local)
                c->P.hist_lds       = c->d_hist_lds;
                c->P.hist_rid_bytes = rid_bytes;
                c->P.hist_distinct  = (int)ids.size();
            } else {
                c->P.hist_lds = nullptr; c->P.hist_rid_bytes = 0; c->P.hist_distinct = 0;
            }
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
    } else if (c->P.packed) {  // records start as "no increments yet"; the table itself goes beside them
        if (!packable_prior(c->prior))
            return fail(c, FBA_EINVAL, "this context stores particles packed (uint16 increments over the prior), which needs prior counts c with "
                                       "c + 65535 exact in fp32; create it with FBA_DENSE_PARTICLES=1 in the environment for other tables");
        HIPCHK(c, hipMemcpyAsync(c->d_prior_dense, c->prior.data(), c->prior.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    } else if (!c->P.ft_packed)
        std::copy(c->prior.begin(), c->prior.end(), padded.begin());
    HIPCHK(c, hipMemcpyAsync(c->d_prior, padded.data(), padded.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FBA_OK;
}

// ---- timed launches ---------------------------------------------------------------------------
int flush_events(fba_ctx* c)
{
    if (c->pending.empty()) return FBA_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (auto& p : c->pending) {
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, p.a, p.b));
        c->k_ms[p.kind] += ms;
        c->k_launches[p.kind] += 1;
        c->free_events.push_back(p);
    }
    c->pending.clear();
    return FBA_OK;
}

template <typename F>
int timed(fba_ctx* c, int kind, F&& launch)
{
    if (!c->timing) {
        launch();
        if (const hipError_t le = launch_take_error(); le != hipSuccess) return fail(c, FBA_EHIP, "inside a kernel launch sequence: %s", hipGetErrorString(le));
        return FBA_OK;
    }
    if (c->pending.size() >= 2048) {
        int rc = flush_events(c);
        if (rc) return rc;
    }
    EventPair p;
    if (!c->free_events.empty()) {
        p = c->free_events.back();
        c->free_events.pop_back();
    } else {
        HIPCHK(c, hipEventCreate(&p.a));
        HIPCHK(c, hipEventCreate(&p.b));
    }
    p.kind = kind;
    HIPCHK(c, hipEventRecord(p.a, c->stream));
    launch();
    HIPCHK(c, hipEventRecord(p.b, c->stream));
    c->pending.push_back(p);
    if (const hipError_t le = launch_take_error(); le != hipSuccess) return fail(c, FBA_EHIP, "inside a kernel launch sequence: %s", hipGetErrorString(le));
    return FBA_OK;
}

int k_update_kind(const fba_ctx* c) { return c->P.belief == FBA_BELIEF_REJECTION ? FBA_K_BELIEF_RS : FBA_K_BELIEF_IS; }

// one real time-step of every active slot
int tick(fba_ctx* c)
{
    int rc;
    if ((rc = timed(c, FBA_K_SEARCH, [&] { launch_search(c->P, c->D, c->stream); }))) return rc;
    if ((rc = timed(c, FBA_K_ENV, [&] { launch_env(c->P, c->D, c->d_n_active, c->stream); }))) return rc;
    if ((rc = timed(c, k_update_kind(c), [&] { launch_belief_update(c->P, c->D, c->stream); }))) return rc;
    if (c->D.trace_on) launch_flush(c->P, c->D, c->stream);
    launch_advance(c->P, c->D, c->d_n_active, c->stream);
    if ((rc = timed(c, FBA_K_BELIEF_INIT, [&] { launch_init(c->P, c->D, c->stream); }))) return rc;
    if (c->P.model != FBA_MODEL_POMDP)
        if ((rc = timed(c, FBA_K_BELIEF_RESET, [&] { launch_reset(c->P, c->D, c->stream); }))) return rc;
    HIPCHK(c, hipGetLastError());
    return FBA_OK;
}

// after a synchronisation point: did a rejection update give up (fba_kernels.h REJECT_MAX_ATTEMPTS)?
int check_fault(fba_ctx* c)
{
    int32_t f = 0;
    HIPCHK(c, hipMemcpy(&f, c->D.fault, sizeof f, hipMemcpyDeviceToHost));
    if (!f) return FBA_OK;
    HIPCHK(c, hipMemset(c->D.fault, 0, sizeof f));
    if (f >= 0x20000000 && f < 0x40000000)
        return fail(c, FBA_ESTATE, "%s in slot %d: the sampled model cannot reproduce the run's history (the reference would never "
                    "return from %s)", c->P.mh == 3 ? "mh-nips" : "mh-within-gibbs", f - 0x20000000,
                    c->P.mh == 3 ? "computePosterior, MHNIPS2018.cpp:39-105" : "rejectionSampleStateHistory, MHwithinGibbs.cpp:38-94");
    if (f >= 0x40000000)
        return fail(c, FBA_ESTATE, "slot %d: more belief updates in one run than the %d (episodes * horizon) a history particle was "
                    "sized for; create the context with FBA_DENSE_PARTICLES=1 in the environment to drive it beyond that", f - 0x40000000, c->P.hist_cap);
    if (f < 0 && c->D.bkt)
        return fail(c, FBA_ESTATE, "the search tree of slot %d outgrew its table of %d buckets (fba_config.tree_buckets; 0 = 2 * (sims + 2), which no search can fill)",
                    -f - 1, 2 * c->D.bkt_lines);
    if (f < 0) return fail(c, FBA_ESTATE, "the search tree of slot %d needed more than the %d node records it has", -f - 1, c->D.max_nodes);
    return fail(c, FBA_ESTATE, "rejection sampling in slot %d accepted fewer than %d particles in %d attempts: no particle of the filter "
                "can produce the observation (the reference loops forever in RejectionSampling.hpp:26-72 here)", f - 1, c->P.N, REJECT_MAX_ATTEMPTS);
}

int set_flags(fba_ctx* c, uint8_t* dev, const uint8_t* mask, uint8_t value)
{
    std::vector<uint8_t> h((size_t)c->P.E, value);
    if (mask)
        for (int e = 0; e < c->P.E; ++e) h[e] = mask[e] ? value : 0;
    HIPCHK(c, hipMemcpyAsync(dev, h.data(), h.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FBA_OK;
}

int ensure_outputs(fba_ctx* c, int runs)
{
    const size_t need = (size_t)std::max(runs, 1) * c->P.episodes;
    if (need > c->returns_cap) {
        int rc;
        if ((rc = dev_alloc(c, &c->D.returns, need))) return rc;
        if ((rc = dev_alloc(c, &c->D.lengths, need))) return rc;
        c->returns_cap = need;
    } else {
        HIPCHK(c, hipMemsetAsync(c->D.returns, 0, need * sizeof(double), c->stream));
        HIPCHK(c, hipMemsetAsync(c->D.lengths, 0, need * sizeof(int32_t), c->stream));
    }
    if (c->cfg.trace) return ensure_trace(c, (size_t)std::max(runs, 1) * c->P.episodes * c->P.horizon);
    return FBA_OK;
}

// Budgeted searches park their state per slot (s_sim / s_nodes / s_depth, fba_state.h).  Whatever re-positions the slots -- a new
// experiment, fba_belief_init, fba_set_position -- makes a parked search stale: the next launch must start a search, not resume one.
int clear_parked_searches(fba_ctx* c)
{
    if (!c->D.s_sim) return FBA_OK;
    const size_t E = (size_t)c->P.E;
    HIPCHK(c, hipMemsetAsync(c->D.s_sim, 0, E * sizeof *c->D.s_sim, c->stream));
    HIPCHK(c, hipMemsetAsync(c->D.s_nodes, 0, E * sizeof *c->D.s_nodes, c->stream));
    HIPCHK(c, hipMemsetAsync(c->D.s_depth, 0, E * sizeof *c->D.s_depth, c->stream));
    HIPCHK(c, hipMemsetAsync(c->D.search_done, 0, E * sizeof *c->D.search_done, c->stream));
    return FBA_OK;
}

int start_experiment(fba_ctx* c, int runs_total)
{
    int rc;
    if ((rc = clear_parked_searches(c))) return rc;
    c->D.runs_total = runs_total;
    c->D.run_offset = c->cfg.run_offset;
    const int32_t n_active = runs_total < 0 ? c->P.E : std::min(c->P.E, runs_total);
    HIPCHK(c, hipMemcpyAsync(c->d_n_active, &n_active, sizeof n_active, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->D.bufsel, 0, (size_t)c->P.E, c->stream));
    HIPCHK(c, hipMemsetAsync(c->D.lazy_reset, 0, (size_t)c->P.E, c->stream));
    if (c->D.bufsel_fc) HIPCHK(c, hipMemsetAsync(c->D.bufsel_fc, 0, (size_t)c->P.E, c->stream));
    if (c->D.bufsel_sh) HIPCHK(c, hipMemsetAsync(c->D.bufsel_sh, 0, (size_t)c->P.E, c->stream));
    launch_start(c->P, c->D, c->stream);
    if ((rc = timed(c, FBA_K_BELIEF_INIT, [&] { launch_init(c->P, c->D, c->stream); }))) return rc;
    if (c->P.model != FBA_MODEL_POMDP)
        if ((rc = timed(c, FBA_K_BELIEF_RESET, [&] { launch_reset(c->P, c->D, c->stream); }))) return rc;
    HIPCHK(c, hipGetLastError());
    c->belief_ready = true;
    return FBA_OK;
}

int run_experiment(fba_ctx* c, fba_stat* stats)
{
    int rc;
    const int runs = c->cfg.runs, E = c->P.E, eps = c->P.episodes;
    if ((rc = ensure_outputs(c, runs))) return rc;
    if ((rc = start_experiment(c, runs))) return rc;
    long long max_ticks = (long long)((runs + E - 1) / E) * eps * c->P.horizon + 2;
    if (c->P.search_budget > 0)   // budgeted launches: a real step takes as many launches as its search needs (<= horizon + 1 iterations per simulation)
        max_ticks *= ((long long)c->P.sims * (c->P.horizon + 1)) / c->P.search_budget + 2;
    for (long long k = 0; k < max_ticks; ++k) {
        if ((rc = tick(c))) return rc;
        int32_t n_active = 0;
        HIPCHK(c, hipMemcpyAsync(&n_active, c->d_n_active, sizeof n_active, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if ((rc = check_fault(c))) return rc;
        if (n_active <= 0) break;
    }
    std::vector<double> ret((size_t)runs * eps);
    HIPCHK(c, hipMemcpy(ret.data(), c->D.returns, ret.size() * sizeof(double), hipMemcpyDeviceToHost));
    std::memset(stats, 0, sizeof(fba_stat) * (size_t)eps);
    for (int r = 0; r < runs; ++r)
        for (int ep = 0; ep < eps; ++ep) fba_stat_add(&stats[ep], ret[(size_t)r * eps + ep]);
    c->started = false;
    return FBA_OK;
}

template <typename T>
uint64_t sum_counter(fba_ctx* c, const T* dev, int n, int* rc)
{
    std::vector<unsigned long long> h((size_t)n);
    hipError_t e = hipMemcpy(h.data(), dev, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        *rc = fail(c, FBA_EHIP, "counter download failed: %s", hipGetErrorString(e));
        return 0;
    }
    uint64_t s = 0;
    for (auto v : h) s += v;
    return s;
}

}  // namespace

// =============================================================================================
// C-ABI
// =============================================================================================
extern "C" {

int fba_abi_version(void) { return FBA_ABI_VERSION; }

void fba_default_config(fba_config* cfg)
{
    std::memset(cfg, 0, sizeof *cfg);
    cfg->domain       = FBA_DOM_TIGER_EPISODIC;
    cfg->model        = FBA_MODEL_POMDP;
    cfg->belief       = FBA_BELIEF_REJECTION;
    cfg->planner      = FBA_PLANNER_POUCT;
    cfg->particles    = 100;
    cfg->sims         = 1000;
    cfg->max_depth    = -1;
    cfg->horizon      = 10;
    cfg->exploration  = 100;
    cfg->discount     = .95;
    cfg->runs         = 1;
    cfg->episodes     = 1;
    cfg->noise        = 0;
    cfg->counts_total = 10000;
    cfg->slots        = 0;
}

const char* fba_last_error(const fba_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

void fba_destroy(fba_ctx* c)
{
    if (!c) return;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& p : c->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto& p : c->free_events) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (void* p : c->allocs) (void)hipFree(p);
    if (c->D.list_count_host) (void)hipHostFree(c->D.list_count_host);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int fba_create(const fba_config* cfg, fba_ctx** out)
{
    if (!cfg || !out) return fail(nullptr, FBA_EINVAL, "fba_create: null argument");
    *out = nullptr;
    (void)hipGetLastError();  // a new context starts from a clean error state
    // ---- validation: the reference's constructor / validate() errors
    // (POUCT.cpp:34-50, RejectionSampling.cpp:7-13, FactoredTiger.cpp:13-17, TigerPriors.cpp:22-25)
    if (cfg->sims < 1 && cfg->planner == FBA_PLANNER_POUCT)
        return fail(nullptr, FBA_EINVAL, "cannot initiate POUCT with %d simulations, must be greater than 0", cfg->sims);
    if (cfg->horizon <= 0) return fail(nullptr, FBA_EINVAL, "cannot initiate POUCT with %d horizon, must be greater than 0", cfg->horizon);
    if (cfg->horizon > 255) return fail(nullptr, FBA_EINVAL, "horizon %d exceeds the 255 steps a stream address holds", cfg->horizon);
    if (cfg->particles < 1) return fail(nullptr, FBA_EINVAL, "cannot initiate belief with n = %d", cfg->particles);
    if (cfg->discount <= 0 || cfg->discount > 1) return fail(nullptr, FBA_EINVAL, "discount must be in (0, 1]");
    if (cfg->runs < 1 || cfg->episodes < 1) return fail(nullptr, FBA_EINVAL, "runs and episodes must be >= 1");
    if (cfg->episodes > 65535) return fail(nullptr, FBA_EINVAL, "episodes %d exceeds the 65535 a stream address holds", cfg->episodes);
    if (cfg->model == FBA_MODEL_POMDP && cfg->episodes != 1) return fail(nullptr, FBA_EINVAL, "planning runs have exactly one episode per run");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, FBA_ENODEVICE, "no HIP device visible: libfba_hip has no CPU path");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, FBA_EINVAL, "device %d out of range (%d visible)", cfg->device, ndev);

    fba_config eff = *cfg;
    const bool point = eff.belief == FBA_BELIEF_POINT;
    if (point) {  // PointEstimation.cpp:36-76 / BAPointEstimation.cpp: the rejection update of a single state
        eff.belief    = FBA_BELIEF_REJECTION;
        eff.particles = 1;
    }
    cfg = &eff;
    fba_ctx* c = new fba_ctx();
    c->cfg     = *cfg;
    Problem& P = c->P;
    P.domain = cfg->domain; P.model = cfg->model; P.belief = cfg->belief; P.planner = cfg->planner;
    P.reinvig = 0;
    P.cheat   = 0;
    P.point   = point ? 1 : 0;
    if (cfg->belief == FBA_BELIEF_CHEATING) {
        // a weighted (importance) filter + a rejection filter of correct-graph particles it copies from when
        // its log likelihood drops below --threshold (prototypes/CheatingReinvigoration.cpp)
        if (cfg->model != FBA_MODEL_BA_FACTORED) {
            fail(nullptr, FBA_EINVAL, "cheating-reinvigoration belief: needs a factored model (fbapomdp)");
            delete c;
            return FBA_EINVAL;
        }
        if (cfg->particles < 1 || cfg->resample_amount < 1) {  // CheatingReinvigoration.cpp:30-34
            fail(nullptr, FBA_EINVAL, "CheatingReinvigoration::cannot initiate belief of size < 1 (%d), or resample size of < 1 (%d)",
                 cfg->particles, cfg->resample_amount);
            delete c;
            return FBA_EINVAL;
        }
        if (cfg->threshold >= 0) {  // :36-40
            fail(nullptr, FBA_EINVAL, "CheatingReinvigoration::cannot initiate with resample_threshold >= 0 (is:%f)", cfg->threshold);
            delete c;
            return FBA_EINVAL;
        }
        P.belief = FBA_BELIEF_IMPORTANCE;
        P.cheat  = cfg->resample_amount;
    }
    P.mh = 0;
    if (cfg->belief == FBA_BELIEF_MH_GIBBS || cfg->belief == FBA_BELIEF_MH_NIPS) {
        // an importance filter + the run's (a, o) history + a Metropolis-Hastings re-draw of the filter when the log
        // likelihood drops below --threshold (MHwithinGibbs.cpp, MHNIPS2018.cpp); built for the priors that have
        // computePriorModel and mutate and a fully enumerable state space: factored tiger, collision avoidance
        const bool nips  = cfg->belief == FBA_BELIEF_MH_NIPS;
        const char* name = nips ? "MHNIPS2018" : "MHwithinGibbs";
        if (cfg->model != FBA_MODEL_BA_FACTORED || cfg->dirichlet_regular ||
            !(is_ftiger(cfg->domain) || is_ca(cfg->domain) || cfg->domain == FBA_DOM_GRIDWORLD || is_sys(cfg->domain)) ||
            (is_ca(cfg->domain) && cfg->structure_prior == FBA_SP_FULLY_CONNECTED)) {
            fail(nullptr, FBA_EINVAL, "%s belief: needs a factored model (fbapomdp) in the expected Dirichlet mode, at most %d structure words per particle",
                 nips ? "mh-nips" : "mh-within-gibbs", MH_MAXVAR);
            delete c;
            return FBA_EINVAL;
        }
        if (nips && is_sys(cfg->domain)) {
            fail(nullptr, FBA_EINVAL, "mh-nips belief on sysadmin: MHNIPS2018::MH scores a particle's counts (grown from the domain's prior, 10000 per row) "
                 "against computePriorModel(structure) (rows {total * p, total - p}, SysAdminFactoredPrior.cpp:98-127): a different prior, "
                 "whose score difference admits no proposal -- the reference does not return from it");
            delete c;
            return FBA_EINVAL;
        }
        if (cfg->particles < 1) {  // MHwithinGibbs.cpp:243-246, MHNIPS2018.cpp:116-119
            fail(nullptr, FBA_EINVAL, "%s::cannot initiate MH with size 0", name);
            delete c;
            return FBA_EINVAL;
        }
        if (cfg->threshold >= 0) {  // :248-252, MHNIPS2018.cpp:121-126
            fail(nullptr, FBA_EINVAL, "%s::cannot initiate with threshold >= 0 (is:%f)", name, cfg->threshold);
            delete c;
            return FBA_EINVAL;
        }
        P.belief = FBA_BELIEF_IMPORTANCE;
        P.mh     = nips ? 3 : cfg->belief_option == 1 ? 2 : 1;
    }
    P.incub = 0;
    if (cfg->belief == FBA_BELIEF_INCUBATOR) {
        // StructureIncubatorSampling(size, reinvigor_amount, threshold) (BABelief.cpp:53-58): the reinvigoration belief's two
        // rejection filters + a weighted shadow filter of bred particles; same domains (a fully connected prior is needed)
        const bool ftiger = cfg->domain == FBA_DOM_FTIGER_EPISODIC || cfg->domain == FBA_DOM_FTIGER_CONTINUOUS;
        if (cfg->model != FBA_MODEL_BA_FACTORED || !(ftiger || is_ca(cfg->domain) || is_sys(cfg->domain)) ||
            (is_ca(cfg->domain) && cfg->structure_prior == FBA_SP_FULLY_CONNECTED)) {
            fail(nullptr, FBA_EINVAL, "incubator belief: needs a factored model (fbapomdp) of factored tiger, collision avoidance or sysadmin");
            delete c;
            return FBA_EINVAL;
        }
        if (cfg->particles < 1 || cfg->resample_amount < 1) {  // StructureIncubatorSampling.cpp:28-33
            fail(nullptr, FBA_EINVAL, "StructureIncubatorSampling::Cannot initiate Incubator belief update with size < 1 (%d) or resample size < 1 (%d)",
                 cfg->particles, cfg->resample_amount);
            delete c;
            return FBA_EINVAL;
        }
        if (cfg->threshold <= 0 || cfg->threshold > 1) {  // :35-39
            fail(nullptr, FBA_EINVAL, "StructureIncubatorSampling::must initiate with 1 < threshold <= 0 (is:%f)", cfg->threshold);
            delete c;
            return FBA_EINVAL;
        }
        if (cfg->resample_amount >= cfg->particles || cfg->resample_amount > 1024) {  // WeightedFilter::leastLikely asserts n < size() (WeightedFilter.cpp:207)
            fail(nullptr, FBA_EINVAL, "incubator belief: the resample amount (%d) must be below the number of particles (%d): WeightedFilter::leastLikely",
                 cfg->resample_amount, cfg->particles);
            delete c;
            return FBA_EINVAL;
        }
        if (cfg->particles > IS_LDS_MAX_N) {
            fail(nullptr, FBA_EINVAL, "incubator belief: at most %d particles per slot", IS_LDS_MAX_N);
            delete c;
            return FBA_EINVAL;
        }
        P.belief = FBA_BELIEF_REJECTION;
        P.incub  = cfg->resample_amount;
    }
    P.nested = 0;
    if (cfg->belief == FBA_BELIEF_NESTED) {
        // NestedBelief(particle_amount, particle_amount^2) (BABelief.cpp:67-70): a weighted filter of count particles, each
        // with its own flat filter of domain states; the BABelief factory only, i.e. bapomdp / fbapomdp
        if (cfg->model == FBA_MODEL_POMDP) {
            fail(nullptr, FBA_EINVAL, "nested belief: needs a Bayes-adaptive model (bapomdp / fbapomdp)");
            delete c;
            return FBA_EINVAL;
        }
        if (cfg->particles < 1) {  // NestedBelief.cpp:22-27
            fail(nullptr, FBA_EINVAL, "NestedBelief: cannot initiate with filter size < 1 (top: %d, bottom: %d)", cfg->particles, cfg->particles * cfg->particles);
            delete c;
            return FBA_EINVAL;
        }
        if (cfg->particles > 256) {
            fail(nullptr, FBA_EINVAL, "nested belief: at most 256 count particles (each carries particles^2 domain states)");
            delete c;
            return FBA_EINVAL;
        }
        P.belief = FBA_BELIEF_IMPORTANCE;
        P.nested = cfg->particles * cfg->particles;
    }
    if (cfg->belief == FBA_BELIEF_REINVIGORATION) {
        // two rejection filters + breeding (ReinvigoratingRejectionSampling.hpp); the reference has fully
        // connected priors for factored tiger, collision avoidance and sysadmin
        // (GridWorldFactBAPrior::sampleFullyConnectedState throws "nyi"): all three are built
        const bool ftiger = cfg->domain == FBA_DOM_FTIGER_EPISODIC || cfg->domain == FBA_DOM_FTIGER_CONTINUOUS;
        if (cfg->model != FBA_MODEL_BA_FACTORED || !(ftiger || is_ca(cfg->domain) || is_sys(cfg->domain)) ||
            (is_ca(cfg->domain) && cfg->structure_prior == FBA_SP_FULLY_CONNECTED)) {
            fail(nullptr, FBA_EINVAL, "reinvigoration belief: needs a factored model (fbapomdp) of factored tiger, collision avoidance or sysadmin");
            delete c;
            return FBA_EINVAL;
        }
        if (cfg->particles < 1 || cfg->resample_amount < 1) {  // ReinvigoratingRejectionSampling.cpp:43-49
            fail(nullptr, FBA_EINVAL, "ReinvigoratingRejectionSampling::cannot initiate belief of size < 1 (%d), or resample size of < 1 (%d)",
                 cfg->particles, cfg->resample_amount);
            delete c;
            return FBA_EINVAL;
        }
        P.belief  = FBA_BELIEF_REJECTION;
        P.reinvig = cfg->resample_amount;
    }
    switch (cfg->domain) {
        case FBA_DOM_TIGER_EPISODIC:
        case FBA_DOM_TIGER_CONTINUOUS: P.S = 2; P.A = 3; P.O = 2; break;
        case FBA_DOM_FTIGER_EPISODIC:
        case FBA_DOM_FTIGER_CONTINUOUS:
            if (cfg->size < 1 || cfg->size > 12) {
                fail(nullptr, FBA_EINVAL, "cannot initiate FactoredTiger with %d irrelevant features", cfg->size);
                delete c;
                return FBA_EINVAL;
            }
            P.S = 2 << cfg->size; P.A = 3; P.O = 2;
            break;
        case FBA_DOM_AGR:  // AGR(10): 21 x 21 (goal, position) states, 21 help actions + work + observe, 21 positions + "none"
            if (cfg->model != FBA_MODEL_POMDP || cfg->belief != FBA_BELIEF_REJECTION) {  // AGR.cpp:307-310 (thrown by the first weighted update)
                fail(nullptr, FBA_EINVAL, "AGR::computeObservationProbability nyi");
                delete c;
                return FBA_EINVAL;
            }
            P.S = 21 * 21; P.A = 23; P.O = 22;
            break;
        case FBA_DOM_COFFEE:
        case FBA_DOM_COFFEE_BOUTILIER:  // CoffeeProblem.hpp: 5 binary features, {GetCoffee, CheckCoffee}, {Want, NotWant}
            if (cfg->model != FBA_MODEL_POMDP) {  // factory::makeBADomainExtension has no coffee entry
                fail(nullptr, FBA_EINVAL, "the coffee problem has no Bayes-adaptive extension: planning only");
                delete c;
                return FBA_EINVAL;
            }
            P.S = 32; P.A = 2; P.O = 2;
            break;
        case FBA_DOM_SYSADMIN_INDEPENDENT:
        case FBA_DOM_SYSADMIN_LINEAR:
            if (cfg->size < 1 || cfg->size > 8) {  // SysAdmin.cpp:17-21; 2N actions <= FBA_MAX_ACTIONS
                fail(nullptr, FBA_EINVAL, cfg->size < 1 ? "Cannot initiate Sysadmin with n %d" : "sysadmin: at most 8 computers (n = %d)", cfg->size);
                delete c;
                return FBA_EINVAL;
            }
            build_sysadmin(c->sysdesc, cfg->size);
            P.S = 1 << cfg->size; P.A = 2 * cfg->size; P.O = 2;
            break;
        case FBA_DOM_COLLISION_AVOID:
        case FBA_DOM_COLLISION_AVOID_CENTERED: {
            const int W = cfg->width, H = cfg->height, n = cfg->size;
            const char* bad = nullptr;
            char msg[160];
            if (W < 1) { snprintf(msg, sizeof msg, "Cannot initiate CollisionAvoidance with width %d", W); bad = msg; }
            else if (H < 1 || H % 2 != 1 || H > 15) { snprintf(msg, sizeof msg, "Cannot initiate CollisionAvoidance with height %d, must be uneven", H); bad = msg; }
            else if (n > W || n < 1 || n > MAXF - 2) { snprintf(msg, sizeof msg, "cannot initiate collision avoidance with more obstacles (%d ) than columns (%d)!", n, W); bad = msg; }
            if (bad) {
                fail(nullptr, FBA_EINVAL, "%s", bad);
                delete c;
                return FBA_EINVAL;
            }
            build_collision_avoidance(c->cadesc, W, H, n, cfg->domain == FBA_DOM_COLLISION_AVOID);
            P.S = W * H * c->cadesc.Hn; P.A = 3; P.O = c->cadesc.Hn;
            break;
        }
        case FBA_DOM_GRIDWORLD:
            if (cfg->size < 3 || cfg->size > 15) {
                fail(nullptr, FBA_EINVAL, "please enter a size larger than 3 to be able to run gridworld (you entered %d)", cfg->size);
                delete c;
                return FBA_EINVAL;
            }
            build_gridworld(c->gdesc, cfg->size);
            P.S = P.O = cfg->size * cfg->size * c->gdesc.G; P.A = 4;
            break;
        default:
            fail(nullptr, FBA_EINVAL, "domain %d is not supported by this build", cfg->domain);
            delete c;
            return FBA_EINVAL;
    }
    P.N = cfg->particles;
    P.sims = cfg->sims;
    P.horizon = cfg->horizon;
    P.max_depth = cfg->max_depth < 0 ? cfg->horizon : cfg->max_depth;  // ArgumentParser.cpp:37-40
    P.episodes = cfg->episodes;
    P.exploration = cfg->exploration;
    P.gamma = cfg->discount;
    P.seed_lo = (uint32_t)cfg->seed;
    P.seed_hi = (uint32_t)(cfg->seed >> 32);
    P.noise = cfg->noise;
    P.counts_total = cfg->counts_total;
    P.structure_prior = cfg->structure_prior;
    P.fd = nullptr;
    P.gw = nullptr;
    P.ca = nullptr;
    P.sys = nullptr;
    P.fd_bytes = 0;
    P.ca_plain = 0;
    P.zig = nullptr;
    P.dirichlet_regular = cfg->dirichlet_regular ? 1 : 0;
    P.packed = 0;
    if (cfg->model == FBA_MODEL_BA_TABLE) {
        P.phi_len = P.S * P.A * P.S;
        P.C       = P.phi_len + P.A * P.S * P.O;
    } else if (cfg->model == FBA_MODEL_POMDP) {
        P.phi_len = 0;
        P.C       = 0;
    } else if (cfg->model == FBA_MODEL_BA_FACTORED) {
        P.phi_len = 0;
        int rc = build_factored_prior(c);
        if (rc) {
            g_create_error = c->err;
            delete c;
            return rc;
        }
        P.C = c->fdesc.ncounts + c->fdesc.nvar;
    } else {
        fail(nullptr, FBA_EINVAL, "model %d is not supported by this build", cfg->model);
        delete c;
        return FBA_EINVAL;
    }
    // record stride (fba_state.h): counts + state word, padded to a power of two up to 64 words
    // (so small particles are whole cache lines), to a multiple of 4 words beyond that
    c->dense_C = P.C;
    // Packed particles (PackedView, fba_device.h): the tabular tiger particle as 24 uint16 increment counts over
    // the shared prior -- 64 bytes instead of 128 in every belief update and every root sample.  Only where every
    // kernel that touches the records is a tiger-table instantiation (launch_search / launch_belief_update), the
    // prior is exact under "+ 65535" (TigerBAPrior: 5000, 0 and the two listen counts below) and one run cannot
    // add more than 65535 to a cell (one T and one O increment per real step).
    if (cfg->model == FBA_MODEL_BA_TABLE && is_tiger(cfg->domain) && !cfg->dirichlet_regular && cfg->planner == FBA_PLANNER_POUCT &&
        (cfg->belief == FBA_BELIEF_REJECTION || cfg->belief == FBA_BELIEF_IMPORTANCE) && cfg->particles <= IS_MAX_CHUNKS * 256 &&
        !std::getenv("FBA_IS_MULTI_MIN") && !std::getenv("FBA_DENSE_PARTICLES") && (long long)cfg->episodes * cfg->horizon <= 65535 &&
        cfg->noise > -.15 && cfg->noise <= .3) {
        const std::vector<float> listen = {(.85f - cfg->noise) * cfg->counts_total, (.15f + cfg->noise) * cfg->counts_total, 5000.f, 0.f};
        if (packable_prior(listen)) {
            P.packed = 1;
            P.C      = (c->dense_C + 1) / 2;  // words holding the uint16 pairs; the state word follows
        }
    }
    // Packed factored-tiger particles (PackedFtigerView, fba_device.h): uint16 increments over a prior that is a function of
    // the cell and the particle's structure bits -- 144 B instead of 288 B at --size 3, so twice the beliefs per gigabyte and
    // per kilobyte of LDS.  Only where every kernel that touches the records is a factored-tiger instantiation (po-uct,
    // plain rejection filter, expected Dirichlet, --size 1..3) and the prior's five values are exact under "+ 65535".
    P.ft_packed = 0; P.ft_acc = P.ft_inacc = P.ft_unif = 0.f;
    if (cfg->model == FBA_MODEL_BA_FACTORED && is_ftiger(cfg->domain) && cfg->belief == FBA_BELIEF_REJECTION && !point &&
        cfg->planner == FBA_PLANNER_POUCT && !cfg->dirichlet_regular && cfg->size >= 1 && cfg->size <= 3 &&
        !std::getenv("FBA_DENSE_PARTICLES") && (long long)cfg->episodes * cfg->horizon <= 65535) {
        const float acc = (.85f - cfg->noise) * cfg->counts_total, inacc = (.15f + cfg->noise) * cfg->counts_total, unif = .5f * cfg->counts_total;
        if (packable_prior({acc, inacc, unif, 5000.f, 0.f})) {
            P.ft_packed = 1;
            P.ft_acc = acc; P.ft_inacc = inacc; P.ft_unif = unif;
            P.C = c->fdesc.ncounts / 2 + 1;   // words of uint16 pairs, the structure word; the state word follows
        }
    }
    // History particles (fba_device.h): the gridworld FBA-POMDP particle as one 4-byte entry per real step over the
    // shared prior tables -- 100-500 bytes instead of 191 KB (N = 7) -- where the importance filter is the plain one,
    // the run fits the record (one entry per belief update), the grid fits 3-bit coordinates and prior + j is exact in
    // fp32 for every count a run can reach (so a row read through the entries is bit for bit the dense row).
    P.hist = 0; P.hist_cap = 0; P.hist_row = 0; P.hist_base = nullptr; P.hist_alt = nullptr;
    P.hist_lds = nullptr; P.hist_rid_bytes = 0; P.hist_distinct = 0;
    if (cfg->model == FBA_MODEL_BA_FACTORED && cfg->domain == FBA_DOM_GRIDWORLD && cfg->belief == FBA_BELIEF_IMPORTANCE &&
        !cfg->dirichlet_regular && cfg->particles <= IS_MAX_CHUNKS * 256 && !std::getenv("FBA_IS_MULTI_MIN") &&
        !std::getenv("FBA_DENSE_PARTICLES") && (long long)cfg->episodes * cfg->horizon <= HIST_MAX_CAP && cfg->size <= HIST_MAX_N) {
        const int cap = cfg->episodes * cfg->horizon;
        build_gridworld_alt_prior(c);
        std::vector<float> counts(c->prior.begin(), c->prior.begin() + c->fdesc.ncounts);
        if (increments_exact(counts, cap + 1) && increments_exact(c->prior_alt, cap + 1)) {
            P.hist     = 1;
            P.hist_cap = cap;
            P.hist_row = std::max(c->gdesc.N, c->gdesc.G);
            P.gw_N = c->gdesc.N; P.gw_G = c->gdesc.G;
            uint32_t cells[4] = {0, 0, 0, 0};
            for (int gq = 0; gq < c->gdesc.G; ++gq)
                cells[gq >> 2] |= (uint32_t)(c->gdesc.goal[gq][0] * c->gdesc.N + c->gdesc.goal[gq][1]) << (8 * (gq & 3));
            P.gw_goalcell0 = cells[0]; P.gw_goalcell1 = cells[1]; P.gw_goalcell2 = cells[2]; P.gw_goalcell3 = cells[3];
            P.C        = 0;  // word 0 = state, word 1 = structure bits, words 2.. = entries
        }
    }
    P.hist_compact = P.hist && !(std::getenv("FBA_HIST_STRIDE") && !std::strcmp(std::getenv("FBA_HIST_STRIDE"), "full")) ? 1 : 0;   // (A/B switch: every record at the full stride)
    if (P.hist) P.Cs = (2 + P.hist_cap + 3) & ~3;
    else if (P.ft_packed) P.Cs = (P.C + 1 + 3) & ~3;   // 36 words at --size 3: the point is the bytes, not a power of two
    else {
        int need = P.C + 1, cs = 4;
        if (need <= 64) { while (cs < need) cs <<= 1; }
        else cs = (need + 3) & ~3;
        P.Cs = cs;
    }
    if (P.A > FBA_MAX_ACTIONS) {
        fail(nullptr, FBA_EINVAL, "more than %d actions", FBA_MAX_ACTIONS);
        delete c;
        return FBA_EINVAL;
    }
    if (cfg->belief == FBA_BELIEF_IMPORTANCE && P.N > (1 << 26)) {
        fail(nullptr, FBA_EINVAL, "importance sampling supports at most %d particles per slot", 1 << 26);
        delete c;
        return FBA_EINVAL;
    }

#define CHK(x)            \
    do {                  \
        int _rc = (x);    \
        if (_rc) {        \
            g_create_error = c->err; \
            fba_destroy(c); \
            return _rc;   \
        }                 \
    } while (0)
#define HIPC(call)                                                                                           \
    do {                                                                                                     \
        hipError_t _e = (call);                                                                              \
        if (_e != hipSuccess) {                                                                              \
            fail(nullptr, FBA_EHIP, "%s failed: %s", #call, hipGetErrorString(_e));                          \
            fba_destroy(c);                                                                                  \
            return FBA_EHIP;                                                                                 \
        }                                                                                                    \
    } while (0)

    HIPC(hipSetDevice(cfg->device));
    HIPC(hipStreamCreate(&c->stream));

    // node record layout (fba_state.h)
    DeviceState& D = c->D;
    const bool hashed = P.A * P.O > 64;
    // a hashed tree with an even number of actions drops the visits word (it is the sum of the action counts) and the
    // padding word with it: 56 -> 48 bytes per node for the four actions of gridworld
    D.cn_off     = (hashed && P.A % 2 == 0 && !std::getenv("FBA_NODE_VISITS")) ? 0 : 1;
    D.cq_off     = (D.cn_off + P.A + 1) & ~1;
    D.child_off  = D.cq_off + 2 * P.A;
    D.node_words = hashed ? D.child_off : ((D.child_off + P.A * P.O + 1) & ~1);
    if (const char* ev = std::getenv("FBA_NODE_WORDS")) D.node_words = std::max(D.node_words, std::atoi(ev));   // (layout experiments: padded records)
    D.max_nodes  = P.sims + 2;  // one new node per simulation at most
    // Episodic (factored) tiger: only `listen` continues an episode and it has two observations, so a node has at most
    // two children and the tree of a search of depth D is at most the complete binary tree with levels 0..D:
    // 2^(D+1) - 1 nodes, whatever the number of simulations (2047 at the default depth 10 against 4098 / 16386
    // records for 4096 / 16384 simulations).  search_kernel refuses to go past max_nodes (fault -> FBA_ESTATE).
    if ((cfg->domain == FBA_DOM_TIGER_EPISODIC || cfg->domain == FBA_DOM_FTIGER_EPISODIC) && P.max_depth >= 0 && P.max_depth < 24)
        D.max_nodes = std::min(D.max_nodes, (1 << (P.max_depth + 1)) + 1);  // (+1: odd, so that the slots' trees are not a power of two
                                                                             // of bytes apart -- at exactly 128 KB the search ran 45 % slower)
    if (const char* ev = std::getenv("FBA_NODE_BOUND")) D.max_nodes = std::max(2, std::atoi(ev));  // tests: exercise the overflow guard
    uint32_t hcap = 0;
    size_t hash_entry = 16;
    D.hash_compact = 0;
    if (hashed) {
        hcap = 64;
        while (hcap < 2u * (uint32_t)(D.max_nodes - 2)) hcap <<= 1;  // at most max_nodes - 2 children (one per simulation): load <= 1/2
        D.hmask = hcap - 1;
        if ((unsigned long long)D.max_nodes * P.A * P.O < (1ull << 27) && !std::getenv("FBA_WIDE_HASH")) {
            D.hash_compact = 1;
            hash_entry     = 8;
        }
    }

    // History-particle searches keep their tree in one table of 64-byte buckets (search_hist2_kernel, fba_state.h) where its keys fit:
    // (parent bucket, action, observation) in 28 bits, counts below the root in 16.  FBA_HIST_TREE=records keeps node records + hash table
    // (search_hist_kernel), for A/B runs and for what does not fit.
    D.bkt = nullptr; D.bkt_lines = 0; D.s_root = nullptr; D.search_order = nullptr;
    size_t bkt_lines = 0;
    const bool force_records = std::getenv("FBA_HIST_TREE") && !std::strcmp(std::getenv("FBA_HIST_TREE"), "records");
    if (P.hist && P.sims <= 65536 && !force_records) {
        long long buckets = cfg->tree_buckets > 0 ? cfg->tree_buckets : 2ll * (P.sims + 2);
        buckets = std::max(buckets, 8ll);
        const long long fit = (1ll << 28) / (4ll * P.O) - 2;   // ((buckets * 4 + 3) * O + O - 1 < 2^28: the root's "bucket" is index `buckets`)
        if (cfg->tree_buckets <= 0 || buckets <= fit) {
            buckets   = std::min(buckets, fit);
            bkt_lines = (size_t)((buckets + 1) / 2);
        }
    }
    if (cfg->tree_buckets < 0 || (cfg->tree_buckets > 0 && !bkt_lines && P.hist && !force_records)) {
        fail(nullptr, FBA_EINVAL, "tree_buckets = %d: not a size this context's search can use (history-particle searches of at most 65536 "
             "simulations, keys of 28 bits)", cfg->tree_buckets);
        fba_destroy(c);
        return FBA_EINVAL;
    }
    if (bkt_lines) {   // no node records, no separate hash table
        D.max_nodes = P.sims + 2;
        hcap = 0;
    }

    // slots
    const size_t per_slot = (size_t)P.N * (2 * 8 + 8 + 4 * (MAXINC + 1) + ((P.reinvig || P.cheat) ? 4 : (P.incub ? 6 : (P.hist ? 1 : 2))) * (size_t)P.Cs * 4) +
                            (bkt_lines ? bkt_lines * 128 : (size_t)D.max_nodes * D.node_words * 4 + (size_t)hcap * hash_entry) + 1024;
    int E = cfg->slots;
    if (E <= 0) {
        size_t free_b = 0, total_b = 0;
        HIPC(hipMemGetInfo(&free_b, &total_b));
        const size_t budget = free_b / 2;
        E = (int)std::min<size_t>(std::max<size_t>(budget / per_slot, 1), 32768);
        E = std::min(E, cfg->runs);
    }
    P.E = E;

    CHK(dev_alloc(c, &D.run, E));
    CHK(dev_alloc(c, &D.episode, E));
    CHK(dev_alloc(c, &D.t, E));
    CHK(dev_alloc(c, &D.active, E));
    CHK(dev_alloc(c, &D.need_update, E));
    CHK(dev_alloc(c, &D.need_reset, E));
    CHK(dev_alloc(c, &D.need_init, E));
    CHK(dev_alloc(c, &D.adv, E));
    CHK(dev_alloc(c, &D.env_state, E));
    CHK(dev_alloc(c, &D.ret, E));
    CHK(dev_alloc(c, &D.disc, E));
    CHK(dev_alloc(c, &D.action, E));
    CHK(dev_alloc(c, &D.obs, E));
    CHK(dev_alloc(c, &D.bufsel, E));
    const bool is = P.belief == FBA_BELIEF_IMPORTANCE || P.incub;   // (the incubator's shadow filter is importance-sampled)
    CHK(dev_alloc(c, &D.p_weight, is ? (size_t)2 * E * P.N : 1));
    // history particles keep ONE record buffer per slot; a resample / reset builds the new filter in a scratch pool shared by a
    // chunk of slots (launch_belief_update / launch_reset: for_each_chunk) -- the second buffer was a quarter of a slot's memory
    D.single_rec = P.hist && !std::getenv("FBA_DOUBLE_BUFFER") ? 1 : 0;
    D.ab_rows_hbm   = std::getenv("FBA_HIST_ROWS") && !std::strcmp(std::getenv("FBA_HIST_ROWS"), "hbm") ? 1 : 0;
    D.ab_hist_multi = std::getenv("FBA_HIST_MULTI") ? (std::atoi(std::getenv("FBA_HIST_MULTI")) != 0 ? 2 : 1) : 0;
    D.ab_no_etiger  = std::getenv("FBA_NO_ETIGER") && std::atoi(std::getenv("FBA_NO_ETIGER")) != 0 ? 1 : 0;
    D.ab_lockstep   = std::getenv("FBA_HIST_LOCKSTEP") ? (std::atoi(std::getenv("FBA_HIST_LOCKSTEP")) != 0 ? 1 : 0) : 1;   // (on; only the bucket tree has it, below)
    D.slot_base = 0;
    D.scratch_slots = D.single_rec ? std::min(E, 1024) : 0;
    if (D.single_rec)
        if (const char* ev = std::getenv("FBA_SCRATCH_SLOTS")) D.scratch_slots = std::max(1, std::min(E, std::atoi(ev)));   // (tests: several chunks with a few slots)
    D.rec_scratch = nullptr; D.copy_pending = nullptr; D.rec_buf = nullptr;
    CHK(dev_alloc(c, &D.p_rec, (D.single_rec ? (size_t)E + D.scratch_slots : (size_t)2 * E) * P.N * P.Cs, false));
    if (D.single_rec) {   // E + scratch_slots buffers; a slot's filter and a scratch place's buffer change places after a resample / reset (swap_buffers_kernel)
        CHK(dev_alloc(c, &D.rec_buf, (size_t)E + D.scratch_slots));
        std::vector<int32_t> ids((size_t)E + D.scratch_slots);
        for (size_t i = 0; i < ids.size(); ++i) ids[i] = (int32_t)i;
        HIPC(hipMemcpy(D.rec_buf, ids.data(), ids.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        CHK(dev_alloc(c, &D.copy_pending, E));
    }
    if (P.reinvig || P.cheat || P.incub) {
        CHK(dev_alloc(c, &D.p_rec_fc, (size_t)2 * E * P.N * P.Cs, false));
        CHK(dev_alloc(c, &D.bufsel_fc, E));
    }
    D.shadow = 0;
    if (P.incub) {
        CHK(dev_alloc(c, &D.p_rec_sh, (size_t)2 * E * P.N * P.Cs, false));
        CHK(dev_alloc(c, &D.p_weight_sh, (size_t)2 * E * P.N));
        CHK(dev_alloc(c, &D.bufsel_sh, E));
        CHK(dev_alloc(c, &D.need_update_sh, E));
        // Which shadow particles are bred anew: the reference asks its weighted filter for the `incub` least likely particles
        // (WeightedFilter.cpp:193-238) at a moment when all N weights are equal, so no particle ever displaces one of the first `incub`
        // and the answer is those indices in the order a binary max-heap of equal keys releases them.  That order is libstdc++'s
        // (std::push_heap / std::pop_heap, which its priority_queue is made of); it is computed here once, with a comparison that
        // finds no key smaller than another.
        std::vector<int32_t> heap, order;
        const auto equal_keys = [](int32_t, int32_t) { return false; };
        for (int i = 0; i < P.incub; ++i) {
            heap.push_back(i);
            std::push_heap(heap.begin(), heap.end(), equal_keys);
        }
        while (!heap.empty()) {
            std::pop_heap(heap.begin(), heap.end(), equal_keys);
            order.push_back(heap.back());
            heap.pop_back();
        }
        int32_t* d_order = nullptr;
        CHK(dev_alloc(c, &d_order, order.size()));
        HIPC(hipMemcpy(d_order, order.data(), order.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        D.inc_order = d_order;
    }
    CHK(dev_alloc(c, &D.wscan, is ? (size_t)E * P.N : 1, false));
    D.side_w = 1 + (P.model == FBA_MODEL_BA_FACTORED ? c->fdesc.FS + c->fdesc.FO : (P.model == FBA_MODEL_BA_TABLE ? 2 : 0));
    if (P.hist) D.side_w = 2;  // {new state, the step's entry}
    CHK(dev_alloc(c, &D.hist_cnt, E));
    CHK(dev_alloc(c, &D.p_side, is ? (size_t)E * P.N * D.side_w : 1, false));
    {
        // one workgroup per slot up to IS_MAX_CHUNKS*256 particles, several launches beyond
        // (FBA_IS_MULTI_MIN lowers the switch-over so tests can exercise the large-filter path)
        int multi_min = IS_MAX_CHUNKS * 256 + 1;
        if (const char* ev = std::getenv("FBA_IS_MULTI_MIN")) multi_min = std::max(1, std::atoi(ev));
        D.is_multi    = is && P.N >= multi_min;
        D.ctot_stride = (P.N + 255) / 256 + 2;
        CHK(dev_alloc(c, &D.ctot, is ? (size_t)E * D.ctot_stride : 1));
        CHK(dev_alloc(c, &D.is_tot, (size_t)2 * E));
    }
    if (P.nested) {
        CHK(dev_alloc(c, &D.nest_s, (size_t)2 * E * P.N * P.nested));
        CHK(dev_alloc(c, &D.nest_sel, E));
        CHK(dev_alloc(c, &D.nest_scan, (size_t)E * P.N));
        CHK(dev_alloc(c, &D.nest_total, E));
    }
    if (P.mh) {
        if (D.is_multi) {
            fail(c, FBA_EINVAL, "mh-within-gibbs belief: at most %d particles per slot", IS_MAX_CHUNKS * 256);
            g_create_error = c->err;
            fba_destroy(c);
            return FBA_EINVAL;
        }
        const size_t cap = (size_t)P.episodes * P.horizon;
        CHK(dev_alloc(c, &D.lik, (size_t)E + 1));
        CHK(dev_alloc(c, &D.cheat_pending, E));
        CHK(dev_alloc(c, &D.mh_a, (size_t)E * cap));
        CHK(dev_alloc(c, &D.mh_o, (size_t)E * cap));
        CHK(dev_alloc(c, &D.mh_ep_len, (size_t)E * (P.episodes + 1)));
        CHK(dev_alloc(c, &D.mh_n_ep, E));
        size_t words = (size_t)3 * P.Cs + (size_t)P.S * P.A * P.S + (size_t)P.A * P.S * P.O + 2;
        words += 2 * ((size_t)(P.horizon + 1) * P.S + P.S) + (size_t)P.episodes * (P.horizon + 1) + 2 * (size_t)P.horizon + 2;
        if (c->fdesc.nvar > MH_MAXVAR) {
            fail(c, FBA_EINVAL, "mh beliefs: at most %d structure words per particle", MH_MAXVAR);
            g_create_error = c->err;
            fba_destroy(c);
            return FBA_EINVAL;
        }
        D.mh_scratch_words = (int32_t)((words + 3) & ~(size_t)3);
        if (!mh_scratch_in_lds(D.mh_scratch_words)) CHK(dev_alloc(c, &D.mh_scratch, (size_t)E * D.mh_scratch_words));  // (else the chain's scratch is LDS)
        const double thr = cfg->threshold;
        HIPC(hipMemcpy(D.lik + E, &thr, sizeof thr, hipMemcpyHostToDevice));
    }
    if (P.cheat) {
        if (D.is_multi) {
            fail(c, FBA_EINVAL, "cheating-reinvigoration belief: at most %d particles per slot", IS_MAX_CHUNKS * 256);
            g_create_error = c->err;
            fba_destroy(c);
            return FBA_EINVAL;
        }
        CHK(dev_alloc(c, &D.lik, (size_t)E + 1));
        CHK(dev_alloc(c, &D.cheat_pending, E));
        const double thr = cfg->threshold;
        HIPC(hipMemcpy(D.lik + E, &thr, sizeof thr, hipMemcpyHostToDevice));
    }
    if (bkt_lines) {
        D.bkt_lines = (int32_t)bkt_lines;
        CHK(dev_alloc(c, &D.bkt, (size_t)E * bkt_lines * 8));   // (zeroed: key 0 carries epoch 0, which no search uses)
        CHK(dev_alloc(c, &D.epoch, E));
        CHK(dev_alloc(c, &D.nodes, 16));
    } else {
        CHK(dev_alloc(c, &D.nodes, (size_t)E * D.max_nodes * D.node_words, false));
        if (hashed) {
            CHK(dev_alloc(c, &D.hash, (size_t)E * hcap * hash_entry / 16));
            CHK(dev_alloc(c, &D.epoch, E));
        }
    }
    CHK(dev_alloc(c, &D.sim_steps, E));
    CHK(dev_alloc(c, &D.belief_steps, E));
    CHK(dev_alloc(c, &D.env_steps, E));
    CHK(dev_alloc(c, &D.upd_particles, E));
    CHK(dev_alloc(c, &D.ep_sums, (size_t)3 * E));
    CHK(dev_alloc(c, &D.upd_attempts, E));
    CHK(dev_alloc(c, &D.upd_entries, E));
    CHK(dev_alloc(c, &D.cur, E));
    CHK(dev_alloc(c, &D.trace_count, 1));
    CHK(dev_alloc(c, &c->d_n_active, 1));
    CHK(dev_alloc(c, &D.fault, 1));
    CHK(dev_alloc(c, &D.lazy_reset, E));
    P.search_budget = (P.hist && cfg->search_budget > 0) ? cfg->search_budget : 0;   // (only search_hist_kernel parks and resumes)
    if (P.search_budget > 0) {
        CHK(dev_alloc(c, &D.s_sim, E));
        CHK(dev_alloc(c, &D.s_nodes, E));
        CHK(dev_alloc(c, &D.s_depth, E));
        CHK(dev_alloc(c, &D.search_done, E));
        if (D.bkt) CHK(dev_alloc(c, &D.s_root, (size_t)6 * E));
        if (D.single_rec) {   // the chunked belief launches run over compacted lists of the slots that have work (fba_state.h)
            CHK(dev_alloc(c, &D.slot_list, E));
            CHK(dev_alloc(c, &D.scratch_idx, E));
            CHK(dev_alloc(c, &D.list_count, 1));
            HIPC(hipHostMalloc(reinterpret_cast<void**>(&D.list_count_host), sizeof(int32_t), hipHostMallocDefault));
        }
    }
    if (D.bkt && D.ab_lockstep) CHK(dev_alloc(c, &D.search_order, (size_t)E));
    if (!D.search_order) D.ab_lockstep = 0;   // (lock-step waves exist on the bucket tree only: search_hist2_kernel)
    CHK(dev_alloc(c, &c->d_prior, P.Cs));
    CHK(dev_alloc(c, &c->d_prior_dense, std::max(c->dense_C, 1)));
    if (P.hist) {
        const HistLayout L(c->gdesc.N, c->gdesc.G, P.A);
        CHK(dev_alloc(c, &c->d_hist_base, (size_t)L.total + 16));   // (+16: a row fetch may run up to a row width past the last row)
        CHK(dev_alloc(c, &c->d_hist_alt, (size_t)L.alt_total + 16));
        CHK(dev_alloc(c, &c->d_hist_lds, (size_t)HistRowIds(c->gdesc.N, c->gdesc.G, P.A).total + 16 + 256 * 16 * sizeof(float)));
        P.hist_base = c->d_hist_base;
        P.hist_alt  = c->d_hist_alt;
    }
    CHK(dev_alloc(c, &c->d_uni_scan, (size_t)P.N + 1));
    CHK(dev_alloc(c, &c->d_log1p, (size_t)P.sims + 2));
    D.prior     = c->d_prior;
    D.prior_dense = c->d_prior_dense;
    D.uni_scan  = c->d_uni_scan;
    D.log1p_tab = c->d_log1p;
    D.trace_on  = cfg->trace ? 1 : 0;
    D.trace_cap = 0;
    D.runs_total = cfg->runs;
    D.run_offset = cfg->run_offset;

    // log1p(m) table: UCB(m, n) = u * sqrt(log1p(m) / n)  (POUCT.cpp:330-338, factorised so the
    // n x n table -- 134 MB at 4096 simulations, int overflow at 65536 -- never exists)
    {
        std::vector<double> t((size_t)P.sims + 2);
        for (size_t m = 0; m < t.size(); ++m) t[m] = std::log1p((double)m);
        HIPC(hipMemcpyAsync(c->d_log1p, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPC(hipStreamSynchronize(c->stream));
    }
    if (is) {
        double* tmp = nullptr;
        CHK(dev_alloc(c, &tmp, (size_t)P.N + 1));
        launch_uniform_scan(P.N, tmp, c->d_uni_scan, c->d_uni_scan + P.N, D.ctot, c->stream);
        HIPC(hipMemcpyAsync(&D.uni_total, c->d_uni_scan + P.N, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPC(hipStreamSynchronize(c->stream));
    }
    if (P.incub && (1.0 / (double)P.N) / D.uni_total > cfg->threshold) {
        // reinvigorateBelief (StructureIncubatorSampling.cpp:155-188) tests normalised shadow weights that are uniform
        // whenever it runs (every update ends in a resample): none is promoted, or all of them -- and then every
        // shadow weight is zero and normalize() divides by their sum
        fail(c, FBA_EINVAL, "incubator belief: with threshold %g every one of the %d shadow particles (normalised weight %g) is promoted at "
             "once; the shadow filter's total weight is then zero and StructureIncubatorSampling.cpp:160-188 divides by it",
             cfg->threshold, P.N, (1.0 / (double)P.N) / D.uni_total);
        g_create_error = c->err;
        fba_destroy(c);
        return FBA_EINVAL;
    }
    if (cfg->model == FBA_MODEL_BA_TABLE) CHK(build_tabular_prior(c));
    if (P.dirichlet_regular) {
        if (cfg->model == FBA_MODEL_POMDP) {
            fail(c, FBA_EINVAL, "the Dirichlet sampling method only exists for Bayes-adaptive models");
            g_create_error = c->err;
            fba_destroy(c);
            return FBA_EINVAL;
        }
        int longest = std::max(P.S, P.O);
        if (cfg->model == FBA_MODEL_BA_FACTORED) {
            longest = 0;
            for (int f = 0; f < c->fdesc.FS; ++f) longest = std::max(longest, c->fdesc.Ssz[f]);
            for (int f = 0; f < c->fdesc.FO; ++f) longest = std::max(longest, c->fdesc.Osz[f]);
        }
        if (longest > MAXROW) {
            fail(c, FBA_EINVAL, "regular Dirichlet mode samples rows of at most %d counts (this model has %d)", MAXROW, longest);
            g_create_error = c->err;
            fba_destroy(c);
            return FBA_EINVAL;
        }
        ZigDesc z;
        build_ziggurat(z);
        CHK(dev_alloc(c, &c->d_zig, 1));
        HIPC(hipMemcpyAsync(c->d_zig, &z, sizeof z, hipMemcpyHostToDevice, c->stream));
        HIPC(hipStreamSynchronize(c->stream));
        P.zig = c->d_zig;
    }
    if (is_ca(cfg->domain)) {
        CHK(dev_alloc(c, &c->d_cadesc, 1));
        HIPC(hipMemcpyAsync(c->d_cadesc, &c->cadesc, sizeof(CADesc), hipMemcpyHostToDevice, c->stream));
        P.ca = c->d_cadesc;
    }
    if (is_sys(cfg->domain)) {
        CHK(dev_alloc(c, &c->d_sysdesc, 1));
        HIPC(hipMemcpyAsync(c->d_sysdesc, &c->sysdesc, sizeof(SysDesc), hipMemcpyHostToDevice, c->stream));
        P.sys = c->d_sysdesc;
    }
    if (cfg->domain == FBA_DOM_GRIDWORLD) {
        CHK(dev_alloc(c, &c->d_gdesc, 1));
        HIPC(hipMemcpyAsync(c->d_gdesc, &c->gdesc, sizeof(GridDesc), hipMemcpyHostToDevice, c->stream));
        P.gw = c->d_gdesc;
    }
    if (cfg->model == FBA_MODEL_BA_FACTORED) {
        FDesc& fdh = c->fdesc;
        const int nnodes = P.A * (fdh.FS + fdh.FO);
        for (int k = 0; k < nnodes; ++k)
            for (int j = 0; j < fdh.nodes[k].nmax; ++j) fdh.nodes[k].psz[j] = (uint8_t)fdh.Ssz[fdh.nodes[k].maxp[j]];
        P.fd_bytes = (int32_t)(offsetof(FDesc, nodes) + (size_t)nnodes * sizeof(FNode));
        P.ca_plain = (is_ca(cfg->domain) && fdh.nvar == 0 && fdh.nodes[2].nmax == 1) ||  // correct graph, no masks (not fully-connected)
                     (is_sys(cfg->domain) && fdh.nvar == 0);
        CHK(dev_alloc(c, &c->d_fdesc, 1));
        HIPC(hipMemcpyAsync(c->d_fdesc, &c->fdesc, sizeof(FDesc), hipMemcpyHostToDevice, c->stream));
        P.fd = c->d_fdesc;
    }
    CHK(upload_prior(c));
    // default positions for the per-step interface: slot e is run run_offset + e
    {
        std::vector<int32_t> run((size_t)E);
        for (int e = 0; e < E; ++e) run[e] = cfg->run_offset + e;
        HIPC(hipMemcpyAsync(D.run, run.data(), run.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPC(hipMemsetAsync(D.active, 1, (size_t)E, c->stream));
        HIPC(hipStreamSynchronize(c->stream));
    }
    HIPC(hipStreamSynchronize(c->stream));
    *out = c;
    return FBA_OK;
#undef CHK
#undef HIPC
}

int fba_domain_sizes(const fba_ctx* c, int32_t* S, int32_t* A, int32_t* O)
{
    if (!c) return FBA_EINVAL;
    *S = c->P.S; *A = c->P.A; *O = c->P.O;
    return FBA_OK;
}
int fba_counts_len(const fba_ctx* c) { return c ? c->dense_C : FBA_EINVAL; }
int fba_slots(const fba_ctx* c) { return c ? c->P.E : FBA_EINVAL; }
int fba_particle_bytes(const fba_ctx* c) { return c ? c->P.Cs * 4 : FBA_EINVAL; }

int fba_set_model_tabular(fba_ctx* c, const float* phi, const float* psi)
{
    if (!c || !phi || !psi) return FBA_EINVAL;
    if (c->P.model != FBA_MODEL_BA_TABLE) return fail(c, FBA_EINVAL, "fba_set_model_tabular needs model = BA_TABLE");
    std::vector<float> keep = c->prior;
    c->prior.assign((size_t)c->dense_C, 0.f);
    std::copy(phi, phi + c->P.phi_len, c->prior.begin());
    std::copy(psi, psi + (c->dense_C - c->P.phi_len), c->prior.begin() + c->P.phi_len);
    const int rc = upload_prior(c);
    if (rc) c->prior = keep;
    // The prior is what Belief::initiate copies into every particle (BAPOMDPPrior::sample).  Packed particles hold
    // increments over the table, so live particles would silently move to the new table: require a new initiate.
    else if (c->P.packed) c->belief_ready = false;
    return rc;
}

int fba_get_factored_layout(const fba_ctx* c, fba_factored_layout* out)
{
    if (!c || !out) return FBA_EINVAL;
    if (c->P.model != FBA_MODEL_BA_FACTORED) return fail(const_cast<fba_ctx*>(c), FBA_EINVAL, "fba_get_factored_layout: not a factored model");
    static_assert(FBA_MAX_FEATURES == MAXF && FBA_MAX_NODES == MAXNODES, "public layout struct mirrors FDesc");
    const FDesc& d = c->fdesc;
    memset(out, 0, sizeof *out);
    out->n_state_features = d.FS;
    out->n_obs_features   = d.FO;
    out->n_nodes          = c->P.A * (d.FS + d.FO);
    out->n_counts         = d.ncounts;
    out->n_mask_words     = d.nvar;
    for (int f = 0; f < d.FS; ++f) out->state_feature_size[f] = d.Ssz[f];
    for (int f = 0; f < d.FO; ++f) out->obs_feature_size[f] = d.Osz[f];
    for (int k = 0; k < out->n_nodes; ++k) {
        const FNode& nd       = d.nodes[k];
        fba_factored_node& o  = out->node[k];
        o.offset = nd.off; o.out = nd.out; o.n_candidates = nd.nmax; o.mask_word = nd.var; o.fixed_mask = nd.fixed_mask;
        for (int j = 0; j < nd.nmax; ++j) { o.candidate[j] = nd.maxp[j]; o.candidate_size[j] = (uint8_t)d.Ssz[nd.maxp[j]]; }
    }
    return FBA_OK;
}

int fba_set_model_factored(fba_ctx* c, const fba_factored_layout* layout, const float* counts)
{
    if (!c || !layout || !counts) return FBA_EINVAL;
    if (c->P.model != FBA_MODEL_BA_FACTORED) return fail(c, FBA_EINVAL, "fba_set_model_factored needs model = BA_FACTORED");
    fba_factored_layout mine;
    int rc = fba_get_factored_layout(c, &mine);
    if (rc) return rc;
    bool same = layout->n_state_features == mine.n_state_features && layout->n_obs_features == mine.n_obs_features &&
                layout->n_nodes == mine.n_nodes && layout->n_counts == mine.n_counts && layout->n_mask_words == mine.n_mask_words;
    for (int k = 0; same && k < mine.n_nodes; ++k) {
        const fba_factored_node &a = layout->node[k], &b = mine.node[k];
        same = a.offset == b.offset && a.out == b.out && a.n_candidates == b.n_candidates && a.mask_word == b.mask_word &&
               std::memcmp(a.candidate, b.candidate, (size_t)b.n_candidates) == 0;
    }
    if (!same)
        return fail(c, FBA_EINVAL, "fba_set_model_factored: the layout is not this context's (obtain it with fba_get_factored_layout and fill "
                                   "the counts where it says each node's rows are)");
    for (int k = 0; k < mine.n_counts; ++k)
        if (!(counts[k] >= 0.f)) return fail(c, FBA_EINVAL, "fba_set_model_factored: count %d is %g: Dirichlet counts cannot be negative", k, (double)counts[k]);
    std::vector<float> keep = c->prior;
    c->prior.assign(counts, counts + mine.n_counts + mine.n_mask_words);
    if (c->P.ft_packed) {
        c->prior = keep;
        return fail(c, FBA_EINVAL, "this context stores factored-tiger particles packed over the built-in prior; create it with "
                                   "FBA_DENSE_PARTICLES=1 in the environment to replace the prior");
    }
    if (c->P.hist) {  // history particles read rows as prior + j: the new table must keep that exact (fba_create checked the built-in one)
        std::vector<float> only(c->prior.begin(), c->prior.begin() + mine.n_counts);
        if (!increments_exact(only, c->P.hist_cap + 1)) {
            c->prior = keep;
            return fail(c, FBA_EINVAL, "this context stores particles as histories over the prior table, which needs prior counts c with c + j exact "
                                       "in fp32 for every j up to episodes * horizon; create it with FBA_DENSE_PARTICLES=1 in the environment for this table");
        }
    }
    rc = upload_prior(c);
    if (rc) c->prior = keep;
    else c->belief_ready = false;
    return rc;
}

int fba_get_prior(const fba_ctx* c, float* counts)
{
    if (!c || !counts) return FBA_EINVAL;
    std::copy(c->prior.begin(), c->prior.end(), counts);
    return FBA_OK;
}

int fba_set_position(fba_ctx* c, const int32_t* run, const int32_t* episode, const int32_t* t)
{
    if (!c) return FBA_EINVAL;
    const size_t n = (size_t)c->P.E * 4;
    int rc;
    if ((rc = clear_parked_searches(c))) return rc;
    if (run || episode) launch_materialize_reset(c->P, c->D, c->stream);  // a pending lazy reset belongs to the old (run, episode)
    if (run) HIPCHK(c, hipMemcpyAsync(c->D.run, run, n, hipMemcpyHostToDevice, c->stream));
    if (episode) HIPCHK(c, hipMemcpyAsync(c->D.episode, episode, n, hipMemcpyHostToDevice, c->stream));
    if (t) HIPCHK(c, hipMemcpyAsync(c->D.t, t, n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FBA_OK;
}

int fba_belief_init(fba_ctx* c)
{
    if (!c) return FBA_EINVAL;
    int rc;
    if ((rc = clear_parked_searches(c))) return rc;
    if ((rc = set_flags(c, c->D.need_init, nullptr, 1))) return rc;
    if ((rc = timed(c, FBA_K_BELIEF_INIT, [&] { launch_init(c->P, c->D, c->stream); }))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->belief_ready = true;
    c->packed_updates.assign((size_t)c->P.E, 0);
    return FBA_OK;
}

int fba_belief_reset_domain_state(fba_ctx* c)
{
    if (!c) return FBA_EINVAL;
    if (c->P.model == FBA_MODEL_POMDP) return fail(c, FBA_EINVAL, "resetDomainStateDistribution exists for Bayes-adaptive beliefs only");
    if (!c->belief_ready) return fail(c, FBA_ESTATE, "belief not initiated");
    int rc;
    if ((rc = set_flags(c, c->D.need_reset, nullptr, 1))) return rc;
    if ((rc = timed(c, FBA_K_BELIEF_RESET, [&] { launch_reset(c->P, c->D, c->stream); }))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FBA_OK;
}

int fba_select_action(fba_ctx* c, const int32_t* hist_len, const uint8_t* active, int32_t* action)
{
    if (!c || !action) return FBA_EINVAL;
    if (!c->belief_ready) return fail(c, FBA_ESTATE, "belief not initiated");
    int rc;
    if (hist_len) HIPCHK(c, hipMemcpyAsync(c->D.t, hist_len, (size_t)c->P.E * 4, hipMemcpyHostToDevice, c->stream));
    if ((rc = set_flags(c, c->D.active, active, 1))) return rc;
    {   // Planner::selectAction returns a finished search: whole searches in one launch, whatever the context's search_budget
        Problem Pw = c->P;
        DeviceState Dw = c->D;
        Pw.search_budget = 0;
        Dw.search_done   = nullptr;
        if ((rc = timed(c, FBA_K_SEARCH, [&] { launch_search(Pw, Dw, c->stream); }))) return rc;
    }
    HIPCHK(c, hipMemcpyAsync(action, c->D.action, (size_t)c->P.E * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_fault(c);
}

int fba_belief_update(fba_ctx* c, const int32_t* action, const int32_t* obs, const uint8_t* active)
{
    if (!c || !action || !obs) return FBA_EINVAL;
    if (!c->belief_ready) return fail(c, FBA_ESTATE, "belief not initiated");
    for (int e = 0; e < c->P.E; ++e) {
        if (active && !active[e]) continue;
        if (action[e] < 0 || action[e] >= c->P.A) return fail(c, FBA_EINVAL, "action %d out of range", action[e]);
        if (obs[e] < 0 || obs[e] >= c->P.O) return fail(c, FBA_EINVAL, "observation %d out of range", obs[e]);
    }
    if (c->P.packed || c->P.ft_packed) {
        // a packed record counts a cell's "+1"s in 16 bits: the experiment loops cannot exceed that (episodes * horizon <= 65535 is
        // part of the packing condition), a host driving the per-step interface can -- refuse instead of wrapping into the next cell
        if (c->packed_updates.size() != (size_t)c->P.E) c->packed_updates.assign((size_t)c->P.E, 0);
        for (int e = 0; e < c->P.E; ++e) {
            if (active && !active[e]) continue;
            if (c->packed_updates[e] >= 65535u)
                return fail(c, FBA_ESTATE, "slot %d: more than 65535 belief updates since fba_belief_init, which packed particle records cannot count; "
                            "create the context with FBA_DENSE_PARTICLES=1 in the environment to drive it beyond that", e);
        }
        for (int e = 0; e < c->P.E; ++e)
            if (!active || active[e]) ++c->packed_updates[e];
    }
    int rc;
    HIPCHK(c, hipMemcpyAsync(c->D.action, action, (size_t)c->P.E * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->D.obs, obs, (size_t)c->P.E * 4, hipMemcpyHostToDevice, c->stream));
    if ((rc = set_flags(c, c->D.need_update, active, 1))) return rc;
    if ((rc = timed(c, k_update_kind(c), [&] { launch_belief_update(c->P, c->D, c->stream); }))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    return check_fault(c);
}

// History particles: the dense count table of one record, as the API speaks of particles -- the prior (the goal-parent
// form of the x / y nodes the structure bits name) plus 1.0f per entry and incremented cell
// (BABNModel::incrementCountsOf BABNModel.cpp:354-382 replayed; `cnt` = the slot's entries per action).
static void hist_materialize(const fba_ctx* c, const uint32_t* rec, uint32_t cnt, float* counts)
{
    const GridDesc& g = c->gdesc;
    const int N = g.N, G = g.G, A = c->P.A;
    const int XY = N * N * G * N, GG = N * N * G * G, NN = N * N, ncounts = c->fdesc.ncounts;
    const uint32_t mask = rec[1];
    std::copy(c->prior.begin(), c->prior.begin() + ncounts, counts);
    for (int a = 0; a < A; ++a)
        for (int f = 0; f < 2; ++f) {
            const bool with_goal = (mask >> (2 * a + f)) & 1u;
            if (with_goal) std::copy(c->prior_alt.begin() + (size_t)(a * 2 + f) * XY, c->prior_alt.begin() + (size_t)(a * 2 + f + 1) * XY,
                                     counts + a * (2 * XY + GG) + f * XY);
            const uint32_t m = with_goal ? 7u : 3u;
            std::memcpy(&counts[ncounts + 2 * a + f], &m, 4);
        }
    int j = 0;
    for (int a = 0; a < A; ++a) {
        const int tbase = a * (2 * XY + GG), obase = A * (2 * XY + GG) + a * (2 * NN + G * G);
        const bool mx = (mask >> (2 * a)) & 1u, my = (mask >> (2 * a + 1)) & 1u;
        for (int q = 0; q < hist_count(cnt, a); ++q, ++j) {
            const uint32_t en = rec[2 + j], s0 = en & 0x3ffu, s1 = (en >> 10) & 0x3ffu, ob = en >> 20;
            const int x = hist_x(s0), y = hist_y(s0), gl = hist_g(s0), cell = x * N + y;
            counts[tbase + (mx ? cell * G + gl : cell) * N + hist_x(s1)] += 1.0f;
            counts[tbase + XY + (my ? cell * G + gl : cell) * N + hist_y(s1)] += 1.0f;
            counts[tbase + 2 * XY + (cell * G + gl) * G + hist_g(s1)] += 1.0f;
            counts[obase + x * N + hist_x(ob)] += 1.0f;
            counts[obase + NN + y * N + hist_y(ob)] += 1.0f;
            counts[obase + 2 * NN + gl * G + hist_g(ob)] += 1.0f;
        }
    }
}

// particles [first, first + n) of slot `slot`'s filter, as the API speaks of particles (fp32 count tables whatever the record format)
static int belief_get_range(fba_ctx* c, int32_t slot, int32_t first, int32_t n, int32_t* state, double* weight, float* counts)
{
    const Problem& P = c->P;
    launch_materialize_reset(c->P, c->D, c->stream);  // a lazily reset rejection filter: write the states out first
    HIPCHK(c, hipStreamSynchronize(c->stream));
    uint8_t sel = 0;
    HIPCHK(c, hipMemcpy(&sel, c->D.bufsel + slot, 1, hipMemcpyDeviceToHost));
    const size_t pb = ((size_t)sel * P.E + slot) * (size_t)P.N;
    if (weight) {
        if (P.belief != FBA_BELIEF_IMPORTANCE) return fail(c, FBA_EINVAL, "the rejection filter is unweighted");
        HIPCHK(c, hipMemcpy(weight, c->D.p_weight + pb + first, (size_t)n * 8, hipMemcpyDeviceToHost));
    }
    if (state || (counts && (P.C || P.hist))) {
        std::vector<float> tmp((size_t)n * P.Cs);
        size_t rb = pb;
        if (c->D.single_rec) {   // (history particles: one record buffer per slot, whichever of the pool it currently is)
            int32_t buf = 0;
            HIPCHK(c, hipMemcpy(&buf, c->D.rec_buf + slot, 4, hipMemcpyDeviceToHost));
            rb = (size_t)buf * (size_t)P.N;
        }
        uint32_t hist_cnt = 0;
        if (P.hist) HIPCHK(c, hipMemcpy(&hist_cnt, c->D.hist_cnt + slot, 4, hipMemcpyDeviceToHost));
        const size_t rs = P.hist ? (size_t)hist_stride(P, hist_total(hist_cnt)) : (size_t)P.Cs;   // words between this slot's records
        HIPCHK(c, hipMemcpy(tmp.data(), c->D.p_rec + rb * P.Cs + (size_t)first * rs, (size_t)n * rs * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) {
            const float* rec = tmp.data() + (size_t)i * rs;
            if (state) std::memcpy(&state[i], &rec[P.C], 4);
            if (counts && P.ft_packed) {  // count = prior(cell, structure) + increments, then the structure word (PackedFtigerView)
                const uint32_t* w = reinterpret_cast<const uint32_t*>(rec);
                const int nc = c->fdesc.ncounts;
                const uint32_t mask = w[nc / 2];
                float* out = counts + (size_t)i * c->dense_C;
                for (int k = 0; k < nc; ++k)
                    out[k] = ftiger_prior_host(c->fdesc.FS, k, mask, P.ft_acc, P.ft_inacc, P.ft_unif) + (float)((k & 1) ? (w[k >> 1] >> 16) : (w[k >> 1] & 0xffffu));
                std::memcpy(&out[nc], &mask, 4);
            } else if (counts && P.hist) hist_materialize(c, reinterpret_cast<const uint32_t*>(rec), hist_cnt, counts + (size_t)i * c->dense_C);
            else if (counts && P.packed) {  // count = prior + number of increments (PackedView)
                const uint32_t* w = reinterpret_cast<const uint32_t*>(rec);
                for (int k = 0; k < c->dense_C; ++k)
                    counts[(size_t)i * c->dense_C + k] = c->prior[k] + (float)((k & 1) ? (w[k >> 1] >> 16) : (w[k >> 1] & 0xffffu));
            } else if (counts && P.C) std::copy(rec, rec + P.C, counts + (size_t)i * P.C);
        }
    }
    return FBA_OK;
}

int fba_belief_get(fba_ctx* c, int32_t slot, int32_t* state, double* weight, float* counts)
{
    if (!c || slot < 0 || slot >= c->P.E) return FBA_EINVAL;
    return belief_get_range(c, slot, 0, c->P.N, state, weight, counts);
}

int fba_belief_get_particle(fba_ctx* c, int32_t slot, int32_t index, int32_t* state, double* weight, float* counts)
{
    if (!c || slot < 0 || slot >= c->P.E) return FBA_EINVAL;
    if (index < 0 || index >= c->P.N) return fail(c, FBA_EINVAL, "particle %d out of range (the filter holds %d)", index, c->P.N);
    if (c->P.nested) return fail(c, FBA_EINVAL, "the nested belief's particles are (model, state filter) pairs: fba_belief_get / fba_belief_get_nested");
    return belief_get_range(c, slot, index, 1, state, weight, counts);
}

int fba_belief_get_fully_connected(fba_ctx* c, int32_t slot, int32_t* state, float* counts)
{
    if (!c || slot < 0 || slot >= c->P.E) return FBA_EINVAL;
    const Problem& P = c->P;
    if (!P.reinvig && !P.cheat && !P.incub) return fail(c, FBA_EINVAL, "only the reinvigoration, incubator and cheating beliefs have a second filter");
    uint8_t sel = 0;
    HIPCHK(c, hipMemcpy(&sel, c->D.bufsel_fc + slot, 1, hipMemcpyDeviceToHost));
    const size_t pb = ((size_t)sel * P.E + slot) * (size_t)P.N;
    std::vector<float> tmp((size_t)P.N * P.Cs);
    HIPCHK(c, hipMemcpy(tmp.data(), c->D.p_rec_fc + pb * P.Cs, tmp.size() * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < P.N; ++i) {
        const float* rec = tmp.data() + (size_t)i * P.Cs;
        if (state) std::memcpy(&state[i], &rec[P.C], 4);
        if (counts) std::copy(rec, rec + P.C, counts + (size_t)i * P.C);
    }
    return FBA_OK;
}

int fba_belief_get_shadow(fba_ctx* c, int32_t slot, int32_t* state, double* weight, float* counts)
{
    if (!c || slot < 0 || slot >= c->P.E) return FBA_EINVAL;
    const Problem& P = c->P;
    if (!P.incub) return fail(c, FBA_EINVAL, "only the incubator belief has a shadow filter");
    uint8_t sel = 0;
    HIPCHK(c, hipMemcpy(&sel, c->D.bufsel_sh + slot, 1, hipMemcpyDeviceToHost));
    const size_t pb = ((size_t)sel * P.E + slot) * (size_t)P.N;
    if (weight) HIPCHK(c, hipMemcpy(weight, c->D.p_weight_sh + pb, (size_t)P.N * sizeof(double), hipMemcpyDeviceToHost));
    std::vector<float> tmp((size_t)P.N * P.Cs);
    HIPCHK(c, hipMemcpy(tmp.data(), c->D.p_rec_sh + pb * P.Cs, tmp.size() * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < P.N; ++i) {
        const float* rec = tmp.data() + (size_t)i * P.Cs;
        if (state) std::memcpy(&state[i], &rec[P.C], 4);
        if (counts) std::copy(rec, rec + P.C, counts + (size_t)i * P.C);
    }
    return FBA_OK;
}

int fba_belief_get_nested(fba_ctx* c, int32_t slot, int32_t* states)
{
    if (!c || slot < 0 || slot >= c->P.E || !states) return FBA_EINVAL;
    const Problem& P = c->P;
    if (!P.nested) return fail(c, FBA_EINVAL, "only the nested belief has flat filters of domain states");
    int32_t sel = 0;
    HIPCHK(c, hipMemcpy(&sel, c->D.nest_sel + slot, sizeof sel, hipMemcpyDeviceToHost));
    const size_t per = (size_t)P.N * P.nested, off = ((size_t)sel * P.E + slot) * per;
    HIPCHK(c, hipMemcpy(states, c->D.nest_s + off, per * sizeof(int32_t), hipMemcpyDeviceToHost));
    return FBA_OK;
}

int fba_belief_set(fba_ctx* c, int32_t slot, const int32_t* state, const double* weight, const float* counts)
{
    if (!c || slot < 0 || slot >= c->P.E) return FBA_EINVAL;
    const Problem& P = c->P;
    if (P.nested) return fail(c, FBA_EINVAL, "the nested belief cannot be set from the host (its weights' prefix sums and flat filters live on the device)");
    launch_materialize_reset(c->P, c->D, c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    uint8_t sel = 0;
    HIPCHK(c, hipMemcpy(&sel, c->D.bufsel + slot, 1, hipMemcpyDeviceToHost));
    const size_t pb = ((size_t)sel * P.E + slot) * (size_t)P.N;
    if (P.hist && (state || counts))
        return fail(c, FBA_EINVAL, "this context stores particles as histories of their own steps over the shared prior (gridworld FBA-POMDP), which "
                                   "cannot take on arbitrary states or counts; create it with FBA_DENSE_PARTICLES=1 in the environment to set them");
    if (state)
        for (int i = 0; i < P.N; ++i)
            if (state[i] < 0 || state[i] >= P.S) return fail(c, FBA_EINVAL, "state %d out of range", state[i]);
    if (weight && P.belief == FBA_BELIEF_IMPORTANCE) HIPCHK(c, hipMemcpy(c->D.p_weight + pb, weight, (size_t)P.N * 8, hipMemcpyHostToDevice));
    if (state || (counts && P.C)) {
        std::vector<float> tmp((size_t)P.N * P.Cs);
        HIPCHK(c, hipMemcpy(tmp.data(), c->D.p_rec + pb * P.Cs, tmp.size() * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < P.N; ++i) {
            float* rec = tmp.data() + (size_t)i * P.Cs;
            if (state) std::memcpy(&rec[P.C], &state[i], 4);
            if (counts && P.ft_packed) {
                uint32_t* w = reinterpret_cast<uint32_t*>(rec);
                const int nc = c->fdesc.ncounts;
                const float* in = counts + (size_t)i * c->dense_C;
                uint32_t mask;
                std::memcpy(&mask, &in[nc], 4);
                for (int k = 0; k < P.C; ++k) w[k] = 0;
                w[nc / 2] = mask;
                for (int k = 0; k < nc; ++k) {
                    const double inc = (double)in[k] - (double)ftiger_prior_host(c->fdesc.FS, k, mask, P.ft_acc, P.ft_inacc, P.ft_unif);
                    if (inc < 0 || inc > 65535 || inc != std::floor(inc))
                        return fail(c, FBA_EINVAL, "particle %d, count %d: %g is not the prior of its structure plus 0..65535 increments, which packed "
                                                   "particles need; create the context with FBA_DENSE_PARTICLES=1 in the environment for arbitrary counts",
                                    i, k, (double)in[k]);
                    w[k >> 1] |= (uint32_t)inc << (16 * (k & 1));
                }
            } else if (counts && P.packed) {
                uint32_t* w = reinterpret_cast<uint32_t*>(rec);
                for (int k = 0; k < P.C; ++k) w[k] = 0;
                for (int k = 0; k < c->dense_C; ++k) {
                    const double inc = (double)counts[(size_t)i * c->dense_C + k] - (double)c->prior[k];
                    if (inc < 0 || inc > 65535 || inc != std::floor(inc))
                        return fail(c, FBA_EINVAL, "particle %d, count %d: %g is not the prior (%g) plus 0..65535 increments, which packed particles "
                                                   "need; create the context with FBA_DENSE_PARTICLES=1 in the environment for arbitrary counts",
                                    i, k, (double)counts[(size_t)i * c->dense_C + k], (double)c->prior[k]);
                    w[k >> 1] |= (uint32_t)inc << (16 * (k & 1));
                }
            } else if (counts && P.C) std::copy(counts + (size_t)i * P.C, counts + (size_t)(i + 1) * P.C, rec);
        }
        HIPCHK(c, hipMemcpy(c->D.p_rec + pb * P.Cs, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice));
    }
    c->belief_ready = true;
    return FBA_OK;
}

int fba_last_step_info(fba_ctx* c, fba_trace_rec* recs)
{
    if (!c || !recs) return FBA_EINVAL;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(recs, c->D.cur, (size_t)c->P.E * sizeof(fba_trace_rec), hipMemcpyDeviceToHost));
    return FBA_OK;
}

int fba_run_planning(fba_ctx* c, fba_stat* stats)
{
    if (!c || !stats) return FBA_EINVAL;
    if (c->P.model != FBA_MODEL_POMDP) return fail(c, FBA_EINVAL, "planning::run needs model = POMDP");
    return run_experiment(c, stats);
}

int fba_run_bapomdp(fba_ctx* c, fba_stat* stats)
{
    if (!c || !stats) return FBA_EINVAL;
    if (c->P.model == FBA_MODEL_POMDP) return fail(c, FBA_EINVAL, "bapomdp::run needs a Bayes-adaptive model");
    return run_experiment(c, stats);
}

int fba_run_ticks(fba_ctx* c, int32_t ticks)
{
    if (!c || ticks < 0) return FBA_EINVAL;
    int rc;
    if (!c->started) {
        if (c->cfg.trace)   // room for one run's worth of records per slot (later ones are counted, not kept)
            if ((rc = ensure_trace(c, (size_t)c->P.E * c->P.episodes * c->P.horizon))) return rc;
        if ((rc = start_experiment(c, -1))) return rc;
        c->started = true;
    }
    if (c->P.search_budget > 0) {
        // slots advance on their own: "ticks" real steps per slot on average = launches until the slots have together made
        // ticks x slots of them (a launch = one budget of search iterations + the step and belief update of the slots that finished)
        int rc2 = FBA_OK;
        const uint64_t base   = sum_counter(c, c->D.env_steps, c->P.E, &rc2);
        const uint64_t target = (uint64_t)ticks * (uint64_t)c->P.E;
        // (a simulation takes at most horizon + 1 loop iterations: the last one finishes it when depth-to-go is 0)
        const long long bound = ((long long)ticks + 2) * (((long long)c->P.sims * (c->P.horizon + 1)) / c->P.search_budget + 2);
        uint64_t made = 0;
        for (long long k = 0; k < bound && !rc2; ++k) {
            made = sum_counter(c, c->D.env_steps, c->P.E, &rc2) - base;   // (synchronises: the launches of the last round are done)
            if (made >= target) break;
            if ((rc = tick(c))) return rc;
        }
        if (rc2) return rc2;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if ((rc = check_fault(c))) return rc;
        if (made < target) {
            made = sum_counter(c, c->D.env_steps, c->P.E, &rc2) - base;
            if (rc2) return rc2;
            if (made < target)
                return fail(c, FBA_ESTATE, "budgeted run_ticks: %llu of %llu real steps made after %lld launches", (unsigned long long)made,
                            (unsigned long long)target, bound);
        }
        return FBA_OK;
    }
    for (int k = 0; k < ticks; ++k)
        if ((rc = tick(c))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_fault(c);
}

int fba_get_returns(const fba_ctx* cc, double* returns, int32_t* lengths)
{
    fba_ctx* c = const_cast<fba_ctx*>(cc);
    if (!c) return FBA_EINVAL;
    const size_t n = (size_t)c->cfg.runs * c->P.episodes;
    if (n > c->returns_cap) return fail(c, FBA_ESTATE, "no experiment has been run on this ctx");
    if (returns) HIPCHK(c, hipMemcpy(returns, c->D.returns, n * sizeof(double), hipMemcpyDeviceToHost));
    if (lengths) HIPCHK(c, hipMemcpy(lengths, c->D.lengths, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return FBA_OK;
}

int fba_get_return_sums(fba_ctx* c, double* out)
{
    if (!c || !out) return FBA_EINVAL;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<double> h((size_t)3 * c->P.E);
    HIPCHK(c, hipMemcpy(h.data(), c->D.ep_sums, h.size() * sizeof(double), hipMemcpyDeviceToHost));
    out[0] = out[1] = out[2] = 0;
    for (int e = 0; e < c->P.E; ++e)
        for (int k = 0; k < 3; ++k) out[k] += h[(size_t)3 * e + k];
    return FBA_OK;
}

int fba_get_counters(fba_ctx* c, fba_counters* out)
{
    if (!c || !out) return FBA_EINVAL;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int rc = FBA_OK;
    out->sim_steps    = sum_counter(c, c->D.sim_steps, c->P.E, &rc);
    out->belief_steps = sum_counter(c, c->D.belief_steps, c->P.E, &rc);
    out->env_steps    = sum_counter(c, c->D.env_steps, c->P.E, &rc);
    return rc;
}

int fba_get_kernel_times(fba_ctx* c, fba_kernel_time* out)
{
    if (!c || !out) return FBA_EINVAL;
    int rc = flush_events(c);
    if (rc) return rc;
    const Problem& P = c->P;
    const uint64_t sim       = sum_counter(c, c->D.sim_steps, P.E, &rc) - c->base_sim;
    const uint64_t attempts  = sum_counter(c, c->D.upd_attempts, P.E, &rc) - c->base_attempts;
    const uint64_t particles = sum_counter(c, c->D.upd_particles, P.E, &rc) - c->base_particles;
    const uint64_t entries   = sum_counter(c, c->D.upd_entries, P.E, &rc) - c->base_entries;
    if (rc) return rc;
    // algorithmic bytes, SURVEY.md section 8(d): Pb = particle payload, Rt / Ro = bytes of the
    // transition / observation rows one step consults
    // (SURVEY's dense figure, 4 bytes per count, unless one of the packed / history formulas below applies: DESIGN.md section 5)
    const uint64_t cell = 4;
    const uint64_t Pb = 4 + cell * (uint64_t)c->dense_C;
    uint64_t Rt = 0, Ro = 0;
    if (P.model == FBA_MODEL_BA_TABLE) { Rt = cell * (uint64_t)P.S; Ro = cell * (uint64_t)P.O; }
    if (P.model == FBA_MODEL_BA_FACTORED) {
        for (int f = 0; f < c->fdesc.FS; ++f) Rt += 4 * (uint64_t)c->fdesc.Ssz[f];
        for (int f = 0; f < c->fdesc.FO; ++f) Ro += 4 * (uint64_t)c->fdesc.Osz[f];
    }
    for (int k = 0; k < FBA_K_COUNT; ++k) {
        out[k].ms = c->k_ms[k];
        out[k].launches = c->k_launches[k];
        out[k].units = 0;
        out[k].bytes = 0;
    }
    out[FBA_K_SEARCH].units = sim;
    if (P.belief == FBA_BELIEF_REJECTION) {
        out[FBA_K_BELIEF_RS].units = particles;
        out[FBA_K_BELIEF_RS].bytes = attempts * (Pb + Rt + Ro) + particles * Pb;
        // reject_tiger_lds_kernel (packed particles, N <= TIGER_LDS_MAX_N): the alternative formula SURVEY 8(d) asks to be
        // stated when the build does not move dense particles -- the attempts run from LDS, so an update reads the
        // filter once to park it, reads the N accepted sources and writes N records, 64 bytes each
        if (P.packed && P.N <= TIGER_LDS_MAX_N) out[FBA_K_BELIEF_RS].bytes = particles * 3 * (uint64_t)(P.Cs * 4);
        // packed factored-tiger records (reject_kernel on 144-byte records at --size 3): SURVEY's formula on the bytes a packed particle has --
        // an attempt reads its source record and one 4-byte word per two-cell Dirichlet row it samples, an accepted one writes a record
        if (P.ft_packed) {
            const uint64_t rec = (uint64_t)P.Cs * 4, rows = 4 * (uint64_t)(c->fdesc.FS + c->fdesc.FO);
            out[FBA_K_BELIEF_RS].bytes = attempts * (rec + rows) + particles * rec;
        }
    } else {
        out[FBA_K_BELIEF_IS].units = particles;
        out[FBA_K_BELIEF_IS].bytes = particles * (32 + Rt + Ro) + particles * (8 + 2 * Pb);
        // history particles (alternative formula, stated in DESIGN.md section 5 before it was measured): per particle the
        // weight read and written (16), the record -- 8 bytes + 4 per entry -- read once by the update, once as a
        // resample source, written once with its new entry (+4); the Dirichlet rows come from the shared tables
        if (P.hist) out[FBA_K_BELIEF_IS].bytes = particles * 44 + entries * 12;
        // packed tiger particles (64-byte records): the same formula on the bytes a packed particle has -- the update reads
        // state + weight + its two rows and writes state + weight + two counts (32 + Rt + Ro = 48), the resample reads the
        // weight and moves a record in and out (8 + 2 * 64)
        if (P.packed) out[FBA_K_BELIEF_IS].bytes = particles * (32 + Rt + Ro) + particles * (8 + 2 * (uint64_t)(P.Cs * 4));
    }
    return FBA_OK;
}

// diagnostic (scripts/tick_probe.py; not part of include/fba_hip.h): every slot's simulated-step counter and time-step
int fba_debug_slot_counters(fba_ctx* c, unsigned long long* sim_steps, int32_t* t)
{
    if (!c) return FBA_EINVAL;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (sim_steps) HIPCHK(c, hipMemcpy(sim_steps, c->D.sim_steps, (size_t)c->P.E * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (t) HIPCHK(c, hipMemcpy(t, c->D.t, (size_t)c->P.E * sizeof(int32_t), hipMemcpyDeviceToHost));
    return FBA_OK;
}

int fba_reset_kernel_times(fba_ctx* c)
{
    if (!c) return FBA_EINVAL;
    int rc = flush_events(c);
    if (rc) return rc;
    for (int k = 0; k < FBA_K_COUNT; ++k) { c->k_ms[k] = 0; c->k_launches[k] = 0; }
    c->base_sim       = sum_counter(c, c->D.sim_steps, c->P.E, &rc);
    c->base_attempts  = sum_counter(c, c->D.upd_attempts, c->P.E, &rc);
    c->base_particles = sum_counter(c, c->D.upd_particles, c->P.E, &rc);
    c->base_entries   = sum_counter(c, c->D.upd_entries, c->P.E, &rc);
    return rc;
}

int fba_trace_count(const fba_ctx* cc)
{
    fba_ctx* c = const_cast<fba_ctx*>(cc);
    if (!c) return FBA_EINVAL;
    int32_t n = 0;
    if (hipMemcpy(&n, c->D.trace_count, sizeof n, hipMemcpyDeviceToHost) != hipSuccess) return FBA_EHIP;
    return std::min(n, c->D.trace_cap);
}

int fba_get_trace(const fba_ctx* cc, fba_trace_rec* out, int32_t cap)
{
    fba_ctx* c = const_cast<fba_ctx*>(cc);
    if (!c || !out) return FBA_EINVAL;
    const int n = std::min(fba_trace_count(c), cap);
    if (n <= 0) return n;
    HIPCHK(c, hipMemcpy(out, c->D.trace, (size_t)n * sizeof(fba_trace_rec), hipMemcpyDeviceToHost));
    // slots finish their ticks in arbitrary order: present the records as the reference would
    // print them, by (run, episode, t)
    std::sort(out, out + n, [](const fba_trace_rec& a, const fba_trace_rec& b) {
        if (a.run != b.run) return a.run < b.run;
        if (a.episode != b.episode) return a.episode < b.episode;
        return a.t < b.t;
    });
    return n;
}

int fba_get_trace_hist(const fba_ctx* cc, uint32_t* out, int32_t cap)
{
    fba_ctx* c = const_cast<fba_ctx*>(cc);
    if (!c || !out) return FBA_EINVAL;
    if (!c->D.trace_hist) return fail(c, FBA_EINVAL, "no histograms were recorded: create the context with trace = 2 (domains of at most %d states)", FBA_TRACE_HIST_BINS);
    const int n = std::min(fba_trace_count(c), cap);
    if (n <= 0) return n;
    std::vector<fba_trace_rec> tr((size_t)n);
    std::vector<uint32_t> h((size_t)n * FBA_TRACE_HIST_BINS);
    HIPCHK(c, hipMemcpy(tr.data(), c->D.trace, (size_t)n * sizeof(fba_trace_rec), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(h.data(), c->D.trace_hist, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::vector<int> order((size_t)n);
    for (int i = 0; i < n; ++i) order[(size_t)i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) {   // the order of fba_get_trace
        if (tr[a].run != tr[b].run) return tr[a].run < tr[b].run;
        if (tr[a].episode != tr[b].episode) return tr[a].episode < tr[b].episode;
        return tr[a].t < tr[b].t;
    });
    for (int i = 0; i < n; ++i)
        std::copy(h.begin() + (size_t)order[(size_t)i] * FBA_TRACE_HIST_BINS, h.begin() + (size_t)(order[(size_t)i] + 1) * FBA_TRACE_HIST_BINS,
                  out + (size_t)i * FBA_TRACE_HIST_BINS);
    return n;
}

int fba_selftest_ucb(fba_ctx* c, const double* L, const int32_t* n, int32_t count, double u, double* out)
{
    if (!c || !L || !n || !out || count <= 0) return FBA_EINVAL;
    ScratchBuf<double> dL, dout;
    ScratchBuf<int32_t> dn;
    HIPCHK(c, dL.alloc((size_t)count));
    HIPCHK(c, dout.alloc((size_t)count));
    HIPCHK(c, dn.alloc((size_t)count));
    HIPCHK(c, hipMemcpy(dL.p, L, (size_t)count * 8, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(dn.p, n, (size_t)count * 4, hipMemcpyHostToDevice));
    launch_selftest_ucb(dL.p, dn.p, count, u, dout.p, c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, dout.p, (size_t)count * 8, hipMemcpyDeviceToHost));
    return FBA_OK;
}

int fba_selftest_lgamma(fba_ctx* c, const double* x, int32_t count, double* out)
{
    if (!c || !x || !out || count <= 0) return FBA_EINVAL;
    ScratchBuf<double> dx, dout;
    HIPCHK(c, dx.alloc((size_t)count));
    HIPCHK(c, dout.alloc((size_t)count));
    HIPCHK(c, hipMemcpy(dx.p, x, (size_t)count * 8, hipMemcpyHostToDevice));
    launch_selftest_lgamma(dx.p, count, dout.p, c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, dout.p, (size_t)count * 8, hipMemcpyDeviceToHost));
    return FBA_OK;
}

int fba_log_bd_score(fba_ctx* c, const float* counts, const float* prior, double* out)
{
    if (!c || !counts || !prior || !out) return FBA_EINVAL;
    if (c->P.model != FBA_MODEL_BA_FACTORED) return fail(c, FBA_EINVAL, "fba_log_bd_score: not a factored model");
    ScratchBuf<float> dc, dp;
    ScratchBuf<double> dout;
    const size_t n = (size_t)c->dense_C;
    HIPCHK(c, dc.alloc(n));
    HIPCHK(c, dp.alloc(n));
    HIPCHK(c, dout.alloc(1));
    HIPCHK(c, hipMemcpy(dc.p, counts, n * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(dp.p, prior, n * 4, hipMemcpyHostToDevice));
    launch_selftest_bd(c->P, dc.p, dp.p, dout.p, c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, dout.p, 8, hipMemcpyDeviceToHost));
    return FBA_OK;
}

// utils::Statistic (reference src/utils/Statistic.cpp:5-46)
void fba_stat_add(fba_stat* s, double v)
{
    s->count += 1;
    const double delta = v - s->mean;
    s->mean += delta / s->count;
    const double delta2 = v - s->mean;
    s->m2 += delta * delta2;
}
double fba_stat_var(const fba_stat* s) { return s->count < 2 ? 0 : s->m2 / (s->count - 1); }
double fba_stat_stder(const fba_stat* s) { return s->count < 2 ? 0 : std::sqrt(fba_stat_var(s) / s->count); }

}  // extern "C"
