// wavetree.hip -- a FLOOR for "one wave per tree" (DESIGN.md section 8.5): what ONE wavefront needs for the
// dependent chain of a UCT search on the episodic tiger POMDP when everything that can be taken off the chain is taken off it:
//   * the whole tree in LDS (no trip past the CU), counts and Q values as the engine keeps them (uint32 / fp64)
//   * uniform draws pre-generated in LDS (Philox costs nothing here: the floor assumes idle lanes made the blocks)
//   * the three UCB values in three lanes (fp64 sqrt and division as parity needs them, log from an LDS table), arg-max by DPP
//   * the back-up's levels in parallel lanes (one fp64 division deep)
//   * no statistics, no trace, no belief beyond one LDS read per simulation
// It is NOT the engine's search (no parity claim, no streams): it has the same dependent operations per tree level, per rollout step
// and per back-up, and nothing else, so a real wave-per-tree kernel cannot be faster than this on this chip.
// Usage (GPU box):  hipcc --offload-arch=gfx950 -O3 wavetree.hip -o wavetree && ./wavetree [sims] [reps]
// Prints one JSON line: microseconds per search of `sims` simulations, tree levels and rollout steps per simulation, cycles per simulation.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int MAXN = 256;      // nodes (the tiger tree of 1 024 simulations has ~50)
constexpr int NU = 2048;       // pre-generated uniforms (a ring)
constexpr int H = 10;          // horizon

struct Out { uint32_t n[3]; double q[3]; uint32_t nodes; uint64_t levels, rollout_steps; long long clocks; };

template <int CTRL> __device__ __forceinline__ double dpp_f64(double v)
{
    const uint64_t b = __double_as_longlong(v);
    const uint32_t lo = __builtin_amdgcn_mov_dpp((int)(uint32_t)b, CTRL, 0xf, 0xf, true);
    const uint32_t hi = __builtin_amdgcn_mov_dpp((int)(uint32_t)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((uint64_t)hi << 32) | lo);
}

__global__ void __launch_bounds__(64) wavetree_kernel(int sims, uint32_t seed, Out* out)
{
    __shared__ uint32_t Nn[MAXN];          // visits of a node
    __shared__ uint32_t na[MAXN][4];       // visits of its actions
    __shared__ double qa[MAXN][4];         // their Q values
    __shared__ uint16_t child[MAXN][2];    // child after (listen, observation)
    __shared__ float uni[NU];
    __shared__ double logtab[2048];
    __shared__ uint32_t p_node[H + 1];
    __shared__ uint32_t p_act[H + 1];
    __shared__ double p_ret[H + 1];
    const int lane = threadIdx.x;
    for (int i = lane; i < MAXN; i += 64) { Nn[i] = 0; child[i][0] = child[i][1] = 0; for (int a = 0; a < 4; ++a) { na[i][a] = 0; qa[i][a] = 0.0; } }
    uint32_t x = seed ^ (lane * 0x9e3779b9u);
    for (int i = lane; i < NU; i += 64) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; uni[i] = (float)(x >> 8) * (1.0f / 16777216.0f); }
    for (int i = lane; i < 2048; i += 64) logtab[i] = log1p((double)i);
    __syncthreads();
    uint32_t k = 0, nodes = 1;
    uint64_t levels = 0, rsteps = 0;
    const long long c0 = wall_clock64();
    for (int s = 0; s < sims; ++s) {
        const int state = __builtin_amdgcn_readfirstlane(uni[k++ & (NU - 1)] < 0.5f);
        uint32_t node = 0;
        int depth = 0;
        double tail = 0.0;      // discounted return below the tree part
        bool more = true;
        while (more) {
            // one tree level: three lanes, one action each
            const int a = lane & 3;
            const uint32_t n = na[node][a], N = Nn[node];
            const double q = qa[node][a];
            const double lg = logtab[N & 2047];
            double ucb = (n == 0 || a == 3) ? (a == 3 ? -1e300 : 1e300) : q + 100.0 * sqrt(lg / (double)n);
            // arg-max over the quad's first three lanes, ties to a uniform draw
            const double m1 = fmax(ucb, dpp_f64<0xB1>(ucb));       // quad_perm [1,0,3,2]
            const double best = fmax(m1, dpp_f64<0x4E>(m1));       // quad_perm [2,3,0,1]
            const uint32_t tie = (uint32_t)__builtin_amdgcn_ballot_w64(ucb == best) & 7u;
            int act = __builtin_ctz(tie);
            if (tie & (tie - 1)) { const int pick = (int)(uni[k++ & (NU - 1)] * (float)__builtin_popcount(tie)); uint32_t t = tie; for (int j = 0; j < pick; ++j) t &= t - 1; act = __builtin_ctz(t); }
            act = __builtin_amdgcn_readfirstlane(act);
            ++levels;
            double r;
            int obs = 0;
            if (act == 2) { r = -1.0; obs = __builtin_amdgcn_readfirstlane((uni[k++ & (NU - 1)] < 0.85f) ? state : 1 - state); }
            else { r = act == state ? -100.0 : 10.0; more = false; }
            if (lane == 0) { p_node[depth] = node; p_act[depth] = act; p_ret[depth] = r; }
            ++depth;
            if (more) {
                uint32_t c = __builtin_amdgcn_readfirstlane(child[node][obs]);
                if (depth >= H) more = false;
                else if (c == 0) {          // expand, then roll out with uniform actions
                    c = nodes < MAXN ? nodes++ : 0;
                    if (lane == 0) child[node][obs] = (uint16_t)c;
                    double disc = 1.0;
                    for (int d = depth; d < H; ++d) {
                        const int ra = __builtin_amdgcn_readfirstlane((int)(uni[k++ & (NU - 1)] * 3.0f));
                        ++rsteps;
                        if (ra == 2) { tail += disc * -1.0; k++; }
                        else { tail += disc * (ra == state ? -100.0 : 10.0); break; }
                        disc *= 0.95;
                    }
                    more = false;
                } else node = c;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // the path is in LDS (one wave: program order)
        // back-up: lane d owns level d; returns by a short serial scan first (depth <= H)
        double G = tail;
        double mine = 0.0;
        for (int d = depth - 1; d >= 0; --d) { G = p_ret[d] + 0.95 * G; if (lane == d) mine = G; }
        if (lane < depth) {
            const uint32_t nd = p_node[lane], a = p_act[lane];
            const uint32_t n = na[nd][a] + 1;
            const double q = qa[nd][a];
            na[nd][a] = n;
            qa[nd][a] = q + (mine - q) / (double)n;
            Nn[nd] += 1;
        }
    }
    const long long c1 = wall_clock64();
    if (lane == 0) {
        for (int a = 0; a < 3; ++a) { out->n[a] = na[0][a]; out->q[a] = qa[0][a]; }
        out->nodes = nodes; out->levels = levels; out->rollout_steps = rsteps; out->clocks = c1 - c0;
    }
}

int main(int argc, char** argv)
{
    const int sims = argc > 1 ? atoi(argv[1]) : 1024;
    const int reps = argc > 2 ? atoi(argv[2]) : 50;
    Out* d; CHK(hipMalloc(&d, sizeof(Out)));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) wavetree_kernel<<<1, 64>>>(sims, 17u + w, d);
    CHK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0.f;
    double clk = 0;
    Out h{};
    for (int r = 0; r < reps; ++r) {
        CHK(hipEventRecord(e0));
        wavetree_kernel<<<1, 64>>>(sims, 1000u + r, d);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best; sum += ms;
        CHK(hipMemcpy(&h, d, sizeof(Out), hipMemcpyDeviceToHost));
        clk += (double)h.clocks;
    }
    // wall_clock64 ticks at 100 MHz
    printf("{\"sims\": %d, \"reps\": %d, \"us_per_search_event_mean\": %.1f, \"us_per_search_event_best\": %.1f, \"us_per_search_loop_only\": %.1f, "
           "\"tree_levels_per_sim\": %.2f, \"rollout_steps_per_sim\": %.2f, \"nodes\": %u, \"root_n\": [%u, %u, %u], \"root_q\": [%.2f, %.2f, %.2f], "
           "\"ns_per_sim_loop_only\": %.0f}\n",
           sims, reps, 1e3 * sum / reps, 1e3 * best, clk / reps / 100.0, (double)h.levels / sims, (double)h.rollout_steps / sims, h.nodes,
           h.n[0], h.n[1], h.n[2], h.q[0], h.q[1], h.q[2], clk / reps / 100.0 * 1e3 / sims);
    return 0;
}
