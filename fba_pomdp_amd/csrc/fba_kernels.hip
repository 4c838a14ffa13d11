// fba_kernels.hip -- the belief, episode and bookkeeping kernels of the BA-POMCP engine (gfx950 / CDNA4, wave64).
// (The tree-search kernels live in fba_search.hip; helpers both share in fba_kernels_common.h.)
//
//   env_kernel           true-environment step + episode loop    (E1-E3, D1/D2)
//   reject_kernel        rejection-sampling belief update        (B4)
//   reinvigorate_kernel  structure reinvigoration (breed + replace) before the rejection update (8f-3)
//   importance_kernel    importance update + scan + resample     (B5, B6, B3)
//   reset_kernel         resetDomainStateDistribution            (B7)
//   init_kernel          Belief::initiate                        (B7)
//   flush_kernel         belief checksum + trace record
//
// None of this work is a dense contraction, so nothing here touches MFMA.  The belief kernels are
// HBM-streaming (one workgroup per slot, wave ballot / prefix-sum compaction, whole-record gathers).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "fba_kernels_common.h"


namespace fba {

// ---------------------------------------------------------------------------------------------
// Episode bookkeeping shared by start_kernel and env_kernel
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void begin_episode(const Problem& P, const DeviceState& D, int e)
{
    D.t[e] = 0;
    Rng g  = slot_rng(P, D, e);
    g.stream(FBA_PHASE_START, 0);
    D.env_state[e] = domain_start(P, g);  // Episode.cpp:32
    D.ret[e]  = 0;
    D.disc[e] = 1;
}

// Sets every slot at the beginning of its first run.
__global__ void start_kernel(Problem P, DeviceState D)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.E) return;
    const int run  = D.run_offset + e;
    const bool on  = D.runs_total < 0 || e < D.runs_total;
    D.run[e]       = run;
    D.episode[e]   = 0;
    D.active[e]    = on;
    D.need_update[e] = 0;
    D.adv[e]         = 0;
    D.need_init[e]   = on;
    D.need_reset[e]  = on && P.model != FBA_MODEL_POMDP;
    D.cur[e].update_count = -1;
    if (on) begin_episode(P, D, e);
}

// env_kernel: the body of episode::run's loop after selectAction (Episode.cpp:42-55) and the
// run / episode loops of the two experiments (PlanningExperiment.cpp:39-52,
// BAPOMDPExperiment.cpp:44-75).  One lane per slot.
__global__ void env_kernel(Problem P, DeviceState D, int32_t* n_active)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.E || !D.active[e]) return;
    if (D.search_done) {   // budgeted searches: only the slots whose search has just finished take their step
        if (!D.search_done[e]) return;
        D.search_done[e] = 0;
    }
    Rng g = slot_rng(P, D, e);
    g.stream(FBA_PHASE_ENV, 0);
    const int a = D.action[e];
    int s = D.env_state[e], o = 0;
    double r = 0;
    const bool term = domain_step(P, g, s, a, o, r);
    D.env_state[e]  = s;
    D.obs[e]        = o;
    D.env_steps[e] += 1;
    const double ret  = D.ret[e] + r * D.disc[e];  // Return::add (Return.cpp:6-9)
    D.ret[e]          = ret;
    D.disc[e]        *= P.gamma;
    const int t       = D.t[e];
    // belief.updateEstimation unless the step was terminal (Episode.cpp:47-50)
    D.need_update[e] = !term;
    if (D.trace_on) {
        fba_trace_rec& rec = D.cur[e];
        rec.run = D.run[e]; rec.episode = D.episode[e]; rec.t = t;
        rec.action = a; rec.state = s; rec.obs = o; rec.terminal = term;
        rec.reward = r; rec.update_count = -1; rec.weight_total = 0;
        rec.belief_hash = 1;  // "flush pending" marker, overwritten by flush_kernel
    }
    // the belief update of this tick still addresses its streams with (run, episode, t): the
    // position only moves in advance_kernel, after that update
    if (!term && t + 1 < P.horizon) {
        D.adv[e] = 1;
        return;
    }
    // episode over: record the return, then the next episode / run of this slot
    const int run = D.run[e], ep = D.episode[e];
    const long long slot_run = (long long)run - D.run_offset;
    if (D.runs_total >= 0) {
        D.returns[(size_t)slot_run * P.episodes + ep] = ret;
        D.lengths[(size_t)slot_run * P.episodes + ep] = t + 1;
    }
    D.ep_sums[3 * e + 0] += 1;
    D.ep_sums[3 * e + 1] += ret;
    D.ep_sums[3 * e + 2] += ret * ret;
    D.adv[e] = 2;
    (void)n_active;
}

// advance_kernel: moves slots whose episode ended to their next episode or run (after this
// tick's belief update has consumed the old stream position).
__global__ void advance_kernel(Problem P, DeviceState D, int32_t* n_active)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.E || !D.active[e]) return;
    const int adv = D.adv[e];
    D.adv[e]      = 0;
    if (adv == 1) { D.t[e] += 1; return; }
    if (adv != 2) return;
    int run = D.run[e], ep = D.episode[e] + 1;
    if (ep < P.episodes) {
        D.episode[e]    = ep;
        D.need_reset[e] = P.model != FBA_MODEL_POMDP;
    } else {
        run += P.E;  // slot e executes runs e, e + E, e + 2E, ...
        if (D.runs_total >= 0 && run - D.run_offset >= D.runs_total) {
            D.active[e] = 0;
            atomicSub(n_active, 1);
            return;
        }
        D.run[e]        = run;
        D.episode[e]    = 0;
        D.need_init[e]  = 1;
        D.need_reset[e] = P.model != FBA_MODEL_POMDP;
    }
    begin_episode(P, D, e);
}

// ---------------------------------------------------------------------------------------------
// Whole-record gather: the m particles listed in s_src[] (LDS) are copied from `src` records to
// consecutive `dst` records, applying the UpdateCounts "+1"s at the blob indices listed in s_inc[k][j], k < ninc,
// and, if s_state is given, overwriting the state word (index C) with the particle's new state.
// A record is C4 float4; a power-of-two group of lanes owns one record so consecutive lanes move
// consecutive 16-byte pieces: every wave instruction reads and writes whole contiguous records.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void bump(float4& v, int d, float add)
{
    v.x += (d == 0) ? add : 0.f;
    v.y += (d == 1) ? add : 0.f;
    v.z += (d == 2) ? add : 0.f;
    v.w += (d == 3) ? add : 0.f;
}
// the same "+1" on a packed record (PackedView): cell c lives in half (c & 1) of word c >> 1
__device__ __forceinline__ void bump_cell(float4& v, int cell, int lo, bool packed)
{
    if (!packed) { bump(v, cell - lo, 1.0f); return; }
    const int d = (cell >> 1) - lo;
    const uint32_t add = (cell & 1) ? 0x10000u : 1u;
    v.x = __uint_as_float(__float_as_uint(v.x) + ((d == 0) ? add : 0u));
    v.y = __uint_as_float(__float_as_uint(v.y) + ((d == 1) ? add : 0u));
    v.z = __uint_as_float(__float_as_uint(v.z) + ((d == 2) ? add : 0u));
    v.w = __uint_as_float(__float_as_uint(v.w) + ((d == 3) ? add : 0u));
}

// s_owner[j] (nullable) = the thread whose LDS columns (s_src, s_state, s_inc) describe output record j
__device__ __forceinline__ void gather_records(float* __restrict__ dst, const float* __restrict__ src, const int32_t* s_owner,
                                               const int32_t* s_src, const int32_t* s_inc, int ninc, int inc_stride,
                                               const int32_t* s_state, int m, int C4, int C, int group, int nthreads,
                                               bool packed = false)
{
    const int gid = threadIdx.x / group, part0 = threadIdx.x % group, ngroups = nthreads / group;
    if (C4 > group && C4 <= 4 * group) {
        // a record of several pieces per lane (the 3.5 KB collision-avoidance particles): a lane group moves TWO records at a time, the
        // pieces of both -- up to eight loads per lane -- in flight before the first is waited for.  One record at a time left a wave
        // with four loads in flight and every record's loads behind the previous record's stores (vmcnt counts both).
        for (int j0 = gid; j0 < m; j0 += 2 * ngroups) {
            float4 v[2][4];
            int t[2];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int j = min(j0 + rr * ngroups, m - 1);
                t[rr] = s_owner ? s_owner[j] : j;
                const float4* sp = reinterpret_cast<const float4*>(src) + (size_t)s_src[t[rr]] * C4;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[rr][q] = sp[min(part0 + q * group, C4 - 1)];
            }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int j = j0 + rr * ngroups;
                if (j >= m) continue;
                float4* dp = reinterpret_cast<float4*>(dst) + (size_t)j * C4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int pq = part0 + q * group;
                    if (pq >= C4) continue;
                    const int lo = pq * 4;
                    for (int k = 0; k < ninc; ++k) bump_cell(v[rr][q], s_inc[k * inc_stride + t[rr]], lo, packed);
                    if (s_state) {  // new domain state in word C
                        const int d = C - lo;
                        const float f = __int_as_float(s_state[t[rr]]);
                        if (d == 0) v[rr][q].x = f; else if (d == 1) v[rr][q].y = f; else if (d == 2) v[rr][q].z = f; else if (d == 3) v[rr][q].w = f;
                    }
                    dp[pq] = v[rr][q];
                }
            }
        }
        return;
    }
    for (int j = gid; j < m; j += ngroups) {
        const int t      = s_owner ? s_owner[j] : j;
        const float4* sp = reinterpret_cast<const float4*>(src) + (size_t)s_src[t] * C4;
        float4* dp       = reinterpret_cast<float4*>(dst) + (size_t)j * C4;
        if (C4 > group) {
            // (records of more than four pieces per lane: the dense gridworld particles) four loads in flight per lane before the first
            // is waited for -- piece by piece, every load also waits for the store before it (vmcnt counts both)
            for (int part = part0; part < C4; part += 4 * group) {
                float4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = sp[min(part + q * group, C4 - 1)];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int pq = part + q * group;
                    if (pq >= C4) continue;
                    const int lo = pq * 4;
                    for (int k = 0; k < ninc; ++k) bump_cell(v[q], s_inc[k * inc_stride + t], lo, packed);
                    if (s_state) {  // new domain state in word C
                        const int d = C - lo;
                        const float f = __int_as_float(s_state[t]);
                        if (d == 0) v[q].x = f; else if (d == 1) v[q].y = f; else if (d == 2) v[q].z = f; else if (d == 3) v[q].w = f;
                    }
                    dp[pq] = v[q];
                }
            }
            continue;
        }
        for (int part = part0; part < C4; part += group) {
            float4 v = sp[part];
            const int lo = part * 4;
            for (int k = 0; k < ninc; ++k) bump_cell(v, s_inc[k * inc_stride + t], lo, packed);
            if (s_state) {  // new domain state in word C
                const int d = C - lo;
                const float f = __int_as_float(s_state[t]);
                if (d == 0) v.x = f; else if (d == 1) v.y = f; else if (d == 2) v.z = f; else if (d == 3) v.w = f;
            }
            dp[part] = v;
        }
    }
}

// The importance filters' gather: output record j is source record s_src[j] with that SOURCE particle's
// pending update applied -- new state and +1s from its row of the side array (written by the update
// pass, which leaves the source buffer untouched: no partial-line write-backs).
__device__ __forceinline__ void gather_records_side(float* __restrict__ dst, const float* __restrict__ src, const int32_t* s_src,
                                                    const int32_t* __restrict__ side, int side_w, int m, int C4, int C, int group,
                                                    int nthreads, bool packed = false)
{
    const int gid = threadIdx.x / group, part0 = threadIdx.x % group, ngroups = nthreads / group;
    const int ninc = side_w - 1;
    for (int j = gid; j < m; j += ngroups) {
        const int p      = s_src[j];
        const float4* sp = reinterpret_cast<const float4*>(src) + (size_t)p * C4;
        float4* dp       = reinterpret_cast<float4*>(dst) + (size_t)j * C4;
        const int32_t* sd = side + (size_t)p * side_w;
        const float nstate = __int_as_float(sd[0]);
        for (int part = part0; part < C4; part += group) {
            float4 v = sp[part];
            const int lo = part * 4;
            for (int k = 0; k < ninc; ++k) bump_cell(v, sd[1 + k], lo, packed);
            const int d = C - lo;
            if (d == 0) v.x = nstate; else if (d == 1) v.y = nstate; else if (d == 2) v.z = nstate; else if (d == 3) v.w = nstate;
            dp[part] = v;
        }
    }
}

// History particles (fba_device.h): output record j = source record s_src[j] -- its state replaced, its structure
// bits kept -- with the source particle's pending entry (side row {new state, entry}) inserted as word `ins`, the
// end of its action's group: the `len` entries behind it move up by one.  A power-of-two group of lanes moves
// one record in 16-byte pieces.
template <class IDX>
__device__ __forceinline__ void gather_hist_records(float* __restrict__ dst, const float* __restrict__ src, const IDX* s_src,
                                                    const int32_t* __restrict__ side, int len, int ins, int m, int C4, int group, int nthreads,
                                                    int C4d = 0)   // C4 / C4d: 16-byte pieces between source / destination records (hist_stride)
{
    if (C4d == 0) C4d = C4;
    const int gid = threadIdx.x / group, part0 = threadIdx.x % group, ngroups = nthreads / group;
    const int n4 = (len + 6) >> 2;  // pieces that hold words 0 .. 2 + len
    if (n4 <= group && m >= 4 * ngroups) {
        // a whole filter at once (every source already drawn): one piece per lane, four records in flight per lane group
        constexpr int FLY = 4;
        const int part = min(part0, n4 - 1);
        for (int j0 = gid; j0 < m; j0 += ngroups * FLY) {
            uint4 cur[FLY];
            uint32_t before[FLY];
            int2 sd[FLY];
#pragma unroll
            for (int q = 0; q < FLY; ++q) {
                const int j     = min(j0 + q * ngroups, m - 1);
                const int p     = (int)s_src[j];
                const uint4* sp = reinterpret_cast<const uint4*>(src) + (size_t)p * C4;
                cur[q]    = sp[part];
                before[q] = part > 0 ? sp[part - 1].w : 0u;
                sd[q]     = *reinterpret_cast<const int2*>(side + (size_t)p * 2);
            }
#pragma unroll
            for (int q = 0; q < FLY; ++q) {
                const int j = j0 + q * ngroups;
                if (j >= m || part0 >= n4) continue;
                const int w0 = part * 4;
                uint4 v;
                v.x = w0 + 0 < ins ? cur[q].x : (w0 + 0 == ins ? (uint32_t)sd[q].y : before[q]);
                v.y = w0 + 1 < ins ? cur[q].y : (w0 + 1 == ins ? (uint32_t)sd[q].y : cur[q].x);
                v.z = w0 + 2 < ins ? cur[q].z : (w0 + 2 == ins ? (uint32_t)sd[q].y : cur[q].y);
                v.w = w0 + 3 < ins ? cur[q].w : (w0 + 3 == ins ? (uint32_t)sd[q].y : cur[q].z);
                if (part == 0) {  // the new state, as an index and as hist_pack (= the step's s')
                    v.x = (uint32_t)sd[q].x;
                    v.y = (v.y & 0xffffu) | ((((uint32_t)sd[q].y >> 10) & 0x3ffu) << 16);
                }
                (reinterpret_cast<uint4*>(dst) + (size_t)j * C4d)[part] = v;
            }
        }
        return;
    }
    for (int j = gid; j < m; j += ngroups) {
        const int p      = s_src[j];
        const uint4* sp  = reinterpret_cast<const uint4*>(src) + (size_t)p * C4;
        uint4* dp        = reinterpret_cast<uint4*>(dst) + (size_t)j * C4d;
        const int2 sd    = *reinterpret_cast<const int2*>(side + (size_t)p * 2);
        for (int part = part0; part < n4; part += group) {
            const uint4 cur = sp[part];
            const uint32_t before = part > 0 ? sp[part - 1].w : 0u;
            const int w0 = part * 4;
            uint4 v;
            v.x = w0 + 0 < ins ? cur.x : (w0 + 0 == ins ? (uint32_t)sd.y : before);
            v.y = w0 + 1 < ins ? cur.y : (w0 + 1 == ins ? (uint32_t)sd.y : cur.x);
            v.z = w0 + 2 < ins ? cur.z : (w0 + 2 == ins ? (uint32_t)sd.y : cur.y);
            v.w = w0 + 3 < ins ? cur.w : (w0 + 3 == ins ? (uint32_t)sd.y : cur.z);
            if (part == 0) {  // the new state, as an index and as hist_pack (= the step's s')
                v.x = (uint32_t)sd.x;
                v.y = (v.y & 0xffffu) | ((((uint32_t)sd.y >> 10) & 0x3ffu) << 16);
            }
            dp[part] = v;
        }
    }
}

// Deferring the increments to the gather pays when a record is a line or two (the update pass then dirties
// nothing); for records of many lines the in-place update touches only the FS + FO lines it increments and the
// gather stays a plain copy, which is cheaper than testing every 16-byte piece against every pending cell.
__device__ __forceinline__ bool defer_increments(const Problem& P) { return P.Cs <= 64; }

__device__ __forceinline__ int record_group(int C4)
{
    int g = 1;
    while (g < C4 && g < 64) g <<= 1;
    return g;
}

// ---------------------------------------------------------------------------------------------
// reject_kernel: beliefs::rejectSample (RejectionSampling.hpp:26-72), one workgroup per slot.
// Attempt k is an independent Philox stream, so a chunk of 256 attempts runs in parallel; the
// accepted ones are compacted in attempt order with a wave ballot + prefix sum, which keeps the
// reference's result: the new filter is the first N accepted attempts, in order, and the
// reported loop count is the index of the N-th accepted attempt + 1.
// ---------------------------------------------------------------------------------------------
// BLK = attempts per chunk = workgroup size.  Measured on the bench workload: 256 and 512 give the same time,
// 1024 is 10 % slower (a larger last chunk runs attempts nobody needs), and halving the resident workgroups
// per CU changes nothing -- the kernel is bound by the memory system's rate for random 128-byte lines.
// `fc` = 1 runs the update on the reinvigoration belief's fully connected filter (launched before
// the main filter's update, which is the one that clears the request flag).
template <bool REG, int TIGER_TABLE, int FTIGER = 0, int BLK = REJECT_BLOCK, bool FTP = false>
__global__ void __launch_bounds__(BLK) reject_kernel(Problem P, DeviceState D, int fc)
{
    if (FTIGER > 0) {  // factored tiger with FTIGER binary state features (see search_kernel)
        P.model = FBA_MODEL_BA_FACTORED;
        P.S = 1 << FTIGER; P.A = 3; P.O = 2;
        if (P.domain != FBA_DOM_FTIGER_CONTINUOUS) P.domain = FBA_DOM_FTIGER_EPISODIC;
    }
    if (TIGER_TABLE) {  // sizes restated as literals (see search_kernel)
        fc = 0;
        P.model = FBA_MODEL_BA_TABLE;
        P.S = 2; P.A = 3; P.O = 2; P.phi_len = 12; P.C = 24; P.Cs = 32;
        if (TIGER_TABLE == 2) { P.C = 12; P.Cs = 16; }  // packed particles (PackedView)
        if (P.domain != FBA_DOM_TIGER_CONTINUOUS) P.domain = FBA_DOM_TIGER_EPISODIC;
    }
    __shared__ __attribute__((aligned(8))) float s_prior[TIGER_TABLE == 2 ? 24 : 2];
    __shared__ int32_t s_src[BLK], s_ns[BLK], s_owner[BLK], s_inc[MAXINC * BLK];
    __shared__ int32_t s_wave[BLK / 64];
    __shared__ int32_t s_count;
    const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (!D.need_update[e]) return;
    if (TIGER_TABLE == 2) {
        if (tid < 24) s_prior[tid] = D.prior_dense[tid];
        __syncthreads();
    }
    const int a = D.action[e], o = D.obs[e], N = P.N;
    uint8_t* bufsel = fc ? D.bufsel_fc : D.bufsel;
    float* recs     = fc ? D.p_rec_fc : D.p_rec;
    const int cur = bufsel[e];
    const size_t sb = pbase(P, e, cur), db = pbase(P, e, cur ^ 1);
    const float* scn   = recs + sb * (size_t)P.Cs;
    float* dcn         = recs + db * (size_t)P.Cs;
    const int C4 = P.Cs / 4, group = record_group(C4);
    const int ninc = model_ninc(P);
    const uint32_t phase = fc ? FBA_PHASE_REJECT_FC : FBA_PHASE_REJECT;
    const bool lazy = slot_lazy(D, e);
    Rng g = slot_rng(P, D, e);

    int acc = 0, base = 0;
    while (acc < N) {
        if (base >= REJECT_MAX_ATTEMPTS) {
            // The reference would spin forever here (no particle can produce the observation); a GPU must
            // not.  Park the slot and tell the host, which turns it into an error.
            if (tid == 0) {
                atomicCAS(D.fault, 0, 1 + e);
                D.need_update[e] = 0;
                D.active[e]      = 0;
            }
            return;
        }
        const int k = base + tid;
        g.stream(phase, (uint32_t)k);
        const int src = P.point ? 0 : g.uniform_int(N);         // FlatFilter::sample; the point estimate copies its one state
        const float* rec = scn + (size_t)src * P.Cs;
        int s = (!fc && lazy) ? lazy_state(P, D, e, src) : rec_state(rec, P.C), so;
        double r;
        // UpdateCounts: the +1s land in the copy
        if (FTIGER > 0 && FTP) ftiger_step_packed<(FTIGER > 0 ? FTIGER : 1)>(P, g, GlobalView{rec}, s, a, so, r, LdsInc<BLK>{s_inc + tid});
        else if (FTIGER > 0) ftiger_step<(FTIGER > 0 ? FTIGER : 1)>(P, g, GlobalView{rec}, s, a, so, r, LdsInc<BLK>{s_inc + tid});
        else if (TIGER_TABLE == 2) tiger_step_packed(P, g, [&](int w) { return __float_as_uint(rec[w]); }, s_prior, s, a, so, r, LdsInc<BLK>{s_inc + tid});
        else sim_step<REG>(P, g, GlobalView{rec}, s, a, so, r, LdsInc<BLK>{s_inc + tid});
        s_src[tid] = src;
        s_ns[tid]  = s;
        const bool ok = (so == o);
        const unsigned long long ballot = __ballot(ok);
        const int prefix = __popcll(ballot & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = __popcll(ballot);
        __syncthreads();
        int woff = 0, chunk = 0;
        for (int w = 0; w < BLK / 64; ++w) {
            if (w < wave) woff += s_wave[w];
            chunk += s_wave[w];
        }
        const int j = woff + prefix;  // position among this chunk's accepted attempts
        if (ok && acc + j < N) {
            s_owner[j] = tid;
            if (acc + j == N - 1) s_count = k + 1;
        }
        __syncthreads();
        const int m = min(chunk, N - acc);
        gather_records(dcn + (size_t)acc * P.Cs, scn, s_owner, s_src, s_inc, ninc, BLK, s_ns, m, C4, P.C, group, BLK, TIGER_TABLE == 2 || FTP);
        acc += m;
        base += BLK;
        __syncthreads();
    }
    if (tid == 0) {
        bufsel[e] = cur ^ 1;
        D.belief_steps[e] += (unsigned long long)s_count;
        D.upd_attempts[e] += (unsigned long long)s_count;
        D.upd_particles[e] += (unsigned long long)N;
        if (!fc) {
            D.need_update[e]      = 0;
            D.lazy_reset[e]       = 0;  // every record of the new buffer carries its real state
            D.cur[e].update_count = s_count;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// reject_tiger_lds_kernel: the same rejectSample for packed tabular-tiger particles with N <= 4096, arranged so
// that the attempts never touch HBM.  For the slot's action a, a step from particle i reads only three words
// of its record -- the (s, a, .) transition pair, the (a, 0, .) and (a, 1, .) observation pairs -- and its state:
//   1. one sequential pass parks those 12 bytes + the state of every particle in LDS (48 + 4 KB);
//   2. attempt k = stream k, as in reject_kernel, samples its source from LDS; the accepted attempts are
//      appended in attempt order to a 16-bit list (source, old state, new state), until N are accepted;
//   3. one pass copies the listed source records to the new filter with their two "+1"s (cells
//      T(s, a, s') and O(a, s', o)) and the new state -- N independent 64-byte copies, no synchronisation.
// Same draws, same order, same result as reject_kernel<false, 2>; two thirds fewer random memory requests.
// ---------------------------------------------------------------------------------------------
struct TigerRowView {   // the counts one step of action a from state s can read, out of the three parked words
    uint32_t wt, wo0, wo1;
    int a;
    const float* prior;
    static constexpr bool row_regs = false;
    __device__ __forceinline__ float at(int k) const
    {
        // (masks, not a chain of ?: -- the compiler turns that into an indexed private array, i.e. scratch)
        const int q = k - 12 - a * 4;
        const uint32_t mt = (uint32_t) - (int)(k < 12), m1 = ~mt & (uint32_t) - (int)(q >= 2), m0 = ~(mt | m1);
        const uint32_t w = (wt & mt) | (wo0 & m0) | (wo1 & m1);
        return prior[k] + (float)((k & 1) ? (w >> 16) : (w & 0xffffu));
    }
};
template <int BLK>
__global__ void __launch_bounds__(BLK) reject_tiger_lds_kernel(Problem P, DeviceState D)
{
    P.model = FBA_MODEL_BA_TABLE;
    P.S = 2; P.A = 3; P.O = 2; P.phi_len = 12; P.C = 12; P.Cs = 16;
    if (P.domain != FBA_DOM_TIGER_CONTINUOUS) P.domain = FBA_DOM_TIGER_EPISODIC;
    extern __shared__ uint32_t s_tab[];  // [N][3] words, then [N] state bytes
    __shared__ uint16_t s_acc[TIGER_LDS_MAX_N];
    __shared__ __attribute__((aligned(8))) float s_prior[24];
    __shared__ int32_t s_wave[BLK / 64];
    __shared__ int32_t s_count;
    const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (!D.need_update[e]) return;
    const int a = D.action[e], o = D.obs[e], N = P.N;
    const int cur = D.bufsel[e];
    const size_t sb = pbase(P, e, cur), db = pbase(P, e, cur ^ 1);
    const float* scn = D.p_rec + sb * (size_t)P.Cs;
    float* dcn       = D.p_rec + db * (size_t)P.Cs;
    uint8_t* s_st    = reinterpret_cast<uint8_t*>(s_tab + 3 * (size_t)N);
    if (tid < 24) s_prior[tid] = D.prior_dense[tid];
    const bool lazy = slot_lazy(D, e);
    for (int i = tid; i < N; i += BLK) {
        const uint32_t* rec = reinterpret_cast<const uint32_t*>(scn) + (size_t)i * 16;
        const uint32_t w0 = rec[a], w1 = rec[3 + a], wo0 = rec[6 + 2 * a], wo1 = rec[7 + 2 * a];  // cell pairs (0,a,.) (1,a,.) | (a,0,.) (a,1,.)
        const int st = lazy ? lazy_state(P, D, e, i) : (int)rec[12];
        s_tab[3 * i + 0] = st ? w1 : w0;
        s_tab[3 * i + 1] = wo0;
        s_tab[3 * i + 2] = wo1;
        s_st[i]          = (uint8_t)st;
    }
    __syncthreads();
    Rng g = slot_rng(P, D, e);
    int acc = 0, base = 0;
    while (acc < N) {
        if (base >= REJECT_MAX_ATTEMPTS) {  // see reject_kernel
            if (tid == 0) {
                atomicCAS(D.fault, 0, 1 + e);
                D.need_update[e] = 0;
                D.active[e]      = 0;
            }
            return;
        }
        const int k = base + tid;
        g.stream(FBA_PHASE_REJECT, (uint32_t)k);
        const int src = P.point ? 0 : g.uniform_int(N);         // FlatFilter::sample
        const int s0  = s_st[src];
        int s = s0, so;
        double r;
        {
            // the three parked words ARE the rows a step of action a from this particle can read: T(s, a, .), O(a, 0, .), O(a, 1, .)
            const uint32_t wt = s_tab[3 * src], wo0 = s_tab[3 * src + 1], wo1 = s_tab[3 * src + 2];
            tiger_step_packed(P, g, [&](int w) { return w < 6 ? wt : (w & 1 ? wo1 : wo0); }, s_prior, s, a, so, r, NoInc{});
        }
        const bool ok = (so == o);
        const unsigned long long ballot = __ballot(ok);
        const int prefix = __popcll(ballot & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = __popcll(ballot);
        __syncthreads();
        int woff = 0, chunk = 0;
        for (int w = 0; w < BLK / 64; ++w) {
            if (w < wave) woff += s_wave[w];
            chunk += s_wave[w];
        }
        const int j = acc + woff + prefix;  // position among all accepted attempts
        if (ok && j < N) {
            s_acc[j] = (uint16_t)(src | (s0 << 12) | (s << 13));
            if (j == N - 1) s_count = k + 1;
        }
        acc += min(chunk, N - acc);
        base += BLK;
        __syncthreads();
    }
    // the new filter: record j = source record with T(s, a, s') and O(a, s', o) bumped and the new state
    const int part = tid & 3;
    constexpr int GROUPS = BLK / 4, UNROLL = 4;  // four independent copies in flight per thread
    for (int j0 = tid >> 2; j0 < N; j0 += GROUPS * UNROLL) {
        uint32_t en[UNROLL];
        float4 v[UNROLL];
#pragma unroll
        for (int q = 0; q < UNROLL; ++q) {
            const int j = j0 + q * GROUPS;
            en[q] = j < N ? s_acc[j] : 0u;
            if (j < N) v[q] = reinterpret_cast<const float4*>(scn)[(size_t)(en[q] & 4095) * 4 + part];
        }
#pragma unroll
        for (int q = 0; q < UNROLL; ++q) {
            const int j = j0 + q * GROUPS;
            if (j >= N) continue;
            const int s0 = (en[q] >> 12) & 1, ns = (en[q] >> 13) & 1;
            bump_cell(v[q], s0 * 6 + a * 2 + ns, part * 4, true);
            bump_cell(v[q], 12 + a * 4 + ns * 2 + o, part * 4, true);
            if (part == 3) v[q].x = __int_as_float(ns);  // the state word (12)
            reinterpret_cast<float4*>(dcn)[(size_t)j * 4 + part] = v[q];
        }
    }
    if (tid == 0) {
        D.bufsel[e] = cur ^ 1;
        D.belief_steps[e] += (unsigned long long)s_count;
        D.upd_attempts[e] += (unsigned long long)s_count;
        D.upd_particles[e] += (unsigned long long)N;
        D.need_update[e]      = 0;
        D.lazy_reset[e]       = 0;  // every record of the new buffer carries its real state
        D.cur[e].update_count = s_count;
    }
}

// ---------------------------------------------------------------------------------------------
// reinvigorate_kernel: ReinvigoratingRejectionSampling::reinvigorateParticles
// (ReinvigoratingRejectionSampling.cpp:121-131), one workgroup per slot, before the two rejection
// updates.  Iteration i breeds one particle (breed, :24-35): the structure of a random particle
// of the main filter with one random edge flipped (FactoredTigerFactoredPrior::mutate
// FactoredTigerPriors.cpp:353-381 -> BABNModel::Structure::flip_random_edge BABNModel.cpp:16-31),
// the counts of a random particle of the fully connected filter marginalised onto that structure
// (BABNModel::marginalizeOut BABNModel.cpp:205-229, DBNNode::marginalizeOut DBNNode.cpp:40-80:
// source rows added in ascending order, in float), the domain state of the structure particle;
// it replaces a random particle of the main filter (FlatFilter::replace FlatFilter.cpp:39-46).
// Iterations are sequential -- a bred particle may be picked by the next one -- and R is small,
// so the draws are made by one thread and the workgroup only shares the record copy.
// Draw order within stream (REINVIG, i): fully connected pick, main pick, edge, victim (g++
// evaluates breed's arguments right to left).
// ---------------------------------------------------------------------------------------------
// mode 1, 2: the incubator belief (StructureIncubatorSampling.cpp).  The bred particle goes into the weighted SHADOW filter:
// mode 1 = reinvigorateShadowBelief (:137-153), P.incub particles, the victims are WeightedFilter::leastLikely of the
// shadow's weights -- uniform whenever this runs, so a fixed list (D.inc_order, computed by the host with the same
// std::priority_queue) -- and WeightedFilter::replace(i, s) gives the weight total / N and adjusts the total
// (WeightedFilter.cpp:70-87); mode 2 = initiate (:81-89), N particles, stream (INIT_SH, i), weight 1 / N.
__global__ void __launch_bounds__(256) reinvigorate_kernel(Problem P, DeviceState D, int mode)
{
    __shared__ int32_t s_fc, s_victim, s_state;
    __shared__ uint32_t s_mask[128];  // one parent mask per variable node (collision avoidance: A * obstacles <= 18; sysadmin: 2N * N <= 128)
    const int e = blockIdx.x, tid = threadIdx.x;
    if (mode == 2 ? !D.need_init[e] : !D.need_update[e]) return;
    const FDesc* fd = P.fd;
    float* recs      = D.p_rec + pbase(P, e, D.bufsel[e]) * (size_t)P.Cs;
    const float* fcs = D.p_rec_fc + pbase(P, e, D.bufsel_fc[e]) * (size_t)P.Cs;
    if (mode == 2 && tid == 0) D.bufsel_sh[e] = 0;
    const size_t shb = mode ? pbase(P, e, mode == 2 ? 0 : D.bufsel_sh[e]) : 0;
    float* dsts      = mode ? D.p_rec_sh + shb * (size_t)P.Cs : recs;
    double sh_total  = D.uni_total;   // the shadow's total weight after a resample / initiate (device order)
    const int nnodes = P.A * (fd->FS + fd->FO);
    const int rounds = mode == 0 ? P.reinvig : (mode == 1 ? P.incub : P.N);
    Rng g = slot_rng(P, D, e);
    if (mode == 2) g.position((uint32_t)D.run[e], 0, 0);
    for (int i = 0; i < rounds; ++i) {
        if (tid == 0) {
            g.stream(mode == 2 ? FBA_PHASE_INIT_SH : FBA_PHASE_REINVIG, (uint32_t)i);
            s_fc = g.uniform_int(P.N);
            const volatile float* srec = recs + (size_t)g.uniform_int(P.N) * P.Cs;  // may have been written an iteration ago
            for (int k = 0; k < fd->nvar; ++k) s_mask[k] = __float_as_uint(srec[fd->ncounts + k]);
            if (dom_is_sys(P.domain)) {
                // SysAdminFactoredPrior::mutate (:47-55): flip_random_edge(&T[action()][computer()], N); under the
                // reference's --std=c++11 g++ evaluates the second subscript first: computer, action, then the edge
                const int mc = g.uniform_int(P.sys->N), ma = g.uniform_int(P.A);
                s_mask[ma * P.sys->N + mc] ^= 1u << g.slow_int(0, fd->FS);
            } else if (dom_is_ca(P.domain)) {  // CollisionAvoidanceFactoredPrior::mutate :455-488: action, obstacle, then the edge
                const int ma = g.uniform_int(P.A), mo = g.uniform_int(P.ca->n);
                s_mask[ma * P.ca->n + mo] ^= 1u << g.slow_int(0, fd->FS);
            } else {                    // FactoredTigerFactoredPrior::mutate: the listen observation node
                s_mask[0] ^= 1u << g.slow_int(0, fd->FS);
            }
            s_state  = __float_as_int(srec[P.C]);
            if (mode == 0) s_victim = g.uniform_int(P.N);
            else {
                s_victim = mode == 1 ? D.inc_order[i] : i;
                double* w = D.p_weight_sh + shb;
                if (mode == 2) w[s_victim] = 1.0 / (double)P.N;
                else {
                    const double nw = sh_total / (double)P.N;
                    sh_total += nw - w[s_victim];
                    w[s_victim] = nw;
                }
            }
        }
        __syncthreads();
        const float* src = fcs + (size_t)s_fc * P.Cs;
        float* dst       = dsts + (size_t)s_victim * P.Cs;
        for (int w = tid; w < P.C; w += 256) dst[w] = src[w];  // nodes with fixed parents: the counts particle's CPTs
        __syncthreads();
        for (int k = 0; k < nnodes; ++k) {
            const FNode& nd = fd->nodes[k];
            if (nd.var < 0) continue;
            const uint32_t om = __float_as_uint(src[fd->ncounts + nd.var]), nm = s_mask[nd.var];
            int rows_max = 1, rows_old = 1, rows_new = 1;
            for (int j = 0; j < nd.nmax; ++j) {
                const int sz = fd->Ssz[nd.maxp[j]];
                rows_max *= sz;
                if ((om >> j) & 1u) rows_old *= sz;
                if ((nm >> j) & 1u) rows_new *= sz;
            }
            for (int idx = tid; idx < rows_max * nd.out; idx += 256) {
                const int row = idx / nd.out, v = idx - row * nd.out;
                float acc = 0.f;
                if (row < rows_new)
                    for (int r = 0; r < rows_old; ++r) {  // ascending source rows, as the reference adds them
                        int rem = r, nr = 0, mul = 1;
                        for (int j = nd.nmax - 1; j >= 0; --j) {
                            const int sz = fd->Ssz[nd.maxp[j]];
                            if ((om >> j) & 1u) {
                                const int digit = rem % sz;
                                rem /= sz;
                                if ((nm >> j) & 1u) { nr += digit * mul; mul *= sz; }
                            }
                        }
                        if (nr == row) acc += src[nd.off + r * nd.out + v];
                    }
                dst[nd.off + idx] = acc;
            }
            if (tid == 0) dst[fd->ncounts + nd.var] = __uint_as_float(nm);
        }
        if (tid == 0) rec_set_state(dst, P.C, s_state);
        __threadfence_block();
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// cheat_kernel: CheatingReinvigoration::cheat (CheatingReinvigoration.cpp:130-143), after the importance
// update of a slot whose log likelihood fell below the threshold: `cheat` times, a random particle of
// the correct-graph filter is copied over a random particle of the weighted filter (which keeps its
// weight).  Draw order within stream (REINVIG, k): the correct filter's sample, then slowRandomInt for
// the victim (g++ evaluates replace's arguments right to left).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) cheat_kernel(Problem P, DeviceState D)
{
    __shared__ int32_t s_src, s_victim;
    const int e = blockIdx.x, tid = threadIdx.x;
    if (!D.cheat_pending[e]) return;
    float* recs      = D.p_rec + pbase(P, e, D.bufsel[e]) * (size_t)P.Cs;
    const float* aux = D.p_rec_fc + pbase(P, e, D.bufsel_fc[e]) * (size_t)P.Cs;
    Rng g = slot_rng(P, D, e);
    for (int k = 0; k < P.cheat; ++k) {
        if (tid == 0) {
            g.stream(FBA_PHASE_REINVIG, (uint32_t)k);
            s_src    = g.uniform_int(P.N);
            s_victim = g.slow_int(0, P.N);
        }
        __syncthreads();
        const float* src = aux + (size_t)s_src * P.Cs;
        float* dst       = recs + (size_t)s_victim * P.Cs;
        for (int w = tid; w <= P.C; w += 256) dst[w] = src[w];  // counts and the state word
        __threadfence_block();
        __syncthreads();
    }
    if (tid == 0) D.cheat_pending[e] = 0;
}


// ---------------------------------------------------------------------------------------------
// mh_kernel: the rest of MHwithinGibbs::updateEstimation (MHwithinGibbs.cpp:316-332) after the importance update and
// resample: record (a, o) in the run's history, add log(likelihood), and when the sum falls below --threshold re-draw
// the whole filter (reinvigorate, :334-395): a Metropolis-Hastings chain over structures -- mutate, prior of the
// structure (computePriorModel, FactoredTigerPriors.cpp:293-321), posterior counts along a sampled state history
// (computePosteriorCounts :397-436), LogBDScore, accept if log(u) < score difference -- inside a Gibbs loop that
// re-samples the state history (msgSampleStateHistory :96-213 or rejectionSampleStateHistory :38-94) after every
// accepted model.
// One WAVE runs one slot's chain.  The chain is sequential by definition (every accept decides what the next proposal is
// conditioned on), so all 64 lanes carry the same Rng and walk the same control flow -- the draws (stream (REINVIG, 0)) and
// every floating-point sum keep the order of a one-lane chain, which is the checker's -- and share out the work BETWEEN
// two draws: record copies, the nodes of computePriorModel, the per-node increments along the history, the (a, s) rows of
// flattenT/O, the states of a message, the lgamma terms of LogBDScore (evaluated side by side, then added up in CPT
// order).  The chain's scratch (three count blobs, T and O, messages, the state sequence) lives in LDS when it fits in
// 64 KB, in HBM otherwise; a block is one wave, so MH_SYNC costs a wait, not a rendezvous.
// ---------------------------------------------------------------------------------------------
#define MH_SYNC() __syncthreads()
struct MhScratch {
    float *prior, *model, *fresh, *T, *O;
    double *msg, *probs;
    int32_t* seq;
    double* terms;
    uint32_t *masks, *nmasks;
    int32_t* nterm;
    int stride;  // longest Dirichlet row + 1
};
__device__ __forceinline__ MhScratch mh_scratch(const Problem& P, float* base, char* head)
{
    MhScratch m;
    m.prior = base; m.model = base + P.Cs; m.fresh = base + 2 * P.Cs;
    m.T = base + 3 * P.Cs;
    m.O = m.T + P.S * P.A * P.S;
    m.msg   = reinterpret_cast<double*>(base + ((3 * P.Cs + P.S * P.A * P.S + P.A * P.S * P.O + 1) & ~1));
    m.probs = m.msg + (size_t)(P.horizon + 1) * P.S;
    m.seq   = reinterpret_cast<int32_t*>(m.probs + P.S);
    m.terms  = reinterpret_cast<double*>(head);
    m.masks  = reinterpret_cast<uint32_t*>(head + MH_TERMS * 8);
    m.nmasks = m.masks + MH_MAXVAR;
    m.nterm  = reinterpret_cast<int32_t*>(m.nmasks + MH_MAXVAR);
    int longest = 1;
    for (int j = 0; j < P.A * (P.fd->FS + P.fd->FO); ++j) longest = max(longest, P.fd->nodes[j].out);
    m.stride = longest + 1;
    return m;
}
__device__ __forceinline__ void mh_copy(float* dst, const float* src, int n, int lane)
{
    for (int k = lane; k < n; k += 64) dst[k] = src[k];
}
// BABNModel::incrementCountsOf (BABNModel.cpp:354-382; observation rows at the OLD state's parent values, App. A #6), the
// count of ONE node: k < FS the transition node of state feature k, else the observation node of feature k - FS
__device__ __forceinline__ void mh_increment_node(const Problem& P, float* cnt, int k, uint64_t fv, uint64_t nf, uint64_t of, int a, float amount)
{
    const FDesc* fd = P.fd;
    const GlobalView v{cnt};
    const bool tr   = k < fd->FS;
    const FNode& nd = tr ? fd->nodes[a * fd->FS + k] : fd->nodes[P.A * fd->FS + a * fd->FO + (k - fd->FS)];
    cnt[node_row(fd, nd, node_mask(fd, nd, v), fv) + (tr ? feat(nf, k) : feat(of, k - fd->FS))] += amount;
}
// (the whole step from one lane: the nested belief's filters)
__device__ __forceinline__ void fact_increment(const Problem& P, float* cnt, int s, int a, int o, int ns, float amount)
{
    const FDesc* fd   = P.fd;
    const uint64_t fv = pack_features(s, fd->Sstep, fd->FS), nf = pack_features(ns, fd->Sstep, fd->FS), of = pack_features(o, fd->Ostep, fd->FO);
    for (int k = 0; k < fd->FS + fd->FO; ++k) mh_increment_node(P, cnt, k, fv, nf, of, a, amount);
}
// (every lane of the wave calls; a lane per node)
__device__ __forceinline__ void mh_increment(const Problem& P, float* cnt, int s, int a, int o, int ns, float amount, int lane)
{
    const FDesc* fd   = P.fd;
    const uint64_t fv = pack_features(s, fd->Sstep, fd->FS), nf = pack_features(ns, fd->Sstep, fd->FS), of = pack_features(o, fd->Ostep, fd->FO);
    MH_SYNC();
    for (int k = lane; k < fd->FS + fd->FO; k += 64) mh_increment_node(P, cnt, k, fv, nf, of, a, amount);
    MH_SYNC();
}
// FBAPOMDPPrior::computePriorModel(structure): factored tiger (FactoredTigerPriors.cpp:293-321) = the prior with the listen
// observation node set for its parent set; collision avoidance (CollisionAvoidancePriors.cpp:490-526) = the prior with every
// obstacle's transition node, per action, set for its parent set.  A lane per node.
__device__ __forceinline__ void mh_compute_prior(const Problem& P, const DeviceState& D, const uint32_t* masks, float* out, int lane)
{
    MH_SYNC();
    mh_copy(out, D.prior, P.C, lane);
    MH_SYNC();
    if (dom_is_sys(P.domain)) {  // SysAdminFactoredPrior::computePriorModel (:98-127): every transition node anew
        const int N = P.sys->N;
        for (int i = lane; i < P.A * N; i += 64) sys_fill_node(P, out, i / N, i % N, masks[i]);
    } else if (dom_is_grid(P.domain)) {  // GridWorldFactBAPrior::computePriorModel (GridWorldBAPriors.cpp:227-254)
        for (int i = lane; i < P.A * 2; i += 64)
            if (masks[P.fd->nodes[(i >> 1) * 3 + (i & 1)].var] == 7u) gw_fill_xy_node_with_goal(P, out, i >> 1, i & 1);
    } else if (dom_is_ca(P.domain)) {
        const int n = P.ca->n;
        for (int i = lane; i < n * P.A; i += 64) ca_fill_obstacle_node(P, out, i % P.A, 2 + i / P.A, masks[(i % P.A) * n + i / P.A]);
    } else if (lane == 0) {
        ftiger_set_observation_model(P, out, masks[0]);
    }
    MH_SYNC();
}
// FBAPOMDP::mutate: FactoredTigerFactoredPrior::mutate (FactoredTigerPriors.cpp:351-381) flips a random edge
// (BABNModel.cpp:16-31) of O[listen][0]; CollisionAvoidanceFactoredPrior::mutate (CollisionAvoidancePriors.cpp:455-488)
// draws an action and an obstacle, then flips a random edge of that transition node
__device__ __forceinline__ void mh_mutate(const Problem& P, Rng& g, uint32_t* masks, int lane)
{
    int word;
    uint32_t bit;
    if (dom_is_sys(P.domain)) {  // SysAdminFactoredPrior::mutate (:47-55): computer, action (g++ evaluates the second subscript first), then the edge
        const int mc = g.uniform_int(P.sys->N), ma = g.uniform_int(P.A);
        word = ma * P.sys->N + mc;
        bit  = 1u << g.slow_int(0, P.fd->FS);
    } else if (dom_is_grid(P.domain)) {  // GridWorldFactBAPrior::mutate (GridWorldBAPriors.cpp:200-225): an action, the x or the y node, the goal edge toggled
        const int a = g.slow_int(0, P.A);
        const int f = g.slow_int(0, 2);
        word = P.fd->nodes[a * 3 + f].var;
        bit  = 4u;
    } else if (dom_is_ca(P.domain)) {
        const int n  = P.ca->n;
        const int a  = g.uniform_int(P.A);
        const int ob = g.uniform_int(n);
        word = a * n + ob;
        bit  = 1u << g.slow_int(0, P.fd->FS);
    } else {
        word = 0;
        bit  = 1u << g.slow_int(0, P.fd->FS);
    }
    MH_SYNC();
    if (lane == 0) masks[word] ^= bit;
    MH_SYNC();
}
// computePosteriorCounts (MHwithinGibbs.cpp:397-436): the prior plus one count per node and step of the history, a lane per node
__device__ __forceinline__ void mh_posterior(const Problem& P, const DeviceState& D, int e, const float* prior, const int32_t* seq, float* out, int lane)
{
    const FDesc* fd = P.fd;
    MH_SYNC();
    mh_copy(out, prior, P.C, lane);
    MH_SYNC();
    const int16_t *ha = D.mh_a + (size_t)e * P.episodes * P.horizon, *ho = D.mh_o + (size_t)e * P.episodes * P.horizon;
    const int32_t* len = D.mh_ep_len + (size_t)e * (P.episodes + 1);
    const int n_ep = D.mh_n_ep[e];
    for (int node = lane; node < fd->FS + fd->FO; node += 64) {
        int k = 0, h = 0;
        for (int ep = 0; ep < n_ep; ++ep) {
            for (int t = 0; t < len[ep]; ++t, ++h, ++k)
                mh_increment_node(P, out, node, pack_features(seq[k], fd->Sstep, fd->FS), pack_features(seq[k + 1], fd->Sstep, fd->FS),
                                  pack_features(ho[h], fd->Ostep, fd->FO), ha[h], 1.0f);
            ++k;
        }
    }
    MH_SYNC();
}
// expectedMult (random.cpp:257-279): float sum, float division, all-zero if the sum underflows
__device__ __forceinline__ void mh_expected(const float* row, int n, float* out)
{
    float sum = row[0];
    for (int i = 1; i < n; ++i) sum += row[i];
    for (int i = 0; i < n; ++i) out[i] = ((double)sum <= 1e-300) ? 0.f : row[i] / sum;
}
// BABNModel::flattenT / flattenO (BABNModel.cpp:89-181), a lane per (action, state).  The expected rows of the state's nodes (FS or FO rows of
// at most MAXROW values) are indexed by feature values, i.e. at run time: they sit in the LDS area of the score terms (MhScratch::terms,
// idle while a model is flattened), `need` floats per lane -- a private array would live in scratch memory, 512 bytes per lane.
__device__ __forceinline__ void mh_flatten(const Problem& P, const float* model, float* T, float* O, const MhScratch& m, int lane)
{
    const FDesc* fd = P.fd;
    const GlobalView v{model};
    const int S = P.S, A = P.A, NO = P.O;
    int need = 1;   // the longest total row length over the nodes of one (action, kind)
    for (int a = 0; a < A; ++a) {
        int ts = 0, os = 0;
        for (int f = 0; f < fd->FS; ++f) ts += fd->nodes[a * fd->FS + f].out;
        for (int f = 0; f < fd->FO; ++f) os += fd->nodes[A * fd->FS + a * fd->FO + f].out;
        need = max(need, max(ts, os));
    }
    const int lanes = min(64, (MH_TERMS * 2) / need);   // (need <= MAXF * MAXROW = 128: at least 16 lanes)
    float* ex = reinterpret_cast<float*>(m.terms) + (size_t)min(lane, lanes - 1) * need;
    MH_SYNC();
    for (int i = lane; i < A * S && lane < lanes; i += lanes) {
        const int a = i / S, s = i % S;
        int off[MAXF + 1];
        const uint64_t fv = pack_features(s, fd->Sstep, fd->FS);
        off[0] = 0;
#pragma unroll
        for (int f = 0; f < MAXF; ++f) {
            off[f + 1] = off[f];
            if (f < fd->FS) {
                const FNode& nd = fd->nodes[a * fd->FS + f];
                mh_expected(model + node_row(fd, nd, node_mask(fd, nd, v), fv), nd.out, ex + off[f]);
                off[f + 1] = off[f] + nd.out;
            }
        }
        for (int ns = 0; ns < S; ++ns) {
            const uint64_t nf = pack_features(ns, fd->Sstep, fd->FS);
            float p = 1;
#pragma unroll
            for (int f = 0; f < MAXF; ++f)
                if (f < fd->FS) p *= ex[off[f] + feat(nf, f)];
            T[((size_t)s * A + a) * S + ns] = p;
        }
#pragma unroll
        for (int f = 0; f < MAXF; ++f) {  // (s plays the new state here)
            off[f + 1] = off[f];
            if (f < fd->FO) {
                const FNode& nd = fd->nodes[A * fd->FS + a * fd->FO + f];
                mh_expected(model + node_row(fd, nd, node_mask(fd, nd, v), fv), nd.out, ex + off[f]);
                off[f + 1] = off[f] + nd.out;
            }
        }
        for (int o = 0; o < NO; ++o) {
            const uint64_t of = pack_features(o, fd->Ostep, fd->FO);
            float p = 1;
#pragma unroll
            for (int f = 0; f < MAXF; ++f)
                if (f < fd->FO) p *= ex[off[f] + feat(of, f)];
            O[((size_t)a * S + s) * NO + o] = p;
        }
    }
    MH_SYNC();
}
// rnd::sample::Dir::sampleFromMult<double> (random.hpp:93-115)
__device__ __forceinline__ int mh_sample_d(Rng& g, const double* m, int n, double total)
{
    const double p = g.u01() * total;
    double sum = m[0];
    for (int i = 1; i < n; ++i) {
        if (p < sum) return i - 1;
        sum += m[i];
    }
    return n - 1;
}
__device__ __forceinline__ bool mh_sample_history(const Problem& P, const DeviceState& D, int e, Rng& g, const MhScratch& m, const float* model, int lane)
{
    const int S = P.S, A = P.A, NO = P.O;
    const int16_t *ha = D.mh_a + (size_t)e * P.episodes * P.horizon, *ho = D.mh_o + (size_t)e * P.episodes * P.horizon;
    const int32_t* len = D.mh_ep_len + (size_t)e * (P.episodes + 1);
    const int n_ep = D.mh_n_ep[e];
    int k = 0, h0 = 0;
    if (P.mh == 2) {  // rejectionSampleStateHistory: one draw decides the next, the wave walks it as one
        MH_SYNC();
        for (int ep = 0; ep < n_ep; ++ep) {
            const int L = len[ep];
            bool ok = false;
            for (int tries = 0; !ok; ++tries) {
                if (tries >= (1 << 22)) return false;
                int s = domain_start(P, g);
                if (lane == 0) m.seq[k] = s;
                ok = true;
                for (int t = 0; t < L; ++t) {
                    int so;
                    double r;
                    fact_step<false>(P, g, GlobalView{model}, s, ha[h0 + t], so, r, NoInc{});
                    if (so != ho[h0 + t]) { ok = false; break; }
                    if (lane == 0) m.seq[k + 1 + t] = s;
                }
            }
            k += L + 1;
            h0 += L;
        }
        MH_SYNC();
        return true;
    }
    // msgSampleStateHistory: a lane per state of a message; the normalising sums run over the states in order
    mh_flatten(P, model, m.T, m.O, m, lane);
    const float init = 1.0f / (float)S;
    const float prior_p = (float)((double)init / (double)(init * (float)S));  // categoricalDistr(size, init)::prob (distributions.cpp:12-35)
    for (int ep = 0; ep < n_ep; ++ep) {
        const int L = len[ep];
        const int16_t *ea = ha + h0, *eo = ho + h0;
        for (int st = lane; st < S; st += 64) m.msg[(size_t)L * S + st] = (double)m.O[((size_t)ea[L - 1] * S + st) * NO + eo[L - 1]];
        MH_SYNC();
        for (int step = L - 1; step >= 0; --step) {
            const int a = ea[step];
            for (int st = lane; st < S; st += 64) {
                double acc = 0.0;
                for (int ns = 0; ns < S; ++ns) acc = acc + (double)m.T[((size_t)st * A + a) * S + ns] * m.msg[(size_t)(step + 1) * S + ns];
                if (step != 0) acc *= (double)m.O[((size_t)ea[step - 1] * S + st) * NO + eo[step - 1]];
                else acc *= (double)prior_p;
                m.msg[(size_t)step * S + st] = acc;
            }
            MH_SYNC();
            double tot = 0;
            for (int st = 0; st < S; ++st) tot += m.msg[(size_t)step * S + st];
            MH_SYNC();
            for (int st = lane; st < S; st += 64) m.msg[(size_t)step * S + st] = m.msg[(size_t)step * S + st] / tot;
            MH_SYNC();
        }
        int st = mh_sample_d(g, m.msg, S, 1);
        if (lane == 0) m.seq[k] = st;
        ++k;
        for (int step = 0; step < L; ++step) {
            MH_SYNC();
            for (int ns = lane; ns < S; ns += 64) m.probs[ns] = (double)m.T[((size_t)st * A + ea[step]) * S + ns] * m.msg[(size_t)(step + 1) * S + ns];
            MH_SYNC();
            double tot = 0;
            for (int ns = 0; ns < S; ++ns) tot += m.probs[ns];
            st = mh_sample_d(g, m.probs, S, tot);
            if (lane == 0) m.seq[k] = st;
            ++k;
        }
        h0 += L;
    }
    MH_SYNC();
    return true;
}
// BABNModel::LogBDScore (log_bd_score, fba_device.h): the lgamma terms of up to 64 Dirichlet rows at a time, a lane per row,
// then ONE running sum over them in CPT order -- the order, and so the double, of the one-lane score
template <class View>
__device__ __forceinline__ double mh_log_bd_score(const Problem& P, const View& cnt, const View& prior, const MhScratch& m, int lane)
{
    const FDesc* fd = P.fd;
    const int per = fd->FS + fd->FO, nn = P.A * per, W = min(64, MH_TERMS / m.stride);
    MH_SYNC();
    if (W < 2) return log_bd_score(P, cnt, prior);  // rows too long to share out
    const auto node_of = [&](int j) -> const FNode& {
        const int a = j / per, k = j % per;
        return k < fd->FS ? fd->nodes[a * fd->FS + k] : fd->nodes[P.A * fd->FS + a * fd->FO + (k - fd->FS)];
    };
    const auto rows_of = [&](const FNode& nd) {
        const uint32_t mask = node_mask(fd, nd, cnt);
        int rows = 1;
        for (int j = 0; j < nd.nmax; ++j)
            if ((mask >> j) & 1u) rows *= nd.psz[j];
        return rows;
    };
    double bd = 0;
    int node = 0, row = 0;
    while (node < nn) {
        int filled = 0, my_node = -1, my_row = 0;
        while (filled < W && node < nn) {  // hand the next rows out, across node boundaries
            const int rows = rows_of(node_of(node)), take = min(W - filled, rows - row);
            if (lane >= filled && lane < filled + take) { my_node = node; my_row = row + (lane - filled); }
            filled += take;
            row += take;
            if (row == rows) { ++node; row = 0; }
        }
        if (my_node >= 0) {
            const FNode& nd = node_of(my_node);
            double* t  = m.terms + lane * m.stride;
            double tot = 0, ptot = 0;
            for (int v = 0; v < nd.out; ++v) {
                const float x = cnt.at(nd.off + my_row * nd.out + v), y = prior.at(nd.off + my_row * nd.out + v);
                tot += (double)x;
                ptot += (double)y;
                t[v] = log_gamma((double)x) - log_gamma((double)y);
            }
            t[nd.out]     = log_gamma(ptot) - log_gamma(tot);
            m.nterm[lane] = nd.out + 1;
        }
        MH_SYNC();
        for (int l = 0; l < filled; ++l) {
            const double* t = m.terms + l * m.stride;
            const int n     = m.nterm[l];
            for (int v = 0; v < n; ++v) bd += t[v];
        }
        MH_SYNC();
    }
    return bd;
}
__global__ void __launch_bounds__(64) mh_kernel(Problem P, DeviceState D, int scratch_in_lds)
{
    extern __shared__ __align__(16) char mh_lds[];
    const int e = blockIdx.x, lane = threadIdx.x;
    if (!D.cheat_pending[e]) return;
    MH_SYNC();  // (every lane has read the flag)
    D.cheat_pending[e] = 0;
    // history.back().add(a, o); log likelihood.  (Every lane stores the same values: each then reads what it wrote.)
    int16_t *ha = D.mh_a + (size_t)e * P.episodes * P.horizon, *ho = D.mh_o + (size_t)e * P.episodes * P.horizon;
    int32_t* len = D.mh_ep_len + (size_t)e * (P.episodes + 1);
    const int n_ep = D.mh_n_ep[e];
    int h = 0, nseq = 0;
    for (int ep = 0; ep < n_ep; ++ep) h += len[ep];
    if (h >= P.episodes * P.horizon) {  // more updates than a run has steps: only the per-step interface can get here
        atomicCAS(D.fault, 0, 0x40000000 + e);
        return;
    }
    const int len_last = len[n_ep - 1];
    const double ll    = D.lik[e] + det_log(D.cur[e].weight_total);
    MH_SYNC();
    ha[h] = (int16_t)D.action[e];
    ho[h] = (int16_t)D.obs[e];
    len[n_ep - 1] = len_last + 1;
    D.lik[e]      = ll;
    if (!(ll < D.lik[P.E])) return;
    __threadfence_block();
    MH_SYNC();

    // reinvigorate
    for (int ep = 0; ep < n_ep; ++ep) nseq += len[ep] + 1;
    const FDesc* fd = P.fd;
    const MhScratch m = mh_scratch(P, scratch_in_lds ? reinterpret_cast<float*>(mh_lds + MH_LDS_HEAD) : D.mh_scratch + (size_t)e * D.mh_scratch_words, mh_lds);
    const int cur = D.bufsel[e], N = P.N;
    const float* old_recs = D.p_rec + pbase(P, e, cur) * (size_t)P.Cs;
    float* new_recs       = D.p_rec + pbase(P, e, cur ^ 1) * (size_t)P.Cs;
    double* new_w         = D.p_weight + pbase(P, e, cur ^ 1);
    Rng g = slot_rng(P, D, e);
    g.stream(FBA_PHASE_REINVIG, 0);
    const double w1 = 1.0 / (double)N;
    const int nvar  = fd->nvar;
    uint32_t *masks = m.masks, *nmasks = m.nmasks;
    int made = 0;
    bool ok  = true;
    if (P.mh == 3) {
        // MHNIPS2018::MH (MHNIPS2018.cpp:208-255): independent proposals -- a particle of the old filter, its structure
        // or (half of the time) a mutation, that structure's prior updated along a freshly simulated history
        // (computePosterior :39-105: a wrong observation undoes the episode's increments and starts it over), accepted on
        // the difference of the LogBDScores
        const int16_t *ca = ha, *co = ho;
        for (int iters = 0; made < N; ++iters) {
            if (iters >= (1 << 24)) { ok = false; break; }
            const float* src = old_recs + (size_t)uniform_weight_pick(D.uni_scan, N, g.u01() * D.uni_total, D.uni_total) * P.Cs;  // old_belief.sample()->model()
            MH_SYNC();
            for (int v = lane; v < nvar; v += 64) masks[v] = __float_as_uint(src[fd->ncounts + v]);
            mh_compute_prior(P, D, masks, m.prior, lane);        // sampled_prior_model
            if (!g.boolean()) mh_mutate(P, g, masks, lane);      // the same structure half of the time
            mh_compute_prior(P, D, masks, m.model, lane);        // new_prior_model
            mh_copy(m.fresh, m.model, P.C, lane);
            MH_SYNC();
            int last = 0, h0 = 0;
            for (int ep = 0; ok && ep < n_ep; ++ep) {
                const int L = len[ep];
                for (int tries = 0;; ++tries) {
                    if (tries >= (1 << 22)) { ok = false; break; }
                    int s = domain_start(P, g), t = 0;
                    for (; t < L; ++t) {
                        const int from = s;
                        int so;
                        double r;
                        fact_step<false>(P, g, GlobalView{m.fresh}, s, ca[h0 + t], so, r, NoInc{});
                        last = s;
                        if (so != co[h0 + t]) break;
                        if (lane == 0) { m.seq[2 * t] = from; m.seq[2 * t + 1] = s; }
                        mh_increment(P, m.fresh, from, ca[h0 + t], so, s, 1.0f, lane);
                    }
                    if (t == L) break;
                    for (int u = 0; u < t; ++u) mh_increment(P, m.fresh, m.seq[2 * u], ca[h0 + u], co[h0 + u], m.seq[2 * u + 1], -1.0f, lane);
                }
                h0 += L;
            }
            if (!ok) break;
            const double old_score = mh_log_bd_score(P, GlobalView{src}, GlobalView{m.prior}, m, lane);
            const double new_score = mh_log_bd_score(P, GlobalView{m.fresh}, GlobalView{m.model}, m, lane);
            if (det_log(g.u01()) < (new_score - old_score)) {
                float* dst = new_recs + (size_t)made * P.Cs;
                mh_copy(dst, m.fresh, P.C, lane);
                if (lane == 0) {
                    rec_set_state(dst, P.C, last);
                    new_w[made] = w1;
                }
                ++made;
            }
        }
    } else {
        {
            const float* src = old_recs + (size_t)uniform_weight_pick(D.uni_scan, N, g.u01() * D.uni_total, D.uni_total) * P.Cs;  // old_belief.sample()->model()
            mh_copy(m.model, src, P.C, lane);
            MH_SYNC();
        }
        ok = mh_sample_history(P, D, e, g, m, m.model, lane);
        for (int v = lane; v < nvar; v += 64) masks[v] = __float_as_uint(m.model[fd->ncounts + v]);
        mh_compute_prior(P, D, masks, m.prior, lane);
        mh_posterior(P, D, e, m.prior, m.seq, m.model, lane);
        double score = mh_log_bd_score(P, GlobalView{m.model}, GlobalView{m.prior}, m, lane);
        for (int iters = 0; ok && made < N; ++iters) {
            if (iters >= (1 << 24)) { ok = false; break; }
            MH_SYNC();
            for (int v = lane; v < nvar; v += 64) nmasks[v] = masks[v];
            mh_mutate(P, g, nmasks, lane);
            mh_compute_prior(P, D, nmasks, m.prior, lane);
            mh_posterior(P, D, e, m.prior, m.seq, m.fresh, lane);
            const double new_score = mh_log_bd_score(P, GlobalView{m.fresh}, GlobalView{m.prior}, m, lane);
            if (det_log(g.u01()) < (new_score - score)) {
                float* dst = new_recs + (size_t)made * P.Cs;
                mh_copy(dst, m.fresh, P.C, lane);
                if (lane == 0) {
                    rec_set_state(dst, P.C, m.seq[nseq - 1]);
                    new_w[made] = w1;
                }
                ++made;
                ok = mh_sample_history(P, D, e, g, m, m.model, lane);   // (from the model of the LAST accepted structure, as the reference does)
                mh_posterior(P, D, e, m.prior, m.seq, m.model, lane);
                for (int v = lane; v < nvar; v += 64) masks[v] = nmasks[v];
                score = mh_log_bd_score(P, GlobalView{m.model}, GlobalView{m.prior}, m, lane);
            }
        }
    }
    if (!ok) {
        atomicCAS(D.fault, 0, 0x20000000 + e);
        return;
    }
    MH_SYNC();
    D.bufsel[e] = cur ^ 1;
    D.lik[e]    = 0.0;
}

// ---------------------------------------------------------------------------------------------
// Device-order prefix sums (DESIGN.md "device-order sums"; oracle/orc.c dev_scan is the CPU twin):
// each lane sums 4 consecutive elements sequentially, a 64-lane Kogge-Stone scan combines the
// lane sums of one 256-element chunk, chunks are chained sequentially.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_inclusive_scan(double v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double t = __shfl_up(v, d, 64);
        if (lane >= d) v = v + t;
    }
    return v;
}

// in: n doubles; out_incl may be null.  s_carry: LDS, at least n/256 + 2 doubles.  Returns the
// total to every thread.  All threads of the block must call.
__device__ __forceinline__ double block_device_scan(const double* in, int n, double* out_incl, double* s_carry)  // (out_incl may be `in`)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int nchunks = (n + 255) >> 8;
    for (int c = wave; c < nchunks; c += nwaves) {
        const int i0 = c * 256 + lane * 4;
        double s = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double x = (i0 + k < n) ? in[i0 + k] : 0.0;
            s = (k == 0) ? x : s + x;
        }
        const double incl = wave_inclusive_scan(s, lane);
        if (lane == 63) s_carry[c + 1] = incl;  // chunk total
    }
    __syncthreads();
    if (tid == 0) {
        double carry = 0;
        s_carry[0]   = 0;
        for (int c = 0; c < nchunks; ++c) {
            const double t = s_carry[c + 1];
            carry          = carry + t;
            s_carry[c + 1] = carry;  // carry into chunk c + 1 (the last one is the total)
        }
    }
    __syncthreads();
    const double total = s_carry[nchunks];
    if (out_incl) {
        for (int c = wave; c < nchunks; c += nwaves) {
            const int i0 = c * 256 + lane * 4;
            double x[4], s = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                x[k] = (i0 + k < n) ? in[i0 + k] : 0.0;
                s    = (k == 0) ? x[k] : s + x[k];
            }
            const double incl = wave_inclusive_scan(s, lane);
            double excl       = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 0.0;
            double run = s_carry[c] + excl;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (i0 + k < n) {
                    run += x[k];
                    out_incl[i0 + k] = run;
                }
            }
        }
    }
    __syncthreads();
    return total;
}

// uniform_scan_kernel: prefix sums of N weights 1/N, computed once per ctx.
__global__ void __launch_bounds__(IS_BLOCK) uniform_scan_kernel(int n, double* w_tmp, double* out, double* total)
{
    __shared__ double s_carry[IS_MAX_CHUNKS + 2];
    const double w = 1.0 / (double)n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) w_tmp[i] = w;
    __syncthreads();
    const double t = block_device_scan(w_tmp, n, out, s_carry);
    if (threadIdx.x == 0) *total = t;
}

// ---------------------------------------------------------------------------------------------
// The nested belief (NestedBelief.cpp; -B nested): a weighted filter of N count particles, each with its own flat
// filter of M = N^2 domain states.  One lane = one count particle: its update is sequential by definition (every
// accepted sample changes the counts the next attempt samples from), the N particles of a slot and the slots run side
// by side.  Dense records only (the increments are 1/M).
//   nested_fill_kernel   initiate (:61-87, stream INIT_FC i: M start states; the count particle itself comes from
//                        init_kernel, stream INIT i, its own domain state released) / resetDomainStateDistribution
//                        (:35-58, stream RESET i: M start states)
//   nested_update_kernel updateEstimation (:127-192, stream REJECT i): until M samples are accepted -- a state of the
//                        filter, one KeepCounts step on the particle's own counts, accept on the observation,
//                        incrementCountsOf(old, a, o, new, 1/M) -- then weight *= 1 / attempts; WeightedFilter::normalize
//                        and the prefix sums sample() needs, in device order.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) nested_fill_kernel(Problem P, DeviceState D, int reset)
{
    __shared__ double s_carry[4];
    const int e = blockIdx.x, i = threadIdx.x;
    if (reset ? D.need_reset[e] != 1 : !D.need_init[e]) return;
    const int M = P.nested;
    if (!reset && i == 0) D.nest_sel[e] = 0;
    __syncthreads();
    const size_t pb = pbase(P, e, D.bufsel[e]);
    if (i < P.N) {
        int32_t* filter = D.nest_s + nest_base(P, e, reset ? D.nest_sel[e] : 0) + (size_t)i * M;
        Rng g = slot_rng(P, D, e);
        g.position((uint32_t)D.run[e], reset ? (uint32_t)D.episode[e] : 0u, 0);
        g.stream(reset ? FBA_PHASE_RESET : FBA_PHASE_INIT_FC, (uint32_t)i);
        for (int j = 0; j < M; ++j) filter[j] = domain_start(P, g);
        if (!reset) rec_set_state(D.p_rec + (pb + i) * (size_t)P.Cs, P.C, 0);  // (the count particle's own domain state is released at once)
    }
    if (reset) return;
    __syncthreads();
    // WeightedFilter(n, alloc): weights 1/n (init_kernel); the prefix sums sample() walks
    const double total = block_device_scan(D.p_weight + pb, P.N, D.nest_scan + (size_t)e * P.N, s_carry);
    if (i == 0) D.nest_total[e] = total;
}

template <bool REG>
__global__ void __launch_bounds__(256) nested_update_kernel(Problem P, DeviceState D)
{
    __shared__ double s_carry[4];
    __shared__ unsigned long long s_attempts;
    const int e = blockIdx.x, i = threadIdx.x;
    if (!D.need_update[e]) return;
    if (i == 0) s_attempts = 0;
    __syncthreads();
    const int M = P.nested, a = D.action[e], o = D.obs[e], cur = D.nest_sel[e];
    const size_t pb = pbase(P, e, D.bufsel[e]);
    double* w = D.p_weight + pb;
    if (i < P.N) {
        const int32_t* filter = D.nest_s + nest_base(P, e, cur) + (size_t)i * M;
        int32_t* fresh        = D.nest_s + nest_base(P, e, cur ^ 1) + (size_t)i * M;
        float* rec            = D.p_rec + (pb + i) * (size_t)P.Cs;
        const float amount    = (float)(1.0 / (double)(float)M);
        Rng g = slot_rng(P, D, e);
        g.stream(FBA_PHASE_REJECT, (uint32_t)i);
        int acc = 0, count = 0;
        while (acc < M) {
            const int old = filter[g.uniform_int(M)];
            int s = old, so;
            double r;
            sim_step<REG>(P, g, GlobalView{rec}, s, a, so, r, NoInc{});
            ++count;
            if (so == o) {
                fresh[acc++] = s;
                if (P.model == FBA_MODEL_BA_FACTORED) fact_increment(P, rec, old, a, o, s, amount);
                else {  // BAFlatModel::incrementCountsOf (BAFlatModel.cpp:130-139)
                    rec[old * P.A * P.S + a * P.S + s] += amount;
                    rec[P.phi_len + a * P.S * P.O + s * P.O + o] += amount;
                }
            }
            if (count >= (1 << 24)) {  // no state of this filter can produce the observation: the reference would never return
                atomicCAS(D.fault, 0, 1 + e);
                break;
            }
        }
        w[i] *= 1.0 / (double)count;
        atomicAdd(&s_attempts, (unsigned long long)count);
    }
    __syncthreads();
    const double total = block_device_scan(w, P.N, nullptr, s_carry);
    if (i < P.N) w[i] = w[i] / total;
    __syncthreads();
    const double norm = block_device_scan(w, P.N, D.nest_scan + (size_t)e * P.N, s_carry);
    if (i == 0) {
        D.nest_total[e]  = norm;
        D.nest_sel[e]    = cur ^ 1;
        D.need_update[e] = 0;
        D.belief_steps[e] += s_attempts;
        D.upd_attempts[e] += s_attempts;
        D.upd_particles[e] += (unsigned long long)P.N * (unsigned long long)M;
        D.cur[e].update_count = (int32_t)s_attempts;
        D.cur[e].weight_total = total;
    }
}

// ---------------------------------------------------------------------------------------------
// importance_kernel: importance_sampling::update + resample (ImportanceSampler.hpp:31-94,
// WeightedFilter::normalize WeightedFilter.cpp:130-143, ::sample :163-191), one workgroup per slot.
//   1. every particle steps in place (UpdateCounts) and multiplies its weight by P(o | a, s')
//   2. total weight, normalisation, prefix sums of the normalised weights     (device order)
//   3. N multinomial draws by binary search on the prefix sums
//   4. whole-record gather into the other buffer, weights reset to 1/N
// ---------------------------------------------------------------------------------------------
// HIST: history particles (gridworld FBA-POMDP): the step reads its rows through the particle's entries, the update
// is one new entry per particle, appended by the gather.
// WLDS: the slot's weights live in LDS from the update pass to the last draw (N <= IS_LDS_MAX_N: 8 bytes each): the
// normalised weights and their prefix sums never go to HBM, the N searches read LDS.  Same sums, same order.
// history particles: all N sources are drawn before the gather when their 16-bit indices fit the workgroup's LDS beside the weights
__host__ __device__ __forceinline__ bool hist_all_draws_first(int N) { return N <= 16384; }
template <bool REG, int TIGER_TABLE, bool HIST = false, bool WLDS = false>
__global__ void __launch_bounds__(IS_BLOCK) importance_kernel(Problem P, DeviceState D)
{
    if (TIGER_TABLE) {  // sizes restated as literals (see search_kernel)
        P.model = FBA_MODEL_BA_TABLE;
        P.S = 2; P.A = 3; P.O = 2; P.phi_len = 12; P.C = 24; P.Cs = 32;
        if (TIGER_TABLE == 2) { P.C = 12; P.Cs = 16; }  // packed particles (PackedView)
        if (P.domain != FBA_DOM_TIGER_CONTINUOUS) P.domain = FBA_DOM_TIGER_EPISODIC;
    }
    __shared__ __attribute__((aligned(8))) float s_prior[TIGER_TABLE == 2 ? 24 : 2];
    __shared__ double s_carry[IS_MAX_CHUNKS + 2];
    __shared__ int32_t s_src[IS_BLOCK], s_inc[HIST ? 1 : (TIGER_TABLE ? 2 : MAXINC) * IS_BLOCK];
    extern __shared__ double s_w[];  // WLDS: [N] weights, then normalised weights, then their inclusive prefix sums
    const int e = chunk_slot(D, blockIdx.x), tid = threadIdx.x;
    if (!D.need_update[e]) return;
    if (TIGER_TABLE == 2) {
        if (tid < 24) s_prior[tid] = D.prior_dense[tid];
        __syncthreads();
    }
    const int a = D.action[e], o = D.obs[e], N = P.N;
    const uint32_t hist_cnt = HIST ? D.hist_cnt[e] : 0u;
    const int hist_n = hist_total(hist_cnt);
    if (HIST && hist_n >= P.hist_cap) {  // more real steps than the records were sized for (episodes * horizon): only the per-step interface can get here
        if (tid == 0) {
            atomicCAS(D.fault, 0, 0x40000000 + e);
            D.need_update[e] = 0;
            D.active[e]      = 0;
        }
        return;
    }
    const int cur = D.bufsel[e];
    const size_t sb = pbase(P, e, cur), db = pbase(P, e, cur ^ 1);
    double* sw   = D.p_weight + sb;
    float* scn   = D.p_rec + rec_base(P, D, e, cur) * (size_t)P.Cs;
    double* dw      = D.p_weight + db;
    float* dcn      = rec_dst(P, D, e, cur ^ 1);
    double* wscan   = WLDS ? s_w : D.wscan + (size_t)e * N;
    double* wcur    = WLDS ? s_w : sw;   // where the update pass leaves the new weights
    int32_t* side   = D.p_side + (size_t)e * N * D.side_w;
    const bool defer = defer_increments(P);
    const int rs = HIST ? hist_stride(P, hist_n) : P.Cs, rd = HIST ? hist_stride(P, hist_n + 1) : P.Cs;   // words between the records read / written
    const int C4 = rs / 4, group = record_group(HIST ? rd / 4 : C4);
    const int ninc = model_ninc(P);
    Rng g = slot_rng(P, D, e);

    for (int i = tid; i < N; i += IS_BLOCK) {
        g.stream(FBA_PHASE_IS_UPDATE, (uint32_t)i);
        float* cnt = scn + (size_t)i * rs;
        int s = rec_state(cnt, P.C), so;
        double r;
        if (HIST) {
            const uint32_t* rec = reinterpret_cast<const uint32_t*>(cnt);
            uint32_t sp = (rec[1] >> 16) & 0x3ffu, entry;
            double prob;
            gridworld_hist_step(P, g, rec + 2 + hist_offset(hist_cnt, a), hist_count(hist_cnt, a), rec[1], sp, a, so, r, entry, o, prob);
            *reinterpret_cast<int2*>(side + (size_t)i * 2) = make_int2(gridworld_unpack_state(P, sp), (int)entry);
            wcur[i] = sw[i] * prob;
            continue;
        }
        if (TIGER_TABLE == 2) tiger_step_packed(P, g, [&](int w) { return __float_as_uint(cnt[w]); }, s_prior, s, a, so, r, LdsInc<IS_BLOCK>{s_inc + tid});
        else sim_step<REG>(P, g, GlobalView{cnt}, s, a, so, r, LdsInc<IS_BLOCK>{s_inc + tid});
        // incrementCountsOf (BAFlatModel.cpp:126-139, BABNModel.cpp:354-382) is deferred to the gather: the new
        // state and the cells go to the side array, the record is only read
        if (TIGER_TABLE == 2 && WLDS) {
            // packed tiger particles: the pending update {new state, T cell, O cell} is 11 bits and stays in LDS
            reinterpret_cast<uint16_t*>(s_w + N)[i] = (uint16_t)(s | (s_inc[tid] << 1) | (s_inc[IS_BLOCK + tid] << 6));
            wcur[i] = sw[i] * sim_obs_prob<REG>(P, g, PendingIncView<PackedView<GlobalView>>{PackedView<GlobalView>{GlobalView{cnt}, s_prior}, s_inc + tid, IS_BLOCK, ninc}, s, a, o);
        } else if (defer) {
            int32_t* sd = side + (size_t)i * D.side_w;
            sd[0] = s;
            for (int q = 0; q < ninc; ++q) sd[1 + q] = s_inc[q * IS_BLOCK + tid];
            // probability from the updated counts
            if (TIGER_TABLE == 2)
                wcur[i] = sw[i] * sim_obs_prob<REG>(P, g, PendingIncView<PackedView<GlobalView>>{PackedView<GlobalView>{GlobalView{cnt}, s_prior}, s_inc + tid, IS_BLOCK, ninc}, s, a, o);
            else wcur[i] = sw[i] * sim_obs_prob<REG>(P, g, PendingIncView<GlobalView>{GlobalView{cnt}, s_inc + tid, IS_BLOCK, ninc}, s, a, o);
        } else {
            for (int q = 0; q < ninc; ++q) cnt[s_inc[q * IS_BLOCK + tid]] += 1.0f;
            rec_set_state(cnt, P.C, s);
            wcur[i] = sw[i] * sim_obs_prob<REG>(P, g, GlobalView{cnt}, s, a, o);
        }
    }
    __syncthreads();
    const double total = block_device_scan(wcur, N, nullptr, s_carry);
    for (int i = tid; i < N; i += IS_BLOCK) wcur[i] /= total;
    __syncthreads();
    const double total_w = block_device_scan(wcur, N, wscan, s_carry);   // (in place when WLDS: every thread reads its four elements before it writes them)
    const double w1 = 1.0 / (double)N;
    if (TIGER_TABLE == 2 && WLDS) {
        // all N draws first (their sources as 16-bit indices in LDS), then one pass of N independent 64-byte copies, four
        // lanes per record and four records in flight per lane, each with its source's pending update applied
        uint16_t* s_side = reinterpret_cast<uint16_t*>(s_w + N);
        uint16_t* s_all  = s_side + N;
        for (int j = tid; j < N; j += IS_BLOCK) {
            g.stream(FBA_PHASE_RESAMPLE, (uint32_t)j);
            s_all[j] = (uint16_t)weighted_pick_guided(wscan, N, g.u01() * total_w, total_w);
            dw[j]    = w1;
        }
        __syncthreads();
        const int part = tid & 3;
        constexpr int GROUPS = IS_BLOCK / 4, UNROLL = 4;
        for (int j0 = tid >> 2; j0 < N; j0 += GROUPS * UNROLL) {
            uint32_t sd[UNROLL];
            float4 v[UNROLL];
#pragma unroll
            for (int q = 0; q < UNROLL; ++q) {
                const int j = j0 + q * GROUPS;
                const int p = j < N ? s_all[j] : 0;
                sd[q] = s_side[p];
                if (j < N) v[q] = reinterpret_cast<const float4*>(scn)[(size_t)p * 4 + part];
            }
#pragma unroll
            for (int q = 0; q < UNROLL; ++q) {
                const int j = j0 + q * GROUPS;
                if (j >= N) continue;
                bump_cell(v[q], (int)((sd[q] >> 1) & 31u), part * 4, true);
                bump_cell(v[q], (int)((sd[q] >> 6) & 31u), part * 4, true);
                if (part == 3) v[q].x = __int_as_float((int)(sd[q] & 1u));  // the state word (12)
                reinterpret_cast<float4*>(dcn)[(size_t)j * 4 + part] = v[q];
            }
        }
    } else if (HIST && hist_all_draws_first(N)) {
        // every source first (16-bit indices in LDS), then ONE pass over the filter with four records in flight per lane group: the
        // 2 N / IS_BLOCK barriers of the round-by-round form, each behind a trip to memory, were most of this kernel
        uint16_t* s_all = reinterpret_cast<uint16_t*>(s_w + (WLDS ? N : 0));
        for (int j = tid; j < N; j += IS_BLOCK) {
            g.stream(FBA_PHASE_RESAMPLE, (uint32_t)j);
            s_all[j] = (uint16_t)weighted_pick_guided(wscan, N, g.u01() * total_w, total_w);
            dw[j]    = w1;
        }
        __syncthreads();
        gather_hist_records(dcn, scn, s_all, side, hist_n, 2 + hist_offset(hist_cnt, a) + hist_count(hist_cnt, a), N, C4, group, IS_BLOCK, rd / 4);
    } else
    for (int j0 = 0; j0 < N; j0 += IS_BLOCK) {
        const int j = j0 + tid;
        if (j < N) {
            g.stream(FBA_PHASE_RESAMPLE, (uint32_t)j);
            const int src = weighted_pick_guided(wscan, N, g.u01() * total_w, total_w);
            s_src[tid]    = src;
            dw[j]         = w1;
        }
        __syncthreads();
        const int m = min(IS_BLOCK, N - j0);
        if (HIST) gather_hist_records(dcn + (size_t)j0 * rd, scn, s_src, side, hist_n, 2 + hist_offset(hist_cnt, a) + hist_count(hist_cnt, a), m, C4, group, IS_BLOCK, rd / 4);
        else if (defer) gather_records_side(dcn + (size_t)j0 * P.Cs, scn, s_src, side, D.side_w, m, C4, P.C, group, IS_BLOCK, TIGER_TABLE == 2);
        else gather_records(dcn + (size_t)j0 * P.Cs, scn, nullptr, s_src, nullptr, 0, 0, nullptr, m, C4, P.C, group, IS_BLOCK);
        __syncthreads();
    }
    if (tid == 0) {
        D.bufsel[e]      = cur ^ 1;
        D.need_update[e] = 0;
        D.belief_steps[e] += (unsigned long long)N;
        D.upd_attempts[e] += (unsigned long long)N;
        D.upd_particles[e] += (unsigned long long)N;
        D.cur[e].update_count = -1;
        D.cur[e].weight_total = total;
        if (HIST) {
            D.hist_cnt[e] = hist_cnt + (1u << (8 * a));
            D.upd_entries[e] += (unsigned long long)N * (unsigned long long)hist_n;
            if (D.single_rec) D.copy_pending[e] = 1;
        }
        if (P.mh) D.cheat_pending[e] = 1;  // mh_kernel: append (a, o) to the history, add log(total), maybe re-draw the filter
        if (P.cheat) {  // CheatingReinvigoration::updateEstimation (CheatingReinvigoration.cpp:117-124)
            double lik = D.lik[e] * total;
            if (det_log(lik) < D.lik[P.E]) {
                D.cheat_pending[e] = 1;
                lik = 1;
            }
            D.lik[e] = lik;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Multi-workgroup importance filter: the same algorithm and the same device-order sums as
// importance_kernel, cut at its block-wide synchronisation points into separate launches so that a
// filter of millions of particles (N > 65536) spreads over the whole chip:
//   is_multi_step   step + weight + totals of each 256-chunk      (one wave per chunk)
//   scan_carry      chunk totals -> carries, sequentially          (one lane per slot)
//   is_multi_norm   w /= total, chunk totals of the normalised w
//   scan_carry
//   scan_write      inclusive prefix sums from carries
//   is_multi_resample  N binary searches + whole-record gather
//   is_multi_finish flip buffers, counters
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double chunk_lane_sum(const double* __restrict__ in, int n, int i0, double (&x)[4])
{
    double s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        x[k] = (i0 + k < n) ? in[i0 + k] : 0.0;
        s    = (k == 0) ? x[k] : s + x[k];
    }
    return s;
}

// totals of every 256-element chunk of w (ctot[c + 1] = total of chunk c); one wave per chunk
__global__ void __launch_bounds__(256) chunk_totals_kernel(const double* w_base, size_t w_stride, int n, double* ctot_base, int ctot_stride,
                                                           const uint8_t* need)
{
    const int e = blockIdx.y;
    if (need && !need[e]) return;
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c * 256 >= n) return;
    double x[4];
    const double incl = wave_inclusive_scan(chunk_lane_sum(w_base + (size_t)e * w_stride, n, c * 256 + lane * 4, x), lane);
    if (lane == 63) ctot_base[(size_t)e * ctot_stride + c + 1] = incl;
}

// One workgroup per slot.  The chain carry_{c+1} = carry_c + total_c is sequential by definition
// (device-order sums); the totals are staged through LDS in tiles so the one lane that chains them
// never waits on HBM.
// (block b works on slot slot_list[slot_base + b] when a list is given -- a chunk of the compacted list of a budgeted context -- else on slot_base + b)
__global__ void __launch_bounds__(256) scan_carry_kernel(double* ctot_base, int ctot_stride, int nchunks, double* total_base,
                                                         int total_stride, int which, const uint8_t* need, int count,
                                                         const int32_t* slot_list = nullptr, int slot_base = 0)
{
    __shared__ double tile[CARRY_TILE];
    __shared__ double s_carry0;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= count) return;
    const int e = slot_list ? slot_list[slot_base + blockIdx.x] : slot_base + (int)blockIdx.x;
    if (need && !need[e]) return;
    double* ct = ctot_base + (size_t)e * ctot_stride;
    if (tid == 0) { s_carry0 = 0; ct[0] = 0; }
    for (int c0 = 0; c0 < nchunks; c0 += CARRY_TILE) {
        const int m = min(CARRY_TILE, nchunks - c0);
        __syncthreads();
        for (int k = tid; k < m; k += 256) tile[k] = ct[c0 + k + 1];
        __syncthreads();
        if (tid == 0) {
            double carry = s_carry0;
            int k = 0;
            for (; k + 8 <= m; k += 8) {  // eight totals in registers before the eight dependent additions
                double t[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) t[q] = tile[k + q];
#pragma unroll
                for (int q = 0; q < 8; ++q) { carry = carry + t[q]; t[q] = carry; }
#pragma unroll
                for (int q = 0; q < 8; ++q) tile[k + q] = t[q];
            }
            for (; k < m; ++k) {
                carry   = carry + tile[k];
                tile[k] = carry;
            }
            s_carry0 = carry;
        }
        __syncthreads();
        for (int k = tid; k < m; k += 256) ct[c0 + k + 1] = tile[k];
    }
    __syncthreads();
    if (tid == 0) total_base[(size_t)e * total_stride + which] = s_carry0;
}

__global__ void __launch_bounds__(256) scan_write_kernel(const double* w_base, size_t w_stride, int n, const double* ctot_base, int ctot_stride,
                                                         double* out_base, size_t out_stride, const uint8_t* need)
{
    const int e = blockIdx.y;
    if (need && !need[e]) return;
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c * 256 >= n) return;
    const int i0 = c * 256 + lane * 4;
    double x[4];
    const double incl = wave_inclusive_scan(chunk_lane_sum(w_base + (size_t)e * w_stride, n, i0, x), lane);
    double excl       = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 0.0;
    double run  = ctot_base[(size_t)e * ctot_stride + c] + excl;
    double* out = out_base + (size_t)e * out_stride;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i0 + k < n) {
            run += x[k];
            out[i0 + k] = run;
        }
}

__global__ void fill_kernel(double* p, int n, double v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// a history-particle slot whose records are full (more real steps than episodes * horizon: only the per-step interface can get there) is
// skipped by every launch of the update; is_multi_finish_kernel raises the fault
__device__ __forceinline__ bool hist_update_refused(const Problem& P, const DeviceState& D, int e)
{
    return P.hist && hist_total(D.hist_cnt[e]) >= P.hist_cap;
}
// HIST: history particles (importance_kernel<HIST> cut at its block-wide synchronisation points, so that a slot's update spreads over
// N / 1024 workgroups instead of walking its chain in one)
// K > 0 (history particles): the prior's Dirichlet rows, deduplicated, in LDS as the search has them (Problem::hist_lds, rows of K floats) -- six rows
// per particle step otherwise come from L2, where the records streaming through evict them (the sweep fetched three times its records' bytes)
template <bool REG, bool HIST = false, int K = 0>
__global__ void __launch_bounds__(256) is_multi_step_kernel(Problem P, DeviceState D)
{
    __shared__ int32_t s_inc[HIST ? 1 : MAXINC * 256];
    extern __shared__ uint4 s_prior_rows[];   // K > 0: one byte per row slot (HistRowIds numbering), then the distinct rows
    const int e = chunk_slot(D, blockIdx.y), tid = threadIdx.x, lane = tid & 63;
    if (!D.need_update[e] || hist_update_refused(P, D, e)) return;
    if (K > 0) {   // every thread, before any wave leaves
        const uint4* src = reinterpret_cast<const uint4*>(P.hist_lds);
        const int n16 = (P.hist_rid_bytes + P.hist_distinct * K * (int)sizeof(float)) / 16;
        for (int i = tid; i < n16; i += 256) s_prior_rows[i] = src[i];
        __syncthreads();
    }
    const int c = blockIdx.x * 4 + (tid >> 6), N = P.N;
    if (c * 256 >= N) return;
    const int a = D.action[e], o = D.obs[e], ninc = model_ninc(P);
    const size_t sb = pbase(P, e, D.bufsel[e]);
    double* sw = D.p_weight + sb;
    float* scn = D.p_rec + rec_base(P, D, e, D.bufsel[e]) * (size_t)P.Cs;
    const uint32_t hist_cnt = HIST ? D.hist_cnt[e] : 0u;
    Rng g = slot_rng(P, D, e);
    const int i0 = c * 256 + lane * 4;
    double sum = 0;
    for (int k = 0; k < 4; ++k) {
        const int i = i0 + k;
        double v = 0.0;
        if (i < N) {
            g.stream(FBA_PHASE_IS_UPDATE, (uint32_t)i);
            float* cnt = scn + (size_t)i * (HIST ? hist_stride(P, hist_total(hist_cnt)) : P.Cs);
            int s = rec_state(cnt, P.C), so;
            double r;
            if (HIST) {
                const uint32_t* rec = reinterpret_cast<const uint32_t*>(cnt);
                uint32_t sp = (rec[1] >> 16) & 0x3ffu, entry;
                double prob;
                if (K > 0) {
                    const HistRowsLds<(K > 0 ? K : 8)> rl{reinterpret_cast<const uint8_t*>(s_prior_rows),
                                                           reinterpret_cast<const float*>(reinterpret_cast<const char*>(s_prior_rows) + P.hist_rid_bytes),
                                                           HistRowIds(P.gw_N, P.gw_G, 4)};
                    gridworld_hist_step(P, g, rl, rec + 2 + hist_offset(hist_cnt, a), hist_count(hist_cnt, a), rec[1], sp, a, so, r, entry, o, prob);
                } else
                    gridworld_hist_step(P, g, rec + 2 + hist_offset(hist_cnt, a), hist_count(hist_cnt, a), rec[1], sp, a, so, r, entry, o, prob);
                *reinterpret_cast<int2*>(D.p_side + ((size_t)e * N + i) * 2) = make_int2(gridworld_unpack_state(P, sp), (int)entry);
                v     = sw[i] * prob;
                sw[i] = v;
                sum   = (k == 0) ? v : sum + v;
                continue;
            }
            sim_step<REG>(P, g, GlobalSearchView{cnt}, s, a, so, r, LdsInc<256>{s_inc + tid});
            if (defer_increments(P)) {  // see importance_kernel
                int32_t* sd = D.p_side + ((size_t)e * N + i) * D.side_w;
                sd[0] = s;
                for (int q = 0; q < ninc; ++q) sd[1 + q] = s_inc[q * 256 + tid];
                v = sw[i] * sim_obs_prob<REG>(P, g, PendingIncView<GlobalView>{GlobalView{cnt}, s_inc + tid, 256, ninc}, s, a, o);
            } else {
                for (int q = 0; q < ninc; ++q) cnt[s_inc[q * 256 + tid]] += 1.0f;
                rec_set_state(cnt, P.C, s);
                v = sw[i] * sim_obs_prob<REG>(P, g, GlobalView{cnt}, s, a, o);
            }
            sw[i] = v;
        }
        sum = (k == 0) ? v : sum + v;
    }
    const double incl = wave_inclusive_scan(sum, lane);
    if (lane == 63) D.ctot[(size_t)e * D.ctot_stride + c + 1] = incl;
}

__global__ void __launch_bounds__(256) is_multi_norm_kernel(Problem P, DeviceState D)
{
    const int e = chunk_slot(D, blockIdx.y), lane = threadIdx.x & 63;
    if (!D.need_update[e] || hist_update_refused(P, D, e)) return;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), N = P.N;
    if (c * 256 >= N) return;
    double* sw = D.p_weight + pbase(P, e, D.bufsel[e]);
    const double total = D.is_tot[2 * e + 0];
    const int i0 = c * 256 + lane * 4;
    double sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // the normalised weight is not stored: is_multi_scan_kernel forms the same quotient again, and the resampled
        // filter's weights are 1 / N whatever it was
        const double v = (i0 + k < N) ? sw[i0 + k] / total : 0.0;
        sum = (k == 0) ? v : sum + v;
    }
    const double incl = wave_inclusive_scan(sum, lane);
    if (lane == 63) D.ctot[(size_t)e * D.ctot_stride + c + 1] = incl;
}

__global__ void __launch_bounds__(256) is_multi_resample_kernel(Problem P, DeviceState D)
{
    __shared__ int32_t s_src[256];
    const int e = chunk_slot(D, blockIdx.y), tid = threadIdx.x;
    if (!D.need_update[e] || hist_update_refused(P, D, e)) return;
    const int N = P.N, j0 = blockIdx.x * 256, j = j0 + tid;
    const int cur = D.bufsel[e];
    const size_t sb = pbase(P, e, cur), db = pbase(P, e, cur ^ 1);
    const double* wscan = D.wscan + (size_t)e * N;
    const int C4 = P.Cs / 4, group = record_group(C4);
    if (j < N) {
        Rng g = slot_rng(P, D, e);
        g.stream(FBA_PHASE_RESAMPLE, (uint32_t)j);
        s_src[tid] = weighted_pick_guided(wscan, N, g.u01() * D.is_tot[2 * e + 1], D.is_tot[2 * e + 1]);
        D.p_weight[db + j] = 1.0 / (double)N;
    }
    __syncthreads();
    if (P.hist) {   // the source record with its particle's pending entry inserted (gather_hist_records), into the slot's scratch place
        const uint32_t hist_cnt = D.hist_cnt[e];
        const int a = D.action[e], hist_n = hist_total(hist_cnt), rs4 = hist_stride(P, hist_n) / 4, rd = hist_stride(P, hist_n + 1);
        gather_hist_records(rec_dst(P, D, e, cur ^ 1) + (size_t)j0 * rd, D.p_rec + rec_base(P, D, e, cur) * (size_t)P.Cs, s_src, D.p_side + (size_t)e * N * 2,
                            hist_n, 2 + hist_offset(hist_cnt, a) + hist_count(hist_cnt, a), min(256, N - j0), rs4, record_group(rd / 4), 256, rd / 4);
        return;
    }
    if (defer_increments(P))
        gather_records_side(D.p_rec + (db + j0) * (size_t)P.Cs, D.p_rec + sb * (size_t)P.Cs, s_src, D.p_side + (size_t)e * N * D.side_w, D.side_w,
                            min(256, N - j0), C4, P.C, group, 256);
    else
        gather_records(D.p_rec + (db + j0) * (size_t)P.Cs, D.p_rec + sb * (size_t)P.Cs, nullptr, s_src, nullptr, 0, 0, nullptr,
                       min(256, N - j0), C4, P.C, group, 256);
}

__global__ void __launch_bounds__(256) is_multi_scan_kernel(Problem P, DeviceState D)
{
    const int e = chunk_slot(D, blockIdx.y);
    if (!D.need_update[e] || hist_update_refused(P, D, e)) return;
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6), N = P.N;
    if (c * 256 >= N) return;
    const double* sw = D.p_weight + pbase(P, e, D.bufsel[e]);
    const double total = D.is_tot[2 * e + 0];
    const int i0 = c * 256 + lane * 4;
    double x[4], s4 = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // the normalised weights (WeightedFilter::normalize), as is_multi_norm_kernel summed them
        x[k] = (i0 + k < N) ? sw[i0 + k] / total : 0.0;
        s4   = (k == 0) ? x[k] : s4 + x[k];
    }
    const double incl = wave_inclusive_scan(s4, lane);
    double excl       = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 0.0;
    double run  = D.ctot[(size_t)e * D.ctot_stride + c] + excl;
    double* out = D.wscan + (size_t)e * N;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i0 + k < N) {
            run += x[k];
            out[i0 + k] = run;
        }
}

__global__ void is_multi_finish_kernel(Problem P, DeviceState D, int count)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= count) return;
    const int e = chunk_slot(D, b);
    if (!D.need_update[e]) return;
    if (P.hist) {
        const uint32_t hist_cnt = D.hist_cnt[e];
        if (hist_total(hist_cnt) >= P.hist_cap) {   // (as importance_kernel<HIST>)
            atomicCAS(D.fault, 0, 0x40000000 + e);
            D.need_update[e] = 0;
            D.active[e]      = 0;
            return;
        }
        D.hist_cnt[e] = hist_cnt + (1u << (8 * D.action[e]));
        D.upd_entries[e] += (unsigned long long)P.N * (unsigned long long)hist_total(hist_cnt);
        if (D.single_rec) D.copy_pending[e] = 1;
    }
    D.bufsel[e] ^= 1;
    D.need_update[e] = 0;
    D.belief_steps[e] += (unsigned long long)P.N;
    D.upd_attempts[e] += (unsigned long long)P.N;
    D.upd_particles[e] += (unsigned long long)P.N;
    D.cur[e].update_count = -1;
    D.cur[e].weight_total = D.is_tot[2 * e + 0];
}

// ---------------------------------------------------------------------------------------------
// init_kernel: Belief::initiate -- N x simulator.sampleStartState()
// (RejectionSampling.cpp:15-20, ImportanceSampler.cpp:45-55; BAPOMDP::sampleStartState
// BAPOMDP.cpp:101-104 = prior->sample(domain start state)).  One workgroup per slot.
// ---------------------------------------------------------------------------------------------
// `fc` = 1: the reinvigoration belief's fully connected filter (ReinvigoratingRejectionSampling.cpp:
// 55-76: N x FBAPOMDP::sampleFullyConnectedState after the N start states).
__global__ void __launch_bounds__(256) init_kernel(Problem P, DeviceState D, int fc)
{
    const int e = blockIdx.y, tid = threadIdx.x;
    if (!D.need_init[e]) return;
    // this workgroup's tile of the slot's particles (PARTICLE_TILE each, so big filters spread over the chip)
    const int i_lo = blockIdx.x * PARTICLE_TILE, i_hi = min(P.N, i_lo + PARTICLE_TILE);
    if (i_lo >= P.N) return;
    const size_t pb = pbase(P, e, fc ? D.bufsel_fc[e] : D.bufsel[e]);
    float* recs     = (fc ? D.p_rec_fc : D.p_rec) + (fc ? pb : rec_base(P, D, e, D.bufsel[e])) * (size_t)P.Cs;
    Rng g = slot_rng(P, D, e);
    g.position((uint32_t)D.run[e], 0, 0);
    if (P.hist) {
        // history particles: a fresh particle is its start state and its structure draws, no entries
        // (GridWorldFactBAPrior::sampleFBAPOMDPState GridWorldBAPriors.cpp:415-441: one boolean per action and x / y node)
        const double w1h = 1.0 / (double)P.N;
        for (int i = i_lo + tid; i < i_hi; i += 256) {
            g.stream(FBA_PHASE_INIT, (uint32_t)i);
            uint32_t* rec = reinterpret_cast<uint32_t*>(recs + (size_t)i * hist_stride(P, 0));
            const int s0 = domain_start(P, g);
            rec[0] = (uint32_t)s0;
            uint32_t mask = 0;
            if (P.structure_prior == FBA_SP_MATCH_UNIFORM)
                for (int k = 0; k < 2 * P.A; ++k)
                    if (g.boolean()) mask |= 1u << k;
            rec[1] = mask | (gridworld_pack_state(P, s0) << 16);
            D.p_weight[pb + i] = w1h;
        }
        return;
    }
    if (P.ft_packed) {
        // packed factored-tiger particles: no increments yet, the structure the prior draws (factored_prior_sample:
        // FactoredTigerPriors.cpp:197-219, 265-291), the start state
        const int FS = P.fd->FS, maskw = (8 * FS + 4 + (2 << FS)) / 2;
        for (int i = i_lo + tid; i < i_hi; i += 256) {
            g.stream(FBA_PHASE_INIT, (uint32_t)i);
            uint32_t* rec = reinterpret_cast<uint32_t*>(recs + (size_t)i * P.Cs);
            for (int k = 0; k < maskw; ++k) rec[k] = 0;
            rec[P.C] = (uint32_t)domain_start(P, g);
            uint32_t mask = 1u;   // the correct structure: the tiger's door
            if (P.structure_prior == FBA_SP_FULLY_CONNECTED) mask = (1u << FS) - 1u;
            else if (P.structure_prior == FBA_SP_UNIFORM || P.structure_prior == FBA_SP_MATCH_UNIFORM) {
                mask = 0;
                for (int f = 0; f < FS; ++f)
                    if (g.boolean()) mask |= 1u << f;
                if (P.structure_prior == FBA_SP_MATCH_UNIFORM) mask |= 1u;
            }
            rec[maskw] = mask;
        }
        return;
    }
    // every particle starts from the prior record ...
    const int C4 = P.Cs / 4;
    const float4* pr = reinterpret_cast<const float4*>(D.prior);
    float4* dp       = reinterpret_cast<float4*>(recs) + (size_t)i_lo * C4;
    const size_t tot = (size_t)(i_hi - i_lo) * C4;
    for (size_t f = tid; f < tot; f += 256) dp[f] = pr[f % C4];
    __syncthreads();
    // ... and its own domain start state
    const double w1 = 1.0 / (double)P.N;
    if (P.cheat && !fc && blockIdx.x == 0 && tid == 0) D.lik[e] = 1.0;  // _likelihood = 1
    for (int i = i_lo + tid; i < i_hi; i += 256) {
        g.stream(fc ? FBA_PHASE_INIT_FC : FBA_PHASE_INIT, (uint32_t)i);
        rec_set_state(recs + (size_t)i * P.Cs, P.C, domain_start(P, g));
        if (fc && P.cheat) continue;  // CheatingReinvigoration::initiate: sampleCorrectGraphState = the base prior record
        if (fc) {  // sampleFullyConnectedState: FactoredTigerPriors.cpp:324-337, CollisionAvoidancePriors.cpp:429-440, SysAdminFactoredPrior.cpp:57-69
            if (dom_is_sys(P.domain)) {
                sys_fill_fully_connected(P, recs + (size_t)i * P.Cs);
            } else if (dom_is_ca(P.domain)) {
                for (int a = 0; a < P.A; ++a)
                    for (int f = 2; f < P.fd->FS; ++f) ca_fill_obstacle_node(P, recs + (size_t)i * P.Cs, a, f, (1u << P.fd->FS) - 1u);
            } else {
                ftiger_set_observation_model(P, recs + (size_t)i * P.Cs, (1u << P.fd->FS) - 1u);
            }
            continue;
        }
        if (P.model == FBA_MODEL_BA_FACTORED) factored_prior_sample(P, g, recs + (size_t)i * P.Cs);
        if (P.belief == FBA_BELIEF_IMPORTANCE) D.p_weight[pb + i] = w1;
    }
}

// ---------------------------------------------------------------------------------------------
// reset_kernel: BABelief::resetDomainStateDistribution.
// Rejection filter (BARejectionSampling.cpp:49-60): every particle keeps its counts and gets a
// fresh domain start state.  Importance filter (BAImportanceSampling.cpp:90-111): N particles are
// re-drawn from the (uniformly weighted) filter, copied, and given a fresh start state.
// ---------------------------------------------------------------------------------------------
// `fc` = 1: the reinvigoration belief's fully connected filter (ReinvigoratingRejectionSampling.cpp:109-119).
__global__ void __launch_bounds__(256) reset_kernel(Problem P, DeviceState D, int fc)
{
    __shared__ int32_t s_src[256], s_ns[256];
    const int e = chunk_slot(D, blockIdx.y), tid = threadIdx.x;
    if (D.need_reset[e] != 1) return;
    const int i_lo = blockIdx.x * PARTICLE_TILE, i_hi = min(P.N, i_lo + PARTICLE_TILE);
    if (i_lo >= P.N) return;
    const int cur = fc ? D.bufsel_fc[e] : D.bufsel[e];
    const size_t sb = pbase(P, e, cur), db = pbase(P, e, cur ^ 1);
    Rng g = slot_rng(P, D, e);
    g.position((uint32_t)D.run[e], (uint32_t)D.episode[e], 0);
    if (P.belief == FBA_BELIEF_REJECTION || fc || P.cheat || P.mh) {  // (the cheating and the mh-within-gibbs beliefs reset their weighted filter in place too, CheatingReinvigoration.cpp:48-62, MHwithinGibbs.cpp:259-275)
        float* recs = (fc ? D.p_rec_fc : D.p_rec) + sb * (size_t)P.Cs;
        for (int i = i_lo + tid; i < i_hi; i += 256) {
            g.stream(fc ? FBA_PHASE_RESET_FC : (D.shadow ? FBA_PHASE_RESET_SH : FBA_PHASE_RESET), (uint32_t)i);
            rec_set_state(recs + (size_t)i * P.Cs, P.C, domain_start(P, g));
        }
        return;
    }
    const int hist_n = P.hist ? hist_total(D.hist_cnt[e]) : 0;
    const int rs = P.hist ? hist_stride(P, hist_n) : P.Cs;
    const int C4 = rs / 4, group = record_group(C4);
    const double w1 = 1.0 / (double)P.N;
    for (int j0 = i_lo; j0 < i_hi; j0 += 256) {
        const int j = j0 + tid;
        if (j < i_hi) {
            g.stream(FBA_PHASE_RESET, (uint32_t)j);
            s_src[tid] = uniform_weight_pick(D.uni_scan, P.N, g.u01() * D.uni_total, D.uni_total);
            s_ns[tid]  = domain_start(P, g);
            D.p_weight[db + j] = w1;
        }
        __syncthreads();
        const int m = min(256, i_hi - j0);
        if (P.hist) {
            // the copy of the drawn particle (state, structure bits, entries) with its new start state
            const int gid = tid / group, part0 = tid % group, ngroups = 256 / group, n4 = (hist_n + 5) >> 2;
            for (int q = gid; q < m; q += ngroups) {
                const float4* sp4 = reinterpret_cast<const float4*>(D.p_rec + rec_base(P, D, e, cur) * (size_t)P.Cs) + (size_t)s_src[q] * C4;
                float4* dp4       = reinterpret_cast<float4*>(rec_dst(P, D, e, cur ^ 1) + (size_t)j0 * rs) + (size_t)q * C4;
                const uint32_t nsp = gridworld_pack_state(P, s_ns[q]);
                for (int part = part0; part < n4; part += group) {
                    float4 v = sp4[part];
                    if (part == 0) {
                        v.x = __int_as_float(s_ns[q]);
                        v.y = __uint_as_float((__float_as_uint(v.y) & 0xffffu) | (nsp << 16));
                    }
                    dp4[part] = v;
                }
            }
            if (D.single_rec && tid == 0) D.copy_pending[e] = 1;
            __syncthreads();
            continue;
        }
        gather_records(D.p_rec + (db + j0) * (size_t)P.Cs, D.p_rec + sb * (size_t)P.Cs, nullptr, s_src, nullptr, 0, 0, s_ns, m, C4,
                       P.C, group, 256);
        __syncthreads();
    }
}

// Plain rejection filter: flag the reset instead of performing it (lazy_state).
__global__ void lazy_reset_kernel(Problem P, DeviceState D)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.E || D.need_reset[e] != 1) return;
    D.lazy_reset[e] = 1;
    D.need_reset[e] = 0;
}
// ... and perform it after all, for a caller that wants to see the records (fba_belief_get / _set).
__global__ void __launch_bounds__(256) materialize_reset_kernel(Problem P, DeviceState D)
{
    const int e = blockIdx.y, tid = threadIdx.x;
    if (!D.lazy_reset[e]) return;
    const int i_lo = blockIdx.x * PARTICLE_TILE, i_hi = min(P.N, i_lo + PARTICLE_TILE);
    float* recs = D.p_rec + pbase(P, e, D.bufsel[e]) * (size_t)P.Cs;
    for (int i = i_lo + tid; i < i_hi; i += 256) rec_set_state(recs + (size_t)i * P.Cs, P.C, lazy_state(P, D, e, i));
}
__global__ void post_materialize_kernel(Problem P, DeviceState D)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < P.E) D.lazy_reset[e] = 0;
}

// clears the request flags after init_kernel / reset_kernel (several workgroups serve one slot)
__global__ void post_init_kernel(Problem P, DeviceState D)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.E) return;
    if (P.hist && D.need_init[e]) D.hist_cnt[e] = 0;
    if (P.mh && D.need_init[e]) {  // MHwithinGibbs::initiate :277-294
        D.mh_n_ep[e] = 1;
        D.mh_ep_len[(size_t)e * (P.episodes + 1)] = 0;
        D.lik[e] = 0.0;
    }
    D.need_init[e] = 0;
}
__global__ void post_reset_kernel(Problem P, DeviceState D)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.E || D.need_reset[e] != 1) return;
    if (P.belief == FBA_BELIEF_IMPORTANCE && !P.cheat && !P.mh && !P.nested) D.bufsel[e] ^= 1;
    if (P.mh) {  // MHwithinGibbs::resetDomainStateDistribution :270-274: a new episode unless the open one is still empty
        int32_t* len = D.mh_ep_len + (size_t)e * (P.episodes + 1);
        const int n = D.mh_n_ep[e];
        if (len[n - 1] != 0 && n <= P.episodes) { len[n] = 0; D.mh_n_ep[e] = n + 1; }
    }
    D.need_reset[e] = 0;
}

// ---------------------------------------------------------------------------------------------
// flush_kernel: belief checksum (sum over particles of a position-keyed hash, so the order of
// the additions does not matter) and hand-over of the tick's trace record.
// ---------------------------------------------------------------------------------------------
// History particles: the same checksum over the particle's whole count table -- walked cell by cell in the dense
// layout, prior value plus the number of entries that incremented the cell -- without ever building the table.
// hist_next_cell: the smallest incremented cell index >= kmin and how many entries incremented it.
__device__ void hist_next_cell(const Problem& P, const uint32_t* rec, uint32_t cnt, int kmin, int& nxt, int& mult)
{
    const int N = P.gw_N, G = P.gw_G, A = P.A;
    const int XY = N * N * G * N, GG = N * N * G * G, NN = N * N;
    const uint32_t mask = rec[1];
    nxt = 0x7fffffff; mult = 0;
    int j = 0;
    for (int a = 0; a < A; ++a) {
        const int tbase = a * (2 * XY + GG), obase = A * (2 * XY + GG) + a * (2 * NN + G * G);
        const bool mx = (mask >> (2 * a)) & 1u, my = (mask >> (2 * a + 1)) & 1u;
        for (int q = 0; q < hist_count(cnt, a); ++q, ++j) {
            const uint32_t en = rec[2 + j], s0 = en & 0x3ffu, s1 = (en >> 10) & 0x3ffu, ob = en >> 20;
            const int x = hist_x(s0), y = hist_y(s0), gl = hist_g(s0), cell = x * N + y;
            const int c[6] = {tbase + (mx ? cell * G + gl : cell) * N + hist_x(s1),
                              tbase + XY + (my ? cell * G + gl : cell) * N + hist_y(s1),
                              tbase + 2 * XY + (cell * G + gl) * G + hist_g(s1),
                              obase + x * N + hist_x(ob),
                              obase + NN + y * N + hist_y(ob),
                              obase + 2 * NN + gl * G + hist_g(ob)};
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (c[k] >= kmin) {
                    if (c[k] < nxt) { nxt = c[k]; mult = 1; }
                    else if (c[k] == nxt) ++mult;
                }
        }
    }
}
__device__ uint64_t hist_hash_counts(const Problem& P, const uint32_t* rec, uint32_t len, uint64_t h)
{
    const HistLayout L(P.gw_N, P.gw_G, P.A);
    const int N = L.N, G = L.G, A = L.A, rows = N * N * G;
    const uint32_t mask = rec[1];
    int k = 0, nxt, mult;  // k: the cell's index in the dense table (the order the checksum is defined in)
    hist_next_cell(P, rec, len, 0, nxt, mult);
    auto visit = [&](float prior) {
        float v = prior;
        if (k == nxt) {
            v = prior + (float)mult;
            hist_next_cell(P, rec, len, k + 1, nxt, mult);
        }
        h = mix64(h ^ ((uint64_t)__float_as_uint(v) + ((uint64_t)k << 32)));
        ++k;
    };
    for (int a = 0; a < A; ++a) {
        for (int f = 0; f < 2; ++f) {
            const bool with_goal = (mask >> (2 * a + f)) & 1u;
            for (int row = 0; row < rows; ++row) {  // the dense node has room for N*N*G rows; without the goal parent N*N are in use, the rest zero
                const float* src = with_goal ? P.hist_alt + (size_t)(a * 2 + f) * L.XY + row * L.NS
                                             : (row < N * N ? P.hist_base + a * L.tstride + f * L.XY + row * L.NS : nullptr);
                for (int i = 0; i < N; ++i) visit(src ? src[i] : 0.f);
            }
        }
        for (int row = 0; row < rows; ++row)
            for (int i = 0; i < G; ++i) visit(P.hist_base[a * L.tstride + 2 * L.XY + row * L.GS + i]);
    }
    for (int a = 0; a < A; ++a)
        for (int f = 0; f < 3; ++f) {
            const int n = f == 2 ? G : N;
            for (int v = 0; v < n; ++v)
                for (int i = 0; i < n; ++i) visit(P.hist_base[L.o_row(a, f, v) + i]);
        }
    for (int w = 0; w < 2 * A; ++w, ++k) h = mix64(h ^ ((uint64_t)(((mask >> w) & 1u) ? 7u : 3u) + ((uint64_t)k << 32)));
    return h;
}

__global__ void __launch_bounds__(256) flush_kernel(Problem P, DeviceState D)
{
    __shared__ unsigned long long s_sum;
    __shared__ uint32_t s_hist[FBA_TRACE_HIST_BINS];
    const int e = blockIdx.x, tid = threadIdx.x;
    if (!D.trace_on || D.cur[e].belief_hash != 1) return;
    if (tid == 0) s_sum = 0;
    if (tid < FBA_TRACE_HIST_BINS) s_hist[tid] = 0;
    const bool hist_on = D.trace_hist != nullptr && !D.cur[e].terminal;   // (no belief update after a terminal step, Episode.cpp:47-50)
    __syncthreads();
    const size_t pb = pbase(P, e, D.bufsel[e]);
    const bool lazy = slot_lazy(D, e);
    unsigned long long local = 0;
    for (int i = tid; i < P.N; i += 256) {
        const float* cnt = P.hist ? D.p_rec + rec_base(P, D, e, D.bufsel[e]) * (size_t)P.Cs + (size_t)i * hist_stride(P, hist_total(D.hist_cnt[e]))
                                  : D.p_rec + (rec_base(P, D, e, D.bufsel[e]) + i) * (size_t)P.Cs;
        const int st = lazy ? lazy_state(P, D, e, i) : rec_state(cnt, P.C);
        if (hist_on && (unsigned)st < (unsigned)FBA_TRACE_HIST_BINS) atomicAdd(&s_hist[st], 1u);
        uint64_t h = mix64((uint64_t)i * 0x9E3779B97F4A7C15ull + (uint64_t)(uint32_t)st);
        const double w = (P.belief == FBA_BELIEF_IMPORTANCE) ? D.p_weight[pb + i] : 0.0;
        h = mix64(h ^ (uint64_t)__double_as_longlong(w));
        if (P.ft_packed) {  // the counts themselves, then the mask word, as the dense blob has them
            const int FS = P.fd->FS, nc = 8 * FS + 4 + (2 << FS);
            for (int k = 0; k <= nc; ++k) {
                float v;
                if (FS == 2) v = packed_ftiger_view<2>(P, GlobalView{cnt}).at(k);
                else if (FS == 3) v = packed_ftiger_view<3>(P, GlobalView{cnt}).at(k);
                else v = packed_ftiger_view<4>(P, GlobalView{cnt}).at(k);
                h = mix64(h ^ ((uint64_t)__float_as_uint(v) + ((uint64_t)k << 32)));
            }
        } else if (P.hist) h = hist_hash_counts(P, reinterpret_cast<const uint32_t*>(cnt), D.hist_cnt[e], h);
        else if (P.packed) {  // the checksum is over the counts themselves, whatever the storage (PackedView)
            const PackedView<GlobalView> pv{GlobalView{cnt}, D.prior_dense};
            const int dense = P.phi_len + P.A * P.S * P.O;
            for (int k = 0; k < dense; ++k) h = mix64(h ^ ((uint64_t)__float_as_uint(pv.at(k)) + ((uint64_t)k << 32)));
        } else
            for (int k = 0; k < P.C; ++k) h = mix64(h ^ ((uint64_t)__float_as_uint(cnt[k]) + ((uint64_t)k << 32)));
        local += h;
    }
    if (P.nested) {  // + every domain state of every flat filter, keyed by its position
        const int32_t* st = D.nest_s + nest_base(P, e, D.nest_sel[e]);
        const size_t tot  = (size_t)P.N * (size_t)P.nested;
        for (size_t k = tid; k < tot; k += 256) local += mix64(((uint64_t)P.N + k) * 0x9E3779B97F4A7C15ull + (uint64_t)(uint32_t)st[k]);
    }
    atomicAdd(&s_sum, local);
    __syncthreads();
    if (tid == 0) {
        D.cur[e].belief_hash = s_sum;
        const int idx = atomicAdd(D.trace_count, 1);
        if (idx < D.trace_cap) D.trace[idx] = D.cur[e];
        s_sum = (unsigned long long)(unsigned)idx;
    }
    if (D.trace_hist) {
        __syncthreads();
        const size_t idx = (size_t)s_sum;
        if (idx < (size_t)D.trace_cap && tid < FBA_TRACE_HIST_BINS) D.trace_hist[idx * FBA_TRACE_HIST_BINS + tid] = hist_on ? s_hist[tid] : 0u;
    }
}

// diagnostics: det_lgamma, and BABNModel::LogBDScore of one particle blob against another (one lane: the sum is sequential)
__global__ void selftest_lgamma_kernel(const double* x, int count, double* out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = det_lgamma(x[i]);
}
__global__ void selftest_bd_kernel(Problem P, const float* cnt, const float* prior, double* out)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *out = log_bd_score(P, GlobalView{cnt}, GlobalView{prior});
}

// diagnostic: the UCB bonus expression exactly as ucb_select evaluates it
__global__ void selftest_ucb_kernel(const double* L, const int32_t* n, int count, double u, double* out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = u * sqrt(L[i] / (double)n[i]);
}

// DeviceState::single_rec: the filter a resample / reset has built in its scratch place's buffer becomes the slot's filter -- the two
// buffer indices change places (one thread per slot of the chunk)
__global__ void __launch_bounds__(256) swap_buffers_kernel(Problem P, DeviceState D, int count)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= count) return;
    const int e = chunk_slot(D, b);
    if (!D.copy_pending[e]) return;
    const int place = P.E + scratch_place(D, e);
    const int32_t mine = D.rec_buf[e];
    D.rec_buf[e]     = D.rec_buf[place];
    D.rec_buf[place] = mine;
}
__global__ void copy_flags_kernel(const uint8_t* src, uint8_t* dst, int n)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) dst[e] = src[e];
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
void launch_start(const Problem& P, const DeviceState& D, hipStream_t st)
{
    hipLaunchKernelGGL(start_kernel, dim3(ceil_div(P.E, 256)), dim3(256), 0, st, P, D);
}
void launch_env(const Problem& P, const DeviceState& D, int32_t* n_active, hipStream_t st)
{
    hipLaunchKernelGGL(env_kernel, dim3(ceil_div(P.E, 256)), dim3(256), 0, st, P, D, n_active);
}
void launch_advance(const Problem& P, const DeviceState& D, int32_t* n_active, hipStream_t st)
{
    hipLaunchKernelGGL(advance_kernel, dim3(ceil_div(P.E, 256)), dim3(256), 0, st, P, D, n_active);
}
// importance_kernel of one slot per workgroup, the instantiation the problem calls for
static void launch_importance_single(const Problem& P, const DeviceState& D, hipStream_t st)
{
    const bool tiger_table = P.model == FBA_MODEL_BA_TABLE && !P.dirichlet_regular &&
                             (P.domain == FBA_DOM_TIGER_EPISODIC || P.domain == FBA_DOM_TIGER_CONTINUOUS);
    // weights and prefix sums of a slot in LDS while its workgroup works on them (8 bytes per particle)
    const bool wlds = P.N <= IS_LDS_MAX_N && !P.dirichlet_regular;
    const size_t wl = (wlds ? (size_t)P.N * (sizeof(double) + ((tiger_table && P.packed) ? 4 : 0)) : 0) +  // (packed tiger: + pending updates and sources, 16 bits each)
                      ((P.hist && hist_all_draws_first(P.N)) ? (size_t)P.N * 2 : 0);                       // (history particles: the drawn sources)
    const int grid_e = D.use_list ? D.scratch_slots /* (set to the chunk's count by the caller) */
                                  : (D.single_rec ? std::min(D.scratch_slots, P.E - D.slot_base) : P.E);   // (single_rec: one chunk of slots per launch)
#define FBA_LAUNCH_IS(...)                                                                                         \
    do {                                                                                                           \
    static unsigned long long raised = 0;   /* one bit per device: function attributes are per device */          \
    int dev_ = 0;                                                                                              \
    (void)hipGetDevice(&dev_);                                                                                 \
    if (wl > 16384 && !((raised >> (dev_ & 63)) & 1ull)) {                                                      \
        /* (a refused attribute leaves the bit clear and the launch below fails: the engine reads hipGetLastError after every tick / update) */ \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&importance_kernel<__VA_ARGS__>),                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, IS_LDS_MAX_N * 12) == hipSuccess)  \
            raised |= 1ull << (dev_ & 63);                                                                     \
    }                                                                                                          \
    hipLaunchKernelGGL((importance_kernel<__VA_ARGS__>), dim3(grid_e), dim3(IS_BLOCK), wl, st, P, D);          \
    } while (0)
    if (P.hist) { if (wlds) FBA_LAUNCH_IS(false, 0, true, true); else FBA_LAUNCH_IS(false, 0, true, false); }
    else if (P.dirichlet_regular) FBA_LAUNCH_IS(true, 0, false, false);
    else if (tiger_table && P.packed) { if (wlds) FBA_LAUNCH_IS(false, 2, false, true); else FBA_LAUNCH_IS(false, 2, false, false); }
    else if (tiger_table) { if (wlds) FBA_LAUNCH_IS(false, 1, false, true); else FBA_LAUNCH_IS(false, 1, false, false); }
    else { if (wlds) FBA_LAUNCH_IS(false, 0, false, true); else FBA_LAUNCH_IS(false, 0, false, false); }
#undef FBA_LAUNCH_IS
}
// the slots whose flag (need_update / need_reset == 1) is set, compacted; a listed slot's scratch place is its position in its chunk
__global__ void __launch_bounds__(256) build_slot_list_kernel(Problem P, DeviceState D, const uint8_t* flag)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= P.E || flag[e] != 1) return;
    const int pos      = atomicAdd(D.list_count, 1);
    D.slot_list[pos]   = e;
    D.scratch_idx[e]   = pos % D.scratch_slots;
}
// a HIP error met inside a launch_* function that has no status to return (the engine's timed() asks after every launch)
static thread_local hipError_t g_launch_err = hipSuccess;
hipError_t launch_take_error()
{
    const hipError_t e = g_launch_err;
    g_launch_err = hipSuccess;
    return e;
}
// DeviceState::single_rec: the slots in chunks of as many as the scratch pool holds -- gather into the pool, copy back, next chunk
template <class F>
static void for_each_chunk(const Problem& P, const DeviceState& D, hipStream_t st, F launch, const uint8_t* flag = nullptr)
{
    if (D.slot_list && flag) {
        // budgeted searches: few slots have work, anywhere among all of them -- chunks of the compacted list instead of E / scratch_slots
        // nearly empty launches (C4: 32 of them took 35 ms per round for a handful of updates).  The count is read on the host.
        hipError_t he = hipMemsetAsync(D.list_count, 0, sizeof(int32_t), st);
        hipLaunchKernelGGL(build_slot_list_kernel, dim3(ceil_div(P.E, 256)), dim3(256), 0, st, P, D, flag);
        int32_t n = 0;
        int32_t* dst = D.list_count_host ? D.list_count_host : &n;
        if (he == hipSuccess) he = hipMemcpyAsync(dst, D.list_count, sizeof n, hipMemcpyDeviceToHost, st);
        if (he == hipSuccess) he = hipStreamSynchronize(st);
        if (he == hipSuccess) n = *dst;
        if (he != hipSuccess) {   // the count is unknown: nothing was launched, the request flags stay set -- the engine must hear of it (timed())
            g_launch_err = he;
            return;
        }
        for (int i0 = 0; i0 < n; i0 += D.scratch_slots) {
            DeviceState Dc = D;
            Dc.slot_base   = i0;
            Dc.use_list    = 1;
            const int cnt  = std::min(D.scratch_slots, n - i0);
            launch(Dc, cnt);
            hipLaunchKernelGGL(swap_buffers_kernel, dim3(ceil_div(cnt, 256)), dim3(256), 0, st, P, Dc, cnt);
        }
        if (n > 0) (void)hipMemsetAsync(D.copy_pending, 0, (size_t)P.E, st);
        return;
    }
    for (int e0 = 0; e0 < P.E; e0 += D.scratch_slots) {
        DeviceState Dc = D;
        Dc.slot_base   = e0;
        const int cnt  = std::min(D.scratch_slots, P.E - e0);
        launch(Dc, cnt);
        hipLaunchKernelGGL(swap_buffers_kernel, dim3(ceil_div(cnt, 256)), dim3(256), 0, st, P, Dc, cnt);
        (void)hipMemsetAsync(D.copy_pending + e0, 0, (size_t)cnt, st);
    }
}
// the importance update of `count` slots (all of them, a chunk, or a chunk of the compacted list: chunk_slot) as seven launches, so that a slot's
// particles spread over many workgroups
static void launch_importance_multi(const Problem& P, const DeviceState& D, int count, hipStream_t st)
{
    const int nchunks = (P.N + 255) / 256;
    const dim3 cgrid(ceil_div(nchunks, 4), count), eg(ceil_div(count, 64));
    const int32_t* list = D.use_list ? D.slot_list : nullptr;
    if (P.hist) {
        // the prior's rows from LDS where the deduplicated blob exists (rows of K floats, K as upload_prior chose it) -- FBA_HIST_ROWS=hbm: from L2
        const bool rows_hbm = D.ab_rows_hbm != 0;
        const int K = P.hist_row <= 8 ? 8 : (P.hist_row <= 10 ? 12 : 0);
        const size_t lds = (size_t)P.hist_rid_bytes + (size_t)P.hist_distinct * K * sizeof(float);
        if (P.hist_lds && K == 8 && !rows_hbm) hipLaunchKernelGGL((is_multi_step_kernel<false, true, 8>), cgrid, dim3(256), lds, st, P, D);
        else if (P.hist_lds && K == 12 && !rows_hbm) hipLaunchKernelGGL((is_multi_step_kernel<false, true, 12>), cgrid, dim3(256), lds, st, P, D);
        else hipLaunchKernelGGL((is_multi_step_kernel<false, true>), cgrid, dim3(256), 0, st, P, D);
    }
    else if (P.dirichlet_regular) hipLaunchKernelGGL((is_multi_step_kernel<true, false>), cgrid, dim3(256), 0, st, P, D);
    else hipLaunchKernelGGL((is_multi_step_kernel<false, false>), cgrid, dim3(256), 0, st, P, D);
    hipLaunchKernelGGL(scan_carry_kernel, dim3(count), dim3(256), 0, st, D.ctot, D.ctot_stride, nchunks, D.is_tot, 2, 0, D.need_update, count, list, D.slot_base);
    hipLaunchKernelGGL(is_multi_norm_kernel, cgrid, dim3(256), 0, st, P, D);
    hipLaunchKernelGGL(scan_carry_kernel, dim3(count), dim3(256), 0, st, D.ctot, D.ctot_stride, nchunks, D.is_tot, 2, 1, D.need_update, count, list, D.slot_base);
    hipLaunchKernelGGL(is_multi_scan_kernel, cgrid, dim3(256), 0, st, P, D);
    hipLaunchKernelGGL(is_multi_resample_kernel, dim3(ceil_div(P.N, 256), count), dim3(256), 0, st, P, D);
    hipLaunchKernelGGL(is_multi_finish_kernel, eg, dim3(64), 0, st, P, D, count);
}
// history particles: several workgroups per slot from this many particles up (FBA_HIST_MULTI=1 / 0 forces / forbids it: tests, A/B runs)
static bool hist_update_multi(const Problem& P, const DeviceState& D)
{
    if (D.ab_hist_multi) return D.ab_hist_multi == 2;
    return P.N >= 4096;
}
void launch_belief_update(const Problem& P, const DeviceState& D, hipStream_t st)
{
    if (P.nested) {
        if (P.dirichlet_regular) hipLaunchKernelGGL(nested_update_kernel<true>, dim3(P.E), dim3(256), 0, st, P, D);
        else hipLaunchKernelGGL(nested_update_kernel<false>, dim3(P.E), dim3(256), 0, st, P, D);
        return;
    }
    if (P.belief == FBA_BELIEF_REJECTION) {
        const bool tiger_table = P.model == FBA_MODEL_BA_TABLE && !P.dirichlet_regular &&
                                 (P.domain == FBA_DOM_TIGER_EPISODIC || P.domain == FBA_DOM_TIGER_CONTINUOUS);
        if (P.reinvig) hipLaunchKernelGGL(reinvigorate_kernel, dim3(P.E), dim3(256), 0, st, P, D, 0);
        if (P.incub) {
            // StructureIncubatorSampling::updateEstimation (:105-131): breed the least likely shadow particles from the two
            // rejection filters as they are now, importance-sample the shadow filter (it has its own copy of the request
            // flags: its kernel clears them), then the two rejection updates below
            hipLaunchKernelGGL(copy_flags_kernel, dim3(ceil_div(P.E, 256)), dim3(256), 0, st, D.need_update, D.need_update_sh, P.E);
            hipLaunchKernelGGL(reinvigorate_kernel, dim3(P.E), dim3(256), 0, st, P, D, 1);
            Problem Ps = P;
            Ps.belief = FBA_BELIEF_IMPORTANCE; Ps.incub = 0;
            DeviceState Ds = D;
            Ds.p_rec = D.p_rec_sh; Ds.p_weight = D.p_weight_sh; Ds.bufsel = D.bufsel_sh; Ds.need_update = D.need_update_sh; Ds.shadow = 1;
            launch_importance_single(Ps, Ds, st);
        }
        const int ft = (P.model == FBA_MODEL_BA_FACTORED && !P.dirichlet_regular &&
                        (P.domain == FBA_DOM_FTIGER_EPISODIC || P.domain == FBA_DOM_FTIGER_CONTINUOUS))
                           ? 31 - __builtin_clz((unsigned)P.S) : 0;  // S = 2^FS
        for (int fc = (P.reinvig || P.incub) ? 1 : 0; fc >= 0; --fc) {  // the main filter's launch clears the request flag: last
            if (ft == 2 && P.ft_packed) hipLaunchKernelGGL((reject_kernel<false, 0, 2, REJECT_BLOCK, true>), dim3(P.E), dim3(REJECT_BLOCK), 0, st, P, D, fc);
            else if (ft == 3 && P.ft_packed) hipLaunchKernelGGL((reject_kernel<false, 0, 3, REJECT_BLOCK, true>), dim3(P.E), dim3(REJECT_BLOCK), 0, st, P, D, fc);
            else if (ft == 4 && P.ft_packed) hipLaunchKernelGGL((reject_kernel<false, 0, 4, REJECT_BLOCK, true>), dim3(P.E), dim3(REJECT_BLOCK), 0, st, P, D, fc);
            else if (ft == 2) hipLaunchKernelGGL((reject_kernel<false, 0, 2>), dim3(P.E), dim3(REJECT_BLOCK), 0, st, P, D, fc);
            else if (ft == 3) hipLaunchKernelGGL((reject_kernel<false, 0, 3>), dim3(P.E), dim3(REJECT_BLOCK), 0, st, P, D, fc);
            else if (ft == 4) hipLaunchKernelGGL((reject_kernel<false, 0, 4>), dim3(P.E), dim3(REJECT_BLOCK), 0, st, P, D, fc);
            else if (P.dirichlet_regular) hipLaunchKernelGGL((reject_kernel<true, 0>), dim3(P.E), dim3(REJECT_BLOCK), 0, st, P, D, fc);
            else if (tiger_table && P.packed && P.N <= TIGER_LDS_MAX_N)
                hipLaunchKernelGGL((reject_tiger_lds_kernel<512>), dim3(P.E), dim3(512), (size_t)P.N * 13 + 16, st, P, D);
            else if (tiger_table && P.packed)  // (chunks of 512 are 4 % faster than 256 / 1024 on packed records)
                hipLaunchKernelGGL((reject_kernel<false, 2, 0, 512>), dim3(P.E), dim3(512), 0, st, P, D, fc);
            else if (tiger_table) hipLaunchKernelGGL((reject_kernel<false, 1>), dim3(P.E), dim3(REJECT_BLOCK), 0, st, P, D, fc);
            else hipLaunchKernelGGL((reject_kernel<false, 0>), dim3(P.E), dim3(REJECT_BLOCK), 0, st, P, D, fc);
        }
        return;
    }
    if (P.cheat) {  // rejectSample on the correct-graph filter first (CheatingReinvigoration.cpp:111)
        if (P.dirichlet_regular) hipLaunchKernelGGL((reject_kernel<true, 0>), dim3(P.E), dim3(REJECT_BLOCK), 0, st, P, D, 1);
        else hipLaunchKernelGGL((reject_kernel<false, 0>), dim3(P.E), dim3(REJECT_BLOCK), 0, st, P, D, 1);
    }
    if (!D.is_multi) {
        if (D.single_rec)
            for_each_chunk(P, D, st, [&](const DeviceState& Dc, int cnt) {
                if (P.hist && hist_update_multi(P, D)) { launch_importance_multi(P, Dc, cnt, st); return; }
                if (!Dc.use_list) { launch_importance_single(P, Dc, st); return; }
                DeviceState Dl   = Dc;           // (list mode: the launch's grid is the chunk's count; scratch places were fixed against
                Dl.scratch_slots = cnt;          //  the pool's size when the list was built)
                launch_importance_single(P, Dl, st);
            }, D.need_update);
        else launch_importance_single(P, D, st);
        if (P.cheat) hipLaunchKernelGGL(cheat_kernel, dim3(P.E), dim3(256), 0, st, P, D);
        if (P.mh) {  // a wave per slot; the chain's scratch in LDS when it fits
            const int in_lds = mh_scratch_in_lds(D.mh_scratch_words);
            hipLaunchKernelGGL(mh_kernel, dim3(P.E), dim3(64), MH_LDS_HEAD + (in_lds ? (size_t)D.mh_scratch_words * 4 : 0), st, P, D, in_lds);
        }
        return;
    }
    launch_importance_multi(P, D, P.E, st);
}
void launch_init(const Problem& P, const DeviceState& D, hipStream_t st)
{
    hipLaunchKernelGGL(init_kernel, dim3(ceil_div(P.N, PARTICLE_TILE), P.E), dim3(256), 0, st, P, D, 0);
    if (P.reinvig || P.cheat || P.incub) hipLaunchKernelGGL(init_kernel, dim3(ceil_div(P.N, PARTICLE_TILE), P.E), dim3(256), 0, st, P, D, 1);
    if (P.incub) hipLaunchKernelGGL(reinvigorate_kernel, dim3(P.E), dim3(256), 0, st, P, D, 2);
    if (P.nested) hipLaunchKernelGGL(nested_fill_kernel, dim3(P.E), dim3(256), 0, st, P, D, 0);
    hipLaunchKernelGGL(post_init_kernel, dim3(ceil_div(P.E, 256)), dim3(256), 0, st, P, D);
}
void launch_materialize_reset(const Problem& P, const DeviceState& D, hipStream_t st)
{
    hipLaunchKernelGGL(materialize_reset_kernel, dim3(ceil_div(P.N, PARTICLE_TILE), P.E), dim3(256), 0, st, P, D);
    hipLaunchKernelGGL(post_materialize_kernel, dim3(ceil_div(P.E, 256)), dim3(256), 0, st, P, D);
}
void launch_reset(const Problem& P, const DeviceState& D, hipStream_t st)
{
    if (P.belief == FBA_BELIEF_REJECTION && !P.reinvig && !P.cheat && !P.incub) {  // the plain rejection filter resets lazily
        hipLaunchKernelGGL(lazy_reset_kernel, dim3(ceil_div(P.E, 256)), dim3(256), 0, st, P, D);
        return;
    }
    if (P.nested) {
        hipLaunchKernelGGL(nested_fill_kernel, dim3(P.E), dim3(256), 0, st, P, D, 1);
        hipLaunchKernelGGL(post_reset_kernel, dim3(ceil_div(P.E, 256)), dim3(256), 0, st, P, D);
        return;
    }
    if (D.single_rec)
        for_each_chunk(P, D, st, [&](const DeviceState& Dc, int cnt) {
            hipLaunchKernelGGL(reset_kernel, dim3(ceil_div(P.N, PARTICLE_TILE), cnt), dim3(256), 0, st, P, Dc, 0);
        }, D.need_reset);
    else hipLaunchKernelGGL(reset_kernel, dim3(ceil_div(P.N, PARTICLE_TILE), P.E), dim3(256), 0, st, P, D, 0);
    if (P.reinvig || P.cheat || P.incub) hipLaunchKernelGGL(reset_kernel, dim3(ceil_div(P.N, PARTICLE_TILE), P.E), dim3(256), 0, st, P, D, 1);
    if (P.incub) {  // StructureIncubatorSampling::resetDomainStateDistribution (:46-61): the shadow filter too, in place
        DeviceState Ds = D;
        Ds.p_rec = D.p_rec_sh; Ds.p_weight = D.p_weight_sh; Ds.bufsel = D.bufsel_sh; Ds.shadow = 1;
        hipLaunchKernelGGL(reset_kernel, dim3(ceil_div(P.N, PARTICLE_TILE), P.E), dim3(256), 0, st, P, Ds, 0);
    }
    hipLaunchKernelGGL(post_reset_kernel, dim3(ceil_div(P.E, 256)), dim3(256), 0, st, P, D);
}
void launch_flush(const Problem& P, const DeviceState& D, hipStream_t st)
{
    hipLaunchKernelGGL(flush_kernel, dim3(P.E), dim3(256), 0, st, P, D);
}
void launch_selftest_lgamma(const double* x, int count, double* out, hipStream_t st)
{
    hipLaunchKernelGGL(selftest_lgamma_kernel, dim3(ceil_div(count, 256)), dim3(256), 0, st, x, count, out);
}
void launch_selftest_bd(const Problem& P, const float* cnt, const float* prior, double* out, hipStream_t st)
{
    hipLaunchKernelGGL(selftest_bd_kernel, dim3(1), dim3(64), 0, st, P, cnt, prior, out);
}
void launch_selftest_ucb(const double* L, const int32_t* n, int count, double u, double* out, hipStream_t st)
{
    hipLaunchKernelGGL(selftest_ucb_kernel, dim3(ceil_div(count, 256)), dim3(256), 0, st, L, n, count, u, out);
}
// prefix sums of n uniform weights 1/n; ctot: n/256 + 2 doubles of scratch (multi-workgroup form)
void launch_uniform_scan(int n, double* w_tmp, double* out, double* total, double* ctot, hipStream_t st)
{
    if (n <= IS_MAX_CHUNKS * 256) {
        hipLaunchKernelGGL(uniform_scan_kernel, dim3(1), dim3(IS_BLOCK), 0, st, n, w_tmp, out, total);
        return;
    }
    const int nchunks = (n + 255) / 256;
    hipLaunchKernelGGL(fill_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, w_tmp, n, 1.0 / (double)n);
    hipLaunchKernelGGL(chunk_totals_kernel, dim3(ceil_div(nchunks, 4), 1), dim3(256), 0, st, w_tmp, (size_t)0, n, ctot, 0, nullptr);
    hipLaunchKernelGGL(scan_carry_kernel, dim3(1), dim3(256), 0, st, ctot, 0, nchunks, total, 0, 0, nullptr, 1);
    hipLaunchKernelGGL(scan_write_kernel, dim3(ceil_div(nchunks, 4), 1), dim3(256), 0, st, w_tmp, (size_t)0, n, ctot, 0, out, (size_t)0, nullptr);
}

}  // namespace fba
