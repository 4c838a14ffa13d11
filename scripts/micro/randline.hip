// randline.hip -- what the MI355X memory system gives for the search kernel's access shape: every lane of 4 096 resident
// waves (16 per CU, one lane = one tree) touches its OWN random cache line, far past the L2 and the Infinity Cache.
// The search kernel is bound by exactly this: one more random line per simulation costs it 45 ms.
//   load16    one 16-byte load of a random 64-byte record                         (a node header)
//   load64    four 16-byte loads of one random 64-byte record                     (a packed particle, a whole node)
//   store8    one 8-byte store into a random record                               (a back-up's Q value, line not resident)
//   store24   a 16-byte and an 8-byte store into one random record                (a back-up's header + Q)
//   rmw       16-byte + 8-byte load, then 16-byte + 8-byte store, same record     (a back-up as the search does it)
//   near64    load64 where each lane's records come from its own 4 KB window      (a lane's small tree: lines recur)
//   pair64    lanes 2k and 2k+1 load 16 bytes of the two 64-byte halves of ONE random 128-byte line: if the memory side moves
//             128-byte lines, records/s = 2 x load16; if it moves 64-byte sectors, records/s = load16
//   line128   one lane loads 16 bytes of BOTH halves of its own random 128-byte line (two records per access pair)
//   far2x64   one lane loads 16 bytes of two UNRELATED random 64-byte records (the control for line128: same requests, no shared line)
// Per-lane "dep" accesses are issued with `fly` of them in flight (independent addresses).
// Usage (GPU box):  hipcc --offload-arch=gfx950 -O3 randline.hip -o randline && ./randline [GiB] [iters]
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE ... -- ./randline ; --pmc WRITE_SIZE ; --pmc TCC_HIT_sum TCC_MISS_sum
// Prints one JSON line per shape: lines/s, and the bytes the shape asks for, to set the counters against.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// rec_mask: record index = hash & rec_mask (records of 64 bytes).  near: records inside a 4 KB window of the lane
template <int MODE, int FLY>
__global__ void __launch_bounds__(64) shape_kernel(uint4* buf, uint32_t rec_mask, int iters, uint32_t seed, uint32_t* sink)
{
    const uint32_t gid = blockIdx.x * 64 + threadIdx.x;
    uint32_t acc = 0;
    const uint32_t window = (mix(gid * 2654435761u + seed) & rec_mask) & ~63u;   // near64: 64 records = 4 KB
    for (int it = 0; it < iters; it += FLY) {
        uint4 v[FLY][4];
        uint4* p[FLY];
#pragma unroll
        for (int f = 0; f < FLY; ++f) {
            const uint32_t h = mix((gid * 0x9e3779b9u) ^ mix((uint32_t)(it + f) * 0x85ebca6bu + seed));
            uint32_t rec = MODE == 5 ? (window | (h & 63u)) : (h & rec_mask);
            if (MODE == 6) {   // the pair of lanes shares the hash of the even lane; each takes its own half of the 128-byte line
                const uint32_t hp = mix(((gid & ~1u) * 0x9e3779b9u) ^ mix((uint32_t)(it + f) * 0x85ebca6bu + seed));
                rec = ((hp & rec_mask) & ~1u) | (gid & 1u);
            }
            if (MODE == 7) rec &= ~1u;
            p[f] = buf + (size_t)rec * 4;
        }
        if (MODE == 0) {
#pragma unroll
            for (int f = 0; f < FLY; ++f) v[f][0] = p[f][0];
#pragma unroll
            for (int f = 0; f < FLY; ++f) acc ^= v[f][0].x;
        } else if (MODE == 6) {
#pragma unroll
            for (int f = 0; f < FLY; ++f) v[f][0] = p[f][0];
#pragma unroll
            for (int f = 0; f < FLY; ++f) acc ^= v[f][0].x;
        } else if (MODE == 7 || MODE == 8) {
#pragma unroll
            for (int f = 0; f < FLY; ++f) {
                v[f][0] = p[f][0];
                if (MODE == 7) v[f][1] = p[f][4];   // the other 64-byte half of the same 128-byte line
                else {
                    const uint32_t h2 = mix((gid * 0x85ebca6bu) ^ mix((uint32_t)(it + f) * 0x9e3779b9u + seed + 77u));
                    v[f][1] = buf[(size_t)(h2 & rec_mask) * 4];
                }
            }
#pragma unroll
            for (int f = 0; f < FLY; ++f) acc ^= v[f][0].x ^ v[f][1].y;
        } else if (MODE == 1 || MODE == 5) {
#pragma unroll
            for (int f = 0; f < FLY; ++f)
#pragma unroll
                for (int q = 0; q < 4; ++q) v[f][q] = p[f][q];
#pragma unroll
            for (int f = 0; f < FLY; ++f) acc ^= v[f][0].x ^ v[f][1].y ^ v[f][2].z ^ v[f][3].w;
        } else if (MODE == 2) {
#pragma unroll
            for (int f = 0; f < FLY; ++f) *reinterpret_cast<uint2*>(p[f] + 1) = make_uint2(gid, it);
        } else if (MODE == 3) {
#pragma unroll
            for (int f = 0; f < FLY; ++f) {
                p[f][0] = make_uint4(gid, it, f, 1);
                *reinterpret_cast<uint2*>(p[f] + 1) = make_uint2(gid, it);
            }
        } else {   // rmw
            uint2 w[FLY];
#pragma unroll
            for (int f = 0; f < FLY; ++f) {
                v[f][0] = p[f][0];
                w[f]    = *reinterpret_cast<uint2*>(p[f] + 1);
            }
#pragma unroll
            for (int f = 0; f < FLY; ++f) {
                v[f][0].x += 1; v[f][0].y += 1;
                w[f].x ^= v[f][0].z;
                p[f][0] = v[f][0];
                *reinterpret_cast<uint2*>(p[f] + 1) = w[f];
            }
        }
    }
    if (acc == 0x12345u) sink[gid & 1023] = acc;
}

template <int MODE, int FLY>
static void run(const char* name, uint4* buf, uint32_t rec_mask, int iters, uint32_t* sink, int waves, double bytes_asked_per_access)
{
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    hipLaunchKernelGGL((shape_kernel<MODE, FLY>), dim3(waves), dim3(64), 0, 0, buf, rec_mask, iters / 8, 1u, sink);   // warm-up
    CHK(hipEventRecord(a));
    hipLaunchKernelGGL((shape_kernel<MODE, FLY>), dim3(waves), dim3(64), 0, 0, buf, rec_mask, iters, 7u, sink);
    CHK(hipEventRecord(b));
    CHK(hipEventSynchronize(b));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, a, b));
    const double accesses = (double)waves * 64.0 * iters;
    printf("{\"shape\": \"%s\", \"in_flight_per_lane\": %d, \"waves\": %d, \"accesses\": %.0f, \"ms\": %.3f, \"G_records_per_s\": %.2f, "
           "\"asked_GBs\": %.1f, \"GBs_if_64B_sectors\": %.1f, \"GBs_if_128B_lines\": %.1f}\n",
           name, FLY, waves, accesses, ms, accesses / ms / 1e6, accesses * bytes_asked_per_access / ms / 1e6, accesses * 64.0 / ms / 1e6,
           accesses * 128.0 / ms / 1e6);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 32.0;
    const int iters  = argc > 2 ? atoi(argv[2]) : 2048;
    const int waves  = argc > 3 ? atoi(argv[3]) : 4096;
    uint64_t recs = (uint64_t)(gib * 1024.0 * 1024.0 * 1024.0 / 64.0);
    uint32_t pow2 = 1;
    while ((uint64_t)pow2 * 2 <= recs) pow2 *= 2;
    uint4* buf;
    uint32_t* sink;
    CHK(hipMalloc(&buf, (size_t)pow2 * 64));
    CHK(hipMalloc(&sink, 4096));
    CHK(hipMemset(buf, 1, (size_t)pow2 * 64));
    printf("{\"buffer_GiB\": %.1f, \"records\": %u, \"iters_per_lane\": %d}\n", pow2 * 64.0 / 1073741824.0, pow2, iters);
    const uint32_t mask = pow2 - 1;
    run<0, 1>("load16", buf, mask, iters, sink, waves, 16);
    run<0, 4>("load16", buf, mask, iters, sink, waves, 16);
    run<1, 1>("load64", buf, mask, iters, sink, waves, 64);
    run<1, 2>("load64", buf, mask, iters, sink, waves, 64);
    run<2, 1>("store8", buf, mask, iters, sink, waves, 8);
    run<2, 4>("store8", buf, mask, iters, sink, waves, 8);
    run<3, 1>("store24", buf, mask, iters, sink, waves, 24);
    run<3, 4>("store24", buf, mask, iters, sink, waves, 24);
    run<4, 1>("rmw", buf, mask, iters, sink, waves, 48);
    run<4, 2>("rmw", buf, mask, iters, sink, waves, 48);
    run<5, 1>("near64", buf, mask, iters, sink, waves, 64);
    run<5, 2>("near64", buf, mask, iters, sink, waves, 64);
    run<6, 1>("pair64", buf, mask, iters, sink, waves, 16);
    run<6, 4>("pair64", buf, mask, iters, sink, waves, 16);
    run<7, 1>("line128", buf, mask, iters, sink, waves, 32);
    run<7, 2>("line128", buf, mask, iters, sink, waves, 32);
    run<8, 1>("far2x64", buf, mask, iters, sink, waves, 32);
    run<8, 2>("far2x64", buf, mask, iters, sink, waves, 32);
    return 0;
}
