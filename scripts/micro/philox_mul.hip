// micro-benchmark: Philox4x32-10 with v_mul_hi_u32 + v_mul_lo_u32 against one 32x32->64 multiply per pair
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int MODE>
__device__ __forceinline__ void philox(uint32_t& x0, uint32_t& x1, uint32_t& x2, uint32_t& x3, uint32_t a, uint32_t b)
{
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        uint32_t hi0, lo0, hi1, lo1;
        if (MODE == 0) {
            hi0 = __umulhi(0xD2511F53u, x0); lo0 = 0xD2511F53u * x0;
            hi1 = __umulhi(0xCD9E8D57u, x2); lo1 = 0xCD9E8D57u * x2;
        } else {
            uint64_t p0, p1;
            if (MODE == 1) {
                p0 = (uint64_t)0xD2511F53u * (uint64_t)x0;
                p1 = (uint64_t)0xCD9E8D57u * (uint64_t)x2;
            } else {
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p0) : "v"(x0), "v"(0xD2511F53u) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p1) : "v"(x2), "v"(0xCD9E8D57u) : "vcc");
            }
            hi0 = (uint32_t)(p0 >> 32); lo0 = (uint32_t)p0;
            hi1 = (uint32_t)(p1 >> 32); lo1 = (uint32_t)p1;
        }
        uint32_t n0 = hi1 ^ x1 ^ a, n2 = hi0 ^ x3 ^ b;
        x0 = n0; x1 = lo1; x2 = n2; x3 = lo0;
        a += 0x9E3779B9u;
        b += 0xBB67AE85u;
    }
}

template <int MODE>
__global__ void k(uint32_t* out, int iters)
{
    uint32_t x0 = threadIdx.x, x1 = blockIdx.x, x2 = 7, x3 = 9;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint32_t y0 = x0 + i, y1 = x1, y2 = x2, y3 = x3;
        philox<MODE>(y0, y1, y2, y3, 123u, 456u);
        acc ^= y0 ^ y1 ^ y2 ^ y3;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE>
void run(const char* name, uint32_t* d, int blocks)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 100);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    uint32_t h[4];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    // one wave per block; waves per SIMD = blocks / 1024
    printf("%-28s blocks %5d: %8.3f ms  %7.1f cycles/block-of-4x32 per wave (2.4 GHz)  check %08x\n", name, blocks, ms,
           ms * 1e-3 * 2.4e9 / iters / ((blocks + 1023) / 1024), h[0] ^ h[1]);
}

int main()
{
    uint32_t* d;
    hipMalloc(&d, 8192 * 64 * 4);
    for (int blocks : {1024, 2048, 4096}) {
        run<0>("mul_hi + mul_lo", d, blocks);
        run<1>("u64 product (compiler)", d, blocks);
        run<2>("v_mad_u64_u32 (asm)", d, blocks);
    }
    return 0;
}
