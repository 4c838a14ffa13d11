// pmc_calibrate.hip -- what FETCH_SIZE / WRITE_SIZE report on gfx950 for the access shapes of the belief kernels,
// on known byte counts (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated: calibrate on a known byte
// count in your own access pattern").  Each kernel runs the shape over `slots` filters of `n` 64-byte records, far
// past the 256 MiB Infinity Cache, one workgroup of 512 threads per filter like reject_tiger_lds_kernel:
//   park_kernel    the pass that parks a filter in LDS: five 4-byte words of every 64-byte record, records in order
//                  (thread i reads record i, i + 512, ...): every 64-byte line of the filter is touched once
//   gather_kernel  the gather: record j of the output = a random record of the SAME filter (256 KB), read as
//                  4 lanes x 16 bytes, written as 4 lanes x 16 bytes
//   stream_kernel  a plain float4 copy (the guide's calibrated case), for reference
// Usage (on the GPU box):  hipcc --offload-arch=gfx950 -O3 pmc_calibrate.hip -o pmc_calibrate
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out/fetch -o run -- ./pmc_calibrate
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out/write -o run -- ./pmc_calibrate
// The program prints the byte counts each kernel is known to move; scripts/pmc_calibration.py divides.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(512) park_kernel(const uint32_t* recs, int n, uint32_t* sink, int a)
{
    extern __shared__ uint32_t s_tab[];
    const uint32_t* f = recs + (size_t)blockIdx.x * n * 16;
    for (int i = threadIdx.x; i < n; i += 512) {
        const uint32_t* rec = f + (size_t)i * 16;
        s_tab[4 * i + 0] = rec[a];
        s_tab[4 * i + 1] = rec[3 + a];
        s_tab[4 * i + 2] = rec[6 + 2 * a] ^ rec[7 + 2 * a];
        s_tab[4 * i + 3] = rec[12];
    }
    __syncthreads();
    uint32_t x = 0;
    for (int i = threadIdx.x; i < 4 * n; i += 512) x ^= s_tab[i];
    if (x == 0x12345u) sink[blockIdx.x] = x;   // keeps the loads alive, practically never stores
}

__global__ void __launch_bounds__(512) gather_kernel(const float4* src, float4* dst, int n, uint32_t seed)
{
    const float4* f = src + (size_t)blockIdx.x * n * 4;
    float4* d       = dst + (size_t)blockIdx.x * n * 4;
    const int part  = threadIdx.x & 3;
    for (int j = threadIdx.x >> 2; j < n; j += 128) {
        uint32_t h = (uint32_t)j * 2654435761u ^ seed ^ (blockIdx.x * 40503u);
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const int s = (int)(h % (uint32_t)n);
        d[(size_t)j * 4 + part] = f[(size_t)s * 4 + part];
    }
}

__global__ void __launch_bounds__(256) stream_kernel(const float4* src, float4* dst, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

int main(int argc, char** argv)
{
    const int slots = argc > 1 ? atoi(argv[1]) : 65536, n = 4096;
    const size_t bytes = (size_t)slots * n * 64;
    uint32_t *a, *b, *sink;
    CHK(hipMalloc(&a, bytes));
    CHK(hipMalloc(&b, bytes));
    CHK(hipMalloc(&sink, slots * 4));
    CHK(hipMemset(a, 1, bytes));
    CHK(hipMemset(b, 2, bytes));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(park_kernel, dim3(slots), dim3(512), (size_t)n * 16, 0, a, n, sink, 2);
        hipLaunchKernelGGL(gather_kernel, dim3(slots), dim3(512), 0, 0, (const float4*)a, (float4*)b, n, 77u + rep);
        hipLaunchKernelGGL(stream_kernel, dim3(8192), dim3(256), 0, 0, (const float4*)a, (float4*)b, bytes / 16);
        CHK(hipDeviceSynchronize());
    }
    // distinct source records a gather launch reads: n draws with replacement from n -> n (1 - 1/e) on average
    printf("{\"slots\": %d, \"records_per_filter\": %d, \"filter_bytes\": %zu, \"total_bytes\": %zu,\n", slots, n, (size_t)n * 64, bytes);
    printf(" \"park_kernel\": {\"fetch_bytes_known\": %zu, \"write_bytes_known\": 0},\n", bytes);
    printf(" \"gather_kernel\": {\"fetch_bytes_if_every_read_misses\": %zu, \"fetch_bytes_distinct_records\": %.0f, \"write_bytes_known\": %zu},\n", bytes,
           (double)bytes * 0.6321205588, bytes);
    printf(" \"stream_kernel\": {\"fetch_bytes_known\": %zu, \"write_bytes_known\": %zu}}\n", bytes, bytes);
    return 0;
}
