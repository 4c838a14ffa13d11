// fba_kernels.h -- launch interface between the host engine and the HIP kernels.
#pragma once

#include <hip/hip_runtime.h>

#include "fba_state.h"

namespace fba {

constexpr int SEARCH_BLOCK  = 64;    // one wave per workgroup, one tree per lane
constexpr int SEARCH_STAGE_WORDS = 128; // particle records up to this many words are staged in LDS by the search
constexpr int ROOT_CHILDREN = 8;        // root child pointers are kept in LDS when A*O is at most this
constexpr int REJECT_BLOCK  = 256;   // attempts per chunk of the rejection filter
constexpr int TIGER_LDS_MAX_N = 4096;  // reject_tiger_lds_kernel: filters up to this many packed tiger particles are parked in LDS
constexpr int REJECT_MAX_ATTEMPTS = 1 << 28;  // per update: beyond this the observation is taken to be impossible under the filter
constexpr int IS_BLOCK      = 512;   // one workgroup per slot in the importance filter (same-box A/B on the bench shape: 1024 -> 37.3 ms, 512 -> 31.1, 256 -> 32.3: three workgroups per CU overlap their barrier-separated phases)
constexpr int PARTICLE_TILE = 4096;  // particles one workgroup initialises / resets
constexpr int CARRY_TILE    = 2048;  // chunk totals staged in LDS per step of the carry chain
constexpr int IS_MAX_CHUNKS = 256;   // 256-element scan chunks per slot => N <= 65536
constexpr int MH_MAXVAR = 128;  // structure words of one particle the MH beliefs' chains handle
constexpr int MH_TERMS  = 1024; // doubles: LogBDScore terms one chain evaluates side by side
constexpr int MH_LDS_HEAD = MH_TERMS * 8 + 2 * MH_MAXVAR * 4 + 64 * 4;  // those terms, two structures, terms per lane
inline bool mh_scratch_in_lds(int scratch_words) { return MH_LDS_HEAD + (size_t)scratch_words * 4 <= 64 * 1024; }  // else [E][words] in HBM
constexpr int IS_LDS_MAX_N  = 8192;  // importance filters up to this many particles keep their weights and prefix sums in LDS (64 KB)

void launch_search(const Problem& P, const DeviceState& D, hipStream_t st);
void launch_start(const Problem& P, const DeviceState& D, hipStream_t st);
void launch_env(const Problem& P, const DeviceState& D, int32_t* n_active, hipStream_t st);
void launch_advance(const Problem& P, const DeviceState& D, int32_t* n_active, hipStream_t st);
void launch_belief_update(const Problem& P, const DeviceState& D, hipStream_t st);
void launch_init(const Problem& P, const DeviceState& D, hipStream_t st);
void launch_reset(const Problem& P, const DeviceState& D, hipStream_t st);
void launch_materialize_reset(const Problem& P, const DeviceState& D, hipStream_t st);
void launch_flush(const Problem& P, const DeviceState& D, hipStream_t st);
void launch_selftest_lgamma(const double* x, int count, double* out, hipStream_t st);
void launch_selftest_bd(const Problem& P, const float* cnt, const float* prior, double* out, hipStream_t st);
void launch_selftest_ucb(const double* L, const int32_t* n, int count, double u, double* out, hipStream_t st);
hipError_t launch_take_error();   // a HIP error a launch_* function met on its way (and clears it); hipSuccess if none
void launch_uniform_scan(int n, double* w_tmp, double* out, double* total, double* ctot, hipStream_t st);

}  // namespace fba
