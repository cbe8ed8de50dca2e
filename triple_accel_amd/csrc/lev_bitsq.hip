// lev_bitsq.hip -- gfx950 instantiations of the small-alphabet bit-parallel band kernel (lev_bitsq_body.h).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <atomic>

#include "lev_bitsq_body.h"
#include "lev_bitsqw_body.h"
#include "lev_plan.h"
#include "ta_internal.h"

namespace ta {

constexpr int BITSQ_WAVES_PER_BLOCK = 4;

template <bool TRANS>
__global__ __launch_bounds__(64 * BITSQ_WAVES_PER_BLOCK) void lev_bitsq_kernel(LevParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t wave = threadIdx.x >> 6, wpb = blockDim.x >> 6, waves = (P.n + 63u) >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0 && P.q_next_count) *P.q_next_count = 0;   // nobody reads it before the next pass
    for (uint32_t w = blockIdx.x * wpb + wave; w < waves; w += gridDim.x * wpb)
        LevBitsQ<DevWave, TRANS>::run(P, w, lds + wave * P.lds_per_wave);
}

// P: a fixed-length unit-cost batch with P.u + 1 (+ 2) <= 33, P.q_table / P.q_shift / P.q_bad_list / P.q_bad_count set
hipError_t lev_bitsq_launch(const LevParams &P0, bool trans, hipStream_t s, uint32_t *grid_out, uint32_t *lds_out) {
    LevParams P = P0;
    P.lds_per_wave = LevBitsQ<DevWave, false>::LDS_PER_WAVE;
    uint32_t wpb = BITSQ_WAVES_PER_BLOCK;
    if (const char *e = env_str("TA_BITS_WPB")) { const int v = atoi(e); if (v >= 1 && v <= BITSQ_WAVES_PER_BLOCK) wpb = (uint32_t)v; }
    const uint32_t waves = (P.n + 63u) / 64u, grid = (waves + wpb - 1) / wpb;
    const size_t lds = (size_t)P.lds_per_wave * wpb;
    if (grid_out) *grid_out = grid;
    if (lds_out) *lds_out = (uint32_t)lds;
    if (grid == 0) return hipSuccess;
    set_last_kernel_name("lev_bitsq_kernel<%s>", trans ? "true" : "false");
    if (trans) hipLaunchKernelGGL(lev_bitsq_kernel<true>, dim3(grid), dim3(64 * wpb), lds, s, P);
    else hipLaunchKernelGGL(lev_bitsq_kernel<false>, dim3(grid), dim3(64 * wpb), lds, s, P);
    return hipGetLastError();
}

// ---- alphabets of up to 32 symbols (lev_bitsqw_body.h): LDS = the byte -> ring table of the workgroup + the wavefronts' rings
constexpr int BITSQW_MAX_WAVES_PER_BLOCK = 8;

template <bool TRANS>
__global__ __launch_bounds__(64 * BITSQW_MAX_WAVES_PER_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void lev_bitsqw_kernel(LevParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    using K = LevBitsQW<DevWave, TRANS>;
    for (uint32_t i = threadIdx.x; i < K::TABLE_BYTES; i += blockDim.x) lds[i] = (uint8_t)lev_bitsqw_entry(i, P.q_shift, P.q_memb, P.q_hi);
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, wpb = blockDim.x >> 6, waves = (P.n + 63u) >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0 && P.q_next_count) *P.q_next_count = 0;   // nobody reads it before the next pass
    for (uint32_t w = blockIdx.x * wpb + wave; w < waves; w += gridDim.x * wpb)
        K::run(P, w, lds + K::TABLE_BYTES + wave * P.lds_per_wave, lds);
}

// P as for lev_bitsq_launch with P.q_shift / q_memb / q_hi / q_ns from lev_bitsqw_hash
constexpr uint32_t BITSQW_TAIL_PAD = 1040;       // 4 * 0xFF + 8 bytes behind the last pair's rings, rounded up to 16
hipError_t lev_bitsqw_launch(const LevParams &P0, bool trans, hipStream_t s, uint32_t *grid_out, uint32_t *lds_out) {
    using K = LevBitsQW<DevWave, false>;
    LevParams P = P0;
    P.lds_per_wave = K::lds_per_wave(P.q_ns);
    // wavefronts per workgroup x workgroups per CU: the most wavefronts (up to 16) the CU's 160 KB of LDS hold (the larger workgroup on a tie)
    uint32_t wpb = 1, best = 0;
    for (uint32_t blocks = 1; blocks <= 4; blocks++) {
        const uint32_t room = 156u * 1024u / blocks;    // (4 KB of the 160 left to the allocation granule)
        if (room < K::TABLE_BYTES + P.lds_per_wave + BITSQW_TAIL_PAD) break;
        uint32_t w = (room - K::TABLE_BYTES - BITSQW_TAIL_PAD) / P.lds_per_wave;
        if (w > (uint32_t)BITSQW_MAX_WAVES_PER_BLOCK) w = BITSQW_MAX_WAVES_PER_BLOCK;
        const uint32_t total = w * blocks > 16u ? 16u : w * blocks;
        if (total > best) { best = total; wpb = w; }
    }
    if (const char *e = env_str("TA_BITS_WPB")) { const int v = atoi(e); if (v >= 1 && v <= BITSQW_MAX_WAVES_PER_BLOCK) wpb = (uint32_t)v; }
    const uint32_t waves = (P.n + 63u) / 64u, grid = (waves + wpb - 1) / wpb;
    // (+ the tail pad: a byte of `b` outside the alphabet looks its ring up at table entry 0xFF -- ring + 1020, an 8-byte read that no answer
    // uses (the pair is re-answered by the byte-test kernel) but that must stay inside the workgroup's allocation for every pair of the block)
    const size_t lds = K::TABLE_BYTES + (size_t)P.lds_per_wave * wpb + BITSQW_TAIL_PAD;
    if (grid_out) *grid_out = grid;
    if (lds_out) *lds_out = (uint32_t)lds;
    if (grid == 0) return hipSuccess;
    // workgroups of more than 64 KB of dynamic LDS need the attribute, once per device (idempotent: a race sets it twice)
    static std::atomic<uint64_t> attr_set{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_set.load(std::memory_order_relaxed) & bit)) {
        e = hipFuncSetAttribute((const void *)lev_bitsqw_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)lev_bitsqw_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set.fetch_or(bit, std::memory_order_relaxed);
    }
    set_last_kernel_name("lev_bitsqw_kernel<%s>", trans ? "true" : "false");
    if (trans) hipLaunchKernelGGL(lev_bitsqw_kernel<true>, dim3(grid), dim3(64 * wpb), lds, s, P);
    else hipLaunchKernelGGL(lev_bitsqw_kernel<false>, dim3(grid), dim3(64 * wpb), lds, s, P);
    return hipGetLastError();
}

}  // namespace ta
