// lev_bitsq.hip -- gfx950 instantiations of the small-alphabet bit-parallel band kernel (lev_bitsq_body.h).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "lev_bitsq_body.h"
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

}  // namespace ta
