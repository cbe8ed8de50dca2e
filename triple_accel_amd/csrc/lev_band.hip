// lev_band.hip -- gfx950 instantiations of the band-wavefront kernel (lev_band_body.h).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "lev_band_body.h"
#include "lev_plan.h"
#include "ta_internal.h"

namespace ta {

constexpr int LEV_WAVES_PER_BLOCK = 4;

// L1: one lane per pair (the whole band in one lane's D diagonals: bands of up to 66 diagonals with 64 pairs per wavefront)
template <int D, bool AFFINE, int TRANS, bool L1>
__global__ __launch_bounds__(64 * LEV_WAVES_PER_BLOCK) void lev_band_kernel(LevParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t wave = threadIdx.x >> 6;
    LevBand<DevWave, D, AFFINE, TRANS, false, L1>::run(P, blockIdx.x * LEV_WAVES_PER_BLOCK + wave, lds + wave * P.lds_per_wave);
}

template <int D>
static hipError_t launch_d(const LevParams &P, bool affine, int trans, uint32_t grid, size_t lds, hipStream_t s) {
    dim3 g(grid), b(64 * LEV_WAVES_PER_BLOCK);
    const bool l1 = P.L == 1;
#define TA_L(A, T) do { if (l1) hipLaunchKernelGGL((lev_band_kernel<D, A, T, true>), g, b, lds, s, P); \
                        else hipLaunchKernelGGL((lev_band_kernel<D, A, T, false>), g, b, lds, s, P); } while (0)
    if (affine) { if (trans == 1) TA_L(true, 1); else if (trans == 2) TA_L(true, 2); else TA_L(true, 0); }
    else { if (trans == 1) TA_L(false, 1); else if (trans == 2) TA_L(false, 2); else TA_L(false, 0); }
#undef TA_L
    return hipGetLastError();
}

// lev_band_score.hip: the score-form instantiations (their own translation unit: as many kernels again)
hipError_t lev_band_score_launch(const LevParams &P, const LevPlan &pl, bool affine, int trans, uint32_t grid, size_t lds, hipStream_t s);

// Launches the kernel for plan `pl`; returns the grid size through *grid_out.
hipError_t lev_band_launch(const LevParams &P, const LevPlan &pl, bool affine, int trans, hipStream_t s,
                           uint32_t *grid_out, uint32_t *lds_out) {
    const uint32_t waves = (P.n + pl.PW - 1) / pl.PW;
    const uint32_t grid = (waves + LEV_WAVES_PER_BLOCK - 1) / LEV_WAVES_PER_BLOCK;
    const size_t lds = (size_t)pl.lds_per_wave * LEV_WAVES_PER_BLOCK;
    if (grid_out) *grid_out = grid;
    if (lds_out) *lds_out = (uint32_t)lds;
    if (grid == 0) return hipSuccess;
    // cells as scores gc (i+j) - dp wherever the costs allow it (lev_plan.h): 5 instructions per affine cell instead of 7
    static const bool no_score = env_str("TA_NO_SCORE_FORM") != nullptr;       // tuning switch (TA_TUNING=1): the cost form always
    if (!no_score && lev_score_form_applies(P.mc, P.gc, trans, P.tc)) return lev_band_score_launch(P, pl, affine, trans, grid, lds, s);
    set_last_kernel_name("lev_band_kernel<%d, %s, %d, %s>", pl.D, affine ? "true" : "false", trans, P.L == 1 ? "true" : "false");
    switch (pl.D) {
#define TA_CASE(d) case d: return launch_d<d>(P, affine, trans, grid, lds, s);
        TA_CASE(2) TA_CASE(4) TA_CASE(6) TA_CASE(8) TA_CASE(10) TA_CASE(12) TA_CASE(16) TA_CASE(18) TA_CASE(20)
        TA_CASE(22) TA_CASE(24) TA_CASE(28) TA_CASE(32) TA_CASE(34) TA_CASE(40) TA_CASE(48) TA_CASE(56) TA_CASE(66)
#undef TA_CASE
        default: return hipErrorInvalidValue;
    }
}

// Trace kernels (single-pair traceback, trace_on = true): D = 16 (bands up to 1024 diagonals) or D = 66.
template <int D, bool AFFINE, int TRANS>
__global__ __launch_bounds__(64) void lev_band_trace_kernel(LevParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    LevBand<DevWave, D, AFFINE, TRANS, true>::run(P, blockIdx.x, lds);
}

hipError_t lev_band_trace_launch(const LevParams &P, const LevPlan &pl, bool affine, bool trans, hipStream_t s) {
    const uint32_t waves = (P.n + pl.PW - 1) / pl.PW;
    dim3 g(waves), b(64);
    const size_t lds = pl.lds_per_wave;
#define TA_T(D_) \
    if (affine) { if (trans) hipLaunchKernelGGL((lev_band_trace_kernel<D_, true, 2>), g, b, lds, s, P); \
                  else hipLaunchKernelGGL((lev_band_trace_kernel<D_, true, 0>), g, b, lds, s, P); } \
    else { if (trans) hipLaunchKernelGGL((lev_band_trace_kernel<D_, false, 2>), g, b, lds, s, P); \
           else hipLaunchKernelGGL((lev_band_trace_kernel<D_, false, 0>), g, b, lds, s, P); }
    if (pl.D == 16) { TA_T(16) }
    else if (pl.D == 66) { TA_T(66) }
    else return hipErrorInvalidValue;
#undef TA_T
    return hipGetLastError();
}

}  // namespace ta
