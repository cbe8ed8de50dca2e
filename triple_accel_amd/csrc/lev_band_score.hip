// lev_band_score.hip -- gfx950 instantiations of the band-wavefront kernel's SCORE form (lev_band_body.h): the cells hold
// gc (i+j) - dp(i,j), gap steps are free, a substitution adds one byte through v_dot4, minima become signed maxima.
// Taken by lev_band_launch (lev_band.hip) whenever lev_score_form_applies (lev_plan.h); same answers as the cost form.
#include <hip/hip_runtime.h>

#include "lev_band_body.h"
#include "lev_plan.h"
#include "ta_internal.h"

namespace ta {

constexpr int LEV_SCORE_WAVES_PER_BLOCK = 4;      // as lev_band.hip (the launcher there sizes the grid and the LDS)

template <int D, bool AFFINE, int TRANS, bool L1>
__global__ __launch_bounds__(64 * LEV_SCORE_WAVES_PER_BLOCK) void lev_band_score_kernel(LevParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t wave = threadIdx.x >> 6, w = blockIdx.x * LEV_SCORE_WAVES_PER_BLOCK + wave;
    if (P.n_dev) {                                     // a list whose length only the device knows (lev_band.hip)
        LevParams Q = P;
        Q.n = *P.n_dev;
        if ((uint64_t)w * P.PW < Q.n) LevBand<DevWave, D, AFFINE, TRANS, false, L1, true>::run(Q, w, lds + wave * P.lds_per_wave);
        return;
    }
    LevBand<DevWave, D, AFFINE, TRANS, false, L1, true>::run(P, w, lds + wave * P.lds_per_wave);
}

// one lane per pair + a fixed-length batch: the LINE form of the fetch (every line of a string requested once, parked in registers)
template <int D, bool AFFINE, int TRANS>
__global__ __launch_bounds__(64 * LEV_SCORE_WAVES_PER_BLOCK) void lev_band_score_line_kernel(LevParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t wave = threadIdx.x >> 6, w = blockIdx.x * LEV_SCORE_WAVES_PER_BLOCK + wave;
    if (P.n_dev) {                                     // a list whose length only the device knows (lev_band.hip)
        LevParams Q = P;
        Q.n = *P.n_dev;
        if ((uint64_t)w * P.PW < Q.n) LevBand<DevWave, D, AFFINE, TRANS, false, true, true, true>::run(Q, w, lds + wave * P.lds_per_wave);
        return;
    }
    LevBand<DevWave, D, AFFINE, TRANS, false, true, true, true>::run(P, w, lds + wave * P.lds_per_wave);
}

template <int D>
static hipError_t launch_score_d(const LevParams &P, bool affine, int trans, uint32_t grid, size_t lds, hipStream_t s) {
    dim3 g(grid), b(64 * LEV_SCORE_WAVES_PER_BLOCK);
    const bool l1 = P.L == 1;
    // (the parked lines are 64 VGPRs: the line form is taken where the kernel still fits 128 -- four wavefronts per SIMD: up to 12 diagonals
    // per lane, without the affine + transposition combination at 12 (147 VGPRs))
    if constexpr (D <= 12) {
        if (l1 && !P.a.off && !P.b.off && P.ch == 32u && (D <= 10 || !(affine && trans == 1)) && !env_int("TA_BAND_NO_LINE")) {
            if (affine) { if (trans == 1) hipLaunchKernelGGL((lev_band_score_line_kernel<D, true, 1>), g, b, lds, s, P);
                          else hipLaunchKernelGGL((lev_band_score_line_kernel<D, true, 0>), g, b, lds, s, P); }
            else { if (trans == 1) hipLaunchKernelGGL((lev_band_score_line_kernel<D, false, 1>), g, b, lds, s, P);
                   else hipLaunchKernelGGL((lev_band_score_line_kernel<D, false, 0>), g, b, lds, s, P); }
            set_last_kernel_name("lev_band_score_line_kernel<%d, %s, %d>", D, affine ? "true" : "false", trans);
            return hipGetLastError();
        }
    }
#define TA_L(A, T) do { if (l1) hipLaunchKernelGGL((lev_band_score_kernel<D, A, T, true>), g, b, lds, s, P); \
                        else hipLaunchKernelGGL((lev_band_score_kernel<D, A, T, false>), g, b, lds, s, P); } while (0)
    if (affine) { if (trans == 1) TA_L(true, 1); else TA_L(true, 0); }
    else { if (trans == 1) TA_L(false, 1); else TA_L(false, 0); }
#undef TA_L
    return hipGetLastError();
}

hipError_t lev_band_score_launch(const LevParams &P, const LevPlan &pl, bool affine, int trans, uint32_t grid, size_t lds, hipStream_t s) {
    if (trans == 2) return hipErrorInvalidValue;
    set_last_kernel_name("lev_band_score_kernel<%d, %s, %d, %s>", pl.D, affine ? "true" : "false", trans, P.L == 1 ? "true" : "false");
    switch (pl.D) {
#define TA_CASE(d) case d: return launch_score_d<d>(P, affine, trans, grid, lds, s);
        TA_CASE(2) TA_CASE(4) TA_CASE(6) TA_CASE(8) TA_CASE(10) TA_CASE(12) TA_CASE(16) TA_CASE(18) TA_CASE(20)
        TA_CASE(22) TA_CASE(24) TA_CASE(28) TA_CASE(32) TA_CASE(34) TA_CASE(40) TA_CASE(48) TA_CASE(56) TA_CASE(66)
#undef TA_CASE
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ta
