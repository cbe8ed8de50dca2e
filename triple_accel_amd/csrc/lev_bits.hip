// lev_bits.hip -- gfx950 instantiations of the bit-parallel band kernel (lev_bits_body.h).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "lev_bits_body.h"
#include "lev_bits2_body.h"
#include "lev_one_body.h"
#include "lev_plan.h"
#include "ta_internal.h"

namespace ta {

constexpr int BITS_WAVES_PER_BLOCK = 4;

template <int NA, bool TRANS, bool STATIC>
__global__ __launch_bounds__(64 * BITS_WAVES_PER_BLOCK) void lev_bits_kernel(LevParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    LevParams Q = P;
    if (P.n_dev) Q.n = *P.n_dev;                       // the pairs of a list a kernel before this one wrote (lev_bitsq's fallback)
    const uint32_t wave = threadIdx.x >> 6, wpb = blockDim.x >> 6, waves = (Q.n + 63u) >> 6;
    // one trip when the grid covers the batch; a smaller (persistent) grid strides over it (TA_BITS_PERSIST, lists of unknown length)
    for (uint32_t w = blockIdx.x * wpb + wave; w < waves; w += gridDim.x * wpb)
        LevBits<DevWave, NA, TRANS, STATIC>::run(Q, w, lds + wave * P.lds_per_wave);
}

// line-form launches (fixed-length batches of strings longer than one line)
template <int NA, bool TRANS, bool STATIC>
__global__ __launch_bounds__(64 * BITS_WAVES_PER_BLOCK) void lev_bits_line_kernel(LevParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    LevParams Q = P;
    if (P.n_dev) Q.n = *P.n_dev;                       // the pairs of a list a kernel before this one wrote (lev_bitsq's fallback)
    const uint32_t wave = threadIdx.x >> 6, wpb = blockDim.x >> 6, waves = (Q.n + 63u) >> 6;
    // one trip when the grid covers the batch; a smaller (persistent) grid strides over it (TA_BITS_PERSIST, lists of unknown length)
    for (uint32_t w = blockIdx.x * wpb + wave; w < waves; w += gridDim.x * wpb)
        LevBits<DevWave, NA, TRANS, STATIC, true>::run(Q, w, lds + wave * P.lds_per_wave);
}

// stride-8 form (bands of up to 33 diagonals), either fetch form
template <bool TRANS, bool LINE, bool EARLY = false>
__global__ __launch_bounds__(64 * BITS_WAVES_PER_BLOCK) void lev_bits_s8_kernel(LevParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    LevParams Q = P;
    if (P.n_dev) Q.n = *P.n_dev;                       // the pairs of a list a kernel before this one wrote (lev_bitsq's fallback)
    const uint32_t wave = threadIdx.x >> 6, wpb = blockDim.x >> 6, waves = (Q.n + 63u) >> 6;
    // one trip when the grid covers the batch; a smaller (persistent) grid strides over it (TA_BITS_PERSIST, lists of unknown length)
    for (uint32_t w = blockIdx.x * wpb + wave; w < waves; w += gridDim.x * wpb)
        LevBits<DevWave, 8, TRANS, false, LINE, true, EARLY>::run(Q, w, lds + wave * P.lds_per_wave);
}

// stride-8 form with checkpoints (lev_bits_body.h, CKPT): the distance pass of ta_levenshtein_trace_batch over a fixed-length batch
template <bool TRANS, bool LINE>
__global__ __launch_bounds__(64 * BITS_WAVES_PER_BLOCK) void lev_bits_s8_ckpt_kernel(LevParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t wave = threadIdx.x >> 6, wpb = blockDim.x >> 6, waves = (P.n + 63u) >> 6;
    for (uint32_t w = blockIdx.x * wpb + wave; w < waves; w += gridDim.x * wpb)
        LevBits<DevWave, 8, TRANS, false, LINE, true, false, false, true>::run(P, w, lds + wave * P.lds_per_wave);
}

#ifdef TA_BITS2_WAVES_PER_SIMD        // A/B builds only: cap the registers for this many wavefronts per SIMD
#define TA_BITS2_ATTR __attribute__((amdgpu_waves_per_eu(TA_BITS2_WAVES_PER_SIMD, TA_BITS2_WAVES_PER_SIMD)))
#else
#define TA_BITS2_ATTR
#endif
template <bool TRANS, bool EARLY = false>
__global__ __launch_bounds__(64 * BITS_WAVES_PER_BLOCK) TA_BITS2_ATTR void lev_bits2_kernel(LevParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t wave = threadIdx.x >> 6, w = blockIdx.x * (blockDim.x >> 6) + wave;
    if (P.n_dev) {                                     // a list whose length only the device knows (the rounds of ta_levenshtein_exp_batch)
        LevParams Q = P;
        Q.n = *P.n_dev;
        if ((uint64_t)w * 128u < Q.n) LevBits2<DevWave, TRANS, EARLY>::run(Q, w, lds + wave * P.lds_per_wave);
        return;
    }
    LevBits2<DevWave, TRANS, EARLY>::run(P, w, lds + wave * P.lds_per_wave);
}

// two pairs per lane (lev_bits2_body.h): 128 pairs per wavefront
hipError_t lev_bits2_launch(const LevParams &P0, const LevBits2Plan &pl, bool trans, hipStream_t s, uint32_t *grid_out, uint32_t *lds_out) {
    LevParams P = P0;
    if (early_out_enabled()) P.tune |= 2u;
    uint32_t wpb = 2;                               // cfg4: 0.1180 ms against 0.1195 with one or four wavefronts per block
    if (const char *e = env_str("TA_BITS_WPB")) { const int v = atoi(e); if (v >= 1 && v <= BITS_WAVES_PER_BLOCK) wpb = (uint32_t)v; }
    const uint32_t waves = (P.n + 127u) / 128u, grid = (waves + wpb - 1) / wpb;
    const size_t lds = (size_t)pl.lds_per_wave * wpb;
    if (grid_out) *grid_out = grid;
    if (lds_out) *lds_out = (uint32_t)lds;
    if (grid == 0) return hipSuccess;
    dim3 g(grid), b(64 * wpb);
    const bool early = (P.tune & 2u) != 0u;
    set_last_kernel_name("lev_bits2_kernel<%s, %s>", trans ? "true" : "false", early ? "true" : "false");
    if (early) {
        if (trans) hipLaunchKernelGGL((lev_bits2_kernel<true, true>), g, b, lds, s, P);
        else hipLaunchKernelGGL((lev_bits2_kernel<false, true>), g, b, lds, s, P);
    } else if (trans) hipLaunchKernelGGL((lev_bits2_kernel<true, false>), g, b, lds, s, P);
    else hipLaunchKernelGGL((lev_bits2_kernel<false, false>), g, b, lds, s, P);
    return hipGetLastError();
}

template <bool TRANS, bool WIDE>
__global__ __launch_bounds__(64) void lev_one_kernel(LevParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    LevOne<DevWave, TRANS, WIDE>::run(P, lds);
}

// one pair, one wavefront (lev_one_body.h); max_len = the longer string (the launcher's promise: <= LEV_ONE_MAX_LEN, band <= 64)
hipError_t lev_one_launch(const LevParams &P, bool trans, uint64_t max_len, hipStream_t s, uint32_t *lds_out) {
    const size_t lds = (LEV_ONE_PAD_LO + (size_t)max_len + LEV_ONE_PAD_HI + 15u) & ~(size_t)15;
    if (lds_out) *lds_out = (uint32_t)lds;
    const bool wide = P.u + 1u + (trans ? 2u : 0u) > 32u;
    set_last_kernel_name("lev_one_kernel<%s, %s>", trans ? "true" : "false", wide ? "true" : "false");
    if (trans) { if (wide) hipLaunchKernelGGL((lev_one_kernel<true, true>), dim3(1), dim3(64), lds, s, P); else hipLaunchKernelGGL((lev_one_kernel<true, false>), dim3(1), dim3(64), lds, s, P); }
    else { if (wide) hipLaunchKernelGGL((lev_one_kernel<false, true>), dim3(1), dim3(64), lds, s, P); else hipLaunchKernelGGL((lev_one_kernel<false, false>), dim3(1), dim3(64), lds, s, P); }
    return hipGetLastError();
}

template <int NA>
static hipError_t launch_na(const LevParams &P, bool trans, bool stat, bool line, uint32_t grid, uint32_t wpb, size_t lds, hipStream_t s) {
    dim3 g(grid), b(64 * wpb);
#define TA_GO(T, S) do { if (line) hipLaunchKernelGGL((lev_bits_line_kernel<NA, T, S>), g, b, lds, s, P); \
                         else hipLaunchKernelGGL((lev_bits_kernel<NA, T, S>), g, b, lds, s, P); return hipGetLastError(); } while (0)
    if constexpr (NA >= 8) {
        if (stat) { if (trans) TA_GO(true, true); else TA_GO(false, true); }
    }
    if (stat) return hipErrorInvalidValue;
    if (trans) TA_GO(true, false); else TA_GO(false, false);
#undef TA_GO
}

hipError_t lev_bits_launch(const LevParams &P0, const LevBitsPlan &pl, bool trans, uint64_t max_len, hipStream_t s,
                           uint32_t *grid_out, uint32_t *lds_out) {
    // CSR batches whose band fits the stride-8 window, under TA_BITS_VLINE=1: the VLINE form (every line of a string requested once, per-lane
    // geometry and alignment; lev_bits_vline.hip).  It runs one column count per pass, so it takes the batches ta_levenshtein_k_batch has
    // ordered by their exact column count (tune bit 2).  Measured, not the default: 0.72 x the chunk form's fabric-side bytes on the ragged
    // cfg2 batch but 8 % more time -- ten times the load instructions (eight lane classes burst in turn) on an issue-bound kernel
    // (profiles/r04/ab_ragged.md).
    if ((P0.a.off || P0.b.off) && pl.s8 && (P0.tune & 4u) && P0.subset && env_int("TA_BITS_VLINE")) return lev_bits_vline_launch(P0, pl, trans, s, grid_out, lds_out);
    LevParams P = P0;
    // fixed-length batches of strings longer than one 128-byte line take the line form of the fetch (lev_bits_body.h); up to one
    // line the chunk form has nothing to refetch and its coarser events (one per 64 columns, not per 16) are cheaper:
    // cfg4 0.150 ms against 0.180 ms (profiles/r02/ab_band_kernel.md).  TA_BITS_NO_COOP=1 pins the chunk form.
    if (max_len <= 128u || env_int("TA_BITS_NO_COOP")) P.tune |= 1u;
    if (early_out_enabled()) P.tune |= 2u;
    const bool line_form = !P.a.off && !P.b.off && !(P.tune & 1u);
    if (pl.s8 && line_form) P.lds_per_wave = 64u * (52u + 36u);       // the stride-8 line form's small rings (lev_bits_body.h)
    const uint32_t waves = (P.n + 63u) / 64u;
    // 4 waves per block while four rings fit a quarter of the CU's LDS; else one wave per block so that the CU packs
    // as many waves as the LDS holds.  TA_BITS_WPB pins the waves per block, TA_BITS_BLOCK_LDS the block's LDS request
    // (= the resident blocks per CU), for the occupancy sweeps of profiles/.
    uint32_t wpb = P.lds_per_wave * 16u <= 160u * 1024u ? BITS_WAVES_PER_BLOCK : 1u;
    if (pl.s8 && line_form) wpb = 1u;          // finer grains at the launch's tail: 0.3138-0.3147 against 0.3159-0.3171 ms on cfg2
    if (const char *e = env_str("TA_BITS_WPB")) { const int v = atoi(e); if (v >= 1 && v <= BITS_WAVES_PER_BLOCK) wpb = (uint32_t)v; }
    uint32_t grid = (waves + wpb - 1) / wpb;
    if (const char *e = env_str("TA_BITS_PERSIST")) { const int v = atoi(e); if (v >= 1 && (uint32_t)v < grid) grid = (uint32_t)v; }   // A/B: persistent grid
    if (P.n_dev && !(P.tune & 8u) && grid > 512u) grid = 512u;   // a list whose length only the device knows and that is usually empty (lev_bitsq's fallback): a small striding grid
                                                                  // (tune bit 3, the rounds of ta_levenshtein_exp_batch: the grid of the list's upper bound; wavefronts behind its end leave at once)
    // CSR batches (chunk form, half lines fetched 64 iterations apart) with strings longer than one 128-byte line: three blocks
    // (12 waves) per CU instead of four -- a quarter fewer pairs in flight lets the 4 MB L2 keep more lines until their second
    // half is read.  Fixed-length batches (line form: every line requested once) run the four blocks the LDS allows: 16 waves
    // per CU measured 8 % faster than 12 once the refetches were gone (profiles/r02/ab_band_kernel.md).
    size_t lds = (size_t)P.lds_per_wave * wpb;
    // (a length-ordered CSR batch -- P.subset set by ta_levenshtein_k_batch -- runs the four blocks: 0.265 against 0.281 ms on the
    // ragged cfg2 batch, profiles/r03/ab_band_kernel.md)
    if (!line_form && wpb == BITS_WAVES_PER_BLOCK && lds < 53000u && max_len > 128u && !(P.subset && (P.a.off || P.b.off))) lds = 53000u;
    if (const char *e = env_str("TA_BITS_BLOCK_LDS")) { const size_t want = (size_t)atoi(e); if (want >= (size_t)P.lds_per_wave * wpb && want <= 160u * 1024u) lds = want; }
    if (grid_out) *grid_out = grid;
    if (lds_out) *lds_out = (uint32_t)lds;
    if (grid == 0) return hipSuccess;
    const bool early = line_form && pl.s8 && (P.tune & 2u) != 0u && !P.ckpt;
    if (pl.s8) set_last_kernel_name("lev_bits_s8_kernel<%s, %s, %s>", trans ? "true" : "false", line_form ? "true" : "false", early ? "true" : "false");
    else set_last_kernel_name("%s<%d, %s, %s>", line_form ? "lev_bits_line_kernel" : "lev_bits_kernel", pl.NA, trans ? "true" : "false", pl.stat ? "true" : "false");
    if (pl.s8 && P.ckpt) {                              // (the caller's promise: fixed-length batch, no subset list, no device-side count)
        dim3 g(grid), b(64 * wpb);
        set_last_kernel_name("lev_bits_s8_ckpt_kernel<%s, %s>", trans ? "true" : "false", line_form ? "true" : "false");
        if (trans) { if (line_form) hipLaunchKernelGGL((lev_bits_s8_ckpt_kernel<true, true>), g, b, lds, s, P); else hipLaunchKernelGGL((lev_bits_s8_ckpt_kernel<true, false>), g, b, lds, s, P); }
        else { if (line_form) hipLaunchKernelGGL((lev_bits_s8_ckpt_kernel<false, true>), g, b, lds, s, P); else hipLaunchKernelGGL((lev_bits_s8_ckpt_kernel<false, false>), g, b, lds, s, P); }
        return hipGetLastError();
    }
    if (pl.s8) {
        dim3 g(grid), b(64 * wpb);
        if (early) {
            if (trans) hipLaunchKernelGGL((lev_bits_s8_kernel<true, true, true>), g, b, lds, s, P);
            else hipLaunchKernelGGL((lev_bits_s8_kernel<false, true, true>), g, b, lds, s, P);
        } else if (trans) { if (line_form) hipLaunchKernelGGL((lev_bits_s8_kernel<true, true>), g, b, lds, s, P); else hipLaunchKernelGGL((lev_bits_s8_kernel<true, false>), g, b, lds, s, P); }
        else { if (line_form) hipLaunchKernelGGL((lev_bits_s8_kernel<false, true>), g, b, lds, s, P); else hipLaunchKernelGGL((lev_bits_s8_kernel<false, false>), g, b, lds, s, P); }
        return hipGetLastError();
    }
    switch (pl.NA) {
#define TA_CASE(n) case n: return launch_na<n>(P, trans, pl.stat, line_form, grid, wpb, lds, s);
        TA_CASE(1) TA_CASE(2) TA_CASE(3) TA_CASE(4) TA_CASE(5) TA_CASE(6) TA_CASE(7) TA_CASE(8)
        TA_CASE(9) TA_CASE(10) TA_CASE(11) TA_CASE(12) TA_CASE(13) TA_CASE(14) TA_CASE(15) TA_CASE(16)
        TA_CASE(18) TA_CASE(20) TA_CASE(22) TA_CASE(24) TA_CASE(26) TA_CASE(28) TA_CASE(30) TA_CASE(32)
#undef TA_CASE
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ta
