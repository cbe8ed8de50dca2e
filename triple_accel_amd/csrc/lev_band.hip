// lev_band.hip -- gfx950 instantiations of the band-wavefront kernel (lev_band_body.h).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "lev_trace_emit.h"
#include "lev_band_body.h"
#include "lev_plan.h"
#include "ta_internal.h"

namespace ta {

constexpr int LEV_WAVES_PER_BLOCK = 4;

// L1: one lane per pair (the whole band in one lane's D diagonals: bands of up to 66 diagonals with 64 pairs per wavefront)
template <int D, bool AFFINE, int TRANS, bool L1>
__global__ __launch_bounds__(64 * LEV_WAVES_PER_BLOCK) void lev_band_kernel(LevParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t wave = threadIdx.x >> 6, w = blockIdx.x * LEV_WAVES_PER_BLOCK + wave;
    if (P.n_dev) {                                     // a list a kernel before this one wrote (the rounds of ta_levenshtein_exp_batch): its length is read here
        LevParams Q = P;
        Q.n = *P.n_dev;
        if ((uint64_t)w * P.PW < Q.n) LevBand<DevWave, D, AFFINE, TRANS, false, L1>::run(Q, w, lds + wave * P.lds_per_wave);
        return;
    }
    LevBand<DevWave, D, AFFINE, TRANS, false, L1>::run(P, w, lds + wave * P.lds_per_wave);
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

// Trace kernels (trace_on = true): D = 16 (bands up to 1024 diagonals), 34 (batches: a band of up to 34 diagonals in ONE lane, 64 pairs per
// wavefront) or 66.
template <int D, bool AFFINE, int TRANS, bool L1 = false>
__global__ __launch_bounds__(64) void lev_band_trace_kernel(LevParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    LevBand<DevWave, D, AFFINE, TRANS, true, L1>::run(P, blockIdx.x, lds);
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
    if (pl.D == 16 && pl.L == 1 && !env_int("TA_TRACE_NO_L1")) {    // a band of up to 16 diagonals in one lane: the L1 instantiation, as for 34 (round 6: the generic one spent 247 instructions per iteration on 16 cells)
        if (affine) { if (trans) hipLaunchKernelGGL((lev_band_trace_kernel<16, true, 2, true>), g, b, lds, s, P);
                      else hipLaunchKernelGGL((lev_band_trace_kernel<16, true, 0, true>), g, b, lds, s, P); }
        else { if (trans) hipLaunchKernelGGL((lev_band_trace_kernel<16, false, 2, true>), g, b, lds, s, P);
               else hipLaunchKernelGGL((lev_band_trace_kernel<16, false, 0, true>), g, b, lds, s, P); }
    }
    else if (pl.D == 16) { TA_T(16) }
    else if (pl.D == 34 && pl.L == 1) {                      // the batch layout: the whole band in one lane -- no neighbour traffic (L1)
        if (affine) { if (trans) hipLaunchKernelGGL((lev_band_trace_kernel<34, true, 2, true>), g, b, lds, s, P);
                      else hipLaunchKernelGGL((lev_band_trace_kernel<34, true, 0, true>), g, b, lds, s, P); }
        else { if (trans) hipLaunchKernelGGL((lev_band_trace_kernel<34, false, 2, true>), g, b, lds, s, P);
               else hipLaunchKernelGGL((lev_band_trace_kernel<34, false, 0, true>), g, b, lds, s, P); }
    }
    else if (pl.D == 34) { TA_T(34) }
    else if (pl.D == 66) { TA_T(66) }
    else return hipErrorInvalidValue;
#undef TA_T
    return hipGetLastError();
}

// ---- batch tracebacks: the walk on the device (src/levenshtein.rs:561-606).  One wavefront per wavefront of the trace kernel: its lanes
// are that wavefront's pairs (PW of them), and ALL its lanes stream the wavefront's records -- one contiguous region, walked from the
// last anti-diagonal down -- through LDS, a window of WIN iterations (both phases, 64 lanes) at a time: a step of the walk is then an
// LDS read (the first version followed the codes through HBM, one dependent load per step: 18 ms per million 256-byte pairs against
// 3.6 for the kernel that computes them).  Phase A: every lane follows the 2-bit argmin codes of its pair from (n, m) back to (0, 0) and
// packs the codes it takes, sixteen per word, into `path` (scratch).  Phase B: the same lane replays its path FORWARD from (0, 0) --
// reading the two strings front to back to tell Match from Mismatch -- and writes the runs as it closes them: ta_edit records {edit, count},
// the reference's Vec<Edit> in its final order, into the pair's slot of `cap` records.  n_edits[pair] = the runs of the script (0 for
// None); a script longer than `cap` is cut (the caller sees n_edits > cap).
__global__ __launch_bounds__(64) void lev_trace_walk_kernel(StrView a, StrView b, const uint32_t *dist, const uint32_t *trace, uint64_t wave_words,
                                                            uint32_t D, uint32_t L, uint32_t PW, uint32_t tw, uint32_t u, uint32_t pair_base,
                                                            uint32_t n_chunk, uint32_t win, ta_edit *edits, uint32_t *n_edits, uint64_t cap,
                                                            uint32_t *path, uint32_t path_words) {
    extern __shared__ __attribute__((aligned(16))) uint32_t rows[];     // win iterations x 2 phases x 64 lanes x tw words
    const uint32_t lane = threadIdx.x, idx = blockIdx.x * PW + lane;
    const bool mine = lane < PW && idx < n_chunk;
    const uint32_t pair = pair_base + (mine ? idx : 0u);
    const uint8_t *x = a.blob, *y = b.blob;
    uint64_t n = 0, m = 0;
    bool some = false;
    if (mine) {
        if (a.off) { x = a.blob + a.off[pair]; n = a.off[pair + 1] - a.off[pair]; } else { x = a.blob + (uint64_t)pair * a.stride; n = a.len; }
        if (b.off) { y = b.blob + b.off[pair]; m = b.off[pair + 1] - b.off[pair]; } else { y = b.blob + (uint64_t)pair * b.stride; m = b.len; }
        some = dist[pair] != 0xFFFFFFFFu;
        if (!some) n_edits[pair] = 0;
    }
    const bool swap = n > m;                                   // the kernel ran the shorter string along the rows (:386-390)
    if (swap) { const uint8_t *t = x; x = y; y = t; const uint64_t tl = n; n = m; m = tl; }
    const uint64_t dlen = m - n, tb = dlen <= u ? (u - dlen) >> 1 : 0;            // lev_pair_offset (lev_plan.h) with n <= m
    const uint32_t o_pair = (uint32_t)tb | 1u;
    const uint32_t *tr = trace + (uint64_t)blockIdx.x * wave_words;
    const uint32_t lane0 = lane * L;
    // the path words of the chunk's pairs are interleaved: word w of pair idx at path[w * pitch + idx] (lev_trace_emit.h)
    const uint64_t pitch = (uint64_t)gridDim.x * PW;
    uint32_t *my_path = path + (mine ? idx : 0u);
    // ---- phase A: backwards through the records, window by window
    uint32_t i = some ? (uint32_t)n : 0u, j = some ? (uint32_t)m : 0u, steps = 0, acc = 0;
    uint32_t top = (i + j) ? (i + j - 1u) >> 1 : 0u;
    for (int off = 32; off >= 1; off >>= 1) { const uint32_t o2 = (uint32_t)__shfl_xor((int)top, off, 64); top = o2 > top ? o2 : top; }
    const uint32_t row_words = 2u * 64u * tw;
    for (int64_t hi = (int64_t)top; hi >= 0; hi -= (int64_t)win) {
        const uint32_t lo = hi + 1 >= (int64_t)win ? (uint32_t)(hi + 1 - (int64_t)win) : 0u;
        const uint32_t nwords = ((uint32_t)hi - lo + 1u) * row_words;                 // a multiple of 128
        __syncthreads();
        {
            typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(16)));
            const u32x4a *src = (const u32x4a *)(tr + (uint64_t)lo * row_words);
            u32x4a *dst = (u32x4a *)rows;
            for (uint32_t q = lane; q < nwords / 4u; q += 64u) dst[q] = src[q];
        }
        __syncthreads();
        while (i + j > 0u) {
            const uint32_t s = i + j, tau = (s - 1u) >> 1;
            if (tau < lo) break;
            const uint32_t p = j + o_pair - i, g = p / D, q = p % D, par = q & 1u, c = q >> 1;
            const uint32_t word = rows[(((tau - lo) * 2u + par) * 64u + lane0 + g) * tw + ((2u * c) >> 5)];
            const uint32_t code = (word >> ((2u * c) & 31u)) & 3u;
            if (code == 0u) { i--; j--; } else if (code == 1u) { j--; } else if (code == 2u) { i--; } else { i -= 2u; j -= 2u; }
            acc |= code << (2u * (steps & 15u));
            if ((steps & 15u) == 15u) { my_path[(uint64_t)(steps >> 4) * pitch] = acc; acc = 0; }
            steps++;
        }
        if (lo == 0u) break;
    }
    __syncthreads();                                           // (phase B reuses the rows' LDS for the lanes' string slots)
    if (!some) return;
    if (steps & 15u) my_path[(uint64_t)(steps >> 4) * pitch] = acc;
    if (cap == 0) return;                                      // (a timing probe, TA_TRACE_SKIP_EMIT=1: the walk without its replay -- no scripts)
    // ---- phase B: the path forwards, runs written as they close (lev_trace_emit.h); the strings 64 bytes at a time through the lane's LDS slots
    uint8_t *xs = (uint8_t *)rows + lane * TRACE_EMIT_SLOT, *ys = (uint8_t *)rows + (64u + lane) * TRACE_EMIT_SLOT;
    n_edits[pair] = trace_emit_runs_lds(my_path, pitch, steps, x, (uint32_t)n, y, (uint32_t)m, swap, edits + (uint64_t)pair * cap, cap, xs, ys);
}

// a chunk of a batch: pairs [pair_base, pair_base + n_chunk) through the trace kernel (records into `trace`), then the walk
hipError_t lev_band_trace_batch_launch(const LevParams &P, const LevPlan &pl, bool affine, bool trans, ta_edit *edits, uint32_t *n_edits,
                                       uint64_t cap, uint32_t *path, uint32_t path_words, hipStream_t s) {
    if (P.n == 0) return hipSuccess;
    hipError_t e = lev_band_trace_launch(P, pl, affine, trans, s);
    if (e != hipSuccess) return e;
    const uint32_t tw = (uint32_t)lev_trace_words(pl.D), waves = (P.n + pl.PW - 1) / pl.PW;
    const uint32_t win = tw >= 4u ? 4u : 16u / tw;                     // <= 8 KB of rows per wavefront: twenty of them per CU
    size_t walk_lds = (size_t)win * 2u * 64u * tw * 4u;
    if (walk_lds < 2u * 64u * TRACE_EMIT_SLOT) walk_lds = 2u * 64u * TRACE_EMIT_SLOT;    // (the replay's string slots live in the same bytes)
    hipLaunchKernelGGL(lev_trace_walk_kernel, dim3(waves), dim3(64), walk_lds, s, P.a, P.b, P.out, P.trace, P.trace_wave_words,
                       (uint32_t)pl.D, pl.L, pl.PW, tw, pl.u, P.pair_base, P.n, win, edits, n_edits, env_int("TA_TRACE_SKIP_EMIT") ? 0ull : cap, path, path_words);
    return hipGetLastError();
}

}  // namespace ta
