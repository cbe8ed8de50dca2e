// lev_wide.hip -- wide-band / full-matrix Levenshtein: one wavefront per pair, row-striped.
//
// Used when the band needs more diagonals than the band-wavefront kernel holds in registers
// (unit_k > ~2100: levenshtein(), rdamerau() and the last doublings of levenshtein_exp on long strings,
// src/levenshtein.rs:1397-1526).  Result contract as lev_band_body.h: d if d <= k else None; recurrence
// of the scalar path (src/levenshtein.rs:471-532) with
//     A(i,j) = min(dp(i,j-1) + sg + gc, A(i,j-1) + gc)      gap along j   (a_gap)
//     B(i,j) = min(dp(i-1,j) + sg + gc, B(i-1,j) + gc)      gap along i   (b_gap)
//
// Mapping.  A stripe is 64*R consecutive rows of `a`; lane t owns R of them and keeps their previous
// column (dp, and A of the column to come) in VGPRs.  At step s lane t computes column j = jlo + s - t of
// its rows top to bottom, so the R-cell dependency chain runs inside a lane and the only cross-lane
// traffic per step is the bottom row handed to lane t+1 (DPP wave_shr:1) plus the column's b byte.
// Only the columns a stripe's rows can reach inside the pair's band (lev_plan.h) are visited.
// Strings longer than one stripe are processed stripe by stripe; the stripe's last row goes to a per-wave
// HBM scratch line and comes back as the next stripe's top boundary (64 columns per coalesced load,
// handed to lane 0 with v_readlane).  Waves are persistent: grid = resident waves, pairs strided.
#include <hip/hip_runtime.h>

#include "ta_internal.h"

namespace ta {

constexpr int WR = 32;                       // rows per lane
constexpr int WROWS = 64 * WR;               // rows per stripe
constexpr uint32_t WINF = 0x3FFFFFFFu;

struct WideScratch {
    uint32_t *buf;          // per wave: 2 (ping-pong) x 3 (dp, B of the row below, dp of the row above) x line
    uint64_t line;          // u32 elements per line (>= max_len + 2)
};

__device__ __forceinline__ uint32_t dpp_from_lower(uint32_t x, uint32_t fill) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)x, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t umin_(uint32_t a, uint32_t b) { return a < b ? a : b; }

__device__ __forceinline__ void wide_str(const StrView &s, uint32_t i, const uint8_t *&p, uint32_t &len) {
    if (s.off) {
        uint64_t o0 = s.off[i], o1 = s.off[i + 1];
        p = s.blob + o0;
        len = (uint32_t)(o1 - o0);
    } else {
        p = s.blob + (uint64_t)i * s.stride;
        len = (uint32_t)s.len;
    }
}

// TRACE (one pair, trace_on = true): 2-bit argmin codes with the scalar tie order (src/levenshtein.rs:493-532), 32 rows =
// 2 dwords per lane per column, at P.trace[((stripe * P.trace_cols + j) * 64 + lane) * 2 + word]
template <bool AFFINE, bool TRANS, bool TRACE = false>
__global__ __launch_bounds__(64) void lev_wide_kernel(LevParams P, WideScratch S) {
    const uint32_t t = threadIdx.x;   // lane
    const uint32_t gc = P.gc, sg = P.sg, sgc = P.sg + P.gc, mc = P.mc, tc = P.tc, u = P.u;
    uint32_t *line_base = S.buf + (uint64_t)blockIdx.x * (6 * S.line);
    auto col0 = [&](uint32_t i) -> uint32_t { return i ? i * gc + sg : 0u; };   // dp(i, 0)

    const uint32_t n_pairs = P.n_dev ? *P.n_dev : P.n;   // (a list whose length only the device knows: the rounds of ta_levenshtein_exp_batch)
    for (uint32_t slot = blockIdx.x; slot < n_pairs; slot += gridDim.x) {
        const uint32_t pair = P.subset ? P.subset[slot] : slot;
        const uint8_t *ap, *bp;
        uint32_t n, m;
        wide_str(P.a, pair, ap, n);
        wide_str(P.b, pair, bp, m);
        const uint32_t diff = n > m ? n - m : m - n;
        if (diff > u) { if (t == 0) P.out[pair] = 0xFFFFFFFFu; continue; }             // :426-428
        if (n == 0 || m == 0) {                                                          // one gap run or ("", "")
            uint32_t d = col0(n + m);
            if (t == 0) P.out[pair] = d <= P.k ? d : 0xFFFFFFFFu;
            continue;
        }
        const uint32_t tband = (u - diff) >> 1;
        const uint32_t below = tband + (n > m ? diff : 0u), above = tband + (m > n ? diff : 0u);
        uint32_t ans = WINF;
        uint32_t plo = 1, phi = 0;                       // previous stripe's column range (empty for stripe 0)
        const uint32_t stripes = (n + WROWS - 1) / WROWS;
        const uint32_t t_ans = ((n - 1) % WROWS) / WR, r_ans = ((n - 1) % WROWS) % WR;
        for (uint32_t q = 0; q < stripes; q++) {
            const uint32_t i0 = q * WROWS;               // the stripe's rows are i0+1 .. i0+WROWS
            const uint32_t i_last = (i0 + WROWS < n) ? i0 + WROWS : n;
            // the pair's band (lev_plan.h): row i visits columns [i - below, i + above]
            const uint32_t jlo = (i0 + 1 > below) ? i0 + 1 - below : 1;
            const uint32_t jhi = ((uint64_t)i_last + above < m) ? i_last + above : m;
            const uint32_t Cn = jhi - jlo + 1;           // >= 1 because |n - m| <= u
            const uint32_t *rd = line_base + (uint64_t)((q + 1) & 1) * 3 * S.line;   // written by stripe q-1
            uint32_t *wr = line_base + (uint64_t)(q & 1) * 3 * S.line;
            const bool last_stripe = (q + 1 == stripes);
            auto top_dp = [&](uint32_t j) -> uint32_t {  // dp(i0, j)
                if (q == 0) return col0(j);
                if (j == 0) return col0(i0);
                return (j >= plo && j <= phi) ? rd[j] : WINF;
            };
            auto top_dp2 = [&](uint32_t j) -> uint32_t { // dp(i0-1, j)
                if (q == 0) return WINF;
                if (j == 0) return col0(i0 - 1);
                return (j >= plo && j <= phi) ? rd[2 * S.line + j] : WINF;
            };

            // this lane's rows: chars packed 4 per VGPR (byte r&3 of word r>>2 = char of row row0 + 1 + r)
            const uint32_t row0 = i0 + t * WR;           // the row just above the lane's first row
            uint32_t A4[WR / 4], AU4[TRANS ? WR / 4 : 1];
#pragma unroll
            for (int w = 0; w < WR / 4; w++) {
                uint32_t v = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    uint32_t ci = row0 + w * 4 + b;      // 0-based index into a
                    v |= ((ci < n) ? (uint32_t)ap[ci] : 0u) << (8 * b);
                }
                A4[w] = v;
            }
            if (TRANS) {                                 // AU4: char of the row above each row
                uint32_t above = (row0 >= 1 && row0 - 1 < n) ? (uint32_t)ap[row0 - 1] : 0u;
#pragma unroll
                for (int w = 0; w < WR / 4; w++)
                    AU4[w] = __builtin_amdgcn_alignbyte(A4[w], w ? A4[w - 1] : (above << 24), 3);
            }

            // state of the column left of the stripe's first column (jlo - 1)
            uint32_t H[WR], GA[WR], P2[TRANS ? WR : 1];
#pragma unroll
            for (int r = 0; r < WR; r++) {
                uint32_t h = (jlo == 1) ? col0(row0 + 1 + r) : WINF;
                H[r] = h;
                GA[r] = h + (AFFINE ? sgc : gc);         // A(i, jlo) = open from dp(i, jlo-1); A(i, jlo-1) itself is INF
                if (TRANS) P2[r] = WINF;
            }
            // delay lines of the two rows above the lane's first row: dp(row0, .) and dp(row0-1, .)
            uint32_t up_dp, up_dp_prev, up_dp_prev2 = WINF, up2_dp = WINF, up2_prev = WINF, up2_prev2 = WINF;
            if (t == 0) {
                up_dp = top_dp(jlo - 1);
                up_dp_prev = (jlo >= 2) ? top_dp(jlo - 2) : WINF;
                if (TRANS) { up2_dp = top_dp2(jlo - 1); up2_prev = (jlo >= 2) ? top_dp2(jlo - 2) : WINF; }
            } else {
                up_dp = (jlo == 1) ? col0(row0) : WINF;
                up_dp_prev = WINF;
                if (TRANS) up2_dp = (jlo == 1) ? col0(row0 - 1) : WINF;
            }
            uint32_t bch = (jlo >= 2) ? (uint32_t)bp[jlo - 2] : 0u;   // char of column jlo - 1
            uint32_t send_dp = WINF, send_gb = WINF, send_dp2 = WINF, send_b = 0;

            uint32_t cb = 0, cdp = WINF, cgb = WINF, cdp2 = WINF;     // lane e holds column jlo + 64*chunk + e
            const uint32_t steps = Cn + 63;
            for (uint32_t s = 0; s < steps; s++) {
                if ((s & 63u) == 0) {                    // coalesced fetch of the next 64 columns for lane 0
                    uint32_t j = jlo + s + t;
                    cb = (j <= m) ? (uint32_t)bp[j - 1] : 0u;
                    if (q == 0) {
                        cdp = col0(j);                   // row 0 (:450-452)
                        cgb = cdp + (AFFINE ? sgc : gc); // B(1, j) = open from dp(0, j)
                        cdp2 = WINF;
                    } else {
                        bool in = (j >= plo && j <= phi);
                        cdp = in ? rd[j] : WINF;
                        cgb = in ? rd[S.line + j] : WINF;
                        cdp2 = (TRANS && in) ? rd[2 * S.line + j] : WINF;
                    }
                }
                const uint32_t e = s & 63u;
                const uint32_t b0 = __builtin_amdgcn_readlane(cb, e);
                const uint32_t t_dp = __builtin_amdgcn_readlane(cdp, e);
                const uint32_t t_gb = __builtin_amdgcn_readlane(cgb, e);
                const uint32_t t_dp2 = TRANS ? __builtin_amdgcn_readlane(cdp2, e) : WINF;

                // what the lane above produced in the previous step (its column j), or the boundary for lane 0
                uint32_t in_dp = dpp_from_lower(send_dp, WINF);
                uint32_t in_gb = dpp_from_lower(send_gb, WINF);
                uint32_t in_dp2 = TRANS ? dpp_from_lower(send_dp2, WINF) : WINF;
                uint32_t in_b = dpp_from_lower(send_b, 0u);
                if (t == 0) { in_dp = t_dp; in_gb = t_gb; in_dp2 = t_dp2; in_b = b0; }

                const bool active = (s >= t) && (s - t < Cn);
                if (active) {
                    const uint32_t j = jlo + s - t;
                    up_dp_prev2 = up_dp_prev; up_dp_prev = up_dp; up_dp = in_dp;
                    if (TRANS) { up2_prev2 = up2_prev; up2_prev = up2_dp; up2_dp = in_dp2; }
                    const uint32_t bprev = bch;
                    bch = in_b;
                    const uint32_t B4 = bch * 0x01010101u;
                    uint32_t F4[WR / 4], Z4[TRANS ? WR / 4 : 1];
#pragma unroll
                    for (int w = 0; w < WR / 4; w++) {
                        // 1 per mismatching row: the XOR plus 0x0C is 12 exactly where the bytes agree, and v_perm_b32 with all-ones
                        // sources maps byte value 12 to 0x00 and every other one to 0xFF (wave.h, W::ne12)
                        F4[w] = __builtin_amdgcn_perm(0xFFFFFFFFu, 0xFFFFFFFFu, A4[w] ^ B4 ^ 0x0C0C0C0Cu) & 0x01010101u;
                        if (TRANS) Z4[w] = (A4[w] ^ (bprev * 0x01010101u)) | (AU4[w] ^ B4);   // 0: a[i]==b[j-1] && a[i-1]==b[j]
                    }
                    uint32_t diag = up_dp_prev;          // dp(row0, j-1)
                    uint32_t gb = in_gb;                 // B(row0 + 1, j)
                    uint32_t p2a = up2_prev2;            // dp(row0 - 1, j-2) -> transposition of row r = 0
                    uint32_t p2b = up_dp_prev2;          // dp(row0,     j-2) -> row r = 1
                    uint32_t v = WINF, v_above = WINF;
                    uint32_t codes[2] = {0u, 0u};
#pragma unroll
                    for (int r = 0; r < WR; r++) {
                        const uint32_t sub = __builtin_amdgcn_udot4(F4[r >> 2], mc << (8 * (r & 3)), diag, false);   // :471-475
                        const uint32_t oldH = H[r];
                        const uint32_t ga = GA[r];
                        v_above = v;
                        v = umin_(umin_(sub, ga), gb);                                   // :493-515
                        uint32_t code = 0;
                        if (TRACE) code = (gb < umin_(sub, ga)) ? 2u : ((ga < sub) ? 1u : 0u);   // sub, then a_gap (<), then b_gap (<)
                        if (TRANS) {
                            const uint32_t td = p2a;                                     // dp(i-2, j-2)
                            p2a = p2b; p2b = P2[r]; P2[r] = oldH;
                            const bool tz = ((Z4[r >> 2] >> (8 * (r & 3))) & 0xffu) == 0u;
                            const uint32_t tv = td + tc;
                            if (TRACE) code = (tz && tv <= v) ? 3u : code;               // transposition wins ties (<=)
                            v = (tz && tv < v) ? tv : v;                                 // :517-532
                        }
                        if (TRACE) codes[r >> 4] |= code << (2 * (r & 15));
                        diag = oldH;
                        H[r] = v;
                        if (AFFINE) {
                            const uint32_t open = v + sgc;
                            GA[r] = umin_(open, ga + gc);                                // A(i, j+1)
                            gb = umin_(open, gb + gc);                                   // B(i+1, j)
                        } else {
                            gb = v + gc;
                            GA[r] = gb;
                        }
                    }
                    if (TRACE) {
                        uint32_t *tr = P.trace + (((uint64_t)q * P.trace_cols + j) * 64u + t) * 2u;
                        tr[0] = codes[0]; tr[1] = codes[1];
                    }
                    send_dp = v; send_gb = gb; send_dp2 = v_above; send_b = bch;
                    if (!last_stripe && t == 63) {       // the stripe's last row -> next stripe's top boundary
                        wr[j] = v;
                        wr[S.line + j] = gb;
                        if (TRANS) wr[2 * S.line + j] = v_above;
                    }
                    if (last_stripe && j == m && t == t_ans) {
                        uint32_t a_ = WINF;
#pragma unroll
                        for (int r = 0; r < WR; r++) a_ = (r_ans == (uint32_t)r) ? H[r] : a_;
                        ans = a_;
                    }
                }
            }
            plo = jlo; phi = jhi;
            if (!last_stripe) __threadfence();           // lane 63's boundary stores -> this wave's next loads (L1 invalidate)
        }
        const uint32_t d = __builtin_amdgcn_readlane(ans, t_ans);
        if (t == 0) P.out[pair] = (d <= P.k && d < WINF) ? d : 0xFFFFFFFFu;             // :539-541
    }
}

// P.lds_per_wave carries max_len + 2 (elements per boundary line) for this kernel.
hipError_t lev_wide_launch(const LevParams &P, bool trans, hipStream_t s, uint32_t *grid_out, uint32_t *lds_out,
                           uint32_t *threads_out, uint32_t *dpt_out) {
    if (lds_out) *lds_out = 0;
    if (threads_out) *threads_out = 64;
    if (dpt_out) *dpt_out = WR;
    int dev = 0, cus = 256;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    uint32_t resident = (uint32_t)cus * 4u * 4u;            // up to 4 waves per SIMD at this register footprint
    uint32_t grid = P.n < resident ? P.n : resident;
    if (grid_out) *grid_out = grid;
    if (grid == 0) return hipSuccess;
    WideScratch S;
    S.line = (uint64_t)P.lds_per_wave;
    // boundary scratch (only touched when a string spans more than one stripe)
    Scratch &sc = tls_scratch(6);
    if (sc.ensure((size_t)grid * 6 * S.line * sizeof(uint32_t)) != TA_OK) return hipErrorOutOfMemory;
    S.buf = (uint32_t *)sc.dev;
    const bool affine = P.sg > 0;
    set_last_kernel_name("lev_wide_kernel<%s, %s, false>", affine ? "true" : "false", trans ? "true" : "false");
    if (affine && trans) hipLaunchKernelGGL((lev_wide_kernel<true, true>), dim3(grid), dim3(64), 0, s, P, S);
    else if (affine) hipLaunchKernelGGL((lev_wide_kernel<true, false>), dim3(grid), dim3(64), 0, s, P, S);
    else if (trans) hipLaunchKernelGGL((lev_wide_kernel<false, true>), dim3(grid), dim3(64), 0, s, P, S);
    else hipLaunchKernelGGL((lev_wide_kernel<false, false>), dim3(grid), dim3(64), 0, s, P, S);
    return hipGetLastError();
}

// trace_on = true for ONE pair (P.n == 1, P.trace / P.trace_cols set): same sweep, argmin codes stored
hipError_t lev_wide_trace_launch(const LevParams &P, bool trans, hipStream_t s) {
    WideScratch S;
    S.line = (uint64_t)P.lds_per_wave;
    Scratch &sc = tls_scratch(6);
    if (sc.ensure((size_t)6 * S.line * sizeof(uint32_t)) != TA_OK) return hipErrorOutOfMemory;
    S.buf = (uint32_t *)sc.dev;
    const bool affine = P.sg > 0;
    if (affine && trans) hipLaunchKernelGGL((lev_wide_kernel<true, true, true>), dim3(1), dim3(64), 0, s, P, S);
    else if (affine) hipLaunchKernelGGL((lev_wide_kernel<true, false, true>), dim3(1), dim3(64), 0, s, P, S);
    else if (trans) hipLaunchKernelGGL((lev_wide_kernel<false, true, true>), dim3(1), dim3(64), 0, s, P, S);
    else hipLaunchKernelGGL((lev_wide_kernel<false, false, true>), dim3(1), dim3(64), 0, s, P, S);
    return hipGetLastError();
}

}  // namespace ta
