// lev_search_body.h -- semi-global (free start in the haystack) Levenshtein search over one haystack tile.
//
// Replaces levenshtein_search_simd_core_* (src/levenshtein.rs:2157-2451); the per-cell recurrence,
// the companion match-length and every tie rule follow the SCALAR text levenshtein_search_naive_with_opts
// (src/levenshtein.rs:1709-1806), which is the bit-exactness target.
//
// One lane owns one tile of consecutive haystack end positions and keeps the whole DP column
// (needle rows 0..n, cost + length + the needle-gap state) in VGPRs, advancing one haystack byte per
// step; a tile starts `halo` bytes early with the fresh-start column so that every cell of cost <= k it
// reports is exact (any alignment of cost <= k spans at most n + unit_k haystack bytes).
// Plain host/device code (no cross-lane traffic), so tests run the same function on the CPU.
#pragma once
#include <stdint.h>

#include "wave.h"

#include "lev_plan.h"

namespace ta {

constexpr uint32_t SRCH_INF = 0x3FFFFFFFu;
// "no gap yet" cost of the packed form (lev_search_tile_packed): every real cost must stay below it (the host checks)
constexpr uint32_t SRCH_PACKED_KINF = 0x7000u;
// Anchored searches (src/levenshtein.rs:1650-1658, :1710-1719): row 0 costs (i+1)*gc + sg, so every cell is bounded by row 0
// at the last column plus a whole needle of gaps / mismatches.  The packed form may only run while that bound stays below
// KINF -- otherwise its continuation KINF + gc would beat a real (larger) candidate.  h = columns the search visits.
static inline bool srch_anchored_packed_ok(uint64_t h, uint32_t needle_len, uint32_t mc, uint32_t gc, uint32_t sg) {
    const uint64_t wc = mc > gc ? mc : gc;
    const uint64_t top = (h + 1) * gc + 2ull * sg + ((uint64_t)needle_len + 1) * wc + gc;
    return top < SRCH_PACKED_KINF;
}

// The candidate filter (lev_filter_body.h) scans with UNIT costs; under any other EditCosts it runs with k' = lev_unit_filter_k
// (lev_plan.h) and is still a SUPERSET filter: every block that holds a weighted hit is flagged, and the exact kernel (which knows the
// real costs) runs on the flagged blocks only.  The scan's left context must cover a unit-cost match of k' edits: needle_len + k' + 2
// columns (>= the exact kernel's halo, since k' >= unit_k = (k - sg) / gc).
static inline uint32_t srch_filter_k(uint32_t k, uint32_t mc, uint32_t gc, uint32_t sg, bool has_t, uint32_t tc) {
    return lev_unit_filter_k(k, mc, gc, sg, has_t, tc);
}

struct SearchCosts {
    uint32_t k, mc, gc, sg, tc;
    uint32_t anchored;
};

// Emit(end_local_index_plus_1, length, cost) is called for every reported column in increasing order.
template <int N, bool TRANS, class Emit>
TA_HD inline void lev_search_tile(const uint8_t *hay, const uint8_t *needle, uint32_t n, const SearchCosts &C,
                                  uint64_t col_begin, uint64_t emit_begin, uint64_t col_end, Emit emit) {
    uint32_t dp1[N + 1], l1[N + 1], ng[N + 1], ngl[N + 1];
    uint32_t dp0[TRANS ? N + 1 : 1], l0[TRANS ? N + 1 : 1];
    const uint32_t sgc = C.sg + C.gc;
#pragma unroll
    for (int j = 0; j <= N; j++) {                       // the closure's first call, :1685-1690
        dp1[j] = (uint32_t)j * C.gc + (j == 0 ? 0u : C.sg);
        l1[j] = 0; ng[j] = SRCH_INF; ngl[j] = 0;
        if (TRANS) { dp0[j] = 0; l0[j] = 0; }
    }
    if (col_begin >= col_end) return;
    uint32_t c_prev = 0;
    uint32_t c_next = hay[col_begin];
    for (uint64_t i = col_begin; i < col_end; i++) {
        const uint32_t c = c_next;
        if (i + 1 < col_end) c_next = hay[i + 1];        // one-ahead prefetch of the lane's byte stream
        const bool first_col = (i == col_begin);         // transposition needs a previous haystack byte (:1769)
        // row 0  (:1710-1721)
        const uint32_t c0 = C.anchored ? ((uint32_t)(i + 1)) * C.gc + C.sg : 0u;
        uint32_t up_dp = c0, up_l = 0;                   // dp2[j-1], length2[j-1]
        uint32_t diag_dp = dp1[0], diag_l = l1[0];       // dp1[j-1], length1[j-1]
        uint32_t hg = SRCH_INF, hgl = 0;                 // haystack_gap_dp[j-1], haystack_gap_length[j-1]
        uint32_t z1_dp = 0, z1_l = 0, z2_dp = 0, z2_l = 0;   // dp0[j-1], dp0[j-2] before they are overwritten
        if (TRANS) { z1_dp = dp0[0]; z1_l = l0[0]; dp0[0] = dp1[0]; l0[0] = l1[0]; }
        dp1[0] = c0; l1[0] = 0;
#pragma unroll
        for (int j = 1; j <= N; j++) {
            if ((uint32_t)j <= n) {
                const uint32_t nb = needle[j - 1];
                const uint32_t old_dp = dp1[j], old_l = l1[j];
                uint32_t sub = diag_dp + (nb != c ? C.mc : 0u);                          // :1724

                uint32_t new_gap = old_dp + sgc;                                         // :1726-1737
                uint32_t cont_gap = ng[j] + C.gc;
                uint32_t g_l = (new_gap < cont_gap) ? old_l
                             : (new_gap > cont_gap) ? ngl[j]
                             : (old_l > ngl[j] ? old_l : ngl[j]);
                ng[j] = new_gap < cont_gap ? new_gap : cont_gap;
                ngl[j] = g_l + 1;

                uint32_t new_gap2 = up_dp + sgc;                                         // :1739-1750
                uint32_t cont_gap2 = hg + C.gc;
                uint32_t h_l = (new_gap2 < cont_gap2) ? up_l
                             : (new_gap2 > cont_gap2) ? hgl
                             : (up_l > hgl ? up_l : hgl);
                hg = new_gap2 < cont_gap2 ? new_gap2 : cont_gap2;
                hgl = h_l;

                uint32_t v = ng[j], vl = ngl[j];                                         // :1752-1753
                if ((hg < v) || (hg == v && up_l > vl)) { v = hg; vl = hgl; }            // :1755-1760 (reads length2[j-1])
                if ((sub < v) || (sub == v && (diag_l + 1) > vl)) { v = sub; vl = diag_l + 1; }   // :1762-1765
                if (TRANS) {
                    const uint32_t t_dp = z2_dp, t_l = z2_l;                             // dp0[j-2], length0[j-2]
                    z2_dp = z1_dp; z2_l = z1_l;
                    z1_dp = dp0[j]; z1_l = l0[j];
                    dp0[j] = old_dp; l0[j] = old_l;
                    if (j > 1 && !first_col && nb == c_prev && (uint32_t)needle[j - 2] == c) {   // :1767-1779
                        uint32_t t = t_dp + C.tc;
                        if (t <= v) { v = t; vl = t_l + 2; }
                    }
                }
                dp1[j] = v; l1[j] = vl;
                diag_dp = old_dp; diag_l = old_l;
                up_dp = v; up_l = vl;
            }
        }
        c_prev = c;
        const uint32_t res = up_dp, len = up_l;          // dp2[len-1], length2[len-1] (:1782-1783)
        if (res <= C.k && i >= emit_begin) emit(i + 1, len, res);     // :1792-1806
    }
}

// Packed form: cost and match length share one VGPR, key = (cost << 16) | (0xFFFF - length), so the reference's
// "cheaper wins, on a tie the LONGER match wins" (src/levenshtein.rs:1726-1750, 1762-1765) is ONE v_min_u32, and
// "length + 1" is "key - 1".  The one rule that is not a plain min -- the haystack-gap candidate is compared with
// length2[j-1] but delivers haystack_gap_length[j] (:1755-1760, quirk Q2) -- splices the two 16-bit fields with a
// v_bfi before the compare.  Valid while every cost stays below 2^16 and lengths below 0xFFFF: the host selects this
// kernel for needle_len <= 32, k <= 30000 and tiles + halo <= 60000 columns (costs are bounded by
// min(j*mc, j*gc+sg) + k <= 8415 + k there), anything else takes lev_search_tile / lev_search_tile_mem.
template <int N, bool TRANS, class Emit>
TA_HD inline void lev_search_tile_packed(const uint8_t *hay, const uint8_t *needle, uint32_t /*n == N*/, const SearchCosts &C,
                                         uint64_t col_begin, uint64_t emit_begin, uint64_t col_end, Emit emit) {
    uint32_t dp1[N + 1], ng[N + 1], dp0[TRANS ? N + 1 : 1];
    const uint32_t SGC = (C.sg + C.gc) << 16, GC = C.gc << 16;
    const uint32_t SUB_MIS = (C.mc << 16) - 1u, SUB_EQ = 0xFFFFFFFFu;   // (+mc, length+1) and (+0, length+1)
    const uint32_t TCK = (C.tc << 16) - 2u;                              // (+tc, length+2)
    constexpr uint32_t KINF = (SRCH_PACKED_KINF << 16) | 0xFFFFu;
#pragma unroll
    for (int j = 0; j <= N; j++) {
        dp1[j] = (((uint32_t)j * C.gc + (j == 0 ? 0u : C.sg)) << 16) | 0xFFFFu;
        ng[j] = KINF;
        if (TRANS) dp0[j] = 0xFFFFu;
    }
    if (col_begin >= col_end) return;
    uint32_t c_prev = 0;
    uint32_t c_next = hay[col_begin];
    for (uint64_t i = col_begin; i < col_end; i++) {
        const uint32_t c = c_next;
        if (i + 1 < col_end) c_next = hay[i + 1];
        const bool later_col = (i != col_begin);
        const uint32_t c0 = C.anchored ? ((((uint32_t)(i + 1)) * C.gc + C.sg) << 16) | 0xFFFFu : 0xFFFFu;
        uint32_t up = c0;                 // dp2[j-1] with its length
        uint32_t diag = dp1[0];           // dp1[j-1]
        uint32_t hg = KINF;               // haystack_gap_dp[j-1] with haystack_gap_length[j-1]
        uint32_t z1 = 0, z2 = 0;
        if (TRANS) { z1 = dp0[0]; dp0[0] = dp1[0]; }
        dp1[0] = c0;
#pragma unroll
        for (int j = 1; j <= N; j++) {
          {                                 // N == needle length exactly (one instantiation per length): no row test
            const uint32_t nb = needle[j - 1 < 32 ? j - 1 : 31];
            const uint32_t old = dp1[j];
            // needle gap (consumes the haystack byte): open from dp1[j] or extend; length + 1          :1726-1737
            uint32_t a1 = old + SGC, a2 = ng[j] + GC;
            const uint32_t ngk = (a1 < a2 ? a1 : a2) - 1u;
            ng[j] = ngk;
            // haystack gap (skips a needle char): open from dp2[j-1] or extend; length unchanged        :1739-1750
            uint32_t h1 = up + SGC, h2 = hg + GC;
            hg = h1 < h2 ? h1 : h2;
            uint32_t v = ngk;                                                                            // :1752-1753
            const uint32_t hq = (hg & 0xFFFF0000u) | (up & 0xFFFFu);     // cost of the gap, length2[j-1]  (Q2)
            v = (hq < v) ? hg : v;                                                                       // :1755-1760
            const uint32_t subk = diag + ((nb != c) ? SUB_MIS : SUB_EQ);                                 // :1724
            v = subk < v ? subk : v;                                                                     // :1762-1765
            if (TRANS) {
                const uint32_t t = z2 + TCK;                              // dp0[j-2] + tc, length0[j-2] + 2
                z2 = z1; z1 = dp0[j]; dp0[j] = old;
                const bool cond = (j > 1) & later_col & (nb == c_prev) & ((uint32_t)needle[j >= 2 ? j - 2 : 0] == c);
                v = (cond & ((t & 0xFFFF0000u) <= v)) ? t : v;                                          // :1767-1779 (<=)
            }
            dp1[j] = v;
            diag = old;
            up = v;
          }
        }
        c_prev = c;
        const uint32_t res_key = up;      // dp2[needle_len] with its length
        const uint32_t res = res_key >> 16;
        if (res <= C.k && i >= emit_begin) emit(i + 1, 0xFFFFu - (res_key & 0xFFFFu), res);             // :1792-1806
    }
}

// Same recurrence with the DP column kept in memory (needles longer than the register kernel's 32 rows).
// Element j of array `arr` of this tile lives at col[(arr * (n + 1) + j) * stride] -- on the GPU `stride` is the
// number of tiles so that neighbouring lanes touch neighbouring addresses (coalesced); arrays: 0 dp1, 1 l1,
// 2 ng, 3 ngl, 4 dp0, 5 l0.
template <class Emit>
TA_HD inline void lev_search_tile_mem(const uint8_t *hay, const uint8_t *needle, uint32_t n, const SearchCosts &C,
                                      bool trans, uint32_t *col, uint64_t stride,
                                      uint64_t col_begin, uint64_t emit_begin, uint64_t col_end, Emit emit) {
    const uint32_t sgc = C.sg + C.gc;
    const uint64_t rows = (uint64_t)n + 1;
    auto at = [&](uint32_t arr, uint32_t j) -> uint32_t & { return col[((uint64_t)arr * rows + j) * stride]; };
    for (uint32_t j = 0; j <= n; j++) {
        at(0, j) = j * C.gc + (j == 0 ? 0u : C.sg);
        at(1, j) = 0; at(2, j) = SRCH_INF; at(3, j) = 0;
        if (trans) { at(4, j) = 0; at(5, j) = 0; }
    }
    if (col_begin >= col_end) return;
    uint32_t c_prev = 0;
    for (uint64_t i = col_begin; i < col_end; i++) {
        const uint32_t c = hay[i];
        const bool first_col = (i == col_begin);
        const uint32_t c0 = C.anchored ? ((uint32_t)(i + 1)) * C.gc + C.sg : 0u;
        uint32_t up_dp = c0, up_l = 0;
        uint32_t diag_dp = at(0, 0), diag_l = at(1, 0);
        uint32_t hg = SRCH_INF, hgl = 0;
        uint32_t z1_dp = 0, z1_l = 0, z2_dp = 0, z2_l = 0;
        if (trans) { z1_dp = at(4, 0); z1_l = at(5, 0); at(4, 0) = diag_dp; at(5, 0) = diag_l; }
        at(0, 0) = c0; at(1, 0) = 0;
        uint32_t nb_prev = 0;
        for (uint32_t j = 1; j <= n; j++) {
            const uint32_t nb = needle[j - 1];
            const uint32_t old_dp = at(0, j), old_l = at(1, j);
            const uint32_t ngj = at(2, j), nglj = at(3, j);
            uint32_t sub = diag_dp + (nb != c ? C.mc : 0u);

            uint32_t new_gap = old_dp + sgc, cont_gap = ngj + C.gc;
            uint32_t g_l = (new_gap < cont_gap) ? old_l : (new_gap > cont_gap) ? nglj : (old_l > nglj ? old_l : nglj);
            const uint32_t ng_new = new_gap < cont_gap ? new_gap : cont_gap, ngl_new = g_l + 1;
            at(2, j) = ng_new; at(3, j) = ngl_new;

            uint32_t new_gap2 = up_dp + sgc, cont_gap2 = hg + C.gc;
            uint32_t h_l = (new_gap2 < cont_gap2) ? up_l : (new_gap2 > cont_gap2) ? hgl : (up_l > hgl ? up_l : hgl);
            hg = new_gap2 < cont_gap2 ? new_gap2 : cont_gap2;
            hgl = h_l;

            uint32_t v = ng_new, vl = ngl_new;
            if ((hg < v) || (hg == v && up_l > vl)) { v = hg; vl = hgl; }
            if ((sub < v) || (sub == v && (diag_l + 1) > vl)) { v = sub; vl = diag_l + 1; }
            if (trans) {
                const uint32_t t_dp = z2_dp, t_l = z2_l;
                z2_dp = z1_dp; z2_l = z1_l;
                z1_dp = at(4, j); z1_l = at(5, j);
                at(4, j) = old_dp; at(5, j) = old_l;
                if (j > 1 && !first_col && nb == c_prev && nb_prev == c) {
                    uint32_t t = t_dp + C.tc;
                    if (t <= v) { v = t; vl = t_l + 2; }
                }
            }
            at(0, j) = v; at(1, j) = vl;
            diag_dp = old_dp; diag_l = old_l;
            up_dp = v; up_l = vl;
            nb_prev = nb;
        }
        c_prev = c;
        if (up_dp <= C.k && i >= emit_begin) emit(i + 1, up_l, up_dp);
    }
}

}  // namespace ta
