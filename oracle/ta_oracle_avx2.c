/* ta_oracle_avx2.c -- TEST / MEASUREMENT INFRASTRUCTURE ONLY (see ta_oracle.h): a hand-written AVX2 restatement of the
 * banded anti-diagonal core with SATURATING u8 CELLS -- the shape and the cell width the reference itself runs for the
 * BASELINE configurations on an AVX2 host.
 *
 * The reference's SIMD core (levenshtein_simd_core_*, src/levenshtein.rs:829-1195) keeps one anti-diagonal of the band
 * per `jewel` vector and picks the narrowest cell type that holds max_k (dispatcher, :766-791): Avx{1,2,4,8}x32x8 =
 * 32/64/128/256 saturating u8 lanes (`_mm256_adds_epu8`, `_mm256_min_epu8`, `_mm256_cmpeq_epi8`; src/jewel.rs:227-233 and
 * the Avx*x32x8 impls) while unit_k <= lanes - 2 and max_k <= 254, then u16, then u32.  Its Rust sources cannot be built
 * here (no toolchain), so this file restates that algorithm in C intrinsics as bench.py's cpu_baseline ("port", never
 * "reference"): cfg2 (k = 32) runs 64 u8 lanes = 2 ymm per anti-diagonal, cfg4 (k = 8) 32 lanes = 1 ymm -- exactly the
 * reference's Avx2x32x8 / Avx1x32x8 choices.  Wider bands than 256 u8 lanes are left to the 16-bit compiler-vectorised
 * restatement (ta_oracle_simd.c) and the scalar one (ta_oracle.c): the ladder 8 -> 16 -> 32 is the caller's
 * (tao_levenshtein_k_batch_ladder below).
 *
 * Layout (mine, not the reference's k1/k2 window queues): diagonal index p = (j - i) + u in [0, 2u]; the newest cell of
 * every diagonal lives in one of two register arrays by the parity of p (E: p = 2e, O: p = 2e + 1).  Step s = i + j
 * updates the array of parity (s + u) & 1 in place: element e needs the elements e-1 / e (or e / e+1) of the other array
 * -- one byte-shift across the ymm registers per step, as in the reference (shift_right_1 / shift_left_1) -- and the
 * characters a[i-1], b[j-1], which are contiguous in e once `a` is reversed: two unaligned loads per 32 cells replace the
 * reference's shifting character windows.  Cells outside the matrix need no masks: rows/columns < 0 start at the
 * saturated maximum and stay there (255 + x = 255), dp(0, j) = j * gc falls out of the recurrence, and rows > n /
 * columns > m never feed a cell inside.  Linear gaps only (start_gap_cost == 0, the BASELINE configs); the transposition
 * term takes the scalar path's value (min with dp(i-2,j-2) + tc under the two character tests, :517-525).
 * Results are identical to tao_levenshtein_k_batch (tests/test_oracle_antidiag.py). */
#include <immintrin.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ta_oracle.h"

#define TAO_AVX2 __attribute__((target("avx2")))
#define TAO_PAD 320            /* bytes of padding either side of the staged strings (>= 256 lanes + 2) */

typedef struct {
    uint8_t *ar, *bb;          /* `a` reversed and `b`, each with TAO_PAD bytes of padding either side */
    size_t cap_a, cap_b;
} avx_ws;

static int avx_reserve(avx_ws *w, size_t n, size_t m) {
    if (n + 2 * TAO_PAD > w->cap_a) {
        free(w->ar);
        w->cap_a = n + 2 * TAO_PAD + 256;
        w->ar = malloc(w->cap_a);
        if (!w->ar) return 0;
    }
    if (m + 2 * TAO_PAD > w->cap_b) {
        free(w->bb);
        w->cap_b = m + 2 * TAO_PAD + 256;
        w->bb = malloc(w->cap_b);
        if (!w->bb) return 0;
    }
    return 1;
}

/* v shifted one byte lane up (lane i <- lane i-1), lane 0 <- the top byte of `below` */
TAO_AVX2 static inline __m256i shift_up_1(__m256i v, __m256i below) {
    const __m256i t = _mm256_permute2x128_si256(below, v, 0x21);        /* [below.hi, v.lo] */
    return _mm256_alignr_epi8(v, t, 15);
}
/* v shifted one byte lane down (lane i <- lane i+1), lane 31 <- the bottom byte of `above` */
TAO_AVX2 static inline __m256i shift_down_1(__m256i v, __m256i above) {
    const __m256i t = _mm256_permute2x128_si256(v, above, 0x21);        /* [v.hi, above.lo] */
    return _mm256_alignr_epi8(t, v, 1);
}

/* NV ymm registers (32 NV u8 lanes) per parity array; NV is a literal at every call site (always_inline) */
TAO_AVX2 static inline __attribute__((always_inline)) uint32_t
lev_avx2_u8(avx_ws *w, const uint8_t *a, size_t n, const uint8_t *b, size_t m, uint32_t k, uint32_t u, uint32_t mc,
            uint32_t gc, int has_t, uint32_t tc, const int NV) {
    /* stage: ar[x] = a[n-1-x], padding = bytes that can only meet cells outside the matrix */
    uint8_t *ar = w->ar + TAO_PAD, *bb = w->bb + TAO_PAD;
    memset(w->ar, 0, TAO_PAD);
    for (size_t x = 0; x < n; x++) ar[x] = a[n - 1 - x];
    memset(ar + n, 0, TAO_PAD);
    memset(w->bb, 0, TAO_PAD);
    memcpy(bb, b, m);
    memset(bb + m, 0, TAO_PAD);

    const __m256i inf = _mm256_set1_epi8((char)0xFF);
    const __m256i vmc = _mm256_set1_epi8((char)mc), vgc = _mm256_set1_epi8((char)gc), vtc = _mm256_set1_epi8((char)tc);
    __m256i V[2][8], P1[2][8];                       /* newest cell per diagonal; the cell before it on the diagonal */
    for (int t = 0; t < 2; t++)
        for (int q = 0; q < NV; q++) { V[t][q] = inf; P1[t][q] = inf; }
    {   /* dp(0,0) = 0 on p = u */
        uint8_t tmp[32 * 8];
        memset(tmp, 0xFF, sizeof(tmp));
        tmp[u >> 1] = 0;
        for (int q = 0; q < NV; q++) V[u & 1][q] = _mm256_loadu_si256((const __m256i *)(tmp + 32 * q));
    }
    const long S = (long)(n + m), lu = (long)u;
    for (long s = 1; s <= S; s++) {
        const int par = (int)((s + lu) & 1);
        /* element 0 of the updated array: d0 = par - u, i0 = (s - d0) / 2, j0 = (s + d0) / 2; element e: i0 - e, j0 + e */
        const long d0 = (long)par - lu, i0 = (s - d0) / 2, j0 = (s + d0) / 2;
        const uint8_t *arp = ar + ((long)n - i0), *bp = bb + (j0 - 1);
#pragma GCC unroll 8
        for (int q = 0; q < NV; q++) {
            const __m256i ca = _mm256_loadu_si256((const __m256i *)(arp + 32 * q));
            const __m256i cb = _mm256_loadu_si256((const __m256i *)(bp + 32 * q));
            __m256i lo, hi;
            if (par == 0) {                          /* p = 2e: p-1 = 2(e-1)+1, p+1 = 2e+1 */
                lo = shift_up_1(V[1][q], q ? V[1][q - 1] : inf);
                hi = V[1][q];
            } else {                                 /* p = 2e+1: p-1 = 2e, p+1 = 2e+2 */
                lo = V[0][q];
                hi = shift_down_1(V[0][q], q + 1 < NV ? V[0][q + 1] : inf);
            }
            const __m256i old = V[par][q];
            const __m256i eq = _mm256_cmpeq_epi8(ca, cb);
            const __m256i sub = _mm256_adds_epu8(old, _mm256_andnot_si256(eq, vmc));        /* :471-475 */
            const __m256i gap = _mm256_adds_epu8(_mm256_min_epu8(lo, hi), vgc);             /* :476-491 */
            __m256i r = _mm256_min_epu8(sub, gap);
            if (has_t) {
                /* a[i-1] == b[j-2] && a[i-2] == b[j-1] (:517-521): ar[e] is a[i-1], ar[e+1] is a[i-2]; b[e-1] is b[j-2] */
                const __m256i cb1 = _mm256_loadu_si256((const __m256i *)(bp + 32 * q - 1));
                const __m256i ca1 = _mm256_loadu_si256((const __m256i *)(arp + 32 * q + 1));
                const __m256i ok = _mm256_and_si256(_mm256_cmpeq_epi8(ca, cb1), _mm256_cmpeq_epi8(ca1, cb));
                const __m256i t = _mm256_or_si256(_mm256_adds_epu8(P1[par][q], vtc), _mm256_xor_si256(ok, inf));
                r = _mm256_min_epu8(r, t);
                P1[par][q] = old;
            }
            V[par][q] = r;
        }
    }
    /* padding bytes are all 0 on both sides: a cell outside the matrix may see a spurious match, which is harmless --
       it can only lower cells that never feed the matrix (rows > n, columns > m) or that sit on saturated rows < 0 */
    const long pa = (long)(m - n) + lu;
    uint8_t out[32 * 8];
    for (int q = 0; q < NV; q++) _mm256_storeu_si256((__m256i *)(out + 32 * q), V[pa & 1][q]);
    const uint32_t d = out[pa >> 1];
    return (d <= k && d < 255u) ? d : TAO_NONE;                                             /* :539-541 */
}

TAO_AVX2 static uint32_t lev_u8_1(avx_ws *w, const uint8_t *a, size_t n, const uint8_t *b, size_t m, uint32_t k, uint32_t u,
                                  uint32_t mc, uint32_t gc, int ht, uint32_t tc) {
    return ht ? lev_avx2_u8(w, a, n, b, m, k, u, mc, gc, 1, tc, 1) : lev_avx2_u8(w, a, n, b, m, k, u, mc, gc, 0, 0, 1);
}
TAO_AVX2 static uint32_t lev_u8_2(avx_ws *w, const uint8_t *a, size_t n, const uint8_t *b, size_t m, uint32_t k, uint32_t u,
                                  uint32_t mc, uint32_t gc, int ht, uint32_t tc) {
    return ht ? lev_avx2_u8(w, a, n, b, m, k, u, mc, gc, 1, tc, 2) : lev_avx2_u8(w, a, n, b, m, k, u, mc, gc, 0, 0, 2);
}
TAO_AVX2 static uint32_t lev_u8_4(avx_ws *w, const uint8_t *a, size_t n, const uint8_t *b, size_t m, uint32_t k, uint32_t u,
                                  uint32_t mc, uint32_t gc, int ht, uint32_t tc) {
    return ht ? lev_avx2_u8(w, a, n, b, m, k, u, mc, gc, 1, tc, 4) : lev_avx2_u8(w, a, n, b, m, k, u, mc, gc, 0, 0, 4);
}
TAO_AVX2 static uint32_t lev_u8_8(avx_ws *w, const uint8_t *a, size_t n, const uint8_t *b, size_t m, uint32_t k, uint32_t u,
                                  uint32_t mc, uint32_t gc, int ht, uint32_t tc) {
    return ht ? lev_avx2_u8(w, a, n, b, m, k, u, mc, gc, 1, tc, 8) : lev_avx2_u8(w, a, n, b, m, k, u, mc, gc, 0, 0, 8);
}

/* One pair through the u8 rungs of the ladder.  Returns 1 and the result when the pair is in the u8 class
 * (max_k <= 254, unit_k <= 254: :766-786 with the lane counts 32/64/128/256), 0 when it needs wider cells. */
static int lev_u8_one(avx_ws *w, const uint8_t *a, size_t n, const uint8_t *b, size_t m, uint32_t k, const tao_costs *c,
                      uint32_t *res, uint32_t *lanes_out) {
    if (n > m) { const uint8_t *t = a; a = b; b = t; size_t z = n; n = m; m = z; }          /* :386-390 */
    const uint32_t mc = c->mismatch_cost, gc = c->gap_cost;
    /* dispatcher clamp (:731-763), start_gap_cost == 0 */
    const uint64_t sub_all = (uint64_t)n * mc, gaps_all = 2ull * n * gc;
    uint64_t bound = (sub_all < gaps_all ? sub_all : gaps_all) + (uint64_t)(m - n) * gc;
    const uint32_t max_k = k < bound ? k : (uint32_t)bound;
    uint64_t u64 = max_k / gc;
    if (u64 > m) u64 = m;
    const uint32_t u = (uint32_t)u64;
    if (max_k > 254u || u > 254u) return 0;
    if (n == 0) { const uint64_t d = (uint64_t)m * gc; *res = d <= k ? (uint32_t)d : TAO_NONE; *lanes_out = 0; return 1; }
    if ((uint64_t)(m - n) > u) { *res = TAO_NONE; *lanes_out = 0; return 1; }               /* :426-428, :860-862 */
    if (!avx_reserve(w, n, m)) return 0;
    const int ht = c->has_transpose != 0;
    const uint32_t tc = c->transpose_cost;
    /* the reference's rungs: unit_k <= lanes - 2 (static_upper_bound() - 2) */
    uint32_t r;
    if (u <= 30u) { r = lev_u8_1(w, a, n, b, m, max_k, u, mc, gc, ht, tc); *lanes_out = 32; }
    else if (u <= 62u) { r = lev_u8_2(w, a, n, b, m, max_k, u, mc, gc, ht, tc); *lanes_out = 64; }
    else if (u <= 126u) { r = lev_u8_4(w, a, n, b, m, max_k, u, mc, gc, ht, tc); *lanes_out = 128; }
    else { r = lev_u8_8(w, a, n, b, m, max_k, u, mc, gc, ht, tc); *lanes_out = 256; }
    *res = r;
    return 1;
}

int tao_have_avx2(void) { return __builtin_cpu_supports("avx2") ? 1 : 0; }

/* Batch driver over the width ladder 8 -> 16 -> 32: u8 AVX2 (this file) while the pair is in the reference's u8 class, the
 * 16-bit anti-diagonal restatement (ta_oracle_simd.c) next, the scalar restatement (ta_oracle.c, u32) last.  Same contract
 * as tao_levenshtein_k_batch for start_gap_cost == 0; returns 0, or -1 when the host has no AVX2 or the costs are affine.
 * `lanes_hist` (may be NULL): 6 counters -- pairs answered without a DP, by 32 / 64 / 128 / 256 u8 lanes, by wider cells. */
int tao_levenshtein_k_batch_ladder(const uint8_t *a_blob, const uint64_t *a_off, const uint8_t *b_blob, const uint64_t *b_off,
                                   size_t n, uint32_t k, const tao_costs *costs, uint32_t *out, int threads, uint64_t *lanes_hist) {
    if (costs->start_gap_cost != 0 || !tao_have_avx2()) return -1;
    uint64_t hist[6] = {0, 0, 0, 0, 0, 0};
#ifdef _OPENMP
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
#endif
    {
        avx_ws w;
        memset(&w, 0, sizeof(w));
        uint64_t h[6] = {0, 0, 0, 0, 0, 0};
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 256)
#endif
        for (long long i = 0; i < (long long)n; i++) {
            const uint8_t *a = a_blob + a_off[i], *b = b_blob + b_off[i];
            const size_t al = (size_t)(a_off[i + 1] - a_off[i]), bl = (size_t)(b_off[i + 1] - b_off[i]);
            uint32_t r = 0, lanes = 0;
            if (lev_u8_one(&w, a, al, b, bl, k, costs, &r, &lanes)) {
                out[i] = r;
                h[lanes == 0 ? 0 : lanes == 32 ? 1 : lanes == 64 ? 2 : lanes == 128 ? 3 : 4]++;
            } else {
                uint64_t o2[2] = {0, al}, o3[2] = {0, bl};
                if (tao_levenshtein_k_batch_antidiag(a, o2, b, o3, 1, k, costs, &r, 1) != 0)
                    tao_levenshtein_k_batch(a, o2, b, o3, 1, k, costs, &r, 1);
                out[i] = r;
                h[5]++;
            }
        }
        free(w.ar); free(w.bb);
#ifdef _OPENMP
#pragma omp critical
#endif
        for (int t = 0; t < 6; t++) hist[t] += h[t];
    }
    if (lanes_hist) for (int t = 0; t < 6; t++) lanes_hist[t] = hist[t];
    return 0;
}
