/* ta_oracle_simd.c -- TEST / MEASUREMENT INFRASTRUCTURE ONLY (see ta_oracle.h): an anti-diagonal restatement of the
 * banded Levenshtein / restricted-Damerau core, the shape of the reference's SIMD path.
 *
 * The reference's SIMD core (levenshtein_simd_core_*, src/levenshtein.rs:829-1195) walks the anti-diagonals of the band
 * [-unit_k, unit_k] with one jewel vector per anti-diagonal (u8 / u16 / u32 lanes, AVX2 or SSE4.1).  Its Rust sources
 * cannot be built here (no toolchain), so bench.py's cpu_baseline would otherwise only have the restated SCALAR path.
 * This file restates the anti-diagonal algorithm in plain C with 16-bit cells, laid out so that gcc vectorises the
 * inner loops (contiguous arrays, `a` reversed once per pair; target_clones: AVX2 when the host has it, baseline x86-64
 * otherwise) -- a "port", not the reference.  Linear gaps only (start_gap_cost == 0: the BASELINE configs), results
 * identical to tao_levenshtein_k_batch (tests/test_oracle_antidiag.py, incl. the reference KATs for the k-bounded entries).
 *
 * Layout: diagonal index p = (j - i) + u in [0, 2u]; the newest cell of every diagonal lives in one of two arrays by the
 * parity of p (E: even p, O: odd p), so that a step s = i + j -- which updates the diagonals with p = s + u (mod 2) --
 * is one contiguous loop: element e of the updated array sits between elements e-1/e (or e/e+1) of the other one,
 * walks `a` backwards and `b` forwards. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ta_oracle.h"

#define SINF ((uint16_t)0x3FFF)

typedef struct {
    uint16_t *v[2], *p1[2];           /* newest cell per diagonal; the cell before it on the diagonal */
    uint8_t *ar, *bb, *tok;           /* `a` reversed (8 bytes of slack), `b` with one byte of left padding, transposition gate */
    size_t cap_band, cap_a, cap_b;
} simd_ws;

static int ws_reserve(simd_ws *w, size_t half, size_t n, size_t m) {
    if (half + 8 > w->cap_band) {
        for (int t = 0; t < 2; t++) {
            free(w->v[t]); free(w->p1[t]);
            w->v[t] = malloc((half + 8) * 2); w->p1[t] = malloc((half + 8) * 2);
            if (!w->v[t] || !w->p1[t]) return 0;
        }
        free(w->tok);
        w->tok = malloc(half + 8);
        if (!w->tok) return 0;
        w->cap_band = half + 8;
    }
    if (n + 16 > w->cap_a) {
        free(w->ar);
        w->ar = calloc(n + 16, 1);
        if (!w->ar) return 0;
        w->cap_a = n + 16;
    }
    if (m + 16 > w->cap_b) {
        free(w->bb);
        w->bb = calloc(m + 16, 1);
        if (!w->bb) return 0;
        w->cap_b = m + 16;
    }
    return 1;
}

/* one step: cur[e] for e in [e0, e1), neighbours lo[e] (diagonal p-1) and hi[e] (diagonal p+1) already offset by the caller */
__attribute__((target_clones("avx2", "default")))
static void step_plain(uint16_t *restrict cur, const uint16_t *restrict lo, const uint16_t *restrict hi,
                       const uint8_t *restrict ar, const uint8_t *restrict b, long cnt, uint16_t mc, uint16_t gc) {
    for (long e = 0; e < cnt; e++) {
        uint16_t sub = (uint16_t)(cur[e] + (ar[e] != b[e] ? mc : 0));       /* :471-475 */
        uint16_t ga = (uint16_t)(lo[e] + gc), gb = (uint16_t)(hi[e] + gc);  /* :476-491 */
        uint16_t m = sub < ga ? sub : ga;
        m = m < gb ? m : gb;
        cur[e] = m < SINF ? m : SINF;
    }
}

__attribute__((target_clones("avx2", "default")))
static void step_trans(uint16_t *restrict cur, uint16_t *restrict p1, const uint16_t *restrict lo,
                       const uint16_t *restrict hi, const uint8_t *restrict ar, const uint8_t *restrict b,
                       const uint8_t *restrict tok, long cnt, uint16_t mc, uint16_t gc, uint16_t tc) {
    for (long e = 0; e < cnt; e++) {
        uint16_t old = cur[e];
        uint16_t sub = (uint16_t)(old + (ar[e] != b[e] ? mc : 0));
        uint16_t ga = (uint16_t)(lo[e] + gc), gb = (uint16_t)(hi[e] + gc);
        uint16_t m = sub < ga ? sub : ga;
        m = m < gb ? m : gb;
        /* a[i-1] == b[j-2] && a[i-2] == b[j-1] (:517-521): ar[e] is a[i-1], ar[e+1] is a[i-2]; b[e] is b[j-1], b[e-1] is b[j-2] */
        uint16_t t = (uint16_t)(p1[e] + tc);                                /* dp(i-2,j-2) + tc */
        uint16_t ok = (uint16_t)((ar[e] == b[e - 1]) & (ar[e + 1] == b[e]) & tok[e]);
        m = (ok && t < m) ? t : m;
        p1[e] = old;
        cur[e] = m < SINF ? m : SINF;
    }
}

static uint32_t lev_antidiag_one(simd_ws *w, const uint8_t *a, size_t n, const uint8_t *b, size_t m, uint32_t k,
                                 const tao_costs *c) {
    if (n > m) { const uint8_t *t = a; a = b; b = t; size_t z = n; n = m; m = z; }       /* :386-390 */
    const uint32_t mc = c->mismatch_cost, gc = c->gap_cost;
    if (n == 0) { uint64_t d = (uint64_t)m * gc; return d <= k ? (uint32_t)d : TAO_NONE; }
    uint64_t u64 = k / gc;
    if (u64 > m) u64 = m;
    const long u = (long)u64;
    if ((long)(m - n) > u) return TAO_NONE;                                                 /* :426-428 */
    const long half = u + 2;
    if (!ws_reserve(w, (size_t)half, n, m)) return TAO_NONE;
    for (size_t x = 0; x < n; x++) w->ar[x] = a[n - 1 - x];
    memset(w->ar + n, 0, 8);
    memcpy(w->bb + 1, b, m);
    b = w->bb + 1;                                                                         /* b[-1] is readable now */
    for (int t = 0; t < 2; t++)
        for (long e = 0; e < half + 4; e++) { w->v[t][e] = SINF; w->p1[t][e] = SINF; }
    /* arrays are used with a +2 offset so that e-1 / e+1 stay inside */
    uint16_t *V[2] = {w->v[0] + 2, w->v[1] + 2}, *P1[2] = {w->p1[0] + 2, w->p1[1] + 2};
    const int has_t = c->has_transpose != 0;
    const uint16_t tc = (uint16_t)c->transpose_cost;
    uint8_t *tok = w->tok;
    V[u & 1][u >> 1] = 0;                                                                   /* dp(0,0) on p = u */
    const long S = (long)(n + m);
    for (long s = 1; s <= S; s++) {
        const int par = (int)((s + u) & 1);          /* parity of the diagonals p updated in this step */
        /* valid cells: 1 <= i <= n, 1 <= j <= m, |d| <= u with d = p - u, i = (s-d)/2, j = (s+d)/2 */
        long dlo = -u, dhi = u;
        if (s - 2 * (long)n > dlo) dlo = s - 2 * (long)n;
        if (2 - s > dlo) dlo = 2 - s;
        if (s - 2 < dhi) dhi = s - 2;
        if (2 * (long)m - s < dhi) dhi = 2 * (long)m - s;
        if (((dlo + s) & 1) != 0) dlo++;
        if (((dhi + s) & 1) != 0) dhi--;
        /* boundary cells of this step: (0, s) on d = s and (s, 0) on d = -s */
        if (s <= u) {
            const long pr = u + s, pl = u - s;
            if (s <= (long)m) { P1[pr & 1][pr >> 1] = V[pr & 1][pr >> 1]; V[pr & 1][pr >> 1] = (uint16_t)((uint64_t)s * gc < SINF ? s * gc : SINF); }
            if (s <= (long)n) { P1[pl & 1][pl >> 1] = V[pl & 1][pl >> 1]; V[pl & 1][pl >> 1] = (uint16_t)((uint64_t)s * gc < SINF ? s * gc : SINF); }
        }
        if (dlo > dhi) continue;
        const long p0 = dlo + u, e0 = p0 >> 1, cnt = ((dhi - dlo) >> 1) + 1;
        const long i0 = (s - dlo) / 2, j0 = (s + dlo) / 2;                 /* cell of element e0; then i-- and j++ per element */
        uint16_t *cur = V[par] + e0;
        const uint16_t *lo, *hi;
        if (par == 0) { lo = V[1] + e0 - 1; hi = V[1] + e0; }              /* p = 2e: p-1 = 2(e-1)+1, p+1 = 2e+1 */
        else { lo = V[0] + e0; hi = V[0] + e0 + 1; }                       /* p = 2e+1: p-1 = 2e, p+1 = 2e+2 */
        const uint8_t *arp = w->ar + (n - (size_t)i0);                      /* a[i-1] = ar[n-i] */
        const uint8_t *bp = b + (j0 - 1);
        if (!has_t) step_plain(cur, lo, hi, arp, bp, cnt, (uint16_t)mc, (uint16_t)gc);
        else {
            for (long e = 0; e < cnt; e++) tok[e] = (uint8_t)((i0 - e >= 2) && (j0 + e >= 2));
            /* ar[e+1] = a[i-2] needs i >= 2 (else reads ar[n-i+1] with i = 1: index n, inside the +8 slack but gated by tok);
               b[e-1] = b[j-2] needs j >= 2 (gated too; bp - 1 >= b - 1 is only dereferenced when j >= 2) */
            step_trans(cur, P1[par] + e0, lo, hi, arp, bp, tok, cnt, (uint16_t)mc, (uint16_t)gc, tc);
        }
    }
    const long pa = (long)(m - n) + u;
    const uint32_t d = V[pa & 1][pa >> 1];
    return (d <= k && d < SINF) ? d : TAO_NONE;                                             /* :539-541 */
}

/* batch driver: same contract as tao_levenshtein_k_batch for start_gap_cost == 0 and k < 0x3F00 (16-bit cells); returns 0,
 * or -1 when the costs / k are outside what this restatement covers (the caller then uses the scalar one) */
int tao_levenshtein_k_batch_antidiag(const uint8_t *a_blob, const uint64_t *a_off, const uint8_t *b_blob, const uint64_t *b_off,
                                     size_t n, uint32_t k, const tao_costs *costs, uint32_t *out, int threads) {
    if (costs->start_gap_cost != 0 || k >= 0x3F00u - 512u) return -1;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
#endif
    {
        simd_ws w;
        memset(&w, 0, sizeof(w));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 256)
#endif
        for (long long i = 0; i < (long long)n; i++)
            out[i] = lev_antidiag_one(&w, a_blob + a_off[i], (size_t)(a_off[i + 1] - a_off[i]), b_blob + b_off[i],
                                      (size_t)(b_off[i + 1] - b_off[i]), k, costs);
        for (int t = 0; t < 2; t++) { free(w.v[t]); free(w.p1[t]); }
        free(w.ar); free(w.bb); free(w.tok);
    }
    return 0;
}
