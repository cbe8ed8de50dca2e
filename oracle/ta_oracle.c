/*
 * ta_oracle.c -- CPU oracle: plain-C restatement of triple_accel's scalar paths.
 *
 * TEST INFRASTRUCTURE ONLY (see ta_oracle.h).  Never linked into the product.
 * Each function cites the reference file:line it follows (relative to
 * /root/reference).  Arithmetic widths follow the Rust text: costs are u32,
 * gap-extension adds saturate, everything else is a plain (wrapping) u32 add.
 */
#include "ta_oracle.h"

#include <stdlib.h>
#include <string.h>

#define U32_MAX 0xFFFFFFFFu

static inline uint32_t sat_add(uint32_t x, uint32_t y) {
    uint32_t s = x + y;
    return s < x ? U32_MAX : s;
}
static inline uint32_t sat_sub(uint32_t x, uint32_t y) { return x > y ? x - y : 0u; }
static inline uint32_t min_u32(uint32_t x, uint32_t y) { return x < y ? x : y; }
static inline size_t min_sz(size_t x, size_t y) { return x < y ? x : y; }
static inline size_t max_sz(size_t x, size_t y) { return x > y ? x : y; }

void tao_free(void *p) { free(p); }

/* ---------------------------------------------------------------- costs */

/* src/levenshtein.rs:44-52 */
int tao_costs_valid(const tao_costs *c) {
    if (!(c->mismatch_cost > 0)) return 0;
    if (!(c->gap_cost > 0)) return 0;
    if (c->has_transpose) {
        if (!(c->transpose_cost > 0)) return 0;
        if (!((c->transpose_cost >> 1) < c->mismatch_cost)) return 0;
        if (!((c->transpose_cost >> 1) < c->gap_cost)) return 0;
    }
    return 1;
}

/* src/levenshtein.rs:67-71 (u8 + u8 in Rust would overflow-panic in debug; widen here) */
int tao_costs_valid_search(const tao_costs *c) {
    if (c->has_transpose) {
        if (!((uint32_t)c->transpose_cost <= (uint32_t)c->start_gap_cost + (uint32_t)c->gap_cost)) return 0;
    }
    return 1;
}

/* ---------------------------------------------------------------- growable vectors */

typedef struct { tao_match *p; size_t n, cap; } match_vec;
static void mv_push(match_vec *v, uint64_t start, uint64_t end, uint32_t k) {
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 16;
        v->p = (tao_match *)realloc(v->p, v->cap * sizeof(tao_match));
    }
    v->p[v->n].start = start; v->p[v->n].end = end; v->p[v->n].k = k; v->p[v->n].pad_ = 0;
    v->n++;
}

typedef struct { tao_edit *p; size_t n, cap; } edit_vec;
/* run-length push: src/levenshtein.rs:304-308, 598-602 */
static void ev_push_rle(edit_vec *v, uint32_t e) {
    if (v->n > 0 && v->p[v->n - 1].edit == e) { v->p[v->n - 1].count++; return; }
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 16;
        v->p = (tao_edit *)realloc(v->p, v->cap * sizeof(tao_edit));
    }
    v->p[v->n].edit = e; v->p[v->n].pad_ = 0; v->p[v->n].count = 1;
    v->n++;
}
static void ev_reverse(edit_vec *v) {
    for (size_t i = 0, j = v->n; i + 1 < j; i++) { j--; tao_edit t = v->p[i]; v->p[i] = v->p[j]; v->p[j] = t; }
}
static void ev_give(edit_vec *v, tao_edit **edits, size_t *n_edits) {
    if (edits) { *edits = v->p; } else { free(v->p); }
    if (n_edits) *n_edits = v->n;
}

/* ---------------------------------------------------------------- hamming */

/* src/hamming.rs:36-47 */
uint32_t tao_hamming_naive(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len) {
    if (a_len != b_len) return TAO_NONE;   /* assert!(len == b.len()) :38 */
    uint32_t res = 0;
    for (size_t i = 0; i < a_len; i++) res += (a[i] != b[i]);
    return res;
}

/* src/hamming.rs:96-146 */
size_t tao_hamming_search_naive_with_opts(const uint8_t *needle, size_t needle_len,
                                          const uint8_t *haystack, size_t haystack_len,
                                          uint32_t k, int search_type, tao_match **out) {
    match_vec mv = {0, 0, 0};
    *out = NULL;
    if (needle_len > haystack_len) return 0;          /* :100-102 */
    size_t len = haystack_len + 1 - needle_len;        /* :104 */
    uint32_t curr_k = k;
    for (size_t i = 0; i < len; i++) {                 /* :109-130 */
        uint32_t final_res = 0;
        int skip = 0;
        for (size_t j = 0; j < needle_len; j++) {
            final_res += (needle[j] != haystack[i + j]);
            if (final_res > curr_k) { skip = 1; break; }   /* early stop :116-119 */
        }
        if (skip) continue;
        if (search_type == TAO_SEARCH_BEST) curr_k = final_res;   /* :122-125 */
        mv_push(&mv, i, i + needle_len, final_res);
    }
    if (search_type == TAO_SEARCH_BEST) {              /* :135-143 filter m.k == curr_k */
        size_t w = 0;
        for (size_t r = 0; r < mv.n; r++) if (mv.p[r].k == curr_k) mv.p[w++] = mv.p[r];
        mv.n = w;
    }
    *out = mv.p;
    return mv.n;
}

/* src/hamming.rs:454-475 + src/lib.rs:237-243 */
int tao_hamming_search_simd_with_opts(const uint8_t *needle, size_t needle_len,
                                      const uint8_t *haystack, size_t haystack_len,
                                      uint32_t k, int search_type, tao_match **out, size_t *n_out) {
    *out = NULL; *n_out = 0;
    if (needle_len > haystack_len) return 0;   /* :455-457 */
    if (needle_len == 0) return 0;             /* :459-461 */
    for (size_t i = 0; i < haystack_len; i++)  /* check_no_null_bytes :463 */
        if (haystack[i] == 0) return 1;
    /* the SIMD cores (:481-552) give the same matches as the scalar routine (:474) */
    *n_out = tao_hamming_search_naive_with_opts(needle, needle_len, haystack, haystack_len, k, search_type, out);
    return 0;
}

/* ---------------------------------------------------------------- full-matrix Levenshtein */

/* src/levenshtein.rs:148-319 */
uint32_t tao_levenshtein_naive_with_opts(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len,
                                         int trace_on, const tao_costs *costs,
                                         tao_edit **edits, size_t *n_edits) {
    int swap = a_len > b_len;                                  /* :157 */
    const uint8_t *an = swap ? b : a; size_t an_len = swap ? b_len : a_len;
    const uint8_t *bn = swap ? a : b; size_t bn_len = swap ? a_len : b_len;
    uint32_t mc = costs->mismatch_cost, gc = costs->gap_cost, sg = costs->start_gap_cost;
    uint32_t tc = costs->has_transpose ? costs->transpose_cost : 0;
    int allow_t = costs->has_transpose;

    size_t len = an_len + 1;                                   /* :171 */
    uint32_t *dp0 = (uint32_t *)calloc(len, 4), *dp1 = (uint32_t *)calloc(len, 4), *dp2 = (uint32_t *)calloc(len, 4);
    uint32_t *ag = (uint32_t *)malloc(len * 4), *bg = (uint32_t *)malloc(len * 4);
    for (size_t i = 0; i < len; i++) { ag[i] = U32_MAX; bg[i] = U32_MAX; }
    uint8_t *tb = trace_on ? (uint8_t *)calloc((bn_len + 1) * len, 1) : NULL;

    for (size_t i = 0; i < len; i++) {                          /* :183-189 */
        dp1[i] = (uint32_t)i * gc + (i == 0 ? 0 : sg);
        if (trace_on) tb[i] = 2;
    }
    for (size_t i = 1; i < bn_len + 1; i++) {                   /* :191 */
        ag[0] = (uint32_t)i * gc + sg;
        dp2[0] = (uint32_t)i * gc + sg;
        if (trace_on) tb[i * len] = 1;
        for (size_t j = 1; j < len; j++) {                       /* :199 */
            uint32_t sub = dp1[j - 1] + (uint32_t)(an[j - 1] != bn[i - 1]) * mc;
            ag[j] = min_u32(dp1[j] + sg + gc, sat_add(ag[j], gc));
            bg[j] = min_u32(dp2[j - 1] + sg + gc, sat_add(bg[j - 1], gc));
            size_t ti = i * len + j;
            dp2[j] = ag[j];
            if (trace_on) tb[ti] = 1;
            if (bg[j] < dp2[j]) { dp2[j] = bg[j]; if (trace_on) tb[ti] = 2; }      /* :217-223 */
            if (sub <= dp2[j]) { dp2[j] = sub; if (trace_on) tb[ti] = 0; }          /* :225-231 */
            if (allow_t && i > 1 && j > 1 && an[j - 1] == bn[i - 2] && an[j - 2] == bn[i - 1]) {  /* :233-247 */
                uint32_t t = dp0[j - 2] + tc;
                if (t <= dp2[j]) { dp2[j] = t; if (trace_on) tb[ti] = 3; }
            }
        }
        uint32_t *t0 = dp0; dp0 = dp1; dp1 = dp2; dp2 = t0;      /* :250-251: swap(dp0,dp1); swap(dp1,dp2) */
    }
    uint32_t result = dp1[an_len];

    if (trace_on) {                                              /* :254-314 */
        edit_vec ev = {0, 0, 0};
        size_t i = bn_len, j = an_len;
        while (i > 0 || j > 0) {
            uint8_t e = tb[i * len + j];
            uint32_t et;
            switch (e) {
            case 0: i--; j--; et = (an[j] == bn[i]) ? TAO_EDIT_MATCH : TAO_EDIT_MISMATCH; break;
            case 1: i--; et = swap ? TAO_EDIT_BGAP : TAO_EDIT_AGAP; break;
            case 2: j--; et = swap ? TAO_EDIT_AGAP : TAO_EDIT_BGAP; break;
            default: i -= 2; j -= 2; et = TAO_EDIT_TRANSPOSE; break;
            }
            ev_push_rle(&ev, et);
        }
        ev_reverse(&ev);
        ev_give(&ev, edits, n_edits);
    } else {
        if (edits) *edits = NULL;
        if (n_edits) *n_edits = 0;
    }
    free(dp0); free(dp1); free(dp2); free(ag); free(bg); free(tb);
    return result;
}

/* ---------------------------------------------------------------- banded Levenshtein */

/* the two-stage clamp shared by :399-421 and :731-757 */
static uint32_t clamp_max_k(uint32_t mn, uint32_t mx, uint32_t k, uint32_t mc, uint32_t gc, uint32_t sg) {
    uint32_t bound = min_u32(mn * mc, (mn << 1) * gc + (mn == 0 ? 0 : sg + (mx == mn ? sg : 0)));
    return min_u32(k, bound + (mx - mn) * gc + (mx == mn ? 0 : sg));
}

/* src/levenshtein.rs:376-607 */
uint32_t tao_levenshtein_naive_k_with_opts(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len,
                                           uint32_t k, int trace_on, const tao_costs *costs,
                                           tao_edit **edits, size_t *n_edits) {
    if (edits) *edits = NULL;
    if (n_edits) *n_edits = 0;
    int swap = a_len > b_len;                                   /* :386 */
    const uint8_t *an = swap ? b : a; size_t an_len = swap ? b_len : a_len;
    const uint8_t *bn = swap ? a : b; size_t bn_len = swap ? a_len : b_len;
    uint32_t mc = costs->mismatch_cost, gc = costs->gap_cost, sg = costs->start_gap_cost;
    uint32_t tc = costs->has_transpose ? costs->transpose_cost : 0;
    int allow_t = costs->has_transpose;

    uint32_t max_k = clamp_max_k((uint32_t)an_len, (uint32_t)bn_len, k, mc, gc, sg);   /* :399-421 */
    size_t unit_k = (size_t)(sat_sub(max_k, sg) / gc);                                   /* :424 */
    if (bn_len - an_len > unit_k) return TAO_NONE;                                       /* :426-428 */

    size_t len = an_len + 1;
    size_t lo = 0;
    size_t hi = min_sz(unit_k + 1, bn_len + 1);
    size_t prev_lo0, prev_lo1 = 0, prev_hi;
    size_t k_len = min_sz((unit_k << 1) + 1, bn_len + 1);                                /* :437 */
    uint32_t *dp0 = (uint32_t *)calloc(k_len, 4), *dp1 = (uint32_t *)calloc(k_len, 4), *dp2 = (uint32_t *)calloc(k_len, 4);
    uint32_t *ag = (uint32_t *)malloc(k_len * 4), *bg = (uint32_t *)malloc(k_len * 4);
    for (size_t i = 0; i < k_len; i++) { ag[i] = U32_MAX; bg[i] = U32_MAX; }
    uint8_t *tb = trace_on ? (uint8_t *)calloc(len * k_len, 1) : NULL;

    for (size_t i = 0; i < hi - lo; i++) {                       /* :450-456 */
        dp1[i] = (uint32_t)i * gc + (i == 0 ? 0 : sg);
        if (trace_on) tb[i] = 1;
    }

    for (size_t i = 1; i < len; i++) {                           /* :458 */
        prev_lo0 = prev_lo1;
        prev_lo1 = lo;
        prev_hi = hi;
        hi = min_sz(hi + 1, bn_len + 1);
        if (i > unit_k) lo += 1;

        for (size_t j = 0; j < hi - lo; j++) {                   /* :469 */
            size_t idx = lo + j;
            uint32_t sub = (idx == 0) ? U32_MAX
                         : dp1[idx - 1 - prev_lo1] + (uint32_t)(an[i - 1] != bn[idx - 1]) * mc;
            ag[j] = (j == 0) ? U32_MAX
                  : min_u32(dp2[j - 1] + sg + gc, sat_add(ag[j - 1], gc));
            bg[j] = (idx >= prev_hi) ? U32_MAX
                  : min_u32(dp1[idx - prev_lo1] + sg + gc, sat_add(bg[idx - prev_lo1], gc));

            dp2[j] = sub;
            size_t ti = i * k_len + j;
            if (trace_on) tb[ti] = 0;
            if (ag[j] < dp2[j]) { dp2[j] = ag[j]; if (trace_on) tb[ti] = 1; }          /* :501-507 */
            if (bg[j] < dp2[j]) { dp2[j] = bg[j]; if (trace_on) tb[ti] = 2; }          /* :509-515 */
            if (allow_t && i > 1 && idx > 1 && an[i - 1] == bn[idx - 2] && an[i - 2] == bn[idx - 1]) {  /* :517-532 */
                uint32_t t = dp0[idx - prev_lo0 - 2] + tc;
                if (t <= dp2[j]) { dp2[j] = t; if (trace_on) tb[ti] = 3; }
            }
        }
        uint32_t *t0 = dp0; dp0 = dp1; dp1 = dp2; dp2 = t0;      /* :535-536 */
    }

    uint32_t result = dp1[hi - lo - 1];
    if (result > max_k) {                                        /* :539-541 */
        free(dp0); free(dp1); free(dp2); free(ag); free(bg); free(tb);
        return TAO_NONE;
    }
    if (trace_on) {                                              /* :547-606 */
        edit_vec ev = {0, 0, 0};
        size_t i = an_len, j = bn_len;
        while (i > 0 || j > 0) {
            uint8_t e = tb[i * k_len + (j - (i > unit_k ? i - unit_k : 0))];
            uint32_t et;
            switch (e) {
            case 0: i--; j--; et = (an[i] == bn[j]) ? TAO_EDIT_MATCH : TAO_EDIT_MISMATCH; break;
            case 1: j--; et = swap ? TAO_EDIT_BGAP : TAO_EDIT_AGAP; break;
            case 2: i--; et = swap ? TAO_EDIT_AGAP : TAO_EDIT_BGAP; break;
            default: i -= 2; j -= 2; et = TAO_EDIT_TRANSPOSE; break;
            }
            ev_push_rle(&ev, et);
        }
        ev_reverse(&ev);
        ev_give(&ev, edits, n_edits);
    }
    free(dp0); free(dp1); free(dp2); free(ag); free(bg); free(tb);
    return result;
}

/* src/levenshtein.rs:714-827: result contract through the scalar fallback (:826) */
uint32_t tao_levenshtein_simd_k_with_opts(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len,
                                          uint32_t k, int trace_on, const tao_costs *costs,
                                          tao_edit **edits, size_t *n_edits) {
    if (a_len == 0 && b_len == 0) {                              /* :721-727 */
        if (edits) *edits = NULL;
        if (n_edits) *n_edits = 0;
        return 0;
    }
    return tao_levenshtein_naive_k_with_opts(a, a_len, b, b_len, k, trace_on, costs, edits, n_edits);
}

/* src/levenshtein.rs:731-791 (AVX2 ladder; static_upper_bound = 32/64/128/256, jewel.rs:127-134) */
void tao_levenshtein_select(size_t a_len, size_t b_len, uint32_t k, const tao_costs *costs,
                            uint32_t *max_k_out, uint32_t *unit_k_out, uint32_t *cell_bits, uint32_t *lanes) {
    uint32_t mn = (uint32_t)min_sz(a_len, b_len), mx = (uint32_t)max_sz(a_len, b_len);
    uint32_t mc = costs->mismatch_cost, gc = costs->gap_cost, sg = costs->start_gap_cost;
    uint32_t max_k = clamp_max_k(mn, mx, k, mc, gc, sg);                 /* :734-757 */
    uint32_t unit_k = min_u32(sat_sub(max_k, sg) / gc, mx);              /* :760-763 */
    *max_k_out = max_k; *unit_k_out = unit_k;
    static const uint32_t ub[4] = {32, 64, 128, 256};
    for (int t = 0; t < 4; t++) {                                        /* :767-786 */
        if (unit_k <= ub[t] - 2 && max_k <= 254u) { *cell_bits = 8; *lanes = ub[t]; return; }
    }
    if (max_k <= 65534u) { *cell_bits = 16; *lanes = 0; return; }        /* :787-788 */
    *cell_bits = 32; *lanes = 0;                                         /* :789-790 */
}

static const tao_costs LEV_COSTS = {1, 1, 0, 0, 0};      /* src/levenshtein.rs:76-81 */
static const tao_costs RDAM_COSTS = {1, 1, 0, 1, 1};     /* src/levenshtein.rs:84-89 */

/* src/levenshtein.rs:1397-1399 */
uint32_t tao_levenshtein(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len) {
    return tao_levenshtein_simd_k_with_opts(a, a_len, b, b_len, U32_MAX, 0, &LEV_COSTS, NULL, NULL);
}
/* src/levenshtein.rs:1419-1423 */
uint32_t tao_rdamerau(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len) {
    return tao_levenshtein_simd_k_with_opts(a, a_len, b, b_len, U32_MAX, 0, &RDAM_COSTS, NULL, NULL);
}
/* src/levenshtein.rs:1480-1494 (k = 30, 60, 120, ...) */
uint32_t tao_levenshtein_exp_with_opts(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len,
                                       int trace_on, const tao_costs *costs,
                                       tao_edit **edits, size_t *n_edits) {
    uint32_t k = 30;
    for (;;) {
        uint32_t r = tao_levenshtein_simd_k_with_opts(a, a_len, b, b_len, k, trace_on, costs, edits, n_edits);
        if (r != TAO_NONE) return r;
        k *= 2;
    }
}
/* src/levenshtein.rs:1445-1454 */
uint32_t tao_levenshtein_exp(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len) {
    return tao_levenshtein_exp_with_opts(a, a_len, b, b_len, 0, &LEV_COSTS, NULL, NULL);
}
/* src/levenshtein.rs:1516-1526 */
uint32_t tao_rdamerau_exp(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len) {
    return tao_levenshtein_exp_with_opts(a, a_len, b, b_len, 0, &RDAM_COSTS, NULL, NULL);
}

/* cells(n,m,u) = sum_{i=1..n} max(0, min(i+u,m) - max(1,i-u) + 1),
 * i.e. the cells the loop at src/levenshtein.rs:458-469 visits, column 0 excluded,
 * with unit_k as computed by the dispatcher (:760-763). */
uint64_t tao_band_cells(size_t a_len, size_t b_len, uint32_t k, const tao_costs *costs) {
    uint32_t max_k, unit_k, bits, lanes;
    tao_levenshtein_select(a_len, b_len, k, costs, &max_k, &unit_k, &bits, &lanes);
    uint64_t n = min_sz(a_len, b_len), m = max_sz(a_len, b_len), u = unit_k, cells = 0;
    for (uint64_t i = 1; i <= n; i++) {
        uint64_t hi = (i + u < m) ? i + u : m;
        uint64_t lo = (i > u + 1) ? i - u : 1;
        if (hi >= lo) cells += hi - lo + 1;
    }
    return cells;
}

/* ---------------------------------------------------------------- search */

/* src/levenshtein.rs:1873, src/hamming.rs:423 */
uint32_t tao_default_search_k(size_t needle_len) {
    return (uint32_t)(needle_len >> 1) + ((uint32_t)needle_len & 1u);
}

/* src/levenshtein.rs:1589-1838 */
int tao_levenshtein_search_naive_with_opts(const uint8_t *needle, size_t needle_len,
                                           const uint8_t *haystack, size_t haystack_len,
                                           uint32_t k, int search_type, const tao_costs *costs,
                                           int anchored, tao_match **out, size_t *n_out) {
    match_vec mv = {0, 0, 0};
    *out = NULL; *n_out = 0;
    uint32_t mc = costs->mismatch_cost, gc = costs->gap_cost, sg = costs->start_gap_cost;
    uint32_t tc = costs->has_transpose ? costs->transpose_cost : 0;
    int allow_t = costs->has_transpose;

    if (needle_len == 0) {                                       /* :1600-1644 */
        if (!anchored) return 0;
        mv_push(&mv, 0, 0, 0);
        if (search_type == TAO_SEARCH_ALL) {                     /* :1604-1634 */
            uint32_t cost = sg;
            for (size_t i = 0; i < haystack_len; ) {
                i += 1;
                cost += gc;
                if (cost <= k) mv_push(&mv, 0, i, cost); else break;   /* from_fn returns None -> iterator ends */
            }
        }
        *out = mv.p; *n_out = mv.n;
        return 0;
    }

    if (!tao_costs_valid_search(costs)) return 1;                /* :1647 */

    size_t len = needle_len + 1;
    size_t iter_len = haystack_len;                              /* :1650-1661 */
    if (anchored) {
        size_t extra = (size_t)sat_sub(k, sg) / (size_t)gc;
        size_t lim = needle_len + extra;
        if (lim < needle_len) lim = (size_t)-1;                  /* saturating_add */
        iter_len = min_sz(haystack_len, lim);
    }

    uint32_t *dp0 = (uint32_t *)calloc(len, 4), *dp1 = (uint32_t *)calloc(len, 4), *dp2 = (uint32_t *)calloc(len, 4);
    uint32_t *ng = (uint32_t *)malloc(len * 4), *hg = (uint32_t *)malloc(len * 4);
    for (size_t j = 0; j < len; j++) { ng[j] = U32_MAX; hg[j] = U32_MAX; }
    size_t *l0 = (size_t *)calloc(len, sizeof(size_t)), *l1 = (size_t *)calloc(len, sizeof(size_t)),
           *l2 = (size_t *)calloc(len, sizeof(size_t));
    size_t *ngl = (size_t *)calloc(len, sizeof(size_t)), *hgl = (size_t *)calloc(len, sizeof(size_t));
    uint32_t curr_k = k;

    /* first call of the closure: :1685-1707 */
    for (size_t j = 0; j < len; j++) dp1[j] = (uint32_t)j * gc + (j == 0 ? 0 : sg);
    if (dp1[len - 1] <= curr_k) {
        if (search_type == TAO_SEARCH_BEST) curr_k = dp1[len - 1];
        mv_push(&mv, 0, 0, dp1[len - 1]);
    }
    /* the Best fold (:1816-1832) consumes the stream in order; emulate it inline */
    /* we first collect the raw emitted stream, then fold. */
    size_t i = 0;
    while (i < iter_len) {                                       /* :1709 */
        uint32_t c0 = anchored ? ((uint32_t)i + 1) * gc + sg : 0;
        ng[0] = c0;
        dp2[0] = c0;
        ngl[0] = 0;
        l2[0] = 0;

        for (size_t j = 1; j < len; j++) {                       /* :1723 */
            uint32_t sub = dp1[j - 1] + (uint32_t)(needle[j - 1] != haystack[i]) * mc;

            uint32_t new_gap = dp1[j] + sg + gc;                 /* :1726-1737 */
            uint32_t cont_gap = sat_add(ng[j], gc);
            if (new_gap < cont_gap) { ng[j] = new_gap; ngl[j] = l1[j] + 1; }
            else if (new_gap > cont_gap) { ng[j] = cont_gap; ngl[j] += 1; }
            else { ng[j] = cont_gap; ngl[j] = max_sz(l1[j], ngl[j]) + 1; }

            new_gap = dp2[j - 1] + sg + gc;                      /* :1739-1750 */
            cont_gap = sat_add(hg[j - 1], gc);
            if (new_gap < cont_gap) { hg[j] = new_gap; hgl[j] = l2[j - 1]; }
            else if (new_gap > cont_gap) { hg[j] = cont_gap; hgl[j] = hgl[j - 1]; }
            else { hg[j] = cont_gap; hgl[j] = max_sz(l2[j - 1], hgl[j - 1]); }

            dp2[j] = ng[j];                                      /* :1752-1753 */
            l2[j] = ngl[j];

            if ((hg[j] < dp2[j]) || (hg[j] == dp2[j] && l2[j - 1] > l2[j])) {   /* :1755-1760 (Q2: reads length2[j-1]) */
                dp2[j] = hg[j];
                l2[j] = hgl[j];
            }
            if ((sub < dp2[j]) || (sub == dp2[j] && (l1[j - 1] + 1) > l2[j])) { /* :1762-1765 */
                dp2[j] = sub;
                l2[j] = l1[j - 1] + 1;
            }
            if (allow_t && i > 0 && j > 1 && needle[j - 1] == haystack[i - 1] && needle[j - 2] == haystack[i]) {  /* :1767-1779 */
                uint32_t t = dp0[j - 2] + tc;
                if (t <= dp2[j]) { dp2[j] = t; l2[j] = l0[j - 2] + 2; }
            }
        }

        uint32_t final_res = dp2[len - 1];
        size_t final_length = l2[len - 1];
        { uint32_t *t0 = dp0; dp0 = dp1; dp1 = dp2; dp2 = t0; }   /* :1785-1788 */
        { size_t *t0 = l0; l0 = l1; l1 = l2; l2 = t0; }
        i += 1;

        if (final_res <= curr_k) {                               /* :1792-1806 */
            if (search_type == TAO_SEARCH_BEST) curr_k = final_res;
            mv_push(&mv, (uint64_t)(i - final_length), (uint64_t)i, final_res);
        }
    }

    if (search_type == TAO_SEARCH_BEST) {                        /* :1812-1835 */
        match_vec rv = {0, 0, 0};
        for (size_t r = 0; r < mv.n; r++) {
            if (rv.n == 0) { mv_push(&rv, mv.p[r].start, mv.p[r].end, mv.p[r].k); }
            else {
                tao_match *last = &rv.p[rv.n - 1];
                if (mv.p[r].start <= last->start) { *last = mv.p[r]; }      /* replace if fully overlapping */
                else mv_push(&rv, mv.p[r].start, mv.p[r].end, mv.p[r].k);
            }
        }
        /* curr_k after the loop == the closure's final curr_k (each m.1 is the running value) */
        size_t w = 0;
        for (size_t r = 0; r < rv.n; r++) if (rv.p[r].k == curr_k) rv.p[w++] = rv.p[r];
        rv.n = w;
        free(mv.p);
        mv = rv;
    }

    free(dp0); free(dp1); free(dp2); free(ng); free(hg);
    free(l0); free(l1); free(l2); free(ngl); free(hgl);
    *out = mv.p; *n_out = mv.n;
    return 0;
}

/* ---------------------------------------------------------------- batch drivers (bench/test convenience)
 * Plain loops over the single-pair restatements above; `threads` > 1 splits the batch with OpenMP. */
#ifdef _OPENMP
#include <omp.h>
#endif

void tao_levenshtein_k_batch(const uint8_t *a_blob, const uint64_t *a_off, const uint8_t *b_blob, const uint64_t *b_off,
                             size_t n, uint32_t k, const tao_costs *costs, uint32_t *out, int threads) {
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads > 0 ? threads : 1)
#endif
    for (long long i = 0; i < (long long)n; i++) {
        out[i] = tao_levenshtein_simd_k_with_opts(a_blob + a_off[i], (size_t)(a_off[i + 1] - a_off[i]),
                                                  b_blob + b_off[i], (size_t)(b_off[i + 1] - b_off[i]),
                                                  k, 0, costs, NULL, NULL);
    }
}

void tao_levenshtein_exp_batch(const uint8_t *a_blob, const uint64_t *a_off, const uint8_t *b_blob, const uint64_t *b_off,
                               size_t n, const tao_costs *costs, uint32_t *out, int threads) {
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads > 0 ? threads : 1)
#endif
    for (long long i = 0; i < (long long)n; i++) {
        out[i] = tao_levenshtein_exp_with_opts(a_blob + a_off[i], (size_t)(a_off[i + 1] - a_off[i]),
                                               b_blob + b_off[i], (size_t)(b_off[i + 1] - b_off[i]),
                                               0, costs, NULL, NULL);
    }
}

void tao_hamming_batch(const uint8_t *a_blob, const uint64_t *a_off, const uint8_t *b_blob, const uint64_t *b_off,
                       size_t n, uint32_t *out, int threads) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (long long i = 0; i < (long long)n; i++) {
        out[i] = tao_hamming_naive(a_blob + a_off[i], (size_t)(a_off[i + 1] - a_off[i]),
                                   b_blob + b_off[i], (size_t)(b_off[i + 1] - b_off[i]));
    }
}

int tao_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
