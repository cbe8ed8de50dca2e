/*
 * ta_oracle.h -- CPU oracle for the triple_accel hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * *scalar* routines (the bit-exactness target named in SURVEY.md section 8).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  The product (triple_accel_amd/) never links, imports or calls it.
 *
 * Parity pin: the reference is Rust and cannot be built in this image (no
 * rustc/cargo), so the oracle is pinned against every known-answer test the
 * reference holds for this path (tests/basic_tests.rs + the doc-test
 * examples), extracted to tests/golden/kats.json by
 * tests/golden/extract_kats.py, plus the cross-implementation property checks
 * of benches/rand_benchmarks.rs (banded == full matrix, search variants agree).
 *
 * Every function cites the reference file:line it follows
 * (paths relative to /root/reference).
 */
#ifndef TA_ORACLE_H
#define TA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/levenshtein.rs:20-26 */
typedef struct {
    uint8_t mismatch_cost;
    uint8_t gap_cost;
    uint8_t start_gap_cost;
    uint8_t has_transpose;   /* Option<u8>::is_some() */
    uint8_t transpose_cost;  /* value when has_transpose */
} tao_costs;

/* src/lib.rs:135-142 */
typedef struct {
    uint64_t start;
    uint64_t end;
    uint32_t k;
    uint32_t pad_;
} tao_match;

/* src/lib.rs:148-165 ; edit codes: 0 Match, 1 Mismatch, 2 AGap, 3 BGap, 4 Transpose */
typedef struct {
    uint32_t edit;
    uint32_t pad_;
    uint64_t count;
} tao_edit;

enum { TAO_EDIT_MATCH = 0, TAO_EDIT_MISMATCH = 1, TAO_EDIT_AGAP = 2, TAO_EDIT_BGAP = 3, TAO_EDIT_TRANSPOSE = 4 };
enum { TAO_SEARCH_ALL = 0, TAO_SEARCH_BEST = 1 };   /* src/lib.rs:171-174 */

#define TAO_NONE 0xFFFFFFFFu   /* stands for Option::None in u32 results */

/* 1 if EditCosts::new would accept (src/levenshtein.rs:38-60), else 0 (Rust: panic) */
int tao_costs_valid(const tao_costs *c);
/* 1 if check_search accepts (src/levenshtein.rs:67-71) */
int tao_costs_valid_search(const tao_costs *c);

/* src/hamming.rs:36-47; returns TAO_NONE where Rust would panic (length mismatch) */
uint32_t tao_hamming_naive(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len);

/* src/hamming.rs:96-146.  Returns number of matches, *out malloc'ed (free with tao_free). */
size_t tao_hamming_search_naive_with_opts(const uint8_t *needle, size_t needle_len,
                                          const uint8_t *haystack, size_t haystack_len,
                                          uint32_t k, int search_type, tao_match **out);

/* public hamming_search path: src/hamming.rs:454-475 (empty needle -> empty, NUL byte -> panic).
 * Returns 0 OK, 1 = would panic (null byte). */
int tao_hamming_search_simd_with_opts(const uint8_t *needle, size_t needle_len,
                                      const uint8_t *haystack, size_t haystack_len,
                                      uint32_t k, int search_type, tao_match **out, size_t *n_out);

/* src/levenshtein.rs:148-319 (full matrix).  edits may be NULL when !trace_on. */
uint32_t tao_levenshtein_naive_with_opts(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len,
                                         int trace_on, const tao_costs *costs,
                                         tao_edit **edits, size_t *n_edits);

/* src/levenshtein.rs:376-607 (banded scalar; THE bit-exactness target).
 * Returns distance or TAO_NONE. */
uint32_t tao_levenshtein_naive_k_with_opts(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len,
                                           uint32_t k, int trace_on, const tao_costs *costs,
                                           tao_edit **edits, size_t *n_edits);

/* src/levenshtein.rs:714-827 result contract via the scalar fallback (:826). */
uint32_t tao_levenshtein_simd_k_with_opts(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len,
                                          uint32_t k, int trace_on, const tao_costs *costs,
                                          tao_edit **edits, size_t *n_edits);

/* The dispatcher arithmetic alone (src/levenshtein.rs:731-791, AVX2 ladder):
 * max_k, unit_k, chosen cell width in bits (8/16/32) and the lane count of the chosen
 * Jewel type (32/64/128/256 for the u8 types; 0 for the Vec-backed Nx16x16 / Nx8x32). */
void tao_levenshtein_select(size_t a_len, size_t b_len, uint32_t k, const tao_costs *costs,
                            uint32_t *max_k, uint32_t *unit_k, uint32_t *cell_bits, uint32_t *lanes);

/* src/levenshtein.rs:1397-1399, 1419-1423, 1445-1454, 1480-1494, 1516-1526 */
uint32_t tao_levenshtein(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len);
uint32_t tao_rdamerau(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len);
uint32_t tao_levenshtein_exp(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len);
uint32_t tao_levenshtein_exp_with_opts(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len,
                                       int trace_on, const tao_costs *costs,
                                       tao_edit **edits, size_t *n_edits);
uint32_t tao_rdamerau_exp(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len);

/* src/levenshtein.rs:1589-1838.  Returns 0 OK, 1 = would panic (check_search). */
int tao_levenshtein_search_naive_with_opts(const uint8_t *needle, size_t needle_len,
                                           const uint8_t *haystack, size_t haystack_len,
                                           uint32_t k, int search_type, const tao_costs *costs,
                                           int anchored, tao_match **out, size_t *n_out);

/* default k of levenshtein_search / hamming_search: ceil(n/2)
 * (src/levenshtein.rs:1873, src/hamming.rs:423) */
uint32_t tao_default_search_k(size_t needle_len);

/* The cells the scalar banded path visits, column 0 excluded:
 * the credited unit of work for GCUPS. */
uint64_t tao_band_cells(size_t a_len, size_t b_len, uint32_t k, const tao_costs *costs);

/* Batch drivers over CSR data (host memory): plain loops over the functions above, optionally
 * split over `threads` OpenMP threads (bench.py's cpu_baseline leg and the large parity tests). */
void tao_levenshtein_k_batch(const uint8_t *a_blob, const uint64_t *a_off, const uint8_t *b_blob, const uint64_t *b_off,
                             size_t n, uint32_t k, const tao_costs *costs, uint32_t *out, int threads);
void tao_levenshtein_exp_batch(const uint8_t *a_blob, const uint64_t *a_off, const uint8_t *b_blob, const uint64_t *b_off,
                               size_t n, const tao_costs *costs, uint32_t *out, int threads);
void tao_hamming_batch(const uint8_t *a_blob, const uint64_t *a_off, const uint8_t *b_blob, const uint64_t *b_off,
                       size_t n, uint32_t *out, int threads);
int tao_max_threads(void);
/* ta_oracle_simd.c: the same contract through an anti-diagonal, auto-vectorised restatement (the shape of the reference's
 * SIMD core, src/levenshtein.rs:829-1195); start_gap_cost == 0 and k < 15,600 only: returns -1 otherwise, 0 on success. */
int tao_levenshtein_k_batch_antidiag(const uint8_t *a_blob, const uint64_t *a_off, const uint8_t *b_blob, const uint64_t *b_off,
                                     size_t n, uint32_t k, const tao_costs *costs, uint32_t *out, int threads);

/* ta_oracle_avx2.c: hand-written AVX2 anti-diagonal restatement with saturating u8 cells (the reference's Avx{1,2,4,8}x32x8
 * classes, src/levenshtein.rs:766-786) and the width ladder 8 -> 16 -> 32 around it (16-bit: ta_oracle_simd.c, 32-bit: the
 * scalar restatement).  start_gap_cost == 0 only; -1 when the host has no AVX2 or the costs are affine.  lanes_hist: 6
 * counters (no DP needed / 32 / 64 / 128 / 256 u8 lanes / wider cells) or NULL. */
int tao_have_avx2(void);
int tao_levenshtein_k_batch_ladder(const uint8_t *a_blob, const uint64_t *a_off, const uint8_t *b_blob, const uint64_t *b_off,
                                   size_t n, uint32_t k, const tao_costs *costs, uint32_t *out, int threads, uint64_t *lanes_hist);

void tao_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
