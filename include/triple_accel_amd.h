/*
 * triple_accel_amd.h -- C ABI of the MI355X-native edit-distance engine.
 *
 * Drop-in boundary for triple_accel's hot path.  The reference
 * (Rust, /root/reference) has no FFI of its own; each entry point below names the
 * reference function (file:line) it replaces, and INTEGRATION.md shows the Rust
 * `extern "C"` shim that maps the reference's public signatures onto it.
 *
 * Conventions
 *   - plain pointers + sizes, no C++/torch types; every function returns a ta_status.
 *   - Rust panics become status codes (the shim re-raises them): TA_ERR_LEN_MISMATCH,
 *     TA_ERR_NULL_BYTE, TA_ERR_BAD_COSTS.  Option::None becomes the sentinel TA_NONE
 *     (a real distance never reaches it: the reference clamps k to a true upper bound,
 *     src/levenshtein.rs:734-757).
 *   - `*_dev` / `*_batch` functions take DEVICE (HBM) pointers and a hipStream_t passed as
 *     void*; they enqueue work and return without synchronising unless stated.  The plain
 *     functions take HOST pointers, run on the current HIP device and synchronise -- except
 *     where the DEVICE SET takes over (ta_set_devices below: host batches, the queue, searches
 *     over big haystacks are partitioned over every device of the set inside the library).
 *   - Device string blobs must be readable for TA_BLOB_SLACK bytes past their last byte
 *     (kernels fetch 16-byte pieces).
 *   - There is NO CPU fallback: with no HIP device every compute entry point returns
 *     TA_ERR_HIP.
 */
#ifndef TRIPLE_ACCEL_AMD_H
#define TRIPLE_ACCEL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TA_NONE 0xFFFFFFFFu
#define TA_BLOB_SLACK 16

typedef enum {
    TA_OK = 0,
    TA_ERR_LEN_MISMATCH = 1, /* assert!(a.len() == b.len())     src/hamming.rs:38,318 */
    TA_ERR_NULL_BYTE = 2,    /* check_no_null_bytes panic        src/lib.rs:237-243    */
    TA_ERR_BAD_COSTS = 3,    /* EditCosts::new / check_search    src/levenshtein.rs:44-52,67-71 */
    TA_ERR_HIP = 4,          /* HIP runtime failure / no device (no CPU fallback) */
    TA_ERR_ARG = 5,          /* null pointer, size over the documented limit */
    TA_ERR_UNSUPPORTED = 6,  /* outside what the GPU path covers (e.g. a traceback needing more than 8 GB of records) */
    TA_ERR_CAPACITY = 7,     /* caller-provided match buffer too small; *n_out holds the need */
    TA_ERR_DIV_ZERO = 8      /* "attempt to divide by zero": hamming_search_naive*, empty needle, Best  src/hamming.rs:136 */
} ta_status;

/* EditCosts, src/levenshtein.rs:20-26 (fields are private there; built via new / the consts) */
typedef struct {
    uint8_t mismatch_cost;
    uint8_t gap_cost;
    uint8_t start_gap_cost;
    uint8_t has_transpose;  /* Option<u8>::is_some() */
    uint8_t transpose_cost;
} ta_edit_costs;

/* Match, src/lib.rs:135-142 (start inclusive, end exclusive) */
typedef struct {
    uint64_t start;
    uint64_t end;
    uint32_t k;
    uint32_t pad_;
} ta_match;

/* Edit / EditType, src/lib.rs:148-165 */
typedef enum { TA_EDIT_MATCH = 0, TA_EDIT_MISMATCH = 1, TA_EDIT_AGAP = 2, TA_EDIT_BGAP = 3, TA_EDIT_TRANSPOSE = 4 } ta_edit_type;
typedef struct {
    uint32_t edit;   /* ta_edit_type */
    uint32_t pad_;
    uint64_t count;
} ta_edit;

/* SearchType, src/lib.rs:171-174 */
typedef enum { TA_SEARCH_ALL = 0, TA_SEARCH_BEST = 1 } ta_search_type;

/* LEVENSHTEIN_COSTS / RDAMERAU_COSTS, src/levenshtein.rs:76-89 */
ta_edit_costs ta_levenshtein_costs(void);
ta_edit_costs ta_rdamerau_costs(void);
/* EditCosts::new validation, src/levenshtein.rs:38-60: TA_OK or TA_ERR_BAD_COSTS */
int ta_edit_costs_new(uint8_t mismatch, uint8_t gap, uint8_t start_gap, int has_transpose, uint8_t transpose,
                      ta_edit_costs *out);
/* check_search, src/levenshtein.rs:67-71 */
int ta_edit_costs_check_search(const ta_edit_costs *c);

/* ---- runtime ------------------------------------------------------------------------- */
const char *ta_version(void);
const char *ta_status_str(int status);
/* Options of the calling thread (0 = off, the default).
 *   TA_OPT_EARLY_OUT: the bit-parallel band kernels of fixed-length unit-cost batches (bands of up to 33 diagonals) stop a
 *   wavefront as soon as none of its pairs can still end at or below k (the top cell of the current band column minus the
 *   downward steps below it -- a lower bound of every cell of the column -- exceeds k; checked every 24-32 columns).  The answers are the same -- those pairs are None either way
 *   (src/levenshtein.rs:539-541) -- but the work then depends on the data: batches of dissimilar strings finish after a few
 *   dozen columns.  Off by default: the reference evaluates its whole band, and so does every benchmark figure of this library.
 *   TA_OPT_UNIT_PREFILTER: batches under weighted EditCosts (ta_levenshtein_k_batch, >= 1024 pairs) first run the unit-cost bit-parallel
 *   pass with k' = max(k / min(mc, tc), (k - sg) / min(mc, gc, tc)): a pair whose UNIT distance exceeds k' has a weighted distance above k --
 *   None (src/levenshtein.rs:539-541) -- and only the others are priced by the DP band kernel with the real costs.  Same answers; the work
 *   depends on the data (dissimilar batches: the unit pass alone, 0.3 of the weighted pass; near pairs: both).  Off by default. */
enum { TA_OPT_EARLY_OUT = 1, TA_OPT_UNIT_PREFILTER = 2 };
int ta_set_option(int option, int value);
/* number of visible HIP devices (0 => every compute call returns TA_ERR_HIP) */
int ta_device_count(void);
/* text of the last HIP error seen on this thread ("" if none) */
const char *ta_last_error(void);
/* the dominant kernel of the calling thread's last pass, as a profiler prints it (no "void ta::", no parameter list); "" before the first pass */
const char *ta_last_kernel_name(void);

/* The dispatcher arithmetic of levenshtein_simd_k_with_opts, src/levenshtein.rs:731-791:
 * clamped max_k, unit_k, and the cell width (8/16/32 bits) the reference would pick.
 * The GPU kernels keep u8/u16/u32 *semantics* by this rule (DESIGN.md "cell width"). */
typedef struct {
    uint32_t max_k;
    uint32_t unit_k;
    uint32_t cell_bits;       /* 8, 16 or 32 */
    uint32_t ref_lanes;       /* 32/64/128/256 for the u8 Jewel types, 0 for the Vec-backed ones */
} ta_lev_select;
int ta_levenshtein_select(size_t a_len, size_t b_len, uint32_t k, const ta_edit_costs *costs, ta_lev_select *out);

/* What the last distance batch launched on this thread used (for tests / debug logging;
 * the analogue of the reference's `debug` feature println, src/levenshtein.rs:840-847). */
typedef struct {
    uint32_t kernel;          /* 1 = band-wavefront (registers+DPP), 2 = wide-band workgroup, 3 = bit-parallel band (unit costs), 4 = bit-parallel full columns (unit costs, long pairs), 5 = pair-sliced systolic band (EXPERIMENTAL builds), 6 = single pair, band <= 64 diagonals (unit costs), 7 = small-alphabet bit-parallel band (ta_levenshtein_k_batch_alphabet), 8 = checkpoint-and-recompute traceback of the unit-cost families (ta_levenshtein_trace_batch, bands of up to 33 diagonals); pairs_per_wave 128 with kernel 3 = two pairs per lane */
    uint32_t diags_per_lane;  /* D */
    uint32_t lanes_per_pair;  /* L */
    uint32_t pairs_per_wave;
    uint32_t band_offset;     /* o: diagonal index of d = 0 */
    uint32_t cell_bits;       /* width class chosen by the reference's rule for the batch's max lengths */
    uint32_t affine;
    uint32_t transpose;
    uint32_t grid;
    uint32_t lds_bytes;
} ta_launch_info;
int ta_last_launch_info(ta_launch_info *out);

/* ---- single-call host API (the reference's function set) ------------------------------- */

/* hamming(a, b), src/hamming.rs:390 -> :317 -> :36-47 */
int ta_hamming(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len, uint32_t *out);

/* levenshtein_simd_k_with_opts(a, b, k, trace_on, costs), src/levenshtein.rs:714-720.
 * *out = distance, or TA_NONE when the distance exceeds k.  trace_on != 0 -> TA_ERR_UNSUPPORTED here: the traceback
 * has its own entry point, ta_levenshtein_trace, because it returns an edit script. */
int ta_levenshtein_simd_k_with_opts(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len,
                                    uint32_t k, int trace_on, const ta_edit_costs *costs, uint32_t *out);
/* levenshtein_simd_k_with_opts(a, b, k, true, costs): distance + run-length traceback (library-owned, ta_free).
 * 2-bit argmin codes come from the band-wavefront kernel; the walk is host code.  Bands wider than 4222 diagonals
 * (unit_k > 4220): the unit-cost families use the row-blocked bit-parallel kernel with 3-bit records, other costs the DP
 * wide kernel with 2-bit codes; more than 8 GB of records returns TA_ERR_UNSUPPORTED. */
int ta_levenshtein_trace(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len, uint32_t k,
                         const ta_edit_costs *costs, uint32_t *out, ta_edit **edits, size_t *n_edits);
/* levenshtein_exp_with_opts(a, b, true, costs), src/levenshtein.rs:1480-1494 */
int ta_levenshtein_exp_trace(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len,
                             const ta_edit_costs *costs, uint32_t *out, ta_edit **edits, size_t *n_edits);
/* levenshtein_simd_k, src/levenshtein.rs:677-684 */
int ta_levenshtein_simd_k(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len, uint32_t k, uint32_t *out);
/* levenshtein, src/levenshtein.rs:1397 ; rdamerau :1419 */
int ta_levenshtein(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len, uint32_t *out);
int ta_rdamerau(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len, uint32_t *out);
/* levenshtein_exp :1445, levenshtein_exp_with_opts :1480 (trace_on unsupported), rdamerau_exp :1516 */
int ta_levenshtein_exp(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len, uint32_t *out);
int ta_levenshtein_exp_with_opts(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len,
                                 int trace_on, const ta_edit_costs *costs, uint32_t *out);
int ta_rdamerau_exp(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len, uint32_t *out);

/* levenshtein_search_simd_with_opts(needle, haystack, k, search_type, costs, anchored),
 * src/levenshtein.rs:1911-1918.  This entry returns the whole result: *out is library-owned (release with ta_free), *n_out its
 * length.  The reference's iterator is lazy; ta_levenshtein_search_first (below) is its `.next()`. */
int ta_levenshtein_search_simd_with_opts(const uint8_t *needle, size_t needle_len,
                                         const uint8_t *haystack, size_t haystack_len,
                                         uint32_t k, int search_type, const ta_edit_costs *costs, int anchored,
                                         ta_match **out, size_t *n_out);
/* levenshtein_search, src/levenshtein.rs:2508 (k = ceil(n/2), Best, LEVENSHTEIN_COSTS, unanchored) */
int ta_levenshtein_search(const uint8_t *needle, size_t needle_len, const uint8_t *haystack, size_t haystack_len,
                          ta_match **out, size_t *n_out);
/* hamming_search_simd_with_opts, src/hamming.rs:454 ; hamming_search :588 */
int ta_hamming_search_simd_with_opts(const uint8_t *needle, size_t needle_len,
                                     const uint8_t *haystack, size_t haystack_len,
                                     uint32_t k, int search_type, ta_match **out, size_t *n_out);
int ta_hamming_search(const uint8_t *needle, size_t needle_len, const uint8_t *haystack, size_t haystack_len,
                      ta_match **out, size_t *n_out);
/* hamming_search_naive_with_opts, src/hamming.rs:96-146 (and hamming_search_naive :70 with k = ceil(n/2), Best): the scalar
 * routine's contract -- no NUL-byte panic, an empty needle matches with k = 0 at every offset 0..=haystack_len. */
int ta_hamming_search_naive_with_opts(const uint8_t *needle, size_t needle_len,
                                      const uint8_t *haystack, size_t haystack_len,
                                      uint32_t k, int search_type, ta_match **out, size_t *n_out);
/* The FIRST element of levenshtein_search_simd_with_opts(.., SearchType::All, ..): what `.next()` on the reference's lazy
 * iterator returns (src/levenshtein.rs:2282-2420; tests/basic_tests.rs:628-632) without scanning the rest of the haystack.
 * The haystack is searched window by window (64 KiB, then four times as much each time, each window behind needle_len +
 * unit_k + 2 bytes of left context -- exact for every cost <= k) and the scan stops at the first window that holds a hit;
 * the host form also uploads only that far.  *found = 0 when there is no match at all.  The bindings' All-mode iterators
 * call this for their first element and run the full search only when a second one is asked for. */
int ta_levenshtein_search_first(const uint8_t *needle, size_t needle_len, const uint8_t *haystack, size_t haystack_len,
                                uint32_t k, const ta_edit_costs *costs, int anchored, ta_match *out, int *found);
/* The All-mode result for a caller that has just taken its first element with ta_levenshtein_search_first on THE SAME haystack (same pointer
 * and length, bytes unchanged: the caller's promise -- the bindings' lazy iterators hold the haystack immutable): the bytes that call uploaded
 * stay on the device, only the rest travels (src/levenshtein.rs:2282-2420: the reference's lazy iterator continues where it stopped).  After
 * any other call of the thread it is ta_levenshtein_search_simd_with_opts(.., TA_SEARCH_ALL, ..). */
int ta_levenshtein_search_resume(const uint8_t *needle, size_t needle_len, const uint8_t *haystack, size_t haystack_len,
                                 uint32_t k, const ta_edit_costs *costs, int anchored, ta_match **out, size_t *n_out);
/* the same over a haystack shard resident in HBM (positions + base; the end == 0 match is the caller's, as with *_search_dev) */
int ta_levenshtein_search_first_dev(const uint8_t *needle_host, size_t needle_len,
                                    const uint8_t *haystack_dev, size_t haystack_len,
                                    uint32_t k, const ta_edit_costs *costs, uint64_t base,
                                    ta_match *out, int *found, void *stream);
/* Limits of the host search forms: needles up to 65,535 bytes (TA_ERR_ARG beyond); the result list is bounded only by memory
 * (a pass that overflows its first hit buffer is repeated once with room for the count it reported). */
void ta_free(void *p);

/* Threading.  Every entry point may be called from any number of threads at once.  The single-call host functions
 * (ta_hamming .. ta_levenshtein_exp, the *_search host forms, the tracebacks) run on a stream the library keeps per calling
 * thread (non-blocking: concurrent callers never serialise on the null stream) and stage short pairs through a pinned,
 * device-mapped buffer of that thread -- one kernel launch and one stream synchronisation per call.  The batch / *_dev
 * functions run on the stream the caller passes; a thread that switches streams between two such calls is ordered by an
 * event the library records at the end of each call (the thread-local scratch of the first call may still be in use).
 * ta_thread_release() frees what the calling thread holds inside the library (device scratch, its stream, the pinned
 * buffer); optional -- without it they live until process exit.
 * Graph capture.  The batch / *_dev entries that say so may be captured into a hipGraph (hipStreamBeginCapture on the stream they are given).
 * A captured call bakes the calling thread's scratch pointers into the graph: run the same call once OUTSIDE the capture first (it sizes the
 * scratch); a call that would have to grow the scratch during a capture returns TA_ERR_UNSUPPORTED instead of allocating.  The graph is
 * invalid after any later call of the thread that grows the scratch (a bigger batch, a wider band) and after ta_thread_release(). */
void ta_thread_release(void);

/* ---- batch API on device-resident data (new surface; N = 1 equals the single-call form) -- */

/* String i of a batch is blob[off[i] .. off[i+1]) (CSR, n+1 offsets, device memory), or, when
 * off == NULL, blob[i*stride .. i*stride+len) (fixed length `len`, byte stride `stride`). */
typedef struct {
    const uint8_t *blob;     /* device */
    const uint64_t *off;     /* device, n+1 entries, or NULL for the strided form */
    uint64_t stride;         /* strided form only */
    uint64_t len;            /* strided form only */
    uint64_t max_len;        /* upper bound on any string length (CSR form; 0 = let the library measure it) */
} ta_strings;

/* N x levenshtein_simd_k_with_opts(a_i, b_i, k, false, costs) -> out[i] (u32 or TA_NONE). */
int ta_levenshtein_k_batch(const ta_strings *a, const ta_strings *b, size_t n, uint32_t k,
                           const ta_edit_costs *costs, uint32_t *out_dev, void *stream);
/* The same for strings written in a small alphabet the caller names: at most four distinct byte values -- DNA, RNA -- for which
 * a two-bit code (byte >> h) & 3 exists, or up to 32 -- IUPAC nucleotide codes, amino acids, digits, one case of the letters -- with a
 * five-bit code (byte >> h) & 31 and the byte's other three bits the same in every symbol.  The match vector of a column becomes a
 * table lookup: about half (four symbols) / two thirds (twenty) of the instructions of the byte test.  The promise is verified on
 * the device: pairs that hold any other byte are answered by the general kernel inside the same call.  Batches the small-alphabet
 * kernels do not cover run ta_levenshtein_k_batch unchanged. */
int ta_levenshtein_k_batch_alphabet(const ta_strings *a, const ta_strings *b, size_t n, uint32_t k, const ta_edit_costs *costs,
                                    const uint8_t *alphabet, size_t alphabet_len, uint32_t *out_dev, void *stream);
/* N x levenshtein_exp_with_opts(a_i, b_i, false, costs): doubling k from 30 over the still-unresolved
 * subset (src/levenshtein.rs:1480-1494).  Batches of >= 1024 pairs: the whole k schedule is enqueued at once -- every list's length stays on
 * the device -- and the call returns without synchronising (capturable in a graph; a CSR side with max_len = 0 costs one synchronisation to
 * measure it).  Smaller batches synchronise the stream between rounds.  The device-driven rounds size every launch for the whole batch (a list's
 * length is only known on the device): a round whose list is short or empty still costs its launches (10-15 us each) and takes the kernel the
 * batch size suggests -- a batch where a handful of long pairs survive to the unbounded pass pays for that; TA_EXP_HOST_ROUNDS=1 (TA_TUNING) keeps
 * the host-driven loop, which sizes every round for the pairs that are left.  The first threshold is the widest the cheapest kernel's window
 * holds (k = 32 for LEVENSHTEIN_COSTS: the price of 30), then the reference's 60, 120, ...: the returned distances do not depend on the schedule. */
int ta_levenshtein_exp_batch(const ta_strings *a, const ta_strings *b, size_t n,
                             const ta_edit_costs *costs, uint32_t *out_dev, void *stream);
/* Batch form of ta_levenshtein_trace (no reference analogue: the reference's trace_on is per call, src/levenshtein.rs:714-720, :561-606):
 * out_dev[i] = distance | TA_NONE, n_edits_dev[i] = runs of pair i's script (0 for None), edits_dev[i * cap .. i * cap + n_edits_dev[i]) =
 * the script front to back -- edit for edit the reference's Vec<Edit>.  All buffers are device memory; the call enqueues kernels on
 * `stream` and returns (no synchronisation).  2 k + 1 records per pair always suffice; a longer script is cut at `cap` records and
 * n_edits_dev[i] says how many it has.  LEVENSHTEIN_COSTS / RDAMERAU_COSTS with k <= 32 (30): no per-cell records -- the distance pass leaves a
 * checkpoint of its column state every 16th column, one more kernel recomputes tile after tile backwards and walks; any
 * other costs / wider bands: 2-bit argmin codes of the DP band kernel + a walk kernel (3.4b).  EditCosts(g, g, 0, None | Some(g)) -- unit costs
 * times g -- ride the checkpoint route with k / g (the script is the unit-cost script, the distance g times the unit one).  Batches whose
 * scratch (checkpoints, run lists) would not fit are worked through in sub-batches of whole wavefronts on the same stream.
 * TA_ERR_UNSUPPORTED for bands beyond the register kernel (> 4222 diagonals). */
int ta_levenshtein_trace_batch(const ta_strings *a, const ta_strings *b, size_t n, uint32_t k, const ta_edit_costs *costs,
                               uint32_t *out_dev, ta_edit *edits_dev, uint32_t *n_edits_dev, size_t cap, void *stream);

/* ta_levenshtein_trace_batch with PACKED records: one 32-bit word per run, (edit type << 29) | count; pair i's script is the
 * min(n_edits_dev[i], cap) words that END at packed_dev[(i + 1) * cap] -- front to back, right-aligned in the pair's slot of `cap` words (a
 * script of more than `cap` runs keeps its LAST cap runs and n_edits_dev[i] says how long it is; the words in front of a script are not
 * written).  A quarter of the bytes of the 16-byte ta_edit form, which stays the ABI's default; the bindings expand the words on the host.
 * On the checkpoint route (LEVENSHTEIN_COSTS / RDAMERAU_COSTS and their multiples EditCosts(g, g, 0, None | Some(g)), k / g <= 32 (30)) the
 * walk writes every run where it belongs: no run lists in scratch, no reversal step.  Strings of 2^29 bytes and more: TA_ERR_UNSUPPORTED. */
int ta_levenshtein_trace_batch_packed(const ta_strings *a, const ta_strings *b, size_t n, uint32_t k, const ta_edit_costs *costs,
                                      uint32_t *out_dev, uint32_t *packed_dev, uint32_t *n_edits_dev, size_t cap, void *stream);

/* N x hamming(a_i, b_i); out[i] = TA_NONE where the lengths differ (Rust: panic). */
int ta_hamming_batch(const ta_strings *a, const ta_strings *b, size_t n, uint32_t *out_dev, void *stream);

/* ---- a queue for callers that produce pairs ONE AT A TIME (the reference's calling convention, src/levenshtein.rs:714-720) but
 * can wait for the answers: a single call costs a kernel launch (22-25 us for a 256-byte pair, against ~2 us on a host core);
 * pushed pairs are copied into pinned staging and answered together by ONE ta_levenshtein_k_batch per flush.
 *   ta_queue_create: every pair of the queue is levenshtein_simd_k_with_opts(a, b, k, false, costs).
 *   ta_queue_push:   copies the pair; *ticket = its index in the next flush's result array.
 *   ta_queue_flush:  uploads, one batch pass (CSR, length-ordered on the device), downloads; *results (library-owned, valid until
 *                    the next push / flush / destroy) holds *n answers (TA_NONE = None) in push order; the queue is empty again.
 * A queue belongs to one thread at a time.  Tickets restart at 0 after every flush. */
typedef struct ta_queue ta_queue;
int ta_queue_create(uint32_t k, const ta_edit_costs *costs, ta_queue **out);
int ta_queue_push(ta_queue *q, const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len, size_t *ticket);
int ta_queue_flush(ta_queue *q, const uint32_t **results, size_t *n);
void ta_queue_destroy(ta_queue *q);

/* ---- the device set: the multi-GPU split behind this boundary ----------------------------------------------------------
 * BASELINE.json north_star: "large batches of independent string pairs -- and for levenshtein_search the haystack itself -- are
 * partitioned across the 8 GPUs of one node", for callers that stay on the reference's calling convention: host slices, one process
 * (src/levenshtein.rs:714-720, 1911-1918, 2508-2511; src/hamming.rs:454-475).  The library keeps one worker thread per entry of the
 * device set (hipSetDevice, its own stream, pinned staging ring and scratch).  Pairs shard as contiguous ranges, a haystack as contiguous
 * byte ranges behind needle_len + unit_k + 2 bytes of left context (Levenshtein) / in front of needle_len - 1 bytes of overlap (Hamming);
 * there is no device-to-device traffic: results return over each device's own PCIe link and are concatenated in shard order.
 *   ta_set_devices(ids, n): the set (NULL / 0: every visible device, also the default).  An id may appear more than once -- that many
 *   workers share the device (how a one-GPU box exercises the N-way logic).  Not to be called while other calls are in flight; resident
 *   handles created before keep the workers they were created on.  A process that owns ONE GPU of a node (one rank per GPU under a launcher,
 *   all GPUs visible) should name it -- ta_set_devices(&mine, 1) -- or its big host calls fan out over its neighbours' devices too.
 * What fans out when the set has more than one entry: the *_host batch entries below, ta_queue_flush (and through it the bindings'
 * levenshtein_simd_k_with_opts_many), and the host search entries ta_levenshtein_search_simd_with_opts (unanchored) /
 * ta_hamming_search_simd_with_opts / ta_hamming_search_naive_with_opts for haystacks of >= 8 MiB (>= 4 MiB per device used).
 * Results are identical to the one-device path's, bit for bit and in the same order. */
int ta_set_devices(const int *devices, size_t n);
int ta_get_devices(int *out, size_t cap, size_t *n_out);

/* Batches whose strings live in HOST memory (ta_strings with host pointers: CSR offsets or the strided form): N x
 * levenshtein_simd_k_with_opts(a_i, b_i, k, false, costs) / levenshtein_exp_with_opts(a_i, b_i, false, costs) / hamming(a_i, b_i)
 * -> out[i] (host; TA_NONE = None / a length mismatch).  Every device of the set takes a contiguous slice of the pairs (no device gets
 * fewer than 4,096) and works through it in chunks of <= 64 MiB of strings: uploaded (big pieces by the runtime's path for pageable memory, CSR
 * offsets rebased through a pinned ring), answered by the batch entry of the same name, the answers downloaded; a device whose slice holds >= 16 MiB
 * of strings is fed by two threads and streams.  Synchronous. */
int ta_levenshtein_k_batch_host(const ta_strings *a_host, const ta_strings *b_host, size_t n, uint32_t k,
                                const ta_edit_costs *costs, uint32_t *out_host);
int ta_levenshtein_exp_batch_host(const ta_strings *a_host, const ta_strings *b_host, size_t n,
                                  const ta_edit_costs *costs, uint32_t *out_host);
int ta_hamming_batch_host(const ta_strings *a_host, const ta_strings *b_host, size_t n, uint32_t *out_host);
/* N x levenshtein_simd_k_with_opts(a_i, b_i, k, true, costs) for HOST strings over the device set: out_host[i] = distance | TA_NONE, n_edits_host[i] =
 * the runs of pair i's script (0 for None), packed_host[i * cap ..]: the script as packed runs, (edit type << 29) | count, front to back, in the LAST
 * min(n_edits_host[i], cap) words of the pair's slot (ta_levenshtein_trace_batch_packed's layout; the words in front of a script are undefined).
 * cap = 2 k + 1 holds every script of cost <= k; a longer script keeps its last cap runs. */
int ta_levenshtein_trace_batch_host(const ta_strings *a_host, const ta_strings *b_host, size_t n, uint32_t k, const ta_edit_costs *costs,
                                    uint32_t *out_host, uint32_t *packed_host, uint32_t *n_edits_host, size_t cap);

/* A pair batch uploaded ONCE and kept resident, sharded over the first n_shards devices of the set (0: as the *_host entries choose);
 * every later call runs the shards side by side and downloads 4 bytes per pair.  ta_sharded_pairs_time_levenshtein_k: `steps` passes back
 * to back with the answers left on the devices; *device_ms = the slowest shard's device time (HIP events on its worker's stream). */
typedef struct ta_sharded_pairs ta_sharded_pairs;
int ta_sharded_pairs_upload(const ta_strings *a_host, const ta_strings *b_host, size_t n, size_t n_shards, ta_sharded_pairs **out);
int ta_sharded_pairs_levenshtein_k(ta_sharded_pairs *s, uint32_t k, const ta_edit_costs *costs, uint32_t *out_host);
int ta_sharded_pairs_levenshtein_exp(ta_sharded_pairs *s, const ta_edit_costs *costs, uint32_t *out_host);
int ta_sharded_pairs_hamming(ta_sharded_pairs *s, uint32_t *out_host);
int ta_sharded_pairs_time_levenshtein_k(ta_sharded_pairs *s, uint32_t k, const ta_edit_costs *costs, int steps, float *device_ms);
int ta_sharded_pairs_shards(const ta_sharded_pairs *s, size_t *n_shards, size_t *n_pairs);
void ta_sharded_pairs_free(ta_sharded_pairs *s);

/* A haystack uploaded ONCE and kept resident as contiguous shards (BASELINE config 5: one shard per GPU), each with `overlap` bytes of its
 * neighbours on either side: every search with needle_len + unit_k + 2 <= overlap (Levenshtein) / needle_len - 1 <= overlap (Hamming) runs
 * on the resident bytes -- TA_ERR_ARG beyond.  The results are those of ta_levenshtein_search_simd_with_opts (unanchored) /
 * ta_hamming_search_simd_with_opts over the whole haystack: library-allocated (ta_free). */
typedef struct ta_sharded_haystack ta_sharded_haystack;
int ta_sharded_haystack_upload(const uint8_t *haystack, size_t len, size_t overlap, size_t n_shards, ta_sharded_haystack **out);
int ta_sharded_haystack_levenshtein_search(ta_sharded_haystack *h, const uint8_t *needle, size_t needle_len, uint32_t k, int search_type,
                                           const ta_edit_costs *costs, ta_match **out, size_t *n_out);
int ta_sharded_haystack_hamming_search(ta_sharded_haystack *h, const uint8_t *needle, size_t needle_len, uint32_t k, int search_type,
                                       ta_match **out, size_t *n_out);
int ta_sharded_haystack_shards(const ta_sharded_haystack *h, size_t *n_shards, size_t *len);
void ta_sharded_haystack_free(ta_sharded_haystack *h);

/* All-mode hits of one haystack shard resident in HBM (levenshtein_search_simd_with_opts with
 * SearchType::All; the order-dependent Best fold is a sequential host pass, ta_search_fold_best).
 * `base` is added to start/end (global offset of this shard inside a larger haystack);
 * `emit_from` suppresses hits whose end <= emit_from (left-halo positions).
 * hits_dev receives up to `cap` records in NO particular order (atomic cursor): sort them by `end` -- unique per
 * hit -- before ta_search_fold_best; *count_host receives the total found
 * (may exceed cap => TA_ERR_CAPACITY after sync).  Synchronises the stream. */
int ta_levenshtein_search_dev(const uint8_t *needle_host, size_t needle_len,
                              const uint8_t *haystack_dev, size_t haystack_len,
                              uint32_t k, const ta_edit_costs *costs, int anchored,
                              uint64_t base, uint64_t emit_from,
                              ta_match *hits_dev, size_t cap, uint64_t *count_host, void *stream);
int ta_hamming_search_dev(const uint8_t *needle_host, size_t needle_len,
                          const uint8_t *haystack_dev, size_t haystack_len, uint32_t k,
                          uint64_t base, ta_match *hits_dev, size_t cap, uint64_t *count_host, void *stream);
/* ta_hamming_search_dev with the hits on the host, sorted by end, library-allocated (ta_free): the hits of an ordinary search (up to 680)
 * arrive through host-mapped pinned memory with the call's one stream synchronisation; more are copied from hits_dev (which must hold
 * `cap` records either way).  Synchronises the stream.  (src/hamming.rs:454-554: the All-mode result of hamming_search_simd_with_opts.) */
int ta_hamming_search_dev_sorted(const uint8_t *needle_host, size_t needle_len,
                                 const uint8_t *haystack_dev, size_t haystack_len, uint32_t k,
                                 uint64_t base, ta_match *hits_dev, size_t cap, ta_match **out, size_t *n_out, void *stream);

/* The Best-mode pass over one shard in a single call (what `levenshtein_search` with SearchType::Best needs from a shard,
 * src/levenshtein.rs:1792-1835): ta_levenshtein_search_dev (unanchored) + ta_search_best_hits_dev fused.  The All-mode hits stay
 * in hits_dev (*count_host of them); *out receives only the hits with the smallest k -- the only ones ta_search_fold_best can
 * keep -- library-allocated (ta_free), sorted by end.  For unit-cost families and needles of up to 64 bytes the whole pass is
 * one fill, two kernels and one stream synchronisation (the number of flagged blocks, the minimum and the selection never leave
 * the device; the result arrives in host-mapped memory).  Synchronises the stream. */
int ta_levenshtein_search_best_dev(const uint8_t *needle_host, size_t needle_len,
                                   const uint8_t *haystack_dev, size_t haystack_len,
                                   uint32_t k, const ta_edit_costs *costs, uint64_t base, uint64_t emit_from,
                                   ta_match *hits_dev, size_t cap, uint64_t *count_host,
                                   ta_match **out, size_t *n_out, void *stream);

/* Best-mode shortcut for a shard whose All-mode hits sit in HBM (the hits_dev / count of ta_*_search_dev): only the hits
 * with the smallest k can survive ta_search_fold_best, so the minimum and the selection run on the device and only those
 * records come back -- library-allocated (ta_free), sorted by end, ready for ta_search_fold_best.  Synchronises. */
int ta_search_best_hits_dev(const ta_match *hits_dev, uint64_t count, ta_match **out, size_t *n_out, void *stream);

/* Sequential Best post-pass over All-mode hits in increasing `end` order (host memory):
 * running curr_k, overlap fold, final filter (src/levenshtein.rs:1792-1796, 1812-1835;
 * hamming: src/hamming.rs:122-143 with fold = 0).  In place; returns the new count. */
size_t ta_search_fold_best(ta_match *hits, size_t n, uint32_t k, int overlap_fold);

#ifdef __cplusplus
}
#endif
#endif
