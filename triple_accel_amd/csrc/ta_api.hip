// ta_api.hip -- the C ABI (include/triple_accel_amd.h): validation, launch planning, staging.
// No CPU fallback lives here: every compute entry point needs a HIP device.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <mutex>
#include <vector>

#include "lev_trace_walk.h"
#include "ta_internal.h"

namespace ta {

static thread_local std::string g_last_error;
static thread_local ta_launch_info g_last_launch = {};
// true when the last distance pass of this thread was ONE kernel whose only store to its result slot is the answer: the
// single-call entry points may then watch the pinned result word instead of synchronising the stream (fetch_u32)
static thread_local bool g_answer_single_store = false;
static thread_local int g_exp_passes = 0;          // distance passes the last ta_levenshtein_exp_batch of this thread launched

void set_last_error(const char *what, hipError_t e) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
}
void set_last_error_msg(const char *msg) { g_last_error = msg; }
static thread_local char g_last_kernel_name[96] = "";
void set_last_kernel_name(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_kernel_name, sizeof(g_last_kernel_name), fmt, ap);
    va_end(ap);
}

bool device_ready() {
    static int state = -1;   // -1 unknown, 0 no, 1 yes
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (state < 0) {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        state = (e == hipSuccess && n > 0) ? 1 : 0;
        if (!state) set_last_error_msg("no HIP device available (triple_accel_amd has no CPU fallback)");
    }
    if (!state) set_last_error_msg("no HIP device available (triple_accel_amd has no CPU fallback)");
    return state == 1;
}

// set by every Scratch::ensure, consumed by the StreamGuard of the call: did this call put thread-local scratch to use?
static thread_local bool g_scratch_in_use = false;

// true while the call's stream is being captured into a graph (StreamGuard): thread-local scratch must not grow then -- hipFree / hipMalloc are
// illegal inside a capture, and the graph bakes the scratch pointers in
static thread_local bool g_call_capturing = false;

int Scratch::ensure(size_t bytes) {
    g_scratch_in_use = true;
    if (bytes <= cap) return TA_OK;
    if (g_call_capturing) {
        set_last_error_msg("stream capture: the call needs more thread-local scratch than this thread holds -- run the same call once outside the capture first "
                           "(the graph stays valid until a later call of the thread grows the scratch, or ta_thread_release)");
        return TA_ERR_UNSUPPORTED;
    }
    if (dev) { (void)hipFree(dev); dev = nullptr; cap = 0; }
    size_t want = bytes < 4096 ? 4096 : bytes + bytes / 4;
    TA_HIP(hipMalloc(&dev, want));
    cap = want;
    return TA_OK;
}
void Scratch::release() {
    if (dev) (void)hipFree(dev);
    dev = nullptr; cap = 0;
}
Scratch::~Scratch() { /* at thread exit the HIP runtime may already be gone (process teardown): ta_thread_release() is the explicit way */ }
Scratch &tls_scratch(int which) {
    static thread_local Scratch s[TA_SCRATCH_SLOTS];
    return s[which];
}

int CallCtx::ensure() {
    if (st && pin) return TA_OK;
    if (!st) TA_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    if (!pin) {
        void *h = nullptr, *d = nullptr;
        TA_HIP(hipHostMalloc(&h, PIN_BYTES, hipHostMallocMapped));
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) d = h;
        pin = (uint8_t *)h; pin_dev = (uint8_t *)d;
    }
    return TA_OK;
}
void CallCtx::release() {
    if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); st = nullptr; }
    if (pin) { (void)hipHostFree(pin); pin = nullptr; pin_dev = nullptr; }
}
CallCtx &call_ctx() {
    static thread_local CallCtx c;
    return c;
}

int PinBox::ensure() {
    if (host) return TA_OK;
    void *h = nullptr, *d = nullptr;
    TA_HIP(hipHostMalloc(&h, SEARCH_REPORT_BYTES, hipHostMallocMapped));
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) d = h;
    host = (uint8_t *)h; dev = (uint8_t *)d;
    return TA_OK;
}
void PinBox::release() {
    if (host) (void)hipHostFree(host);
    host = dev = nullptr;
}
PinBox &search_report_box() {
    static thread_local PinBox b;
    return b;
}

struct LastUse { hipStream_t st = nullptr; hipEvent_t ev = nullptr; bool pending = false; };
static LastUse &last_use() {
    static thread_local LastUse u;
    return u;
}
// A call that put thread-local scratch to use records an event at its end (its kernels may still be reading that scratch when
// it returns); a later call of the thread on ANOTHER stream waits for that event first.  Calls that touch no scratch -- the
// fixed-length batch passes, ta_hamming_batch: the hot paths -- record nothing (an event record per call was a fifth of a small
// batch call's cost).  The event is recorded while the stream is certainly alive: the caller may destroy it afterwards.
// A stream that is being CAPTURED into a graph neither waits for the event nor records it: an event recorded inside a capture belongs to
// the capture (a later wait on it from an ordinary stream is not a wait for the replayed graph's kernels -- it put the kernels of the
// following calls out of order on gfx950: a memory fault in bench.py --unit-prefilter under its hipGraph), and the graph's own edges
// order the captured calls.  Whoever replays a captured call next to ordinary calls of the same thread on ANOTHER stream orders them
// himself (one stream, or an event of his own): the thread-local scratch is shared.
static bool stream_is_capturing(hipStream_t s) {
    if (!s) return false;                                 // (the null stream cannot be captured)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}
StreamGuard::StreamGuard(hipStream_t s) : st(s) {
    LastUse &u = last_use();
    was_capturing = g_call_capturing;
    capturing = stream_is_capturing(s);
    g_call_capturing = capturing;
    if (u.pending && u.st != s && u.ev && !capturing) (void)hipStreamWaitEvent(s, u.ev, 0);
}
StreamGuard::~StreamGuard() {
    g_call_capturing = was_capturing;
    if (!g_scratch_in_use) return;
    g_scratch_in_use = false;
    if (capturing) return;
    LastUse &u = last_use();
    if (!u.ev && hipEventCreateWithFlags(&u.ev, hipEventDisableTiming) != hipSuccess) { u.ev = nullptr; return; }
    if (hipEventRecord(u.ev, st) == hipSuccess) { u.st = st; u.pending = true; }
}

// Test / tuning switches.  They are honoured only when TA_TUNING was set in the environment when the
// library was loaded: a production process reads the environment exactly once (here) and never on the call path.
static const bool g_tuning = getenv("TA_TUNING") != nullptr;
bool tuning_enabled() { return g_tuning; }
const char *env_str(const char *name) { return g_tuning ? getenv(name) : nullptr; }
int env_int(const char *name) {
    const char *v = env_str(name);
    return v ? atoi(v) : 0;
}

static thread_local bool g_opt_early_out = false;
bool early_out_enabled() { return g_opt_early_out || env_int("TA_EARLY_OUT"); }
static thread_local bool g_opt_unit_prefilter = false;
bool unit_prefilter_enabled() { return g_opt_unit_prefilter || env_int("TA_UNIT_PREFILTER"); }

static bool costs_ok(const ta_edit_costs *c) {   // EditCosts::new, src/levenshtein.rs:44-52
    if (!c) return false;
    if (!(c->mismatch_cost > 0) || !(c->gap_cost > 0)) return false;
    if (c->has_transpose) {
        if (!(c->transpose_cost > 0)) return false;
        if (!((c->transpose_cost >> 1) < c->mismatch_cost)) return false;
        if (!((c->transpose_cost >> 1) < c->gap_cost)) return false;
    }
    return true;
}

static StrView view_of(const ta_strings *s) { return StrView{s->blob, s->off, s->stride, s->len}; }

// longest string of a batch side: given, implied (strided) or measured on the device
static int side_max_len(const ta_strings *s, uint32_t n, hipStream_t st, uint64_t *out) {
    if (!s->off) { *out = s->len; return TA_OK; }
    if (s->max_len) { *out = s->max_len; return TA_OK; }
    Scratch &sc = tls_scratch(3);
    int rc = sc.ensure(16);
    if (rc) return rc;
    TA_HIP(hipMemsetAsync(sc.dev, 0, 4, st));
    TA_HIP(strings_maxlen_launch(view_of(s), n, (uint32_t *)sc.dev, st));
    uint32_t v = 0;
    TA_HIP(hipMemcpyAsync(&v, sc.dev, 4, hipMemcpyDeviceToHost, st));
    TA_HIP(hipStreamSynchronize(st));
    *out = v;
    return TA_OK;
}

// One k-bounded distance pass over the batch (or over `subset`); the heart of every distance entry point.
// columns_ordered: `subset` lists the pairs by their exact column count (length_order_launch): 64 consecutive ones share it -- what
// the VLINE form of the bit-parallel band kernel wants (one pass per wavefront)
// n_dev (optional): the length of `subset` as a kernel before this pass left it on the device -- n_work is then its upper bound (grids,
// kernel choice) and the kernels read the real count themselves (the rounds of ta_levenshtein_exp_batch: no host round trip between them)
static int lev_pass(const ta_strings *a, const ta_strings *b, uint32_t n_work, const uint32_t *subset, uint32_t k,
                    const ta_edit_costs *c, uint64_t max_len, uint32_t *out_dev, hipStream_t st, bool columns_ordered = false,
                    const uint32_t *n_dev = nullptr) {
    const uint32_t gc = c->gap_cost, sg = c->start_gap_cost;
    if (env_int("TA_FAIL_PASS")) { set_last_error_msg("TA_FAIL_PASS: a distance pass that fails (tests)"); return TA_ERR_UNSUPPORTED; }
    // unit costs times g: the unit-cost kernels with k / g, then the answers times g (lev_plan.h: lev_unit_scale)
    if (const uint32_t g = lev_unit_scale(c->mismatch_cost, gc, sg, c->has_transpose != 0, c->transpose_cost);
        g && !env_int("TA_NO_BITS") && !env_int("TA_NO_UNIT_SCALE") && !env_int("TA_FORCE_D") && !env_int("TA_FORCE_L")) {
        const ta_edit_costs uc = {1, 1, 0, (uint8_t)(c->has_transpose ? 1 : 0), (uint8_t)(c->has_transpose ? 1 : 0)};
        int rc = lev_pass(a, b, n_work, subset, k / g, &uc, max_len, out_dev, st, columns_ordered, n_dev);
        if (rc) return rc;
        TA_HIP(scale_results_launch(out_dev, subset, n_work, n_dev, g, st));
        g_answer_single_store = false;                 // two stores to the result slot: the single-call path waits for the stream
        ta_lev_select sel;
        ta_levenshtein_select((size_t)max_len, (size_t)max_len, k, c, &sel);
        g_last_launch.cell_bits = sel.cell_bits;       // (the reference's width class is that of the caller's costs and k)
        return TA_OK;
    }
    LevPlan pl = lev_make_plan(k, c->mismatch_cost, gc, sg, max_len, env_int("TA_FORCE_D"), env_int("TA_FORCE_L"), env_int("TA_FORCE_CH"));
    LevParams P;
    P.a = view_of(a); P.b = view_of(b);
    P.subset = subset; P.trace = nullptr; P.out = out_dev; P.n = n_work; P.k = k;
    P.mc = c->mismatch_cost; P.gc = gc; P.sg = sg; P.tc = c->has_transpose ? c->transpose_cost : 0;
    P.u = pl.u; P.o = pl.o;
    if (columns_ordered && subset) P.tune |= 4u;
    if (n_dev) { P.n_dev = n_dev; P.tune |= 8u; }     // (tune bit 3: the grid of the upper bound, lev_bits.hip)
    const bool affine = sg > 0 || env_int("TA_FORCE_AFFINE"), trans = c->has_transpose != 0;
    ta_launch_info li = {};
    li.band_offset = pl.o; li.affine = affine; li.transpose = trans;
    g_answer_single_store = true;                     // every branch below is one launch except the tiled row-blocked form
    ta_lev_select sel;
    ta_levenshtein_select((size_t)max_len, (size_t)max_len, k, c, &sel);
    li.cell_bits = sel.cell_bits;
    // unit-cost families (levenshtein(), rdamerau(), levenshtein_simd_k(), the exp loop) have bit-parallel kernels
    const uint32_t tcost = c->has_transpose ? c->transpose_cost : 0;
    const LevBitsPlan bp = lev_bits_make_plan(k, c->mismatch_cost, gc, sg, trans, tcost, max_len, env_int("TA_FORCE_NA"), env_int("TA_FORCE_CH"),
                                              env_int("TA_BITS_STATIC"));
    const bool dp_forced = env_int("TA_NO_BITS") || env_int("TA_FORCE_D") || env_int("TA_FORCE_L") || env_int("TA_FORCE_AFFINE") ||
                           env_int("TA_FORCE_TRANS_SELECT") || env_int("TA_FORCE_WIDE");
    // a pass of few pairs is as slow as its slowest wavefront: the chooser then compares wavefronts, not pairs (lev_plan.h);
    // the switches that pin a kernel layout keep the throughput choice
    const bool pinned = env_int("TA_NO_LATENCY_RULE") || env_int("TA_FORCE_NA") || env_int("TA_BITS_STATIC") || env_int("TA_FORCE_CH") ||
                        env_int("TA_FORCE_SLICED");
    LevChoice ch = lev_choose(k, c->mismatch_cost, gc, sg, trans, tcost, max_len, dp_forced, pinned ? 0xFFFFFFFFu : n_work);
    const bool unit = c->mismatch_cost == 1 && gc == 1 && sg == 0 && (!trans || tcost == 1);
    // a handful of long fixed-length pairs: the tiled row-blocked form uses many wavefronts per pair, whatever the band
    if (unit && !dp_forced && ch.kernel != LEV_K_BITS && (n_work == 1 || (n_work <= 16 && !subset)) && !a->off && !b->off &&
        (a->len < b->len ? a->len : b->len) > 2ull * 4096ull && !env_int("TA_WB_NO_TILES")) {
        ch.kernel = LEV_K_WIDEBITS;
        ch.rows_per_lane = 64;
    }
    if (env_int("TA_FORCE_WIDEBITS") && unit && !dp_forced) {
        ch.kernel = LEV_K_WIDEBITS;
        ch.rows_per_lane = env_int("TA_FORCE_WIDEBITS") == 64 ? 64 : 32;
    }
    // fixed-length batches with a band of 25..45 diagonals: 32 pairs per register, three band cells per lane (lev_sliced.hip);
    // opt-in -- fewer instructions than the bit-parallel band kernel but bound by the refetch of its string lines
#ifdef TA_EXPERIMENTAL
    uint32_t sl_strips = 0;
    const bool sliced = ch.kernel == LEV_K_BITS && unit && !trans && !subset && !dp_forced && env_int("TA_FORCE_SLICED") &&
                        lev_sliced_applies(P.a, P.b, bp.u, &sl_strips);
    if (sliced) {
        uint32_t grid = 0, lds = 0, ppw = 0;
        TA_HIP(lev_sliced_launch(P.a, P.b, n_work, k, bp.u, out_dev, st, &grid, &lds, &ppw));
        li.kernel = 5; li.diags_per_lane = 3; li.lanes_per_pair = sl_strips; li.pairs_per_wave = ppw;
        li.grid = grid; li.lds_bytes = lds; li.band_offset = 0;
        if (env_int("TA_DEBUG")) fprintf(stderr, "[triple_accel_amd] lev pass: n=%u k=%u u=%u kernel=5 (pair-sliced) grid=%u lds=%u\n", n_work, k, bp.u, grid, lds);
        g_last_launch = li;
        return TA_OK;                                  // (like the single-pair branch below: nothing else may write `out`)
    }
#endif
    uint32_t u_one = 0;
    if (n_work == 1 && !dp_forced && !pinned && !env_int("TA_FORCE_WIDEBITS") && !env_int("TA_NO_ONE") &&
        lev_one_applies(k, c->mismatch_cost, gc, sg, trans, tcost, max_len, &u_one)) {
        // a lone pair with a band of up to 64 diagonals: match vectors 64 columns at a time, the recurrence on the scalar unit
        P.u = u_one; P.o = 0; P.L = 64; P.PW = 1; P.lds_per_wave = 0; P.Tw = 0; P.ch = 0;
        uint32_t lds = 0;
        TA_HIP(lev_one_launch(P, trans, max_len, st, &lds));
        li.kernel = 6; li.diags_per_lane = 64; li.lanes_per_pair = 64; li.pairs_per_wave = 1;
        li.grid = 1; li.lds_bytes = lds; li.band_offset = 0;
        if (env_int("TA_DEBUG")) fprintf(stderr, "[triple_accel_amd] lev pass: n=1 k=%u u=%u kernel=6 (single pair, scalar-unit recurrence) lds=%u\n", k, u_one, lds);
        g_last_launch = li;
        return TA_OK;
    }
    const LevBits2Plan b2 = lev_bits2_make_plan(k, c->mismatch_cost, gc, sg, trans, tcost, max_len, !a->off && !b->off, n_work);
    if (ch.kernel == LEV_K_BITS && b2.ok && !pinned && !env_int("TA_NO_BITS2")) {
        // narrow band, big fixed-length batch: two pairs per lane share the recurrence AND the byte test (lev_bits2_body.h, the
        // stride-8 window with both pairs' bytes in every register): 27 % fewer instructions per pair, 16 wavefronts per CU;
        // cfg4 0.118 against 0.135 ms (profiles/r03/ab_band_kernel.md).  TA_NO_BITS2=1 keeps one pair per lane.
        P.u = b2.u; P.o = 0; P.L = 1; P.PW = 128; P.lds_per_wave = b2.lds_per_wave; P.Tw = b2.Tw; P.ch = 64;
        uint32_t grid = 0, lds = 0;
        TA_HIP(lev_bits2_launch(P, b2, trans, st, &grid, &lds));
        li.kernel = 3; li.diags_per_lane = 15; li.lanes_per_pair = 1; li.pairs_per_wave = 128;
        li.grid = grid; li.lds_bytes = lds; li.band_offset = 0;
    } else if (ch.kernel == LEV_K_BITS) {
        P.u = bp.u; P.o = 0; P.L = 1; P.PW = 64; P.lds_per_wave = bp.lds_per_wave; P.Tw = bp.Tw; P.ch = bp.ch;
        uint32_t grid = 0, lds = 0;
        TA_HIP(lev_bits_launch(P, bp, trans, max_len, st, &grid, &lds));
        li.kernel = 3; li.diags_per_lane = bp.s8 ? 33u : 4u * (uint32_t)bp.NA; li.lanes_per_pair = 1; li.pairs_per_wave = 64;
        li.grid = grid; li.lds_bytes = lds; li.band_offset = 0;
    } else if (ch.kernel == LEV_K_WIDEBITS && (n_work == 1 || (n_work <= 16 && !subset)) && !a->off && !b->off &&
               a->len <= 0xFFFFFFF0ull && b->len <= 0xFFFFFFF0ull &&
               (a->len < b->len ? a->len : b->len) > 2ull * 64ull * (uint64_t)ch.rows_per_lane &&
               (a->len > b->len ? a->len - b->len : b->len - a->len) <= bp.u && !env_int("TA_WB_NO_TILES")) {
        // ONE long pair (the single-call API) or a handful of them (fixed-length batch): a pair's stripe sweeps are cut into
        // tiles and spread over many wavefronts, pair after pair
        uint32_t launches = 0;
        g_answer_single_store = false;                 // many launches per pair; only the last tile stores the answer
        for (uint32_t p = 0; p < n_work; p++)
            TA_HIP(lev_widebits_huge_launch(a->blob + (uint64_t)p * a->stride, (uint32_t)a->len, b->blob + (uint64_t)p * b->stride,
                                            (uint32_t)b->len, bp.u, k, ch.rows_per_lane, trans, out_dev + p, st, &launches));
        li.kernel = 4; li.diags_per_lane = (uint32_t)ch.rows_per_lane; li.lanes_per_pair = 64; li.pairs_per_wave = 1;
        li.grid = launches; li.lds_bytes = 0; li.band_offset = 0;
    } else if (ch.kernel == LEV_K_WIDEBITS) {
        P.u = bp.u; P.o = 0; P.L = 64; P.PW = 1; P.lds_per_wave = 0; P.Tw = 0; P.ch = 0;
        uint32_t grid = 0, lds = 0;
        TA_HIP(lev_widebits_launch(P, ch.rows_per_lane, max_len, trans, st, &grid, &lds));
        li.kernel = 4; li.diags_per_lane = (uint32_t)ch.rows_per_lane; li.lanes_per_pair = 64; li.pairs_per_wave = 1;
        li.grid = grid; li.lds_bytes = lds; li.band_offset = 0;
    } else if (pl.ok && !env_int("TA_FORCE_WIDE")) {
        P.L = pl.L; P.PW = pl.PW; P.lds_per_wave = pl.lds_per_wave; P.Tw = pl.Tw; P.ch = pl.ch;
        uint32_t grid = 0, lds = 0;
        const int tmode = !trans ? 0 : ((2u * P.mc <= 255u + P.tc && !env_int("TA_FORCE_TRANS_SELECT")) ? 1 : 2);
        TA_HIP(lev_band_launch(P, pl, affine, tmode, st, &grid, &lds));
        li.kernel = 1; li.diags_per_lane = pl.D; li.lanes_per_pair = pl.L; li.pairs_per_wave = pl.PW;
        li.grid = grid; li.lds_bytes = lds;
    } else {
        P.L = 0; P.PW = 1; P.Tw = 0;
        P.lds_per_wave = (uint32_t)((max_len + 2 > 0xFFFFFFF0ull) ? 0xFFFFFFF0ull : max_len + 2);   // boundary line length
        uint32_t grid = 0, lds = 0, threads = 0, dpt = 0;
        TA_HIP(lev_wide_launch(P, trans, st, &grid, &lds, &threads, &dpt));
        li.kernel = 2; li.diags_per_lane = dpt; li.lanes_per_pair = threads; li.pairs_per_wave = 0;
        li.grid = grid; li.lds_bytes = lds; li.affine = 1;
    }
    if (env_int("TA_DEBUG"))
        fprintf(stderr, "[triple_accel_amd] lev pass: n=%u k=%u u=%u kernel=%u D=%u L=%u pairs/wave=%u width=%u-bit grid=%u lds=%u\n",
                n_work, k, pl.u, li.kernel, li.diags_per_lane, li.lanes_per_pair, li.pairs_per_wave, li.cell_bits, li.grid,
                li.lds_bytes);
    g_last_launch = li;
    return TA_OK;
}

// The pairs of a ragged batch in length order (util_kernels.hip: length_order_launch) into thread-local scratch; *order_out = the list.
// The histogram scratch must be zero on entry and every complete pass leaves it zero: `clean` says the last pass of this thread was
// complete -- after one that failed midway the histogram is zeroed again instead of trusted.
static int order_pairs(const ta_strings *a, const ta_strings *b, uint32_t n, uint32_t u, uint64_t max_len, bool by_steps, bool exact,
                       hipStream_t st, const uint32_t **order_out, bool *exact_columns) {
    static thread_local bool clean = false;
    Scratch &ord = tls_scratch(13), &bins = tls_scratch(14);
    constexpr size_t BINS_BYTES = 2 * 1024 * 32 * 4;                    // histogram + cursors: 1024 bins x 32 counters each
    const bool fresh = bins.cap < BINS_BYTES || !clean;
    int rc;
    if ((rc = ord.ensure((size_t)n * 4)) || (rc = bins.ensure(BINS_BYTES))) return rc;
    if (fresh) TA_HIP(hipMemsetAsync(bins.dev, 0, BINS_BYTES / 2, st));
    clean = false;
    TA_HIP(length_order_launch(view_of(a), view_of(b), nullptr, n, u, max_len, by_steps, (uint32_t *)bins.dev, (uint32_t *)ord.dev, st, exact, exact_columns));
    clean = true;
    *order_out = (const uint32_t *)ord.dev;
    return TA_OK;
}

static int batch_max_len(const ta_strings *a, const ta_strings *b, uint32_t n, hipStream_t st, uint64_t *out) {
    uint64_t ma = 0, mb = 0;
    int rc = side_max_len(a, n, st, &ma);
    if (rc) return rc;
    rc = side_max_len(b, n, st, &mb);
    if (rc) return rc;
    *out = ma > mb ? ma : mb;
    return TA_OK;
}

static int check_batch_args(const ta_strings *a, const ta_strings *b, size_t n, const void *out) {
    if (!a || !b || (!out && n) || n > 0xFFFFFFF0ull) { set_last_error_msg("bad batch arguments"); return TA_ERR_ARG; }
    if (n && (!a->blob || !b->blob)) { set_last_error_msg("null blob"); return TA_ERR_ARG; }
    return TA_OK;
}

}  // namespace ta

using namespace ta;

extern "C" {

#ifdef TA_EXPERIMENTAL
const char *ta_version(void) { return "triple_accel_amd 0.3 (gfx950) +experimental"; }
#else
const char *ta_version(void) { return "triple_accel_amd 0.3 (gfx950)"; }
#endif

const char *ta_status_str(int s) {
    switch (s) {
        case TA_OK: return "ok";
        case TA_ERR_LEN_MISMATCH: return "length mismatch (reference: assert!(a.len() == b.len()))";
        case TA_ERR_NULL_BYTE: return "No zero/null bytes allowed in the string!";
        case TA_ERR_BAD_COSTS: return "invalid EditCosts";
        case TA_ERR_HIP: return "HIP error / no device";
        case TA_ERR_ARG: return "bad argument";
        case TA_ERR_UNSUPPORTED: return "unsupported on the GPU path";
        case TA_ERR_CAPACITY: return "output capacity exceeded";
        case TA_ERR_DIV_ZERO: return "attempt to divide by zero";
        default: return "unknown";
    }
}

int ta_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *ta_last_error(void) { return g_last_error.c_str(); }
const char *ta_last_kernel_name(void) { return g_last_kernel_name; }
int ta_set_option(int option, int value) {
    if (option == TA_OPT_EARLY_OUT) { g_opt_early_out = value != 0; return TA_OK; }
    if (option == TA_OPT_UNIT_PREFILTER) { g_opt_unit_prefilter = value != 0; return TA_OK; }
    set_last_error_msg("unknown option");
    return TA_ERR_ARG;
}

ta_edit_costs ta_levenshtein_costs(void) { ta_edit_costs c = {1, 1, 0, 0, 0}; return c; }
ta_edit_costs ta_rdamerau_costs(void) { ta_edit_costs c = {1, 1, 0, 1, 1}; return c; }

int ta_edit_costs_new(uint8_t mismatch, uint8_t gap, uint8_t start_gap, int has_transpose, uint8_t transpose,
                      ta_edit_costs *out) {
    ta_edit_costs c = {mismatch, gap, start_gap, (uint8_t)(has_transpose ? 1 : 0), (uint8_t)(has_transpose ? transpose : 0)};
    if (!costs_ok(&c)) return TA_ERR_BAD_COSTS;
    if (out) *out = c;
    return TA_OK;
}

int ta_edit_costs_check_search(const ta_edit_costs *c) {   // src/levenshtein.rs:67-71
    if (!c) return TA_ERR_ARG;
    if (c->has_transpose && !((uint32_t)c->transpose_cost <= (uint32_t)c->start_gap_cost + (uint32_t)c->gap_cost))
        return TA_ERR_BAD_COSTS;
    return TA_OK;
}

// src/levenshtein.rs:731-791
int ta_levenshtein_select(size_t a_len, size_t b_len, uint32_t k, const ta_edit_costs *c, ta_lev_select *out) {
    if (!c || !out) return TA_ERR_ARG;
    if (!costs_ok(c)) return TA_ERR_BAD_COSTS;
    const uint32_t mn = (uint32_t)(a_len < b_len ? a_len : b_len), mx = (uint32_t)(a_len < b_len ? b_len : a_len);
    const uint32_t mc = c->mismatch_cost, gc = c->gap_cost, sg = c->start_gap_cost;
    uint32_t sub_all = mn * mc;
    uint32_t gaps_all = (mn << 1) * gc + (mn == 0 ? 0u : sg + (mx == mn ? sg : 0u));
    uint32_t bound = (sub_all < gaps_all ? sub_all : gaps_all) + (mx - mn) * gc + (mx == mn ? 0u : sg);
    uint32_t max_k = k < bound ? k : bound;
    uint32_t unit_k = lev_sat_sub(max_k, sg) / gc;
    if (unit_k > mx) unit_k = mx;
    out->max_k = max_k; out->unit_k = unit_k;
    out->cell_bits = 32; out->ref_lanes = 0;
    static const uint32_t ub[4] = {32, 64, 128, 256};
    if (max_k <= 254u) {
        for (int t = 0; t < 4; t++)
            if (unit_k <= ub[t] - 2) { out->cell_bits = 8; out->ref_lanes = ub[t]; return TA_OK; }
    }
    if (max_k <= 65534u) out->cell_bits = 16;
    return TA_OK;
}

int ta_last_launch_info(ta_launch_info *out) {
    if (!out) return TA_ERR_ARG;
    *out = g_last_launch;
    return TA_OK;
}

void ta_free(void *p) { free(p); }

/* Frees what the calling thread holds inside the library: its device scratch, its stream and its pinned staging buffer.
 * Optional -- a thread that exits without it leaves them to process exit (nothing is freed from a thread-exit hook: the HIP
 * runtime may already be gone by then). */
void ta_thread_release(void) {
    (void)hipDeviceSynchronize();
    for (int i = 0; i < TA_SCRATCH_SLOTS; i++) tls_scratch(i).release();
    call_ctx().release();
    search_report_box().release();
    search_resident_reset();
    LastUse &u = last_use();                     // the event that orders this thread's calls across streams
    if (u.ev) (void)hipEventDestroy(u.ev);
    u = LastUse{};
}

/* ---------------------------------------------------------------- batch API */

int ta_levenshtein_k_batch(const ta_strings *a, const ta_strings *b, size_t n, uint32_t k,
                           const ta_edit_costs *costs, uint32_t *out_dev, void *stream) {
    int rc = check_batch_args(a, b, n, out_dev);
    if (rc) return rc;
    if (!costs_ok(costs)) return TA_ERR_BAD_COSTS;
    if (!device_ready()) return TA_ERR_HIP;
    if (n == 0) return TA_OK;
    hipStream_t st = (hipStream_t)stream;
    StreamGuard guard(st);
    uint64_t max_len = 0;
    rc = batch_max_len(a, b, (uint32_t)n, st, &max_len);
    if (rc) return rc;
    // Ragged (CSR) batches: the band kernels run a wavefront to the longest of its 64 / L pairs, so the pairs are taken in the
    // order of a counting sort on their length class (util_kernels.hip: three small launches, ~20 us per million pairs) --
    // every wavefront then sees pairs within 8 bytes of each other.  TA_NO_LENGTH_ORDER=1 keeps the batch order.
    const uint32_t *order = nullptr;
    bool exact_columns = false;
    if ((a->off || b->off) && n >= 4096 && max_len >= 16 && !env_int("TA_NO_LENGTH_ORDER")) {
        const uint32_t u = lev_batch_unit_k(k, costs->mismatch_cost, costs->gap_cost, costs->start_gap_cost, max_len);
        // the key counts what the kernel of this pass iterates over: columns (bit-parallel kernels) or anti-diagonal steps (DP band kernel).
        // (the kernels whose time follows max(len_a, len_b) or len_a -- two pairs per lane, the single-pair kernel, the row-blocked
        // one -- never see a length-ordered CSR wavefront of many pairs: the first takes fixed-length batches only, the other two run ONE pair
        // per wavefront, where the order only shapes the launch's tail; len_b is the right key for everything that is ordered by columns.)
        const bool unit = (costs->mismatch_cost == 1 && costs->gap_cost == 1 && costs->start_gap_cost == 0 && (!costs->has_transpose || costs->transpose_cost == 1)) ||
                          lev_unit_scale(costs->mismatch_cost, costs->gap_cost, costs->start_gap_cost, costs->has_transpose != 0, costs->transpose_cost);
        const bool by_steps = !unit || env_int("TA_NO_BITS") || env_int("TA_FORCE_D") || env_int("TA_FORCE_L");
        if ((rc = order_pairs(a, b, (uint32_t)n, u, max_len, by_steps, env_int("TA_BITS_VLINE") != 0, st, &order, &exact_columns))) return rc;
    }
    // TA_OPT_UNIT_PREFILTER (an option, off by default: the work then depends on the data): a weighted batch first runs the UNIT-cost
    // bit-parallel band pass with k' = lev_unit_filter_k -- "unit distance > k'" implies "weighted distance > k", i.e. None
    // (src/levenshtein.rs:539-541) -- and the DP band kernel prices only the pairs that pass answered (a list whose length stays on the
    // device).  Dissimilar batches cost the unit pass alone; a batch of near pairs pays for both.
    if (unit_prefilter_enabled() && n >= 1024 && !env_int("TA_NO_BITS") && !env_int("TA_FORCE_D") && !env_int("TA_FORCE_L")) {
        const bool trans = costs->has_transpose != 0;
        const bool unit_family = (costs->mismatch_cost == 1 && costs->gap_cost == 1 && costs->start_gap_cost == 0 && (!trans || costs->transpose_cost == 1)) ||
                                 lev_unit_scale(costs->mismatch_cost, costs->gap_cost, costs->start_gap_cost, trans, costs->transpose_cost);
        const uint32_t kf = lev_unit_filter_k(k, costs->mismatch_cost, costs->gap_cost, costs->start_gap_cost, trans, costs->transpose_cost);
        if (!unit_family && kf != 0xFFFFFFFFu && lev_choose(kf, 1, 1, 0, trans, trans ? 1u : 0u, max_len, false, (uint32_t)n).kernel == LEV_K_BITS) {
            const ta_edit_costs uc = {1, 1, 0, (uint8_t)(trans ? 1 : 0), (uint8_t)(trans ? 1 : 0)};
            Scratch &lst = tls_scratch(4), &cnt = tls_scratch(3);
            if ((rc = lst.ensure(n * 4)) || (rc = cnt.ensure(16))) return rc;
            if ((rc = lev_pass(a, b, (uint32_t)n, order, kf, &uc, max_len, out_dev, st, false))) return rc;
            TA_HIP(fill_u32_launch((uint32_t *)cnt.dev, 0u, 1, st));
            TA_HIP(compact_some_launch(out_dev, order, (uint32_t)n, (uint32_t *)lst.dev, (uint32_t *)cnt.dev, st));
            rc = lev_pass(a, b, (uint32_t)n, (const uint32_t *)lst.dev, k, costs, max_len, out_dev, st, false, (const uint32_t *)cnt.dev);
            g_answer_single_store = false;
            return rc;
        }
    }
    return lev_pass(a, b, (uint32_t)n, order, k, costs, max_len, out_dev, st, exact_columns);
}

/* ta_levenshtein_k_batch for strings written in a SMALL ALPHABET the caller names (up to four byte values -- DNA, RNA: lev_bitsq_body.h;
 * up to 32 -- IUPAC codes, amino acids: lev_bitsqw_body.h): the column's match vector is then a table lookup instead of a byte test --
 * the same answers, bit for bit.
 * The promise is verified on the device, byte by byte: a pair that holds any other byte is answered by the general kernel in the
 * same call (a second, usually empty, launch over the list of such pairs; no host round trip).  Batches the small-alphabet kernel
 * does not cover (CSR batches, general EditCosts, bands beyond 33 diagonals, fewer than 16384 pairs, alphabets of more than 32
 * symbols or without a code: two bits for up to four symbols, else five bits with the byte's other three bits the same in every
 * symbol -- one case of the letters, the digits) run ta_levenshtein_k_batch as they are. */
int ta_levenshtein_k_batch_alphabet(const ta_strings *a, const ta_strings *b, size_t n, uint32_t k, const ta_edit_costs *costs,
                                    const uint8_t *alphabet, size_t alphabet_len, uint32_t *out_dev, void *stream) {
    int rc = check_batch_args(a, b, n, out_dev);
    if (rc) return rc;
    if (!costs_ok(costs)) return TA_ERR_BAD_COSTS;
    if (!device_ready()) return TA_ERR_HIP;
    if (n == 0) return TA_OK;
    const bool fixed = !a->off && !b->off;
    const bool trans = costs->has_transpose != 0;
    const uint64_t max_len = fixed ? (a->len > b->len ? a->len : b->len) : 0;
    uint32_t u = 0, q_shift = 0, q_table = 0, q_memb = 0, q_hi = 0;
    const bool pinned = env_int("TA_NO_BITS") || env_int("TA_FORCE_NA") || env_int("TA_BITS_STATIC") || env_int("TA_FORCE_D") || env_int("TA_NO_BITSQ");
    if (!fixed || pinned || !alphabet ||
        !lev_bitsq_applies(k, costs->mismatch_cost, costs->gap_cost, costs->start_gap_cost, trans, trans ? costs->transpose_cost : 0, max_len, true, n, &u) ||
        alphabet_len == 0)
        return ta_levenshtein_k_batch(a, b, n, k, costs, out_dev, stream);
    // at most four symbols with a two-bit code: lev_bitsq_body.h; up to 32 with a five-bit code: lev_bitsqw_body.h
    const bool force_wide = env_int("TA_BITSQ_WIDE") != 0;
    const bool narrow = !force_wide && lev_bitsq_hash(alphabet, alphabet_len, &q_shift, &q_table);
    if (!narrow) {
        if (env_int("TA_NO_BITSQW") || !lev_bitsqw_hash(alphabet, alphabet_len, &q_shift, &q_memb, &q_hi))
            return ta_levenshtein_k_batch(a, b, n, k, costs, out_dev, stream);
        // Where the 5-bit-code kernel pays (measured, profiles/r04/ab_alphabet.md): its per-16-rows conversion grows with the groups of
        // four codes that hold a symbol, and narrow bands have the two-pairs-per-lane byte test.  Beyond 4 groups (IUPAC's 16 letters
        // and the 20 amino acids: 7) or at 15 diagonals and fewer the byte-test kernels are as fast or faster: the general path runs.
        uint32_t groups = 0;
        for (uint32_t g4 = 0; g4 < 8; g4++) groups += ((q_memb >> (4u * g4)) & 15u) ? 1u : 0u;
        if (!force_wide && (groups > 4 || (uint64_t)u + 1u + (trans ? 2u : 0u) <= 15u))
            return ta_levenshtein_k_batch(a, b, n, k, costs, out_dev, stream);
    }
    hipStream_t st = (hipStream_t)stream;
    StreamGuard guard(st);
    Scratch &bad = tls_scratch(15), &cnt = tls_scratch(16);
    // two counters taken in turn: this pass appends to one and zeroes the other for the next pass (no fill per call; both are
    // zeroed once, when the scratch is allocated)
    // (`clean`: the last pass of this thread ran to its last launch -- the counter it left behind is zero.  A pass that failed midway leaves
    // it unknown: both counters are zeroed again and the turn restarts.)
    static thread_local uint32_t turn = 0;
    static thread_local bool clean = false;
    const bool fresh = cnt.cap < 64 || !clean;
    if ((rc = bad.ensure(n * 4)) || (rc = cnt.ensure(64))) return rc;
    if (fresh) { TA_HIP(hipMemsetAsync(cnt.dev, 0, 64, st)); turn = 0; }
    clean = false;
    uint32_t *counters = (uint32_t *)cnt.dev;
    const uint32_t mine = turn & 1u;
    turn++;
    LevParams P;
    P.a = view_of(a); P.b = view_of(b);
    P.subset = nullptr; P.trace = nullptr; P.out = out_dev; P.n = (uint32_t)n; P.k = k;
    P.mc = 1; P.gc = 1; P.sg = 0; P.tc = trans ? 1 : 0;
    P.u = u; P.o = 0; P.L = 1; P.PW = 64; P.lds_per_wave = 0; P.Tw = 0; P.ch = 0;
    P.q_table = q_table; P.q_shift = q_shift; P.q_memb = q_memb; P.q_hi = q_hi; P.q_ns = (uint32_t)alphabet_len; P.q_bad_count = counters + 8u * mine; P.q_bad_list = (uint32_t *)bad.dev;
    P.q_next_count = counters + 8u * (mine ^ 1u);
    ta_launch_info li = {};
    li.transpose = trans;
    ta_lev_select sel;
    ta_levenshtein_select((size_t)max_len, (size_t)max_len, k, costs, &sel);
    li.cell_bits = sel.cell_bits;
    uint32_t grid = 0, lds = 0;
    if (narrow) TA_HIP(lev_bitsq_launch(P, trans, st, &grid, &lds));
    else TA_HIP(lev_bitsqw_launch(P, trans, st, &grid, &lds));
    li.kernel = 7; li.diags_per_lane = 33; li.lanes_per_pair = 1; li.pairs_per_wave = 64; li.grid = grid; li.lds_bytes = lds;
    // the pairs that hold a byte outside the alphabet: the byte-test kernel over the list the first kernel wrote (its length is read
    // on the device; a small grid strides over it)
    const LevBitsPlan bp = lev_bits_make_plan(k, 1, 1, 0, trans, trans ? 1u : 0u, max_len, 0, 0, 0);
    LevParams F = P;
    F.subset = (const uint32_t *)bad.dev; F.n_dev = counters + 8u * mine;
    F.q_bad_count = nullptr; F.q_bad_list = nullptr; F.q_next_count = nullptr;
    F.u = bp.u; F.lds_per_wave = bp.lds_per_wave; F.Tw = bp.Tw; F.ch = bp.ch;
    char name[96];
    snprintf(name, sizeof(name), "%s", ta_last_kernel_name());
    TA_HIP(lev_bits_launch(F, bp, trans, max_len, st, nullptr, nullptr));
    set_last_kernel_name("%s", name);                   // the pass's dominant kernel is the first one
    if (env_int("TA_DEBUG")) fprintf(stderr, "[triple_accel_amd] lev pass: n=%zu k=%u u=%u kernel=7 (small alphabet, %s, shift %u table %08x) grid=%u lds=%u\n", n, k, u, narrow ? "2-bit codes" : "5-bit codes", q_shift, narrow ? q_table : q_memb, grid, lds);
    g_last_launch = li;
    g_answer_single_store = false;
    clean = true;
    return TA_OK;
}

int ta_levenshtein_exp_batch(const ta_strings *a, const ta_strings *b, size_t n,
                             const ta_edit_costs *costs, uint32_t *out_dev, void *stream) {
    int rc = check_batch_args(a, b, n, out_dev);
    if (rc) return rc;
    if (!costs_ok(costs)) return TA_ERR_BAD_COSTS;
    if (!device_ready()) return TA_ERR_HIP;
    if (n == 0) return TA_OK;
    hipStream_t st = (hipStream_t)stream;
    StreamGuard guard(st);
    uint64_t max_len = 0;
    rc = batch_max_len(a, b, (uint32_t)n, st, &max_len);
    if (rc) return rc;
    // subset ping-pong buffers + counter
    Scratch &s0 = tls_scratch(4), &s1 = tls_scratch(5), &cnt = tls_scratch(3), &bnd = tls_scratch(10), &wrk = tls_scratch(11);
    if ((rc = s0.ensure(n * 4)) || (rc = s1.ensure(n * 4)) || (rc = cnt.ensure(16))) return rc;
    // Bag lower bound per pair (util_kernels.hip): a doubling round only takes the unresolved pairs whose bound admits its
    // threshold -- the others could only return None from it.  Same return values, fewer rounds for dissimilar strings.
    const bool bounded = n >= 64 && max_len >= 64 && !env_int("TA_EXP_NO_BOUND") && !env_int("TA_EXP_FAITHFUL");
    if (bounded) {
        if ((rc = bnd.ensure(n * 4)) || (rc = wrk.ensure(n * 4))) return rc;
        TA_HIP(fill_u32_launch(out_dev, 0xFFFFFFFFu, (uint32_t)n, st));
        TA_HIP(bag_bound_launch(view_of(a), view_of(b), (uint32_t)n, costs->mismatch_cost, costs->gap_cost, (uint32_t *)bnd.dev, st));
    }
    g_exp_passes = 0;
    uint32_t *sub_in = nullptr, *bufs[2] = {(uint32_t *)s0.dev, (uint32_t *)s1.dev};
    uint32_t n_left = (uint32_t)n;                          // unresolved pairs (sub_in == nullptr: all of them)
    uint32_t k = 30;                                        // src/levenshtein.rs:1446, 1486, 1517
    // The FIRST threshold as wide as the cheapest kernel's window: the stride-8 form holds 33 diagonals whatever k <= 32 asks for (31 with the
    // transposition term's two extra ones: k = 30), so unit costs start at k = 32 (g x unit costs: 32 g) for the price of 30 -- a pair at distance
    // 31 or 32 is answered by the first round instead of the 61-diagonal one (cfg3 on similar strings: 32 substitutions per pair; 2.29 -> 0.9 ms per
    // 100K pairs).  levenshtein_exp returns the distance: any schedule of thresholds finds the same value (:1445-1454); from the second round on
    // the reference's 60, 120, ... stand.  TA_EXP_FAITHFUL=1 keeps 30.
    uint32_t k_first = 30;
    if (!env_int("TA_EXP_FAITHFUL") && !costs->has_transpose && costs->start_gap_cost == 0 && costs->mismatch_cost == costs->gap_cost)
        k_first = 32u * costs->gap_cost;                    // (gap_cost <= 255: no overflow)
    k = k_first;
    // ragged (CSR) batches: the rounds take their pairs in length order, as ta_levenshtein_k_batch does (the list of the still
    // unresolved pairs is compacted from the ordered one, which keeps it ordered block by block)
    if ((a->off || b->off) && n >= 4096 && max_len >= 16 && !env_int("TA_NO_LENGTH_ORDER")) {
        const uint32_t u0 = lev_batch_unit_k(k, costs->mismatch_cost, costs->gap_cost, costs->start_gap_cost, max_len);
        const bool unit = (costs->mismatch_cost == 1 && costs->gap_cost == 1 && costs->start_gap_cost == 0 && (!costs->has_transpose || costs->transpose_cost == 1)) ||
                          lev_unit_scale(costs->mismatch_cost, costs->gap_cost, costs->start_gap_cost, costs->has_transpose != 0, costs->transpose_cost);
        const uint32_t *ordered = nullptr;
        if ((rc = order_pairs(a, b, (uint32_t)n, u0, max_len, !unit || env_int("TA_NO_BITS"), false, st, &ordered, nullptr))) return rc;
        sub_in = (uint32_t *)ordered;
    }
    int flip = 0;
    const uint32_t tcx = costs->has_transpose ? costs->transpose_cost : 0;
    const bool dpo = env_int("TA_NO_BITS") != 0, faithful = env_int("TA_EXP_FAITHFUL") != 0;
    // Big batches: the rounds are DEVICE-DRIVEN -- every list (the pairs a round works on, the pairs it leaves unresolved) carries its
    // length in a device counter that the next kernel reads, every launch is sized for the list's upper bound (the batch) and the
    // wavefronts behind its end leave at once: the whole k schedule is enqueued without a host round trip between rounds, the call
    // returns while the device works (and can be captured in a hipGraph).  The schedule itself is the host-driven one's for a list that
    // never shrinks: k = 30, 60, ... until a bounded pass would cost more than a quarter of the unbounded one, then k = u32::MAX -- a
    // round whose list is empty costs its (empty) launches, 10-15 us.  Small batches (a lone pair's first round is the unbounded one;
    // the kernel choice follows the number of pairs left) keep the host-driven loop below.  TA_EXP_HOST_ROUNDS=1 pins it.
    if (n >= 1024 && !faithful && !env_int("TA_EXP_HOST_ROUNDS")) {
        constexpr int MAX_ROUNDS = 40;
        if ((rc = cnt.ensure(2 * MAX_ROUNDS * 4))) return rc;
        TA_HIP(fill_u32_launch((uint32_t *)cnt.dev, 0u, 2 * MAX_ROUNDS, st));
        uint32_t *counters = (uint32_t *)cnt.dev;
        const uint32_t *n_in_dev = nullptr;                 // the length of sub_in (nullptr: n, exactly)
        for (int round = 0; round < MAX_ROUNDS; round++) {
            if (!faithful && k != 0xFFFFFFFFu) {
                const double c_this = lev_choose(k, costs->mismatch_cost, costs->gap_cost, costs->start_gap_cost, costs->has_transpose != 0, tcx, max_len, dpo, (uint32_t)n).cost;
                const double c_full = lev_choose(0xFFFFFFFFu, costs->mismatch_cost, costs->gap_cost, costs->start_gap_cost, costs->has_transpose != 0, tcx, max_len, dpo, (uint32_t)n).cost;
                if (c_this > 0.25 * c_full) k = 0xFFFFFFFFu;
            }
            const uint32_t *work = sub_in, *work_n_dev = n_in_dev;
            if (bounded && k != 0xFFFFFFFFu) {
                TA_HIP(compact_bound_launch(out_dev, (const uint32_t *)bnd.dev, k, sub_in, (uint32_t)n, n_in_dev, (uint32_t *)wrk.dev, counters + 2 * round, st));
                work = (const uint32_t *)wrk.dev;
                work_n_dev = counters + 2 * round;
            }
            rc = lev_pass(a, b, (uint32_t)n, work, k, costs, max_len, out_dev, st, false, work_n_dev);
            g_exp_passes++;
            if (rc) return rc;
            if (k == 0xFFFFFFFFu) break;                    // the unbounded pass answers every pair it is given (all that were left)
            TA_HIP(compact_none_launch(out_dev, sub_in, (uint32_t)n, n_in_dev, bufs[flip], counters + 2 * round + 1, st));
            sub_in = bufs[flip];
            n_in_dev = counters + 2 * round + 1;
            flip ^= 1;
            k = (k > 0x7FFFFFFFu) ? 0xFFFFFFFFu : (round == 0 && k == k_first && k_first != 30u ? 60u * costs->gap_cost : k * 2);    // k *= 2 (:1452); saturate instead of wrapping
        }
        g_answer_single_store = false;
        return TA_OK;
    }
    for (int round = 0; round < 40 && n_left > 0; round++) {
        // Once a bounded pass would cost more than a quarter of the unbounded one (kernel cost model, lev_plan.h: per pair for
        // big passes, per wavefront for small ones), it is cheaper in expectation to finish the unresolved pairs with
        // k = u32::MAX right away (same return values -- levenshtein_exp returns the distance, whatever k schedule finds it).
        if (!faithful && k != 0xFFFFFFFFu) {
            const double c_this = lev_choose(k, costs->mismatch_cost, costs->gap_cost, costs->start_gap_cost, costs->has_transpose != 0, tcx, max_len, dpo, n_left).cost;
            const double c_full = lev_choose(0xFFFFFFFFu, costs->mismatch_cost, costs->gap_cost, costs->start_gap_cost, costs->has_transpose != 0, tcx, max_len, dpo, n_left).cost;
            if (c_this > 0.25 * c_full) k = 0xFFFFFFFFu;
        }
        const uint32_t *work = sub_in;
        uint32_t n_work = n_left;
        if (bounded && k != 0xFFFFFFFFu) {
            TA_HIP(hipMemsetAsync(cnt.dev, 0, 4, st));
            TA_HIP(compact_bound_launch(out_dev, (const uint32_t *)bnd.dev, k, sub_in, n_left, nullptr, (uint32_t *)wrk.dev, (uint32_t *)cnt.dev, st));
            TA_HIP(hipMemcpyAsync(&n_work, cnt.dev, 4, hipMemcpyDeviceToHost, st));
            TA_HIP(hipStreamSynchronize(st));
            work = (const uint32_t *)wrk.dev;
        }
        if (n_work > 0) {
            // (the compacted lists keep the order block by block only -- a wavefront may straddle two blocks' shares: the chunk form,
            // which takes any mix of lengths in one pass, stays the rounds' kernel)
            rc = lev_pass(a, b, n_work, work, k, costs, max_len, out_dev, st);
            g_exp_passes++;
            if (rc) return rc;
            if (k == 0xFFFFFFFFu) break;                    // the unbounded pass answers every pair it is given (all that were left)
            TA_HIP(hipMemsetAsync(cnt.dev, 0, 4, st));
            TA_HIP(compact_none_launch(out_dev, sub_in, n_left, nullptr, bufs[flip], (uint32_t *)cnt.dev, st));
            uint32_t left = 0;
            TA_HIP(hipMemcpyAsync(&left, cnt.dev, 4, hipMemcpyDeviceToHost, st));
            TA_HIP(hipStreamSynchronize(st));
            sub_in = bufs[flip];
            flip ^= 1;
            n_left = left;
        }
        k = (k > 0x7FFFFFFFu) ? 0xFFFFFFFFu : (round == 0 && k == k_first && k_first != 30u ? 60u * costs->gap_cost : k * 2);        // k *= 2 (:1452); saturate instead of wrapping
    }
    return TA_OK;
}

int ta_hamming_batch(const ta_strings *a, const ta_strings *b, size_t n, uint32_t *out_dev, void *stream) {
    int rc = check_batch_args(a, b, n, out_dev);
    if (rc) return rc;
    if (!device_ready()) return TA_ERR_HIP;
    if (n == 0) return TA_OK;
    StreamGuard guard((hipStream_t)stream);
    TA_HIP(hamming_batch_launch(view_of(a), view_of(b), (uint32_t)n, out_dev, (hipStream_t)stream));
    return TA_OK;
}

/* ---------------------------------------------------------------- queue: pairs pushed one at a time, answered by one batch pass */
}  // extern "C"

struct ta_queue {
    uint32_t k;
    ta_edit_costs costs;
    std::vector<uint8_t> blob[2];          // the pushed strings back to back (a: 0, b: 1)
    std::vector<uint64_t> off[2];          // CSR offsets (n + 1)
    uint64_t max_len = 0;
    std::vector<uint32_t> results;
    void *dev = nullptr;                   // device staging: [a blob | b blob | a offsets | b offsets | results]
    size_t dev_cap = 0;
    hipStream_t st = nullptr;
};

extern "C" {

int ta_queue_create(uint32_t k, const ta_edit_costs *costs, ta_queue **out) {
    if (!out) return TA_ERR_ARG;
    *out = nullptr;
    if (!costs_ok(costs)) return TA_ERR_BAD_COSTS;
    if (!device_ready()) return TA_ERR_HIP;
    ta_queue *q = new ta_queue();
    q->k = k; q->costs = *costs;
    q->off[0].push_back(0); q->off[1].push_back(0);
    if (hipStreamCreateWithFlags(&q->st, hipStreamNonBlocking) != hipSuccess) { delete q; return TA_ERR_HIP; }
    *out = q;
    return TA_OK;
}

int ta_queue_push(ta_queue *q, const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len, size_t *ticket) {
    if (!q || (!a && a_len) || (!b && b_len) || a_len > 0xFFFFFFF0ull || b_len > 0xFFFFFFF0ull) return TA_ERR_ARG;
    if (q->off[0].size() > 0xFFFFFFF0ull) return TA_ERR_CAPACITY;
    if (ticket) *ticket = q->off[0].size() - 1;
    q->blob[0].insert(q->blob[0].end(), a, a + a_len);
    q->blob[1].insert(q->blob[1].end(), b, b + b_len);
    q->off[0].push_back(q->blob[0].size());
    q->off[1].push_back(q->blob[1].size());
    const uint64_t m = a_len > b_len ? a_len : b_len;
    if (m > q->max_len) q->max_len = m;
    return TA_OK;
}

// A flush that fails (an over-wide band, an allocation failure) DROPS the queued pairs: the queue is empty afterwards and accepts new
// pairs -- a queue that kept them would fail the same way on every later flush.
static int queue_flush_impl(ta_queue *q, const uint32_t **results, size_t n);
int ta_queue_flush(ta_queue *q, const uint32_t **results, size_t *n_out) {
    if (!q || !results || !n_out) return TA_ERR_ARG;
    const size_t n = q->off[0].size() - 1;
    *results = nullptr; *n_out = n;
    q->results.assign(n, 0);
    if (n == 0) return TA_OK;
    const int rc = queue_flush_impl(q, results, n);
    for (int s = 0; s < 2; s++) { q->blob[s].clear(); q->off[s].assign(1, 0); }
    q->max_len = 0;
    if (rc) { *n_out = 0; *results = nullptr; }
    return rc;
}
static int queue_flush_impl(ta_queue *q, const uint32_t **results, size_t n) {
    // more than one device in the set and enough pairs for more than one of them: contiguous slices of the queue, one per device, each
    // uploaded through its worker's pinned ring and answered on its own GPU (ta_multi.hip)
    if (multi_pair_shards(n) > 1) {
        const ta_strings A = {q->blob[0].data(), q->off[0].data(), 0, 0, 0}, B = {q->blob[1].data(), q->off[1].data(), 0, 0, 0};
        int rc = ta_levenshtein_k_batch_host(&A, &B, n, q->k, &q->costs, q->results.data());
        if (rc) return rc;
        *results = q->results.data();
        return TA_OK;
    }
    auto pad = [](size_t x) { return (x + TA_BLOB_SLACK + 255) & ~(size_t)255; };
    const size_t sa = pad(q->blob[0].size()), sb = pad(q->blob[1].size()), so = pad((n + 1) * 8), sr = pad(n * 4);
    const size_t need = sa + sb + 2 * so + sr;
    if (need > q->dev_cap) {
        if (q->dev) (void)hipFree(q->dev);
        q->dev = nullptr; q->dev_cap = 0;
        TA_HIP(hipMalloc(&q->dev, need + need / 2));
        q->dev_cap = need + need / 2;
    }
    uint8_t *base = (uint8_t *)q->dev;
    uint8_t *da = base, *db = base + sa, *doa = base + sa + sb, *dob = doa + so, *dr = dob + so;
    hipStream_t st = q->st;
    if (!q->blob[0].empty()) TA_HIP(hipMemcpyAsync(da, q->blob[0].data(), q->blob[0].size(), hipMemcpyHostToDevice, st));
    if (!q->blob[1].empty()) TA_HIP(hipMemcpyAsync(db, q->blob[1].data(), q->blob[1].size(), hipMemcpyHostToDevice, st));
    TA_HIP(hipMemcpyAsync(doa, q->off[0].data(), (n + 1) * 8, hipMemcpyHostToDevice, st));
    TA_HIP(hipMemcpyAsync(dob, q->off[1].data(), (n + 1) * 8, hipMemcpyHostToDevice, st));
    ta_strings A = {da, (const uint64_t *)doa, 0, 0, q->max_len ? q->max_len : 1}, B = {db, (const uint64_t *)dob, 0, 0, q->max_len ? q->max_len : 1};
    int rc = ta_levenshtein_k_batch(&A, &B, n, q->k, &q->costs, (uint32_t *)dr, st);
    if (rc) return rc;
    TA_HIP(hipMemcpyAsync(q->results.data(), dr, n * 4, hipMemcpyDeviceToHost, st));
    TA_HIP(hipStreamSynchronize(st));
    *results = q->results.data();
    return TA_OK;
}

void ta_queue_destroy(ta_queue *q) {
    if (!q) return;
    if (q->st) { (void)hipStreamSynchronize(q->st); (void)hipStreamDestroy(q->st); }
    if (q->dev) (void)hipFree(q->dev);
    delete q;
}

/* ---------------------------------------------------------------- single-call host API */

// One (a, b) pair of a single call.  Short pairs go into the thread's pinned, device-mapped buffer: the kernel reads the
// strings in place and writes its answer next to them (`*out_host` then reads it after one stream synchronisation).  Long
// pairs are copied to thread-local device scratch on the thread's stream.  Either way the call runs on the thread's own
// non-blocking stream `*st`.
constexpr uint32_t TA_SLOT_EMPTY = 0xFFFFFFFEu;    // "no answer yet": distances stay below 0xFFFFFFF0, None is 0xFFFFFFFF
struct Staged {
    ta_strings sa, sb;
    uint32_t *out_dev;             // where the kernel writes
    volatile uint32_t *out_host;   // host view of out_dev (pinned case) or nullptr
    hipStream_t st;
};
static int stage_pair(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len, Staged *S) {
    if ((!a && a_len) || (!b && b_len)) return TA_ERR_ARG;
    if (a_len > 0xFFFFFFF0ull || b_len > 0xFFFFFFF0ull) return TA_ERR_ARG;
    if (!device_ready()) return TA_ERR_HIP;
    CallCtx &cx = call_ctx();
    int rc = cx.ensure();
    if (rc) return rc;
    S->st = cx.st;
    const size_t a_pad = (a_len + TA_BLOB_SLACK + 255) & ~(size_t)255, b_pad = (b_len + TA_BLOB_SLACK + 255) & ~(size_t)255;
    if (a_pad + b_pad <= CallCtx::RESULT_OFF) {
        if (a_len) memcpy(cx.pin, a, a_len);
        if (b_len) memcpy(cx.pin + a_pad, b, b_len);
        S->sa = ta_strings{cx.pin_dev, nullptr, 0, a_len, a_len};
        S->sb = ta_strings{cx.pin_dev + a_pad, nullptr, 0, b_len, b_len};
        S->out_dev = (uint32_t *)(cx.pin_dev + CallCtx::RESULT_OFF);
        S->out_host = (volatile uint32_t *)(cx.pin + CallCtx::RESULT_OFF);
        *S->out_host = TA_SLOT_EMPTY;
        return TA_OK;
    }
    Scratch &sc = tls_scratch(0);
    if ((rc = sc.ensure(a_pad + b_pad + 256))) return rc;
    uint8_t *base = (uint8_t *)sc.dev;
    if (a_len) TA_HIP(hipMemcpyAsync(base, a, a_len, hipMemcpyHostToDevice, cx.st));
    if (b_len) TA_HIP(hipMemcpyAsync(base + a_pad, b, b_len, hipMemcpyHostToDevice, cx.st));
    S->sa = ta_strings{base, nullptr, 0, a_len, a_len};
    S->sb = ta_strings{base + a_pad, nullptr, 0, b_len, b_len};
    S->out_dev = (uint32_t *)(base + a_pad + b_pad);
    S->out_host = nullptr;
    return TA_OK;
}

// `single_store`: the call was ONE kernel whose only write to the result slot is its answer.  The slot then starts out as a
// value no answer can take, and the host watches the pinned word instead of going through the runtime's completion
// machinery (the store becomes visible no later than the kernel's end-of-kernel release); after 20 ms without an answer the
// ordinary stream synchronisation takes over (and reports a fault, if that is what happened).
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#else
    __asm__ __volatile__("" ::: "memory");
#endif
}
static int fetch_u32(const Staged &S, uint32_t *out, bool single_store = false) {
    if (S.out_host) {
        if (single_store) {
            // watch the word (the answer of a short pair arrives within ~10-25 us, of a 4 KiB pair within ~0.6 ms; the runtime's own
            // wait costs ~40 us more than the watch), for at most 2 ms: then the runtime's wait takes over
            const auto t0 = std::chrono::steady_clock::now();
            for (uint32_t spin = 0; *S.out_host == TA_SLOT_EMPTY; spin++) {
                cpu_relax();
                if ((spin & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
            }
            const uint32_t v = *S.out_host;
            if (v != TA_SLOT_EMPTY) { *out = v; return TA_OK; }
        }
        TA_HIP(hipStreamSynchronize(S.st));
        *out = *S.out_host;
        return TA_OK;
    }
    TA_HIP(hipMemcpyAsync(out, S.out_dev, 4, hipMemcpyDeviceToHost, S.st));
    TA_HIP(hipStreamSynchronize(S.st));
    return TA_OK;
}

int ta_hamming(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len, uint32_t *out) {
    if (!out) return TA_ERR_ARG;
    if (a_len != b_len) return TA_ERR_LEN_MISMATCH;          // src/hamming.rs:318
    Staged S;
    int rc = stage_pair(a, a_len, b, b_len, &S);
    if (rc) return rc;
    rc = ta_hamming_batch(&S.sa, &S.sb, 1, S.out_dev, S.st);
    if (rc) return rc;
    return fetch_u32(S, out, true);                          // one kernel, one store
}

int ta_levenshtein_simd_k_with_opts(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len,
                                    uint32_t k, int trace_on, const ta_edit_costs *costs, uint32_t *out) {
    if (!out) return TA_ERR_ARG;
    if (!costs_ok(costs)) return TA_ERR_BAD_COSTS;
    if (trace_on) return TA_ERR_UNSUPPORTED;
    if (!device_ready()) return TA_ERR_HIP;
    if (a_len == 0 && b_len == 0) { *out = 0; return TA_OK; }   // src/levenshtein.rs:721-727
    Staged S;
    int rc = stage_pair(a, a_len, b, b_len, &S);
    if (rc) return rc;
    rc = ta_levenshtein_k_batch(&S.sa, &S.sb, 1, k, costs, S.out_dev, S.st);
    if (rc) return rc;
    return fetch_u32(S, out, g_answer_single_store);         // lev_pass says whether the pass was one kernel with one store
}

}  // extern "C"

// trace_on = true beyond the register band (unit costs): row-blocked bit-parallel kernel with 3-bit records + host walk
static int trace_widebits(const uint8_t *x, size_t n, const uint8_t *y, size_t m, bool swap, uint32_t k, const ta_edit_costs *costs,
                          uint32_t *out, ta_edit **edits, size_t *n_edits, uint64_t rec_words, uint64_t tcols) {
    Staged S;
    int rc = stage_pair(x, n, y, m, &S);
    if (rc) return rc;
    ta_strings &sa = S.sa, &sb = S.sb;
    uint32_t *od = S.out_dev;
    hipStream_t tst = S.st;
    Scratch &ts = tls_scratch(9), &bl = tls_scratch(6);
    if ((rc = ts.ensure((size_t)rec_words * 4)) || (rc = bl.ensure((size_t)6 * (m + 66) * 4))) return rc;
    LevParams P;
    P.a = view_of(&sa); P.b = view_of(&sb);
    P.subset = nullptr; P.out = od; P.n = 1; P.k = k;
    P.mc = 1; P.gc = 1; P.sg = 0; P.tc = costs->has_transpose ? 1 : 0;
    P.u = lev_batch_unit_k(k, 1, 1, 0, m);
    P.o = 0; P.L = 64; P.PW = 1; P.lds_per_wave = 0; P.Tw = 0; P.ch = 0;
    P.trace = (uint32_t *)ts.dev; P.trace_cols = tcols;
    P.bnd = (uint32_t *)bl.dev; P.bnd_line = (uint64_t)m + 66;
    TA_HIP(lev_widebits_trace_launch(P, costs->has_transpose != 0, tst));
    uint32_t d = 0;
    rc = fetch_u32(S, &d);
    if (rc) return rc;
    *out = d;
    if (d == TA_NONE) return TA_OK;
    std::vector<uint32_t> rec((size_t)rec_words);
    TA_HIP(hipMemcpy(rec.data(), ts.dev, (size_t)rec_words * 4, hipMemcpyDeviceToHost));
    WbTrace T{rec.data(), tcols, 2u, (uint32_t)n, (uint32_t)m, P.u};
    std::vector<ta_edit> res;
    size_t ci = n, cj = m;
    const bool ok = wb_trace_walk(T, x, y, d, costs->has_transpose != 0, [&](int code) {
        uint32_t e;
        switch (code) {                                                        // relabelling as in ta_levenshtein_trace (:561-603)
            case 0: ci--; cj--; e = (x[ci] == y[cj]) ? TA_EDIT_MATCH : TA_EDIT_MISMATCH; break;
            case 1: cj--; e = swap ? TA_EDIT_BGAP : TA_EDIT_AGAP; break;
            case 2: ci--; e = swap ? TA_EDIT_AGAP : TA_EDIT_BGAP; break;
            default: ci -= 2; cj -= 2; e = TA_EDIT_TRANSPOSE; break;
        }
        if (!res.empty() && res.back().edit == e) res.back().count++;
        else res.push_back(ta_edit{e, 0u, 1u});
    });
    if (!ok) { set_last_error_msg("traceback records are inconsistent"); return TA_ERR_HIP; }
    *n_edits = res.size();
    if (!res.empty()) {
        *edits = (ta_edit *)malloc(res.size() * sizeof(ta_edit));
        for (size_t t = 0; t < res.size(); t++) (*edits)[t] = res[res.size() - 1 - t];   // :605 reverse
    }
    return TA_OK;
}

// trace_on = true beyond the register band, any EditCosts: the DP wide kernel with 2-bit argmin codes + the usual walk
static int trace_wide(const uint8_t *x, size_t n, const uint8_t *y, size_t m, bool swap, uint32_t k, const ta_edit_costs *costs,
                      uint32_t *out, ta_edit **edits, size_t *n_edits, uint64_t code_words, uint64_t tcols) {
    Staged S;
    int rc = stage_pair(x, n, y, m, &S);
    if (rc) return rc;
    ta_strings &sa = S.sa, &sb = S.sb;
    uint32_t *od = S.out_dev;
    hipStream_t tst = S.st;
    Scratch &ts = tls_scratch(9);
    if ((rc = ts.ensure((size_t)code_words * 4))) return rc;
    LevParams P;
    P.a = view_of(&sa); P.b = view_of(&sb);
    P.subset = nullptr; P.out = od; P.n = 1; P.k = k;
    P.mc = costs->mismatch_cost; P.gc = costs->gap_cost; P.sg = costs->start_gap_cost;
    P.tc = costs->has_transpose ? costs->transpose_cost : 0;
    P.u = lev_batch_unit_k(k, P.mc, P.gc, P.sg, m);
    P.o = 0; P.L = 0; P.PW = 1; P.Tw = 0; P.ch = 0;
    P.lds_per_wave = (uint32_t)(m + 2);                                         // boundary line length
    P.trace = (uint32_t *)ts.dev; P.trace_cols = tcols;
    TA_HIP(lev_wide_trace_launch(P, costs->has_transpose != 0, tst));
    uint32_t d = 0;
    rc = fetch_u32(S, &d);
    if (rc) return rc;
    *out = d;
    if (d == TA_NONE) return TA_OK;
    std::vector<uint32_t> tr((size_t)code_words);
    TA_HIP(hipMemcpy(tr.data(), ts.dev, (size_t)code_words * 4, hipMemcpyDeviceToHost));
    std::vector<ta_edit> res;
    size_t i = n, j = m;
    while (i > 0 || j > 0) {                                                    // :561-603
        uint32_t code;
        if (i == 0) code = 1;                                                   // row 0 is one a_gap run, column 0 one b_gap run
        else if (j == 0) code = 2;
        else {
            const size_t q = (i - 1) / 2048, r = (i - 1) % 2048, lane = r / 32, rr = r % 32;
            code = (tr[((q * tcols + j) * 64 + lane) * 2 + (rr >> 4)] >> (2 * (rr & 15))) & 3u;
        }
        uint32_t e;
        switch (code) {
            case 0: i--; j--; e = (x[i] == y[j]) ? TA_EDIT_MATCH : TA_EDIT_MISMATCH; break;
            case 1: j--; e = swap ? TA_EDIT_BGAP : TA_EDIT_AGAP; break;
            case 2: i--; e = swap ? TA_EDIT_AGAP : TA_EDIT_BGAP; break;
            default: i -= 2; j -= 2; e = TA_EDIT_TRANSPOSE; break;
        }
        if (!res.empty() && res.back().edit == e) res.back().count++;
        else res.push_back(ta_edit{e, 0u, 1u});
    }
    *n_edits = res.size();
    if (!res.empty()) {
        *edits = (ta_edit *)malloc(res.size() * sizeof(ta_edit));
        for (size_t t = 0; t < res.size(); t++) (*edits)[t] = res[res.size() - 1 - t];   // :605 reverse
    }
    return TA_OK;
}

// The checkpoint-and-recompute traceback of one (sub-)batch of the unit-cost families, bands of up to 33 diagonals: the
// caller (ta_levenshtein_trace_batch) has validated the arguments and bounded n so that n x runs_cap fits 32 bits and the scratch fits memory.
// packed != nullptr: the packed form -- the walk writes the runs straight into the caller's buffer (cap words per pair, right-aligned): no run
// lists in scratch, no last step.
static int trace_bits_batch(const ta_strings *a, const ta_strings *b, size_t n, uint32_t k, const ta_edit_costs *costs, uint64_t max_len,
                            uint32_t *out_dev, ta_edit *edits_dev, uint32_t *n_edits_dev, size_t cap, hipStream_t st, uint32_t *packed = nullptr) {
    int rc;
    const bool trans = costs->has_transpose != 0;
    const uint32_t u = lev_batch_unit_k(k, 1, 1, 0, max_len);
    // Fixed-length batches: the distance pass IS the forward sweep -- the stride-8 kernel's CKPT instantiation stores the column
    // state in front of every 16th column (rows = the shorter string: the views are swapped for it where a is the longer one; the
    // distance is symmetric; CSR batches: pair by pair inside the kernel).  TA_TRACE_OWN_SWEEP=1 pins the trace kernel's own sweep
    // (TA_TRACE_CSR_OWN_SWEEP=1: for CSR batches only, an A/B).
    const bool fixed = !a->off && !b->off;
    const LevBitsPlan bp8 = lev_bits_make_plan(k, 1, 1, 0, trans, trans ? 1u : 0u, max_len, 0, 0, 3);
    const bool fold = bp8.ok && bp8.s8 && !env_int("TA_TRACE_OWN_SWEEP") && !env_int("TA_TRACE_TILE") && (fixed || !env_int("TA_TRACE_CSR_OWN_SWEEP"));
    const uint32_t tile = fold ? 16u : lev_bits_trace_tile(), tiles = (uint32_t)((max_len + tile - 1) / tile) + 1u, waves = (uint32_t)((n + 63) / 64);
    // CSR batches: both kernels run a wavefront to the longest of its 64 pairs -- the pairs are taken in length order (as
    // ta_levenshtein_k_batch takes them), the same list for the distance pass and the trace kernel.  TA_NO_LENGTH_ORDER=1 keeps the batch order.
    const uint32_t *order = nullptr;
    if (!fixed && n >= 4096 && max_len >= 16 && !env_int("TA_NO_LENGTH_ORDER") &&
        (rc = order_pairs(a, b, (uint32_t)n, u, max_len, false, false, st, &order, nullptr))) return rc;
    if (!fold && (rc = lev_pass(a, b, (uint32_t)n, order, k, costs, max_len, out_dev, st))) return rc;
    // (a script of cost <= u has at most 2 u + 1 runs, and never more than n + m)
    uint32_t runs_cap = (uint32_t)(2 * max_len + 1 < 2ull * u + 2 ? 2 * max_len + 1 : 2ull * u + 2);
    Scratch &cs = tls_scratch(9), &ps = tls_scratch(8), &ss = tls_scratch(7);
    if ((rc = cs.ensure((size_t)waves * tiles * lev_bits_trace_ckpt_words(trans) * 64u * 4u)) || (!packed && (rc = ps.ensure((size_t)n * runs_cap * 4u))) ||
        (rc = ss.ensure((size_t)n * 4u))) return rc;
    LevBitsTraceParams T;
    T.a = view_of(a); T.b = view_of(b); T.dist = out_dev; T.n = (uint32_t)n; T.u = u;
    T.ckpt = (uint32_t *)cs.dev; T.ckpt_tiles = tiles; T.runs = (uint32_t *)ps.dev; T.runs_cap = env_int("TA_TRACE_SKIP_WALK") ? 0u : runs_cap; T.n_runs = (uint32_t *)ss.dev;
    T.subset = order;
    if (packed) { T.runs = packed; T.packed_cap = (uint32_t)cap; }
    if (fold) {
        const bool sw = fixed && a->len > b->len;              // (CSR batches: the kernel swaps pair by pair)
        LevParams P;
        P.a = view_of(sw ? b : a); P.b = view_of(sw ? a : b);
        P.subset = order; P.trace = nullptr; P.out = out_dev; P.n = (uint32_t)n; P.k = k;
        P.mc = 1; P.gc = 1; P.sg = 0; P.tc = trans ? 1 : 0;
        P.u = bp8.u; P.o = 0; P.L = 1; P.PW = 64; P.lds_per_wave = bp8.lds_per_wave; P.Tw = bp8.Tw; P.ch = bp8.ch;
        P.ckpt = (uint32_t *)cs.dev; P.ckpt_tiles = tiles;
        TA_HIP(lev_bits_launch(P, bp8, trans, max_len, st, nullptr, nullptr));
        ta_launch_info l0 = {};
        ta_lev_select sel;
        ta_levenshtein_select((size_t)max_len, (size_t)max_len, k, costs, &sel);
        l0.cell_bits = sel.cell_bits; l0.transpose = trans;
        g_last_launch = l0;
    }
    ta_launch_info li = g_last_launch;                 // (the distance pass's: kernel 3)
    uint32_t grid = 0, lds = 0;
    TA_HIP(lev_bits_trace_launch(T, trans, fold, edits_dev, n_edits_dev, cap, st, &grid, &lds));
    li.kernel = 8; li.diags_per_lane = 33; li.lanes_per_pair = 1; li.pairs_per_wave = 64; li.grid = grid; li.lds_bytes = lds;
    g_last_launch = li;
    return TA_OK;
}

extern "C" {

/* levenshtein_simd_k_with_opts(..., trace_on = true): distance + run-length edit script.
 * The band-wavefront kernel stores a 2-bit argmin code per cell (tie order of src/levenshtein.rs:493-532);
 * the walk from (n, m) back to (0, 0) is the host part (:561-606). */
int ta_levenshtein_trace(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len, uint32_t k,
                         const ta_edit_costs *costs, uint32_t *out, ta_edit **edits, size_t *n_edits) {
    if (!out || !edits || !n_edits) return TA_ERR_ARG;
    *edits = nullptr; *n_edits = 0;
    if (!costs_ok(costs)) return TA_ERR_BAD_COSTS;
    if (!device_ready()) return TA_ERR_HIP;
    if (a_len == 0 && b_len == 0) { *out = 0; return TA_OK; }                   // :721-727 (Some(vec![]))
    const bool swap = a_len > b_len;                                            // :386-390
    const uint8_t *x = swap ? b : a, *y = swap ? a : b;
    const size_t n = swap ? b_len : a_len, m = swap ? a_len : b_len;
    ta_lev_select sel;
    ta_levenshtein_select(n, m, k, costs, &sel);
    const uint32_t gc = costs->gap_cost, sg = costs->start_gap_cost;
    LevPlan pl = lev_make_plan(sel.max_k, costs->mismatch_cost, gc, sg, m, 16, 0);
    if (!pl.ok) pl = lev_make_plan(sel.max_k, costs->mismatch_cost, gc, sg, m, 66, 0);
    if (!pl.ok) {
        // band too wide for the register kernel: the unit-cost families take the row-blocked bit-parallel kernel with records
        const bool unit = costs->mismatch_cost == 1 && gc == 1 && sg == 0 && (!costs->has_transpose || costs->transpose_cost == 1);
        const uint64_t stripes = (n + 4095) / 4096, tcols = (uint64_t)m + 64;
        const uint64_t rec_words = stripes * tcols * 64ull * 6ull;
        const uint64_t code_words = ((n + 2047) / 2048) * tcols * 64ull * 2ull;           // DP wide kernel: 2 bits per cell
        if (m > 0xFFFFFF00ull || (unit ? rec_words : code_words) * 4ull > (8ull << 30)) {
            set_last_error_msg("traceback: more than 8 GB of traceback records");
            return TA_ERR_UNSUPPORTED;
        }
        if (unit) return trace_widebits(x, n, y, m, swap, k, costs, out, edits, n_edits, rec_words, tcols);
        return trace_wide(x, n, y, m, swap, k, costs, out, edits, n_edits, code_words, tcols);
    }
    Staged S;
    int rc = stage_pair(x, n, y, m, &S);
    if (rc) return rc;
    ta_strings &sa = S.sa, &sb = S.sb;
    uint32_t *od = S.out_dev;
    hipStream_t tst = S.st;
    const uint32_t tw = (uint32_t)lev_trace_words(pl.D);
    const size_t taus = (n + m + 1) / 2 + 1;
    const size_t trace_words = taus * 2 * 64 * tw;
    Scratch &ts = tls_scratch(6);
    if ((rc = ts.ensure(trace_words * 4))) return rc;
    LevParams P;
    P.a = view_of(&sa); P.b = view_of(&sb);
    P.subset = nullptr; P.out = od; P.n = 1; P.k = k;
    P.mc = costs->mismatch_cost; P.gc = gc; P.sg = sg; P.tc = costs->has_transpose ? costs->transpose_cost : 0;
    P.u = pl.u; P.o = pl.o; P.L = pl.L; P.PW = pl.PW; P.lds_per_wave = pl.lds_per_wave; P.Tw = pl.Tw; P.ch = pl.ch;
    P.trace = (uint32_t *)ts.dev;
    TA_HIP(lev_band_trace_launch(P, pl, sg > 0, costs->has_transpose != 0, tst));
    uint32_t d = 0;
    rc = fetch_u32(S, &d);
    if (rc) return rc;
    *out = d;
    if (d == TA_NONE) return TA_OK;
    std::vector<uint32_t> tr(trace_words);
    TA_HIP(hipMemcpy(tr.data(), ts.dev, trace_words * 4, hipMemcpyDeviceToHost));
    std::vector<ta_edit> res;
    const uint32_t o_pair = lev_pair_offset(pl.u, n, m);
    size_t i = n, j = m;
    while (i > 0 || j > 0) {                                                    // :561-603
        const uint32_t s = (uint32_t)(i + j), p = (uint32_t)(j + o_pair - i);
        const uint32_t g = p / (uint32_t)pl.D, q = p % (uint32_t)pl.D, par = q & 1u, c = q >> 1, tau = (s - 1) >> 1;
        const uint32_t word = tr[(((size_t)tau * 2 + par) * 64 + g) * tw + ((2 * c) >> 5)];
        const uint32_t code = (word >> ((2 * c) & 31)) & 3u;
        uint32_t e;
        switch (code) {
            case 0: i--; j--; e = (x[i] == y[j]) ? TA_EDIT_MATCH : TA_EDIT_MISMATCH; break;
            case 1: j--; e = swap ? TA_EDIT_BGAP : TA_EDIT_AGAP; break;
            case 2: i--; e = swap ? TA_EDIT_AGAP : TA_EDIT_BGAP; break;
            default: i -= 2; j -= 2; e = TA_EDIT_TRANSPOSE; break;
        }
        if (!res.empty() && res.back().edit == e) res.back().count++;
        else res.push_back(ta_edit{e, 0u, 1u});
    }
    *n_edits = res.size();
    if (!res.empty()) {
        *edits = (ta_edit *)malloc(res.size() * sizeof(ta_edit));
        for (size_t t = 0; t < res.size(); t++) (*edits)[t] = res[res.size() - 1 - t];   // :605 reverse
    }
    return TA_OK;
}

/* levenshtein_simd_k_with_opts(a_i, b_i, k, trace_on = true, costs) for a whole batch, everything on the device: out_dev[i] = the distance or
 * TA_NONE, n_edits_dev[i] = the runs of pair i's script (0 for None), edits_dev[i * cap ..] = the script, front to back, as ta_edit records --
 * edit for edit what the reference returns (its tie order, its swap of the shorter string onto the rows, its gap relabelling).  Nothing
 * comes back to the host and nothing synchronises: the DP band kernel stores 2-bit argmin codes (the records: scratch, bounded by
 * working through the batch in chunks), a second kernel walks them, one lane per pair.  A script of more than `cap` runs is cut
 * (n_edits_dev[i] > cap says so); 2 k + 1 runs always hold a script of cost <= k.  Bands beyond the register kernel (more than 4222
 * diagonals) are TA_ERR_UNSUPPORTED here -- ta_levenshtein_trace serves those pairs one by one. */
static int trace_batch_impl(const ta_strings *a, const ta_strings *b, size_t n, uint32_t k, const ta_edit_costs *costs,
                            uint32_t *out_dev, ta_edit *edits_dev, uint32_t *packed_dev, uint32_t *n_edits_dev, size_t cap, void *stream);
int ta_levenshtein_trace_batch(const ta_strings *a, const ta_strings *b, size_t n, uint32_t k, const ta_edit_costs *costs,
                               uint32_t *out_dev, ta_edit *edits_dev, uint32_t *n_edits_dev, size_t cap, void *stream) {
    if (n && !edits_dev) { set_last_error_msg("bad batch arguments"); return TA_ERR_ARG; }
    return trace_batch_impl(a, b, n, k, costs, out_dev, edits_dev, nullptr, n_edits_dev, cap, stream);
}
/* The same with PACKED records: one word per run, (edit << 29) | count, pair i's script the min(n_edits_dev[i], cap) words that END at
 * packed_dev[(i + 1) * cap] -- front to back, right-aligned in the pair's slot of `cap` words (a script of more runs keeps its LAST cap runs;
 * words in front of the script are not written).  A quarter of the 16-byte records' bytes; on the checkpoint route (LEVENSHTEIN_COSTS /
 * RDAMERAU_COSTS and their multiples, k / g <= 32 (30)) the walk writes every run where it belongs: no run lists in scratch, no reversal step.
 * Strings of 2^29 bytes and more: TA_ERR_UNSUPPORTED (a run's count has 29 bits). */
int ta_levenshtein_trace_batch_packed(const ta_strings *a, const ta_strings *b, size_t n, uint32_t k, const ta_edit_costs *costs,
                                      uint32_t *out_dev, uint32_t *packed_dev, uint32_t *n_edits_dev, size_t cap, void *stream) {
    if (n && !packed_dev) { set_last_error_msg("bad batch arguments"); return TA_ERR_ARG; }
    return trace_batch_impl(a, b, n, k, costs, out_dev, nullptr, packed_dev, n_edits_dev, cap, stream);
}
}  // extern "C"

static int trace_batch_impl(const ta_strings *a, const ta_strings *b, size_t n, uint32_t k, const ta_edit_costs *costs,
                            uint32_t *out_dev, ta_edit *edits_dev, uint32_t *packed_dev, uint32_t *n_edits_dev, size_t cap, void *stream) {
    int rc = check_batch_args(a, b, n, out_dev);
    if (rc) return rc;
    if (n && (!n_edits_dev || cap == 0)) { set_last_error_msg("bad batch arguments"); return TA_ERR_ARG; }
    if (!costs_ok(costs)) return TA_ERR_BAD_COSTS;
    if (!device_ready()) return TA_ERR_HIP;
    if (n == 0) return TA_OK;
    hipStream_t st = (hipStream_t)stream;
    StreamGuard guard(st);
    uint64_t max_len = 0;
    if ((rc = batch_max_len(a, b, (uint32_t)n, st, &max_len))) return rc;
    const uint32_t gc = costs->gap_cost, sg = costs->start_gap_cost;
    if (packed_dev && (max_len >= (1ull << 29) || cap > 0xFFFFFFFFull)) { set_last_error_msg("packed tracebacks: strings of 2^29 bytes and more"); return TA_ERR_UNSUPPORTED; }
    // Unit costs times g (lev_unit_scale): every alignment costs g times its unit cost, every comparison of the recurrence and of the walk
    // keeps its outcome -- the script IS the unit-cost script for k / g, the distance g times the unit one (3.4c): the checkpoint kernel
    // serves EditCosts(g, g, 0, None | Some(g)) too.  TA_NO_UNIT_SCALE=1 keeps the DP band kernel's records.
    if (const uint32_t g = lev_unit_scale(costs->mismatch_cost, gc, sg, costs->has_transpose != 0, costs->transpose_cost);
        g && !env_int("TA_NO_BITS") && !env_int("TA_NO_UNIT_SCALE") && !env_int("TA_TRACE_NO_BITS")) {
        const bool tr = costs->has_transpose != 0;
        const uint32_t uu = lev_batch_unit_k(k / g, 1, 1, 0, max_len);
        if ((uint64_t)uu + 1u + (tr ? 2u : 0u) <= 33u && max_len < (1ull << 29)) {
            const ta_edit_costs uc = {1, 1, 0, (uint8_t)(tr ? 1 : 0), (uint8_t)(tr ? 1 : 0)};
            if ((rc = trace_batch_impl(a, b, n, k / g, &uc, out_dev, edits_dev, packed_dev, n_edits_dev, cap, stream))) return rc;
            TA_HIP(scale_results_launch(out_dev, nullptr, (uint32_t)n, nullptr, g, st));
            ta_lev_select sel;
            ta_levenshtein_select((size_t)max_len, (size_t)max_len, k, costs, &sel);
            g_last_launch.cell_bits = sel.cell_bits;
            return TA_OK;
        }
    }
    // LEVENSHTEIN_COSTS / RDAMERAU_COSTS with a band of up to 33 diagonals (k <= 32; 30 with the transposition term): no per-cell records --
    // the distance pass, then ONE kernel that sweeps the columns forwards with a checkpoint (8 bytes per pair) every 16 columns, recomputes
    // tile after tile backwards with the column states of the tile in LDS, walks, and writes the runs (lev_bits_trace_body.h): 2 x the
    // strings' bytes instead of 21 x.  TA_TRACE_NO_BITS=1 keeps the DP band kernel's records for these cost sets too.
    {
        const bool trans = costs->has_transpose != 0;
        const bool unit = costs->mismatch_cost == 1 && gc == 1 && sg == 0 && (!trans || costs->transpose_cost == 1);
        const uint32_t u = lev_batch_unit_k(k, 1, 1, 0, max_len);
        if (unit && (uint64_t)u + 1u + (trans ? 2u : 0u) <= 33u && max_len < (1ull << 29) /* a run's count has 29 bits */ && !env_int("TA_TRACE_NO_BITS") && !env_int("TA_NO_BITS")) {
            // The kernel indexes its run lists with 32 bits (pair slot x runs_cap) and its scratch -- checkpoints, run lists, run counts -- is
            // proportional to the pairs of a launch: a batch beyond either bound is worked through in chunks of whole wavefronts, each an
            // ordinary sub-batch on the same stream (65M pairs at k = 32 wrapped the index; a huge batch asked for all of its
            // scratch at once where the record path below chunks by free memory).  TA_TRACE_CHUNK_PAIRS=n pins the chunk (tests).
            const uint32_t runs_cap = (uint32_t)(2 * max_len + 1 < 2ull * u + 2 ? 2 * max_len + 1 : 2ull * u + 2);
            const uint64_t tiles16 = (max_len + 15) / 16 + 1;
            const uint64_t per_pair = tiles16 * lev_bits_trace_ckpt_words(trans) * 4ull + (uint64_t)runs_cap * 4ull + 4ull + 8ull;
            uint64_t chunk = 0xFFFFFFF0ull / (packed_dev ? (cap > runs_cap ? (uint64_t)cap : runs_cap) : (runs_cap ? runs_cap : 1u));
            const uint64_t held = tls_scratch(9).cap + tls_scratch(8).cap + tls_scratch(7).cap;
            if (per_pair * n > held && !stream_is_capturing(st)) {   // the scratch has to grow: not beyond half of what is free right now
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                    const uint64_t by_mem = (((uint64_t)free_b + held) / 2) / per_pair;
                    if (by_mem < chunk) chunk = by_mem;
                }
            }
            if (env_int("TA_TRACE_CHUNK_PAIRS") > 0) chunk = (uint64_t)env_int("TA_TRACE_CHUNK_PAIRS");
            chunk &= ~63ull;
            if (chunk < 64) chunk = 64;
            if (chunk >= n) return trace_bits_batch(a, b, n, k, costs, max_len, out_dev, edits_dev, n_edits_dev, cap, st, packed_dev);
            for (uint64_t lo = 0; lo < n; lo += chunk) {
                const size_t cnt = (size_t)((n - lo) < chunk ? (n - lo) : chunk);
                const ta_strings sa = a->off ? ta_strings{a->blob, a->off + lo, 0, 0, max_len} : ta_strings{a->blob + lo * a->stride, nullptr, a->stride, a->len, a->len};
                const ta_strings sb = b->off ? ta_strings{b->blob, b->off + lo, 0, 0, max_len} : ta_strings{b->blob + lo * b->stride, nullptr, b->stride, b->len, b->len};
                if ((rc = trace_bits_batch(&sa, &sb, cnt, k, costs, max_len, out_dev + lo, edits_dev ? edits_dev + lo * cap : nullptr, n_edits_dev + lo, cap, st,
                                           packed_dev ? packed_dev + lo * cap : nullptr))) return rc;
            }
            return TA_OK;
        }
    }
    if (packed_dev) {
        // the record routes write ta_edit records: sub-batches through a bounded ta_edit scratch (<= 256 MiB, whole wavefronts), packed as they come
        // (the ta_edit form cuts a long script at its FRONT `cap` runs, the packed form keeps the LAST ones: the scratch takes whole scripts -- a
        // script has at most n + m runs, and one of cost <= k at most 2 k + 1: every edit costs at least 1)
        const uint64_t cap_in = 2ull * k + 1 < 2 * max_len + 2 ? 2ull * k + 1 : 2 * max_len + 2;
        uint64_t chunk = (256ull << 20) / (cap_in * sizeof(ta_edit));
        chunk &= ~63ull;
        if (chunk < 64) chunk = 64;
        if (chunk > n) chunk = n;
        Scratch &es = tls_scratch(12);
        if ((rc = es.ensure((size_t)(chunk * cap_in * sizeof(ta_edit))))) return rc;
        for (uint64_t lo = 0; lo < n; lo += chunk) {
            const size_t cnt = (size_t)((n - lo) < chunk ? (n - lo) : chunk);
            const ta_strings sa = a->off ? ta_strings{a->blob, a->off + lo, 0, 0, max_len} : ta_strings{a->blob + lo * a->stride, nullptr, a->stride, a->len, a->len};
            const ta_strings sb = b->off ? ta_strings{b->blob, b->off + lo, 0, 0, max_len} : ta_strings{b->blob + lo * b->stride, nullptr, b->stride, b->len, b->len};
            if ((rc = trace_batch_impl(&sa, &sb, cnt, k, costs, out_dev + lo, (ta_edit *)es.dev, nullptr, n_edits_dev + lo, (size_t)cap_in, stream))) return rc;
            TA_HIP(pack_edits_launch((const ta_edit *)es.dev, n_edits_dev + lo, (uint32_t)cnt, cap_in, packed_dev + lo * cap, cap, st));
        }
        return TA_OK;
    }
    // the trace kernels exist for 16, 34 and 66 diagonals per lane: the cheapest layout per pair that holds the band
    LevPlan pl = {};
    pl.ok = false;
    double best = 1e300;
    for (int D : {34, 16, 66}) {
        const LevPlan c = lev_make_plan(k, costs->mismatch_cost, gc, sg, max_len, D, 0);
        const double cost = c.ok ? (5.0 * c.D + 24.0) / c.PW * (c.D > 40 ? 1.25 : 1.0) : 1e300;
        if (c.ok && cost < best) { best = cost; pl = c; }
    }
    if (!pl.ok) { set_last_error_msg("batch traceback: band wider than the register kernel (4222 diagonals)"); return TA_ERR_UNSUPPORTED; }
    const uint32_t tw = (uint32_t)lev_trace_words(pl.D);
    const uint64_t taus = (2 * max_len + 1) / 2 + 1;
    const uint64_t wave_words = taus * 2 * 64 * tw;
    // the records of one chunk: at most ~4 GiB and at most a quarter of the device memory that is free right now (or one wavefront's, if
    // that is more)
    const uint64_t waves_all = (n + pl.PW - 1) / pl.PW;
    uint64_t waves_per_chunk = ((4ull << 30) / 4) / wave_words;
    if (waves_per_chunk == 0) waves_per_chunk = 1;
    if (waves_per_chunk > waves_all) waves_per_chunk = waves_all;
    if (wave_words * waves_per_chunk * 4 > tls_scratch(6).cap) {      // the scratch has to grow: not beyond a quarter of what is free right now
        size_t free_b = 0, total_b = 0;                                // (asked only then: a captured call must not query the device)
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const uint64_t budget = ((uint64_t)free_b + tls_scratch(6).cap) / 4;
            if (wave_words * waves_per_chunk * 4 > budget) waves_per_chunk = (budget / 4) / wave_words;
            if (waves_per_chunk == 0) waves_per_chunk = 1;
        }
    }
    if (wave_words * waves_per_chunk * 4 > (48ull << 30)) { set_last_error_msg("batch traceback: more than 48 GB of records for one wavefront"); return TA_ERR_UNSUPPORTED; }
    Scratch &ts = tls_scratch(6), &ps = tls_scratch(9);
    const uint32_t path_words = (uint32_t)((2 * max_len) / 16 + 2);                  // 2-bit codes of a pair's path, sixteen per word
    if ((rc = ts.ensure((size_t)(wave_words * waves_per_chunk * 4))) || (rc = ps.ensure((size_t)(waves_per_chunk * pl.PW) * path_words * 4))) return rc;
    LevParams P;
    P.a = view_of(a); P.b = view_of(b);
    P.subset = nullptr; P.out = out_dev; P.k = k;
    P.mc = costs->mismatch_cost; P.gc = gc; P.sg = sg; P.tc = costs->has_transpose ? costs->transpose_cost : 0;
    P.u = pl.u; P.o = pl.o; P.L = pl.L; P.PW = pl.PW; P.lds_per_wave = pl.lds_per_wave; P.Tw = pl.Tw; P.ch = pl.ch;
    P.trace = (uint32_t *)ts.dev; P.trace_wave_words = wave_words;
    const uint64_t pairs_per_chunk = waves_per_chunk * pl.PW;
    for (uint64_t lo = 0; lo < n; lo += pairs_per_chunk) {
        P.pair_base = (uint32_t)lo;
        P.n = (uint32_t)((n - lo) < pairs_per_chunk ? (n - lo) : pairs_per_chunk);
        TA_HIP(lev_band_trace_batch_launch(P, pl, sg > 0, costs->has_transpose != 0, edits_dev, n_edits_dev, cap, (uint32_t *)ps.dev, path_words, st));
    }
    ta_launch_info li = {};
    li.kernel = 1; li.diags_per_lane = pl.D; li.lanes_per_pair = pl.L; li.pairs_per_wave = pl.PW; li.band_offset = pl.o;
    li.affine = sg > 0; li.transpose = costs->has_transpose != 0; li.grid = (uint32_t)waves_per_chunk; li.lds_bytes = pl.lds_per_wave;
    ta_lev_select sel;
    ta_levenshtein_select((size_t)max_len, (size_t)max_len, k, costs, &sel);
    li.cell_bits = sel.cell_bits;
    g_last_launch = li;
    set_last_kernel_name("lev_band_trace_kernel<%d, %s, %d>", pl.D, sg > 0 ? "true" : "false", costs->has_transpose ? 2 : 0);
    return TA_OK;
}

extern "C" {

/* levenshtein_exp_with_opts(..., trace_on = true), src/levenshtein.rs:1480-1494 */
int ta_levenshtein_exp_trace(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len,
                             const ta_edit_costs *costs, uint32_t *out, ta_edit **edits, size_t *n_edits) {
    uint32_t k = 30;
    for (int round = 0; round < 40; round++) {
        int rc = ta_levenshtein_trace(a, a_len, b, b_len, k, costs, out, edits, n_edits);
        if (rc != TA_OK || *out != TA_NONE) return rc;
        k = (k > 0x7FFFFFFFu) ? 0xFFFFFFFFu : k * 2;
    }
    return TA_OK;
}

int ta_levenshtein_simd_k(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len, uint32_t k, uint32_t *out) {
    ta_edit_costs c = ta_levenshtein_costs();
    return ta_levenshtein_simd_k_with_opts(a, a_len, b, b_len, k, 0, &c, out);
}
int ta_levenshtein(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len, uint32_t *out) {
    return ta_levenshtein_simd_k(a, a_len, b, b_len, 0xFFFFFFFFu, out);       // :1398
}
int ta_rdamerau(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len, uint32_t *out) {
    ta_edit_costs c = ta_rdamerau_costs();
    return ta_levenshtein_simd_k_with_opts(a, a_len, b, b_len, 0xFFFFFFFFu, 0, &c, out);   // :1420
}
int ta_levenshtein_exp_with_opts(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len,
                                 int trace_on, const ta_edit_costs *costs, uint32_t *out) {
    if (!out) return TA_ERR_ARG;
    if (!costs_ok(costs)) return TA_ERR_BAD_COSTS;
    if (trace_on) return TA_ERR_UNSUPPORTED;
    if (!device_ready()) return TA_ERR_HIP;
    if (a_len == 0 && b_len == 0) { *out = 0; return TA_OK; }
    Staged S;
    int rc = stage_pair(a, a_len, b, b_len, &S);
    if (rc) return rc;
    rc = ta_levenshtein_exp_batch(&S.sa, &S.sb, 1, costs, S.out_dev, S.st);
    if (rc) return rc;
    // one pass, one kernel, one store (a short pair goes straight to the unbounded pass): the pinned word can be watched; after
    // several rounds the slot already holds an earlier round's None
    return fetch_u32(S, out, g_exp_passes == 1 && g_answer_single_store);
}
int ta_levenshtein_exp(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len, uint32_t *out) {
    ta_edit_costs c = ta_levenshtein_costs();
    return ta_levenshtein_exp_with_opts(a, a_len, b, b_len, 0, &c, out);
}
int ta_rdamerau_exp(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len, uint32_t *out) {
    ta_edit_costs c = ta_rdamerau_costs();
    return ta_levenshtein_exp_with_opts(a, a_len, b, b_len, 0, &c, out);
}

}  // extern "C"
