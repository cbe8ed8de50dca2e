// ta_internal.h -- host-side plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/triple_accel_amd.h"
#include "lev_band_body.h"
#include "lev_plan.h"

namespace ta {

void set_last_error(const char *what, hipError_t e);
void set_last_error_msg(const char *msg);
// the dominant kernel of this thread's last pass as a profiler prints it, without "void ta::" and the parameter list
// (ta_last_kernel_name: bench.py refuses to splice committed counter figures recorded for another kernel)
void set_last_kernel_name(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

#define TA_HIP(expr)                                        \
    do {                                                    \
        hipError_t e__ = (expr);                            \
        if (e__ != hipSuccess) {                            \
            ::ta::set_last_error(#expr, e__);               \
            return TA_ERR_HIP;                              \
        }                                                   \
    } while (0)

// Thread-local grow-only device scratch.
struct Scratch {
    void *dev = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes);   // TA_OK / TA_ERR_HIP
    void release();
    ~Scratch();
};
constexpr int TA_SCRATCH_SLOTS = 18;
constexpr int TA_SLOT_SEARCH_HAY = 17;        // the host search entries' haystack staging: nothing else writes it (ta_levenshtein_search_resume relies on that)
Scratch &tls_scratch(int which);

// Per-thread context of the single-call host API: its own non-blocking stream (concurrent callers never meet on the null
// stream) and a pinned, device-mapped staging buffer -- a short pair is memcpy'd there, the kernel reads it in place and
// writes the answer back into it: one launch and one stream synchronisation per call, no staging copies.
struct CallCtx {
    static constexpr size_t PIN_BYTES = 64 * 1024, RESULT_OFF = PIN_BYTES - 64;
    hipStream_t st = nullptr;
    uint8_t *pin = nullptr;       // host address
    uint8_t *pin_dev = nullptr;   // the same bytes as the device addresses them
    int ensure();                 // TA_OK / TA_ERR_HIP
    void release();
};
CallCtx &call_ctx();

// The batch / *_dev entry points run on the CALLER's stream but keep state in thread-local scratch that their kernels may
// still be using when the call returns.  A later call from the same thread on a DIFFERENT stream first makes that stream
// wait (device-side, hipStreamWaitEvent) for the event recorded at the end of the previous call; calls that stay on one
// stream -- the normal case -- are ordered by the stream itself and cost nothing extra.
struct StreamGuard {
    hipStream_t st;
    bool capturing = false, was_capturing = false;     // the stream is being captured into a graph (Scratch::ensure refuses to grow then)
    explicit StreamGuard(hipStream_t s);
    ~StreamGuard();
};

// test / tuning switches: nullptr / 0 unless TA_TUNING was set when the library was loaded (no getenv on the call path)
bool tuning_enabled();
const char *env_str(const char *name);
int env_int(const char *name);

// ta_set_option(TA_OPT_EARLY_OUT) of the calling thread
bool early_out_enabled();
bool unit_prefilter_enabled();
// true when a HIP device is usable (lazy, cached)
bool device_ready();

// kernels (defined in the .hip files)
// trans: 0 none, 1 dot4-penalty form (2*mc <= 255 + tc), 2 select form
hipError_t lev_band_launch(const LevParams &P, const LevPlan &pl, bool affine, int trans, hipStream_t s,
                           uint32_t *grid_out, uint32_t *lds_out);
hipError_t lev_band_trace_launch(const LevParams &P, const LevPlan &pl, bool affine, bool trans, hipStream_t s);
// batch tracebacks (lev_band.hip): trace kernel over the pairs [P.pair_base, P.pair_base + P.n), then the device-side walk
// (path: n x path_words u32 of scratch, path_words >= (a_len + b_len) / 16 + 1 for every pair)
hipError_t lev_band_trace_batch_launch(const LevParams &P, const LevPlan &pl, bool affine, bool trans, ta_edit *edits, uint32_t *n_edits,
                                       uint64_t cap, uint32_t *path, uint32_t path_words, hipStream_t s);
hipError_t lev_bits_launch(const LevParams &P, const LevBitsPlan &pl, bool trans, uint64_t max_len, hipStream_t s,
                           uint32_t *grid_out, uint32_t *lds_out);
// the VLINE fetch form (lev_bits_vline.hip): CSR batches through the stride-8 window
hipError_t lev_bits_vline_launch(const LevParams &P, const LevBitsPlan &pl, bool trans, hipStream_t s, uint32_t *grid_out, uint32_t *lds_out);
hipError_t lev_bitsq_launch(const LevParams &P, bool trans, hipStream_t s, uint32_t *grid_out, uint32_t *lds_out);
hipError_t lev_bitsqw_launch(const LevParams &P, bool trans, hipStream_t s, uint32_t *grid_out, uint32_t *lds_out);
hipError_t lev_bits2_launch(const LevParams &P, const LevBits2Plan &pl, bool trans, hipStream_t s, uint32_t *grid_out, uint32_t *lds_out);
hipError_t lev_one_launch(const LevParams &P, bool trans, uint64_t max_len, hipStream_t s, uint32_t *lds_out);
bool lev_sliced_applies(const StrView &a, const StrView &b, uint32_t unit_k, uint32_t *strips_out);
hipError_t lev_sliced_launch(const StrView &a, const StrView &b, uint32_t n, uint32_t k, uint32_t unit_k, uint32_t *out,
                             hipStream_t st, uint32_t *grid_out, uint32_t *lds_out, uint32_t *pairs_per_wave);
hipError_t lev_widebits_launch(const LevParams &P, int rows_per_lane, uint64_t max_len, bool trans, hipStream_t s,
                               uint32_t *grid_out, uint32_t *lds_out);
hipError_t lev_widebits_huge_launch(const uint8_t *a, uint32_t a_len, const uint8_t *b, uint32_t b_len, uint32_t u, uint32_t k,
                                    int rows_per_lane, bool trans, uint32_t *out, hipStream_t s, uint32_t *launches_out);
hipError_t lev_wide_trace_launch(const LevParams &P, bool trans, hipStream_t s);
hipError_t lev_widebits_trace_launch(const LevParams &P, bool trans, hipStream_t s);
hipError_t lev_wide_launch(const LevParams &P, bool trans, hipStream_t s, uint32_t *grid_out, uint32_t *lds_out,
                           uint32_t *threads_out, uint32_t *dpt_out);
hipError_t hamming_batch_launch(const StrView &a, const StrView &b, uint32_t n, uint32_t *out, hipStream_t s);
hipError_t strings_maxlen_launch(const StrView &s, uint32_t n, uint32_t *out_max /*device, pre-zeroed*/, hipStream_t st);
hipError_t compact_none_launch(const uint32_t *out, const uint32_t *subset_in, uint32_t n_in, const uint32_t *n_in_dev /*optional: the list's length on the device*/,
                               uint32_t *subset_out, uint32_t *count, hipStream_t st);
// batch tracebacks of the unit-cost families by checkpoints + recomputation (lev_bits_trace.hip; bands of up to 33 diagonals)
uint32_t lev_bits_trace_ckpt_words(bool trans);
uint32_t lev_bits_trace_tile();
hipError_t lev_bits_trace_launch(const LevBitsTraceParams &P, bool trans, bool have_ckpt, ta_edit *edits, uint32_t *n_edits, uint64_t cap, hipStream_t s,
                                 uint32_t *grid_out, uint32_t *lds_out);
hipError_t fill_u32_launch(uint32_t *p, uint32_t v, uint32_t n, hipStream_t st);   // p[0..n) = v, as a kernel (graph-safe)
hipError_t compact_some_launch(const uint32_t *out, const uint32_t *subset_in, uint32_t n_in, uint32_t *subset_out, uint32_t *count /*device, pre-zeroed*/, hipStream_t st);
hipError_t scale_results_launch(uint32_t *out, const uint32_t *list /*pairs, or nullptr: 0..n*/, uint32_t n, const uint32_t *n_dev, uint32_t g, hipStream_t st);
hipError_t pack_edits_launch(const ta_edit *edits, const uint32_t *n_edits, uint32_t n, uint64_t cap_in, uint32_t *packed, uint64_t cap_out, hipStream_t st);
hipError_t compact_bound_launch(const uint32_t *out, const uint32_t *bound, uint32_t k, const uint32_t *list_in, uint32_t n_in,
                                const uint32_t *n_in_dev, uint32_t *list_out, uint32_t *count, hipStream_t st);
hipError_t bag_bound_launch(const StrView &a, const StrView &b, uint32_t n, uint32_t mc, uint32_t gc, uint32_t *bound, hipStream_t st);
hipError_t hits_best_launch(const ta_match *hits, uint64_t n, uint32_t *min_k /*device, preset to ~0*/, ta_match *out, uint32_t cap,
                            uint32_t *count /*device, pre-zeroed*/, hipStream_t st);
// counting sort of the pairs of a ragged batch by length class (util_kernels.hip): subset_out = the pairs (of subset_in, or
// 0..n) ordered so that 64 consecutive ones are within a few bytes of each other; bins = 2 x 32768 u32 of device scratch, the
// first half zero on entry (it is zero again on exit); *exact_columns: 64 consecutive pairs of the order share their exact column count
hipError_t length_order_launch(const StrView &a, const StrView &b, const uint32_t *subset_in, uint32_t n, uint32_t u, uint64_t max_len, bool by_steps,
                               uint32_t *bins, uint32_t *subset_out, hipStream_t st, bool exact = false, bool *exact_columns = nullptr);

struct SearchParams {
    const uint8_t *hay;       // device
    uint64_t hay_len;
    uint8_t needle[64];       // by value (kernarg): the lane-per-tile register kernels read it as scalars (needles <= 32), the
                              // wavefront-per-block kernel one byte per lane straight from the kernarg segment (needles <= 64)
    const uint8_t *needle_dev;   // device copy (any length)
    uint32_t *col_scratch;    // memory-backed column (long needles)
    uint32_t needle_len;
    uint32_t k, mc, gc, sg, tc;
    uint32_t anchored;
    uint32_t halo;            // bytes of left context each tile recomputes
    uint32_t tile;            // haystack positions emitted per lane
    uint64_t base, emit_from;
    ta_match *hits;           // device
    uint64_t cap;
    unsigned long long *count;   // device
};
hipError_t lev_search_launch(const SearchParams &P, bool packed, bool trans, hipStream_t s);
hipError_t lev_filter_launch(const SearchParams &P, bool trans, uint32_t *list, uint32_t list_cap, unsigned int *list_count,
                             hipStream_t s);
hipError_t lev_search_list_launch(const SearchParams &P, bool trans, const uint32_t *list, uint32_t n_list, hipStream_t s);

// Device-side control block of one filtered search pass (zeroed by ONE memset) and the report its last wavefront writes
// into host-mapped pinned memory: the host learns everything it needs from one stream synchronisation, with no copy and
// no round trip between the filter and the exact kernel.
constexpr uint32_t SEARCH_DONE_GROUPS = 16;
struct SearchCtl {
    unsigned long long count;     // hits emitted (may exceed the caller's cap)
    uint32_t n_list;              // 64-column blocks the filter flagged
    uint32_t pad0;
    uint32_t done2;               // groups of workgroups that have finished (the last one writes the report)
    uint32_t pad1;
    uint32_t pad[10];             // [0], [1]: timestamps of wavefront 0
    uint32_t done[SEARCH_DONE_GROUPS * 16];   // finished workgroups per group, one counter per 64-byte line: 512 bumps of ONE address
                                  // serialise at the memory side (~70 ns each); 16 lines of 32 run side by side
};
struct SearchReport {             // 64 bytes, followed by up to SEARCH_REPORT_SEL selected ta_match records (Best passes)
    uint64_t count;
    uint32_t n_list;
    uint32_t dense;               // 1: too many flagged blocks -- nothing was searched, the lane-per-tile kernel must run over everything
    uint32_t sel_count;           // Best: hits with the smallest k ...
    uint32_t sel_state;           // ... 1: all of them follow this header, 2: too many to select here (use ta_search_best_hits_dev)
    uint32_t min_k;
    uint32_t n_slots;
    uint32_t t[8];                // 100 MHz timestamps (s_memrealtime, low dword): [0] wavefront 0 enters, [1] it has finished its blocks,
                                  // [2] the last workgroup starts the report, [3] ... has written it   (scripts/measure_search_parts.py)
};
// Best passes: one slot per flagged block
struct SearchSlot {
    uint32_t min_cost, cnt;       // the block's best cost and its number of hits (0: none -- the slot is as the fill left it)
    uint64_t idx0;                // its hits are hits[idx0 .. idx0 + cnt)
};
constexpr uint32_t SEARCH_SLOT_CAP = 1u << 18;                                      // flagged blocks a fused Best pass handles (4 MiB of slots)
constexpr uint32_t SEARCH_REPORT_SEL = 680;                                         // 64 + 680 * 24 = 16 KiB
constexpr size_t SEARCH_REPORT_BYTES = 64 + (size_t)SEARCH_REPORT_SEL * sizeof(ta_match);
// thread-local pinned, device-mapped landing zone of the report
struct PinBox {
    uint8_t *host = nullptr, *dev = nullptr;
    int ensure();                 // TA_OK / TA_ERR_HIP
    void release();
};
PinBox &search_report_box();
void search_resident_reset();        // forget what ta_levenshtein_search_first left in the haystack staging buffer (ta_search.hip)
// one wavefront per flagged block (lev_search_wave_body.h), persistent grid reading n_list on the device; needles <= 64 bytes
hipError_t lev_search_wave_launch(const SearchParams &P, bool trans, bool best, const uint32_t *list, uint32_t cap_list,
                                  SearchCtl *ctl, SearchSlot *slots /* SEARCH_SLOT_CAP of them; Best passes */, uint8_t *report_dev,
                                  hipStream_t s);
hipError_t search_report_copy_launch(const unsigned long long *count, const uint32_t *nul_flag, const ta_match *hits, uint64_t cap, uint8_t *box, hipStream_t s);
hipError_t hamming_search_launch(const SearchParams &P, hipStream_t s, uint32_t *nul_flag = nullptr, bool *nul_done = nullptr);

// ta_multi.hip: the device set (one worker thread per entry).  multi_search_shards / multi_pair_shards: over how many of them a host
// haystack / a host batch of that size is spread (1: the calling thread's own device path).  The search forms return the All-mode hits
// sorted by end (best: only those with each shard's smallest k -- all the Best fold can keep), without the end == 0 match.
size_t multi_search_shards(size_t haystack_len);
size_t multi_pair_shards(size_t n);
int multi_levenshtein_search_host(const uint8_t *needle, size_t n, const uint8_t *hay, size_t h, uint32_t k, bool best, const ta_edit_costs *costs,
                                  std::vector<ta_match> &hits);
int multi_hamming_search_host(const uint8_t *needle, size_t n, const uint8_t *hay, size_t h, uint32_t k, bool check_nul, std::vector<ta_match> &hits);
// ta_hamming_search_dev under the scalar routine's contract (NUL bytes are fine, src/hamming.rs:96-146)
int hamming_search_dev_nocheck(const uint8_t *needle_host, size_t needle_len, const uint8_t *haystack_dev, size_t haystack_len, uint32_t k,
                               uint64_t base, ta_match *hits_dev, size_t cap, uint64_t *count_host, void *stream);

}  // namespace ta
