// ta_search.hip -- search entry points of the C ABI: special cases, haystack tiling, hit gathering,
// and the order-dependent Best post-pass (host).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "lev_filter_body.h"
#include "lev_search_body.h"
#include "ta_internal.h"

namespace ta {
hipError_t has_zero_byte_launch(const uint8_t *p, uint64_t len, uint32_t *flag, hipStream_t s);

static bool search_costs_ok(const ta_edit_costs *c) {
    if (!c || !(c->mismatch_cost > 0) || !(c->gap_cost > 0)) return false;
    if (c->has_transpose) {
        if (!(c->transpose_cost > 0) || !((c->transpose_cost >> 1) < c->mismatch_cost) ||
            !((c->transpose_cost >> 1) < c->gap_cost))
            return false;
    }
    return true;
}

// tile size: enough tiles to fill the chip (>= ~128K lanes) while keeping the halo overhead small
static uint32_t pick_tile(uint64_t hay_len, uint32_t halo) {
    if (const char *e = env_str("TA_SEARCH_TILE")) { long v = atol(e); if (v > 0) return (uint32_t)v; }
    uint64_t t = (hay_len + 524287) / 524288;          // ~2 full sets of resident lanes (256 CUs x 32 waves x 64)
    uint64_t lo = (uint64_t)halo * 2;
    if (t < lo) t = lo;
    if (t < 64) t = 64;
    if (t > 32768) t = 32768;
    return (uint32_t)t;
}

static void fill_params(SearchParams &P, const uint8_t *needle, size_t n, const uint8_t *hay, size_t h, uint32_t k,
                        const ta_edit_costs *c, int anchored, uint64_t base, uint64_t emit_from, ta_match *hits,
                        size_t cap, unsigned long long *count) {
    memset(&P, 0, sizeof(P));
    P.hay = hay; P.hay_len = h;
    memcpy(P.needle, needle, n < sizeof(P.needle) ? n : sizeof(P.needle));
    P.needle_len = (uint32_t)n;
    P.k = k;
    if (c) {
        P.mc = c->mismatch_cost; P.gc = c->gap_cost; P.sg = c->start_gap_cost;
        P.tc = c->has_transpose ? c->transpose_cost : 0;
    }
    P.anchored = anchored ? 1 : 0;
    P.base = base; P.emit_from = emit_from; P.hits = hits; P.cap = cap; P.count = count;
}

}  // namespace ta

using namespace ta;

// The hits of a device-resident All-mode result that can survive the Best fold -- those with the smallest k -- sorted by end.
static int best_hits_of(const ta_match *hits_dev, uint64_t count, std::vector<ta_match> &v, hipStream_t st) {
    v.clear();
    if (count == 0) return TA_OK;
    const uint32_t cap = 1u << 16;                                // best hits kept on the first try (rarely more than a handful)
    Scratch &sel = tls_scratch(12), &cnt = tls_scratch(2);
    int rc;
    if ((rc = sel.ensure((size_t)cap * sizeof(ta_match))) || (rc = cnt.ensure(64))) return rc;
    uint32_t *ctr = (uint32_t *)cnt.dev;                          // [0] count, [1] min k
    TA_HIP(hipMemsetAsync(ctr, 0, 4, st));
    TA_HIP(hipMemsetAsync(ctr + 1, 0xFF, 4, st));
    TA_HIP(hits_best_launch(hits_dev, count, ctr + 1, (ta_match *)sel.dev, cap, ctr, st));
    uint32_t host[2] = {0, 0};
    TA_HIP(hipMemcpyAsync(host, ctr, 8, hipMemcpyDeviceToHost, st));
    TA_HIP(hipStreamSynchronize(st));
    if (host[0] <= cap) {
        v.resize(host[0]);
        if (host[0]) TA_HIP(hipMemcpyAsync(v.data(), sel.dev, (size_t)host[0] * sizeof(ta_match), hipMemcpyDeviceToHost, st));
        TA_HIP(hipStreamSynchronize(st));
    } else {                                                      // a flood of equally good hits: take everything and filter here
        std::vector<ta_match> all(count);
        TA_HIP(hipMemcpyAsync(all.data(), hits_dev, (size_t)count * sizeof(ta_match), hipMemcpyDeviceToHost, st));
        TA_HIP(hipStreamSynchronize(st));
        for (const ta_match &h : all)
            if (h.k == host[1]) v.push_back(h);
    }
    return TA_OK;
}

static void sort_by_end(std::vector<ta_match> &v) {
    std::sort(v.begin(), v.end(), [](const ta_match &x, const ta_match &y) { return x.end != y.end ? x.end < y.end : x.start < y.start; });
}

// One search pass over a shard resident in HBM.  best == nullptr: All-mode hits into hits_dev, their number into *count_host.
// best != nullptr: additionally the hits with the smallest k (the only ones the Best fold can keep), sorted by end.
//
// Unit-cost families with a short needle: a bit-parallel scan (lev_filter_body.h) flags the 64-column blocks that hold a cost
// <= k and only those go through the exact kernel.  For needles of up to 64 bytes the whole pass is ONE fill, TWO kernels and
// ONE stream synchronisation: the exact kernel (one wavefront per flagged block, lev_search_wave_body.h) reads the number of
// flagged blocks on the device, its last workgroup selects the best hits and writes the report into host-mapped memory.
static int search_dev_core(const uint8_t *needle_host, size_t needle_len, const uint8_t *haystack_dev, size_t haystack_len,
                           uint32_t k, const ta_edit_costs *costs, int anchored, uint64_t base, uint64_t emit_from,
                           ta_match *hits_dev, size_t cap, uint64_t *count_host, std::vector<ta_match> *best, hipStream_t st) {
    if (!search_costs_ok(costs)) return TA_ERR_BAD_COSTS;
    if (ta_edit_costs_check_search(costs) != TA_OK) return TA_ERR_BAD_COSTS;         // :1965
    if (needle_len == 0) { set_last_error_msg("empty needle is handled by the host entry point"); return TA_ERR_ARG; }
    if (needle_len > 0xFFFFu) { set_last_error_msg("needle longer than 65535 bytes"); return TA_ERR_ARG; }
    if (!device_ready()) return TA_ERR_HIP;
    StreamGuard guard(st);
    Scratch &cnt = tls_scratch(2);
    int rc = cnt.ensure(sizeof(SearchCtl) + 64);
    if (rc) return rc;
    SearchParams P;
    size_t h = haystack_len;
    const uint32_t unit_k = lev_sat_sub(k, costs->start_gap_cost) / costs->gap_cost;
    if (anchored) {                                                                    // :1650-1658
        uint64_t lim = (uint64_t)needle_len + unit_k;
        if (lim < h) h = (size_t)lim;
    }
    fill_params(P, needle_host, needle_len, haystack_dev, h, k, costs, anchored, base, emit_from, hits_dev, cap,
                (unsigned long long *)cnt.dev);
    uint64_t halo64 = (uint64_t)needle_len + unit_k + 2;
    if (halo64 > 0x7FFFFFFFull) halo64 = 0x7FFFFFFFull;
    P.halo = (uint32_t)halo64;
    P.tile = anchored ? 0x7FFFFFFFu : pick_tile(h, P.halo);
    if (anchored) P.halo = 0;
    if (needle_len > 32) {
        // memory-backed column: needle on the device, 6 arrays of (n+1) u32 per tile; keep the scratch <= ~256 MB
        Scratch &nd = tls_scratch(7);
        if ((rc = nd.ensure(needle_len + 16))) return rc;
        TA_HIP(hipMemcpyAsync(nd.dev, needle_host, needle_len, hipMemcpyHostToDevice, st));
        P.needle_dev = (const uint8_t *)nd.dev;
    }
    auto mem_column_scratch = [&]() -> int {           // the lane-per-tile kernel over everything, needle > 32 bytes
        Scratch &cs = tls_scratch(6);
        const uint64_t per_tile = 6ull * (needle_len + 1) * 4ull;
        uint64_t max_tiles = (256ull << 20) / per_tile;
        if (max_tiles < 64) max_tiles = 64;
        if (!anchored) {
            uint64_t t = (h + max_tiles - 1) / max_tiles;
            if (t > P.tile) P.tile = (uint32_t)(t > 0x7FFFFFFFull ? 0x7FFFFFFFull : t);
        }
        const uint64_t tiles = (h + P.tile - 1) / P.tile;
        int r = cs.ensure((size_t)(per_tile * (tiles ? tiles : 1)));
        if (r) return r;
        P.col_scratch = (uint32_t *)cs.dev;
        return TA_OK;
    };
    // packed cost/length kernel whenever every cost and length provably fits 16 bits
    bool packed = needle_len <= 32 && k <= 30000u && (uint64_t)P.tile + P.halo <= 60000u && !env_str("TA_SEARCH_UNPACKED");
    if (anchored)           // every cost must stay below the packed form's "no gap yet" marker (lev_search_body.h)
        packed = needle_len <= 32 && k <= 30000u && h <= 60000u && !env_str("TA_SEARCH_UNPACKED") &&
                 srch_anchored_packed_ok(h, (uint32_t)needle_len, costs->mismatch_cost, costs->gap_cost, costs->start_gap_cost);
    const bool unit = costs->mismatch_cost == 1 && costs->gap_cost == 1 && costs->start_gap_cost == 0 &&
                      (!costs->has_transpose || costs->transpose_cost == 1);
    const bool trans = costs->has_transpose != 0;
    // The filter scans with unit costs; under any other EditCosts it runs with k' = srch_filter_k (lev_search_body.h): a superset
    // filter -- every alignment of weighted cost <= k has at most k' unit edits -- in front of the exact kernel, which knows the
    // real costs.  With k' >= needle_len every position matches: no filter.  (TA_SEARCH_NOWFILTER=1: unit costs only, round 4's rule.)
    const uint32_t kf = unit ? k : srch_filter_k(k, costs->mismatch_cost, costs->gap_cost, costs->start_gap_cost, trans, costs->transpose_cost);
    const bool filter_ok = (unit || !env_str("TA_SEARCH_NOWFILTER")) && !anchored && needle_len <= 512 && kf < needle_len && h >= 4096 &&
                           !env_str("TA_SEARCH_NOFILTER");
    bool searched = false;
    unsigned long long c = 0;
    if (filter_ok) {
        Scratch &ls = tls_scratch(5);
        uint64_t cap_list = h / FILTER_BLOCK + 2;
        if (cap_list > (4u << 20)) cap_list = 4u << 20;
        if ((rc = ls.ensure((size_t)cap_list * 4))) return rc;
        SearchCtl *ctl = (SearchCtl *)cnt.dev;
        TA_HIP(hipMemsetAsync(ctl, 0, sizeof(SearchCtl), st));           // the pass's one fill
        P.count = &ctl->count;
        SearchParams F = P;
        F.k = kf;
        if (F.halo < (uint32_t)needle_len + kf + 2) F.halo = (uint32_t)needle_len + kf + 2;       // left context of a unit-cost match of k' edits
        uint64_t ft = (h + 524287) / 524288;                          // two sets of resident lanes (256 CUs x 16 waves x 64): 0.557 against 0.601 ms
                                                                      // per GiB with one set of 8 waves per CU (profiles/r03/ab_search.md)
        if (ft < 4 * (uint64_t)F.halo) ft = 4 * (uint64_t)F.halo;    // keep the left-context overhead under 25 %
        ft = (ft + 2 * FILTER_BLOCK - 1) / (2 * FILTER_BLOCK) * (2 * FILTER_BLOCK);      // whole 128-byte lines per lane
        if (const char *e = env_str("TA_FILTER_TILE")) { long v = atol(e); if (v >= 64) ft = (uint64_t)v / FILTER_BLOCK * FILTER_BLOCK; }
        F.tile = (uint32_t)(ft > 0x7FFFFFC0ull ? 0x7FFFFFC0ull : ft);
        TA_HIP(lev_filter_launch(F, trans, (uint32_t *)ls.dev, (uint32_t)cap_list, &ctl->n_list, st));
        if (needle_len <= 64 && k <= 30000u) {
            // one wavefront per flagged block; no host round trip: the report arrives with the stream synchronisation
            PinBox &box = search_report_box();
            Scratch &cd = tls_scratch(12);
            if ((rc = box.ensure()) || (best && (rc = cd.ensure((size_t)SEARCH_SLOT_CAP * sizeof(SearchSlot))))) return rc;
            TA_HIP(lev_search_wave_launch(P, trans, best != nullptr, (const uint32_t *)ls.dev, (uint32_t)cap_list, ctl, (SearchSlot *)cd.dev,
                                          box.dev, st));
            TA_HIP(hipStreamSynchronize(st));
            const SearchReport *rep = (const SearchReport *)box.host;
            if (!rep->dense) {
                searched = true;
                c = rep->count;
                if (best && c <= cap && rep->sel_state == 1) {
                    const ta_match *sel = (const ta_match *)(box.host + sizeof(SearchReport));
                    best->assign(sel, sel + rep->sel_count);
                    sort_by_end(*best);
                    *count_host = c;
                    return TA_OK;
                }
            }
        } else {
            // needles beyond the wavefront kernel: the memory-backed column, one lane per flagged block (needs the count here)
            unsigned int n_list = 0;
            TA_HIP(hipMemcpyAsync(&n_list, &ctl->n_list, 4, hipMemcpyDeviceToHost, st));
            TA_HIP(hipStreamSynchronize(st));
            // dense matches: the exact kernel over everything is cheaper than (64 + halo) columns per flagged block
            if (n_list <= cap_list && (uint64_t)n_list * (FILTER_BLOCK + P.halo) < h / 2) {
                if (n_list) {                                            // one memory-backed column per flagged block
                    Scratch &cs = tls_scratch(6);
                    if ((rc = cs.ensure((size_t)(6ull * (needle_len + 1) * 4ull * n_list)))) return rc;
                    P.col_scratch = (uint32_t *)cs.dev;
                }
                TA_HIP(lev_search_list_launch(P, trans, (const uint32_t *)ls.dev, n_list, st));
                TA_HIP(hipMemcpyAsync(&c, &ctl->count, 8, hipMemcpyDeviceToHost, st));
                TA_HIP(hipStreamSynchronize(st));
                searched = true;
            }
        }
    }
    if (!searched) {                                                      // the lane-per-tile kernel over everything
        if (needle_len > 32 && (rc = mem_column_scratch())) return rc;
        P.count = (unsigned long long *)cnt.dev;
        TA_HIP(hipMemsetAsync(cnt.dev, 0, 8, st));
        TA_HIP(lev_search_launch(P, packed, trans, st));
        TA_HIP(hipMemcpyAsync(&c, cnt.dev, 8, hipMemcpyDeviceToHost, st));
        TA_HIP(hipStreamSynchronize(st));
    }
    *count_host = c;
    if (c > cap) return TA_ERR_CAPACITY;
    if (best) {
        if ((rc = best_hits_of(hits_dev, c, *best, st))) return rc;
        sort_by_end(*best);
    }
    return TA_OK;
}

extern "C" {

int ta_levenshtein_search_dev(const uint8_t *needle_host, size_t needle_len,
                              const uint8_t *haystack_dev, size_t haystack_len,
                              uint32_t k, const ta_edit_costs *costs, int anchored,
                              uint64_t base, uint64_t emit_from,
                              ta_match *hits_dev, size_t cap, uint64_t *count_host, void *stream) {
    if (!count_host || (!needle_host && needle_len) || (!haystack_dev && haystack_len)) return TA_ERR_ARG;
    return search_dev_core(needle_host, needle_len, haystack_dev, haystack_len, k, costs, anchored, base, emit_from, hits_dev, cap,
                           count_host, nullptr, (hipStream_t)stream);
}

// (measurement aid, not part of the ABI header) the report of this thread's last filtered search pass, as 16 dwords
int ta_debug_last_search_report(uint32_t *out16) {
    PinBox &box = search_report_box();
    if (!box.host || !out16) return TA_ERR_ARG;
    memcpy(out16, box.host, 64);
    return TA_OK;
}

// The Best-mode pass over a shard: the All-mode hits stay in hits_dev, only the ones with the smallest k come back.
int ta_levenshtein_search_best_dev(const uint8_t *needle_host, size_t needle_len,
                                   const uint8_t *haystack_dev, size_t haystack_len,
                                   uint32_t k, const ta_edit_costs *costs, uint64_t base, uint64_t emit_from,
                                   ta_match *hits_dev, size_t cap, uint64_t *count_host, ta_match **out, size_t *n_out, void *stream) {
    if (!out || !n_out || !count_host || (!needle_host && needle_len) || (!haystack_dev && haystack_len)) return TA_ERR_ARG;
    *out = nullptr; *n_out = 0;
    std::vector<ta_match> v;
    int rc = search_dev_core(needle_host, needle_len, haystack_dev, haystack_len, k, costs, 0, base, emit_from, hits_dev, cap,
                             count_host, &v, (hipStream_t)stream);
    if (rc) return rc;
    *n_out = v.size();
    if (!v.empty()) {
        *out = (ta_match *)malloc(v.size() * sizeof(ta_match));
        if (!*out) return TA_ERR_ARG;
        memcpy(*out, v.data(), v.size() * sizeof(ta_match));
    }
    return TA_OK;
}

}  // extern "C"

// shared by the SIMD-contract entry (NUL bytes in the haystack are an error, src/hamming.rs:463) and the naive-contract one.
// host_hits (optional): the hits, unsorted, straight from the report box when they fit it (SEARCH_REPORT_SEL records; *in_box says so) --
// the caller need not copy them from hits_dev then.
static int hamming_search_dev_impl(const uint8_t *needle_host, size_t needle_len,
                                   const uint8_t *haystack_dev, size_t haystack_len, uint32_t k,
                                   uint64_t base, ta_match *hits_dev, size_t cap, uint64_t *count_host, void *stream, bool check_nul,
                                   std::vector<ta_match> *host_hits = nullptr, bool *in_box = nullptr) {
    if (!count_host || (!needle_host && needle_len) || (!haystack_dev && haystack_len)) return TA_ERR_ARG;
    if (!device_ready()) return TA_ERR_HIP;
    hipStream_t st = (hipStream_t)stream;
    *count_host = 0;
    if (in_box) *in_box = false;
    if (needle_len == 0 || needle_len > haystack_len) return TA_OK;                     // src/hamming.rs:455-461
    StreamGuard guard(st);
    Scratch &cnt = tls_scratch(2);
    PinBox &box = search_report_box();
    int rc = cnt.ensure(16);
    if (rc || (rc = box.ensure())) return rc;
    TA_HIP(fill_u32_launch((uint32_t *)cnt.dev, 0u, 4, st));
    SearchParams P;
    fill_params(P, needle_host, needle_len, haystack_dev, haystack_len, k, nullptr, 0, base, 0, hits_dev, cap,
                (unsigned long long *)cnt.dev);
    // needles of up to 64 bytes travel in the kernel arguments (SearchParams::needle): no upload
    // (the round-1 SWAR kernel -- needles beyond 64 bytes the phased filter does not take, or TA_HAMMING_SEARCH_SWAR=1 -- reads the device copy)
    P.needle_dev = nullptr;
    if (needle_len > 64 || env_str("TA_HAMMING_SEARCH_SWAR") || (needle_len > 32 && tuning_enabled())) {   // (any A/B switch may route to the kernel that reads the device copy)
        Scratch &nd = tls_scratch(7);
        if ((rc = nd.ensure(needle_len + 16))) return rc;
        TA_HIP(hipMemcpyAsync(nd.dev, needle_host, needle_len, hipMemcpyHostToDevice, st));
        P.needle_dev = (const uint8_t *)nd.dev;
    }
    // the NUL-byte scan of the SIMD contract (:463) rides inside the search kernel where that kernel reads every byte anyway
    uint32_t *nul_flag = (uint32_t *)((uint8_t *)cnt.dev + 8);
    bool nul_done = false;
    TA_HIP(hamming_search_launch(P, st, check_nul ? nul_flag : nullptr, &nul_done));
    if (check_nul && !nul_done) TA_HIP(has_zero_byte_launch(haystack_dev, haystack_len, nul_flag, st));
    // count, NUL flag and (few) hits into host-mapped memory: one synchronisation delivers them
    TA_HIP(search_report_copy_launch((const unsigned long long *)cnt.dev, nul_flag, hits_dev, cap, box.dev, st));
    TA_HIP(hipStreamSynchronize(st));
    const SearchReport *rep = (const SearchReport *)box.host;
    if (check_nul && rep->dense) return TA_ERR_NULL_BYTE;
    *count_host = rep->count;
    if (rep->count > cap) return TA_ERR_CAPACITY;
    if (host_hits && rep->sel_state == 1) {
        const ta_match *sel = (const ta_match *)(box.host + sizeof(SearchReport));
        host_hits->assign(sel, sel + rep->sel_count);
        if (in_box) *in_box = true;
    }
    return TA_OK;
}

namespace ta {
int hamming_search_dev_nocheck(const uint8_t *needle_host, size_t needle_len, const uint8_t *haystack_dev, size_t haystack_len, uint32_t k,
                               uint64_t base, ta_match *hits_dev, size_t cap, uint64_t *count_host, void *stream) {
    return hamming_search_dev_impl(needle_host, needle_len, haystack_dev, haystack_len, k, base, hits_dev, cap, count_host, stream, false);
}
}  // namespace ta

extern "C" {

int ta_hamming_search_dev(const uint8_t *needle_host, size_t needle_len,
                          const uint8_t *haystack_dev, size_t haystack_len, uint32_t k,
                          uint64_t base, ta_match *hits_dev, size_t cap, uint64_t *count_host, void *stream) {
    return hamming_search_dev_impl(needle_host, needle_len, haystack_dev, haystack_len, k, base, hits_dev, cap, count_host, stream, true);
}

/* ta_hamming_search_dev with the hits on the HOST, sorted by end (library-allocated: ta_free): the few hits of an ordinary search arrive
 * through host-mapped memory with the call's one stream synchronisation; more than the report box holds are copied from hits_dev. */
int ta_hamming_search_dev_sorted(const uint8_t *needle_host, size_t needle_len, const uint8_t *haystack_dev, size_t haystack_len, uint32_t k,
                                 uint64_t base, ta_match *hits_dev, size_t cap, ta_match **out, size_t *n_out, void *stream) {
    if (!out || !n_out) return TA_ERR_ARG;
    *out = nullptr; *n_out = 0;
    std::vector<ta_match> v;
    bool in_box = false;
    uint64_t count = 0;
    int rc = hamming_search_dev_impl(needle_host, needle_len, haystack_dev, haystack_len, k, base, hits_dev, cap, &count, stream, true, &v, &in_box);
    if (rc) return rc;
    if (!in_box && count) {
        v.resize(count);
        TA_HIP(hipMemcpy(v.data(), hits_dev, count * sizeof(ta_match), hipMemcpyDeviceToHost));
    }
    sort_by_end(v);
    *n_out = v.size();
    if (!v.empty()) {
        *out = (ta_match *)malloc(v.size() * sizeof(ta_match));
        if (!*out) return TA_ERR_ARG;
        memcpy(*out, v.data(), v.size() * sizeof(ta_match));
    }
    return TA_OK;
}

// The hits of a device-resident All-mode result that can survive the Best fold -- those with the smallest k -- sorted by end.
int ta_search_best_hits_dev(const ta_match *hits_dev, uint64_t count, ta_match **out, size_t *n_out, void *stream) {
    if (!out || !n_out || (!hits_dev && count)) return TA_ERR_ARG;
    *out = nullptr; *n_out = 0;
    if (count == 0) return TA_OK;
    if (!device_ready()) return TA_ERR_HIP;
    hipStream_t st = (hipStream_t)stream;
    StreamGuard guard(st);
    std::vector<ta_match> v;
    int rc = best_hits_of(hits_dev, count, v, st);
    if (rc) return rc;
    sort_by_end(v);
    *n_out = v.size();
    if (!v.empty()) {
        *out = (ta_match *)malloc(v.size() * sizeof(ta_match));
        if (!*out) return TA_ERR_ARG;
        memcpy(*out, v.data(), v.size() * sizeof(ta_match));
    }
    return TA_OK;
}

// src/levenshtein.rs:1792-1796 + :1812-1835 (levenshtein) / src/hamming.rs:122-143 (overlap_fold = 0)
size_t ta_search_fold_best(ta_match *hits, size_t n, uint32_t k, int overlap_fold) {
    uint32_t curr_k = k;
    size_t w = 0;
    for (size_t r = 0; r < n; r++) {
        if (hits[r].k > curr_k) continue;          // emitted only if res <= running curr_k
        curr_k = hits[r].k;
        if (overlap_fold && w > 0 && hits[r].start <= hits[w - 1].start) hits[w - 1] = hits[r];   // fully overlapping: replace
        else hits[w++] = hits[r];
    }
    size_t o = 0;
    for (size_t r = 0; r < w; r++)
        if (hits[r].k == curr_k) hits[o++] = hits[r];
    return o;
}

}  // extern "C"

static int give(std::vector<ta_match> &v, ta_match **out, size_t *n_out) {
    *n_out = v.size();
    *out = nullptr;
    if (!v.empty()) {
        *out = (ta_match *)malloc(v.size() * sizeof(ta_match));
        if (!*out) return TA_ERR_ARG;
        memcpy(*out, v.data(), v.size() * sizeof(ta_match));
    }
    return TA_OK;
}

// stage a haystack into device scratch (with read slack) and collect the All-mode hits, sorted by end; everything runs on
// the calling thread's own stream.  The hit buffer starts at min(haystack_len + 2, 4M) records; a denser result (All mode
// over a big haystack, k >= needle_len) reports its true count, and the pass is repeated once with room for exactly that.
// What ta_levenshtein_search_first left in this thread's haystack staging buffer (tls_scratch(TA_SLOT_SEARCH_HAY)): the first `upto` bytes of the caller's
// haystack.  ta_levenshtein_search_resume -- "the rest of the All-mode result over THE SAME haystack" -- uploads only what is missing.
struct ResidentHay { const uint8_t *host = nullptr; size_t len = 0; uint64_t upto = 0; const void *dev = nullptr; uint8_t fp[128] = {}; };
// a content fingerprint of the uploaded prefix (its first and last 64 bytes): a caller that reuses the pointer for other bytes -- the C header
// states the promise, only the Rust borrow enforces it -- degrades ta_levenshtein_search_resume to the plain call instead of wrong matches
static void hay_fingerprint(const uint8_t *h, uint64_t upto, uint8_t *fp) {
    memset(fp, 0, 128);
    const size_t n = upto < 64 ? (size_t)upto : 64;
    memcpy(fp, h, n);
    memcpy(fp + 64, h + upto - n, n);
}
static ResidentHay &resident_hay() { static thread_local ResidentHay r; return r; }
namespace ta { void search_resident_reset() { resident_hay() = ResidentHay{}; } }     // (ta_thread_release: the staging buffer is gone)
static thread_local uint64_t g_resume_upto = 0;      // set by ta_levenshtein_search_resume around its call of the full search

template <class Launch>
// resident_upto: bytes [0, resident_upto) of the haystack are already in that buffer (0: none)
static int run_search_host(const uint8_t *haystack, size_t haystack_len, std::vector<ta_match> &hits, Launch launch, uint64_t resident_upto = 0) {
    if (!device_ready()) return TA_ERR_HIP;
    CallCtx &cx = call_ctx();
    int rc = cx.ensure();
    if (rc) return rc;
    // a short haystack goes through the thread's pinned, device-mapped buffer like a short pair does (ta_api.hip): the kernels
    // read it in place and write their hits next to it -- no staging copy in, no copy of the hits out.  Room for the haystack
    // and at least 64 hits; a denser result falls through to the general path below.
    const size_t hay_pad = (haystack_len + TA_BLOB_SLACK + 255) & ~(size_t)255;
    if (!resident_upto && hay_pad + 64 * sizeof(ta_match) <= CallCtx::RESULT_OFF) {
        const size_t pcap = (CallCtx::RESULT_OFF - hay_pad) / sizeof(ta_match);
        if (haystack_len) memcpy(cx.pin, haystack, haystack_len);
        memset(cx.pin + haystack_len, 0, TA_BLOB_SLACK);
        uint64_t count = 0;
        rc = launch((const uint8_t *)cx.pin_dev, (ta_match *)(cx.pin_dev + hay_pad), pcap, &count, cx.st);   // synchronises
        if (rc == TA_OK) {
            const ta_match *h = (const ta_match *)(cx.pin + hay_pad);
            hits.assign(h, h + count);
            std::sort(hits.begin(), hits.end(), [](const ta_match &x, const ta_match &y) {
                return x.end != y.end ? x.end < y.end : x.start < y.start;
            });
            return TA_OK;
        }
        if (rc != TA_ERR_CAPACITY) return rc;
    }
    Scratch &hs = tls_scratch(TA_SLOT_SEARCH_HAY), &ob = tls_scratch(1);
    if (hs.cap < haystack_len + TA_BLOB_SLACK + 64) resident_upto = 0;                          // (the buffer is about to be re-allocated)
    if ((rc = hs.ensure(haystack_len + TA_BLOB_SLACK + 64))) return rc;
    size_t cap = haystack_len + 2;
    if (cap > (1u << 22)) cap = (1u << 22);
    if ((rc = ob.ensure(cap * sizeof(ta_match)))) return rc;
    if (resident_upto > haystack_len || resident_hay().dev != hs.dev) resident_upto = 0;       // (the buffer was re-allocated: nothing is resident)
    if (resident_upto && env_int("TA_DEBUG")) fprintf(stderr, "[triple_accel_amd] search resume: %llu of %zu haystack bytes already resident\n", (unsigned long long)resident_upto, haystack_len);
    if (haystack_len > resident_upto)
        TA_HIP(hipMemcpyAsync((uint8_t *)hs.dev + resident_upto, haystack + resident_upto, haystack_len - resident_upto, hipMemcpyHostToDevice, cx.st));
    resident_hay() = ResidentHay{};                                // (whatever runs next starts from scratch)
    uint64_t count = 0;
    rc = launch((const uint8_t *)hs.dev, (ta_match *)ob.dev, cap, &count, cx.st);
    if (rc == TA_ERR_CAPACITY && count > cap && count <= (uint64_t)haystack_len + 2) {
        cap = (size_t)count;
        if ((rc = ob.ensure(cap * sizeof(ta_match)))) return rc;
        rc = launch((const uint8_t *)hs.dev, (ta_match *)ob.dev, cap, &count, cx.st);
    }
    if (rc) return rc;
    hits.resize(count);
    if (count) {
        TA_HIP(hipMemcpyAsync(hits.data(), ob.dev, count * sizeof(ta_match), hipMemcpyDeviceToHost, cx.st));
        TA_HIP(hipStreamSynchronize(cx.st));
    }
    std::sort(hits.begin(), hits.end(), [](const ta_match &x, const ta_match &y) {
        return x.end != y.end ? x.end < y.end : x.start < y.start;
    });
    return TA_OK;
}

extern "C" {

int ta_levenshtein_search_simd_with_opts(const uint8_t *needle, size_t needle_len,
                                         const uint8_t *haystack, size_t haystack_len,
                                         uint32_t k, int search_type, const ta_edit_costs *costs, int anchored,
                                         ta_match **out, size_t *n_out) {
    if (!out || !n_out || (!needle && needle_len) || (!haystack && haystack_len)) return TA_ERR_ARG;
    *out = nullptr; *n_out = 0;
    if (!search_costs_ok(costs)) return TA_ERR_BAD_COSTS;
    std::vector<ta_match> res;
    if (needle_len == 0) {                                                 // src/levenshtein.rs:1919-1963
        if (!anchored) return TA_OK;
        res.push_back(ta_match{0, 0, 0, 0});
        if (search_type == TA_SEARCH_ALL) {
            uint32_t cost = costs->start_gap_cost;
            for (size_t i = 0; i < haystack_len;) {
                i += 1;
                cost += costs->gap_cost;
                if (cost <= k) res.push_back(ta_match{0, i, cost, 0}); else break;
            }
        }
        return give(res, out, n_out);
    }
    if (ta_edit_costs_check_search(costs) != TA_OK) return TA_ERR_BAD_COSTS;   // :1965
    std::vector<ta_match> hits;
    int rc;
    // a big haystack and more than one device in the set (ta_set_devices; default: every visible one): contiguous shards, one per device,
    // each behind needle_len + unit_k + 2 bytes of left context, uploaded and searched side by side (ta_multi.hip).  An anchored search only
    // reads the haystack's first needle_len + unit_k bytes: one device.
    if (!anchored && needle_len <= 65535 && multi_search_shards(haystack_len) > 1) {
        if (!device_ready()) return TA_ERR_HIP;
        rc = multi_levenshtein_search_host(needle, needle_len, haystack, haystack_len, k, search_type == TA_SEARCH_BEST, costs, hits);
    } else
    rc = run_search_host(haystack, haystack_len, hits, [&](const uint8_t *hd, ta_match *od, size_t cap, uint64_t *cnt, hipStream_t st) {
        if (haystack_len == 0) { *cnt = 0; return (int)TA_OK; }
        return ta_levenshtein_search_dev(needle, needle_len, hd, haystack_len, k, costs, anchored, 0, 0, od, cap, cnt, st);
    }, g_resume_upto);
    if (rc) return rc;
    // the match that ends before the first haystack byte (:1693-1706, SIMD :2394-2399)
    const uint32_t whole_gap = (uint32_t)needle_len * costs->gap_cost + costs->start_gap_cost;
    if (whole_gap <= k) res.push_back(ta_match{0, 0, whole_gap, 0});
    res.insert(res.end(), hits.begin(), hits.end());
    if (search_type == TA_SEARCH_BEST) res.resize(ta_search_fold_best(res.data(), res.size(), k, 1));
    return give(res, out, n_out);
}

}  // extern "C"

// windows of a resident haystack until one holds a hit; `upload(lo, hi)` makes bytes [lo, hi) present (host form) or is a no-op
template <class Upload>
static int first_hit_windows(const uint8_t *needle, size_t needle_len, const uint8_t *hay_dev, size_t h, uint32_t k,
                             const ta_edit_costs *costs, uint64_t base, ta_match *out, int *found, hipStream_t st, Upload upload) {
    *found = 0;
    const uint32_t unit_k = lev_sat_sub(k, costs->start_gap_cost) / costs->gap_cost;
    const uint64_t halo = (uint64_t)needle_len + unit_k + 2;
    Scratch &ob = tls_scratch(1);
    const size_t cap = 1u << 16;                                  // hits kept per window; a denser window is cut down (below)
    int rc = ob.ensure(cap * sizeof(ta_match));
    if (rc) return rc;
    std::vector<ta_match> hits;
    uint64_t lo = 0, chunk = 64u << 10, uploaded = 0;
    while (lo < h) {
        const uint64_t hi = lo + chunk < h ? lo + chunk : h, ctx = lo > halo ? lo - halo : 0;
        if (hi > uploaded) { if ((rc = upload(uploaded, hi))) return rc; uploaded = hi; }
        uint64_t count = 0;
        // the window [ctx, hi) as a haystack of its own: a DP started fresh `halo` bytes before `lo` is exact for every cost <= k
        rc = search_dev_core(needle, needle_len, hay_dev + ctx, (size_t)(hi - ctx), k, costs, 0, base + ctx, base + lo,
                             (ta_match *)ob.dev, cap, &count, nullptr, st);
        if (rc == TA_ERR_CAPACITY) {                              // more hits than the buffer holds: the first one is in a shorter window
            if (hi - lo <= 2048) return rc;                       // (cannot happen: a window emits at most one hit per end position)
            chunk = (hi - lo) / 16 < 2048 ? 2048 : (hi - lo) / 16;
            continue;
        }
        if (rc) return rc;
        if (count) {
            hits.resize((size_t)count);
            TA_HIP(hipMemcpyAsync(hits.data(), ob.dev, (size_t)count * sizeof(ta_match), hipMemcpyDeviceToHost, st));
            TA_HIP(hipStreamSynchronize(st));
            const ta_match *best = &hits[0];
            for (const ta_match &m : hits)
                if (m.end < best->end || (m.end == best->end && m.start < best->start)) best = &m;
            *out = *best;
            *found = 1;
            return TA_OK;
        }
        lo = hi;
        if (chunk < (256u << 20)) chunk *= 4;
    }
    return TA_OK;
}

extern "C" {

int ta_levenshtein_search_first_dev(const uint8_t *needle_host, size_t needle_len,
                                    const uint8_t *haystack_dev, size_t haystack_len,
                                    uint32_t k, const ta_edit_costs *costs, uint64_t base,
                                    ta_match *out, int *found, void *stream) {
    if (!out || !found || (!needle_host && needle_len) || (!haystack_dev && haystack_len)) return TA_ERR_ARG;
    *found = 0;
    if (!search_costs_ok(costs) || ta_edit_costs_check_search(costs) != TA_OK) return TA_ERR_BAD_COSTS;
    if (needle_len == 0 || haystack_len == 0) return TA_OK;
    return first_hit_windows(needle_host, needle_len, haystack_dev, haystack_len, k, costs, base, out, found, (hipStream_t)stream,
                             [](uint64_t, uint64_t) { return (int)TA_OK; });
}

int ta_levenshtein_search_first(const uint8_t *needle, size_t needle_len, const uint8_t *haystack, size_t haystack_len,
                                uint32_t k, const ta_edit_costs *costs, int anchored, ta_match *out, int *found) {
    if (!out || !found || (!needle && needle_len) || (!haystack && haystack_len)) return TA_ERR_ARG;
    *found = 0;
    resident_hay() = ResidentHay{};                                            // (set again below if this call uploads a prefix)
    if (!search_costs_ok(costs)) return TA_ERR_BAD_COSTS;
    if (needle_len == 0) {                                                      // src/levenshtein.rs:1919-1963
        if (anchored) { *out = ta_match{0, 0, 0, 0}; *found = 1; }
        return TA_OK;
    }
    if (ta_edit_costs_check_search(costs) != TA_OK) return TA_ERR_BAD_COSTS;     // :1965
    if (needle_len > 65535) { set_last_error_msg("needle longer than 65535 bytes"); return TA_ERR_ARG; }   // (before any 32-bit arithmetic on it)
    if (!device_ready()) return TA_ERR_HIP;
    const uint32_t whole_gap = (uint32_t)needle_len * costs->gap_cost + costs->start_gap_cost;
    if (whole_gap <= k) { *out = ta_match{0, 0, whole_gap, 0}; *found = 1; return TA_OK; }     // the end == 0 match comes first (:1693-1706)
    if (haystack_len == 0) return TA_OK;
    if (anchored) {                                                             // a prefix of needle_len + unit_k bytes: one small pass
        ta_match *all = nullptr;
        size_t n_all = 0;
        int rc = ta_levenshtein_search_simd_with_opts(needle, needle_len, haystack, haystack_len, k, TA_SEARCH_ALL, costs, 1, &all, &n_all);
        if (rc) return rc;
        if (n_all) { *out = all[0]; *found = 1; }
        free(all);
        return TA_OK;
    }
    if (!device_ready()) return TA_ERR_HIP;
    CallCtx &cx = call_ctx();
    int rc = cx.ensure();
    if (rc) return rc;
    Scratch &hs = tls_scratch(TA_SLOT_SEARCH_HAY);
    if ((rc = hs.ensure(haystack_len + TA_BLOB_SLACK + 64))) return rc;
    uint8_t *hd = (uint8_t *)hs.dev;
    hipStream_t st = cx.st;
    uint64_t upto = 0;
    rc = first_hit_windows(needle, needle_len, hd, haystack_len, k, costs, 0, out, found, st, [&](uint64_t from, uint64_t to) -> int {
        TA_HIP(hipMemcpyAsync(hd + from, haystack + from, (size_t)(to - from), hipMemcpyHostToDevice, st));   // only as far as the scan gets
        if (from == upto) upto = to;                               // (the windows' uploads are consecutive)
        return (int)TA_OK;
    });
    resident_hay() = ResidentHay{};
    if (rc == TA_OK && upto) {
        ResidentHay &r = resident_hay();
        r.host = haystack; r.len = haystack_len; r.upto = upto; r.dev = hs.dev;
        hay_fingerprint(haystack, upto, r.fp);
    }
    return rc;
}

/* The All-mode result of ta_levenshtein_search_simd_with_opts for a caller that has just taken its first element with
 * ta_levenshtein_search_first ON THE SAME HAYSTACK (same pointer, same length, unchanged bytes -- the caller's promise; the bindings' lazy
 * iterators hold the haystack borrowed / immutable): the bytes that call uploaded stay where they are, only the rest of the haystack
 * travels, then the whole search runs on the resident copy.  Anything else in between on this thread, a different haystack or a
 * re-allocated staging buffer make it the plain call.  (src/levenshtein.rs:2282-2420: the reference's iterator continues where it stopped.) */
int ta_levenshtein_search_resume(const uint8_t *needle, size_t needle_len, const uint8_t *haystack, size_t haystack_len,
                                 uint32_t k, const ta_edit_costs *costs, int anchored, ta_match **out, size_t *n_out) {
    const ResidentHay r = resident_hay();
    uint64_t upto = (r.host == haystack && r.len == haystack_len && haystack_len) ? r.upto : 0;
    if (upto) {                                                                  // same pointer, same length -- and still the same bytes?
        uint8_t fp[128];
        hay_fingerprint(haystack, upto, fp);
        if (memcmp(fp, r.fp, 128) != 0) upto = 0;
    }
    g_resume_upto = upto;
    const int rc = ta_levenshtein_search_simd_with_opts(needle, needle_len, haystack, haystack_len, k, TA_SEARCH_ALL, costs, anchored, out, n_out);
    g_resume_upto = 0;
    return rc;
}

int ta_levenshtein_search(const uint8_t *needle, size_t needle_len, const uint8_t *haystack, size_t haystack_len,
                          ta_match **out, size_t *n_out) {                     // src/levenshtein.rs:1866-1878
    ta_edit_costs c = ta_levenshtein_costs();
    uint32_t k = (uint32_t)(needle_len >> 1) + ((uint32_t)needle_len & 1u);
    return ta_levenshtein_search_simd_with_opts(needle, needle_len, haystack, haystack_len, k, TA_SEARCH_BEST, &c, 0,
                                                out, n_out);
}

int ta_hamming_search_simd_with_opts(const uint8_t *needle, size_t needle_len,
                                     const uint8_t *haystack, size_t haystack_len,
                                     uint32_t k, int search_type, ta_match **out, size_t *n_out) {
    if (!out || !n_out || (!needle && needle_len) || (!haystack && haystack_len)) return TA_ERR_ARG;
    *out = nullptr; *n_out = 0;
    if (needle_len > haystack_len) return TA_OK;                                // src/hamming.rs:455-457
    if (needle_len == 0) return TA_OK;                                          // :459-461
    std::vector<ta_match> hits;
    int rc;
    if (multi_search_shards(haystack_len) > 1) {                                // the windows that start in a shard: its bytes + needle_len - 1 more
        if (!device_ready()) return TA_ERR_HIP;
        rc = multi_hamming_search_host(needle, needle_len, haystack, haystack_len, k, true, hits);
    } else
    rc = run_search_host(haystack, haystack_len, hits, [&](const uint8_t *hd, ta_match *od, size_t cap, uint64_t *cnt, hipStream_t st) {
        return ta_hamming_search_dev(needle, needle_len, hd, haystack_len, k, 0, od, cap, cnt, st);
    });
    if (rc) return rc;
    if (search_type == TA_SEARCH_BEST) hits.resize(ta_search_fold_best(hits.data(), hits.size(), k, 0));
    return give(hits, out, n_out);
}

/* hamming_search_naive_with_opts (src/hamming.rs:96-146): the same scan under the scalar routine's contract -- NUL bytes in the
 * haystack are fine, and an empty needle matches (with k = 0) at every offset 0..=haystack_len in All mode; in Best mode the
 * reference divides by the needle length (:136) and panics: TA_ERR_DIV_ZERO. */
int ta_hamming_search_naive_with_opts(const uint8_t *needle, size_t needle_len,
                                      const uint8_t *haystack, size_t haystack_len,
                                      uint32_t k, int search_type, ta_match **out, size_t *n_out) {
    if (!out || !n_out || (!needle && needle_len) || (!haystack && haystack_len)) return TA_ERR_ARG;
    *out = nullptr; *n_out = 0;
    if (needle_len > haystack_len) return TA_OK;                                // :100-102
    std::vector<ta_match> hits;
    if (needle_len == 0) {                                                      // the loop body never runs: final_res = 0 everywhere
        // Best collects into Vec::with_capacity(haystack_len / needle_len) (:136): the reference panics on the division
        if (search_type == TA_SEARCH_BEST) return TA_ERR_DIV_ZERO;
        hits.reserve(haystack_len + 1);                                         // (no kernel: this answer needs no device)
        for (size_t i = 0; i <= haystack_len; i++) hits.push_back(ta_match{i, i, 0u, 0u});
        return give(hits, out, n_out);
    }
    int rc;
    if (multi_search_shards(haystack_len) > 1) {
        if (!device_ready()) return TA_ERR_HIP;
        rc = multi_hamming_search_host(needle, needle_len, haystack, haystack_len, k, false, hits);
    } else
    rc = run_search_host(haystack, haystack_len, hits, [&](const uint8_t *hd, ta_match *od, size_t cap, uint64_t *cnt, hipStream_t st) {
        return hamming_search_dev_impl(needle, needle_len, hd, haystack_len, k, 0, od, cap, cnt, st, false);
    });
    if (rc) return rc;
    if (search_type == TA_SEARCH_BEST) hits.resize(ta_search_fold_best(hits.data(), hits.size(), k, 0));
    return give(hits, out, n_out);
}

int ta_hamming_search(const uint8_t *needle, size_t needle_len, const uint8_t *haystack, size_t haystack_len,
                      ta_match **out, size_t *n_out) {                         // src/hamming.rs:422-424
    uint32_t k = ((uint32_t)needle_len >> 1) + ((uint32_t)needle_len & 1u);
    return ta_hamming_search_simd_with_opts(needle, needle_len, haystack, haystack_len, k, TA_SEARCH_BEST, out, n_out);
}

}  // extern "C"
