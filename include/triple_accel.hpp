// triple_accel.hpp -- header-only C++ mirror of triple_accel's public Rust API (src/lib.rs:121-127 and the
// `levenshtein` / `hamming` modules) over the C ABI of triple_accel_amd.h.  Same names, argument order and
// meaning; Option<T> -> std::optional<T>, panic!/assert! -> triple_accel::panic_error, the boxed Match
// iterator -> std::vector<Match>.  (The Rust shim itself is in INTEGRATION.md; Rust is not in this image.)
#pragma once
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "triple_accel_amd.h"

namespace triple_accel {

struct panic_error : std::runtime_error { using std::runtime_error::runtime_error; };   // where Rust panics
struct device_error : std::runtime_error { using std::runtime_error::runtime_error; };  // HIP failure / no GPU (no CPU fallback)
struct unsupported_error : std::runtime_error { using std::runtime_error::runtime_error; };

struct Match { std::size_t start, end; std::uint32_t k; bool operator==(const Match &o) const { return start == o.start && end == o.end && k == o.k; } };   // src/lib.rs:135-142
enum class SearchType { All = 0, Best = 1 };                                            // src/lib.rs:171-174
using bytes = std::basic_string_view<std::uint8_t>;

inline void check_(int rc) {
    switch (rc) {
        case TA_OK: return;
        case TA_ERR_LEN_MISMATCH: throw panic_error("assertion failed: a.len() == b.len()");
        case TA_ERR_NULL_BYTE: throw panic_error("No zero/null bytes allowed in the string!");
        case TA_ERR_BAD_COSTS: throw panic_error("invalid EditCosts");
        case TA_ERR_DIV_ZERO: throw panic_error("attempt to divide by zero");
        case TA_ERR_UNSUPPORTED: throw unsupported_error(ta_status_str(rc));
        default: throw device_error(std::string(ta_status_str(rc)) + ": " + ta_last_error());
    }
}

// src/levenshtein.rs:20-71
class EditCosts {
public:
    EditCosts(std::uint8_t mismatch_cost, std::uint8_t gap_cost, std::uint8_t start_gap_cost, std::optional<std::uint8_t> transpose_cost) {
        check_(ta_edit_costs_new(mismatch_cost, gap_cost, start_gap_cost, transpose_cost.has_value(), transpose_cost.value_or(0), &c_));
    }
    const ta_edit_costs *raw() const { return &c_; }
private:
    ta_edit_costs c_;
};
inline const EditCosts LEVENSHTEIN_COSTS{1, 1, 0, std::nullopt};   // :76-81
inline const EditCosts RDAMERAU_COSTS{1, 1, 0, std::uint8_t{1}};   // :84-89

inline std::optional<std::uint32_t> opt_(std::uint32_t v) { return v == TA_NONE ? std::nullopt : std::optional<std::uint32_t>(v); }

inline std::uint32_t hamming(bytes a, bytes b) {                                          // src/hamming.rs:390
    std::uint32_t o; check_(ta_hamming(a.data(), a.size(), b.data(), b.size(), &o)); return o;
}
inline std::optional<std::uint32_t> levenshtein_simd_k(bytes a, bytes b, std::uint32_t k) {   // src/levenshtein.rs:677
    std::uint32_t o; check_(ta_levenshtein_simd_k(a.data(), a.size(), b.data(), b.size(), k, &o)); return opt_(o);
}
// src/lib.rs:145-168
enum class EditType { Match = 0, Mismatch = 1, AGap = 2, BGap = 3, Transpose = 4 };
struct Edit { EditType edit; std::size_t count; bool operator==(const Edit &o) const { return edit == o.edit && count == o.count; } };
using Traceback = std::optional<std::vector<Edit>>;

inline std::vector<Edit> take_edits_(ta_edit *e, std::size_t n) {
    std::vector<Edit> v; v.reserve(n);
    for (std::size_t i = 0; i < n; i++) v.push_back(Edit{(EditType)e[i].edit, (std::size_t)e[i].count});
    ta_free(e);
    return v;
}
// :714 -- Some((distance, traceback)) / None; trace_on = true returns the run-length edit script (more than 8 GB of
// traceback records: unsupported_error -- use the scalar routine there)
inline std::optional<std::pair<std::uint32_t, Traceback>> levenshtein_simd_k_with_opts(bytes a, bytes b, std::uint32_t k, bool trace_on, const EditCosts &costs) {
    std::uint32_t o;
    if (trace_on) {
        ta_edit *e = nullptr; std::size_t n = 0;
        check_(ta_levenshtein_trace(a.data(), a.size(), b.data(), b.size(), k, costs.raw(), &o, &e, &n));
        if (o == TA_NONE) return std::nullopt;
        return std::make_pair(o, Traceback(take_edits_(e, n)));
    }
    check_(ta_levenshtein_simd_k_with_opts(a.data(), a.size(), b.data(), b.size(), k, 0, costs.raw(), &o));
    if (o == TA_NONE) return std::nullopt;
    return std::make_pair(o, Traceback(std::nullopt));
}
// :1480
inline std::pair<std::uint32_t, Traceback> levenshtein_exp_with_opts(bytes a, bytes b, bool trace_on, const EditCosts &costs) {
    std::uint32_t o;
    if (trace_on) {
        ta_edit *e = nullptr; std::size_t n = 0;
        check_(ta_levenshtein_exp_trace(a.data(), a.size(), b.data(), b.size(), costs.raw(), &o, &e, &n));
        return std::make_pair(o, Traceback(take_edits_(e, n)));
    }
    check_(ta_levenshtein_exp_with_opts(a.data(), a.size(), b.data(), b.size(), 0, costs.raw(), &o));
    return std::make_pair(o, Traceback(std::nullopt));
}
inline std::uint32_t levenshtein(bytes a, bytes b) { std::uint32_t o; check_(ta_levenshtein(a.data(), a.size(), b.data(), b.size(), &o)); return o; }          // :1397
inline std::uint32_t rdamerau(bytes a, bytes b) { std::uint32_t o; check_(ta_rdamerau(a.data(), a.size(), b.data(), b.size(), &o)); return o; }                // :1419
inline std::uint32_t levenshtein_exp(bytes a, bytes b) { std::uint32_t o; check_(ta_levenshtein_exp(a.data(), a.size(), b.data(), b.size(), &o)); return o; }  // :1445
inline std::uint32_t rdamerau_exp(bytes a, bytes b) { std::uint32_t o; check_(ta_rdamerau_exp(a.data(), a.size(), b.data(), b.size(), &o)); return o; }        // :1516

inline std::vector<Match> take_(ta_match *m, std::size_t n) {
    std::vector<Match> v; v.reserve(n);
    for (std::size_t i = 0; i < n; i++) v.push_back(Match{(std::size_t)m[i].start, (std::size_t)m[i].end, m[i].k});
    ta_free(m);
    return v;
}
inline std::vector<Match> levenshtein_search_simd_with_opts(bytes needle, bytes haystack, std::uint32_t k, SearchType st, const EditCosts &costs, bool anchored) {   // :1911
    ta_match *m; std::size_t n;
    check_(ta_levenshtein_search_simd_with_opts(needle.data(), needle.size(), haystack.data(), haystack.size(), k, (int)st, costs.raw(), anchored, &m, &n));
    return take_(m, n);
}
// pairs produced one at a time, answered together by ONE batch pass per flush (ta_queue_*; no reference analogue)
class Queue {
public:
    Queue(std::uint32_t k, const EditCosts &costs) { check_(ta_queue_create(k, costs.raw(), &q_)); }
    ~Queue() { ta_queue_destroy(q_); }
    Queue(const Queue &) = delete;
    Queue &operator=(const Queue &) = delete;
    std::size_t push(bytes a, bytes b) { std::size_t t = 0; check_(ta_queue_push(q_, a.data(), a.size(), b.data(), b.size(), &t)); return t; }
    std::vector<std::optional<std::uint32_t>> flush() {
        const std::uint32_t *r = nullptr; std::size_t n = 0;
        check_(ta_queue_flush(q_, &r, &n));
        std::vector<std::optional<std::uint32_t>> v; v.reserve(n);
        for (std::size_t i = 0; i < n; i++) v.push_back(opt_(r[i]));
        return v;
    }
private:
    ta_queue *q_ = nullptr;
};
// what a caller's loop over levenshtein_simd_k_with_opts(a, b, k, false, costs) computes, answered by one batch pass per 65536 pairs
template <class PairRange>
inline std::vector<std::optional<std::uint32_t>> levenshtein_simd_k_with_opts_many(const PairRange &pairs, std::uint32_t k, const EditCosts &costs) {
    Queue q(k, costs);
    std::vector<std::optional<std::uint32_t>> out;
    std::size_t pending = 0;
    for (const auto &pr : pairs) {
        q.push(pr.first, pr.second);
        if (++pending >= (std::size_t(1) << 16)) { auto r = q.flush(); out.insert(out.end(), r.begin(), r.end()); pending = 0; }
    }
    if (pending) { auto r = q.flush(); out.insert(out.end(), r.begin(), r.end()); }
    return out;
}
// The device set (include/triple_accel_amd.h): the GPUs the host entry points fan out over (empty: every visible device, the default).  With
// more than one entry the *_many / *_slices functions, Queue::flush and the searches over haystacks of >= 8 MiB are partitioned inside the library.
inline void set_devices(const std::vector<int> &devices) { check_(ta_set_devices(devices.empty() ? nullptr : devices.data(), devices.size())); }
inline std::vector<int> get_devices() {
    std::size_t n = 0;
    check_(ta_get_devices(nullptr, 0, &n));
    std::vector<int> v(n);
    check_(ta_get_devices(v.data(), n, &n));
    return v;
}
// pairs that already sit in memory: gathered into one CSR blob per side and handed to the host-pointer batch entry (no queue copy)
inline std::vector<std::optional<std::uint32_t>> levenshtein_simd_k_with_opts_slices(const std::vector<std::pair<bytes, bytes>> &pairs, std::uint32_t k,
                                                                                     const EditCosts &costs) {
    std::vector<std::uint8_t> blob[2];
    std::vector<std::uint64_t> off[2] = {{0}, {0}};
    for (const auto &pr : pairs) {
        blob[0].insert(blob[0].end(), pr.first.begin(), pr.first.end());
        blob[1].insert(blob[1].end(), pr.second.begin(), pr.second.end());
        off[0].push_back(blob[0].size());
        off[1].push_back(blob[1].size());
    }
    const ta_strings a = {blob[0].data(), off[0].data(), 0, 0, 0}, b = {blob[1].data(), off[1].data(), 0, 0, 0};
    std::vector<std::uint32_t> out(pairs.size());
    check_(ta_levenshtein_k_batch_host(&a, &b, pairs.size(), k, costs.raw(), out.data()));
    std::vector<std::optional<std::uint32_t>> v;
    v.reserve(out.size());
    for (std::uint32_t d : out) v.push_back(opt_(d));
    return v;
}
// `.next()` on the reference's lazy All-mode iterator (src/levenshtein.rs:2282-2420): the first match, found without scanning the rest
inline std::optional<Match> levenshtein_search_first(bytes needle, bytes haystack, std::uint32_t k, const EditCosts &costs, bool anchored) {
    ta_match m; int found = 0;
    check_(ta_levenshtein_search_first(needle.data(), needle.size(), haystack.data(), haystack.size(), k, costs.raw(), anchored, &m, &found));
    if (!found) return std::nullopt;
    return Match{(std::size_t)m.start, (std::size_t)m.end, m.k};
}
inline std::vector<Match> levenshtein_search(bytes needle, bytes haystack) {             // :2508
    ta_match *m; std::size_t n;
    check_(ta_levenshtein_search(needle.data(), needle.size(), haystack.data(), haystack.size(), &m, &n));
    return take_(m, n);
}
inline std::vector<Match> hamming_search_simd_with_opts(bytes needle, bytes haystack, std::uint32_t k, SearchType st) {   // src/hamming.rs:454
    ta_match *m; std::size_t n;
    check_(ta_hamming_search_simd_with_opts(needle.data(), needle.size(), haystack.data(), haystack.size(), k, (int)st, &m, &n));
    return take_(m, n);
}
inline std::vector<Match> hamming_search(bytes needle, bytes haystack) {                 // src/hamming.rs:588
    ta_match *m; std::size_t n;
    check_(ta_hamming_search(needle.data(), needle.size(), haystack.data(), haystack.size(), &m, &n));
    return take_(m, n);
}

// ---- the reference's scalar entry points: the kernels implement the scalar rules, so these are the same calls under the other names
inline std::uint32_t hamming_naive(bytes a, bytes b) { return hamming(a, b); }                                  // src/hamming.rs:36
inline std::uint32_t hamming_words_64(bytes a, bytes b) { return hamming(a, b); }                               // :176
inline std::uint32_t hamming_words_128(bytes a, bytes b) { return hamming(a, b); }                              // :249
inline std::uint32_t hamming_simd_parallel(bytes a, bytes b) { return hamming(a, b); }                          // :317
inline std::uint32_t hamming_simd_movemask(bytes a, bytes b) { return hamming(a, b); }                          // :354
inline std::vector<Match> hamming_search_naive_with_opts(bytes needle, bytes haystack, std::uint32_t k, SearchType st) {   // :96 (no NUL-byte panic)
    ta_match *m; std::size_t n;
    check_(ta_hamming_search_naive_with_opts(needle.data(), needle.size(), haystack.data(), haystack.size(), k, (int)st, &m, &n));
    return take_(m, n);
}
inline std::vector<Match> hamming_search_naive(bytes needle, bytes haystack) {                                   // :70
    return hamming_search_naive_with_opts(needle, haystack, (std::uint32_t)((needle.size() >> 1) + (needle.size() & 1)), SearchType::Best);
}
inline std::optional<std::uint32_t> levenshtein_naive_k(bytes a, bytes b, std::uint32_t k) { return levenshtein_simd_k(a, b, k); }    // src/levenshtein.rs:342
inline std::optional<std::pair<std::uint32_t, Traceback>> levenshtein_naive_k_with_opts(bytes a, bytes b, std::uint32_t k, bool trace_on, const EditCosts &costs) {
    return levenshtein_simd_k_with_opts(a, b, k, trace_on, costs);                                               // :376
}
inline std::pair<std::uint32_t, Traceback> levenshtein_naive_with_opts(bytes a, bytes b, bool trace_on, const EditCosts &costs) {   // :148
    return *levenshtein_simd_k_with_opts(a, b, 0xFFFFFFFFu, trace_on, costs);
}
inline std::uint32_t levenshtein_naive(bytes a, bytes b) { return levenshtein(a, b); }                          // :105
inline std::vector<Match> levenshtein_search_naive_with_opts(bytes needle, bytes haystack, std::uint32_t k, SearchType st, const EditCosts &costs, bool anchored) {
    return levenshtein_search_simd_with_opts(needle, haystack, k, st, costs, anchored);                          // :1589
}
inline std::vector<Match> levenshtein_search_naive(bytes needle, bytes haystack) { return levenshtein_search(needle, haystack); }   // :1549
inline std::vector<Match> levenshtein_search_simd(bytes needle, bytes haystack) { return levenshtein_search(needle, haystack); }    // :1866

}  // namespace triple_accel
