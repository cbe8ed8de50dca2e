// mirror_check.cpp -- exercises include/triple_accel.hpp (the C++ mirror of the reference's public API) against known
// answers taken from the reference's own doc-tests and tests/basic_tests.rs.  Built and run by tests/test_cpp_mirror.py.
//   mirror_check nogpu : without a device every compute call must throw device_error (no CPU fallback); host-only parts work
//   mirror_check gpu   : the known answers
#include <cstdio>
#include <cstring>
#include <string>

#include "triple_accel.hpp"

namespace ta = triple_accel;
static ta::bytes B(const char *s) { return ta::bytes(reinterpret_cast<const std::uint8_t *>(s), std::strlen(s)); }
static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL line %d: %s\n", __LINE__, #c); fails++; } } while (0)

int main(int argc, char **argv) {
    const std::string mode = argc > 1 ? argv[1] : "gpu";
    // host-only: cost validation behaves like EditCosts::new (src/levenshtein.rs:38-60)
    bool threw = false;
    try { ta::EditCosts bad(0, 1, 0, std::nullopt); } catch (const ta::panic_error &) { threw = true; }
    CHECK(threw);
    ta::EditCosts ok(1, 1, 0, std::uint8_t{1});
    (void)ok;
    if (mode == "nogpu") {
        threw = false;
        try { ta::hamming(B("abc"), B("abd")); } catch (const ta::device_error &) { threw = true; }
        CHECK(threw);
        threw = false;
        try { ta::levenshtein(B("abc"), B("abd")); } catch (const ta::device_error &) { threw = true; }
        CHECK(threw);
        threw = false;
        try { ta::levenshtein_search(B("abc"), B("  abd")); } catch (const ta::device_error &) { threw = true; }
        CHECK(threw);
        std::printf(fails ? "mirror_check nogpu: %d failures\n" : "mirror_check nogpu: ok\n", fails);
        return fails ? 1 : 0;
    }
    CHECK(ta::hamming(B("abc"), B("abd")) == 1);
    threw = false;
    try { ta::hamming(B("abc"), B("ab")); } catch (const ta::panic_error &) { threw = true; }
    CHECK(threw);
    CHECK(ta::levenshtein(B("abc"), B("abcd")) == 1);
    CHECK(ta::levenshtein(B("kitten"), B("sitting")) == 3);
    CHECK(ta::rdamerau(B("abcd"), B("bacd")) == 1);
    CHECK(ta::levenshtein(B("abcd"), B("bacd")) == 2);
    CHECK(ta::levenshtein_exp(B("abc"), B("abcd")) == 1);
    CHECK(ta::rdamerau_exp(B("abc"), B("acb")) == 1);
    CHECK(ta::levenshtein_simd_k(B("abc"), B("ab"), 1).value() == 1);
    {   // the device set: device 0 listed three times -- the host-pointer batch entry shards the pairs over three workers
        ta::set_devices({0, 0, 0});
        CHECK(ta::get_devices().size() == 3);
        const auto r = ta::levenshtein_simd_k_with_opts_slices({{B("kitten"), B("sitting")}, {B("abc"), B("abd")}, {B(""), B("xy")}, {B("abcdefgh"), B("zzzzzzzz")}}, 3, ta::LEVENSHTEIN_COSTS);
        CHECK(r.size() == 4 && r[0].value() == 3 && r[1].value() == 1 && r[2].value() == 2 && !r[3].has_value());
        ta::set_devices({});
        CHECK(ta::get_devices().size() >= 1);
    }
    CHECK(!ta::levenshtein_simd_k(B("abcdef"), B("uvwxyz"), 3).has_value());
    auto r = ta::levenshtein_simd_k_with_opts(B("abc"), B("ab"), 1, true, ta::LEVENSHTEIN_COSTS);
    CHECK(r.has_value() && r->first == 1 && r->second.has_value());
    if (r && r->second) {
        const std::vector<ta::Edit> want{{ta::EditType::Match, 2}, {ta::EditType::BGap, 1}};
        CHECK(*r->second == want);
    }
    auto e = ta::levenshtein_exp_with_opts(B("abc"), B("abXc"), true, ta::EditCosts(1, 1, 0, std::nullopt));
    CHECK(e.first == 1 && e.second.has_value());
    auto hits = ta::levenshtein_search(B("abc"), B("  abd"));
    CHECK(hits.size() == 1 && hits[0] == (ta::Match{2, 5, 1}));       // src/lib.rs doc-test of levenshtein_search
    auto all = ta::levenshtein_search_simd_with_opts(B("abc"), B("  abd"), 1, ta::SearchType::All, ta::LEVENSHTEIN_COSTS, false);
    CHECK(all.size() == 2);
    auto hh = ta::hamming_search(B("abc"), B("  abd"));
    CHECK(hh.size() == 1 && hh[0] == (ta::Match{2, 5, 1}));
    threw = false;
    try { ta::hamming_search(B("a"), ta::bytes(reinterpret_cast<const std::uint8_t *>("a\0b"), 3)); } catch (const ta::panic_error &) { threw = true; }
    CHECK(threw);
    // round 3: the first match of the lazy All-mode iterator (tests/basic_tests.rs:628-632) and the queue of single pairs
    auto first = ta::levenshtein_search_first(B("tst"), B("testing 123 tasting!"), 1, ta::LEVENSHTEIN_COSTS, false);
    CHECK(first.has_value() && *first == (ta::Match{0, 4, 1}));
    CHECK(!ta::levenshtein_search_first(B("abc"), B("xyzxyz"), 0, ta::LEVENSHTEIN_COSTS, false).has_value());
    {
        ta::Queue q(2, ta::LEVENSHTEIN_COSTS);
        CHECK(q.push(B("kitten"), B("sitting")) == 0);
        CHECK(q.push(B("abc"), B("abd")) == 1);
        CHECK(q.push(B(""), B("")) == 2);
        auto res = q.flush();
        CHECK(res.size() == 3 && !res[0].has_value() && res[1].value_or(99) == 1 && res[2].value_or(99) == 0);
        CHECK(q.flush().empty());
    }
    threw = false;
    try { ta::hamming_search_naive_with_opts(B(""), B("abc"), 1, ta::SearchType::Best); } catch (const ta::panic_error &) { threw = true; }
    CHECK(threw);                                                       // src/hamming.rs:136: haystack_len / 0
    std::printf(fails ? "mirror_check gpu: %d failures\n" : "mirror_check gpu: ok\n", fails);
    return fails ? 1 : 0;
}
