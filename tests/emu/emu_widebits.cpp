// emu_widebits.cpp -- host emulation driver of the row-blocked bit-parallel kernel body.  TESTS ONLY (see emu_wave.h).
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "emu_wave.h"
#include "lev_band_body.h"
#include "lev_plan.h"

using namespace ta;

// ---- row-blocked bit-parallel full-column kernel (lev_widebits_body.h)
#include "lev_widebits_body.h"

extern "C" int emu_lev_widebits(const uint8_t *a_blob, const uint64_t *a_off, const uint8_t *b_blob, const uint64_t *b_off,
                                uint32_t n, uint32_t k, int has_t, uint64_t max_len, int nwl, uint32_t nwaves, uint32_t *out) {
    LevParams P;
    P.a = StrView{a_blob, a_off, 0, 0};
    P.b = StrView{b_blob, b_off, 0, 0};
    P.subset = nullptr; P.trace = nullptr; P.out = out; P.n = n; P.k = k;
    P.mc = 1; P.gc = 1; P.sg = 0; P.tc = has_t ? 1 : 0;
    P.u = lev_batch_unit_k(k, 1, 1, 0, max_len);
    P.o = 0; P.L = 64; P.PW = 1; P.lds_per_wave = 0; P.Tw = 0; P.ch = 0;
    std::vector<uint32_t> lines((size_t)nwaves * 6 * (max_len + 66));
    P.bnd = lines.data(); P.bnd_line = max_len + 66;
    uint8_t *lds = (uint8_t *)calloc(33 * 64 * 2 * 4 + 64, 1);
    for (uint32_t w = 0; w < nwaves; w++) {
        if (nwl == 1) { if (has_t) LevWideBits<EmuWave, 1, true>::run(P, w, nwaves, lds); else LevWideBits<EmuWave, 1, false>::run(P, w, nwaves, lds); }
        else { if (has_t) LevWideBits<EmuWave, 2, true>::run(P, w, nwaves, lds); else LevWideBits<EmuWave, 2, false>::run(P, w, nwaves, lds); }
    }
    free(lds);
    return 0;
}

// one huge pair as tiles: the launches of the host driver are simulated diagonal by diagonal; within a diagonal the
// tiles run in the given order (0 = ascending stripes, 1 = descending) -- both must give the same answer
template <int NWL, bool TR>
static uint32_t huge_run(const uint8_t *a, uint32_t alen, const uint8_t *b, uint32_t blen, uint32_t k, uint32_t CB, int order) {
    using K = LevWideBits<EmuWave, NWL, TR>;
    const bool swap = alen > blen;
    typename K::Huge H;
    H.ap = swap ? b : a; H.bp = swap ? a : b;
    H.n = swap ? blen : alen; H.m = swap ? alen : blen;
    H.u = lev_batch_unit_k(k, 1, 1, 0, H.m); H.k = k; H.CB = CB;
    if (H.m - H.n > H.u) return 0xFFFFFFFFu;
    if (H.n == 0) return H.m <= k ? H.m : 0xFFFFFFFFu;
    const uint32_t stripes = K::huge_stripes(H.n);
    H.line = (uint64_t)H.m + 66;
    std::vector<uint32_t> lines((size_t)stripes * 3 * H.line, 0xDEADBEEFu), state((size_t)stripes * 64 * 16, 0xDEADBEEFu);
    uint32_t out = 0xDEADBEEFu;
    H.lines = lines.data(); H.state = state.data(); H.out = &out;
    uint8_t *lds = (uint8_t *)calloc(33 * 64 * 2 * 4 + 64, 1);
    const uint32_t D = K::huge_diagonals(H.n, H.m, H.u, CB);
    for (uint32_t d = 0; d < D; d++)
        for (uint32_t i = 0; i < stripes; i++) K::run_tile(H, order ? stripes - 1 - i : i, d, lds);
    free(lds);
    return out;
}

extern "C" uint32_t emu_lev_widebits_huge(const uint8_t *a, uint32_t alen, const uint8_t *b, uint32_t blen, uint32_t k, int has_t,
                                          int nwl, uint32_t CB, int order) {
    if (nwl == 1) return has_t ? huge_run<1, true>(a, alen, b, blen, k, CB, order) : huge_run<1, false>(a, alen, b, blen, k, CB, order);
    return has_t ? huge_run<2, true>(a, alen, b, blen, k, CB, order) : huge_run<2, false>(a, alen, b, blen, k, CB, order);
}

// ---- TRACE: one pair through the row-blocked kernel with records, then the host walk (lev_trace_walk.h)
#include "lev_trace_walk.h"

struct EmuEdit { uint32_t code; uint32_t pad; uint64_t count; };   // code: 0 match, 1 mismatch, 2 a_gap, 3 b_gap, 4 transpose

template <int NWL, bool TR>
static int trace_run(const uint8_t *a, uint32_t alen, const uint8_t *b, uint32_t blen, uint32_t k, uint32_t *dist,
                     std::vector<EmuEdit> &edits) {
    using K = LevWideBits<EmuWave, NWL, TR, true>;
    const bool swap = alen > blen;
    const uint8_t *x = swap ? b : a, *y = swap ? a : b;
    const uint32_t n = swap ? blen : alen, m = swap ? alen : blen;
    uint64_t xoff[2] = {0, n}, yoff[2] = {0, m};
    LevParams P;
    P.a = StrView{x, xoff, 0, 0}; P.b = StrView{y, yoff, 0, 0};
    uint32_t out = 0;
    P.subset = nullptr; P.out = &out; P.n = 1; P.k = k;
    P.mc = 1; P.gc = 1; P.sg = 0; P.tc = TR ? 1 : 0;
    P.u = lev_batch_unit_k(k, 1, 1, 0, m);
    P.o = 0; P.L = 64; P.PW = 1; P.lds_per_wave = 0; P.Tw = 0; P.ch = 0;
    const uint32_t rows = 64u * 32u * NWL, stripes = n ? (n + rows - 1) / rows : 1;
    P.trace_cols = (uint64_t)m + 64;
    std::vector<uint32_t> rec((size_t)stripes * P.trace_cols * 64 * 3 * NWL, 0xDEADBEEFu), lines((size_t)6 * (m + 66));
    P.trace = rec.data(); P.bnd = lines.data(); P.bnd_line = (uint64_t)m + 66;
    uint8_t *lds = (uint8_t *)calloc(33 * 64 * 2 * 4 + 64, 1);
    K::run(P, 0, 1, lds);
    free(lds);
    *dist = out;
    if (out == 0xFFFFFFFFu) return 0;
    WbTrace T{rec.data(), P.trace_cols, (uint32_t)NWL, n, m, P.u};
    std::vector<EmuEdit> rev;
    uint64_t ci = n, cj = m;
    const bool ok = wb_trace_walk(T, x, y, out, TR, [&](int code) {
        uint32_t e;
        switch (code) {
            case 0: ci--; cj--; e = (x[ci] == y[cj]) ? 0u : 1u; break;
            case 1: cj--; e = swap ? 3u : 2u; break;
            case 2: ci--; e = swap ? 2u : 3u; break;
            default: ci -= 2; cj -= 2; e = 4u; break;
        }
        if (!rev.empty() && rev.back().code == e) rev.back().count++;
        else rev.push_back(EmuEdit{e, 0u, 1u});
    });
    if (!ok) return 2;
    edits.assign(rev.rbegin(), rev.rend());
    return 0;
}

extern "C" int emu_lev_widebits_trace(const uint8_t *a, uint32_t alen, const uint8_t *b, uint32_t blen, uint32_t k, int has_t, int nwl,
                                      uint32_t *dist, EmuEdit *out, uint64_t cap, uint64_t *n_out) {
    std::vector<EmuEdit> e;
    int rc;
    if (nwl == 1) rc = has_t ? trace_run<1, true>(a, alen, b, blen, k, dist, e) : trace_run<1, false>(a, alen, b, blen, k, dist, e);
    else rc = has_t ? trace_run<2, true>(a, alen, b, blen, k, dist, e) : trace_run<2, false>(a, alen, b, blen, k, dist, e);
    *n_out = e.size();
    for (uint64_t i = 0; i < e.size() && i < cap; i++) out[i] = e[i];
    return rc;
}
