// emu_bits.cpp -- host emulation driver of the bit-parallel band kernel body.  TESTS ONLY (see emu_wave.h).
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "emu_wave.h"
#include "lev_band_body.h"
#include "lev_plan.h"

using namespace ta;

extern int g_emu_force_ch;
static int g_emu_bits_fixed_chunk = 0;      // 1: fixed-length batches take the chunk form (as the launcher does up to one line per string)
extern "C" void emu_bits_set_fixed_chunk(int on) { g_emu_bits_fixed_chunk = on; }
static int g_emu_bits_vline = 0;            // 1: the VLINE form of the fetch (what the launcher gives CSR batches), whatever the batch's form
extern "C" void emu_bits_set_vline(int on) { g_emu_bits_vline = on; }
static uint32_t g_emu_bits_tune = 0;        // LevParams::tune bits (2: early out)
extern "C" void emu_bits_set_tune(uint32_t bits) { g_emu_bits_tune = bits; }

// ---- bit-parallel band kernel (lev_bits_body.h)
#include "lev_bits_body.h"

template <int NA> static void run_bits(const LevParams &P, bool trans, bool stat, bool line, uint32_t waves) {
    uint8_t *lds = (uint8_t *)malloc(P.lds_per_wave + 64);
    for (uint32_t w = 0; w < waves; w++) {
        memset(lds, g_emu_bits_vline ? 0xA5 : 0, P.lds_per_wave + 64);
#define GO(T, S) do { if (g_emu_bits_vline) LevBits<EmuWave, NA, T, S, false, false, false, true>::run(P, w, lds); \
                      else if (line) LevBits<EmuWave, NA, T, S, true>::run(P, w, lds); else LevBits<EmuWave, NA, T, S, false>::run(P, w, lds); } while (0)
        if constexpr (NA >= 8) {
            if (stat) { if (trans) GO(true, true); else GO(false, true); continue; }
        }
        if (trans) GO(true, false); else GO(false, false);
#undef GO
    }
    free(lds);
}

// a_off / b_off == nullptr: fixed-length (strided) batch of a_len / b_len bytes per string -- the coalesced fetch form;
// subset (may be nullptr): pair indices, as the levenshtein_exp rounds pass them
extern "C" int emu_lev_bits_any(const uint8_t *a_blob, const uint64_t *a_off, uint64_t a_len, const uint8_t *b_blob,
                                const uint64_t *b_off, uint64_t b_len, const uint32_t *subset,
                                uint32_t n, uint32_t k, int has_t, uint64_t max_len, int force_NA, int force_static, uint32_t *out,
                                uint32_t *plan_out /* NA, u, Tw, static */);

extern "C" int emu_lev_bits(const uint8_t *a_blob, const uint64_t *a_off, const uint8_t *b_blob, const uint64_t *b_off,
                            uint32_t n, uint32_t k, int has_t, uint64_t max_len, int force_NA, int force_static, uint32_t *out,
                            uint32_t *plan_out /* NA, u, Tw, static */) {
    return emu_lev_bits_any(a_blob, a_off, 0, b_blob, b_off, 0, nullptr, n, k, has_t, max_len, force_NA, force_static, out, plan_out);
}

extern "C" int emu_lev_bits_any(const uint8_t *a_blob, const uint64_t *a_off, uint64_t a_len, const uint8_t *b_blob,
                                const uint64_t *b_off, uint64_t b_len, const uint32_t *subset,
                                uint32_t n, uint32_t k, int has_t, uint64_t max_len, int force_NA, int force_static, uint32_t *out,
                                uint32_t *plan_out /* NA, u, Tw, static */) {
    LevBitsPlan pl = lev_bits_make_plan(k, 1, 1, 0, has_t != 0, 1, max_len, force_NA, g_emu_force_ch, force_static);
    if (!pl.ok) return 1;
    LevParams P;
    P.a = StrView{a_blob, a_off, a_off ? 0 : a_len, a_off ? 0 : a_len};
    P.b = StrView{b_blob, b_off, b_off ? 0 : b_len, b_off ? 0 : b_len};
    P.subset = subset; P.trace = nullptr; P.out = out; P.n = n; P.k = k;
    P.mc = 1; P.gc = 1; P.sg = 0; P.tc = has_t ? 1 : 0;
    P.u = pl.u; P.o = 0; P.L = 1; P.PW = 64; P.lds_per_wave = pl.lds_per_wave; P.Tw = pl.Tw; P.ch = pl.ch;
    if (g_emu_bits_vline && P.lds_per_wave < LEV_BITS_VLINE_LDS) P.lds_per_wave = LEV_BITS_VLINE_LDS;
    P.tune = g_emu_bits_tune;
    if (plan_out) { plan_out[0] = pl.NA; plan_out[1] = pl.u; plan_out[2] = pl.Tw; plan_out[3] = pl.s8 ? 3 : pl.stat; }
    const uint32_t waves = (n + 63) / 64;
    struct RangeGuard { ~RangeGuard() { EmuWave::clear_ranges(); } } range_guard;     // (the other drivers read unchecked)
    EmuWave::clear_ranges();                               // VLINE reads whole lines: bytes outside the blobs (+ 16 of slack) read as 0xA5
    if (g_emu_bits_vline && !subset) {                      // (checked reads only where whole lines are read; n = the batch then)
        EmuWave::add_range(a_blob, (a_off ? a_off[n] : (uint64_t)n * a_len) + 16);
        EmuWave::add_range(b_blob, (b_off ? b_off[n] : (uint64_t)n * b_len) + 16);
    }
    // fixed-length batches take the line form here whatever their length (the launcher keeps the chunk form up to one line per
    // string -- a speed choice; the emulation covers the line form on short strings too)
    if (pl.s8) {
        uint8_t *lds = (uint8_t *)calloc(P.lds_per_wave + 64, 1);
        const bool line = !a_off && !b_off && !g_emu_bits_fixed_chunk;
        for (uint32_t w = 0; w < waves; w++) {
            if (g_emu_bits_vline) {
                memset(lds, 0xA5, P.lds_per_wave + 64);
                if (has_t) LevBits<EmuWave, 8, true, false, false, true, false, true>::run(P, w, lds); else LevBits<EmuWave, 8, false, false, false, true, false, true>::run(P, w, lds);
                continue;
            }
            if (line && (P.tune & 2u)) {           // as the launcher: the early-out instantiation under the option
                if (has_t) LevBits<EmuWave, 8, true, false, true, true, true>::run(P, w, lds); else LevBits<EmuWave, 8, false, false, true, true, true>::run(P, w, lds);
            } else if (has_t) { if (line) LevBits<EmuWave, 8, true, false, true, true>::run(P, w, lds); else LevBits<EmuWave, 8, true, false, false, true>::run(P, w, lds); }
            else { if (line) LevBits<EmuWave, 8, false, false, true, true>::run(P, w, lds); else LevBits<EmuWave, 8, false, false, false, true>::run(P, w, lds); }
        }
        free(lds);
        return 0;
    }
    switch (pl.NA) {
#define CASE(d) case d: run_bits<d>(P, has_t != 0, pl.stat, !a_off && !b_off && !g_emu_bits_fixed_chunk, waves); break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12)
        CASE(13) CASE(14) CASE(15) CASE(16) CASE(18) CASE(20) CASE(22) CASE(24) CASE(26) CASE(28) CASE(30) CASE(32)
#undef CASE
        default: return 2;
    }
    return 0;
}


// ---- two pairs per lane (lev_bits2_body.h): fixed-length batches, narrow bands
#include "lev_bits2_body.h"

extern "C" int emu_lev_bits2(const uint8_t *a_blob, uint64_t a_len, const uint8_t *b_blob, uint64_t b_len, const uint32_t *subset,
                             uint32_t n, uint32_t k, int has_t, uint32_t *out, uint32_t *plan_out /* NA, u, Tw */) {
    const uint64_t max_len = a_len > b_len ? a_len : b_len;
    LevBits2Plan pl = lev_bits2_make_plan(k, 1, 1, 0, has_t != 0, 1, max_len, true, LEV_BITS2_MIN_PAIRS);
    if (!pl.ok) return 1;
    LevParams P;
    P.a = StrView{a_blob, nullptr, a_len, a_len};
    P.b = StrView{b_blob, nullptr, b_len, b_len};
    P.subset = subset; P.trace = nullptr; P.out = out; P.n = n; P.k = k;
    P.mc = 1; P.gc = 1; P.sg = 0; P.tc = has_t ? 1 : 0;
    P.u = pl.u; P.o = 0; P.L = 1; P.PW = 128; P.lds_per_wave = pl.lds_per_wave; P.Tw = pl.Tw; P.ch = 64;
    P.tune = g_emu_bits_tune;
    if (plan_out) { plan_out[0] = pl.NA; plan_out[1] = pl.u; plan_out[2] = pl.Tw; }
    uint8_t *lds = (uint8_t *)calloc(P.lds_per_wave + 64, 1);
    const uint32_t waves = (n + 127) / 128;
    for (uint32_t w = 0; w < waves; w++) {
        if (P.tune & 2u) { if (has_t) LevBits2<EmuWave, true, true>::run(P, w, lds); else LevBits2<EmuWave, false, true>::run(P, w, lds); }
        else if (has_t) LevBits2<EmuWave, true>::run(P, w, lds);
        else LevBits2<EmuWave, false>::run(P, w, lds);
    }
    free(lds);
    return 0;
}

// ---- small alphabets (lev_bitsq_body.h): fixed-length batches, the match vector from per-symbol tables
#include "lev_bitsq_body.h"

// out entries of pairs that hold a byte outside the alphabet stay untouched; bad_out[0] = their number, bad_out[1..] = the pairs
extern "C" int emu_lev_bitsq(const uint8_t *a_blob, uint64_t a_len, const uint8_t *b_blob, uint64_t b_len, const uint32_t *subset,
                             uint32_t n, uint32_t k, int has_t, const uint8_t *sym, uint32_t n_sym, uint32_t *out, uint32_t *bad_out) {
    const uint64_t max_len = a_len > b_len ? a_len : b_len;
    uint32_t u = 0;
    if (!lev_bitsq_applies(k, 1, 1, 0, has_t != 0, 1, max_len, true, LEV_BITSQ_MIN_PAIRS, &u)) return 1;
    LevParams P;
    P.a = StrView{a_blob, nullptr, a_len, a_len};
    P.b = StrView{b_blob, nullptr, b_len, b_len};
    P.subset = subset; P.trace = nullptr; P.out = out; P.n = n; P.k = k;
    P.mc = 1; P.gc = 1; P.sg = 0; P.tc = has_t ? 1 : 0;
    P.u = u; P.o = 0; P.L = 1; P.PW = 64; P.lds_per_wave = LevBitsQ<EmuWave, false>::LDS_PER_WAVE; P.Tw = 0; P.ch = 0;
    if (!lev_bitsq_hash(sym, n_sym, &P.q_shift, &P.q_table)) return 3;
    bad_out[0] = 0;
    P.q_bad_count = bad_out; P.q_bad_list = bad_out + 1;
    uint8_t *lds = (uint8_t *)malloc(P.lds_per_wave + 64);
    const uint32_t waves = (n + 63) / 64;
    for (uint32_t w = 0; w < waves; w++) {
        memset(lds, 0xA5, P.lds_per_wave + 64);            // LDS starts out as garbage on the device
        if (has_t) LevBitsQ<EmuWave, true>::run(P, w, lds);
        else LevBitsQ<EmuWave, false>::run(P, w, lds);
    }
    free(lds);
    return 0;
}

// ---- alphabets of up to 32 symbols (lev_bitsqw_body.h)
#include "lev_bitsqw_body.h"

extern "C" int emu_lev_bitsqw(const uint8_t *a_blob, uint64_t a_len, const uint8_t *b_blob, uint64_t b_len, const uint32_t *subset,
                              uint32_t n, uint32_t k, int has_t, const uint8_t *sym, uint32_t n_sym, uint32_t *out, uint32_t *bad_out) {
    const uint64_t max_len = a_len > b_len ? a_len : b_len;
    uint32_t u = 0;
    if (!lev_bitsq_applies(k, 1, 1, 0, has_t != 0, 1, max_len, true, LEV_BITSQ_MIN_PAIRS, &u)) return 1;
    LevParams P;
    P.a = StrView{a_blob, nullptr, a_len, a_len};
    P.b = StrView{b_blob, nullptr, b_len, b_len};
    P.subset = subset; P.trace = nullptr; P.out = out; P.n = n; P.k = k;
    P.mc = 1; P.gc = 1; P.sg = 0; P.tc = has_t ? 1 : 0;
    P.u = u; P.o = 0; P.L = 1; P.PW = 64; P.Tw = 0; P.ch = 0;
    if (!lev_bitsqw_hash(sym, n_sym, &P.q_shift, &P.q_memb, &P.q_hi)) return 3;
    P.q_ns = n_sym;
    P.lds_per_wave = LevBitsQW<EmuWave, false>::lds_per_wave(P.q_ns);
    bad_out[0] = 0;
    P.q_bad_count = bad_out; P.q_bad_list = bad_out + 1;
    uint8_t tab[256];
    for (uint32_t i = 0; i < 256; i++) tab[i] = (uint8_t)lev_bitsqw_entry(i, P.q_shift, P.q_memb, P.q_hi);
    const size_t slack = 1100;                             // a byte outside the alphabet looks up "ring" 0xFF: read, never used
    uint8_t *lds = (uint8_t *)malloc(P.lds_per_wave + slack);
    const uint32_t waves = (n + 63) / 64;
    for (uint32_t w = 0; w < waves; w++) {
        memset(lds, 0xA5, P.lds_per_wave + slack);         // LDS starts out as garbage on the device
        if (has_t) LevBitsQW<EmuWave, true>::run(P, w, lds, tab);
        else LevBitsQW<EmuWave, false>::run(P, w, lds, tab);
    }
    free(lds);
    return 0;
}

// ---- one pair, one wavefront (lev_one_body.h)
#include "lev_one_body.h"

extern "C" int emu_lev_one(const uint8_t *a, uint64_t a_len, const uint8_t *b, uint64_t b_len, uint32_t k, int has_t, uint32_t *out) {
    const uint64_t max_len = a_len > b_len ? a_len : b_len;
    uint32_t u = 0;
    if (!lev_one_applies(k, 1, 1, 0, has_t != 0, 1, max_len, &u)) return 1;
    LevParams P;
    P.a = StrView{a, nullptr, a_len, a_len};
    P.b = StrView{b, nullptr, b_len, b_len};
    P.subset = nullptr; P.trace = nullptr; P.out = out; P.n = 1; P.k = k;
    P.mc = 1; P.gc = 1; P.sg = 0; P.tc = has_t ? 1 : 0;
    P.u = u; P.o = 0; P.L = 64; P.PW = 1; P.lds_per_wave = 0; P.Tw = 0; P.ch = 0;
    const size_t lds_bytes = (LEV_ONE_PAD_LO + max_len + LEV_ONE_PAD_HI + 15u) & ~(size_t)15;
    uint8_t *lds = (uint8_t *)malloc(lds_bytes);
    memset(lds, 0xA5, lds_bytes);                      // nothing may depend on what the pads hold
    const bool wide = u + 1u + (has_t ? 2u : 0u) > 32u;
    if (has_t) { if (wide) LevOne<EmuWave, true, true>::run(P, lds); else LevOne<EmuWave, true, false>::run(P, lds); }
    else { if (wide) LevOne<EmuWave, false, true>::run(P, lds); else LevOne<EmuWave, false, false>::run(P, lds); }
    free(lds);
    return 0;
}
