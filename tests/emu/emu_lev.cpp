// emu_lev.cpp -- runs the band-wavefront kernel body on the host, 64 lanes in lock-step.
// TESTS ONLY (see emu_wave.h).  Built by tests/emu/Makefile into tests/emu/libta_emu.so.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "emu_wave.h"
#include "lev_band_body.h"
#include "lev_plan.h"

using namespace ta;

template <int D> static void run_d(const LevParams &P, bool affine, int trans, uint32_t waves) {
    uint8_t *lds = (uint8_t *)calloc(P.lds_per_wave + 64, 1);
    for (uint32_t w = 0; w < waves; w++) {
        if (affine) {
            if (trans == 1) LevBand<EmuWave, D, true, 1>::run(P, w, lds);
            else if (trans == 2) LevBand<EmuWave, D, true, 2>::run(P, w, lds);
            else LevBand<EmuWave, D, true, 0>::run(P, w, lds);
        } else {
            if (trans == 1) LevBand<EmuWave, D, false, 1>::run(P, w, lds);
            else if (trans == 2) LevBand<EmuWave, D, false, 2>::run(P, w, lds);
            else LevBand<EmuWave, D, false, 0>::run(P, w, lds);
        }
    }
    free(lds);
}

static int g_force_ch = 0;
extern "C" void emu_lev_set_chunk(int ch) { g_force_ch = ch; }   // 0 = planner's choice, else 16 / 32 / 64

extern "C" int emu_lev_band(const uint8_t *a_blob, const uint64_t *a_off, const uint8_t *b_blob, const uint64_t *b_off,
                            uint32_t n, uint32_t k, uint32_t mc, uint32_t gc, uint32_t sg, int has_t, uint32_t tc,
                            uint64_t max_len, int force_D, int force_L, int force_affine, uint32_t *out,
                            uint32_t *plan_out /* D, L, PW, u, o */) {
    LevPlan pl = lev_make_plan(k, mc, gc, sg, max_len, force_D, force_L, g_force_ch);
    if (!pl.ok) return 1;
    LevParams P;
    P.a = StrView{a_blob, a_off, 0, 0};
    P.b = StrView{b_blob, b_off, 0, 0};
    P.subset = nullptr; P.trace = nullptr; P.out = out; P.n = n; P.k = k;
    P.mc = mc; P.gc = gc; P.sg = sg; P.tc = tc;
    P.u = pl.u; P.o = pl.o; P.L = pl.L; P.PW = pl.PW; P.lds_per_wave = pl.lds_per_wave; P.Tw = pl.Tw; P.ch = pl.ch;
    if (plan_out) { plan_out[0] = pl.D; plan_out[1] = pl.L; plan_out[2] = pl.PW; plan_out[3] = pl.u; plan_out[4] = pl.o; }
    uint32_t waves = (n + pl.PW - 1) / pl.PW;
    bool affine = sg > 0 || force_affine;
    int trans = !has_t ? 0 : ((2u * mc <= 255u + tc && !(force_affine & 2)) ? 1 : 2);   // force_affine bit 1: force the select form
    switch (pl.D) {
#define CASE(d) case d: run_d<d>(P, affine, trans, waves); break;
        CASE(2) CASE(4) CASE(6) CASE(8) CASE(10) CASE(12) CASE(16) CASE(18) CASE(20) CASE(22) CASE(24) CASE(28)
        CASE(32) CASE(34) CASE(40) CASE(48) CASE(56) CASE(66)
#undef CASE
        default: return 2;
    }
    return 0;
}

// ---- bit-parallel band kernel (lev_bits_body.h)
#include "lev_bits_body.h"

template <int NA> static void run_bits(const LevParams &P, bool trans, uint32_t waves) {
    uint8_t *lds = (uint8_t *)calloc(P.lds_per_wave + 64, 1);
    for (uint32_t w = 0; w < waves; w++) {
        if (trans) LevBits<EmuWave, NA, true>::run(P, w, lds);
        else LevBits<EmuWave, NA, false>::run(P, w, lds);
    }
    free(lds);
}

extern "C" int emu_lev_bits(const uint8_t *a_blob, const uint64_t *a_off, const uint8_t *b_blob, const uint64_t *b_off,
                            uint32_t n, uint32_t k, int has_t, uint64_t max_len, int force_NA, uint32_t *out,
                            uint32_t *plan_out /* NA, u, Tw */) {
    LevBitsPlan pl = lev_bits_make_plan(k, 1, 1, 0, has_t != 0, 1, max_len, force_NA, g_force_ch);
    if (!pl.ok) return 1;
    LevParams P;
    P.a = StrView{a_blob, a_off, 0, 0};
    P.b = StrView{b_blob, b_off, 0, 0};
    P.subset = nullptr; P.trace = nullptr; P.out = out; P.n = n; P.k = k;
    P.mc = 1; P.gc = 1; P.sg = 0; P.tc = has_t ? 1 : 0;
    P.u = pl.u; P.o = 0; P.L = 1; P.PW = 64; P.lds_per_wave = pl.lds_per_wave; P.Tw = pl.Tw; P.ch = pl.ch;
    if (plan_out) { plan_out[0] = pl.NA; plan_out[1] = pl.u; plan_out[2] = pl.Tw; }
    const uint32_t waves = (n + 63) / 64;
    switch (pl.NA) {
#define CASE(d) case d: run_bits<d>(P, has_t != 0, waves); break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12)
        CASE(13) CASE(14) CASE(15) CASE(16) CASE(18) CASE(20) CASE(22) CASE(24) CASE(26) CASE(28) CASE(30) CASE(32)
#undef CASE
        default: return 2;
    }
    return 0;
}

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
