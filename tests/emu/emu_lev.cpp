// emu_lev.cpp -- runs the band-wavefront kernel body on the host, 64 lanes in lock-step.
// TESTS ONLY (see emu_wave.h).  Built by tests/emu/Makefile into tests/emu/libta_emu.so.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "emu_wave.h"
#include "lev_band_body.h"
#include "lev_plan.h"

using namespace ta;

int g_emu_score = -1;   // -1: as the launcher (score form wherever lev_score_form_applies), 0: cost form always
extern "C" void emu_lev_set_score(int v) { g_emu_score = v; }
extern "C" int emu_lev_score_applies(uint32_t mc, uint32_t gc, int trans, uint32_t tc) { return lev_score_form_applies(mc, gc, trans, tc) ? 1 : 0; }

int g_emu_band_line = 0;   // 1: fixed-length batches in the one-lane-per-pair layout take the LINE form of the fetch (score form; as the launcher)
extern "C" void emu_lev_set_line(int v) { g_emu_band_line = v; }

template <int D, bool L1> static void run_dl(const LevParams &P, bool affine, int trans, uint32_t waves) {
    uint8_t *lds = (uint8_t *)calloc(P.lds_per_wave + 64, 1);
    const bool score = g_emu_score != 0 && lev_score_form_applies(P.mc, P.gc, trans, P.tc);
    for (uint32_t w = 0; w < waves; w++) {
        if constexpr (L1) {
            if (score && g_emu_band_line && !P.a.off && !P.b.off) {
                memset(lds, 0xA5, P.lds_per_wave + 64);
                if (affine) {
                    if (trans == 1) LevBand<EmuWave, D, true, 1, false, true, true, true>::run(P, w, lds);
                    else LevBand<EmuWave, D, true, 0, false, true, true, true>::run(P, w, lds);
                } else {
                    if (trans == 1) LevBand<EmuWave, D, false, 1, false, true, true, true>::run(P, w, lds);
                    else LevBand<EmuWave, D, false, 0, false, true, true, true>::run(P, w, lds);
                }
                continue;
            }
        }
        if (score) {
            if (affine) {
                if (trans == 1) LevBand<EmuWave, D, true, 1, false, L1, true>::run(P, w, lds);
                else LevBand<EmuWave, D, true, 0, false, L1, true>::run(P, w, lds);
            } else {
                if (trans == 1) LevBand<EmuWave, D, false, 1, false, L1, true>::run(P, w, lds);
                else LevBand<EmuWave, D, false, 0, false, L1, true>::run(P, w, lds);
            }
        } else if (affine) {
            if (trans == 1) LevBand<EmuWave, D, true, 1, false, L1>::run(P, w, lds);
            else if (trans == 2) LevBand<EmuWave, D, true, 2, false, L1>::run(P, w, lds);
            else LevBand<EmuWave, D, true, 0, false, L1>::run(P, w, lds);
        } else {
            if (trans == 1) LevBand<EmuWave, D, false, 1, false, L1>::run(P, w, lds);
            else if (trans == 2) LevBand<EmuWave, D, false, 2, false, L1>::run(P, w, lds);
            else LevBand<EmuWave, D, false, 0, false, L1>::run(P, w, lds);
        }
    }
    free(lds);
}
// as the launcher (lev_band.hip): the one-lane-per-pair instantiation whenever the plan says L == 1
template <int D> static void run_d(const LevParams &P, bool affine, int trans, uint32_t waves) {
    if (P.L == 1) run_dl<D, true>(P, affine, trans, waves);
    else run_dl<D, false>(P, affine, trans, waves);
}

int g_emu_force_ch = 0;
extern "C" void emu_lev_set_chunk(int ch) { g_emu_force_ch = ch; }   // 0 = planner's choice, else 16 / 32 / 64

// geometry of the pair-sliced kernel (lev_plan.h): out = {ok, S, dhi, c_ans, e_ans, dabs, steps}
extern "C" void emu_sliced_plan(uint64_t a_len, uint64_t b_len, uint32_t unit_k, int64_t *out) {
    const LevSlicedPlan p = lev_sliced_make_plan(a_len, b_len, unit_k);
    out[0] = p.ok; out[1] = p.S; out[2] = p.dhi; out[3] = p.c_ans; out[4] = p.e_ans; out[5] = p.dabs; out[6] = p.steps;
}

extern "C" int emu_lev_band(const uint8_t *a_blob, const uint64_t *a_off, const uint8_t *b_blob, const uint64_t *b_off,
                            uint32_t n, uint32_t k, uint32_t mc, uint32_t gc, uint32_t sg, int has_t, uint32_t tc,
                            uint64_t max_len, int force_D, int force_L, int force_affine, uint32_t *out,
                            uint32_t *plan_out /* D, L, PW, u, o */) {
    LevPlan pl = lev_make_plan(k, mc, gc, sg, max_len, force_D, force_L, g_emu_force_ch);
    if (!pl.ok) return 1;
    LevParams P;
    P.a = StrView{a_blob, a_off, 0, 0};
    P.b = StrView{b_blob, b_off, 0, 0};
    if (g_emu_band_line && n) {                            // a batch whose strings all have one length per side: the strided view the launcher sees
        bool uni = true;
        const uint64_t la = a_off[1] - a_off[0], lb = b_off[1] - b_off[0];
        for (uint32_t i = 0; i < n && uni; i++) uni = (a_off[i + 1] - a_off[i] == la) && (b_off[i + 1] - b_off[i] == lb);
        if (uni && a_off[0] == 0 && b_off[0] == 0) { P.a = StrView{a_blob, nullptr, la, la}; P.b = StrView{b_blob, nullptr, lb, lb}; }
    }
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

