// emu_trace.cpp -- host emulation driver of the checkpoint-and-recompute batch traceback (lev_bits_trace_body.h).  TESTS ONLY (see emu_wave.h).
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "emu_wave.h"
#include "lev_band_body.h"
#include "lev_plan.h"

using namespace ta;

// ---- batch tracebacks by checkpoints + recomputation (lev_bits_trace_body.h): distances in, the scripts' runs (last run first) out
#include "lev_bits_trace_body.h"
#include "lev_plan.h"
// dist[n]: the pass's answers (0xFFFFFFFF = None); runs[n * runs_cap] ((edit type << 29) | count), n_runs[n] out.  tile = 8, 16 or 32.  Reads are range-checked: a byte
// outside the blobs (+ 16 of slack) reads as 0xA5.
// packed form (LevBitsTraceParams::packed_cap): `runs` is then the packed script buffer, packed_cap words per pair, scripts right-aligned
static uint32_t g_packed_cap = 0;
extern "C" void emu_lev_bits_trace_set_packed(uint32_t cap) { g_packed_cap = cap; }
extern "C" int emu_lev_bits_trace(const uint8_t *a_blob, const uint64_t *a_off, uint64_t a_len, const uint8_t *b_blob, const uint64_t *b_off,
                                  uint64_t b_len, uint32_t n, uint32_t u, int has_t, int tile, const uint32_t *dist, uint64_t max_len,
                                  uint32_t *runs, uint32_t runs_cap, uint32_t *n_runs) {
    if (u + 1u + (has_t ? 2u : 0u) > 33u || (tile != 8 && tile != 16 && tile != 32 && tile != 116)) return 1;
    LevBitsTraceParams P;
    P.a = StrView{a_blob, a_off, a_off ? 0 : a_len, a_off ? 0 : a_len};
    P.b = StrView{b_blob, b_off, b_off ? 0 : b_len, b_off ? 0 : b_len};
    P.dist = dist; P.n = n; P.u = u;
    const uint32_t cols_per_tile = tile == 116 ? 16u : (uint32_t)tile;
    P.ckpt_tiles = (uint32_t)((max_len + (uint64_t)cols_per_tile - 1) / (uint64_t)cols_per_tile) + 1u;
    const uint32_t waves = (n + 63) / 64;
    std::vector<uint32_t> ck((size_t)waves * P.ckpt_tiles * 5u * 64u, 0xDEADBEEFu);
    P.ckpt = ck.data(); P.runs = runs; P.runs_cap = runs_cap; P.n_runs = n_runs; P.packed_cap = g_packed_cap;
    struct RangeGuard { ~RangeGuard() { EmuWave::clear_ranges(); } } range_guard;
    EmuWave::clear_ranges();
    EmuWave::add_range(a_blob, (a_off ? a_off[n] : (uint64_t)n * a_len) + 16);
    EmuWave::add_range(b_blob, (b_off ? b_off[n] : (uint64_t)n * b_len) + 16);
    const size_t lds_bytes = 64u * 256u + 64u * 4u * 128u;
    uint8_t *lds = (uint8_t *)malloc(lds_bytes);
    if (tile == 116) {
        // tile 116 = tiles of 16 columns with the forward sweep done by the distance kernel's CKPT instantiation (fixed-length batches: the
        // launcher's route): rows = the shorter string -- the views swapped where a is the longer one -- the line form beyond one line per string
        const bool sw = !a_off && !b_off && a_len > b_len;       // (CSR batches: the kernel swaps pair by pair)
        const LevBitsPlan pl = lev_bits_make_plan(u, 1, 1, 0, has_t != 0, 1, max_len, 0, 0, 3);
        if (!pl.ok || !pl.s8) { free(lds); return 4; }
        std::vector<uint32_t> dist2(n, 0xDEADBEEFu);
        LevParams L;
        L.a = sw ? P.b : P.a; L.b = sw ? P.a : P.b;
        L.subset = nullptr; L.trace = nullptr; L.out = dist2.data(); L.n = n; L.k = u;
        L.mc = 1; L.gc = 1; L.sg = 0; L.tc = has_t ? 1 : 0;
        L.u = pl.u; L.o = 0; L.L = 1; L.PW = 64; L.lds_per_wave = pl.lds_per_wave; L.Tw = pl.Tw; L.ch = pl.ch;
        L.ckpt = ck.data(); L.ckpt_tiles = P.ckpt_tiles;
        const bool line = max_len > 128 && !a_off && !b_off;       // (the line form is the fixed-length batches')
        if (line) L.lds_per_wave = 64u * (52u + 36u); else L.tune |= 1u;
        uint8_t *lds2 = (uint8_t *)calloc((size_t)L.lds_per_wave + 64, 1);
        for (uint32_t w = 0; w < waves; w++) {
            if (has_t) { if (line) LevBits<EmuWave, 8, true, false, true, true, false, false, true>::run(L, w, lds2); else LevBits<EmuWave, 8, true, false, false, true, false, false, true>::run(L, w, lds2); }
            else { if (line) LevBits<EmuWave, 8, false, false, true, true, false, false, true>::run(L, w, lds2); else LevBits<EmuWave, 8, false, false, false, true, false, false, true>::run(L, w, lds2); }
        }
        free(lds2);
        // (the distances of that pass must be the caller's wherever the caller's is <= u: the same kernel code as the plain distance pass)
        for (uint32_t i = 0; i < n; i++) if (dist[i] != 0xFFFFFFFFu && dist[i] <= u && dist2[i] != dist[i]) { free(lds); return 5; }
        for (uint32_t w = 0; w < waves; w++) {
            memset(lds, 0xA5, lds_bytes);
            if (has_t) LevBitsTrace<EmuWave, true, 16, 64, true>::run(P, w, lds); else LevBitsTrace<EmuWave, false, 16, 64, true>::run(P, w, lds);
        }
        free(lds);
        return 0;
    }
    for (uint32_t w = 0; w < waves; w++) {
        memset(lds, 0xA5, lds_bytes);
        // (tile 8: string tiles of 32 columns, the device's default; 16: of 64; 32: of 64)
        if (tile == 8) { if (has_t) LevBitsTrace<EmuWave, true, 8, 32>::run(P, w, lds); else LevBitsTrace<EmuWave, false, 8, 32>::run(P, w, lds); }
        else if (tile == 16) { if (has_t) LevBitsTrace<EmuWave, true, 16, 64>::run(P, w, lds); else LevBitsTrace<EmuWave, false, 16, 64>::run(P, w, lds); }
        else { if (has_t) LevBitsTrace<EmuWave, true, 32, 64>::run(P, w, lds); else LevBitsTrace<EmuWave, false, 32, 64>::run(P, w, lds); }
    }
    free(lds);
    return 0;
}
