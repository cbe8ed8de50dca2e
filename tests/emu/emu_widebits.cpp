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
