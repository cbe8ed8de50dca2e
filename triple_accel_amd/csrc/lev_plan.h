// lev_plan.h -- host-side launch planning for the band-wavefront kernel (lev_band_body.h).
// Pure integer logic, shared by the product's C ABI (ta_api.hip) and the test-only emulation.
#pragma once
#include <stdint.h>

namespace ta {

// Diagonals-per-lane values the kernel is instantiated for (all even).
static const int LEV_D_SET[] = {2, 4, 6, 8, 10, 12, 16, 18, 20, 22, 24, 28, 32, 34, 40, 48, 56, 66};
static const int LEV_D_COUNT = (int)(sizeof(LEV_D_SET) / sizeof(LEV_D_SET[0]));

struct LevPlan {
    uint32_t u, o, need;     // band half width, diagonal index of d=0, diagonals needed (o+u+1)
    int D;                   // diagonals per lane
    uint32_t L, PW;          // lanes per pair, pairs per wave
    uint32_t lds_per_wave;
    uint32_t Tw;             // warm-up iterations
    bool ok;                 // false: band too wide for one wavefront (needs the wide-band kernel)
};

static inline uint32_t lev_sat_sub(uint32_t a, uint32_t b) { return a > b ? a - b : 0u; }

// u: the farthest any alignment of cost <= k strays from the main diagonal (src/levenshtein.rs:760-763);
// max_len clamps it exactly like the dispatcher does.  force_D > 0 pins D (tuning / tests).
static inline LevPlan lev_make_plan(uint32_t k, uint32_t gc, uint32_t sg, uint64_t max_len, int force_D, int force_L) {
    LevPlan p;
    uint64_t u64 = lev_sat_sub(k, sg) / gc;
    if (u64 > max_len) u64 = max_len;
    p.u = (uint32_t)u64;
    p.o = p.u | 1u;
    p.need = p.o + p.u + 1u;
    p.ok = false;
    p.D = 0; p.L = 0; p.PW = 0; p.lds_per_wave = 0;
    double best = 1e30;
    for (int t = 0; t < LEV_D_COUNT; t++) {
        int D = LEV_D_SET[t];
        if (force_D > 0 && D != force_D) continue;
        uint32_t L = (p.need + D - 1) / D;
        if (force_L > 0) { if ((uint32_t)force_L < L) continue; L = (uint32_t)force_L; }
        if (L > 64) continue;
        uint32_t PW = 64 / L;
        // VALU work per wave-iteration ~ 5 ops per cell (D cells) + ~24 ops of window/loop overhead,
        // shared by PW pairs; registers beyond ~120 halve the occupancy, so penalise very large D.
        double cost = (5.0 * D + 24.0) / PW * (D > 40 ? 1.25 : 1.0);
        if (cost < best) { best = cost; p.D = D; p.L = L; p.PW = PW; p.ok = true; }
    }
    if (p.ok) {
        p.lds_per_wave = (2u * p.PW * 132u + 15u) & ~15u;   // LEV_SLOT bytes per (pair, string)
        // The ring chunk of iteration block kc holds a[64 kc - ea ..) and b[64 kc - eb ..), ea = (Tw - h) rounded up to
        // 16, eb likewise.  Pad the warm-up so that both are multiples of 64: every 64-byte chunk then maps onto ONE
        // 64-byte line of a line-aligned string instead of straddling two (which costs a second HBM fetch when the line
        // has left L2 by the next refill).
        const uint32_t base = p.L * (uint32_t)(p.D / 2), h = (p.o + 1) >> 1;
        auto lined = [](uint32_t c) { uint32_t r = c & 63u; return r == 0 || r >= 49; };   // 16-byte round-up reaches a multiple of 64
        p.Tw = base;
        for (uint32_t w = 0; w < 64; w++)
            if (lined(base + w - h) && lined(w + h)) { p.Tw = base + w; break; }
    }
    return p;
}

}  // namespace ta
