// lev_plan.h -- host-side launch planning for the band-wavefront kernel (lev_band_body.h).
// Pure integer logic, shared by the product's C ABI (ta_api.hip) and the test-only emulation.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace ta {

// Diagonals-per-lane values the kernel is instantiated for (all even).
static const int LEV_D_SET[] = {2, 4, 6, 8, 10, 12, 16, 18, 20, 22, 24, 28, 32, 34, 40, 48, 56, 66};
static const int LEV_D_COUNT = (int)(sizeof(LEV_D_SET) / sizeof(LEV_D_SET[0]));

struct LevPlan {
    uint32_t u, o, need;     // unit_k of the batch, diagonal index of d=0 for an equal-length pair, diagonal slots (u+2)
    int D;                   // diagonals per lane
    uint32_t L, PW;          // lanes per pair, pairs per wave
    uint32_t lds_per_wave;
    uint32_t Tw;             // warm-up iterations
    uint32_t ch;             // bytes per string per streamed chunk (LDS ring = 2 chunks)
    bool ok;                 // false: band too wide for one wavefront (needs the wide-band kernel)
};

static inline uint32_t lev_sat_sub(uint32_t a, uint32_t b) { return a > b ? a - b : 0u; }

// The band.  An alignment of cost <= K that strays t diagonals outside [min(0,delta), max(0,delta)]
// (delta = b_len - a_len) pays for at least 2t + |delta| gap characters and one gap opening, so
//     2t + |delta| <= (K - sg) / gc = unit_k                     (unit_k as in src/levenshtein.rs:426, :760-763)
// and the diagonals d = j - i in [min(0,delta) - t, max(0,delta) + t], t = (unit_k - |delta|) / 2, hold every such
// alignment: at most unit_k + 1 diagonals, half of the reference's [-unit_k, unit_k] (:434-438, :866).  Results are
// unchanged -- inside either band every cell on an optimal alignment of cost <= K is exact, and a banded value is
// never below the true one, so "d <= k ? Some(d) : None" (:539-541) comes out the same.
// A pair's diagonal index is p = d + o_pair with o_pair odd (dp(0,0) must sit on an odd slot):
static inline uint32_t lev_pair_offset(uint32_t u, uint64_t a_len, uint64_t b_len) {
    const uint64_t diff = a_len > b_len ? a_len - b_len : b_len - a_len;
    const uint64_t t = diff <= u ? (u - diff) >> 1 : 0;
    return (uint32_t)(t + (a_len > b_len ? a_len - b_len : 0)) | 1u;
}

// unit_k of a batch whose strings are at most max_len long: K = min(k, the largest distance any such pair can have)
// -- the dispatcher's max_k clamp (:734-757) taken over the batch -- then (K - sg) / gc, and never more than the
// 2 max_len diagonals a matrix has.
static inline uint32_t lev_batch_unit_k(uint32_t k, uint32_t mc, uint32_t gc, uint32_t sg, uint64_t max_len) {
    const uint64_t all_gaps = 2 * max_len * gc + 2ull * sg;                       // delete a, insert b
    const uint64_t subs = max_len * (mc > gc ? mc : gc) + sg;                     // substitute, then one gap run
    uint64_t K = all_gaps < subs ? all_gaps : subs;
    if (K > k) K = k;
    uint64_t u = (K > sg ? K - sg : 0) / gc;
    if (u > 2 * max_len) u = 2 * max_len;
    return (uint32_t)(u > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : u);
}

// force_D > 0 pins D (tuning / tests).
static inline LevPlan lev_make_plan(uint32_t k, uint32_t mc, uint32_t gc, uint32_t sg, uint64_t max_len, int force_D, int force_L,
                                    int force_ch = 0) {
    LevPlan p;
    p.u = lev_batch_unit_k(k, mc, gc, sg, max_len);
    p.o = (p.u >> 1) | 1u;
    p.need = p.u + 2u;
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
        // LDS ring: 2 chunks per (pair, string).  64-byte chunks when a wave holds <= 32 pairs; 32-byte ones at 64
        // pairs per wave, so that a wave's rings stay under ~8.5 KB and 4 waves per SIMD fit next to each other.
        p.ch = (force_ch == 16 || force_ch == 32 || force_ch == 64) ? (uint32_t)force_ch : (p.PW > 32 ? 32u : 64u);
        p.lds_per_wave = (2u * p.PW * (2u * p.ch + 4u) + 15u) & ~15u;
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

// ---- costs that are a unit-cost family times g: EditCosts(g, g, 0, None) or (g, g, 0, Some(g)), g >= 2.  Every alignment then costs g
// times its unit cost, so  d = g d_unit  and  d <= k  <=>  d_unit <= k / g: the pass runs the unit-cost (bit-parallel) kernels with
// k / g and multiplies the answers -- the one family of weighted costs that rides the bit-vector kernels as it is.  0: not such costs.
static inline uint32_t lev_unit_scale(uint32_t mc, uint32_t gc, uint32_t sg, bool has_t, uint32_t tc) {
    return (sg == 0 && mc == gc && gc >= 2 && (!has_t || tc == gc)) ? gc : 0u;
}

// ---- bit-parallel band kernel (lev_bits_body.h): unit costs only, one pair per lane, window of 4*NA diagonals
static const int LEV_BITS_MAX_NA = 32;      // kernels exist for NA = 1..16 and the even NA up to 32
static const uint32_t LEV_BITS_S8_MIN = 25; // narrowest band (diagonals) the planner gives to the stride-8 form
static const uint32_t LEV_BITS_VLINE_LDS = 64u * 156u;   // VLINE fetch form: per pair two rings of 64 + 8 bytes, an 8-byte dump (+ 4: an odd dword stride)

struct LevBitsPlan {
    bool ok;                 // false: costs are not a unit-cost family, or the band is wider than the window
    uint32_t u;              // unit_k of the batch
    int NA;                  // packed dwords of `a` under the window (window = 4*NA bits, 4*NA - 3 in the static form)
    bool stat;               // static form: the window registers move a dword every 4th column (lev_bits_body.h)
    bool s8;                 // stride-8 form: 33 diagonals in 8 registers, no shifts and no moves (lev_bits_body.h); NA = 8, stat = false
    uint32_t Tw, ch, lds_per_wave;
};

// force_static: 0 = planner's choice, 1 = sliding form, 2 = static form, 3 = stride-8 form (bands of up to 33 diagonals)
static inline LevBitsPlan lev_bits_make_plan(uint32_t k, uint32_t mc, uint32_t gc, uint32_t sg, bool has_t, uint32_t tc,
                                             uint64_t max_len, int force_NA = 0, int force_ch = 0, int force_static = 0) {
    LevBitsPlan p;
    p.ok = mc == 1 && gc == 1 && sg == 0 && (!has_t || tc == 1);   // LEVENSHTEIN_COSTS / RDAMERAU_COSTS (src/levenshtein.rs:79-91)
    p.u = lev_batch_unit_k(k, mc, gc, sg, max_len);
    const uint64_t w = (uint64_t)p.u + 1u + (has_t ? 2u : 0u);     // the transposition test looks one row past each band edge
    uint64_t na = (w + 3) / 4, na_st = (w + 3 + 3) / 4;            // the static form gives up 3 window bits
    if (na_st < 8) na_st = 8;                                       // static kernels exist for NA >= 8
    if (force_NA > 0 && (uint64_t)force_NA >= na) na = (uint64_t)force_NA;
    if (force_NA > 0 && (uint64_t)force_NA >= na_st) na_st = (uint64_t)force_NA;
    if (na > 16) na += na & 1;
    if (na_st > 16) na_st += na_st & 1;
    // one more dword of compares (5 instructions) buys back NA v_alignbyte per column: worth it from 8 dwords on
    p.stat = force_static == 2 || (force_static == 0 && na >= 8 && na_st <= (uint64_t)LEV_BITS_MAX_NA);
    if (p.stat && na_st > (uint64_t)LEV_BITS_MAX_NA) p.stat = false;
    if (p.stat) na = na_st;
    if (na > (uint64_t)LEV_BITS_MAX_NA) { p.ok = false; na = LEV_BITS_MAX_NA; }
    // bands of LEV_BITS_S8_MIN..33 diagonals: the stride-8 form (39 instructions per column whatever the width; the sliding form
    // needs 7 dwords from 25 diagonals on: 44, the static one 43 / 48 at 8 / 9 dwords)
    p.s8 = force_NA <= 0 && w <= 33u && (force_static == 3 || (force_static == 0 && w >= LEV_BITS_S8_MIN));
    if (force_static == 3 && !p.s8) p.ok = false;
    if (p.s8) { p.stat = false; na = 8; }
    p.NA = (int)na;
    (void)force_ch;
    p.ch = 64u;                                                    // one 64-byte line per string per refill
    p.lds_per_wave = 64u * (84u + 68u);                            // per pair: `a` 64 + 16 look-ahead + 4 pad, `b` 64 + 4 pad
    // columns start at iteration Tw >= the deepest band (rows that must slide in first); a multiple of 64 keeps the
    // chunks of `b` on 64-byte lines
    p.Tw = (p.u + (has_t ? 1u : 0u) + 63u) & ~63u;
    return p;
}

// ---- two pairs per lane (lev_bits2_body.h): fixed-length unit-cost batches whose band (+ the transposition test's two extra
// rows) is at most 15 diagonals wide, and big enough that halving the number of wavefronts still leaves >= 2 per SIMD
struct LevBits2Plan {
    bool ok;
    int NA;                  // window registers (8: the stride-8 layout, 15 diagonals per pair)
    uint32_t u, Tw, lds_per_wave;
};
constexpr uint32_t LEV_BITS2_MIN_PAIRS = 262144;     // 128 pairs per wavefront x 2 wavefronts x 1024 SIMDs
static inline LevBits2Plan lev_bits2_make_plan(uint32_t k, uint32_t mc, uint32_t gc, uint32_t sg, bool has_t, uint32_t tc, uint64_t max_len,
                                               bool fixed_length, uint64_t pairs) {
    LevBits2Plan p;
    p.u = lev_batch_unit_k(k, mc, gc, sg, max_len);
    const uint64_t w = (uint64_t)p.u + 1u + (has_t ? 2u : 0u);
    p.ok = mc == 1 && gc == 1 && sg == 0 && (!has_t || tc == 1) && fixed_length && w <= 15 && max_len <= 65000 && max_len >= 1 &&
           pairs >= LEV_BITS2_MIN_PAIRS;
    p.NA = 8;                                                  // the stride-8 window: 8 registers of 2 + 2 bytes
    p.Tw = (p.u + (has_t ? 1u : 0u) + 2u + 63u) & ~63u;        // (the stream of `a` runs two iterations ahead of the window's last row)
    p.lds_per_wave = 128u * (36u + 20u);                       // rings of 2 + 1 sixteen-byte pieces per pair (+ wrap copies)
    return p;
}

// ---- score form of the DP band kernel (lev_band_body.h, SCORE): the cells hold gc (i+j) - dp, so the substitution adds the byte
// 2 gc - mc [a != b]: both bytes must be in 0..255, and so must the transposition's 4 gc - tc (>= 0 always: tc / 2 < gc).
// trans: 0 none, 1 dot4 penalty, 2 select form (cost form only).
static inline bool lev_score_form_applies(uint32_t mc, uint32_t gc, int trans, uint32_t tc = 0) {
    return trans != 2 && 2u * gc <= 255u && mc <= 2u * gc && (trans == 0 || (tc <= 4u * gc && 4u * gc - tc <= 255u));
}

// ---- small alphabets (lev_bitsq_body.h): fixed-length unit-cost batches whose band (+ the transposition test's two extra rows) fits
// the 33-diagonal window; the caller names at most four symbols (the launcher checks that a code hash exists for them)
constexpr uint32_t LEV_BITSQ_MIN_PAIRS = 16384;
static inline bool lev_bitsq_applies(uint32_t k, uint32_t mc, uint32_t gc, uint32_t sg, bool has_t, uint32_t tc, uint64_t max_len,
                                     bool fixed_length, uint64_t pairs, uint32_t *u_out) {
    const uint32_t u = lev_batch_unit_k(k, mc, gc, sg, max_len);
    if (u_out) *u_out = u;
    return mc == 1 && gc == 1 && sg == 0 && (!has_t || tc == 1) && fixed_length && (uint64_t)u + 1u + (has_t ? 2u : 0u) <= 33u &&
           max_len >= 1 && max_len <= 0x7FFFFF00ull && pairs >= LEV_BITSQ_MIN_PAIRS;
}

// h in 0..6 with ((s >> h) & 3) distinct over the n <= 4 distinct symbols, and the table whose byte c is the symbol of code c
// (a byte that does NOT hash to c where no symbol does: nothing matches it).  false: no such shift (or bad arguments).
static inline bool lev_bitsq_hash(const uint8_t *sym, size_t n, uint32_t *shift_out, uint32_t *table_out) {
    if (!sym || n == 0 || n > 4) return false;
    for (size_t i = 0; i < n; i++)
        for (size_t j = i + 1; j < n; j++)
            if (sym[i] == sym[j]) return false;
    for (uint32_t h = 0; h <= 6; h++) {
        uint32_t seen = 0, table = 0;
        bool ok = true;
        for (size_t i = 0; i < n && ok; i++) {
            const uint32_t c = (sym[i] >> h) & 3u;
            if (seen & (1u << c)) ok = false;
            seen |= 1u << c;
            table |= (uint32_t)sym[i] << (8 * c);
        }
        if (!ok) continue;
        for (uint32_t c = 0; c < 4; c++)
            if (!(seen & (1u << c))) table |= ((((c + 1u) & 3u) << h) & 0xFFu) << (8 * c);
        *shift_out = h; *table_out = table;
        return true;
    }
    return false;
}

// The same for alphabets of 5 .. 32 symbols (lev_bitsqw_body.h): h in 0..3 with ((s >> h) & 31) distinct over the symbols AND the other
// three bits of the byte the same in every symbol (the kernel verifies `a` by comparing them).  memb: bit c = code c is a symbol;
// hi = (mask of the other bits) | (their common value << 8).
static inline bool lev_bitsqw_hash(const uint8_t *sym, size_t n, uint32_t *shift_out, uint32_t *memb_out, uint32_t *hi_out) {
    if (!sym || n == 0 || n > 32) return false;
    for (uint32_t h = 0; h <= 3; h++) {
        const uint32_t other = ~(31u << h) & 0xFFu;
        uint32_t memb = 0;
        bool ok = true;
        for (size_t i = 0; i < n && ok; i++) {
            const uint32_t c = (sym[i] >> h) & 31u;
            if ((memb >> c) & 1u) ok = false;
            memb |= 1u << c;
            if ((sym[i] & other) != (sym[0] & other)) ok = false;
        }
        if (!ok) continue;
        *shift_out = h; *memb_out = memb; *hi_out = other | ((sym[0] & other) << 8);
        return true;
    }
    return false;
}

// ---- ONE pair, band of at most 64 diagonals (lev_one_body.h: match vectors 64 columns at a time, the recurrence on the scalar unit)
static inline bool lev_one_applies(uint32_t k, uint32_t mc, uint32_t gc, uint32_t sg, bool has_t, uint32_t tc, uint64_t max_len, uint32_t *u_out) {
    const uint32_t u = lev_batch_unit_k(k, mc, gc, sg, max_len);
    if (u_out) *u_out = u;
    return mc == 1 && gc == 1 && sg == 0 && (!has_t || tc == 1) && (uint64_t)u + 1u + (has_t ? 2u : 0u) <= 64u && max_len <= 32000u;
}

// ---- which kernel runs a k-bounded pass, and roughly what it costs (wave-instructions per pair; only ratios matter)
enum LevKernel { LEV_K_BAND = 1, LEV_K_WIDE = 2, LEV_K_BITS = 3, LEV_K_WIDEBITS = 4 };

struct LevChoice {
    int kernel;              // LevKernel
    int rows_per_lane;       // LEV_K_WIDEBITS: 32 or 64
    double cost;
};

// `pairs`: how many pairs the pass holds.  Up to LEV_LATENCY_PAIRS of them (about one wavefront per SIMD on the GPU) the
// pass lasts as long as its longest wavefront, so the kernels are compared by the instructions ONE wavefront issues, not
// by the instructions per pair: a kernel that spreads a pair over a whole wavefront then beats one that packs 64 pairs.
constexpr uint32_t LEV_LATENCY_PAIRS = 1024;
static inline LevChoice lev_choose(uint32_t k, uint32_t mc, uint32_t gc, uint32_t sg, bool has_t, uint32_t tc, uint64_t max_len,
                                   bool dp_only, uint32_t pairs = 0xFFFFFFFFu) {
    const double n = (double)(max_len ? max_len : 1);
    const bool latency = pairs <= LEV_LATENCY_PAIRS;
    const LevBitsPlan bp = lev_bits_make_plan(k, mc, gc, sg, has_t, tc, max_len);
    const LevPlan pl = lev_make_plan(k, mc, gc, sg, max_len, 0, 0);
    const bool unit = mc == 1 && gc == 1 && sg == 0 && (!has_t || tc == 1);
    LevChoice c;
    if (bp.ok && !dp_only) {                                   // one pair per lane, ~6 instructions per window dword + 25
        c.kernel = LEV_K_BITS; c.rows_per_lane = 0;
        c.cost = n * (6.0 * bp.NA + 25.0 + (has_t ? 8.0 : 0.0)) / (latency ? 1.0 : 64.0);
        if (!latency) return c;
    } else {
        c.kernel = 0; c.rows_per_lane = 0; c.cost = 1e300;
    }
    // DP band: (4.5 D + 25) instructions per iteration, shared by the PW pairs of a wave; wide DP kernel: ~6 per cell
    const double band = pl.ok ? n * (4.5 * pl.D + 25.0 + (has_t ? 1.5 * pl.D : 0.0)) / (latency ? 1.0 : pl.PW) : 1e300;
    const double uu = pl.u < n ? (double)pl.u : n;
    const double wide = (2.0 * n * uu - uu * uu) * 6.0 / 64.0 + n * 40.0;
    if (c.kernel == 0) {                                       // (the bit-parallel band kernel, where it applies, beats the DP kernels)
        c.kernel = band <= wide ? LEV_K_BAND : LEV_K_WIDE; c.rows_per_lane = 0;
        c.cost = band <= wide ? band : wide;
    }
    if (unit && !dp_only) {                                    // one pair per wave, one column of a 2048/4096-row stripe per ~45 instructions
        const int rpl = max_len > 2048 ? 64 : 32;
        const double rows = 64.0 * rpl, stripes = (double)((max_len + (uint64_t)rows - 1) / (uint64_t)rows);
        const double cols = (rows + pl.u < n ? rows + pl.u : n) + 64.0;          // columns a stripe visits inside the band
        const double wb = stripes * cols * (rpl == 64 ? 45.0 : 35.0) + (has_t ? stripes * cols * 10.0 : 0.0) + 600.0 * stripes;
        if (wb < c.cost) { c.kernel = LEV_K_WIDEBITS; c.rows_per_lane = rpl; c.cost = wb; }
    }
    return c;
}

// Geometry of the pair-sliced systolic band kernel (lev_sliced.hip) for a fixed-length batch: the band of 3.1 cut into S
// strips of three window cells (S odd: both entry lanes of a DPP row then hold the same set), 9 <= S <= 15.
struct LevSlicedPlan {
    bool ok;
    uint32_t S;            // strips per group
    int32_t dhi;           // highest diagonal (j - i) of the band: window cell w of column j is row j - dhi + w
    uint32_t c_ans, e_ans; // strip and cell of the answer diagonal (window cell dhi - delta)
    uint32_t dabs;         // |b_len - a_len|
    uint32_t steps;        // time steps: every strip through b_len columns, rounded to whole 64-step epochs
};
inline LevSlicedPlan lev_sliced_make_plan(uint64_t a_len, uint64_t b_len, uint32_t unit_k) {
    LevSlicedPlan p = {};
    if (a_len == 0 || b_len == 0 || a_len > 0x7FFFFFF0ull || b_len > 512ull) return p;
    const uint64_t dabs = a_len > b_len ? a_len - b_len : b_len - a_len;
    if (dabs > unit_k) return p;
    const uint32_t t = (unit_k - (uint32_t)dabs) / 2u, W = (uint32_t)dabs + 2u * t + 1u;
    p.S = ((W + 2u) / 3u) | 1u;
    p.dabs = (uint32_t)dabs;
    p.dhi = (int32_t)((b_len > a_len ? (uint32_t)dabs : 0u) + t);
    const uint32_t w_ans = (uint32_t)((int64_t)p.dhi - ((int64_t)b_len - (int64_t)a_len));
    p.c_ans = w_ans / 3u; p.e_ans = w_ans % 3u;
    p.steps = ((2u * (uint32_t)b_len + p.S + 2u) + 63u) / 64u * 64u;
    p.ok = p.S >= 9u && p.S <= 15u;
    return p;
}

// A unit-cost threshold that bounds a weighted one from below: for any EditCosts,
//     k' = max( k / min(mc, tc),  (k - sg) / min(mc, gc, tc) )              (tc only when the costs have a transposition)
// An alignment the reference's recurrence prices at C <= k (src/levenshtein.rs:1709-1806, :434-541: every cell value is the cost of one
// concrete path) has x mismatches, g gap characters in o >= [g > 0] gap runs and t transpositions: C = x mc + g gc + o sg + t tc.
// Its number of unit edits U = x + g + t (the unit restricted-Damerau recurrence counts a transposition as ONE edit) obeys
// U min(mc, gc, tc) <= C - o sg: with a gap that is <= k - sg, without one g = 0 and U min(mc, tc) <= k.  The unit-cost distance is the
// minimum over all paths, so it is <= U <= k': "unit distance > k'" implies "weighted distance > k".  Unit costs times g give k / g exactly.
// Users: the search's candidate filter (lev_search_body.h: srch_filter_k) and the unit-cost pre-pass of weighted pair batches
// (TA_OPT_UNIT_PREFILTER, ta_api.hip).
static inline uint32_t lev_unit_filter_k(uint32_t k, uint32_t mc, uint32_t gc, uint32_t sg, bool has_t, uint32_t tc) {
    uint32_t m_nogap = mc, m_all = mc < gc ? mc : gc;
    if (has_t) {
        if (tc < m_nogap) m_nogap = tc;
        if (tc < m_all) m_all = tc;
    }
    if (m_all == 0) return 0xFFFFFFFFu;                       // (no bound: a free edit)
    const uint32_t with_gap = (k > sg ? k - sg : 0u) / m_all, without = k / m_nogap;
    return with_gap > without ? with_gap : without;
}

}  // namespace ta
