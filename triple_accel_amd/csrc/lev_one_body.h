// lev_one_body.h -- ONE pair, one wavefront, shortest time to the answer: the kernel behind the single-call entry points
// (levenshtein_simd_k, levenshtein_simd_k_with_opts, ... src/levenshtein.rs:677, :714) for the unit-cost families when the band
// fits 64 diagonals.
//
// A lone pair cannot fill a GPU; what counts is the length of the dependent chain.  The bit-parallel band recurrence of
// lev_bits_body.h is serial in the columns, but only its ~18 bitwise operations per column are: the match vector of a column (the
// byte compares, about as much work again per 32 diagonals) depends on nothing.  So the wavefront splits the two:
//   * PARALLEL, 64 columns at a time: lane t builds the match vector of column j0 + t -- its window of `a` comes out of LDS (the
//     whole string is staged there once, XOR 0x0C) with unaligned ds_read_b32, the byte test is the v_perm_b32 / v_dot4 one --
//     and leaves it in two VGPRs as a 64-bit word already positioned on the window's bits;
//   * SERIAL: v_readlane_b32 hands column after column to the SCALAR unit, where the whole 64-bit window is one SGPR pair:
//     D0 = (((PM & VP) + VP) ^ VP) | PM | VN and the rest are s_and_b64 / s_or_b64 / s_xor_b64 / s_lshr_b64, the carry chain one
//     s_add_u32 + s_addc_u32, the zero-difference count on the answer diagonal s_bitcmp1_b64 -- no multi-word carries, no masks.
// One wavefront issues about one instruction per four cycles whatever the unit, so a column costs ~20 issue slots instead of the
// band kernel's ~58 (all of which it would run on ONE useful lane for a lone pair).
//
// Window: 64 bits, or 32 when the band fits (one v_readlane and the carry less per column).  Bit i <-> diagonal d_hi - i with
// d_hi = 63 (31) - nlo; the band of lev_plan.h (w = unit_k + 1 diagonals,
// + 2 rows for the transposition test) occupies the TOP w bits (the lowest diagonal -nlo is bit 63), the bits below it are
// diagonals above the band and are fed "mismatch": their cells can only over-estimate, which never changes an answer <= k.
// Result contract as everywhere: d if d <= k else None (src/levenshtein.rs:539-541); None at once if |n - m| > unit_k (:426-428).
#pragma once
#include <type_traits>

#include "lev_band_body.h"

namespace ta {

constexpr uint32_t LEV_ONE_PAD_LO = 128, LEV_ONE_PAD_HI = 192;     // LDS bytes before / after the staged string (rows outside [1, n])
constexpr uint32_t LEV_ONE_MAX_LEN = 32000;                        // the staged string must fit the 64 KB a block may ask for
constexpr uint32_t LEV_ONE_MAX_W = 64;

// WIDE: 64-bit window (bands of 33..64 rows); else 32 bits -- one v_readlane and no carry per column less
template <class W, bool TRANS, bool WIDE>
struct LevOne {
    static constexpr uint32_t WBITS = WIDE ? 64u : 32u;
    typedef typename std::conditional<WIDE, uint64_t, uint32_t>::type Win;
    using U32 = typename W::U32;
    using Bool = typename W::Bool;
    using Ptr = typename W::Ptr;
    using Q = typename W::Q;

    static TA_HD inline void run(const LevParams &P, uint8_t *lds) {
        const U32 lane = W::lane();
        const Bool all = (lane == lane);
        const U32 pair = P.subset ? W::load_u32(P.subset, W::splat(0), all, 0u) : W::splat(0);
        Ptr aptr, bptr;
        U32 alen_v, blen_v;
        W::load_str(P.a, pair, all, aptr, alen_v);      // rows
        W::load_str(P.b, pair, all, bptr, blen_v);      // columns
        const uint32_t alen = W::readlane(alen_v, 0), blen = W::readlane(blen_v, 0);

        const uint32_t diff = blen >= alen ? blen - alen : alen - blen;
        const bool inband = diff <= P.u;
        const uint32_t w = P.u + 1u + (TRANS ? 2u : 0u);                      // band rows per column (<= 64: the launcher's promise)
        const uint32_t nlo = inband ? ((P.u - diff) >> 1) + (blen >= alen ? 0u : diff) + (TRANS ? 1u : 0u) : 0u;
        const uint32_t dhi = WBITS - 1u - nlo, ans = inband ? dhi + alen - blen : 0u, sh = WBITS - w;
        const uint64_t band = (w >= 64u ? ~0ull : (((1ull << w) - 1ull) << sh)) & (WIDE ? ~0ull : 0xFFFFFFFFull);

        auto load_b = [&](uint32_t j0) {                                        // lane t: b[j0 + t] (column j0 + t + 1)
            const U32 x = W::splat(j0) + lane;
            return W::gload_u8(W::ptr_add(bptr, W::sel(x < blen, x, W::splat(0))), x < blen);
        };
        U32 c_next = load_b(0);                                                 // in flight while `a` is staged
        // ---- stage `a` between two pads
        for (uint32_t off = 0; off < LEV_ONE_PAD_LO; off += 256u) W::lds_write32(lds, W::splat(off) + lane * 4u, W::splat(0));
        for (uint32_t off = 0; off < alen; off += 1024u) {
            const U32 x = W::splat(off) + lane * 16u;
            const Q q = W::gload16(W::ptr_add(aptr, x), x < alen);           // (blobs carry 16 bytes of slack past the string)
            W::lds_store16(lds, x + LEV_ONE_PAD_LO, q, x < alen);
        }
        {   // bytes past the string up to the pad's end: whatever the last 16-byte piece left there is fine (rows > n never
            // feed rows <= n), but the reads must stay inside the block's LDS -- which the launcher sized
        }
        W::lds_wave_sync();

        // ---- column 0: D[r][0] = |r|; rows r = 1 - d_hi + i >= 1 step up (+1), rows <= 0 step down (-1)
        const Win below = (Win)(dhi >= 64u ? ~0ull : ((1ull << dhi) - 1ull));
        Win VP = (Win)~below, VN = below, PMp = 0, D0p = (Win)~(Win)0;
        uint32_t cnt = 0;

        // lane t's match vector for column j = j0 + t + 1, positioned: bit i set <=> a[j - d_hi + i - 1] == b[j - 1], band bits only
        auto match_vectors = [&](uint32_t j0, U32 c, U32 &pm_lo, U32 &pm_hi) {
            // window bit i <-> LDS byte PAD + (j - d_hi + i) - 1; the band's bits i = sh .. 63 are w consecutive bytes from `base`
            const U32 base = W::splat(LEV_ONE_PAD_LO + j0 + 1u + sh - 1u - dhi) + lane;       // (PAD + sh >= d_hi always)
            const U32 Bs = W::splat_byte(c) ^ 0x0C0C0C0Cu;                       // the byte test looks for 12: a ^ b ^ 0x0C
            U32 w0 = W::splat(0), w1 = W::splat(0);                             // mismatch bits of band bytes 0..31 / 32..63
            const uint32_t nd = (w + 3u) / 4u;
            for (uint32_t m = 0; m < nd; m++) {
                const U32 d = W::lds_read32u(lds, base + 4u * m);
                const U32 nib = W::sdot4_first(W::ne12(d ^ Bs), W::splat(0xF8FCFEFFu));     // 4 mismatch flags -> 4 bits
                if (m < 8u) w0 = w0 | (nib << (4u * m));
                else w1 = w1 | (nib << (4u * (m - 8u)));
            }
            // ne = (w1:w0) << sh, then PM = ~ne & band  (sh uniform; a 32-bit window has no high word)
            U32 lo, hi;
            if (!WIDE) { lo = sh ? (w0 << sh) : w0; hi = W::splat(0); }
            else if (sh == 0u) { lo = w0; hi = w1; }
            else if (sh < 32u) { lo = w0 << sh; hi = (w1 << sh) | (w0 >> (32u - sh)); }
            else if (sh == 32u) { lo = W::splat(0); hi = w0; }
            else { lo = W::splat(0); hi = w0 << (sh - 32u); }
            pm_lo = ~lo & (uint32_t)band;
            pm_hi = ~hi & (uint32_t)(band >> 32);
        };

        for (uint32_t j0 = 0; j0 < blen; j0 += 64u) {
            U32 pm_lo, pm_hi;
            match_vectors(j0, c_next, pm_lo, pm_hi);
            if (j0 + 64u < blen) c_next = load_b(j0 + 64u);                     // in flight while the serial part runs
            const uint32_t nt = blen - j0 < 64u ? blen - j0 : 64u;
            auto column = [&](uint32_t t) {
                const Win PM = WIDE ? (Win)(((uint64_t)W::readlane(pm_hi, t) << 32) | (uint64_t)W::readlane(pm_lo, t)) : (Win)W::readlane(pm_lo, t);
                Win D0 = (Win)((((PM & VP) + VP) ^ VP) | PM) | VN;              // Hyyro 2003
                if (TRANS) D0 |= (Win)~D0p & (Win)(PM << 1) & (Win)(PMp >> 1);  // src/levenshtein.rs:517-525
                const Win HP = VN | (Win)~(D0 | VP), HN = D0 & VP;
                const Win D0s = D0 >> 1;                                        // the window moves one row down
                VP = HN | (Win)~(D0s | HP);
                VN = D0s & HP;
                cnt += (uint32_t)((D0 >> ans) & (Win)1);
                if (TRANS) { PMp = PM; D0p = D0; }
            };
            uint32_t t = 0;
            for (; t + 4u <= nt; t += 4u) { column(t); column(t + 1u); column(t + 2u); column(t + 3u); }
            for (; t < nt; t++) column(t);
        }
        const uint32_t d = diff + blen - cnt;                                   // |delta| + columns - zero-difference steps
        W::store_u32(P.out, pair, W::splat(inband && d <= P.k ? d : 0xFFFFFFFFu), lane == 0u);
    }
};

}  // namespace ta
