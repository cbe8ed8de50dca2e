// lev_bits2_body.h -- the bit-parallel band kernel (lev_bits_body.h) for NARROW bands, TWO pairs per lane.
//
// When the band of a fixed-length batch is at most 16 diagonals wide (unit_k <= 15, or 13 with the transposition term: what
// small thresholds like cfg4's k = 8 give), a pair's column state -- VP, VN and the match vector, one bit per diagonal -- fills
// half a VGPR.  The other half then carries a SECOND pair: lane l of wavefront w owns the pairs 128 w + l (bits 0..15) and
// 128 w + 64 + l (bits 16..31).  Everything after the match vectors -- Hyyro's recurrence (about half of a column's
// instructions for such windows) -- is bitwise and serves both pairs at once; what stays per pair is the byte window of `a`
// (NA dwords each) and its v_xor / v_perm / v_dot4 match test.  The halves must not talk to each other:
//   * the one addition, (PM & VP) + VP, cannot carry from bit 15 into bit 16 because both terms are kept inside the window
//     (VP is masked every column; at most 16 bits wide, its sum with a subset of itself stays below 2^17 only if the window is
//     16 bits -- so windows are limited to 15 bits: WB = 4 NA <= 12, or the planner's 13..15 served by NA = 4 with one bit unused);
//   * right shifts (D0 >> 1, PM' >> 1) move pair B's bit 0 into pair A's bit 15, which lies outside every window and is masked
//     off before it can take part in the addition.
// Same result contract as lev_bits_body.h (d if d <= k else None, src/levenshtein.rs:539-541), same sliding window (one row down
// per column, v_alignbyte), same chunk form of the string streaming with two LDS slots per lane.  Fixed-length batches only:
// the band geometry (rows that slide in first, answer diagonal) is then one number for the whole launch.
#pragma once
#include "lev_bits_body.h"

namespace ta {

template <class W, int NA, bool TRANS>
struct LevBits2 {
    static_assert(NA >= 1 && NA <= 4, "windows of 4..16 diagonals");
    static constexpr int WB = 4 * NA < 15 ? 4 * NA : 15;      // window bits per pair (bit 15 stays free: see the header)
    static constexpr uint32_t WM = (1u << WB) - 1u, WM2 = WM | (WM << 16);
    using U32 = typename W::U32;
    using Bool = typename W::Bool;
    using Ptr = typename W::Ptr;
    using Q = typename W::Q;
    static constexpr uint32_t SLOT_A = 84, SLOT_B = 68;        // as in lev_bits_body.h, per pair
    static constexpr uint32_t LDS_PER_WAVE = 128u * (SLOT_A + SLOT_B);

    struct State {
        U32 VP, VN, PMp, D0p;        // both pairs: bits 0..15 pair A, 16..31 pair B
        U32 AW[2][NA];               // byte windows of `a` (^ 0x0C), one set per pair
    };

    static TA_HD inline __attribute__((always_inline)) void advance_a(U32 (&AW)[NA], U32 a_in) {
#pragma unroll
        for (int k = 0; k < NA - 1; k++) AW[k] = W::template alignbyte<1>(AW[k + 1], AW[k]);
        AW[NA - 1] = W::template alignbyte<1>(a_in ^ 0x0Cu, AW[NA - 1]);
    }

    // mismatch bits of one pair's window against the column character (bit i = window byte i differs), continuing the
    // Horner chain `acc` (the other pair's bits, already in place above)
    static TA_HD inline __attribute__((always_inline)) U32 ne_bits(const U32 (&AW)[NA], U32 b_in, U32 acc, bool have_acc) {
        const U32 Bs = W::splat_byte(b_in);
        U32 ne = acc;
        bool first = !have_acc;
#pragma unroll
        for (int p = 1; p >= 0; p--) {
            const int k0 = 2 * p;
            if (k0 >= NA) continue;
            U32 a2 = first ? W::sdot4_first(W::ne12(AW[k0] ^ Bs), W::splat(0xF8FCFEFFu))
                           : W::sdot4(W::ne12(AW[k0] ^ Bs), W::splat(0xF8FCFEFFu), ne << 8);
            first = false;
            if (k0 + 1 < NA) a2 = W::sdot4(W::ne12(AW[k0 + 1] ^ Bs), W::splat(0x80C0E0F0u), a2);
            ne = a2;
        }
        return ne;
    }

    // One column of both pairs: bA / bB = the pairs' column characters.  ans = bit index of the answer diagonal (the same for
    // both pairs and the whole launch); cnt accumulates the zero-difference steps on it, pair A in bits 0..15, pair B in 16..31.
    static TA_HD inline __attribute__((always_inline)) void column(State &st, U32 bA, U32 bB, uint32_t ans, U32 &cnt) {
        // pair B's bits first: the chain then shifts them up by 16 while pair A's come in (NA <= 2: one byte group per pair, one
        // shift by 16 - 8 = 8 more is needed; handled by the generic (ne << 8) of ne_bits plus the fix-up below)
        U32 ne = ne_bits(st.AW[1], bB, W::splat(0), false);
        if (NA <= 2) ne = ne << 8;                             // one group per pair: lift pair B to bits 8.., ne_bits lifts 8 more
        ne = ne_bits(st.AW[0], bA, ne, true);
        const U32 PM = ~ne & WM2;
        const U32 s = (PM & st.VP) + st.VP;                    // st.VP is inside the windows: no carry leaves bit 15
        U32 D0 = ((s ^ st.VP) | PM) | st.VN;
        if (TRANS) {
            // a[i-1] == b[j-2] && a[i-2] == b[j-1] and the diagonal step before was +1 (src/levenshtein.rs:517-525)
            const U32 pml = PM << 1, pmr = st.PMp >> 1;        // (bit 16 -> 15 of pmr is cleared by the & WM2 below)
            D0 = D0 | (~st.D0p & pml & pmr);
        }
        D0 = D0 & WM2;
        const U32 HP = st.VN | ~(D0 | st.VP);
        const U32 HN = D0 & st.VP;
        const U32 D0s = D0 >> 1;                               // next window's rows; pair B's bit 0 lands on bit 15 (outside WM)
        st.VP = (HN | ~(D0s | HP)) & WM2;
        st.VN = D0s & HP;
        cnt = cnt + ((D0 >> ans) & 0x00010001u);
        if (TRANS) { st.PMp = PM; st.D0p = D0; }
    }

    // (the launcher guarantees: fixed-length batch, band + transposition rows <= WB, P.Tw a multiple of 64)
    static TA_HD inline void run(const LevParams &P, uint32_t wave_index, uint8_t *lds) {
        const U32 lane = W::lane();
        const Bool active = (lane == lane);
        Bool valid[2];
        U32 pair[2];
        Ptr aptr[2], bptr[2];
        U32 alen[2], blen[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const U32 slot_idx = lane + wave_index * 128u + 64u * (uint32_t)h;
            valid[h] = slot_idx < P.n;
            pair[h] = P.subset ? W::load_u32(P.subset, slot_idx, valid[h], 0u) : slot_idx;
            W::load_str(P.a, pair[h], valid[h], aptr[h], alen[h]);
            W::load_str(P.b, pair[h], valid[h], bptr[h], blen[h]);
        }
        // the batch's geometry (lev_plan.h): diagonals d = j - i in [-nlo, d_hi]; window bit i <-> diagonal d_hi - i
        const uint32_t alen_u = (uint32_t)P.a.len, blen_u = (uint32_t)P.b.len;
        const uint32_t diff_u = blen_u >= alen_u ? blen_u - alen_u : alen_u - blen_u;
        const bool inband = diff_u <= P.u;                     // else None for every pair (:426-428, :860-862)
        const uint32_t nlo = inband ? ((P.u - diff_u) >> 1) + (blen_u >= alen_u ? 0u : diff_u) + (TRANS ? 1u : 0u) : 0u;
        const uint32_t dhi = (uint32_t)WB - 1u - nlo;
        const uint32_t ans = inband ? dhi + alen_u - blen_u : 0u;  // row a_len at column b_len

        State st;
        {   // column 0, D[r][0] = |r|: rows r = 1 - d_hi + i >= 1 step up (+1), rows <= 0 step down (-1)
            const uint32_t below = dhi ? ((1u << dhi) - 1u) & WM : 0u;
            st.VN = W::splat(below | (below << 16));
            st.VP = W::splat((~below & WM) | ((~below & WM) << 16));
            st.PMp = W::splat(0);
            st.D0p = W::splat(WM2);
        }
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int k = 0; k < NA; k++) st.AW[h][k] = W::splat(0);
        U32 cnt = W::splat(0);

        // iteration tp inserts a[tp - ca] into the windows and, from tp = T0 on, runs column tp - T0 + 1 with b[tp - T0].  A byte
        // enters at byte 4 NA - 1 of the window registers; with NA = 4 that is byte 15, one above the 15-bit window's top row, so
        // the stream of `a` runs that one iteration ahead (it reaches byte 14 = bit 14 exactly when its row becomes the top row)
        constexpr uint32_t AHEAD = 4u * NA - (uint32_t)WB;
        const uint32_t T0 = P.Tw, ca = T0 - nlo - AHEAD;
        const uint32_t da = (16u - (ca & 15u)) & 15u, ea = ca + da;
        const uint32_t tp0 = ca, iters = T0 + blen_u;

        // ---- string streaming: the chunk form of lev_bits_body.h, one (84 + 68)-byte slot pair per pair
        U32 a_slot[2], b_slot[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            a_slot[h] = (lane + 64u * (uint32_t)h) * SLOT_A;
            b_slot[h] = (lane + 64u * (uint32_t)h) * SLOT_B + 128u * SLOT_A;
        }
        Q S[2][8];
        auto fetch = [&](uint32_t kc) {
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const uint32_t y0 = kc * 64u + 16u * (uint32_t)p;
                    const bool ina = ea <= y0 && y0 - ea < alen_u, inb = T0 <= y0 && y0 - T0 < blen_u;
                    S[h][p] = W::gload16(W::ptr_add(aptr[h], W::splat(ina ? y0 - ea : 0u)), ina ? valid[h] : W::bfalse());
                    S[h][4 + p] = W::gload16(W::ptr_add(bptr[h], W::splat(inb ? y0 - T0 : 0u)), inb ? valid[h] : W::bfalse());
                }
        };
        auto commit_main = [&]() {
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    W::lds_store16(lds, a_slot[h] + 16u * p, S[h][p], active);
                    W::lds_store16(lds, b_slot[h] + 16u * p, S[h][4 + p], active);
                }
        };
        auto commit_look = [&]() {
#pragma unroll
            for (int h = 0; h < 2; h++) W::lds_store16(lds, a_slot[h] + 64u, S[h][0], active);
        };
        const uint32_t kc0 = tp0 / 64u;
        fetch(kc0);
        commit_main();
        fetch(kc0 + 1);
        W::lds_wave_sync();

        for (uint32_t kc = kc0; kc * 64u < iters; kc++) {
            const uint32_t t_lo = kc * 64u;
            const uint32_t t_hi = (t_lo + 64u < iters) ? t_lo + 64u : iters;
            const uint32_t oa = da - t_lo, ob = 0u - t_lo;     // LDS address = slot + o + tp (wraps mod 2^32 like the sum does)
            uint32_t tp = t_lo > tp0 ? t_lo : tp0;
            for (int part = 0; part < 2; part++) {
                const uint32_t p_hi = part == 0 ? (t_lo + 48u < t_hi ? t_lo + 48u : t_hi) : t_hi;
                if (part == 1) { commit_look(); W::lds_wave_sync(); }
                for (; tp < p_hi && tp < T0; tp++) {           // warm-up: rows 1..nlo slide in
                    advance_a(st.AW[0], W::lds_u8(lds, a_slot[0] + (oa + tp)));
                    advance_a(st.AW[1], W::lds_u8(lds, a_slot[1] + (oa + tp)));
                }
                for (; tp < p_hi; tp++) {
                    const U32 a0 = W::lds_u8(lds, a_slot[0] + (oa + tp)), a1 = W::lds_u8(lds, a_slot[1] + (oa + tp));
                    const U32 b0 = W::lds_u8(lds, b_slot[0] + (ob + tp)), b1 = W::lds_u8(lds, b_slot[1] + (ob + tp));
                    advance_a(st.AW[0], a0);
                    advance_a(st.AW[1], a1);
                    column(st, b0, b1, ans, cnt);
                }
            }
            if (t_hi < iters) {
                commit_main();
                fetch(kc + 2);
                W::lds_wave_sync();
            }
        }

        // |delta| + columns - zero-difference steps on the answer diagonal, per half
        const U32 dA = W::splat(diff_u + blen_u) - (cnt & 0xFFFFu), dB = W::splat(diff_u + blen_u) - (cnt >> 16);
        const Bool okA = W::splat(inband ? 1u : 0u) != 0u, okB = okA;
        W::store_u32(P.out, pair[0], W::sel(okA & (dA <= P.k), dA, W::splat(0xFFFFFFFFu)), valid[0]);
        W::store_u32(P.out, pair[1], W::sel(okB & (dB <= P.k), dB, W::splat(0xFFFFFFFFu)), valid[1]);
    }
};

}  // namespace ta
