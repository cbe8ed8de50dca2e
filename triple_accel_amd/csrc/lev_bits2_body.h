// lev_bits2_body.h -- the bit-parallel band kernel (lev_bits_body.h) for NARROW bands: TWO pairs per lane, stride-8 window.
//
// When the band of a fixed-length batch is at most 15 diagonals wide (unit_k <= 14, or 12 with the transposition term: what
// small thresholds like cfg4's k = 8 give), a pair's column state -- VP, VN and the match vector, one bit per diagonal -- fills
// half a VGPR.  The other half carries a SECOND pair: lane l of wavefront w owns the pairs 128 w + l (bits 0..14) and
// 128 w + 64 + l (bits 16..30).  Hyyro's recurrence -- half of a column's instructions for such windows -- is bitwise and serves
// both pairs at once, and so does the byte test, because the window bytes are laid out the stride-8 way (lev_bits_body.h, S8):
//
//   * register m (0..7) holds, for BOTH pairs, the bytes of `a` (^ 0x0C) under window bits m and m + 8:
//         byte 0 = pair A row m, byte 1 = pair A row m + 8, byte 2 = pair B row m, byte 3 = pair B row m + 8;
//     the column characters are splatted pair-wise, Bs = (bA, bA, bB, bB) -- one v_perm_b32 of the two pairs' `b` dwords --
//     so v_xor + v_perm (W::ne12) give the four mismatch flags of a register and a tree of seven v_bfi_b32 over the eight registers
//     drops them onto their window bits (m, m + 8 | 16 + m, 24 + m): 23 instructions per column for two pairs, no Horner shifts;
//   * the window moves one row down per column by RENAMING the registers (unrolled 8 columns): the register whose rows were
//     (0, 8) becomes the one of rows (7, 15) by one v_perm_b32 that drops byte 0 / byte 2 and takes the two pairs' entering bytes
//     (pre-interleaved, one v_perm_b32 per two columns).  Row 15 (bit 15 / bit 31) is a spare: a byte enters there one column
//     before it is needed (the stream of `a` runs that one iteration ahead) and every vector is masked to 15 + 15 bits;
//   * the halves do not talk to each other: the one addition, (PM & VP) + VP, cannot carry out of bit 14 into bit 16 because bit
//     15 of both terms is zero; right shifts move pair B's bit 0 into bit 15, which is masked; PM is masked before PM << 1.
//
// Zero steps are counted on the window's TOP diagonal (bit 0 / bit 16, two 16-bit counters in one register) and the answer
// cell is reached over the last column's vertical differences, as in lev_bits_body.h.  43.5 VALU instructions per column of
// 128 pairs with the transposition term (39.5 without) against 32 per column of 64 pairs in the one-pair sliding form.
//
// Strings: fixed-length batches only (the band geometry is one number per launch, every event is wave-uniform).  Each lane
// requests HALF a 128-byte line (four 16-byte pieces) of each of its four strings at a time and parks it in registers (64 VGPRs);
// LDS holds per pair a ring of 2 pieces of `a` and 1 of `b` (36 + 20 bytes with the wrap copies): 7 KB per wavefront; the
// 119 - 123 VGPRs allow 16 wavefronts per CU.  A piece moves registers -> LDS every 16 columns; the burst for the next half line follows the commit
// of a half line's last piece.  Same result contract as lev_bits_body.h (d if d <= k else None, src/levenshtein.rs:539-541).
#pragma once
#include "lev_bits_body.h"

namespace ta {

template <class W, bool TRANS, bool EARLY = false>
struct LevBits2 {
    static constexpr int WB = 15;                              // window bits per pair (bit 15 / 31: the spare row)
    static constexpr uint32_t WM = (1u << WB) - 1u, WM2 = WM | (WM << 16);
    using U32 = typename W::U32;
    using Bool = typename W::Bool;
    using Ptr = typename W::Ptr;
    using Q = typename W::Q;
    // ring pieces of `a` / `b` per pair: a span of 16 iterations reads 16 + 15 bytes of `a` from where its first byte sits in a
    // piece (two pieces) and one piece of `b` (its stream starts on a piece boundary: T0 is a multiple of 64); the piece a span
    // needs last is committed at the span's start into the slot of the piece the span before finished with
    static constexpr int32_t RA = 2, RB = 1;
    static constexpr uint32_t SLOT_A = 16u * RA + 4u, SLOT_B = 16u * RB + 4u;   // + a wrap copy of the ring's first dword
    static constexpr uint32_t LDS_PER_WAVE = 128u * (SLOT_A + SLOT_B);
#ifndef TA_BITS2_BURST
#define TA_BITS2_BURST 4
#endif
    static constexpr int BURST = TA_BITS2_BURST, LOGB = BURST == 4 ? 2 : (BURST == 2 ? 1 : 3);   // pieces per request: half a line (A/B builds: 2, 8)

    struct State {
        U32 VP, VN, PMp, D0p;        // both pairs: bits 0..14 pair A, 16..30 pair B
        U32 AW[8];                   // the stride-8 byte window of both pairs (see the header)
        U32 cnt;                     // zero steps on the top diagonals: pair A in bits 0..15, pair B in 16..31
    };

    // iteration with C = tp % 8: (COLUMN) one column of both pairs, then the window moves one row down.
    // bA / bB: the pairs' `b` dwords whose byte C & 3 is this column's character; X: the entering bytes of `a` (^ 0x0C) of
    // two iterations, [A(even), B(even), A(odd), B(odd)].
    template <int C, bool COLUMN>
    static TA_HD inline __attribute__((always_inline)) void step(State &st, U32 bA, U32 bB, U32 X) {
        if (COLUMN) {
            constexpr uint32_t c = (uint32_t)(C & 3);
            const U32 Bs = W::template perm<c * 0x0101u + (4u + c) * 0x01010000u>(bB, bA);          // (bA, bA, bB, bB)
            // bit m of every byte from the byte masks of register (C + m) & 7 (register C holds bits 0, 8, 16, 24): a tree of seven
            // v_bfi -- each takes the bits of its mask from one side and ALL the others from the other side, whose stray bits the
            // next level drops (a chain of v_and_or needs eight)
            U32 M[8];
#pragma unroll
            for (int m = 0; m < 8; m++) M[m] = W::ne12(st.AW[(C + m) & 7] ^ Bs);
            const U32 q01 = W::bfi_k(0x01010101u, M[0], M[1]), q23 = W::bfi_k(0x04040404u, M[2], M[3]);
            const U32 q45 = W::bfi_k(0x10101010u, M[4], M[5]), q67 = W::bfi_k(0x40404040u, M[6], M[7]);
            const U32 t = W::bfi_k(0x0F0F0F0Fu, W::bfi_k(0x03030303u, q01, q23), W::bfi_k(0x30303030u, q45, q67));
            const U32 PM = ~t & WM2;
            const U32 s = (PM & st.VP) + st.VP;                // st.VP is inside the windows: no carry leaves bit 14
            U32 D0 = ((s ^ st.VP) | PM) | st.VN;
            if (TRANS) {
                // a[i-1] == b[j-2] && a[i-2] == b[j-1] and the diagonal step before was +1 (src/levenshtein.rs:517-525)
                const U32 pml = PM << 1, pmr = st.PMp >> 1;    // (pair B's bit 0 lands on bit 15: cleared by the & WM2 below)
                D0 = D0 | (~st.D0p & pml & pmr);
            }
            D0 = D0 & WM2;
            st.cnt = st.cnt + (D0 & 0x00010001u);
            const U32 HP = st.VN | ~(D0 | st.VP);
            const U32 HN = D0 & st.VP;
            const U32 D0s = D0 >> 1;                           // next window's rows
            st.VP = (HN | ~(D0s | HP)) & WM2;
            st.VN = D0s & HP;
            if (TRANS) { st.PMp = PM; st.D0p = D0; }
        }
        // rows (0, 8) -> rows (7, 15): [AW.1, A's entering byte, AW.3, B's entering byte]
        constexpr uint32_t e = (uint32_t)(C & 1) * 2u;
        st.AW[C] = W::template perm<0x00030001u | ((4u + e) << 8) | ((5u + e) << 24)>(X, st.AW[C]);
    }

    // (the launcher guarantees: fixed-length batch, band + transposition rows <= WB, P.Tw a multiple of 64, >= band + 2)
    static TA_HD inline void run(const LevParams &P, uint32_t wave_index, uint8_t *lds) {
        const U32 lane = W::lane();
        const Bool active = (lane == lane);
        Bool valid[2];
        U32 pair[2];
        Ptr aptr[2], bptr[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const U32 slot_idx = lane + wave_index * 128u + 64u * (uint32_t)h;
            valid[h] = slot_idx < P.n;
            pair[h] = P.subset ? W::load_u32(P.subset, slot_idx, valid[h], 0u) : slot_idx;
            U32 la, lb;
            W::load_str(P.a, pair[h], valid[h], aptr[h], la);
            W::load_str(P.b, pair[h], valid[h], bptr[h], lb);
        }
        // the batch's geometry (lev_plan.h): diagonals d = j - i in [-nlo, d_hi]; window bit i <-> diagonal d_hi - i
        const uint32_t alen_u = (uint32_t)P.a.len, blen_u = (uint32_t)P.b.len;
        const uint32_t diff_u = blen_u >= alen_u ? blen_u - alen_u : alen_u - blen_u;
        const bool inband = diff_u <= P.u;                     // else None for every pair (:426-428, :860-862)
        const uint32_t nlo = inband ? ((P.u - diff_u) >> 1) + (blen_u >= alen_u ? 0u : diff_u) + (TRANS ? 1u : 0u) : 0u;
        const uint32_t dhi = (uint32_t)WB - 1u - nlo;
        const uint32_t idx_ans = inband ? dhi + alen_u - blen_u : 0u;   // row a_len at column b_len, rows below the top diagonal

        State st;
        {   // column 0, D[r][0] = |r|: rows r = 1 - d_hi + i >= 1 step up (+1), rows <= 0 step down (-1)
            const uint32_t below = dhi ? ((1u << dhi) - 1u) & WM : 0u;
            st.VN = W::splat(below | (below << 16));
            st.VP = W::splat((~below & WM) | ((~below & WM) << 16));
            st.PMp = W::splat(0);
            st.D0p = W::splat(WM2);
            st.cnt = W::splat(0);
        }
#pragma unroll
        for (int m = 0; m < 8; m++) st.AW[m] = W::splat(0);

        // iteration tp inserts a[tp - ca_s] into the windows (it is row 15 -- the spare -- during iteration tp + 1 and row 14 from
        // tp + 2 on) and, from tp = T0 on, runs column tp - T0 + 1 with b[tp - T0]
        const uint32_t T0 = P.Tw;
        const int32_t ca_s = (int32_t)T0 - (int32_t)nlo - 2;
        const uint32_t iters = T0 + blen_u;

        U32 a_slot[2], b_slot[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            a_slot[h] = (lane + 64u * (uint32_t)h) * SLOT_A;
            b_slot[h] = (lane + 64u * (uint32_t)h) * SLOT_B + 128u * SLOT_A;
        }
        Q SA[2][BURST], SB[2][BURST];
        auto fetch = [&](Q (&S)[2][BURST], const Ptr (&ptr)[2], uint32_t len_u, int32_t m) {
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int c = 0; c < BURST; c++) {
                    const int32_t off = 16 * BURST * m + 16 * c;
                    // (every lane loads: a lane without a pair points at the batch's first pair -- load_str -- and its bytes go nowhere)
                    const Bool ok = (off >= 0 && (uint32_t)off < len_u) ? active : W::bfalse();
                    S[h][c] = W::gload16(W::ptr_add(ptr[h], W::splat(off >= 0 ? (uint32_t)off : 0u)), ok);
                }
        };
        auto put = [&](const Q (&S)[2][BURST], int32_t piece, const U32 (&slot)[2], uint32_t at, uint32_t wrap_copy_at, uint32_t x) {
            switch (piece & (BURST - 1)) {                     // wave-uniform: one of four stores per pair
#define TA_PUT(c_) case c_: _Pragma("unroll") for (int h = 0; h < 2; h++) { constexpr int c = c_ < BURST ? c_ : 0; const Q q = x ? W::qxor(S[h][c], x) : S[h][c]; W::lds_store16(lds, slot[h] + at, q, active); \
                              if (wrap_copy_at) W::lds_write32(lds, slot[h] + at + wrap_copy_at, W::qword(q, 0)); } break;
                TA_PUT(0) TA_PUT(1) TA_PUT(2) TA_PUT(3) TA_PUT(4) TA_PUT(5) TA_PUT(6) TA_PUT(7)
#undef TA_PUT
            }
        };
        auto fmod = [](int32_t x, int32_t m) -> uint32_t { const int32_t r = x % m; return (uint32_t)(r < 0 ? r + m : r); };
        auto commit_a = [&](int32_t piece) {
            const uint32_t slot = fmod(piece, RA);
            put(SA, piece, a_slot, 16u * slot, slot == 0u ? 16u * RA : 0u, 0x0C0C0C0Cu);
            if ((piece & (BURST - 1)) == BURST - 1) fetch(SA, aptr, alen_u, (piece >> LOGB) + 1);
        };
        auto commit_b = [&](int32_t piece) {
            const uint32_t slot = fmod(piece, RB);
            put(SB, piece, b_slot, 16u * slot, slot == 0u ? 16u * RB : 0u, 0u);
            if ((piece & (BURST - 1)) == BURST - 1) fetch(SB, bptr, blen_u, (piece >> LOGB) + 1);
        };

        // one block of 8 iterations starting at tp (a multiple of 8); `left` of its columns run (8 but for the batch's last block)
        auto block = [&](uint32_t tp, uint32_t left, auto column_tag) {
            constexpr bool COLUMN = decltype(column_tag)::value;
            const uint32_t oa = fmod((int32_t)tp - ca_s, 16 * RA), oa4 = fmod((int32_t)tp + 4 - ca_s, 16 * RA);
            const U32 aA0 = W::lds_read32u(lds, a_slot[0] + oa), aB0 = W::lds_read32u(lds, a_slot[1] + oa);
            const U32 aA1 = W::lds_read32u(lds, a_slot[0] + oa4), aB1 = W::lds_read32u(lds, a_slot[1] + oa4);
            // entering bytes of two iterations per register: [A(even), B(even), A(odd), B(odd)]
            const U32 X01 = W::template perm<0x05010400u>(aB0, aA0), X23 = W::template perm<0x07030602u>(aB0, aA0);
            const U32 X45 = W::template perm<0x05010400u>(aB1, aA1), X67 = W::template perm<0x07030602u>(aB1, aA1);
            U32 bA0 = W::splat(0), bB0 = bA0, bA1 = bA0, bB1 = bA0;
            if (COLUMN) {
                const uint32_t ob = fmod((int32_t)tp - (int32_t)T0, 16 * RB), ob4 = fmod((int32_t)tp + 4 - (int32_t)T0, 16 * RB);
                bA0 = W::lds_read32u(lds, b_slot[0] + ob); bB0 = W::lds_read32u(lds, b_slot[1] + ob);
                bA1 = W::lds_read32u(lds, b_slot[0] + ob4); bB1 = W::lds_read32u(lds, b_slot[1] + ob4);
            }
            if (left >= 8u) {
                step<0, COLUMN>(st, bA0, bB0, X01); step<1, COLUMN>(st, bA0, bB0, X01);
                step<2, COLUMN>(st, bA0, bB0, X23); step<3, COLUMN>(st, bA0, bB0, X23);
                step<4, COLUMN>(st, bA1, bB1, X45); step<5, COLUMN>(st, bA1, bB1, X45);
                step<6, COLUMN>(st, bA1, bB1, X67); step<7, COLUMN>(st, bA1, bB1, X67);
            } else {                                           // the batch's last columns
                if (left > 0u) step<0, COLUMN>(st, bA0, bB0, X01);
                if (left > 1u) step<1, COLUMN>(st, bA0, bB0, X01);
                if (left > 2u) step<2, COLUMN>(st, bA0, bB0, X23);
                if (left > 3u) step<3, COLUMN>(st, bA0, bB0, X23);
                if (left > 4u) step<4, COLUMN>(st, bA1, bB1, X45);
                if (left > 5u) step<5, COLUMN>(st, bA1, bB1, X45);
                if (left > 6u) step<6, COLUMN>(st, bA1, bB1, X67);
            }
        };

        const bool early = EARLY && (P.tune & 2u) != 0u && P.k < 0x7FFFFFFFu;               // stop once no pair of the wavefront can end at or below k (lev_bits_body.h)
        bool dead = false;
        if (inband) {
            uint32_t tp = (uint32_t)ca_s & ~7u;                // whole blocks: the extra leading iterations slide bytes in that leave again
            const uint32_t tb0 = tp & ~15u;
            // pieces the first span reads: a string offset x lives in piece x >> 4 (arithmetic shift: offsets before the string are
            // pieces < 0, delivered as zeros)
            int32_t qa = ((int32_t)tb0 - ca_s) >> 4, qb = ((int32_t)tb0 - (int32_t)T0) >> 4;
            fetch(SA, aptr, alen_u, qa >> LOGB);
            fetch(SB, bptr, blen_u, qb >> LOGB);
            for (int32_t x = qa; x < qa + RA - 1; x++) commit_a(x);
            for (int32_t x = qb; x < qb + RB - 1; x++) commit_b(x);
            for (uint32_t tb = tb0; tb < iters; tb += 16u, qa++, qb++) {
                if (EARLY && early && tb > T0 && (tb & 16u) == 0u) {    // every 32 columns (lev_bits_body.h, EARLY OUT): tb - T0 columns are done
                    const U32 topA = W::splat(dhi + (tb - T0)) - (st.cnt & 0xFFFFu), topB = W::splat(dhi + (tb - T0)) - (st.cnt >> 16);
                    const Bool liveA = valid[0] & (topA <= W::bcnt(st.VN & WM, W::splat(P.k)));
                    const Bool liveB = valid[1] & (topB <= W::bcnt(st.VN & (WM << 16), W::splat(P.k)));
                    if (!W::any(liveA | liveB)) { dead = true; break; }
                }
                commit_a(qa + RA - 1);                         // into the slot of piece qa - 1, which the last span finished
                commit_b(qb + RB - 1);
                W::lds_wave_sync();
                const uint32_t hi = tb + 16u < iters ? tb + 16u : iters;
                for (; tp < hi && tp < T0; tp += 8u) block(tp, 8u, std::false_type());      // warm-up: rows slide in (T0 is a multiple of 64)
                for (; tp + 8u <= hi; tp += 8u) block(tp, 8u, std::true_type());
                if (tp < hi) { block(tp, hi - tp, std::true_type()); tp = hi; }
            }
        }

        // the answer cells: idx_ans rows below the top diagonals of the last column
        const uint32_t mb = idx_ans ? ((1u << idx_ans) - 1u) : 0u;
        const U32 zero = W::splat(0);
        const U32 downA = W::bcnt(st.VP & mb, zero) - W::bcnt(st.VN & mb, zero);
        const U32 downB = W::bcnt(st.VP & (mb << 16), zero) - W::bcnt(st.VN & (mb << 16), zero);
        const U32 base = W::splat(dhi + blen_u);               // the top diagonal starts at d_hi; + columns - zero steps + way down
        const U32 dA = base - (st.cnt & 0xFFFFu) + downA, dB = base - (st.cnt >> 16) + downB;
        const Bool ok = W::splat(inband && !dead ? 1u : 0u) != 0u;
        W::store_u32(P.out, pair[0], W::sel(ok & (dA <= P.k), dA, W::splat(0xFFFFFFFFu)), valid[0]);
        W::store_u32(P.out, pair[1], W::sel(ok & (dB <= P.k), dB, W::splat(0xFFFFFFFFu)), valid[1]);
    }
};

}  // namespace ta
