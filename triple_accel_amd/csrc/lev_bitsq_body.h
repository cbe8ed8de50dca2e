// lev_bitsq_body.h -- the bit-parallel band kernel (lev_bits_body.h) for SMALL ALPHABETS: the match vector of a column is a
// table lookup instead of a byte test.
//
// In lev_bits_body.h two thirds of a column's instructions build the match vector: 32 window bytes of `a` against the column's
// character, three instructions per four bytes.  When the strings are written in at most FOUR distinct byte values -- DNA / RNA,
// the reference's home domain -- the classic Myers table does it in one read: per pair and symbol s, Peq[s] has bit x set where
// a[x] == s; the match vector of the column with character c is the 33-bit window of Peq[c] that starts at the window's top row.
// Same recurrence, same band, same result contract (d if d <= k else None, src/levenshtein.rs:539-541), bit for bit.
//
//   * One pair per lane, 33-diagonal window as in the stride-8 form: 32 bits in one word plus the band's bottom diagonal
//     (`match | carry`).  Unit-cost families, fixed-length batches (the geometry is one number per launch), unit_k <= 32 (30 with
//     the transposition term).
//   * Peq lives in LDS as a RING of 128 rows per symbol (4 dwords + a wrap copy of the first): the window of column j covers the
//     rows j - d_hi - 1 .. + 32, so a 64-bit read at dword (row & 127) >> 5 and ONE v_alignbit_b32 by row & 31 -- the same
//     shift in every lane -- deliver the 32 window bits; the bottom diagonal's bit is bit (row & 31) of the upper dword.
//     84 bytes per pair (21 dwords: odd, the lanes of a wavefront start in different banks), 5.4 KB per wavefront.
//   * Symbols -> codes: the host finds a shift h with ((s >> h) & 3) distinct over the alphabet (A C G T / a c g t / A C G U:
//     h = 1), so a dword of text becomes four codes by a shift and a mask.  Every byte is checked against the symbol its code
//     stands for (one v_perm_b32 lookup); a pair that holds any other byte is NOT answered here: it goes to P.q_bad_list and the
//     launcher runs the byte-test kernel over that list (the caller's promise about the alphabet is verified, not trusted).
//   * `a`: every 16 columns a 16-byte piece of the string (requested a whole 128-byte line at a time and parked in registers, as
//     in the line form of lev_bits_body.h) becomes 16 bits of each symbol's ring: the two code bit-planes are packed by
//     v_dot4_u32_u8 with the weights 1, 2, 4 .. 128 and combined into the four masks (45 instructions per 16 rows).
//     `b`: the piece of the 16 columns ahead becomes, in registers, the byte offsets of its symbols' rings (code * 20): a
//     column's lookup address is ring + offset byte + window dword.
//   * 15 VALU instructions + one ds_read_b64 per column (20 with the transposition term) + 4.5 per column of conversions,
//     against 42 in the stride-8 form.
#pragma once
#include "lev_bits_body.h"

namespace ta {

template <class W, bool TRANS>
struct LevBitsQ {
    using U32 = typename W::U32;
    using Bool = typename W::Bool;
    using Ptr = typename W::Ptr;
    using Q = typename W::Q;
    static constexpr uint32_t SYM_STRIDE = 20;                 // bytes per symbol: 4 ring dwords + the wrap copy
    static constexpr uint32_t PAIR_STRIDE = 4 * SYM_STRIDE + 4;
    static constexpr uint32_t LDS_PER_WAVE = 64 * PAIR_STRIDE;

    struct State {
        U32 VP, VN;              // vertical +1 / -1 differences at the window's 32 upper rows (the bottom diagonal's step is always +1)
        U32 PMp, D0p, PMb;       // TRANS: the previous column's match vector and D0, and its bottom-diagonal match (bit 0)
        U32 acc;                 // D0 of the window's top diagonal, the last columns' bits from bit 31 down
    };

    // one column: PM = match bits of window bits 0..31, mb bit 0 = the bottom diagonal's characters match (the other bits: anything)
    static TA_HD inline __attribute__((always_inline)) void column(State &st, U32 PM, U32 mb) {
        U32 sum, D0, d0_bot;
        if constexpr (TRANS) {
            const Bool m_bot = (mb & 1u) != 0u;
            Bool carry = W::bfalse();
            W::addc(PM & st.VP, st.VP, carry, sum, carry);
            D0 = ((sum ^ st.VP) | PM) | st.VN;                     // Hyyro 2003 (lev_bits_body.h)
            const U32 pml = PM << 1, pmr = W::template alignbit<1>(st.PMb, st.PMp);
            D0 = D0 | (~st.D0p & pml & pmr);
            d0_bot = W::sel(carry | m_bot, W::splat(1), W::splat(0));      // the bottom diagonal: match | carry
            st.PMb = W::sel(m_bot, W::splat(1), W::splat(0));
        } else {
            // the carry stays a wavefront mask: bit 0 of d0_bot (all the shift below takes) is one v_cndmask away
            const typename W::Mask cm = W::add_carry_mask(PM & st.VP, st.VP, sum);
            D0 = ((sum ^ st.VP) | PM) | st.VN;
            d0_bot = W::sel_mask(cm, W::splat(1), mb);
        }
        st.acc = W::template alignbit<1>(D0, st.acc);
        const U32 HP = st.VN | ~(D0 | st.VP);
        const U32 HN = D0 & st.VP;
        const U32 D0s = W::template alignbit<1>(d0_bot, D0);
        st.VP = HN | ~(D0s | HP);
        st.VN = D0s & HP;
        if (TRANS) { st.PMp = PM; st.D0p = D0; }
    }

    // (the launcher guarantees: fixed-length batch, unit costs, band + transposition rows <= 33, P.q_table / P.q_shift from
    // lev_bitsq_hash, P.q_bad_list with room for every pair of the launch)
    static TA_HD inline void run(const LevParams &P, uint32_t wave_index, uint8_t *lds) {
        const U32 lane = W::lane();
        const Bool active = (lane == lane);
        const U32 slot_idx = lane + wave_index * 64u;
        const Bool valid = slot_idx < P.n;
        const U32 pair = P.subset ? W::load_u32(P.subset, slot_idx, valid, 0u) : slot_idx;
        Ptr aptr, bptr;
        U32 la, lb;
        W::load_str(P.a, pair, valid, aptr, la);
        W::load_str(P.b, pair, valid, bptr, lb);

        // the batch's geometry (lev_plan.h): diagonals d = j - i in [-nlo, d_hi]; window bit i <-> diagonal d_hi - i
        const uint32_t alen_u = (uint32_t)P.a.len, blen_u = (uint32_t)P.b.len;
        const uint32_t diff_u = blen_u >= alen_u ? blen_u - alen_u : alen_u - blen_u;
        if (diff_u > P.u) {                                    // None for every pair (:426-428, :860-862)
            W::store_u32(P.out, pair, W::splat(0xFFFFFFFFu), valid);
            return;
        }
        const uint32_t nlo = ((P.u - diff_u) >> 1) + (blen_u >= alen_u ? 0u : diff_u) + (TRANS ? 1u : 0u);
        const uint32_t dhi = 32u - nlo;
        const uint32_t idx_ans = dhi + alen_u - blen_u;        // row a_len at column b_len, rows below the top diagonal (<= 32)

        State st;
        {   // column 0, D[r][0] = |r|: rows r = 1 - d_hi + i >= 1 step up (+1), rows <= 0 step down (-1)
            const uint32_t below = dhi >= 32u ? 0xFFFFFFFFu : ((1u << dhi) - 1u);
            st.VN = W::splat(below); st.VP = W::splat(~below);
            st.PMp = W::splat(0); st.D0p = W::splat(0xFFFFFFFFu); st.PMb = W::splat(0);
            st.acc = W::splat(0);
        }
        U32 cnt = W::splat(0), bad = W::splat(0);
        const U32 ring = lane * PAIR_STRIDE;
        const uint32_t hs = P.q_shift;
        const U32 table = W::splat(P.q_table);
#pragma unroll
        for (uint32_t q = 0; q < PAIR_STRIDE / 4u; q++) W::lds_write32(lds, ring + 4u * q, W::splat(0));   // rows before the string: no matches

        Q SA[8], SB[8];
        auto fetch = [&](Q (&S)[8], Ptr ptr, uint32_t len_u, int32_t m) {
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const uint32_t off = 128u * (uint32_t)m + 16u * (uint32_t)c;
                // (every lane loads: a lane without a pair points at the batch's first pair -- load_str -- and its bytes go nowhere)
                const Bool ok = off < len_u ? active : W::bfalse();
                S[c] = W::gload16(W::ptr_add(ptr, W::splat(off < len_u ? off : 0u)), ok);
            }
        };
        auto take = [&](const Q (&S)[8], uint32_t piece) -> Q {   // wave-uniform: one of eight parked pieces
            switch (piece & 7u) {
                case 0: return S[0]; case 1: return S[1]; case 2: return S[2]; case 3: return S[3];
                case 4: return S[4]; case 5: return S[5]; case 6: return S[6]; default: return S[7];
            }
        };
        // the four codes of a dword of text; bytes at or beyond `inside` (0..4, wave-uniform) lie past the string's end
        auto codes_of = [&](U32 dw, uint32_t inside) -> U32 {
            const U32 c = W::shr_u(dw, hs) & 0x03030303u;
            if (inside) {
                const uint32_t m = inside >= 4u ? 0xFFFFFFFFu : ((1u << (8u * inside)) - 1u);
                bad = bad | ((W::perm_sel(table, table, c) ^ dw) & m);     // the symbol the code stands for must be the byte itself
            }
            return c;
        };
        auto inside_of = [](uint32_t len_u, uint32_t x) -> uint32_t { return x >= len_u ? 0u : (len_u - x >= 4u ? 4u : len_u - x); };
        // rows 16 piece .. 16 piece + 15 of `a` -> 16 bits of every symbol's ring
        auto commit_a = [&](uint32_t piece) {
            const Q q = take(SA, piece);
            U32 c[4];
#pragma unroll
            for (int d = 0; d < 4; d++) c[d] = codes_of(W::qword(q, d), inside_of(alen_u, 16u * piece + 4u * (uint32_t)d));
            const U32 lo_w = W::splat(0x08040201u), hi_w = W::splat(0x80402010u);
            U32 h0 = W::dot4(c[1] & 0x01010101u, hi_w, W::dot4(c[0] & 0x01010101u, lo_w, W::splat(0)));
            U32 g0 = W::dot4(c[3] & 0x01010101u, hi_w, W::dot4(c[2] & 0x01010101u, lo_w, W::splat(0)));
            U32 h1 = W::dot4(c[1] & 0x02020202u, hi_w, W::dot4(c[0] & 0x02020202u, lo_w, W::splat(0)));     // (twice the mask)
            U32 g1 = W::dot4(c[3] & 0x02020202u, hi_w, W::dot4(c[2] & 0x02020202u, lo_w, W::splat(0)));
            const U32 P0 = h0 | (g0 << 8), P1 = (h1 | (g1 << 8)) >> 1;
            const U32 M[4] = {~(P0 | P1) & 0xFFFFu, P0 & ~P1, P1 & ~P0, P0 & P1};
            const uint32_t hw = 2u * (piece & 7u);             // the halfword's byte offset in the 16-byte ring
#pragma unroll
            for (uint32_t s = 0; s < 4; s++) {
                W::lds_write16(lds, ring + s * SYM_STRIDE + hw, M[s]);
                if (hw < 4u) W::lds_write16(lds, ring + s * SYM_STRIDE + 16u + hw, M[s]);      // the wrap copy of the ring's first dword
            }
            if ((piece & 7u) == 7u) fetch(SA, aptr, alen_u, (int32_t)(piece >> 3) + 1);
        };
        // the 16 columns of piece `piece` of `b` -> the byte offsets of their symbols' rings (code * SYM_STRIDE), in registers
        auto convert_b = [&](uint32_t piece, U32 (&bo)[4]) {
            const Q q = take(SB, piece);
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const U32 c = codes_of(W::qword(q, d), inside_of(blen_u, 16u * piece + 4u * (uint32_t)d));
                bo[d] = W::lshl_add(c, 4, c << 2);             // * 20
            }
            if ((piece & 7u) == 7u) fetch(SB, bptr, blen_u, (int32_t)(piece >> 3) + 1);
        };
        // column t + 1 (t = c0 + c): the window's top row is a[t - d_hi]
        auto col = [&](uint32_t t, U32 bo_d, int byte) {
            const uint32_t g = (t - dhi) & 127u, w4 = (g >> 5) << 2, s = g & 31u;
            U32 lo, hi;
            W::lds_read64(lds, ring + W::byte_of(bo_d, byte) + w4, lo, hi);
            column(st, W::alignbit_rt(hi, lo, s), W::shr_u(hi, s));
        };

        fetch(SA, aptr, alen_u, 0);
        fetch(SB, bptr, blen_u, 0);
        // before the span of columns t + 1 .. t + 16 the rows up to the bottom diagonal's, t + 15 + nlo, must be in the rings
        const uint32_t p0 = (15u + nlo) >> 4;
        for (uint32_t p = 0; p <= p0; p++) commit_a(p);
        uint32_t nacc = 0;
        for (uint32_t t = 0; t < blen_u; t += 16u) {
            if (t) commit_a(p0 + (t >> 4));
            U32 bo[4];
            convert_b(t >> 4, bo);
            W::lds_wave_sync();
            if (nacc == 32u) { cnt = W::bcnt(st.acc, cnt); nacc = 0; }
            const uint32_t left = blen_u - t;
            if (left >= 16u) {
#pragma unroll
                for (int c = 0; c < 16; c++) col(t + (uint32_t)c, bo[c >> 2], c & 3);
                nacc += 16u;
            } else {                                           // the batch's last columns
#pragma unroll
                for (int c = 0; c < 16; c++)
                    if ((uint32_t)c < left) col(t + (uint32_t)c, bo[c >> 2], c & 3);
                nacc += left;
            }
        }
        if (nacc) cnt = W::bcnt(st.acc >> (32u - nacc), cnt);

        // the answer cell: idx_ans rows below the top diagonal of the last column
        const uint32_t mb = idx_ans >= 32u ? 0xFFFFFFFFu : ((1u << idx_ans) - 1u);
        const U32 tail = W::bcnt(st.VP & mb, W::splat(0)) - W::bcnt(st.VN & mb, W::splat(0));
        const U32 d = (W::splat(dhi + blen_u) - cnt) + tail;   // the top diagonal starts at d_hi; + columns - zero steps + way down
        const Bool foreign = bad != 0u;                        // a byte outside the alphabet: the byte-test kernel answers this pair
        W::store_u32(P.out, pair, W::sel(d <= P.k, d, W::splat(0xFFFFFFFFu)), valid & !foreign);
        W::append_u32(P.q_bad_list, P.q_bad_count, pair, valid & foreign);
    }
};

}  // namespace ta
