// lev_bitsqw_body.h -- the small-alphabet bit-parallel band kernel (lev_bitsq_body.h) for alphabets of up to 32 SYMBOLS: IUPAC nucleotide
// codes (16 letters), the amino acids (20 - 25 letters), digits, one case of the Latin letters.
//
// Same recurrence, same 33-diagonal window, same band geometry and result contract as lev_bitsq_body.h (whose `column` this file calls):
// the match vector of a column is the 33-bit window of the column character's row mask Peq[c], read from LDS.  What changes with the
// alphabet's size:
//   * Rings.  Peq[s] is a ring of 64 rows per symbol -- 2 dwords, 8 bytes -- and a pair holds one ring per symbol OF THE ALPHABET (dense,
//     in code order) plus one spare slot: 20 symbols = 172 bytes per pair, 11 KB per wavefront (14 wavefronts per CU; 16 up to 18
//     symbols).  The rows a span of 16 columns reads and the rows committed ahead of it are at most 63 (run(): the piece a commit
//     overwrites died a span ago), so 64 rows do.  A column is one ds_read_b64 of its symbol's ring; the 33 window bits start at bit
//     g = row & 63 and run on from the second dword into the first when g >= 32 -- the two dwords trade places, a wave-uniform choice (two
//     v_cndmask on a scalar condition) -- then one v_alignbit by g & 31.  The rings of eight columns are requested before the first runs.
//   * `a`: symbols -> codes by a shift and a mask, code = (byte >> h) & 31, where the host found h with distinct codes over the alphabet
//     whose OTHER three bits are the same in every symbol (lev_bitsqw_hash; one case of the letters, the digits: h = 0).  Every 16 rows the
//     five code bit-planes of a 16-byte piece are packed by v_dot4_u32_u8 (as in lev_bitsq_body.h), and the mask of a code is ONE
//     three-input operation on the product of planes {0,1} and the planes {2,3,4}.  The codes are walked in order, four (one product of
//     planes 2..4) under ONE wave-uniform test of the membership word, nothing branching inside a group: a code that is no symbol writes
//     its mask where the next symbol writes its own right after.  The promise about the alphabet is verified: the three other bits of
//     every byte are compared, and a row that no symbol's mask covers holds a code outside the alphabet.
//   * `b`: a column's ring is looked up BY THE BYTE ITSELF in a 256-entry table in LDS (one per workgroup, built by the kernel from h, the
//     membership word and the other bits: entry = 2 * rank of the symbol, 0xFF for a byte outside the alphabet): one ds_read_u8 per
//     column, no hash, and the verification for free.
//   * A pair that holds any byte outside the alphabet is not answered here (P.q_bad_list; the launcher's byte-test pass answers it).
// Measured (profiles/r04/ab_alphabet.md): 37 VALU + 19 scalar instructions per column of 64 pairs on the 20 amino acids (17 the recurrence
// and the swap, 4 the lookups of `b`, 16 the conversion of `a`) against 41.9 + ~0 for the byte test of lev_bits_body.h at 16 wavefronts
// per CU: 0.322 against 0.290 ms on cfg2's geometry.  It pays for few symbols in few groups of codes at wide bands (ACGTN: 0.272 against
// 0.290) -- which is where ta_levenshtein_k_batch_alphabet sends batches here; the rest runs the byte-test kernels.
#pragma once
#include "lev_bitsq_body.h"

namespace ta {

// the table entry of a byte value (host: emulation; device: the kernel's prologue)
TA_HD inline uint32_t lev_bitsqw_entry(uint32_t byte, uint32_t shift, uint32_t memb, uint32_t hi) {
    const uint32_t code = (byte >> shift) & 31u;
    const bool ok = ((memb >> code) & 1u) != 0u && (byte & (hi & 0xFFu)) == ((hi >> 8) & 0xFFu);
    return ok ? 2u * (uint32_t)__builtin_popcount(memb & ((1u << code) - 1u)) : 0xFFu;
}

template <class W, bool TRANS>
struct LevBitsQW {
    using U32 = typename W::U32;
    using Bool = typename W::Bool;
    using Ptr = typename W::Ptr;
    using Q = typename W::Q;
    using Base = LevBitsQ<W, TRANS>;
    using State = typename Base::State;
    static constexpr uint32_t SYM_STRIDE = 8;                  // bytes per symbol: the ring's 2 dwords
    static constexpr uint32_t TABLE_BYTES = 256;
    static TA_HD inline uint32_t pair_stride(uint32_t ns) { return 4u * ((2u * (ns + 1u) + 1u) | 1u); }   // + a spare slot; an odd number of dwords
    static TA_HD inline uint32_t lds_per_wave(uint32_t ns) { return 64u * pair_stride(ns); }

    // (the launcher guarantees: fixed-length batch, unit costs, band + transposition rows <= 33, P.q_shift / q_memb / q_hi / q_ns from
    // lev_bitsqw_hash, `tab` filled with lev_bitsqw_entry, P.q_bad_list with room for every pair of the launch)
    static TA_HD inline void run(const LevParams &P, uint32_t wave_index, uint8_t *lds, const uint8_t *tab) {
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
        const uint32_t stride = pair_stride(P.q_ns);
        const U32 ring = lane * stride;
        const uint32_t hs = P.q_shift, memb = P.q_memb;
        const uint32_t himask = (P.q_hi & 0xFFu) * 0x01010101u, hival = ((P.q_hi >> 8) & 0xFFu) * 0x01010101u;
        for (uint32_t q = 0; q < stride; q += 4u) W::lds_write32(lds, ring + q, W::splat(0));       // rows before the string: no matches

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
        auto inside_of = [](uint32_t len_u, uint32_t x) -> uint32_t { return x >= len_u ? 0u : (len_u - x >= 4u ? 4u : len_u - x); };
        // rows 16 piece .. 16 piece + 15 of `a` -> 16 bits of the ring of every symbol of the alphabet
        auto commit_a = [&](uint32_t piece) {
            const Q q = take(SA, piece);
            U32 c[4];
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const U32 dw = W::qword(q, d);
                const uint32_t inside = inside_of(alen_u, 16u * piece + 4u * (uint32_t)d);
                c[d] = W::shr_u(dw, hs) & 0x1F1F1F1Fu;
                if (inside) {                                  // the bits outside the code are the alphabet's
                    const uint32_t m = inside >= 4u ? 0xFFFFFFFFu : ((1u << (8u * inside)) - 1u);
                    bad = bad | (((dw & himask) ^ hival) & m);
                }
            }
            const U32 lo_w = W::splat(0x08040201u), hi_w = W::splat(0x80402010u);
            U32 Pl[5];
#pragma unroll
            for (int p = 0; p < 5; p++) {
                const uint32_t mk = 0x01010101u << p;
                const U32 h = W::dot4(c[1] & mk, hi_w, W::dot4(c[0] & mk, lo_w, W::splat(0)));      // 2^p times the plane's rows 0..7
                const U32 g = W::dot4(c[3] & mk, hi_w, W::dot4(c[2] & mk, lo_w, W::splat(0)));      // rows 8..15
                Pl[p] = W::shr_u(h | (g << 8), (uint32_t)p);
            }
            U32 A[4], B[8];
#pragma unroll
            for (int j = 0; j < 4; j++) A[j] = ((j & 1) ? Pl[0] : ~Pl[0]) & ((j & 2) ? Pl[1] : ~Pl[1]);
#pragma unroll
            for (int j = 0; j < 8; j++) B[j] = ((j & 1) ? Pl[2] : ~Pl[2]) & ((j & 2) ? Pl[3] : ~Pl[3]) & ((j & 4) ? Pl[4] : ~Pl[4]);
            const uint32_t hw = 2u * (piece & 3u);             // the halfword's byte offset in the ring
            const U32 ring_hw = ring + hw;
            U32 seen = W::splat(0);
            // The symbols are walked in code order, four codes (one product of planes 2..4) under ONE wave-uniform test; inside a group
            // nothing branches: a code that is no symbol writes its mask where the next symbol will write its own right after (the
            // address does not move on), and the slot behind the last ring takes what follows the last symbol.
            const uint32_t memb_now = W::opaque_s(memb);
            U32 at = ring_hw;
#pragma unroll
            for (int grp = 0; grp < 8; grp++) {
                if ((memb_now >> (4 * grp)) & 15u) {           // wave-uniform
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t is_sym = (memb_now >> (4 * grp + j)) & 1u;
                        const U32 M = A[j] & B[grp];
                        seen = seen | (M & (0u - is_sym));
                        W::lds_write16(lds, at, M);
                        at = at + SYM_STRIDE * is_sym;
                    }
                }
            }
            // a row of the string that no symbol's mask covers holds a code outside the alphabet
            const uint32_t row0 = 16u * piece;
            const uint32_t nin = row0 >= alen_u ? 0u : (alen_u - row0 >= 16u ? 16u : alen_u - row0);
            if (nin) bad = bad | (~seen & ((1u << nin) - 1u));
            if ((piece & 7u) == 7u) fetch(SA, aptr, alen_u, (int32_t)(piece >> 3) + 1);
        };
        // the 16 columns of piece `piece` of `b` -> the LDS addresses of their symbols' rings, in registers
        auto convert_b = [&](uint32_t piece, U32 (&ba)[16]) {
            const Q q = take(SB, piece);
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const U32 dw = W::qword(q, d);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const U32 e = W::lds_u8(tab, W::byte_of(dw, j));
                    if (16u * piece + 4u * (uint32_t)d + (uint32_t)j < blen_u) bad = bad | (e & 0x80u);
                    ba[4 * d + j] = W::lshl_add(e, 2, ring);
                }
            }
            if ((piece & 7u) == 7u) fetch(SB, bptr, blen_u, (int32_t)(piece >> 3) + 1);
        };
        // column t + 1: the window's top row is a[t - d_hi], bit g = (t - d_hi) & 63 of the ring; the 33 window bits run on from the ring's
        // second dword into its first (rows 64 on) when g >= 32: the two dwords trade places (a wave-uniform choice)
        auto window = [&](uint32_t t, U32 x0, U32 x1) {
            const uint32_t g = (t - dhi) & 63u, s = g & 31u;
            const bool up = g >= 32u;
            const U32 lo = up ? x1 : x0, hi = up ? x0 : x1;
            Base::column(st, W::alignbit_rt(hi, lo, s), W::shr_u(hi, s));
        };
        auto col = [&](uint32_t t, U32 ring_of_symbol) {
            U32 x0, x1;
            W::lds_read64(lds, ring_of_symbol, x0, x1);
            window(t, x0, x1);
        };

        fetch(SA, aptr, alen_u, 0);
        fetch(SB, bptr, blen_u, 0);
        // before the span of columns t + 1 .. t + 16 the rows up to the bottom diagonal's, t + 15 + nlo, must be in the rings: pieces up
        // to p0 + t / 16.  Live rows: t - 32 + nlo .. t + 16 p0 + 15, at most 63 (16 p0 <= nlo + 15); the commit for the span writes over
        // rows t + 16 p0 - 64 .. - 49, below the lowest live one.
        const uint32_t p0 = (15u + nlo) >> 4;
        for (uint32_t p = 0; p <= p0; p++) commit_a(p);
        uint32_t nacc = 0;
        for (uint32_t t = 0; t < blen_u; t += 16u) {
            if (t) commit_a(p0 + (t >> 4));
            U32 ba[16];
            convert_b(t >> 4, ba);
            W::lds_wave_sync();
            if (nacc == 32u) { cnt = W::bcnt(st.acc, cnt); nacc = 0; }
            const uint32_t left = blen_u - t;
            if (left >= 16u) {
                // the rings of eight columns are requested before the first of them runs: one LDS latency per eight columns
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    U32 x0[8], x1[8];
#pragma unroll
                    for (int c = 0; c < 8; c++) W::lds_read64(lds, ba[8 * h + c], x0[c], x1[c]);
#pragma unroll
                    for (int c = 0; c < 8; c++) window(t + (uint32_t)(8 * h + c), x0[c], x1[c]);
                }
                nacc += 16u;
            } else {                                           // the batch's last columns
#pragma unroll
                for (int c = 0; c < 16; c++)
                    if ((uint32_t)c < left) col(t + (uint32_t)c, ba[c]);
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
