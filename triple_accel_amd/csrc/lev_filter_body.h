// lev_filter_body.h -- bit-parallel candidate filter in front of the search kernel (lev_search_body.h).
//
// For the unit-cost families, levenshtein_search's cost at an end position -- dp2[len-1] of
// levenshtein_search_naive_with_opts (src/levenshtein.rs:1782-1806) -- is the classic semi-global edit distance
// of the needle against the best-starting substring ending there, which Myers' bit-vector scan (1999; Hyyro 2003
// for the restricted-Damerau term) delivers in ~20 instructions per haystack byte for needles up to 32 bytes.
// The scan does NOT know the match length (the reference's tie rules, quirk Q2, live in the length companion),
// so it only marks 64-column blocks of end positions that contain at least one cost <= k; the exact kernel
// then runs on those blocks alone (each with its own left halo).  On haystacks where matches are rare -- the
// normal case for search -- this removes almost all of the exact DP.
//
// Needle row j sits on bit (32 - n + j); the bits below are wildcard rows (they match every byte, their DP value
// is 0 for ever), which keeps the needle's last row on bit 31 where v_add_co reads it as the carry-out.
// Plain per-lane code, no cross-lane traffic: tests run the same function on the CPU.
#pragma once
#include <stdint.h>

#include "wave.h"

namespace ta {

constexpr uint32_t FILTER_BLOCK = 64;   // end positions per candidate block

// peq[c] for all 256 byte values (device: built in LDS by the block, see lev_search.hip)
TA_HD inline uint32_t lev_filter_peq(const uint8_t *needle, uint32_t n, uint32_t c) {
    uint32_t m = n < 32 ? ((1u << (32 - n)) - 1u) : 0u;            // wildcard rows
    for (uint32_t j = 0; j < n; j++)
        if ((uint32_t)needle[j] == c) m |= 1u << (32 - n + j);
    return m;
}

struct FilterState {
    uint32_t Pv, Mv, D0p, Eqp, score;
};

TA_HD inline void lev_filter_reset(FilterState &s, uint32_t n) {
    s.Pv = n < 32 ? ~((1u << (32 - n)) - 1u) : 0xFFFFFFFFu;         // D[j][start] = j on the needle rows, 0 on the wildcard rows
    s.Mv = 0; s.D0p = 0xFFFFFFFFu; s.Eqp = 0; s.score = n;
}

// one haystack byte; returns the semi-global cost of the needle ending at this byte
template <bool TRANS>
TA_HD inline __attribute__((always_inline)) uint32_t lev_filter_step(FilterState &s, uint32_t Eq) {
    uint32_t D0 = (((Eq & s.Pv) + s.Pv) ^ s.Pv) | Eq | s.Mv;
    if (TRANS) {                                                     // src/levenshtein.rs:1767-1779 (tc = 1)
        D0 |= ((~s.D0p & Eq) << 1) & s.Eqp;
        s.D0p = D0; s.Eqp = Eq;
    }
    const uint32_t Ph = s.Mv | ~(D0 | s.Pv);
    const uint32_t Mh = D0 & s.Pv;
    s.score += (Ph >> 31);
    s.score -= (Mh >> 31);
    const uint32_t Phs = Ph << 1, Mhs = Mh << 1;                     // row 0 is free: D[0][i] = 0 for every i
    s.Pv = Mhs | ~(D0 | Phs);
    s.Mv = Phs & D0;
    return s.score;
}

// The same step without the score: the last row's horizontal -1 bits are shifted into a 32-column register (one v_alignbit_b32) and
// the score is settled once per 32 columns by lev_filter_fold32 -- which reads it off the column itself (round 4: the +1 bits were
// collected too, one more instruction per byte; the score of a column is the sum of its vertical steps, two popcounts per 32 bytes).
template <bool TRANS>
TA_HD inline __attribute__((always_inline)) void lev_filter_step_h(FilterState &s, uint32_t Eq, uint32_t &MH) {
    uint32_t D0 = (((Eq & s.Pv) + s.Pv) ^ s.Pv) | Eq | s.Mv;
    if (TRANS) {
        D0 |= ((~s.D0p & Eq) << 1) & s.Eqp;
        s.D0p = D0; s.Eqp = Eq;
    }
    const uint32_t Ph = s.Mv | ~(D0 | s.Pv);
    const uint32_t Mh = D0 & s.Pv;
#if defined(__HIP_DEVICE_COMPILE__)
    MH = __builtin_amdgcn_alignbit(MH, Mh, 31);                      // (MH << 1) | (Mh >> 31)
#else
    MH = (MH << 1) | (Mh >> 31);
#endif
    const uint32_t Phs = Ph << 1, Mhs = Mh << 1;
    s.Pv = Mhs | ~(D0 | Phs);
    s.Mv = Phs & D0;
}

// After exactly 32 lev_filter_step_h: could one of those 32 columns have cost <= k?  The cost moves by one per column, so it
// never went below (score before) - (number of -1 steps): a LOWER BOUND -- the answer may be yes for a block without a hit
// (the exact kernel then finds nothing there), never no for a block with one.  On text where matches are rare the -1 steps
// are rare too and the bound is as good as the exact minimum; it saves 1.3 of the scan's 16 instructions per byte.
// The score after those columns = the last row's value = the sum of the column's vertical steps over the needle's rows (row 0 is 0 and the
// wildcard rows below the needle carry no steps): popcount(Pv) - popcount(Mv).
TA_HD inline __attribute__((always_inline)) bool lev_filter_fold32(FilterState &s, uint32_t MH, uint32_t k) {
    const uint32_t down = (uint32_t)__builtin_popcount(MH);
    const bool any = s.score <= k + down;
    s.score = (uint32_t)__builtin_popcount(s.Pv) - (uint32_t)__builtin_popcount(s.Mv);
    return any;
}

// Scan columns [col_begin, col_end) of `hay`; for every FILTER_BLOCK-aligned block of columns >= emit_begin that
// holds a column of cost <= k, call mark(block_index) once.  emit_begin must be a multiple of FILTER_BLOCK.
template <bool TRANS, class Peq, class Mark>
TA_HD inline void lev_filter_tile(const uint8_t *hay, Peq peq, uint32_t n, uint32_t k, uint64_t col_begin,
                                  uint64_t emit_begin, uint64_t col_end, Mark mark) {
    FilterState s;
    lev_filter_reset(s, n);
    bool any = false;
    for (uint64_t i = col_begin; i < col_end; i++) {
        const uint32_t cost = lev_filter_step<TRANS>(s, peq(hay[i]));
        if (i >= emit_begin) {
            any |= cost <= k;
            if ((i & (FILTER_BLOCK - 1)) == FILTER_BLOCK - 1 || i + 1 == col_end) {
                if (any) mark(i / FILTER_BLOCK);
                any = false;
            }
        }
    }
}

// as lev_filter_tile with the score settled per 32 columns (lev_filter_step_h / lev_filter_fold32) on whole blocks: marks a
// SUPERSET of lev_filter_tile's blocks.  emit_begin must be a multiple of FILTER_BLOCK.
template <bool TRANS, class Peq, class Mark>
TA_HD inline void lev_filter_tile_lb(const uint8_t *hay, Peq peq, uint32_t n, uint32_t k, uint64_t col_begin,
                                     uint64_t emit_begin, uint64_t col_end, Mark mark) {
    FilterState s;
    lev_filter_reset(s, n);
    uint64_t i = col_begin;
    for (; i < emit_begin && i < col_end; i++) lev_filter_step<TRANS>(s, peq(hay[i]));     // left context: exact score
    while (i + FILTER_BLOCK <= col_end) {
        bool any = false;
        for (int half = 0; half < 2; half++) {
            uint32_t MH = 0;
            for (int b = 0; b < 32; b++) lev_filter_step_h<TRANS>(s, peq(hay[i + 32 * half + b]), MH);
            any |= lev_filter_fold32(s, MH, k);
        }
        if (any) mark(i / FILTER_BLOCK);
        i += FILTER_BLOCK;
    }
    if (i < col_end) {                                               // the partial last block: exact
        bool any = false;
        const uint64_t blk = i / FILTER_BLOCK;
        for (; i < col_end; i++) any |= lev_filter_step<TRANS>(s, peq(hay[i])) <= k;
        if (any) mark(blk);
    }
}

// ---- needles of 33..256 bytes: the same scan on NWF-dword bit-vectors (needle on the top bits of the vector)

// word w (0 = lowest) of the match vector of byte value c
TA_HD inline uint32_t lev_filter_peq_word(const uint8_t *needle, uint32_t n, uint32_t nwf, uint32_t c, uint32_t w) {
    const uint32_t pad = 32u * nwf - n;                             // wildcard rows below the needle
    uint32_t m = 0;
    for (uint32_t b = 0; b < 32; b++) {
        const uint32_t bit = 32u * w + b;
        if (bit < pad || (uint32_t)needle[bit - pad] == c) m |= 1u << b;
    }
    return m;
}

template <int NWF>
struct FilterStateN {
    uint32_t Pv[NWF], Mv[NWF], D0p[NWF], Eqp[NWF], score;
};

template <int NWF>
TA_HD inline void lev_filter_reset_n(FilterStateN<NWF> &s, uint32_t n) {
    const uint32_t pad = 32u * NWF - n;
    for (int w = 0; w < NWF; w++) {
        const uint32_t lo = 32u * (uint32_t)w;
        const uint32_t wild = pad >= lo + 32u ? 0xFFFFFFFFu : (pad <= lo ? 0u : ((1u << (pad - lo)) - 1u));
        s.Pv[w] = ~wild; s.Mv[w] = 0; s.D0p[w] = 0xFFFFFFFFu; s.Eqp[w] = 0;
    }
    s.score = n;
}

template <int NWF, bool TRANS>
TA_HD inline __attribute__((always_inline)) uint32_t lev_filter_step_n(FilterStateN<NWF> &s, const uint32_t (&Eq)[NWF]) {
    uint32_t D0[NWF], Ph[NWF], Mh[NWF];
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < NWF; w++) {
        const uint64_t t = (uint64_t)(Eq[w] & s.Pv[w]) + s.Pv[w] + carry;
        carry = (uint32_t)(t >> 32);
        D0[w] = (((uint32_t)t) ^ s.Pv[w]) | Eq[w] | s.Mv[w];
    }
    if (TRANS) {
#pragma unroll
        for (int w = 0; w < NWF; w++) {
            const uint32_t x = ~s.D0p[w] & Eq[w], xl = w ? (~s.D0p[w - 1] & Eq[w - 1]) : 0u;
            D0[w] |= ((x << 1) | (xl >> 31)) & s.Eqp[w];
        }
#pragma unroll
        for (int w = 0; w < NWF; w++) { s.D0p[w] = D0[w]; s.Eqp[w] = Eq[w]; }
    }
#pragma unroll
    for (int w = 0; w < NWF; w++) {
        Ph[w] = s.Mv[w] | ~(D0[w] | s.Pv[w]);
        Mh[w] = D0[w] & s.Pv[w];
    }
    s.score += (Ph[NWF - 1] >> 31);
    s.score -= (Mh[NWF - 1] >> 31);
#pragma unroll
    for (int w = NWF - 1; w >= 0; w--) {
        const uint32_t Phs = (Ph[w] << 1) | (w ? (Ph[w - 1] >> 31) : 0u);   // row 0 is free: nothing shifts in
        const uint32_t Mhs = (Mh[w] << 1) | (w ? (Mh[w - 1] >> 31) : 0u);
        s.Pv[w] = Mhs | ~(D0[w] | Phs);
        s.Mv[w] = Phs & D0[w];
    }
    return s.score;
}

// as lev_filter_tile; peq(c, Eq) fills the NWF words of byte value c
template <int NWF, bool TRANS, class Peq, class Mark>
TA_HD inline void lev_filter_tile_n(const uint8_t *hay, Peq peq, uint32_t n, uint32_t k, uint64_t col_begin,
                                    uint64_t emit_begin, uint64_t col_end, Mark mark) {
    FilterStateN<NWF> s;
    lev_filter_reset_n<NWF>(s, n);
    bool any = false;
    for (uint64_t i = col_begin; i < col_end; i++) {
        uint32_t Eq[NWF];
        peq(hay[i], Eq);
        const uint32_t cost = lev_filter_step_n<NWF, TRANS>(s, Eq);
        if (i >= emit_begin) {
            any |= cost <= k;
            if ((i & (FILTER_BLOCK - 1)) == FILTER_BLOCK - 1 || i + 1 == col_end) {
                if (any) mark(i / FILTER_BLOCK);
                any = false;
            }
        }
    }
}

}  // namespace ta
