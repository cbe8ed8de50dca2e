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

}  // namespace ta
