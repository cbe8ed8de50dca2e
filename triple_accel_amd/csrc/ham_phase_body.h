// ham_phase_body.h -- hamming_search through bit-sliced mismatch counters over a SUBSET of the needle's positions, Q phases per dword.
//
// Contract as ham_swar_body.h (src/hamming.rs:454-554, scalar text :89-145): mismatches(p) <= k for every offset p in [0, h - n].
//
// ham_bits_body.h ages every alignment by one needle position per haystack byte and ripple-adds one mismatch bit per position:
// 3 B + 3 instructions per byte for a needle of up to 32 bytes, whatever its length.  Two observations make that cheaper and open it
// to longer needles:
//   * an alignment with at most k mismatches over the whole needle has at most k mismatches on ANY subset S of its positions, so a
//     count over S is a filter (a superset of the hits; the rare survivors are recounted over the whole needle, which is what the
//     exact form does with its hits anyway).  A needle of 33.. bytes is filtered on 32 of its positions.
//   * with S = {0, Q, 2Q, ...} the alignment at offset p only ever looks at haystack bytes of its own residue p mod Q: the haystack falls
//     into Q independent PHASES, each with its own counters of L = |S| positions -- and Q * L <= 32 of them share one dword.  One step
//     takes Q haystack bytes (byte r belongs to phase r), ONE ripple add serves all phases: (3 B + 3 + 2 Q) / Q instructions per byte.
//
// Layout: phase r owns bits [r W, r W + W), W = 32 / Q; position i of S (needle byte Q i) sits on bit r W + W - L + i, a new alignment
// enters on bit r W + W - L with the counter value 2^B - 1 - k (so "more than k mismatches on S" is a counter overflow that travels with
// its alignment and sticks), and after the step that read haystack byte p + Q (L - 1) the lane-top bit r W + W - 1 of OV is its verdict.
// The lane tops are cleared after every step (KEEP): nothing of a finished alignment shifts into the next phase's bits.
// Plain per-lane code: the tests run the same functions on the CPU.
#pragma once
#include <stdint.h>

#include "ham_bits_body.h"

namespace ta {

struct HamPhaseGeom {
    uint32_t Q, W, L;          // phases per dword, bits per phase, positions counted per phase
    uint32_t keep;             // every bit but the phase tops
    uint32_t pos;              // the position bits of every phase
    uint32_t entry;            // the entry bit of every phase
    uint32_t span;             // Q (L - 1): haystack bytes between an alignment's first byte and its verdict byte
};
TA_HD inline HamPhaseGeom ham_phase_geom(uint32_t Q, uint32_t L) {
    HamPhaseGeom g;
    g.Q = Q; g.W = 32u / Q; g.L = L;
    g.keep = 0; g.pos = 0; g.entry = 0;
    for (uint32_t r = 0; r < Q; r++) {
        const uint32_t lane = g.W >= 32u ? 0xFFFFFFFFu : (((1u << g.W) - 1u) << (r * g.W));
        const uint32_t top = 1u << (r * g.W + g.W - 1u);
        const uint32_t below = L >= g.W ? 0u : (((1u << (g.W - L)) - 1u) << (r * g.W));
        g.keep |= lane & ~top;
        g.pos |= lane & ~below;
        g.entry |= 1u << (r * g.W + g.W - L);
    }
    g.span = Q * (L - 1u);
    return g;
}
// The plan for (needle length, k): Q phases of L positions and B counter bits; false where this form does not apply or does not pay.
// Selectivity: the subset must still tell a hit from noise -- L >= 2 k and L - k >= 4 (random bytes pass a position with probability 1/256:
// C(L, k) / 256^(L - k) false candidates per offset; a four-letter text passes 2.7 % of its offsets at L = 16, k = 8 and 11 % at L = 8, k = 4,
// each recounted over the whole needle).  (Round 5 asked for L - k >= 6: a 16-byte needle with k = 4 then ran ONE phase of 16 positions at
// 14 instructions per byte, 0.61 ms per GiB -- slower than a 32-byte needle's two phases at 0.49; with two phases of 8 it is 8 per byte.)
// A filter over ALL positions (n <= 32, Q = 1) is exact and needs no margin.
TA_HD inline bool ham_phase_plan(uint32_t n, uint32_t k, uint32_t &Q, uint32_t &L, int &B) {
    B = ham_bits_planes(k);
    if (!B || n == 0 || k >= n) return false;
    for (uint32_t q = 4; q >= 1; q >>= 1) {
        const uint32_t w = 32u / q;
        uint32_t l = (n + q - 1u) / q;                       // Q (L - 1) <= n - 1
        if (l > w) l = w;
        const bool exact = q == 1u && l == n;
        if (!exact && (l < 2u * k || l < k + 4u)) continue;
        Q = q; L = l;
        return true;
    }
    return false;
}
// instructions per haystack byte of this form (what the launcher compares with the SWAR form's 3 per needle dword + 4)
TA_HD inline uint32_t ham_phase_cost_x4(uint32_t Q, int B) { return 4u * (3u * (uint32_t)B + 3u + 2u * Q) / Q; }

// Mis[c] of phase 0: bit W - L + i = needle[Q i] != c
TA_HD inline uint32_t ham_phase_mis(const uint8_t *needle, const HamPhaseGeom &g, uint32_t c) {
    uint32_t m = 0;
    for (uint32_t i = 0; i < g.L; i++)
        if ((uint32_t)needle[g.Q * i] != c) m |= 1u << (g.W - g.L + i);
    return m;
}
template <int B> TA_HD inline void ham_phase_reset(HamBitsState<B> &s, const HamPhaseGeom &g) {
#pragma unroll
    for (int b = 0; b < B; b++) s.P[b] = 0;
    s.OV = g.pos & g.keep;        // the alignments "in flight" at a tile's start are nobody's: overflowed
}
template <int B> TA_HD inline void ham_phase_bias(uint32_t k, const HamPhaseGeom &g, uint32_t (&bias)[B]) {
    const uint32_t v = ((1u << B) - 1u) - k;
#pragma unroll
    for (int b = 0; b < B; b++) bias[b] = ((v >> b) & 1u) ? g.entry : 0u;
}
// one step = Q haystack bytes, mis = their phase-0 table entries combined (phase r's on bits r W ..); returns OV before the tops are
// cleared: bit r W + W - 1 set = the alignment of phase r that ends with this step has MORE than k mismatches on S
// (ONE phase: its top is bit 31 and shifts out by itself -- no KEEP, the step is ham_bits_body.h's)
template <int B, bool MASKED = true>
TA_HD inline __attribute__((always_inline)) uint32_t ham_phase_step(HamBitsState<B> &s, uint32_t mis, const uint32_t (&bias)[B], uint32_t keep) {
    uint32_t carry = mis;
#pragma unroll
    for (int b = 0; b < B; b++) {
        const uint32_t t = (s.P[b] << 1) | bias[b];
        s.P[b] = MASKED ? ((t ^ carry) & keep) : (t ^ carry);
        carry = t & carry;
    }
    const uint32_t ov = (s.OV << 1) | carry;
    s.OV = MASKED ? (ov & keep) : ov;
    return ov;
}

}  // namespace ta
