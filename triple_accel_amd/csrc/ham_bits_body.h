// ham_bits_body.h -- hamming_search for needles of 9..32 bytes and small k: BIT-SLICED mismatch counters.
//
// Contract as ham_swar_body.h (src/hamming.rs:454-554): mismatches(p) <= k for every offset p in [0, h - n].
//
// One lane scans a tile of the haystack.  Needle position j sits on bit 32 - n + j of B bit-plane dwords: plane b holds bit b of the n
// counters "mismatches so far of the alignment that is j bytes old".  A haystack byte c costs
//     s_b = (P_b << 1) | bias_b      every alignment ages by one position; a NEW one enters on bit 32 - n with the counter value
//                                    bias = 2^B - 1 - k  (so that "more than k mismatches" is "the counter overflowed")
//     P_b = s_b ^ carry_b,  carry_{b+1} = s_b & carry_b,  carry_0 = Mis[c]     a ripple add of the one-bit vector Mis[c] (bit 32 - n + j =
//                                    needle[j] != c; one LDS lookup per byte, the table kept once per lane: no bank conflicts)
//     OV = (OV << 1) | carry_B       the overflow bit travels with its alignment and sticks
// and bit 31 of OV is the verdict of the alignment that just completed: 3 B + 1 instructions + the lookup + one v_alignbit that collects
// the verdicts of 32 bytes -- 15 for k <= 14 -- against the SWAR form's 3 per needle dword (27 for 32 bytes).  The exact count of a hit
// (rare) is recounted byte by byte.  Plain per-lane code: the tests run the same functions on the CPU.
#pragma once
#include <stdint.h>

#include "wave.h"

namespace ta {

// planes needed for threshold k (2^B - 1 >= k); 0: k too large for the bit-sliced form
TA_HD inline int ham_bits_planes(uint32_t k) {
    for (int B = 1; B <= 5; B++) if (((1u << B) - 1u) >= k) return B;
    return 0;
}
// Mis[c]
TA_HD inline uint32_t ham_bits_mis(const uint8_t *needle, uint32_t n, uint32_t c) {
    uint32_t m = 0;
    for (uint32_t j = 0; j < n; j++)
        if ((uint32_t)needle[j] != c) m |= 1u << (32u - n + j);
    return m;
}
template <int B> struct HamBitsState { uint32_t P[B], OV; };
template <int B> TA_HD inline void ham_bits_reset(HamBitsState<B> &s, uint32_t n) {
#pragma unroll
    for (int b = 0; b < B; b++) s.P[b] = 0;
    // the alignments "in flight" at a tile's start are nobody's: overflowed (the bits below the needle's stay zero -- a new alignment's
    // overflow bit is shifted in from there)
    s.OV = n >= 32u ? 0xFFFFFFFFu : ~((1u << (32u - n)) - 1u);
}
// bias_b: bit b of 2^B - 1 - k on the entry position
template <int B> TA_HD inline void ham_bits_bias(uint32_t k, uint32_t n, uint32_t (&bias)[B]) {
    const uint32_t v = ((1u << B) - 1u) - k;
#pragma unroll
    for (int b = 0; b < B; b++) bias[b] = ((v >> b) & 1u) << (32u - n);
}
// one haystack byte; returns OV (bit 31: the alignment that ends at this byte has MORE than k mismatches)
template <int B>
TA_HD inline __attribute__((always_inline)) uint32_t ham_bits_step(HamBitsState<B> &s, uint32_t mis, const uint32_t (&bias)[B]) {
    uint32_t carry = mis;
#pragma unroll
    for (int b = 0; b < B; b++) {
        const uint32_t t = (s.P[b] << 1) | bias[b];
        s.P[b] = t ^ carry;
        carry = t & carry;
    }
    s.OV = (s.OV << 1) | carry;
    return s.OV;
}

}  // namespace ta
