// ham_swar_body.h -- hamming_search for needles of up to 64 bytes, SWAR form: one lane owns 16 consecutive haystack offsets.
//
// Contract (src/hamming.rs:454-554, scalar text :89-145): for every offset p in [0, h - n] the number of mismatching bytes between
// needle and haystack[p .. p+n); reported when <= k.
//
// The lane holds the 4 + NW aligned dwords that cover its 16 windows (NW = needle dwords): three 16-byte loads, all in flight before the
// first compare.  A window starting at byte 4 i + r is one v_alignbyte of dwords i, i + 1 -- computed ONCE per (i, r) and used by every
// (offset group g, needle dword j) with g + j = i: 3 (3 + NW) shifts instead of 12 NW -- and a compare is v_xor, v_perm (wave.h ne12:
// the needle dword carries ^ 0x0C, so a byte is 12 exactly where window and needle agree; 0x00 there, 0xFF elsewhere), v_bcnt with its
// accumulate operand: a mismatching byte adds 8.  3 NW + 0.75 (3 + NW) / 4 ... instructions per offset: 6.9 for an 8-byte needle, 26 for 32.
// Plain per-lane code: the tests run the same function on the CPU.
#pragma once
#include <stdint.h>

#include "wave.h"

namespace ta {

TA_HD inline uint32_t ham_ne12(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(0xFFFFFFFFu, 0xFFFFFFFFu, x);
#else
    uint32_t f = 0;
    for (int b = 0; b < 4; b++) if (((x >> (8 * b)) & 0xffu) != 12u) f |= 0xffu << (8 * b);
    return f;
#endif
}
template <int R> TA_HD inline uint32_t ham_window(uint32_t hi, uint32_t lo) {
    if (R == 0) return lo;
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, R);
#else
    return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (8 * R));
#endif
}
TA_HD inline uint32_t ham_popc(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_popcount(x);
#else
    uint32_t c = 0; for (; x; x &= x - 1) c++; return c;
#endif
}

// w: the aligned dwords of haystack bytes [B, B + 16 + 4 NW); nd12[j] = needle dword j ^ 0x0C0C0C0C, the bytes past the needle's end in the
// last dword forced to 12 by tail_mask / tail_pad (they compare equal).  cnt[4 g + r] = EIGHT TIMES the mismatches of the window at byte B + 4 g + r.
template <int NW>
TA_HD inline void ham_swar_lane(const uint32_t (&w)[4 + NW], const uint32_t (&nd12)[NW], uint32_t tail_mask, uint32_t tail_pad,
                                uint32_t (&cnt)[16]) {
    uint32_t win[3 + NW][4];
#pragma unroll
    for (int i = 0; i < 3 + NW; i++) {
        win[i][0] = w[i];
        win[i][1] = ham_window<1>(w[i + 1], w[i]);
        win[i][2] = ham_window<2>(w[i + 1], w[i]);
        win[i][3] = ham_window<3>(w[i + 1], w[i]);
    }
#pragma unroll
    for (int q = 0; q < 16; q++) cnt[q] = 0;
#pragma unroll
    for (int j = 0; j < NW; j++) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                uint32_t x = win[g + j][r] ^ nd12[j];
                if (j == NW - 1) x = (x & tail_mask) | tail_pad;
                cnt[4 * g + r] += ham_popc(ham_ne12(x));
            }
        }
    }
    // (cnt[q] = 8 x the mismatches: the caller compares against 8 k + 7 and shifts only what it reports)
}

// the needle as the lane wants it
template <int NW>
TA_HD inline void ham_swar_needle(const uint8_t *needle, uint32_t n, uint32_t (&nd12)[NW], uint32_t &tail_mask, uint32_t &tail_pad) {
#pragma unroll
    for (int j = 0; j < NW; j++) {
        uint32_t v = 0;
        for (int b = 0; b < 4; b++) {
            const uint32_t idx = 4u * (uint32_t)j + (uint32_t)b;
            v |= (uint32_t)(idx < n ? needle[idx] : 0u) << (8 * b);
        }
        nd12[j] = v ^ 0x0C0C0C0Cu;
    }
    const uint32_t used = n - 4u * (uint32_t)(NW - 1);          // bytes of the last dword that belong to the needle: 1..4
    tail_mask = used >= 4u ? 0xFFFFFFFFu : ((1u << (8u * used)) - 1u);
    tail_pad = 0x0C0C0C0Cu & ~tail_mask;
}

}  // namespace ta
