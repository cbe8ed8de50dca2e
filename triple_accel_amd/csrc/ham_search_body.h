// ham_search_body.h -- hamming_search for needles of up to 32 bytes as a "shift-add" scan (Baeza-Yates & Gonnet 1992).
//
// Contract (src/hamming.rs:454-554, scalar text :89-145): for every offset p in [0, h - n] the number of
// mismatching bytes between needle and haystack[p .. p+n); reported when <= k.
//
// One lane scans a tile of offsets.  Its state is n byte-counters packed in NWS dwords: after haystack byte i,
// counter j holds the mismatches of haystack[i-j .. i] against needle[0 .. j].  A step is
//     S = (S << 8) + T[c]          T[c].byte[j] = (needle[j] != c)
// i.e. NWS v_alignbyte + NWS v_add (counters never exceed 32, so no carry crosses a byte), with T -- 256 x NWS
// dwords -- in LDS; counter n-1 is the finished count of offset i-n+1.  ~20 instructions per haystack byte for a
// 32-byte needle, where the SWAR compare-and-popcount kernel (any needle length, lev_search.hip) needs ~54.
// Plain per-lane code: tests run the same function on the CPU.
#pragma once
#include <stdint.h>

#include "wave.h"

namespace ta {

// word w of T[c]
TA_HD inline uint32_t ham_sa_table_word(const uint8_t *needle, uint32_t n, uint32_t c, uint32_t w) {
    uint32_t m = 0;
    for (uint32_t b = 0; b < 4; b++) {
        const uint32_t j = 4u * w + b;
        if (j < n && (uint32_t)needle[j] != c) m |= 1u << (8 * b);
    }
    return m;
}

template <int NWS>
struct HamSaState {
    uint32_t S[NWS];
};

// one haystack byte (Tc = the NWS words of T[c]); returns counter n-1
template <int NWS>
TA_HD inline __attribute__((always_inline)) uint32_t ham_sa_step(HamSaState<NWS> &s, const uint32_t (&Tc)[NWS], uint32_t n) {
#pragma unroll
    for (int w = NWS - 1; w >= 1; w--) s.S[w] = ((s.S[w] << 8) | (s.S[w - 1] >> 24)) + Tc[w];
    s.S[0] = (s.S[0] << 8) + Tc[0];
    const uint32_t j = n - 1u;
    uint32_t word = s.S[0];
#pragma unroll
    for (int w = 1; w < NWS; w++) word = (j >> 2) == (uint32_t)w ? s.S[w] : word;
    return (word >> (8u * (j & 3u))) & 0xffu;
}

// offsets [off_begin, off_end) of `hay` (every offset needs hay[p .. p+n)); emit(p, count) for count <= k
template <int NWS, class Table, class Emit>
TA_HD inline void ham_sa_tile(const uint8_t *hay, Table table, uint32_t n, uint32_t k, uint64_t off_begin, uint64_t off_end,
                              Emit emit) {
    HamSaState<NWS> s;
    for (int w = 0; w < NWS; w++) s.S[w] = 0;
    if (off_begin >= off_end) return;
    const uint64_t last = off_end - 1 + (n - 1);                   // last haystack byte this tile reads
    for (uint64_t i = off_begin; i <= last; i++) {
        uint32_t Tc[NWS];
        table(hay[i], Tc);
        const uint32_t cnt = ham_sa_step<NWS>(s, Tc, n);
        if (i >= off_begin + (n - 1) && cnt <= k) emit(i - (n - 1), cnt);
    }
}

}  // namespace ta
