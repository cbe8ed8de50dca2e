// lev_search_wave_body.h -- the exact search recurrence of lev_search_body.h on ONE WAVEFRONT per block of end positions.
//
// Replaces levenshtein_search_simd_core_* (src/levenshtein.rs:2157-2451) for the blocks the candidate filter flags; the
// per-cell rules (cost, companion match length, every tie) are the scalar text's (src/levenshtein.rs:1709-1806), in the
// packed form of lev_search_tile_packed: key = (cost << 16) | (0xFFFF - length), "cheaper wins, then longer wins" = one
// unsigned minimum, quirk Q2 (:1755-1760) a v_bfi splice.
//
// Layout (the reference's own, SURVEY.md C.2, turned from SIMD lanes into a wavefront): lane r owns needle row r + 1 and
// the lanes run SKEWED -- at step s lane r computes haystack column s - r -- so everything a cell needs from the row above
// was produced by lane r - 1 one step earlier and arrives by DPP wave_shr:1:
//     up    = V(r-1, i)      lane r-1's newest value              (haystack gap opens from it, quirk Q2 reads its length)
//     hg    = HG(r-1, i)     lane r-1's haystack-gap chain
//     diag  = V(r-1, i-1)    lane r-1's value BEFORE its last update
//     z     = V(r-2, i-2)    the diag lane r-1 used two steps ago (transposition)
//     c     = hay[i]         the byte lane r-1 used one step ago; lane 0 takes step s's byte from the block's bytes, which
//                            the 64 lanes loaded with one coalesced access per 64 columns
// A block of 64 end positions behind `halo` bytes of left context is (64 + halo + n - 1) dependent steps of ~20
// instructions instead of (64 + halo) x n cells in one lane (the lane-per-block form: 140 us for cfg5's ~1.5 K flagged
// blocks at < 3 % occupancy).  Valid under the packed form's conditions (every cost < SRCH_PACKED_KINF, lengths < 0xFFFF);
// unanchored searches only (row 0 is the constant 0).  Needles up to 64 bytes.
// Written over the wave policy W (wave.h): tests/emu runs the same body on the host.
#pragma once
#include <stdint.h>

#include "lev_search_body.h"
#include "wave.h"

namespace ta {

constexpr uint32_t SRCH_WAVE_MAX_COLS = 256;    // bytes a block may span (emitted columns + left context): 4 registers of 64

// The last row's value of every emitted column goes to LDS as it is finished -- one ds_write under the lane mask of lane n - 1 per
// step, nothing of it on the dependent chain (rounds 2-3 read it back to the scalar unit every step: v_readlane, s_cmp, a branch, 44 us
// for cfg5's flagged blocks) -- and ONE vector pass after the last step tests them all: lane t holds emitted column t (a block emits at
// most SRCH_WAVE_MAX_HITS = 64 end positions), flush(hit, keys, cols) with hit = "cost <= k" per lane, cost = keys >> 16, length =
// 0xFFFF - (keys & 0xFFFF), end = col_begin + cols + 1.  (One atomic cursor bump per block.)  lds: 64 dwords of this wavefront's own.
// `needle` must be readable per lane (global / kernarg memory on the device).
constexpr uint32_t SRCH_WAVE_MAX_HITS = 64;
template <class W, bool TRANS, class Flush>
TA_HD inline void lev_search_block_wave(const uint8_t *hay, const uint8_t *needle, uint32_t n, const SearchCosts &C,
                                        uint64_t col_begin, uint64_t emit_begin, uint64_t col_end, uint8_t *lds, Flush flush) {
    using U32 = typename W::U32;
    using Bool = typename W::Bool;
    if (col_begin >= col_end || n == 0) return;
    const uint32_t ncols = (uint32_t)(col_end - col_begin);               // <= SRCH_WAVE_MAX_COLS (the caller's contract)
    const uint32_t e0 = (uint32_t)(emit_begin - col_begin);
    const U32 lane = W::lane();
    const U32 nb = W::gload_u8(W::ptr_add(W::ptr_splat(needle), lane), lane < W::splat(n));     // needle[j-1], j = lane + 1
    const U32 nb2 = W::from_lower(nb, W::splat(0u));                                             // needle[j-2]
    U32 B[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const U32 idx = lane + W::splat(64u * (uint32_t)q);
        B[q] = W::gload_u8(W::ptr_add(W::ptr_splat(hay + col_begin), idx), idx < W::splat(ncols));
    }
    const uint32_t SGC = (C.sg + C.gc) << 16, GC = C.gc << 16;
    const uint32_t SUB_MIS = (C.mc << 16) - 1u, SUB_EQ = 0xFFFFFFFFu;     // (+mc, length+1) and (+0, length+1)
    const uint32_t TCK = (C.tc << 16) - 2u;                                // (+tc, length+2)
    constexpr uint32_t KINF = (SRCH_PACKED_KINF << 16) | 0xFFFFu;
    constexpr uint32_t ROW0 = 0xFFFFu;                                     // row 0 of an unanchored search: cost 0, length 0
    // the fresh-start column (:1685-1690): dp1[j] = j*gc + sg
    U32 v = (((lane + W::splat(1u)) * W::splat(C.gc) + W::splat(C.sg)) << 16) | W::splat(0xFFFFu);
    U32 ng = W::splat(KINF), hg = W::splat(KINF);
    U32 oldp = v;                              // this lane's value before its last update
    U32 dcur = W::splat(ROW0), dprev = W::splat(ROW0);   // the diag this lane used one / two steps ago
    U32 c = W::splat(0u);                     // the byte this lane used one step ago = hay[i-1]
    const Bool last_row = lane == W::splat(n - 1u);
    const uint32_t steps = ncols + n - 1;
    const uint32_t first_emit = e0 + n - 1;    // lane n-1 reaches column e0 at this step
    uint32_t s = 0;
#pragma unroll
    for (int q = 0; q < 5; q++) {              // 4 byte registers, then the n-1 draining steps (no new bytes)
        const uint32_t s_end = q < 4 ? (steps < 64u * (uint32_t)(q + 1) ? steps : 64u * (uint32_t)(q + 1)) : steps;
        for (; s < s_end; s++) {
            const uint32_t cb = (q < 4 && s < ncols) ? W::readlane(B[q < 4 ? q : 0], s & 63u) : 0u;
            const U32 cn = W::from_lower(c, W::splat(cb));
            const U32 up = W::from_lower(v, W::splat(ROW0));
            const U32 hgp = W::from_lower(hg, W::splat(KINF));
            const U32 diag = W::from_lower(oldp, W::splat(ROW0));
            const U32 old = v;
            // needle gap (consumes the haystack byte): open from dp1[j] or extend; length + 1          :1726-1737
            const U32 a = W::umin(old + W::splat(SGC - 1u), ng + W::splat(GC - 1u));
            // haystack gap (skips a needle char): open from dp2[j-1] or extend; length unchanged        :1739-1750
            const U32 h = W::umin(up + W::splat(SGC), hgp + W::splat(GC));
            const U32 hq = W::bfi(0xFFFF0000u, h, up);                     // cost of the gap, length2[j-1]  (Q2)
            U32 vv = W::sel(hq < a, h, a);                                 // :1752-1760
            const U32 subk = diag + W::sel(nb != cn, W::splat(SUB_MIS), W::splat(SUB_EQ));   // :1724
            vv = W::umin(vv, subk);                                        // :1762-1765
            if (TRANS) {                                                   // :1767-1779 (<=)
                const U32 z = W::from_lower(dprev, W::splat(ROW0));        // dp0[j-2], length0[j-2]
                const U32 t = z + W::splat(TCK);
                const Bool cond = W::land(W::land(lane > W::splat(0u), lane < W::splat(s)), W::land(nb == c, nb2 == cn));
                vv = W::sel(W::land(cond, (t & W::splat(0xFFFF0000u)) <= vv), t, vv);
                dprev = dcur; dcur = diag;
            }
            const Bool act = lane <= W::splat(s);                          // lane r starts at step r with the fresh column
            v = W::sel(act, vv, v);
            ng = W::sel(act, a, ng);
            hg = h; oldp = old; c = cn;
            if (s >= first_emit && s - first_emit < SRCH_WAVE_MAX_HITS)   // lane n-1 has just finished column s - (n-1)
                W::lds_write32p(lds, W::splat(4u * (s - first_emit)), v, last_row);
        }
    }
    W::lds_wave_sync();
    const uint32_t n_emit = steps - first_emit < SRCH_WAVE_MAX_HITS ? steps - first_emit : SRCH_WAVE_MAX_HITS;      // = ncols - e0
    const Bool mine = lane < W::splat(n_emit);
    const U32 key = W::sel(mine, W::lds_read32(lds, lane << 2), W::splat(0xFFFFFFFFu));
    const Bool hit = W::land(mine, (key >> 16) <= W::splat(C.k));                                   // :1792-1806
    if (W::any(hit)) flush(hit, key, lane + W::splat(e0));
    W::lds_wave_sync();                                                  // (the next block of this wavefront writes the same words)
}

}  // namespace ta
