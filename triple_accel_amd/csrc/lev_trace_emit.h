// lev_trace_emit.h -- the last step of a batch traceback: a pair's walk (2-bit codes, sixteen per word, the walk's first step = the script's
// LAST edit first) replayed FORWARDS from (0, 0), reading the two strings front to back to tell Match from Mismatch, the runs written as they
// close: ta_edit records {edit, count}, the reference's Vec<Edit> in its final order (src/levenshtein.rs:561-606: the reference walks backwards
// and reverses), into the pair's slot of `cap` records.  Returns the runs of the script; a script of more than `cap` runs is cut (the caller
// sees n_edits > cap).  x = the rows' string (the shorter one), y = the columns'; swap: the caller's a is the longer string (:386-390), the
// gaps are relabelled.  Device code, one lane per pair (the DP band kernel's walk and the bit-parallel trace kernel end with it).
#pragma once
#include <stdint.h>

#include "../../include/triple_accel_amd.h"

namespace ta {

__device__ __forceinline__ uint32_t trace_emit_runs(const uint32_t *my_path, uint32_t steps, const uint8_t *x, const uint8_t *y, bool swap,
                                                    ta_edit *slot, uint64_t cap) {
    uint32_t runs = 0, cur = 0xFFFFFFFFu, fi = 0, fj = 0, wcache = 0;
    uint64_t cnt = 0;
    // the strings eight bytes at a time (the blobs carry 16 bytes of slack): a load per eight steps instead of two per step
    typedef uint64_t u64u __attribute__((aligned(1)));
    uint64_t xc = 0, yc = 0;
    uint32_t xb = 0xFFFFFFFFu, yb = 0xFFFFFFFFu;                 // which 8-byte group the caches hold
    for (uint32_t t = steps; t-- > 0u;) {
        if ((t & 15u) == 15u || t == steps - 1u) wcache = my_path[t >> 4];
        if ((t & 15u) == 15u && wcache == 0u) {
            // a whole word of diagonal steps (sixteen of them): if the sixteen characters agree they are sixteen Matches
            const uint64_t x0 = *(const u64u *)(x + fi), x1 = *(const u64u *)(x + fi + 8u), y0 = *(const u64u *)(y + fj), y1 = *(const u64u *)(y + fj + 8u);
            if (x0 == y0 && x1 == y1) {
                if (cur != TA_EDIT_MATCH) {
                    if (cur != 0xFFFFFFFFu) { if (runs < cap) slot[runs] = ta_edit{cur, 0u, cnt}; runs++; }
                    cur = TA_EDIT_MATCH; cnt = 0;
                }
                cnt += 16; fi += 16u; fj += 16u; t -= 15u;
                continue;
            }
        }
        const uint32_t code = (wcache >> (2u * (t & 15u))) & 3u;
        uint32_t e;
        if (code == 0u) {
            if ((fi >> 3) != xb) { xb = fi >> 3; xc = *(const u64u *)(x + 8u * (uint64_t)xb); }
            if ((fj >> 3) != yb) { yb = fj >> 3; yc = *(const u64u *)(y + 8u * (uint64_t)yb); }
            e = (((xc >> (8u * (fi & 7u))) ^ (yc >> (8u * (fj & 7u)))) & 0xFFu) == 0u ? TA_EDIT_MATCH : TA_EDIT_MISMATCH; fi++; fj++;
        }
        else if (code == 1u) { e = swap ? TA_EDIT_BGAP : TA_EDIT_AGAP; fj++; }
        else if (code == 2u) { e = swap ? TA_EDIT_AGAP : TA_EDIT_BGAP; fi++; }
        else { e = TA_EDIT_TRANSPOSE; fi += 2u; fj += 2u; }
        if (e == cur) { cnt++; continue; }
        if (cur != 0xFFFFFFFFu) { if (runs < cap) slot[runs] = ta_edit{cur, 0u, cnt}; runs++; }
        cur = e; cnt = 1;
    }
    if (cur != 0xFFFFFFFFu) { if (runs < cap) slot[runs] = ta_edit{cur, 0u, cnt}; runs++; }
    return runs;
}

// The same replay for the walk KERNEL (lev_band.hip, round 6).  Round 4's form above read the strings eight bytes at a time and its path a word at a
// time straight from memory, every lane its own pair: 64 different lines per load instruction, gone from the L2 by the time the lane came back
// for the next eight bytes -- 11.7 GB of fabric-side reads per million 256-byte pairs for 0.5 GB of strings, 81 % of the kernel's wave-cycles
// waiting (profiles/r06/ab_wtrace.md).  Here a lane fetches 64 bytes of a string at once -- four 16-byte loads back to back: one trip to the
// fabric per half line instead of eight -- into its own LDS slot (72 bytes apart: at most two lanes per bank) and reads its bytes from there; the
// path words are interleaved over the chunk's pairs (word w of pair idx at path[w * pitch + idx]): the 64 lanes of a wavefront, all about as far
// along their paths, touch one or two lines per word instead of 64.
constexpr uint32_t TRACE_EMIT_SLOT = 72u;                          // bytes per lane and string in LDS
__device__ __forceinline__ void trace_emit_fill(const uint8_t *g, uint32_t len, uint8_t *slot, uint32_t base) {
    typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(1)));
    typedef uint32_t u32x2a __attribute__((ext_vector_type(2), aligned(8)));
    u32x4u v[4];
#pragma unroll
    for (uint32_t p = 0; p < 4u; p++) v[p] = (base + 16u * p <= len) ? *(const u32x4u *)(g + base + 16u * p) : u32x4u{0, 0, 0, 0};   // (a piece that starts at or before the end: inside the blob's 16 bytes of slack)
#pragma unroll
    for (uint32_t p = 0; p < 4u; p++) {
        *(u32x2a *)(slot + 16u * p) = u32x2a{v[p].x, v[p].y};
        *(u32x2a *)(slot + 16u * p + 8u) = u32x2a{v[p].z, v[p].w};
    }
}
__device__ __forceinline__ uint32_t trace_emit_runs_lds(const uint32_t *path, uint64_t pitch, uint32_t steps, const uint8_t *x, uint32_t n, const uint8_t *y, uint32_t m,
                                                        bool swap, ta_edit *slot, uint64_t cap, uint8_t *xs, uint8_t *ys) {
    uint32_t runs = 0, cur = 0xFFFFFFFFu, fi = 0, fj = 0, wcache = 0;
    uint64_t cnt = 0;
    typedef uint64_t u64a __attribute__((aligned(8)));
    typedef uint32_t u32u __attribute__((aligned(1)));
    uint32_t xbase = 0xFFFFFF00u, ybase = 0xFFFFFF00u;            // the 64-byte chunk each slot holds (nothing yet)
    uint64_t xc = 0, yc = 0;
    uint32_t xb = 0xFFFFFFFFu, yb = 0xFFFFFFFFu;                   // which 8-byte group the register caches hold
    for (uint32_t t = steps; t-- > 0u;) {
        if ((t & 15u) == 15u || t == steps - 1u) wcache = path[(uint64_t)(t >> 4) * pitch];
        if ((t & 15u) == 15u && wcache == 0u) {
            // a whole word of diagonal steps (sixteen of them): if the sixteen characters agree they are sixteen Matches
            if (fi - xbase >= 64u) { xbase = fi & ~63u; trace_emit_fill(x, n, xs, xbase); xb = 0xFFFFFFFFu; }
            if (fj - ybase >= 64u) { ybase = fj & ~63u; trace_emit_fill(y, m, ys, ybase); yb = 0xFFFFFFFFu; }
            if (fi - xbase <= 48u && fj - ybase <= 48u) {           // both runs of sixteen inside their chunks
                const uint8_t *px = xs + (fi - xbase), *py = ys + (fj - ybase);
                const bool eq = *(const u32u *)px == *(const u32u *)py && *(const u32u *)(px + 4) == *(const u32u *)(py + 4) &&
                                *(const u32u *)(px + 8) == *(const u32u *)(py + 8) && *(const u32u *)(px + 12) == *(const u32u *)(py + 12);
                if (eq) {
                    if (cur != TA_EDIT_MATCH) {
                        if (cur != 0xFFFFFFFFu) { if (runs < cap) slot[runs] = ta_edit{cur, 0u, cnt}; runs++; }
                        cur = TA_EDIT_MATCH; cnt = 0;
                    }
                    cnt += 16; fi += 16u; fj += 16u; t -= 15u;
                    continue;
                }
            }
        }
        const uint32_t code = (wcache >> (2u * (t & 15u))) & 3u;
        uint32_t e;
        if (code == 0u) {
            if ((fi >> 3) != xb) {
                if (fi - xbase >= 64u) { xbase = fi & ~63u; trace_emit_fill(x, n, xs, xbase); }
                xb = fi >> 3; xc = *(const u64a *)(xs + ((fi & 63u) & ~7u));
            }
            if ((fj >> 3) != yb) {
                if (fj - ybase >= 64u) { ybase = fj & ~63u; trace_emit_fill(y, m, ys, ybase); }
                yb = fj >> 3; yc = *(const u64a *)(ys + ((fj & 63u) & ~7u));
            }
            e = (((xc >> (8u * (fi & 7u))) ^ (yc >> (8u * (fj & 7u)))) & 0xFFu) == 0u ? TA_EDIT_MATCH : TA_EDIT_MISMATCH; fi++; fj++;
        }
        else if (code == 1u) { e = swap ? TA_EDIT_BGAP : TA_EDIT_AGAP; fj++; }
        else if (code == 2u) { e = swap ? TA_EDIT_AGAP : TA_EDIT_BGAP; fi++; }
        else { e = TA_EDIT_TRANSPOSE; fi += 2u; fj += 2u; }
        if (e == cur) { cnt++; continue; }
        if (cur != 0xFFFFFFFFu) { if (runs < cap) slot[runs] = ta_edit{cur, 0u, cnt}; runs++; }
        cur = e; cnt = 1;
    }
    if (cur != 0xFFFFFFFFu) { if (runs < cap) slot[runs] = ta_edit{cur, 0u, cnt}; runs++; }
    return runs;
}

}  // namespace ta
