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

}  // namespace ta
