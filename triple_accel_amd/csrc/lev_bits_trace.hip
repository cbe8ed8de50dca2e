// lev_bits_trace.hip -- gfx950 instantiation of the checkpoint-and-recompute batch traceback for the unit-cost families
// (lev_bits_trace_body.h) + the last step that writes the runs as ta_edit records.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "lev_bits_trace_body.h"
#include "ta_internal.h"

namespace ta {

// One wavefront per block, a pair per lane: forward sweep with checkpoints, backward tile recomputation + walk (the script's runs, last run
// first, into the pair's run list); then every lane turns its own list round and writes it as ta_edit records -- the reference's Vec<Edit> in
// its final order (src/levenshtein.rs:561-606: the reference walks backwards and reverses) -- into the pair's slot of `cap` records (a script
// of more runs is cut: n_edits says how long it is).
template <bool TRANS, int TILE, int STILE, bool HAVE_CKPT = false>
__global__ __launch_bounds__(64) void lev_bits_trace_kernel(LevBitsTraceParams P, ta_edit *edits, uint32_t *n_edits, uint64_t cap) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint32_t nr = 0;
    LevBitsTrace<DevWave, TRANS, TILE, STILE, HAVE_CKPT>::run(P, blockIdx.x, lds, &nr);
    const uint32_t slot_idx = blockIdx.x * 64u + threadIdx.x;
    if (slot_idx >= P.n) return;
    const uint32_t pair = P.subset ? P.subset[slot_idx] : slot_idx;                                   // (the run list below: this lane's own stores, program order)
    n_edits[pair] = nr;
    if (P.packed_cap) return;                                                                         // (packed form: the walk has written the script in place)
    const uint32_t have = nr < P.runs_cap ? nr : P.runs_cap;
    const uint32_t *mine = P.runs + (uint64_t)pair * P.runs_cap;
    ta_edit *slot = edits + (uint64_t)pair * cap;
    // eight runs per round trip: loads and stores share one counter here, so a load waited for alone also waits for the stores in front of it
    // (one run per iteration: a memory round trip per run, 45 in a row for the longest list of a wavefront -- a quarter of its life)
    const uint32_t lim = (uint64_t)have < cap ? have : (uint32_t)cap;
    for (uint32_t t0 = 0; t0 < lim; t0 += 8u) {
        uint32_t w[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) w[u] = mine[t0 + u < lim ? have - 1u - (t0 + u) : 0u];
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++)
            if (t0 + u < lim) slot[t0 + u] = ta_edit{w[u] >> 29, 0u, (uint64_t)(w[u] & 0x1FFFFFFFu)};
    }
}

uint32_t lev_bits_trace_ckpt_words(bool trans) { return trans ? 5u : 2u; }
// columns per checkpoint (TA_TRACE_TILE = 8 | 16 | 32) and per fetch of the strings (TA_TRACE_STILE = 32 | 64)
uint32_t lev_bits_trace_tile() {
    if (const char *e = env_str("TA_TRACE_TILE")) { const int v = atoi(e); if (v == 8 || v == 16 || v == 32) return (uint32_t)v; }
    return 16u;                  // cfg2t with the kernel's own sweep, ms per million pairs at (TILE, STILE) = (8, 32) 1.62, (8, 64) 1.50, (16, 32) 1.54, (16, 64) 1.48, (32, 64) 1.86
                                 // (folded sweep: (16, 32) 1.29, (16, 64) 1.23; the round's first version: 2.12 / 2.04 / 2.04 / 2.02 / 2.31)
}
static uint32_t lev_bits_trace_stile(uint32_t tile) {
    uint32_t st = 64u;
    if (const char *e = env_str("TA_TRACE_STILE")) { const int v = atoi(e); if (v == 32 || v == 64 || v == 128) st = (uint32_t)v; }
    return st < tile ? tile : st;
}

// have_ckpt: the distance pass (lev_bits_s8_ckpt_kernel) left the checkpoints of tiles of 16 columns in P.ckpt: no forward sweep here
hipError_t lev_bits_trace_launch(const LevBitsTraceParams &P, bool trans, bool have_ckpt, ta_edit *edits, uint32_t *n_edits, uint64_t cap, hipStream_t s,
                                 uint32_t *grid_out, uint32_t *lds_out) {
    const uint32_t waves = (P.n + 63u) / 64u, tile = have_ckpt ? 16u : lev_bits_trace_tile();
    uint32_t stile = lev_bits_trace_stile(tile);
    if (grid_out) *grid_out = waves;
    if (waves == 0) return hipSuccess;
    if (have_ckpt) {
        set_last_kernel_name("lev_bits_trace_kernel<%s, 16, %u, true>", trans ? "true" : "false", stile);
#define TA_BTC(T_, SL_) do { const uint32_t lds = LevBitsTrace<DevWave, T_, 16, SL_, true>::LDS_PER_WAVE + 1024u * (uint32_t)env_int("TA_TRACE_LDS_PAD_KB"); if (lds_out) *lds_out = lds; \
        hipLaunchKernelGGL((lev_bits_trace_kernel<T_, 16, SL_, true>), dim3(waves), dim3(64), lds, s, P, edits, n_edits, cap); } while (0)
        if (stile == 128u) { if (trans) TA_BTC(true, 128); else TA_BTC(false, 128); }      // (an A/B: every line of a string touched 2-3 times instead of 3-4, at 34 more VGPRs)
        else if (stile == 64u) { if (trans) TA_BTC(true, 64); else TA_BTC(false, 64); }
        else { if (trans) TA_BTC(true, 32); else TA_BTC(false, 32); }
#undef TA_BTC
        return hipGetLastError();
    }
    set_last_kernel_name("lev_bits_trace_kernel<%s, %u, %u>", trans ? "true" : "false", tile, stile);
#define TA_BT(T_, TL_, SL_) do { const uint32_t lds = LevBitsTrace<DevWave, T_, TL_, SL_>::LDS_PER_WAVE; if (lds_out) *lds_out = lds; \
        hipLaunchKernelGGL((lev_bits_trace_kernel<T_, TL_, SL_>), dim3(waves), dim3(64), lds, s, P, edits, n_edits, cap); } while (0)
#define TA_BT2(TL_, SL_) do { if (trans) TA_BT(true, TL_, SL_); else TA_BT(false, TL_, SL_); } while (0)
    if (stile > 64u) stile = 64u;                        // (128-column string tiles exist for the folded sweep only)
    if (tile == 8u) { if (stile == 64u) TA_BT2(8, 64); else TA_BT2(8, 32); }
    else if (tile == 16u) { if (stile == 64u) TA_BT2(16, 64); else TA_BT2(16, 32); }
    else TA_BT2(32, 64);
#undef TA_BT2
#undef TA_BT
    return hipGetLastError();
}

}  // namespace ta
