// lev_bits_trace.hip -- gfx950 instantiation of the checkpoint-and-recompute batch traceback for the unit-cost families
// (lev_bits_trace_body.h) + the forward replay that writes the runs (lev_trace_emit.h).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "lev_bits_trace_body.h"
#include "lev_trace_emit.h"
#include "ta_internal.h"

namespace ta {

// One wavefront per block, a pair per lane: forward sweep with checkpoints, backward tile recomputation + walk (codes into the pair's path
// words), then every lane replays its own path forwards and writes its runs.
template <bool TRANS, int TILE>
__global__ __launch_bounds__(64) void lev_bits_trace_kernel(LevBitsTraceParams P, ta_edit *edits, uint32_t *n_edits, uint64_t cap) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    LevBitsTrace<DevWave, TRANS, TILE>::run(P, blockIdx.x, lds);
    const uint32_t pair = blockIdx.x * 64u + threadIdx.x;
    if (pair >= P.n) return;
    const uint32_t steps = P.steps[pair];                      // (this lane's own stores: program order)
    if (P.dist[pair] == 0xFFFFFFFFu) { n_edits[pair] = 0; return; }
    const uint8_t *x, *y;
    uint64_t n, m;
    if (P.a.off) { x = P.a.blob + P.a.off[pair]; n = P.a.off[pair + 1] - P.a.off[pair]; } else { x = P.a.blob + (uint64_t)pair * P.a.stride; n = P.a.len; }
    if (P.b.off) { y = P.b.blob + P.b.off[pair]; m = P.b.off[pair + 1] - P.b.off[pair]; } else { y = P.b.blob + (uint64_t)pair * P.b.stride; m = P.b.len; }
    const bool swap = n > m;                                   // the kernel ran the shorter string along the rows (:386-390)
    if (swap) { const uint8_t *t = x; x = y; y = t; }
    n_edits[pair] = trace_emit_runs(P.path + (uint64_t)pair * P.path_words, steps, x, y, swap, edits + (uint64_t)pair * cap, cap);
}

uint32_t lev_bits_trace_ckpt_words(bool trans) { return trans ? 5u : 2u; }
uint32_t lev_bits_trace_tile() {
    if (const char *e = env_str("TA_TRACE_TILE")) { if (atoi(e) == 32) return 32u; if (atoi(e) == 16) return 16u; }
    return 16u;
}

hipError_t lev_bits_trace_launch(const LevBitsTraceParams &P, bool trans, ta_edit *edits, uint32_t *n_edits, uint64_t cap, hipStream_t s,
                                 uint32_t *grid_out, uint32_t *lds_out) {
    const uint32_t waves = (P.n + 63u) / 64u, tile = lev_bits_trace_tile();
    if (grid_out) *grid_out = waves;
    if (waves == 0) return hipSuccess;
    set_last_kernel_name("lev_bits_trace_kernel<%s, %u>", trans ? "true" : "false", tile);
#define TA_BT(T_, TL_) do { const uint32_t lds = LevBitsTrace<DevWave, T_, TL_>::LDS_PER_WAVE; if (lds_out) *lds_out = lds; \
        hipLaunchKernelGGL((lev_bits_trace_kernel<T_, TL_>), dim3(waves), dim3(64), lds, s, P, edits, n_edits, cap); } while (0)
    if (tile == 32u) { if (trans) TA_BT(true, 32); else TA_BT(false, 32); }
    else { if (trans) TA_BT(true, 16); else TA_BT(false, 16); }
#undef TA_BT
    return hipGetLastError();
}

}  // namespace ta
