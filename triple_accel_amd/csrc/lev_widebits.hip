// lev_widebits.hip -- gfx950 instantiations of the row-blocked bit-parallel kernel (lev_widebits_body.h).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "lev_widebits_body.h"
#include "ta_internal.h"

namespace ta {

template <int NWL, bool TRANS>
__global__ __launch_bounds__(64) void lev_widebits_kernel(LevParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    if (P.n_dev) {                                     // a list whose length only the device knows (the rounds of ta_levenshtein_exp_batch)
        LevParams Q = P;
        Q.n = *P.n_dev;
        LevWideBits<DevWave, NWL, TRANS>::run(Q, blockIdx.x, gridDim.x, lds);
        return;
    }
    LevWideBits<DevWave, NWL, TRANS>::run(P, blockIdx.x, gridDim.x, lds);
}

// rows_per_lane: 32 or 64.  Persistent grid: one wavefront per block, as many blocks as fit next to each other.
hipError_t lev_widebits_launch(const LevParams &P0, int rows_per_lane, uint64_t max_len, bool trans, hipStream_t s,
                               uint32_t *grid_out, uint32_t *lds_out) {
    LevParams P = P0;
    const int nwl = rows_per_lane / 32;
    const uint32_t lds = 21u * 64u * (uint32_t)nwl * 4u;
    int dev = 0, cus = 256;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint32_t per_cu = nwl == 2 ? 14u : 28u;           // 160 KB of LDS per CU / table bytes
    uint32_t resident = (uint32_t)cus * per_cu;
    if (const char *e = env_str("TA_WB_WAVES_PER_CU")) { int v = atoi(e); if (v > 0) resident = (uint32_t)cus * (uint32_t)v; }
    uint32_t grid = P.n < resident ? P.n : resident;
    P.bnd = nullptr; P.bnd_line = 0;
    if (max_len > 64ull * (uint64_t)rows_per_lane) {
        // pairs may span several stripes: 6 boundary lines (2 x {HP, HN, transposition term}) of one u32 per column per wave
        P.bnd_line = max_len + 66;
        const uint64_t per_wave = 6ull * P.bnd_line * sizeof(uint32_t), budget = 8ull << 30;
        if ((uint64_t)grid * per_wave > budget) grid = (uint32_t)(budget / per_wave ? budget / per_wave : 1);
        Scratch &sc = tls_scratch(6);
        if (sc.ensure((size_t)grid * per_wave) != TA_OK) return hipErrorOutOfMemory;
        P.bnd = (uint32_t *)sc.dev;
    }
    if (grid_out) *grid_out = grid;
    if (lds_out) *lds_out = lds;
    if (grid == 0) return hipSuccess;
    dim3 g(grid), b(64);
    set_last_kernel_name("lev_widebits_kernel<%d, %s>", nwl == 2 ? 2 : 1, trans ? "true" : "false");
    if (nwl == 2) { if (trans) hipLaunchKernelGGL((lev_widebits_kernel<2, true>), g, b, lds, s, P); else hipLaunchKernelGGL((lev_widebits_kernel<2, false>), g, b, lds, s, P); }
    else { if (trans) hipLaunchKernelGGL((lev_widebits_kernel<1, true>), g, b, lds, s, P); else hipLaunchKernelGGL((lev_widebits_kernel<1, false>), g, b, lds, s, P); }
    return hipGetLastError();
}

// ---- trace_on = true for ONE pair: the sweep also stores 3 bits per visited cell (P.trace, P.trace_cols set by the caller)
template <bool TRANS>
__global__ __launch_bounds__(64) void lev_widebits_trace_kernel(LevParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    LevWideBits<DevWave, 2, TRANS, true>::run(P, 0, 1, lds);
}

hipError_t lev_widebits_trace_launch(const LevParams &P, bool trans, hipStream_t s) {
    const uint32_t lds = 21u * 64u * 2u * 4u;
    if (trans) hipLaunchKernelGGL(lev_widebits_trace_kernel<true>, dim3(1), dim3(64), lds, s, P);
    else hipLaunchKernelGGL(lev_widebits_trace_kernel<false>, dim3(1), dim3(64), lds, s, P);
    return hipGetLastError();
}

// ---- one huge pair: tiles of the stripes' sweeps, one launch per diagonal of the (stripe, tile) grid (lev_widebits_body.h)
template <int NWL, bool TRANS>
__global__ __launch_bounds__(64) void lev_widebits_huge_kernel(typename LevWideBits<DevWave, NWL, TRANS>::Huge H, uint32_t d) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    LevWideBits<DevWave, NWL, TRANS>::run_tile(H, blockIdx.x, d, lds);
}

template <int NWL, bool TRANS>
static hipError_t huge_launch_t(const uint8_t *ap, const uint8_t *bp, uint32_t n, uint32_t m, uint32_t u, uint32_t k, uint32_t *out,
                                hipStream_t s, uint32_t *launches_out) {
    using K = LevWideBits<DevWave, NWL, TRANS>;
    typename K::Huge H;
    H.ap = ap; H.bp = bp; H.n = n; H.m = m; H.u = u; H.k = k; H.out = out;
    const uint32_t stripes = K::huge_stripes(n);
    uint64_t cb = ((uint64_t)m / 16 + 63) & ~(uint64_t)63;         // ~16 tiles per stripe, 1024..8192 steps each
    if (cb < 1024) cb = 1024;
    if (cb > 8192) cb = 8192;
    if (const char *e = env_str("TA_WB_TILE_STEPS")) { long v = atol(e); if (v >= 64) cb = (uint64_t)v & ~(uint64_t)63; }
    H.CB = (uint32_t)cb;
    H.line = (uint64_t)m + 66;
    Scratch &ls = tls_scratch(6), &ss = tls_scratch(8);    // (slots 4 and 5 hold the exp loop's subsets while this runs)
    if (ls.ensure((size_t)stripes * 3 * H.line * sizeof(uint32_t)) != TA_OK) return hipErrorOutOfMemory;
    if (ss.ensure((size_t)stripes * 64 * 16 * sizeof(uint32_t)) != TA_OK) return hipErrorOutOfMemory;
    H.lines = (uint32_t *)ls.dev; H.state = (uint32_t *)ss.dev;
    const uint32_t D = K::huge_diagonals(n, m, u, H.CB);
    const uint32_t lds = K::LDS_BYTES;
    for (uint32_t d = 0; d < D; d++) {
        // stripes q with a tile on this diagonal form a contiguous range; launching all stripes keeps the host loop trivial
        hipLaunchKernelGGL((lev_widebits_huge_kernel<NWL, TRANS>), dim3(stripes), dim3(64), lds, s, H, d);
    }
    if (launches_out) *launches_out = D;
    return hipGetLastError();
}

// a and b are device pointers to the ONE pair's strings (with read slack); out = the pair's result slot
hipError_t lev_widebits_huge_launch(const uint8_t *a, uint32_t a_len, const uint8_t *b, uint32_t b_len, uint32_t u, uint32_t k,
                                    int rows_per_lane, bool trans, uint32_t *out, hipStream_t s, uint32_t *launches_out) {
    const bool swap = a_len > b_len;                             // rows <- the shorter string
    const uint8_t *ap = swap ? b : a, *bp = swap ? a : b;
    const uint32_t n = swap ? b_len : a_len, m = swap ? a_len : b_len;
    if (rows_per_lane == 64) return trans ? huge_launch_t<2, true>(ap, bp, n, m, u, k, out, s, launches_out)
                                          : huge_launch_t<2, false>(ap, bp, n, m, u, k, out, s, launches_out);
    return trans ? huge_launch_t<1, true>(ap, bp, n, m, u, k, out, s, launches_out)
                 : huge_launch_t<1, false>(ap, bp, n, m, u, k, out, s, launches_out);
}

}  // namespace ta
