// lev_widebits.hip -- gfx950 instantiations of the row-blocked bit-parallel kernel (lev_widebits_body.h).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "lev_widebits_body.h"
#include "ta_internal.h"

namespace ta {

template <int NWL, bool TRANS>
__global__ __launch_bounds__(64) void lev_widebits_kernel(LevParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
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
    if (const char *e = getenv("TA_WB_WAVES_PER_CU")) { int v = atoi(e); if (v > 0) resident = (uint32_t)cus * (uint32_t)v; }
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
    if (nwl == 2) { if (trans) hipLaunchKernelGGL((lev_widebits_kernel<2, true>), g, b, lds, s, P); else hipLaunchKernelGGL((lev_widebits_kernel<2, false>), g, b, lds, s, P); }
    else { if (trans) hipLaunchKernelGGL((lev_widebits_kernel<1, true>), g, b, lds, s, P); else hipLaunchKernelGGL((lev_widebits_kernel<1, false>), g, b, lds, s, P); }
    return hipGetLastError();
}

}  // namespace ta
