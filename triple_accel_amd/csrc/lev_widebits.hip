// lev_widebits.hip -- gfx950 instantiations of the row-blocked bit-parallel kernel (lev_widebits_body.h).
#include <hip/hip_runtime.h>

#include "lev_widebits_body.h"
#include "ta_internal.h"

namespace ta {

template <int NWL, bool TRANS>
__global__ __launch_bounds__(64) void lev_widebits_kernel(LevParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    LevWideBits<DevWave, NWL, TRANS>::run(P, blockIdx.x, gridDim.x, lds);
}

// rows_per_lane: 32 or 64.  Persistent grid: one wavefront per block, as many blocks as fit next to each other.
hipError_t lev_widebits_launch(const LevParams &P, int rows_per_lane, bool trans, hipStream_t s, uint32_t *grid_out,
                               uint32_t *lds_out) {
    const int nwl = rows_per_lane / 32;
    const uint32_t lds = 33u * 64u * (uint32_t)nwl * 4u;
    int dev = 0, cus = 256;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint32_t per_cu = nwl == 2 ? 8u : 16u;            // 160 KB of LDS per CU / table bytes, rounded to whole waves per SIMD
    const uint32_t resident = (uint32_t)cus * per_cu;
    const uint32_t grid = P.n < resident ? P.n : resident;
    if (grid_out) *grid_out = grid;
    if (lds_out) *lds_out = lds;
    if (grid == 0) return hipSuccess;
    dim3 g(grid), b(64);
    if (nwl == 2) { if (trans) hipLaunchKernelGGL((lev_widebits_kernel<2, true>), g, b, lds, s, P); else hipLaunchKernelGGL((lev_widebits_kernel<2, false>), g, b, lds, s, P); }
    else { if (trans) hipLaunchKernelGGL((lev_widebits_kernel<1, true>), g, b, lds, s, P); else hipLaunchKernelGGL((lev_widebits_kernel<1, false>), g, b, lds, s, P); }
    return hipGetLastError();
}

}  // namespace ta
