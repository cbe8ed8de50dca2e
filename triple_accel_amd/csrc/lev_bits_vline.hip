// lev_bits_vline.hip -- gfx950 instantiations of the bit-parallel band kernel (lev_bits_body.h) in the VLINE form of the fetch:
// CSR batches, every pair with its own lengths, band geometry and alignment; each 128-byte line that holds bytes of a string is
// requested once, whole, by the pair's lane.  Its own translation unit: lev_bits.hip already takes two minutes to compile.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "lev_bits_body.h"
#include "lev_plan.h"
#include "ta_internal.h"

namespace ta {

// one wavefront per block (as the stride-8 line form: finer grains at the launch's tail); 128 VGPRs = four wavefronts per SIMD
template <bool TRANS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void lev_bits_s8v_kernel(LevParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    LevParams Q = P;
    if (P.n_dev) Q.n = *P.n_dev;
    const uint32_t waves = (Q.n + 63u) >> 6;
    for (uint32_t w = blockIdx.x; w < waves; w += gridDim.x)
        LevBits<DevWave, 8, TRANS, false, false, true, false, true>::run(Q, w, lds);
}

hipError_t lev_bits_vline_launch(const LevParams &P, const LevBitsPlan &pl, bool trans, hipStream_t s, uint32_t *grid_out, uint32_t *lds_out) {
    if (!pl.s8) return hipErrorInvalidValue;
    const uint32_t waves = (P.n + 63u) / 64u;
    uint32_t grid = waves;
    if (P.n_dev && grid > 512u) grid = 512u;
    const size_t lds = LEV_BITS_VLINE_LDS;
    if (grid_out) *grid_out = grid;
    if (lds_out) *lds_out = (uint32_t)lds;
    if (grid == 0) return hipSuccess;
    set_last_kernel_name("lev_bits_s8v_kernel<%s>", trans ? "true" : "false");
    if (trans) hipLaunchKernelGGL((lev_bits_s8v_kernel<true>), dim3(grid), dim3(64), lds, s, P);
    else hipLaunchKernelGGL((lev_bits_s8v_kernel<false>), dim3(grid), dim3(64), lds, s, P);
    return hipGetLastError();
}

}  // namespace ta
