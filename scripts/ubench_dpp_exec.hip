// Does a DPP row shift treat an EXEC-disabled source lane as invalid (dst keeps `old`) on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *out) {
    unsigned lane = threadIdx.x, src = 1000 + lane, r_shl = 7, r_shr = 7, r_shl_bc = 7;
    if ((lane & 15) < 11) {
        r_shl = (unsigned)__builtin_amdgcn_update_dpp((int)(500 + lane), (int)src, 0x101, 0xf, 0xf, false);   // row_shl:1  lane i <- i+1
        r_shr = (unsigned)__builtin_amdgcn_update_dpp((int)(500 + lane), (int)src, 0x111, 0xf, 0xf, false);   // row_shr:1  lane i <- i-1
        r_shl_bc = (unsigned)__builtin_amdgcn_update_dpp(0, (int)src, 0x101, 0xf, 0xf, true);
    }
    out[lane] = r_shl; out[64 + lane] = r_shr; out[128 + lane] = r_shl_bc;
}
int main() {
    unsigned *d, h[192]; (void)hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d); (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int r = 0; r < 3; r++) { for (int i = 0; i < 32; i++) printf("%u ", h[r * 64 + i]); printf("\n"); }
    return 0;
}
