// v_perm_b32 with all-ones sources as a byte test: selector 12 -> 0x00, everything else -> 0xFF; v_dot4_i32_i8 packs the flags.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(const uint32_t *in, uint32_t *out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x = in[i];
    uint32_t f = __builtin_amdgcn_perm(0xFFFFFFFFu, 0xFFFFFFFFu, x);
    out[2 * i] = f;
    out[2 * i + 1] = (uint32_t)__builtin_amdgcn_sdot4((int)f, (int)0xF8FCFEFFu, 0, false);
}
int main() {
    const int n = 1 << 16;
    uint32_t *h = new uint32_t[n], *o = new uint32_t[2 * n], *di, *dout;
    for (int i = 0; i < n; i++) { uint32_t r = i * 2654435761u; h[i] = (i < 65536 / 2) ? ((i & 255) | ((i >> 8) << 8) | 0x0C0C0000u * (i & 1)) : r; if (i % 7 == 0) h[i] = (h[i] & 0xFFFF00FFu) | 0x0C00u; }
    (void)hipMalloc(&di, n * 4); (void)hipMalloc(&dout, n * 8);
    (void)hipMemcpy(di, h, n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(di, dout, n);
    (void)hipMemcpy(o, dout, n * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) {
        uint32_t f = 0, m = 0;
        for (int b = 0; b < 4; b++) if (((h[i] >> (8 * b)) & 255) != 12) { f |= 0xFFu << (8 * b); m |= 1u << b; }
        if (o[2 * i] != f || o[2 * i + 1] != m) { if (bad++ < 5) printf("x=%08x f=%08x want %08x dot=%u want %u\n", h[i], o[2 * i], f, o[2 * i + 1], m); }
    }
    printf("bad %d of %d\n", bad, n);
    return 0;
}
