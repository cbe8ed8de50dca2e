// wave.h wave_max (DPP reduction) against a plain loop, every lane pattern that matters
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../triple_accel_amd/csrc/wave.h"
__global__ void k(const uint32_t *in, uint32_t *out) {
    out[blockIdx.x] = ta::DevWave::wave_max(in[blockIdx.x * 64 + threadIdx.x]);
}
int main() {
    const int n = 4096;
    uint32_t *h = new uint32_t[n * 64], *o = new uint32_t[n], *di, *dout;
    for (int b = 0; b < n; b++) for (int l = 0; l < 64; l++) {
        uint32_t r = (b * 64 + l) * 2654435761u;
        h[b * 64 + l] = b < 64 ? (l == b ? 1000u + b : (r & 255)) : b < 128 ? (l == (b - 64) ? 0xFFFFFFFFu : r) : b == 128 ? 0 : r >> (b & 31);
    }
    (void)hipMalloc(&di, n * 256); (void)hipMalloc(&dout, n * 4);
    (void)hipMemcpy(di, h, n * 256, hipMemcpyHostToDevice);
    k<<<n, 64>>>(di, dout);
    (void)hipMemcpy(o, dout, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < n; b++) { uint32_t m = 0; for (int l = 0; l < 64; l++) m = h[b * 64 + l] > m ? h[b * 64 + l] : m; if (m != o[b]) { if (bad++ < 5) printf("block %d got %u want %u\n", b, o[b], m); } }
    printf("bad %d of %d\n", bad, n);
    return 0;
}
