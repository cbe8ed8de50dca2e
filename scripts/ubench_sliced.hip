// Micro-benchmark of the pair-sliced systolic band step (no transposition, synthetic planes in LDS): how many cycles does
// one lane-step (2 cells x 32 pairs per lane) cost?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t U32;
#define DPP_SHR(x) (U32) __builtin_amdgcn_update_dpp(0, (int)(x), 0x138, 0xf, 0xf, true)
#define DPP_SHL(x) (U32) __builtin_amdgcn_update_dpp(0, (int)(x), 0x130, 0xf, 0xf, true)
#define B3(a, b, c, t) (U32) __builtin_amdgcn_bitop3_b32((a), (b), (c), (t))

struct Bank {
    uint4 Plo, Phi, Qlo, Qhi;   // a-rows of the two cells (roles alternate every column)
    U32 Pv0, Mv0, Pv1, Mv1;     // this bank's dv outputs of its previous column
    U32 Ph, Mh;                 // dh output of the last cell
};
struct Planes { uint4 alo, ahi, blo, bhi; };

__device__ __forceinline__ U32 neq8(const uint4 &alo, const uint4 &ahi, const uint4 &blo, const uint4 &bhi) {
    U32 ne = alo.x ^ blo.x;
    ne = B3(ne, alo.y, blo.y, 0xF6); ne = B3(ne, alo.z, blo.z, 0xF6); ne = B3(ne, alo.w, blo.w, 0xF6);
    ne = B3(ne, ahi.x, bhi.x, 0xF6); ne = B3(ne, ahi.y, bhi.y, 0xF6); ne = B3(ne, ahi.z, bhi.z, 0xF6);
    return B3(ne, ahi.w, bhi.w, 0xF6);
}

__device__ __forceinline__ Planes fetch(const uint8_t *lds, U32 addr_a, U32 addr_b, int HI) {
    Planes p;
    p.alo = *(const uint4 *)(lds + addr_a); p.ahi = *(const uint4 *)(lds + addr_a + HI);
    p.blo = *(const uint4 *)(lds + addr_b); p.bhi = *(const uint4 *)(lds + addr_b + HI);
    return p;
}

template <bool FLIP>
__device__ __forceinline__ void step(Bank &me, const Bank &other, const Planes &in, U32 topm, U32 ntop, U32 botm, U32 nbot, U32 &zacc) {
    if (FLIP) { me.Plo = in.alo; me.Phi = in.ahi; } else { me.Qlo = in.alo; me.Qhi = in.ahi; }
    const uint4 &r0lo = FLIP ? me.Qlo : me.Plo, &r0hi = FLIP ? me.Qhi : me.Phi, &r1lo = FLIP ? me.Plo : me.Qlo, &r1hi = FLIP ? me.Phi : me.Qhi;
    const U32 ne0 = neq8(r0lo, r0hi, in.blo, in.bhi), ne1 = neq8(r1lo, r1hi, in.blo, in.bhi);
    const U32 ph = DPP_SHR(other.Ph) | topm, mh = DPP_SHR(other.Mh) & ntop;
    const U32 pvb = DPP_SHL(other.Pv0) | botm, mvb = DPP_SHL(other.Mv0) & nbot;
    const U32 z0 = B3(ne0, me.Mv1, mh, 0xEF);
    const U32 pv0 = B3(z0, ph, mh, 0xAB), mv0 = z0 & ph;
    const U32 ph0 = B3(z0, me.Pv1, me.Mv1, 0xAB), mh0 = z0 & me.Pv1;
    const U32 z1 = B3(ne1, mvb, mh0, 0xEF);
    me.Pv1 = B3(z1, ph0, mh0, 0xAB); me.Mv1 = z1 & ph0;
    me.Ph = B3(z1, pvb, mvb, 0xAB); me.Mh = z1 & pvb;
    me.Pv0 = pv0; me.Mv0 = mv0;
    zacc ^= z0;
}

template <int HI, int MODE>
__global__ __launch_bounds__(64) void k(uint32_t *out, int steps, int lds_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const U32 lane = threadIdx.x;
    for (int i = lane; i < lds_bytes / 4; i += 64) ((uint32_t *)lds)[i] = (i * 2654435761u) ^ (blockIdx.x * 40503u);
    __syncthreads();
    Bank b0, b1;
    b0.Plo = b0.Phi = b0.Qlo = b0.Qhi = b1.Plo = b1.Phi = b1.Qlo = b1.Qhi = make_uint4(lane, lane * 3, lane * 5, lane * 7);
    b0.Pv0 = b0.Pv1 = b1.Pv0 = b1.Pv1 = ~0u; b0.Mv0 = b0.Mv1 = b1.Mv0 = b1.Mv1 = 0; b0.Ph = b1.Ph = ~0u; b0.Mh = b1.Mh = 0;
    const U32 c = lane % 17, g = (lane / 17) % 3;
    const U32 topm = c == 0 ? ~0u : 0u, botm = c == 16 ? ~0u : 0u, ntop = ~topm, nbot = ~botm;
    // per (group, bank): a ring 64 positions x 16 B (lo planes) + the same for hi planes, for a and for b: 4 KB
    const U32 base_a0 = g * 8192, base_b0 = base_a0 + 2048, base_a1 = base_a0 + 4096, base_b1 = base_a0 + 6144;
    U32 ta = ((c * 3 / 2) * 16) & 1023, tb = ((c / 2) * 16) & 1023, zacc = 0;
    Planes n0 = fetch(lds, (ta & 1023) | base_a0, (tb & 1023) | base_b0, HI);
    for (int s = 0; s < steps; s += 4) {
        Planes n1 = n0;
        if (MODE != 1) n1 = fetch(lds, (ta & 1023) | base_a1, (tb & 1023) | base_b1, HI);
        if (MODE != 2) step<false>(b0, b1, n0, topm, ntop, botm, nbot, zacc); else zacc ^= n0.alo.x ^ n0.ahi.y ^ n0.blo.z ^ n0.bhi.w;
        ta += 16; tb += 16;
        if (MODE != 1) n0 = fetch(lds, (ta & 1023) | base_a0, (tb & 1023) | base_b0, HI);
        if (MODE != 2) step<false>(b1, b0, n1, topm, ntop, botm, nbot, zacc); else zacc ^= n1.alo.x ^ n1.ahi.y ^ n1.blo.z ^ n1.bhi.w;
        if (MODE != 1) n1 = fetch(lds, (ta & 1023) | base_a1, (tb & 1023) | base_b1, HI);
        if (MODE != 2) step<true>(b0, b1, n0, topm, ntop, botm, nbot, zacc); else zacc ^= n0.alo.x ^ n0.ahi.y ^ n0.blo.z ^ n0.bhi.w;
        ta += 16; tb += 16;
        if (MODE != 1) n0 = fetch(lds, (ta & 1023) | base_a0, (tb & 1023) | base_b0, HI);
        if (MODE != 2) step<true>(b1, b0, n1, topm, ntop, botm, nbot, zacc); else zacc ^= n1.alo.x ^ n1.ahi.y ^ n1.blo.z ^ n1.bhi.w;
        if (MODE == 1) { n0.alo.x += zacc; n0.bhi.y ^= ta; }
    }
    out[blockIdx.x * 64 + lane] = zacc ^ b0.Pv0 ^ b1.Mv1 ^ n0.alo.x;
}

int main(int argc, char **argv) {
    const int steps = 532, waves = argc > 1 ? atoi(argv[1]) : 5209, lds_bytes = argc > 2 ? atoi(argv[2]) : 31 * 1024;
    uint32_t *out; (void)hipMalloc(&out, waves * 64 * 4);
    const int mode = argc > 3 ? atoi(argv[3]) : 0;
    auto kern = mode == 0 ? k<1024, 0> : mode == 1 ? k<1024, 1> : k<1024, 2>;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        kern<<<waves, 64, lds_bytes>>>(out, steps, lds_bytes);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d waves %d lds %d: %.3f ms  (%.1f clk @2.4GHz per step per SIMD)\n", mode, waves, lds_bytes, ms, ms * 1e-3 * 2.4e9 / ((double)waves * steps / 1024.0));
    }
    return 0;
}
