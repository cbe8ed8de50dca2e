// Micro-benchmark behind DESIGN.md 3.8 (the reference's 8/16/32-bit cell ladder, src/levenshtein.rs:766-791): one lane's share
// of the general-cost DP band recurrence -- 32 diagonals, in-place cells, alternating parities, linear gaps, mismatch flags
// from the v_perm byte test -- written twice:
//   u32 : one cell per VGPR, substitution = v_dot4_u32_u8 (flag byte x mismatch cost + cell), three-way v_min3_u32, even cells
//         stored biased by +gc, odd ones raw with a +2gc copy (what lev_band_body.h does: 2.5 arithmetic instructions per cell)
//   u16 : two cells per VGPR, v_pk_add_u16 clamp / v_pk_min_u16 (saturating like the reference's u16 jewel vectors), the
//         lower neighbour pair through one v_alignbit, flags widened to 16-bit masks by one v_perm
// Both compute the same recurrence on the same pseudo-random characters; the result checksums must agree (cells stay far
// below 65535).  Reports ns per cell-update per lane and the VGPR budget of each form.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench_cellwidth.hip -o scripts/ubench_cellwidth
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ u16x2 as16(unsigned x) { return __builtin_bit_cast(u16x2, x); }
static __device__ __forceinline__ unsigned as32(u16x2 x) { return __builtin_bit_cast(unsigned, x); }
static __device__ __forceinline__ unsigned pk_add_sat(unsigned a, unsigned b) { return as32(__builtin_elementwise_add_sat(as16(a), as16(b))); }
static __device__ __forceinline__ unsigned pk_min(unsigned a, unsigned b) { return as32(__builtin_elementwise_min(as16(a), as16(b))); }
static __device__ __forceinline__ unsigned umin3(unsigned a, unsigned b, unsigned c) { a = a < b ? a : b; return a < c ? a : c; }
static __device__ __forceinline__ unsigned ne12(unsigned x) { return __builtin_amdgcn_perm(0xFFFFFFFFu, 0xFFFFFFFFu, x); }

constexpr int H = 16;        // cells per parity per lane (32 diagonals)

// characters: two streams of pseudo-random bytes over a 4-letter alphabet (a quarter of the cells match)
static __device__ __forceinline__ unsigned next_chars(unsigned &s) { s = s * 1664525u + 1013904223u; return (s >> 8) & 0x03030303u; }

// u32: even cells stored biased by + gc (Eb), odd cells raw (O) with a + 2 gc copy (O2): 2 instructions per even cell, 3 per odd
__global__ __launch_bounds__(256) void k_u32(unsigned *out, unsigned steps, unsigned mc, unsigned gc) {
    unsigned Eb[H], O[H], O2[H], sa = threadIdx.x * 2654435761u + 1u, sb = sa ^ 0x9E3779B9u;
    const unsigned INF = 0x3FFFFFFFu;
#pragma unroll
    for (int c = 0; c < H; c++) { Eb[c] = 2u * c * gc + gc; O[c] = (2u * c + 1u) * gc; O2[c] = O[c] + 2u * gc; }
    for (unsigned s = 0; s < steps; s++) {
        unsigned F[H / 4];
#pragma unroll
        for (int w = 0; w < H / 4; w++) F[w] = ne12(next_chars(sa) ^ next_chars(sb) ^ 0x0C0C0C0Cu) & 0x01010101u;
#pragma unroll
        for (int c = 0; c < H; c++)                         // even parity: e + gc = min(e_old + gc + sub, o[c-1] + 2 gc, o[c] + 2 gc)
            Eb[c] = umin3(__builtin_amdgcn_udot4(F[c >> 2], mc << (8 * (c & 3)), Eb[c], false), c ? O2[c - 1] : INF, O2[c]);
#pragma unroll
        for (int w = 0; w < H / 4; w++) F[w] = ne12(next_chars(sa) ^ next_chars(sb) ^ 0x0C0C0C0Cu) & 0x01010101u;
#pragma unroll
        for (int c = 0; c < H; c++) {                       // odd parity: o = min(o_old + sub, e[c] + gc, e[c+1] + gc)
            O[c] = umin3(__builtin_amdgcn_udot4(F[c >> 2], mc << (8 * (c & 3)), O[c], false), Eb[c], c + 1 < H ? Eb[c + 1] : INF);
            O2[c] = O[c] + 2u * gc;
        }
    }
    unsigned acc = 0;
#pragma unroll
    for (int c = 0; c < H; c++) acc += (Eb[c] - gc) * 3u + O[c];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// packed u16, the same bookkeeping: Eb = {e + gc} pairs, O raw pairs, O2 = O + 2 gc; saturating adds (cells never come near 65535 here)
__global__ __launch_bounds__(256) void k_u16(unsigned *out, unsigned steps, unsigned mc, unsigned gc) {
    unsigned Eb[H / 2], O[H / 2], O2[H / 2], sa = threadIdx.x * 2654435761u + 1u, sb = sa ^ 0x9E3779B9u;
    const unsigned MC = mc | (mc << 16), GC2 = (2u * gc) | ((2u * gc) << 16);
#pragma unroll
    for (int i = 0; i < H / 2; i++) {
        Eb[i] = (2u * (2 * i) * gc + gc) | ((2u * (2 * i + 1) * gc + gc) << 16);
        O[i] = ((2u * (2 * i) + 1u) * gc) | (((2u * (2 * i + 1) + 1u) * gc) << 16);
        O2[i] = pk_add_sat(O[i], GC2);
    }
    for (unsigned s = 0; s < steps; s++) {
        unsigned F[H / 4];
#pragma unroll
        for (int w = 0; w < H / 4; w++) F[w] = ne12(next_chars(sa) ^ next_chars(sb) ^ 0x0C0C0C0Cu);      // 0x00 / 0xFF per cell
#pragma unroll
        for (int i = 0; i < H / 2; i++) {                   // even parity, cells 2i and 2i+1
            const unsigned m16 = __builtin_amdgcn_perm(0u, F[i >> 1], (i & 1) ? 0x03030202u : 0x01010000u) & MC;   // bytes -> 16-bit masks & mc
            const unsigned lo = i ? __builtin_amdgcn_alignbit(O2[i], O2[i - 1], 16) : ((O2[0] << 16) | 0xFFFFu);   // {o[2i-1], o[2i]} + 2 gc
            Eb[i] = pk_min(pk_add_sat(Eb[i], m16), pk_min(lo, O2[i]));
        }
#pragma unroll
        for (int w = 0; w < H / 4; w++) F[w] = ne12(next_chars(sa) ^ next_chars(sb) ^ 0x0C0C0C0Cu);
#pragma unroll
        for (int i = 0; i < H / 2; i++) {                   // odd parity: neighbours {e[2i], e[2i+1]} and {e[2i+1], e[2i+2]}, both + gc
            const unsigned m16 = __builtin_amdgcn_perm(0u, F[i >> 1], (i & 1) ? 0x03030202u : 0x01010000u) & MC;
            const unsigned hi = i + 1 < H / 2 ? __builtin_amdgcn_alignbit(Eb[i + 1], Eb[i], 16) : ((Eb[i] >> 16) | 0xFFFF0000u);
            O[i] = pk_min(pk_add_sat(O[i], m16), pk_min(Eb[i], hi));
            O2[i] = pk_add_sat(O[i], GC2);
        }
    }
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < H / 2; i++)
        acc += ((Eb[i] & 0xFFFFu) - gc) * 3u + ((Eb[i] >> 16) - gc) * 3u + (O[i] & 0xFFFFu) + (O[i] >> 16);
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    unsigned *out;
    (void)hipMalloc(&out, 256 * 64 * 256 * sizeof(unsigned));
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const unsigned steps = 2000, mc = 3, gc = 2;
    unsigned *h0 = (unsigned *)malloc(256 * sizeof(unsigned)), *h1 = (unsigned *)malloc(256 * sizeof(unsigned));
    hipLaunchKernelGGL(k_u32, dim3(1), dim3(256), 0, 0, out, 40u, mc, gc);
    (void)hipMemcpy(h0, out, 256 * 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k_u16, dim3(1), dim3(256), 0, 0, out, 40u, mc, gc);
    (void)hipMemcpy(h1, out, 256 * 4, hipMemcpyDeviceToHost);
    int same = 1;
    for (int i = 0; i < 256; i++) same &= h0[i] == h1[i];
    printf("device %s; u32 and u16 forms agree on 40 steps: %s\n", p.gcnArchName, same ? "yes" : "NO");
    struct { const char *name; void (*k)(unsigned *, unsigned, unsigned, unsigned); } tab[] = {
        {"u32 cells (v_dot4_u32_u8 + v_min3_u32)", k_u32}, {"packed u16 cells (v_pk_add_u16 clamp + v_pk_min_u16)", k_u16}};
    for (int wps = 2; wps <= 4; wps++) {
        const int blocks = p.multiProcessorCount * wps;
        for (auto &e : tab) {
            hipEvent_t a, b;
            (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, steps, mc, gc);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(a);
            for (int r = 0; r < 3; r++) hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, steps, mc, gc);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            float ms;
            (void)hipEventElapsedTime(&ms, a, b);
            const double cells_per_simd = 3.0 * wps * steps * 2.0 * H;      // wave-level cell updates per SIMD (64 lanes each)
            printf("%d waves/SIMD  %-52s %8.3f ms  %.2f ns per 64-lane cell update per SIMD = %.1f cycles at %.1f GHz\n", wps, e.name, ms,
                   ms * 1e6 / cells_per_simd, ms * 1e6 / cells_per_simd * p.clockRate / 1e6, p.clockRate / 1e6);
        }
    }
    return 0;
}
