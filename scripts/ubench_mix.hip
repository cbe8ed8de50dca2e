// Micro-benchmark: what does a MIXED VALU stream cost on gfx950?  (scripts/ubench_valu*.hip price every opcode on its own.)
//
//  part 1 -- two-opcode streams, independent registers, at 1..4 waves per SIMD:
//      full-rate only (v_xor), half-rate only (v_perm), strictly alternating, blocks of 8 / 8, blocks of 64 / 64,
//      v_bitop3 alternating with v_perm, and the kernel's own triple v_xor -> v_perm -> v_dot4 (dependent inside a triple).
//      If a full-rate op costs 2.3 cycles wherever it stands, "alternating" must come out at (2.3 + 4.2) / 2 = 3.25.
//  part 2 -- the band kernel's column code itself (LevBits::column, lev_bits_body.h) on register-resident inputs: the same
//      instruction stream as the kernel's inner loop minus the LDS reads and the HBM refills, at the kernel's occupancy.
//      ns per column per wavefront here against the same figure of the full kernel = how much of the kernel is VALU issue.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I triple_accel_amd/csrc scripts/ubench_mix.hip -o scripts/ubench_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "lev_bits_body.h"

#define REP8(x) x x x x x x x x
#define ITER 256

#define KERNEL(NAME, ASM, NINSTR)                                                                \
    __global__ __launch_bounds__(256) void NAME(unsigned *out) {                                 \
        unsigned r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4,           \
                 r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7, s = blockIdx.x | 1;                      \
        for (int i = 0; i < ITER; i++) {                                                         \
            asm volatile(REP8(ASM) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4),           \
                         "+v"(r5), "+v"(r6), "+v"(r7) : "v"(s));                                  \
        }                                                                                        \
        out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;             \
    }                                                                                            \
    static const int NAME##_n = NINSTR;

#define X(r) "v_xor_b32 %" #r ", %" #r ", %8\n"
#define P(r) "v_perm_b32 %" #r ", %" #r ", %8, %8\n"
#define B(r) "v_bitop3_b32 %" #r ", %" #r ", %8, %8 bitop3:0x48\n"
#define D(r) "v_dot4_u32_u8 %" #r ", %8, %8, %" #r "\n"

KERNEL(k_xor, X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7), 8)
KERNEL(k_perm, P(0) P(1) P(2) P(3) P(4) P(5) P(6) P(7), 8)
KERNEL(k_alt, X(0) P(1) X(2) P(3) X(4) P(5) X(6) P(7), 8)
KERNEL(k_blk8, X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) P(0) P(1) P(2) P(3) P(4) P(5) P(6) P(7), 16)
KERNEL(k_bitop_perm, B(0) P(1) B(2) P(3) B(4) P(5) B(6) P(7), 8)
KERNEL(k_bitop_xor, B(0) X(1) B(2) X(3) B(4) X(5) B(6) X(7), 8)
// the kernel's own match-vector triple: xor -> perm -> dot4 on one register, eight registers round robin
KERNEL(k_triple, X(0) X(1) X(2) X(3) P(0) P(1) P(2) P(3) D(0) D(1) D(2) D(3) X(4) X(5) X(6) X(7) P(4) P(5) P(6) P(7) D(4) D(5) D(6) D(7), 24)

// blocks of 64 xor then 64 perm
__global__ __launch_bounds__(256) void k_blk64(unsigned *out) {
    unsigned r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7, s = blockIdx.x | 1;
    for (int i = 0; i < ITER / 2; i++) {
        asm volatile(REP8(X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(s));
        asm volatile(REP8(P(0) P(1) P(2) P(3) P(4) P(5) P(6) P(7)) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(s));
    }
    out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
}
static const int k_blk64_n = 8;   // per REP8 element on average: ITER/2 trips x 128 = ITER x 64

// ---- part 2: the column code on register-resident inputs.  MODE 0 = sliding window, 1 = static window, 2 = stride-8 window (NA = 8)
template <int NA, bool TRANS, int MODE>
__global__ __launch_bounds__(256) void k_columns(unsigned *out, unsigned groups, unsigned seed) {
    extern __shared__ unsigned char lds_dummy[];        // only sizes the block's LDS footprint (= the kernel's occupancy)
    using LB = ta::LevBits<ta::DevWave, NA, TRANS, MODE == 1, false, MODE == 2>;
    typename LB::State st;
    unsigned x = (threadIdx.x + 1u) * 2654435761u ^ seed, y = x * 40503u + 7u;
#pragma unroll
    for (int q = 0; q < LB::NW; q++) { st.VP[q] = x >> q; st.VN[q] = ~st.VP[q] & (y << q); st.PMp[q] = 0; st.D0p[q] = ~0u; }
#pragma unroll
    for (int k = 0; k < NA; k++) st.AW[k] = x + 0x01010101u * k;
    st.acc = 0;
    unsigned cnt = 0;
    for (unsigned g = 0; g < groups; g++) {             // a group = 4 columns (stride-8 form: every other trip closes a block of 8)
        x = (x >> 1) ^ y; y += 0x9E3779B9u;               // stands in for the two ds_read_b32 of a group (full-rate ops)
        if constexpr (MODE == 2) {
            const unsigned ax = x ^ 0x0C0C0C0Cu;
            if (g & 1u) {
                LB::template step8<false, 4, true>(st, y, x, ax, true); LB::template step8<false, 5, true>(st, y, x, ax, true);
                LB::template step8<false, 6, true>(st, y, x, ax, true); LB::template step8<false, 7, true>(st, y, x, ax, true);
            } else {
                LB::template step8<false, 0, true>(st, y, x, ax, true); LB::template step8<false, 1, true>(st, y, x, ax, true);
                LB::template step8<false, 2, true>(st, y, x, ax, true); LB::template step8<false, 3, true>(st, y, x, ax, true);
            }
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int k2 = 0; k2 < NA - 1; k2++) st.AW[k2] = st.AW[k2 + 1];
            st.AW[NA - 1] = x ^ 0x0C0C0C0Cu;
            LB::template column<false, 0>(st, y, true);
            LB::template column<false, 1>(st, y, true);
            LB::template column<false, 2>(st, y, true);
            LB::template column<false, 3>(st, y, true);
        } else {
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) {
                LB::advance_a(st, (x >> (8 * s4)) & 0xFFu);
                LB::template column<false>(st, (y >> (8 * s4)) & 0xFFu, true);
            }
        }
        if ((g & 7u) == 7u) { cnt += __builtin_popcount(st.acc); }      // the zero-step register: 32 columns
    }
    unsigned acc = cnt;
#pragma unroll
    for (int q = 0; q < LB::NWF; q++) acc += st.VP[q] ^ st.VN[q];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (lds_dummy[0] == 77 && groups == 0xFFFFFFFFu) out[0] = lds_dummy[threadIdx.x];
}

typedef void (*kern_t)(unsigned *);
struct Entry { const char *name; kern_t k; int n; };

int main() {
    Entry tab[] = {{"v_xor only (full rate)", k_xor, k_xor_n}, {"v_perm only (half rate)", k_perm, k_perm_n},
                   {"xor/perm strictly alternating", k_alt, k_alt_n}, {"8 xor then 8 perm", k_blk8, k_blk8_n},
                   {"64 xor then 64 perm", k_blk64, k_blk64_n}, {"bitop3/perm alternating", k_bitop_perm, k_bitop_perm_n},
                   {"bitop3/xor alternating", k_bitop_xor, k_bitop_xor_n}, {"xor->perm->dot4 triples", k_triple, k_triple_n}};
    unsigned *out;
    hipMalloc(&out, 256 * 256 * 16 * 4 * sizeof(unsigned));
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("device %s  CUs %d  clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    const double ghz = p.clockRate / 1e6;
    for (int wps = 1; wps <= 4; wps++) {
        int blocks = p.multiProcessorCount * wps;
        printf("-- part 1, %d wave(s) per SIMD: cycles per wave-instruction per SIMD at %.2f GHz\n", wps, ghz);
        for (auto &e : tab) {
            hipEvent_t a, b;
            hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out);
            hipDeviceSynchronize();
            hipEventRecord(a);
            for (int r = 0; r < 5; r++) hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            double waveinstr = 5.0 * blocks * 4 * (double)ITER * 8 * e.n;
            double per_simd_per_s = waveinstr / (ms * 1e-3) / (p.multiProcessorCount * 4);
            printf("%-34s %8.3f ms  %.2f cycles/instr\n", e.name, ms, ghz * 1e9 / per_simd_per_s);
        }
    }
    // part 2: blocks of 4 waves; dynamic LDS picks the occupancy: 53000 B -> 3 blocks = 12 waves per CU (cfg2's launch), 38912 -> 16
    struct Col { const char *name; void (*k)(unsigned *, unsigned, unsigned); int vgpr_note; };
    Col cols[] = {{"cfg2 column code: stride-8 window, LEVENSHTEIN", k_columns<8, false, 2>, 0},
                  {"33 diagonals, static window (NA=9), LEVENSHTEIN", k_columns<9, false, 1>, 0},
                  {"cfg4 column code: NA=3, sliding window, RDAMERAU", k_columns<3, true, 0>, 0}};
    const unsigned groups = 2048;     // 8192 columns per wavefront
    for (auto &c : cols) {
        for (unsigned lds : {80000u, 53000u, 38912u}) {
            hipFuncSetAttribute((const void *)c.k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            const int blocks_per_cu = (int)(160u * 1024u / lds), blocks = p.multiProcessorCount * blocks_per_cu;
            hipEvent_t a, b;
            hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), lds, 0, out, groups, 1u);
            hipDeviceSynchronize();
            hipEventRecord(a);
            for (int r = 0; r < 3; r++) hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), lds, 0, out, groups, 2u + r);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            // every SIMD runs blocks_per_cu waves side by side, each 4 * groups columns; time per column per SIMD:
            const double cols_per_simd = 3.0 * blocks_per_cu * 4.0 * groups;
            const double ns = ms * 1e6 / cols_per_simd;
            printf("%-56s %2d waves/CU  %8.3f ms  %.1f ns per wave-column per SIMD = %.0f cycles at %.2f GHz\n", c.name,
                   blocks_per_cu * 4, ms, ns, ns * ghz, ghz);
        }
    }
    return 0;
}
