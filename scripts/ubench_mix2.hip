// Micro-benchmark (round 4): can the cfg2 column's byte test leave the half-rate class?  scripts/ubench_mix.hip showed that a stream that
// alternates v_xor (2.2 cycles alone) and v_perm (4.2) costs 4.05 per instruction, not 3.2.  Here: how a FEW half-rate instructions among
// many full-rate ones are priced (1 in 4 / 8 / 16 / 32), and what the opcodes a SWAR byte test would be made of cost alone and mixed
// (v_bitop3, v_add_u32, v_lshrrev_b32 with a constant, v_and / v_or).  Eight independent registers, in-place, 1..4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench_mix2.hip -o scripts/ubench_mix2
#include <hip/hip_runtime.h>
#include <stdio.h>

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
#define A(r) "v_add_u32 %" #r ", %" #r ", %8\n"
#define N(r) "v_and_b32 %" #r ", %" #r ", %8\n"
#define S(r) "v_lshrrev_b32 %" #r ", 3, %" #r "\n"
#define L(r) "v_lshlrev_b32 %" #r ", 3, %" #r "\n"
#define P(r) "v_perm_b32 %" #r ", %" #r ", %8, %8\n"
#define B(r) "v_bitop3_b32 %" #r ", %" #r ", %8, %8 bitop3:0x48\n"
#define F(r) "v_bfi_b32 %" #r ", %8, %" #r ", %8\n"
#define G(r) "v_alignbit_b32 %" #r ", %8, %" #r ", 1\n"
#define X8 X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
KERNEL(k_xor, X8, 8)
KERNEL(k_add, A(0) A(1) A(2) A(3) A(4) A(5) A(6) A(7), 8)
KERNEL(k_lshr, S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7), 8)
KERNEL(k_lshl, L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7), 8)
KERNEL(k_bitop, B(0) B(1) B(2) B(3) B(4) B(5) B(6) B(7), 8)
KERNEL(k_bfi, F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7), 8)
KERNEL(k_alignbit, G(0) G(1) G(2) G(3) G(4) G(5) G(6) G(7), 8)
KERNEL(k_3x1p, X(0) X(1) X(2) P(3) X(4) X(5) X(6) P(7), 8)
KERNEL(k_7x1p, X(0) X(1) X(2) X(3) X(4) X(5) X(6) P(7), 8)
KERNEL(k_15x1p, X8 X(0) X(1) X(2) X(3) X(4) X(5) X(6) P(7), 16)
KERNEL(k_31x1p, X8 X8 X8 X(0) X(1) X(2) X(3) X(4) X(5) X(6) P(7), 32)
// the SWAR byte test's triple (bitop3, add, bitop3) x 8, then 7 shifts and a 7-deep bitop3 mux tree: what the match vector would be
KERNEL(k_swar, B(0) A(0) B(0) B(1) A(1) B(1) B(2) A(2) B(2) B(3) A(3) B(3) B(4) A(4) B(4) B(5) A(5) B(5) B(6) A(6) B(6) B(7) A(7) B(7)
               S(0) S(1) S(2) S(3) S(4) S(5) S(6) B(0) B(2) B(4) B(6) B(1) B(5) B(3), 38)
// today's match vector: 8 x (xor, perm), 7 bfi
KERNEL(k_today, X(0) P(0) X(1) P(1) X(2) P(2) X(3) P(3) X(4) P(4) X(5) P(5) X(6) P(6) X(7) P(7) F(0) F(2) F(4) F(6) F(1) F(5) F(3), 23)
// the SWAR match vector followed by the rest of today's column (10 half-rate-class, 7 full-rate instructions)
KERNEL(k_swar_col, B(0) A(0) B(0) B(1) A(1) B(1) B(2) A(2) B(2) B(3) A(3) B(3) B(4) A(4) B(4) B(5) A(5) B(5) B(6) A(6) B(6) B(7) A(7) B(7)
                   S(0) S(1) S(2) S(3) S(4) S(5) S(6) B(0) B(2) B(4) B(6) B(1) B(5) B(3)
                   P(0) P(1) G(2) G(3) F(4) P(5) G(6) F(7) P(0) P(1) X(2) X(3) B(4) B(5) B(6) X(7) X(0), 55)
KERNEL(k_today_col, X(0) P(0) X(1) P(1) X(2) P(2) X(3) P(3) X(4) P(4) X(5) P(5) X(6) P(6) X(7) P(7) F(0) F(2) F(4) F(6) F(1) F(5) F(3)
                    P(0) P(1) G(2) G(3) F(4) P(5) G(6) F(7) P(0) P(1) X(2) X(3) B(4) B(5) B(6) X(7) X(0), 40)
typedef void (*kern_t)(unsigned *);
struct Entry { const char *name; kern_t k; int n; };
int main() {
    Entry tab[] = {{"v_xor_b32 only", k_xor, k_xor_n}, {"v_add_u32 only", k_add, k_add_n}, {"v_lshrrev_b32 (const) only", k_lshr, k_lshr_n},
                   {"v_lshlrev_b32 (const) only", k_lshl, k_lshl_n}, {"v_bitop3_b32 only", k_bitop, k_bitop_n}, {"v_bfi_b32 only", k_bfi, k_bfi_n},
                   {"v_alignbit_b32 only", k_alignbit, k_alignbit_n},
                   {"3 xor : 1 perm", k_3x1p, k_3x1p_n}, {"7 xor : 1 perm", k_7x1p, k_7x1p_n}, {"15 xor : 1 perm", k_15x1p, k_15x1p_n},
                   {"31 xor : 1 perm", k_31x1p, k_31x1p_n},
                   {"match vector, SWAR form (38 instr)", k_swar, k_swar_n}, {"match vector, today (23 instr)", k_today, k_today_n},
                   {"whole column, SWAR match vector (55)", k_swar_col, k_swar_col_n}, {"whole column, today's mix (40)", k_today_col, k_today_col_n}};
    unsigned *out;
    hipMalloc(&out, 256 * 256 * 16 * 4 * sizeof(unsigned));
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("device %s  CUs %d  clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    const double ghz = p.clockRate / 1e6;
    for (int wps = 2; wps <= 4; wps += 2) {
        int blocks = p.multiProcessorCount * wps;
        printf("-- %d wave(s) per SIMD: cycles per wave-instruction per SIMD at %.2f GHz; cycles per pattern\n", wps, ghz);
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
            double cpi = ghz * 1e9 / per_simd_per_s;
            printf("%-40s %8.3f ms  %.2f cycles/instr  %7.1f cycles/pattern\n", e.name, ms, cpi, cpi * e.n);
        }
    }
    return 0;
}
