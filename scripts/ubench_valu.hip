// Micro-benchmark: issue throughput of the VALU instructions the band kernel is made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O2 scripts/ubench_valu.hip -o scripts/ubench_valu ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define REP8(x) x x x x x x x x
#define ITER 512

#define KERNEL(NAME, ASM)                                                                   \
    __global__ __launch_bounds__(256) void NAME(unsigned *out) {                            \
        unsigned r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4,      \
                 r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7, s = blockIdx.x | 1;                 \
        for (int i = 0; i < ITER; i++) {                                                    \
            asm volatile(REP8(ASM) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4),      \
                         "+v"(r5), "+v"(r6), "+v"(r7) : "v"(s));                             \
        }                                                                                   \
        out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;        \
    }

// each ASM string = 8 independent instructions (one per register); REP8 -> 64 per loop trip
KERNEL(k_add, "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n")
KERNEL(k_min3, "v_min3_u32 %0, %0, %8, %1\n v_min3_u32 %1, %1, %8, %2\n v_min3_u32 %2, %2, %8, %3\n v_min3_u32 %3, %3, %8, %4\n v_min3_u32 %4, %4, %8, %5\n v_min3_u32 %5, %5, %8, %6\n v_min3_u32 %6, %6, %8, %7\n v_min3_u32 %7, %7, %8, %0\n")
KERNEL(k_min, "v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_min_u32 %3, %3, %8\n v_min_u32 %4, %4, %8\n v_min_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_min_u32 %7, %7, %8\n")
KERNEL(k_dot4, "v_dot4_u32_u8 %0, %8, %8, %0\n v_dot4_u32_u8 %1, %8, %8, %1\n v_dot4_u32_u8 %2, %8, %8, %2\n v_dot4_u32_u8 %3, %8, %8, %3\n v_dot4_u32_u8 %4, %8, %8, %4\n v_dot4_u32_u8 %5, %8, %8, %5\n v_dot4_u32_u8 %6, %8, %8, %6\n v_dot4_u32_u8 %7, %8, %8, %7\n")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %8, %1 bitop3:0x48\n v_bitop3_b32 %1, %1, %8, %2 bitop3:0x48\n v_bitop3_b32 %2, %2, %8, %3 bitop3:0x48\n v_bitop3_b32 %3, %3, %8, %4 bitop3:0x48\n v_bitop3_b32 %4, %4, %8, %5 bitop3:0x48\n v_bitop3_b32 %5, %5, %8, %6 bitop3:0x48\n v_bitop3_b32 %6, %6, %8, %7 bitop3:0x48\n v_bitop3_b32 %7, %7, %8, %0 bitop3:0x48\n")
KERNEL(k_alignbyte, "v_alignbyte_b32 %0, %0, %8, 1\n v_alignbyte_b32 %1, %1, %8, 1\n v_alignbyte_b32 %2, %2, %8, 1\n v_alignbyte_b32 %3, %3, %8, 1\n v_alignbyte_b32 %4, %4, %8, 1\n v_alignbyte_b32 %5, %5, %8, 1\n v_alignbyte_b32 %6, %6, %8, 1\n v_alignbyte_b32 %7, %7, %8, 1\n")
KERNEL(k_dpp, "v_mov_b32_dpp %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_rowdpp, "v_mov_b32_dpp %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_sdwa, "v_add_u32_sdwa %0, %8, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %1, %8, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %2, %8, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %3, %8, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %4, %8, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %5, %8, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %6, %8, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %7, %8, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n")
KERNEL(k_pkadd, "v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n v_pk_add_u16 %4, %4, %8\n v_pk_add_u16 %5, %5, %8\n v_pk_add_u16 %6, %6, %8\n v_pk_add_u16 %7, %7, %8\n")
KERNEL(k_pkmin, "v_pk_min_u16 %0, %0, %8\n v_pk_min_u16 %1, %1, %8\n v_pk_min_u16 %2, %2, %8\n v_pk_min_u16 %3, %3, %8\n v_pk_min_u16 %4, %4, %8\n v_pk_min_u16 %5, %5, %8\n v_pk_min_u16 %6, %6, %8\n v_pk_min_u16 %7, %7, %8\n")
KERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8\n")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
KERNEL(k_lshr, "v_lshrrev_b32 %0, 7, %0\n v_lshrrev_b32 %1, 7, %1\n v_lshrrev_b32 %2, 7, %2\n v_lshrrev_b32 %3, 7, %3\n v_lshrrev_b32 %4, 7, %4\n v_lshrrev_b32 %5, 7, %5\n v_lshrrev_b32 %6, 7, %6\n v_lshrrev_b32 %7, 7, %7\n")
// dependent chain: one register, 8 back-to-back dependent adds
KERNEL(k_add_dep, "v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n")
KERNEL(k_min3_dep, "v_min3_u32 %0, %0, %8, %1\n v_min3_u32 %0, %0, %8, %1\n v_min3_u32 %0, %0, %8, %1\n v_min3_u32 %0, %0, %8, %1\n v_min3_u32 %0, %0, %8, %1\n v_min3_u32 %0, %0, %8, %1\n v_min3_u32 %0, %0, %8, %1\n v_min3_u32 %0, %0, %8, %1\n")

typedef void (*kern_t)(unsigned *);
struct Entry { const char *name; kern_t k; };

int main() {
    Entry tab[] = {{"v_add_u32", k_add}, {"v_min_u32", k_min}, {"v_min3_u32", k_min3}, {"v_dot4_u32_u8", k_dot4},
                   {"v_bitop3_b32", k_bitop3}, {"v_alignbyte_b32", k_alignbyte}, {"v_mov_dpp wave_shr", k_dpp},
                   {"v_mov_dpp row_shr", k_rowdpp}, {"v_add_u32_sdwa", k_sdwa}, {"v_pk_add_u16", k_pkadd},
                   {"v_pk_min_u16", k_pkmin}, {"v_mul_u32_u24", k_mul24}, {"v_cndmask_b32", k_cndmask},
                   {"v_lshrrev_b32", k_lshr}, {"v_add_u32 (dependent)", k_add_dep}, {"v_min3_u32 (dependent)", k_min3_dep}};
    unsigned *out;
    hipMalloc(&out, 256 * 256 * 16 * 4 * sizeof(unsigned));
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("device %s  CUs %d  clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    for (int wps = 1; wps <= 8; wps *= 2) {           // waves per SIMD
        int blocks = p.multiProcessorCount * wps;      // 256-thread blocks = 4 waves = 1 wave/SIMD each
        printf("-- %d wave(s) per SIMD\n", wps);
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
            double waveinstr = 5.0 * blocks * 4 * (double)ITER * 64;     // wave-instructions issued
            double per_simd_per_s = waveinstr / (ms * 1e-3) / (p.multiProcessorCount * 4);
            printf("%-26s %8.3f ms  %7.2f G wave-instr/s/chip  %.2f cycles/instr/SIMD @2.4GHz\n", e.name, ms,
                   waveinstr / (ms * 1e-3) / 1e9, 2.4e9 / per_simd_per_s);
        }
    }
    return 0;
}
