// wave.h -- the tiny "wavefront" vocabulary the kernels are written in.
//
// Kernel bodies are templates over a wave policy W.  DevWave maps every operation to one
// CDNA4 instruction (U32 = a VGPR, cross-lane = DPP wave_shr/wave_shl, LDS = ds_*).
// EmuWave (emu/emu_wave.h, host only, TESTS ONLY) runs the same body with U32 = 64 lanes
// in lock-step so the index math can be checked in a GPU-less container -- it is not a
// product fallback and is never linked into libtriple_accel_amd.so.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define TA_HD __host__ __device__
#else
#define TA_HD
#endif

namespace ta {

struct StrView {            // device view of ta_strings (include/triple_accel_amd.h)
    const uint8_t *blob;
    const uint64_t *off;    // n+1 CSR offsets or nullptr (strided form)
    uint64_t stride;
    uint64_t len;
};

struct Q128 { uint32_t x, y, z, w; };

#if defined(__HIPCC__)

struct DevWave {
    using U32 = uint32_t;
    using Bool = bool;
    using Ptr = const uint8_t *;
    using Q = Q128;

    static __device__ __forceinline__ U32 lane() {
        return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    }
    static __device__ __forceinline__ U32 splat(uint32_t x) { return x; }
    static __device__ __forceinline__ Bool bfalse() { return false; }
    static __device__ __forceinline__ U32 sel(Bool c, U32 a, U32 b) { return c ? a : b; }
    static __device__ __forceinline__ Bool land(Bool a, Bool b) { return a && b; }   // per-lane AND of two predicates (s_and_b64)
    static __device__ __forceinline__ U32 umin(U32 a, U32 b) { return a < b ? a : b; }
    static __device__ __forceinline__ U32 umin3(U32 a, U32 b, U32 c) { return umin(umin(a, b), c); }
    // signed maxima of values held in U32 (the band kernel's score form) -> v_max_i32 / v_max3_i32
    static __device__ __forceinline__ U32 imax(U32 a, U32 b) { return (int32_t)a > (int32_t)b ? a : b; }
    static __device__ __forceinline__ U32 imax3(U32 a, U32 b, U32 c) { return imax(imax(a, b), c); }
    // a where the mask bit is set, b elsewhere -> v_bfi_b32
    static __device__ __forceinline__ U32 sel_bits(U32 m, U32 a, U32 b) { return (a & m) | (b & ~m); }
    static __device__ __forceinline__ U32 udiv(U32 a, uint32_t d) { return a / d; }
    // ({hi,lo} >> 8*n)[31:0], n in 0..3  -> v_alignbyte_b32
    template <int N> static __device__ __forceinline__ U32 alignbyte(U32 hi, U32 lo) {
        return __builtin_amdgcn_alignbyte(hi, lo, N);
    }
    // value barrier: stops instcombine from re-associating across it
    static __device__ __forceinline__ U32 opaque(U32 x) { asm volatile("" : "+v"(x)); return x; }
    // a wave-uniform value the optimiser must take as new at this point (e.g. so that tests of a kernel argument's bits inside a loop stay
    // scalar compares there instead of 64-bit condition masks hoisted out of it, two SGPRs apiece)
    static __device__ __forceinline__ uint32_t opaque_s(uint32_t x) { asm volatile("" : "+s"(x)); return x; }
    // acc + byte n of x * m  (m <= 255)  -> v_dot4_u32_u8 with a one-hot multiplier
    static __device__ __forceinline__ U32 dot4_byte(U32 x, int n, uint32_t m, U32 acc) {
        return __builtin_amdgcn_udot4(x, m << (8 * n), acc, false);
    }
    // acc + sum of the four byte products of a and b -> v_dot4_u32_u8
    static __device__ __forceinline__ U32 dot4(U32 a, U32 b, U32 acc) { return __builtin_amdgcn_udot4(a, b, acc, false); }
    // acc + sum of the four SIGNED byte products -> v_dot4_i32_i8
    static __device__ __forceinline__ U32 sdot4(U32 a, U32 b, U32 acc) { return (U32)__builtin_amdgcn_sdot4((int)a, (int)b, (int)acc, false); }
    // the same without an accumulator (clamped form: the three-address encoding, no v_mov of a zero first; the sums here
    // never come near the clamp)
    static __device__ __forceinline__ U32 sdot4_first(U32 a, U32 b) { return (U32)__builtin_amdgcn_sdot4((int)a, (int)b, 0, true); }
    // 0x00 in every byte of x that equals 12, 0xFF in the others -> ONE v_perm_b32: with all-ones sources every byte
    // selector reads 0xFF (0-7: a source byte, 8-11: a replicated sign bit, >= 13: the constant 0xFF) except 12, the
    // constant 0x00.  XOR-ing one side of a byte compare with 0x0C turns this into a per-byte != test.
    static __device__ __forceinline__ U32 ne12(U32 x) { return __builtin_amdgcn_perm(0xFFFFFFFFu, 0xFFFFFFFFu, x); }
    // the low byte of x in all four bytes -> v_perm_b32
    static __device__ __forceinline__ U32 splat_byte(U32 x) { return __builtin_amdgcn_perm(x, x, 0x04040404u); }
    // sum = a + b + cin, cout = carry out -> v_add_co_u32 / v_addc_co_u32
    static __device__ __forceinline__ void addc(U32 a, U32 b, Bool cin, U32 &sum, Bool &cout) {
        unsigned int co;
        sum = __builtin_addc(a, b, cin ? 1u : 0u, &co);
        cout = co != 0u;
    }
    // acc + popcount(x) -> v_bcnt_u32_b32
    static __device__ __forceinline__ U32 bcnt(U32 x, U32 acc) { return (U32)__builtin_popcount(x) + acc; }
    // ({hi,lo} >> N)[31:0], N in 1..31 -> v_alignbit_b32
    template <int N> static __device__ __forceinline__ U32 alignbit(U32 hi, U32 lo) { return __builtin_amdgcn_alignbit(hi, lo, N); }
    // bits [off, off + width) of x -> v_bfe_u32 (kept opaque so that a following shift is not folded back into a mask)
    static __device__ __forceinline__ U32 bfe(U32 x, uint32_t off, uint32_t width) { return __builtin_amdgcn_ubfe(x, off, width); }
    // NOTE: no real instructions in inline asm in this file -- hipcc's hazard recogniser does not look inside asm
    // statements (a v_dot4 result consumed by an asm VALU instruction two slots later read a stale value on gfx950).
    // (a << s) + b -> v_lshl_add_u32
    static __device__ __forceinline__ U32 lshl_add(U32 a, uint32_t s, U32 b) { return (a << s) + b; }
    // per-lane shift amounts (< 32)
    // ({hi, lo} >> 8 (s & 3)) [31:0] with the byte count in a register (v_alignbyte_b32)
    static __device__ __forceinline__ U32 alignbyte_v(U32 hi, U32 lo, U32 s) { return __builtin_amdgcn_alignbyte(hi, lo, s); }
    static __device__ __forceinline__ U32 clz(U32 x) { return (U32)__clz((int)x); }        // leading zero bits, 32 for 0
    static __device__ __forceinline__ U32 shlv(U32 x, U32 s) { return x << s; }
    static __device__ __forceinline__ U32 shrv(U32 x, U32 s) { return x >> s; }
    // byte N of x, zero-extended (folds into the consumer as an SDWA byte select)
    static __device__ __forceinline__ U32 byte_of(U32 x, int n) { return (x >> (8 * n)) & 0xffu; }
    // (a & mask) | (b & ~mask) -> v_bfi_b32
    static __device__ __forceinline__ U32 bfi(uint32_t mask, U32 a, U32 b) { return (a & mask) | (b & ~mask); }

    // lane i <- lane i-1 (lane 0 <- fill): DPP wave_shr:1
    static __device__ __forceinline__ U32 from_lower(U32 x, U32 fill) {
        return (U32)__builtin_amdgcn_update_dpp((int)fill, (int)x, 0x138, 0xf, 0xf, false);
    }
    // lane i <- lane i+1 (lane 63 <- fill): DPP wave_shl:1
    static __device__ __forceinline__ U32 from_upper(U32 x, U32 fill) {
        return (U32)__builtin_amdgcn_update_dpp((int)fill, (int)x, 0x130, 0xf, 0xf, false);
    }
    // same moves when the edge lane's value is overridden by the caller anyway: no tied `old` operand
    static __device__ __forceinline__ U32 from_lower0(U32 x) { return (U32)__builtin_amdgcn_mov_dpp((int)x, 0x138, 0xf, 0xf, true); }
    static __device__ __forceinline__ U32 from_upper0(U32 x) { return (U32)__builtin_amdgcn_mov_dpp((int)x, 0x130, 0xf, 0xf, true); }
    static __device__ __forceinline__ U32 shfl(U32 x, U32 src_lane) {
        return (U32)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)x);
    }
    static __device__ __forceinline__ Ptr shfl_ptr(Ptr p, U32 src_lane) {
        uint64_t v = (uint64_t)p;
        uint32_t lo = shfl((uint32_t)v, src_lane), hi = shfl((uint32_t)(v >> 32), src_lane);
        return (Ptr)(((uint64_t)hi << 32) | lo);
    }
    static __device__ __forceinline__ bool any(Bool c) { return __builtin_amdgcn_ballot_w64(c) != 0ull; }
    // x of the first lane where c holds (c must hold somewhere) -> s_ff1 + v_readlane
    static __device__ __forceinline__ uint32_t first_u32(U32 x, Bool c) {
        return __builtin_amdgcn_readlane(x, (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(c)));
    }
    // wave-wide unsigned maximum in six DPP steps (no LDS round trips): row_shr 1, 2, 4, 8 leave a row's maximum in its
    // lane 15 (a lane without a source reads 0, the identity), row_bcast:15 / row_bcast:31 carry it on to lane 63
    static __device__ __forceinline__ uint32_t wave_max(U32 x) {
#define TA_DPP_MAX(ctrl, rows)                                                                        \
    {                                                                                                 \
        const U32 y = (U32)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, rows, 0xf, true);              \
        x = x > y ? x : y;                                                                            \
    }
        TA_DPP_MAX(0x111, 0xf) TA_DPP_MAX(0x112, 0xf) TA_DPP_MAX(0x114, 0xf) TA_DPP_MAX(0x118, 0xf)
        TA_DPP_MAX(0x142, 0xa) TA_DPP_MAX(0x143, 0xc)
#undef TA_DPP_MAX
        return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
    }

    static __device__ __forceinline__ void load_str(const StrView &s, U32 idx, Bool valid, Ptr &p, U32 &len) {
        if (valid) {
            if (s.off) {
                uint64_t o0 = s.off[idx], o1 = s.off[idx + 1];
                p = s.blob + o0;
                len = (uint32_t)(o1 - o0);
            } else {
                p = s.blob + (uint64_t)idx * s.stride;
                len = (uint32_t)s.len;
            }
        } else {
            p = s.blob;
            len = 0;
        }
    }
    static __device__ __forceinline__ U32 load_u32(const uint32_t *p, U32 idx, Bool valid, uint32_t dflt) {
        return valid ? p[idx] : dflt;
    }
    static __device__ __forceinline__ void store_u32(uint32_t *p, U32 idx, U32 v, Bool pred) {
        if (pred) p[idx] = v;
    }
    static __device__ __forceinline__ Ptr ptr_add(Ptr p, U32 off) { return p + off; }
    static __device__ __forceinline__ Ptr ptr_splat(const uint8_t *p) { return p; }
    // the pieces of pointer arithmetic the VLINE fetch form needs (lev_bits_body.h): low 32 bits of an address, p - off, the 128-byte line of p
    static __device__ __forceinline__ U32 ptr_lo32(Ptr p) { return (uint32_t)(uintptr_t)p; }
    static __device__ __forceinline__ Ptr ptr_sub(Ptr p, U32 off) { return p - off; }
    static __device__ __forceinline__ Ptr ptr_piece(Ptr p) { return p - ((uint32_t)(uintptr_t)p & 15u); }                 // its 16-byte piece
    static __device__ __forceinline__ Ptr ptr_line(Ptr p) { return p - ((uint32_t)(uintptr_t)p & 127u); }   // (no int -> pointer cast: the loads stay global_load)
    static __device__ __forceinline__ Q128 qzero() { return Q128{0u, 0u, 0u, 0u}; }
    // the eight 16-byte pieces of the 128-byte line at `line` (128-byte aligned) into S, rotated by the wave-uniform kappa:
    // S[j] <- piece (j + kappa) & 7; the lanes where !pred keep what they hold.  ONE asm statement -- the lane mask into EXEC, a scalar
    // jump to one of eight runs of eight global_load_dwordx4 (immediate offsets, destinations TIED to the registers S lives in), EXEC
    // restored.  Written as `if (pred) S[j] = load` under a switch, hipcc keeps a second set of 32 registers and copies the old values over
    // before every burst (191 VGPRs instead of 124).  The loads are invisible to the compiler's s_waitcnt bookkeeping: the caller waits
    // for them with wait_vm0() before it reads S.  (Called with all 64 lanes active; the mask comes from a VALU compare and is read
    // by a SALU instruction: interlocked.)
    static __device__ __forceinline__ void gload_line_keep(Q128 (&S)[8], Ptr line, Bool pred, uint32_t kappa) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 t0 = {S[0].x, S[0].y, S[0].z, S[0].w}, t1 = {S[1].x, S[1].y, S[1].z, S[1].w}, t2 = {S[2].x, S[2].y, S[2].z, S[2].w},
              t3 = {S[3].x, S[3].y, S[3].z, S[3].w}, t4 = {S[4].x, S[4].y, S[4].z, S[4].w}, t5 = {S[5].x, S[5].y, S[5].z, S[5].w},
              t6 = {S[6].x, S[6].y, S[6].z, S[6].w}, t7 = {S[7].x, S[7].y, S[7].z, S[7].w};
        const uint64_t m = __builtin_amdgcn_ballot_w64(pred);
        const uint32_t k = __builtin_amdgcn_readfirstlane(kappa & 7u);
        uint64_t save;
        uint32_t k7;
        asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                     "s_cbranch_execz .Lvl%=_end\n\t"
                     "s_getpc_b64 vcc\n"
                     ".Lvl%=_pc:\n\t"
                     "s_lshl_b32 %[k7], %[k], 7\n\t"
                     "s_add_u32 %[k7], %[k7], .Lvl%=_0-.Lvl%=_pc\n\t"
                     "s_add_u32 vcc_lo, vcc_lo, %[k7]\n\t"
                     "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
                     "s_setpc_b64 vcc\n\t"
                     ".p2align 7\n"
                     ".Lvl%=_0:\n\t"
                     "global_load_dwordx4 %[t0], %[p], off offset:0\n\t"
                     "global_load_dwordx4 %[t1], %[p], off offset:16\n\t"
                     "global_load_dwordx4 %[t2], %[p], off offset:32\n\t"
                     "global_load_dwordx4 %[t3], %[p], off offset:48\n\t"
                     "global_load_dwordx4 %[t4], %[p], off offset:64\n\t"
                     "global_load_dwordx4 %[t5], %[p], off offset:80\n\t"
                     "global_load_dwordx4 %[t6], %[p], off offset:96\n\t"
                     "global_load_dwordx4 %[t7], %[p], off offset:112\n\t"
                     "s_branch .Lvl%=_end\n\t"
                     ".p2align 7\n"
                     ".Lvl%=_1:\n\t"
                     "global_load_dwordx4 %[t0], %[p], off offset:16\n\t"
                     "global_load_dwordx4 %[t1], %[p], off offset:32\n\t"
                     "global_load_dwordx4 %[t2], %[p], off offset:48\n\t"
                     "global_load_dwordx4 %[t3], %[p], off offset:64\n\t"
                     "global_load_dwordx4 %[t4], %[p], off offset:80\n\t"
                     "global_load_dwordx4 %[t5], %[p], off offset:96\n\t"
                     "global_load_dwordx4 %[t6], %[p], off offset:112\n\t"
                     "global_load_dwordx4 %[t7], %[p], off offset:0\n\t"
                     "s_branch .Lvl%=_end\n\t"
                     ".p2align 7\n"
                     ".Lvl%=_2:\n\t"
                     "global_load_dwordx4 %[t0], %[p], off offset:32\n\t"
                     "global_load_dwordx4 %[t1], %[p], off offset:48\n\t"
                     "global_load_dwordx4 %[t2], %[p], off offset:64\n\t"
                     "global_load_dwordx4 %[t3], %[p], off offset:80\n\t"
                     "global_load_dwordx4 %[t4], %[p], off offset:96\n\t"
                     "global_load_dwordx4 %[t5], %[p], off offset:112\n\t"
                     "global_load_dwordx4 %[t6], %[p], off offset:0\n\t"
                     "global_load_dwordx4 %[t7], %[p], off offset:16\n\t"
                     "s_branch .Lvl%=_end\n\t"
                     ".p2align 7\n"
                     ".Lvl%=_3:\n\t"
                     "global_load_dwordx4 %[t0], %[p], off offset:48\n\t"
                     "global_load_dwordx4 %[t1], %[p], off offset:64\n\t"
                     "global_load_dwordx4 %[t2], %[p], off offset:80\n\t"
                     "global_load_dwordx4 %[t3], %[p], off offset:96\n\t"
                     "global_load_dwordx4 %[t4], %[p], off offset:112\n\t"
                     "global_load_dwordx4 %[t5], %[p], off offset:0\n\t"
                     "global_load_dwordx4 %[t6], %[p], off offset:16\n\t"
                     "global_load_dwordx4 %[t7], %[p], off offset:32\n\t"
                     "s_branch .Lvl%=_end\n\t"
                     ".p2align 7\n"
                     ".Lvl%=_4:\n\t"
                     "global_load_dwordx4 %[t0], %[p], off offset:64\n\t"
                     "global_load_dwordx4 %[t1], %[p], off offset:80\n\t"
                     "global_load_dwordx4 %[t2], %[p], off offset:96\n\t"
                     "global_load_dwordx4 %[t3], %[p], off offset:112\n\t"
                     "global_load_dwordx4 %[t4], %[p], off offset:0\n\t"
                     "global_load_dwordx4 %[t5], %[p], off offset:16\n\t"
                     "global_load_dwordx4 %[t6], %[p], off offset:32\n\t"
                     "global_load_dwordx4 %[t7], %[p], off offset:48\n\t"
                     "s_branch .Lvl%=_end\n\t"
                     ".p2align 7\n"
                     ".Lvl%=_5:\n\t"
                     "global_load_dwordx4 %[t0], %[p], off offset:80\n\t"
                     "global_load_dwordx4 %[t1], %[p], off offset:96\n\t"
                     "global_load_dwordx4 %[t2], %[p], off offset:112\n\t"
                     "global_load_dwordx4 %[t3], %[p], off offset:0\n\t"
                     "global_load_dwordx4 %[t4], %[p], off offset:16\n\t"
                     "global_load_dwordx4 %[t5], %[p], off offset:32\n\t"
                     "global_load_dwordx4 %[t6], %[p], off offset:48\n\t"
                     "global_load_dwordx4 %[t7], %[p], off offset:64\n\t"
                     "s_branch .Lvl%=_end\n\t"
                     ".p2align 7\n"
                     ".Lvl%=_6:\n\t"
                     "global_load_dwordx4 %[t0], %[p], off offset:96\n\t"
                     "global_load_dwordx4 %[t1], %[p], off offset:112\n\t"
                     "global_load_dwordx4 %[t2], %[p], off offset:0\n\t"
                     "global_load_dwordx4 %[t3], %[p], off offset:16\n\t"
                     "global_load_dwordx4 %[t4], %[p], off offset:32\n\t"
                     "global_load_dwordx4 %[t5], %[p], off offset:48\n\t"
                     "global_load_dwordx4 %[t6], %[p], off offset:64\n\t"
                     "global_load_dwordx4 %[t7], %[p], off offset:80\n\t"
                     "s_branch .Lvl%=_end\n\t"
                     ".p2align 7\n"
                     ".Lvl%=_7:\n\t"
                     "global_load_dwordx4 %[t0], %[p], off offset:112\n\t"
                     "global_load_dwordx4 %[t1], %[p], off offset:0\n\t"
                     "global_load_dwordx4 %[t2], %[p], off offset:16\n\t"
                     "global_load_dwordx4 %[t3], %[p], off offset:32\n\t"
                     "global_load_dwordx4 %[t4], %[p], off offset:48\n\t"
                     "global_load_dwordx4 %[t5], %[p], off offset:64\n\t"
                     "global_load_dwordx4 %[t6], %[p], off offset:80\n\t"
                     "global_load_dwordx4 %[t7], %[p], off offset:96\n\t"
                     "\n.Lvl%=_end:\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [t3] "+v"(t3), [t4] "+v"(t4), [t5] "+v"(t5), [t6] "+v"(t6), [t7] "+v"(t7),
                       [sv] "=&s"(save), [k7] "=&s"(k7)
                     : [p] "v"(line), [m] "s"(m), [k] "s"(k)
                     : "memory", "scc", "vcc");
        S[0] = Q128{t0.x, t0.y, t0.z, t0.w}; S[1] = Q128{t1.x, t1.y, t1.z, t1.w}; S[2] = Q128{t2.x, t2.y, t2.z, t2.w}; S[3] = Q128{t3.x, t3.y, t3.z, t3.w};
        S[4] = Q128{t4.x, t4.y, t4.z, t4.w}; S[5] = Q128{t5.x, t5.y, t5.z, t5.w}; S[6] = Q128{t6.x, t6.y, t6.z, t6.w}; S[7] = Q128{t7.x, t7.y, t7.z, t7.w};
    }
    // marks a switch case as its own (an empty asm statement whose text holds N): keeps the optimiser from merging cases that differ only
    // in a register array's index into one body with a run-time index
    template <int N> static __device__ __forceinline__ void case_tag() { asm volatile("; case %0" : : "n"(N)); }
    // every global load of this wavefront has delivered (the loads of gload_line_keep are not tracked by the compiler)
    static __device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    static __device__ __forceinline__ Ptr sel_ptr(Bool c, Ptr a, Ptr b) { return c ? a : b; }
    // 16 bytes from an arbitrarily aligned global address (zeros where !pred)
    static __device__ __forceinline__ Q128 gload16(Ptr p, Bool pred) {
        Q128 q = {0u, 0u, 0u, 0u};
        if (pred) {
            typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(1)));
            u32x4u v = *(const u32x4u *)p;
            q.x = v.x; q.y = v.y; q.z = v.z; q.w = v.w;
        }
        return q;
    }
    // 16 bytes from an arbitrarily aligned global address, EVERY lane (all addresses readable): no branch and no zero-fill, so the load's
    // destination can be a register that lives across iterations -- a prefetch stays in flight until the value is read
    static __device__ __forceinline__ Q128 gload16_all(Ptr p) {
        typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(1)));
        const u32x4u v = *(const u32x4u *)p;
        Q128 q = {v.x, v.y, v.z, v.w};
        return q;
    }
    static __device__ __forceinline__ Q128 qkeep(Q128 q, Bool keep) { Q128 r = {keep ? q.x : 0u, keep ? q.y : 0u, keep ? q.z : 0u, keep ? q.w : 0u}; return r; }
    // same, streamed: the line is consumed whole by this one load, do not keep it in L2 (nt)
    static __device__ __forceinline__ Q128 gload16_nt(Ptr p, Bool pred) {
        Q128 q = {0u, 0u, 0u, 0u};
        if (pred) {
            typedef uint32_t u32u __attribute__((aligned(1)));
            const u32u *w = (const u32u *)p;
            q.x = __builtin_nontemporal_load(w); q.y = __builtin_nontemporal_load(w + 1);
            q.z = __builtin_nontemporal_load(w + 2); q.w = __builtin_nontemporal_load(w + 3);
        }
        return q;
    }
    static __device__ __forceinline__ U32 qword(const Q128 &q, int i) { return i == 0 ? q.x : i == 1 ? q.y : i == 2 ? q.z : q.w; }
    static __device__ __forceinline__ Q128 qxor_v(Q128 q, U32 c) { q.x ^= c; q.y ^= c; q.z ^= c; q.w ^= c; return q; }
    static __device__ __forceinline__ Q128 qxor(Q128 q, uint32_t c) { q.x ^= c; q.y ^= c; q.z ^= c; q.w ^= c; return q; }
    static __device__ __forceinline__ void lds_store16(uint8_t *lds, U32 off, Q128 q, Bool pred) {
        if (pred) {   // 4-byte aligned only (slot stride is an odd number of dwords)
            uint32_t *d = (uint32_t *)(lds + off);
            d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
        }
    }
    static __device__ __forceinline__ U32 lds_u8(const uint8_t *lds, U32 off) { return lds[off]; }
    // lane-private dwords in LDS (4-byte aligned offsets)
    static __device__ __forceinline__ U32 lds_read32(const uint8_t *lds, U32 off) { return *(const uint32_t *)(lds + off); }
    // a dword at ANY byte offset: one ds_read_b32 (gfx950 DS accesses need no alignment)
    static __device__ __forceinline__ U32 lds_read32u(const uint8_t *lds, U32 off) {
        typedef uint32_t __attribute__((aligned(1))) u32u;
        return *(const u32u *)(lds + off);
    }
    // byte N of x in all four bytes -> v_perm_b32
    template <int N>
    static __device__ __forceinline__ U32 splat_byte_n(U32 x) { return __builtin_amdgcn_perm(x, x, 0x04040404u + 0x01010101u * (uint32_t)N); }
    // (lo >> 8) | (byte N of hi << 24): the dword slides one byte down, byte N of `hi` enters on top -> one v_perm_b32
    template <int N>
    static __device__ __forceinline__ U32 slide_in_byte(U32 hi, U32 lo) { return __builtin_amdgcn_perm(hi, lo, 0x00030201u | ((4u + (uint32_t)N) << 24)); }
    // bytes picked by the constant selector SEL out of {hi: 4..7, lo: 0..3} (0x0C: the constant 0x00) -> v_perm_b32
    template <uint32_t SEL>
    static __device__ __forceinline__ U32 perm(U32 hi, U32 lo) { return __builtin_amdgcn_perm(hi, lo, SEL); }
    // the same with a per-lane selector (bytes 0..3: lo, 4..7: hi) -> v_perm_b32
    static __device__ __forceinline__ U32 perm_sel(U32 hi, U32 lo, U32 sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
    // x >> s and ({hi,lo} >> s)[31:0] with a wave-uniform run-time s (< 32): v_lshrrev_b32 / v_alignbit_b32 with a scalar shift
    static __device__ __forceinline__ U32 shr_u(U32 x, uint32_t s) { return x >> s; }
    static __device__ __forceinline__ U32 alignbit_rt(U32 hi, U32 lo, uint32_t s) { return __builtin_amdgcn_alignbit(hi, lo, s); }
    // two consecutive lane-private dwords in LDS (4-byte aligned offset) -> ds_read2_b32
    static __device__ __forceinline__ void lds_read64(const uint8_t *lds, U32 off, U32 &lo, U32 &hi) {
        const uint32_t *p = (const uint32_t *)(lds + off);
        lo = p[0]; hi = p[1];
    }
    static __device__ __forceinline__ void lds_write16(uint8_t *lds, U32 off, U32 v) { *(uint16_t *)(lds + off) = (uint16_t)v; }
    // list[(*counter)++] = v in the lanes where pred holds (rare paths: one global atomic per such lane)
    static __device__ __forceinline__ void append_u32(uint32_t *list, uint32_t *counter, U32 v, Bool pred) {
        if (pred) list[atomicAdd(counter, 1u)] = v;
    }
    // (a & m) | c -> v_and_or_b32
    // (hipcc would rather emit v_and per term and join three terms per v_or3: 11 instructions for 8 terms instead of 8)
    static __device__ __forceinline__ U32 and_or(U32 a, uint32_t m, U32 c) {
        U32 d;
        asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(m), "v"(c));
        return d;
    }
    // (a & m) | (b & ~m) with a constant mask -> v_bfi_b32, as written (hipcc would re-associate a tree of these into v_and + v_or3)
    static __device__ __forceinline__ U32 bfi_k(uint32_t m, U32 a, U32 b) {
        U32 d;
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "s"(m), "v"(a), "v"(b));
        return d;
    }
    // byte N of x == byte N of y (hipcc: v_bitop3 (x ^ y) & mask, v_cmp_eq 0)
    // A carry that stays a wavefront mask in an SGPR pair (hipcc turns a `bool` that crosses an asm statement into a VGPR and back):
    // sum = a + b, returns the carry-out mask -> v_add_co_u32
    using Mask = uint64_t;
    static __device__ __forceinline__ Mask add_carry_mask(U32 a, U32 b, U32 &sum) {
        Mask cm;
        asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(sum), "=s"(cm) : "v"(a), "v"(b));
        return cm;
    }
    // a where the lane's bit of m is set, else b.  The mask goes through VCC by an s_mov: a VALU instruction that reads an SGPR pair a VALU
    // instruction wrote needs wait states the compiler cannot count across asm statements; SALU reads and writes are interlocked.
    static __device__ __forceinline__ U32 sel_mask(Mask m, U32 a, U32 b) {
        U32 d;
        asm("s_mov_b64 vcc, %3\n\tv_cndmask_b32_e32 %0, %2, %1, vcc" : "=v"(d) : "v"(a), "v"(b), "s"(m) : "vcc");
        return d;
    }
    // 1 where byte N of x == byte N of y or the lane's bit of cm is set, else 0 -> v_cmp_eq_u32_sdwa (the byte selects do the
    // extraction), s_or_b64, v_cndmask_b32: two VALU instructions (hipcc's own sequence for byte_eq() | c is v_bitop3, v_cmp_eq,
    // s_or, v_cndmask: three)
    template <int N>
    static __device__ __forceinline__ U32 byte_eq_or(U32 x, U32 y, Mask cm) {
        static_assert(N >= 0 && N < 4, "byte index");
        U32 d;
        if constexpr (N == 0) asm("v_cmp_eq_u32_sdwa vcc, %1, %2 src0_sel:BYTE_0 src1_sel:BYTE_0\n\ts_or_b64 vcc, vcc, %3\n\tv_cndmask_b32_e64 %0, 0, 1, vcc" : "=v"(d) : "v"(x), "v"(y), "s"(cm) : "vcc", "scc");      // (s_or_b64 sets SCC)
        if constexpr (N == 1) asm("v_cmp_eq_u32_sdwa vcc, %1, %2 src0_sel:BYTE_1 src1_sel:BYTE_1\n\ts_or_b64 vcc, vcc, %3\n\tv_cndmask_b32_e64 %0, 0, 1, vcc" : "=v"(d) : "v"(x), "v"(y), "s"(cm) : "vcc", "scc");      // (s_or_b64 sets SCC)
        if constexpr (N == 2) asm("v_cmp_eq_u32_sdwa vcc, %1, %2 src0_sel:BYTE_2 src1_sel:BYTE_2\n\ts_or_b64 vcc, vcc, %3\n\tv_cndmask_b32_e64 %0, 0, 1, vcc" : "=v"(d) : "v"(x), "v"(y), "s"(cm) : "vcc", "scc");      // (s_or_b64 sets SCC)
        if constexpr (N == 3) asm("v_cmp_eq_u32_sdwa vcc, %1, %2 src0_sel:BYTE_3 src1_sel:BYTE_3\n\ts_or_b64 vcc, vcc, %3\n\tv_cndmask_b32_e64 %0, 0, 1, vcc" : "=v"(d) : "v"(x), "v"(y), "s"(cm) : "vcc", "scc");      // (s_or_b64 sets SCC)
        return d;
    }
    template <int N>
    static __device__ __forceinline__ Bool byte_eq(U32 x, U32 y) { return ((x >> (8 * N)) & 0xFFu) == ((y >> (8 * N)) & 0xFFu); }
    static __device__ __forceinline__ void lds_write32(uint8_t *lds, U32 off, U32 v) { *(uint32_t *)(lds + off) = v; }
    static __device__ __forceinline__ void lds_write32p(uint8_t *lds, U32 off, U32 v, Bool pred) { if (pred) *(uint32_t *)(lds + off) = v; }
    static __device__ __forceinline__ void lds_or32(uint8_t *lds, U32 off, U32 v, Bool pred) {   // ds_or_b32, no return
        if (pred) (void)__hip_atomic_fetch_or((uint32_t *)(lds + off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    // this wave's global stores become visible to its own later global loads from other lanes (L1 invalidate)
    static __device__ __forceinline__ void mem_fence() { __threadfence(); }
    // value of x in lane l (l wave-uniform) -> v_readlane_b32
    static __device__ __forceinline__ uint32_t readlane(U32 x, uint32_t l) { return __builtin_amdgcn_readlane(x, l); }
    // x with lane l (wave-uniform) replaced by the wave-uniform value v (v_cmp + v_cndmask; hipcc has no writelane builtin)
    static __device__ __forceinline__ U32 writelane(U32 x, uint32_t v, uint32_t l) { return lane() == l ? v : x; }
    static __device__ __forceinline__ U32 gload_u8(Ptr p, Bool pred) { return pred ? (U32)*p : 0u; }
    static __device__ __forceinline__ uint32_t wave_sum(U32 x) {
        for (int m = 32; m >= 1; m >>= 1) x += shfl(x, lane() ^ (uint32_t)m);
        return __builtin_amdgcn_readfirstlane(x);
    }
    // this wave's LDS writes become visible to its own later LDS reads (same-wave DS ops are
    // ordered in hardware; this only stops the compiler from moving them)
    static __device__ __forceinline__ void lds_wave_sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
};

#endif  // __HIPCC__

}  // namespace ta
