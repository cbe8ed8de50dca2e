// emu_wave.h -- 64-lane lock-step host emulation of the wave policy (triple_accel_amd/csrc/wave.h).
//
// TESTS ONLY.  Lets the kernel bodies (lev_band_body.h, ...) run in a GPU-less container so
// their index math can be checked against the oracle.  Never linked into the product library,
// never a fallback: the product's compute entry points fail with TA_ERR_HIP without a GPU.
#pragma once
#include <stdint.h>
#include <string.h>

#include "wave.h"

namespace ta {

struct VB {
    bool v[64];
};
struct V32 {
    uint32_t v[64];
    V32() {}
    V32(uint32_t x) { for (int i = 0; i < 64; i++) v[i] = x; }
};
struct VP {
    const uint8_t *v[64];
};

#define TA_EMU_BIN(op)                                                                    \
    static inline V32 operator op(const V32 &a, const V32 &b) {                           \
        V32 r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] op b.v[i]; return r; }
TA_EMU_BIN(+) TA_EMU_BIN(-) TA_EMU_BIN(*) TA_EMU_BIN(^) TA_EMU_BIN(|) TA_EMU_BIN(&)
#undef TA_EMU_BIN
static inline V32 operator~(const V32 &a) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = ~a.v[i]; return r; }
static inline V32 operator>>(const V32 &a, int s) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] >> s; return r; }
static inline V32 operator<<(const V32 &a, int s) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] << s; return r; }
#define TA_EMU_CMP(op)                                                                    \
    static inline VB operator op(const V32 &a, const V32 &b) {                            \
        VB r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] op b.v[i]; return r; }
TA_EMU_CMP(==) TA_EMU_CMP(!=) TA_EMU_CMP(<) TA_EMU_CMP(<=) TA_EMU_CMP(>) TA_EMU_CMP(>=)
#undef TA_EMU_CMP
static inline VB operator&(const VB &a, const VB &b) { VB r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] && b.v[i]; return r; }
static inline VB operator|(const VB &a, const VB &b) { VB r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] || b.v[i]; return r; }
static inline VB operator!(const VB &a) { VB r; for (int i = 0; i < 64; i++) r.v[i] = !a.v[i]; return r; }

struct EmuWave {
    using U32 = V32;
    using Bool = VB;
    using Ptr = VP;

    static U32 lane() { V32 r; for (int i = 0; i < 64; i++) r.v[i] = i; return r; }
    static U32 splat(uint32_t x) { return V32(x); }
    static Bool bfalse() { VB r; for (int i = 0; i < 64; i++) r.v[i] = false; return r; }
    static U32 sel(const Bool &c, const U32 &a, const U32 &b) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = c.v[i] ? a.v[i] : b.v[i]; return r; }
    static Bool land(const Bool &a, const Bool &b) { return a & b; }
    static U32 umin(const U32 &a, const U32 &b) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] < b.v[i] ? a.v[i] : b.v[i]; return r; }
    static U32 umin3(const U32 &a, const U32 &b, const U32 &c) { return umin(umin(a, b), c); }
    static U32 imax(const U32 &a, const U32 &b) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = (int32_t)a.v[i] > (int32_t)b.v[i] ? a.v[i] : b.v[i]; return r; }
    static U32 imax3(const U32 &a, const U32 &b, const U32 &c) { return imax(imax(a, b), c); }
    static U32 sel_bits(const U32 &m, const U32 &a, const U32 &b) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = (a.v[i] & m.v[i]) | (b.v[i] & ~m.v[i]); return r; }
    static U32 udiv(const U32 &a, uint32_t d) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] / d; return r; }
    template <int N> static U32 alignbyte(const U32 &hi, const U32 &lo) {
        V32 r;
        for (int i = 0; i < 64; i++) r.v[i] = (uint32_t)(((((uint64_t)hi.v[i]) << 32) | lo.v[i]) >> (8 * N));
        return r;
    }
    static U32 opaque(const U32 &x) { return x; }
    static uint32_t opaque_s(uint32_t x) { return x; }
    static U32 dot4_byte(const U32 &x, int n, uint32_t m, const U32 &acc) {
        V32 r; for (int i = 0; i < 64; i++) r.v[i] = acc.v[i] + ((x.v[i] >> (8 * n)) & 0xffu) * (m & 0xffu); return r;
    }
    static U32 dot4(const U32 &a, const U32 &b, const U32 &acc) {
        V32 r;
        for (int i = 0; i < 64; i++) {
            uint32_t t = acc.v[i];
            for (int k = 0; k < 4; k++) t += ((a.v[i] >> (8 * k)) & 0xffu) * ((b.v[i] >> (8 * k)) & 0xffu);
            r.v[i] = t;
        }
        return r;
    }
    static U32 sdot4(const U32 &a, const U32 &b, const U32 &acc) {
        V32 r;
        for (int i = 0; i < 64; i++) {
            int32_t t = (int32_t)acc.v[i];
            for (int k = 0; k < 4; k++) t += (int32_t)(int8_t)(a.v[i] >> (8 * k)) * (int32_t)(int8_t)(b.v[i] >> (8 * k));
            r.v[i] = (uint32_t)t;
        }
        return r;
    }
    static U32 sdot4_first(const U32 &a, const U32 &b) { V32 z; for (int i = 0; i < 64; i++) z.v[i] = 0; return sdot4(a, b, z); }
    static U32 ne12(const U32 &x) {
        V32 r;
        for (int i = 0; i < 64; i++) {
            uint32_t f = 0;
            for (int k = 0; k < 4; k++) if (((x.v[i] >> (8 * k)) & 0xffu) != 12u) f |= 0xffu << (8 * k);
            r.v[i] = f;
        }
        return r;
    }
    static U32 splat_byte(const U32 &x) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = (x.v[i] & 0xffu) * 0x01010101u; return r; }
    static void addc(const U32 &a, const U32 &b, const Bool &cin, U32 &sum, Bool &cout) {
        for (int i = 0; i < 64; i++) {
            uint64_t t = (uint64_t)a.v[i] + b.v[i] + (cin.v[i] ? 1u : 0u);
            sum.v[i] = (uint32_t)t; cout.v[i] = (t >> 32) != 0;
        }
    }
    static U32 bcnt(const U32 &x, const U32 &acc) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = acc.v[i] + (uint32_t)__builtin_popcount(x.v[i]); return r; }
    template <int N> static U32 alignbit(const U32 &hi, const U32 &lo) {
        V32 r; for (int i = 0; i < 64; i++) r.v[i] = (uint32_t)(((((uint64_t)hi.v[i]) << 32) | lo.v[i]) >> N); return r;
    }
    static U32 bfe(const U32 &x, uint32_t off, uint32_t width) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = (x.v[i] >> off) & ((1u << width) - 1u); return r; }
    static U32 lshl_add(const U32 &a, uint32_t s, const U32 &b) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = (a.v[i] << s) + b.v[i]; return r; }
    static U32 alignbyte_v(const U32 &hi, const U32 &lo, const U32 &s) {
        V32 r; for (int i = 0; i < 64; i++) r.v[i] = (uint32_t)(((((uint64_t)hi.v[i]) << 32) | lo.v[i]) >> (8 * (s.v[i] & 3u))); return r; }
    static U32 clz(const U32 &x) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = x.v[i] ? (uint32_t)__builtin_clz(x.v[i]) : 32u; return r; }
    static U32 shlv(const U32 &x, const U32 &s) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = x.v[i] << (s.v[i] & 31); return r; }
    static U32 shrv(const U32 &x, const U32 &s) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = x.v[i] >> (s.v[i] & 31); return r; }
    static U32 byte_of(const U32 &x, int n) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = (x.v[i] >> (8 * n)) & 0xffu; return r; }
    static U32 bfi(uint32_t mask, const U32 &a, const U32 &b) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = (a.v[i] & mask) | (b.v[i] & ~mask); return r; }
    static U32 from_lower(const U32 &x, const U32 &fill) { V32 r; r.v[0] = fill.v[0]; for (int i = 1; i < 64; i++) r.v[i] = x.v[i - 1]; return r; }
    static U32 from_upper(const U32 &x, const U32 &fill) { V32 r; r.v[63] = fill.v[63]; for (int i = 0; i < 63; i++) r.v[i] = x.v[i + 1]; return r; }
    static U32 from_lower0(const U32 &x) { return from_lower(x, V32(0xDEADBEEFu)); }   // edge value must not matter
    static U32 from_upper0(const U32 &x) { return from_upper(x, V32(0xDEADBEEFu)); }
    static U32 shfl(const U32 &x, const U32 &src) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = x.v[src.v[i] & 63]; return r; }
    static Ptr shfl_ptr(const Ptr &p, const U32 &src) { VP r; for (int i = 0; i < 64; i++) r.v[i] = p.v[src.v[i] & 63]; return r; }
    static bool any(const Bool &c) { for (int i = 0; i < 64; i++) if (c.v[i]) return true; return false; }
    static uint32_t first_u32(const U32 &x, const Bool &c) { for (int i = 0; i < 64; i++) if (c.v[i]) return x.v[i]; return 0; }
    static uint32_t wave_max(const U32 &x) { uint32_t m = 0; for (int i = 0; i < 64; i++) if (x.v[i] > m) m = x.v[i]; return m; }

    static void load_str(const StrView &s, const U32 &idx, const Bool &valid, Ptr &p, U32 &len) {
        for (int i = 0; i < 64; i++) {
            if (valid.v[i]) {
                if (s.off) {
                    p.v[i] = s.blob + s.off[idx.v[i]];
                    len.v[i] = (uint32_t)(s.off[idx.v[i] + 1] - s.off[idx.v[i]]);
                } else {
                    p.v[i] = s.blob + (uint64_t)idx.v[i] * s.stride;
                    len.v[i] = (uint32_t)s.len;
                }
            } else {
                p.v[i] = s.blob;
                len.v[i] = 0;
            }
        }
    }
    static U32 load_u32(const uint32_t *p, const U32 &idx, const Bool &valid, uint32_t dflt) {
        V32 r; for (int i = 0; i < 64; i++) r.v[i] = valid.v[i] ? p[idx.v[i]] : dflt; return r;
    }
    static void store_u32(uint32_t *p, const U32 &idx, const U32 &v, const Bool &pred) {
        for (int i = 0; i < 64; i++) if (pred.v[i]) p[idx.v[i]] = v.v[i];
    }
    static Ptr ptr_splat(const uint8_t *p) { VP r; for (int i = 0; i < 64; i++) r.v[i] = p; return r; }
    static Ptr ptr_add(const Ptr &p, const U32 &off) { VP r; for (int i = 0; i < 64; i++) r.v[i] = p.v[i] + off.v[i]; return r; }
    static Ptr sel_ptr(const Bool &c, const Ptr &a, const Ptr &b) { VP r; for (int i = 0; i < 64; i++) r.v[i] = c.v[i] ? a.v[i] : b.v[i]; return r; }

    struct Q128V { Q128 v[64]; };
    using Q = Q128V;
    static Q128V gload16(const Ptr &p, const Bool &pred) {
        Q128V q;
        for (int i = 0; i < 64; i++) {
            q.v[i] = Q128{0, 0, 0, 0};
            if (pred.v[i]) {
                if (n_ranges()) { uint8_t tmp[16]; for (int b = 0; b < 16; b++) tmp[b] = safe_byte(p.v[i] + b); memcpy(&q.v[i], tmp, 16); }
                else memcpy(&q.v[i], p.v[i], 16);
            }
        }
        return q;
    }
    static Q128V gload16_nt(const Ptr &p, const Bool &pred) { return gload16(p, pred); }
    static Q128V gload16_all(const Ptr &p) { Bool all; for (int l = 0; l < 64; l++) all.v[l] = true; return gload16(p, all); }
    static Q128V qkeep(Q128V q, const Bool &keep) { for (int l = 0; l < 64; l++) if (!keep.v[l]) q.v[l] = Q128{0u, 0u, 0u, 0u}; return q; }
    // ---- VLINE fetch form: whole 128-byte lines are read, also the bytes of a line that lie outside the string (on the device: the
    // same line, mapped memory).  The drivers register the blobs; a byte outside every registered range reads as 0xA5.
    struct Range { const uint8_t *lo, *hi; };
    static Range *ranges() { static Range r[8]; return r; }
    static int &n_ranges() { static int n = 0; return n; }
    static void clear_ranges() { n_ranges() = 0; }
    static void add_range(const uint8_t *lo, uint64_t bytes) { if (n_ranges() < 8) ranges()[n_ranges()++] = Range{lo, lo + bytes}; }
    static uint8_t safe_byte(const uint8_t *p) {
        if (n_ranges() == 0) return *p;                    // nothing registered: unchecked
        for (int i = 0; i < n_ranges(); i++) if (p >= ranges()[i].lo && p < ranges()[i].hi) return *p;
        return 0xA5;
    }
    static U32 ptr_lo32(const Ptr &p) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = (uint32_t)(uintptr_t)p.v[i]; return r; }
    static Ptr ptr_sub(const Ptr &p, const U32 &off) { VP r; for (int i = 0; i < 64; i++) r.v[i] = p.v[i] - off.v[i]; return r; }
    static Ptr ptr_piece(const Ptr &p) { VP r; for (int i = 0; i < 64; i++) r.v[i] = (const uint8_t *)((uintptr_t)p.v[i] & ~(uintptr_t)15); return r; }
    static Ptr ptr_line(const Ptr &p) { VP r; for (int i = 0; i < 64; i++) r.v[i] = (const uint8_t *)((uintptr_t)p.v[i] & ~(uintptr_t)127); return r; }
    static void wait_vm0() {}
    template <int N> static void case_tag() {}
    static Q128V qzero() { Q128V q; for (int i = 0; i < 64; i++) q.v[i] = Q128{0, 0, 0, 0}; return q; }
    static void gload_line_keep(Q128V (&S)[8], const Ptr &line, const Bool &pred, uint32_t KAPPA) {
        for (int i = 0; i < 64; i++) {
            if (!pred.v[i]) continue;
            for (int j = 0; j < 8; j++) {
                uint8_t tmp[16];
                const uint8_t *src = line.v[i] + 16 * ((j + KAPPA) & 7);
                for (int b = 0; b < 16; b++) tmp[b] = safe_byte(src + b);
                memcpy(&S[j].v[i], tmp, 16);
            }
        }
    }
    static U32 qword(const Q128V &q, int i) {
        V32 r;
        for (int l = 0; l < 64; l++) r.v[l] = i == 0 ? q.v[l].x : i == 1 ? q.v[l].y : i == 2 ? q.v[l].z : q.v[l].w;
        return r;
    }
    static Q128V qxor_v(Q128V q, const U32 &c) { for (int l = 0; l < 64; l++) { q.v[l].x ^= c.v[l]; q.v[l].y ^= c.v[l]; q.v[l].z ^= c.v[l]; q.v[l].w ^= c.v[l]; } return q; }
    static Q128V qxor(Q128V q, uint32_t c) { for (int l = 0; l < 64; l++) { q.v[l].x ^= c; q.v[l].y ^= c; q.v[l].z ^= c; q.v[l].w ^= c; } return q; }
    static void lds_store16(uint8_t *lds, const U32 &off, const Q128V &q, const Bool &pred) {
        for (int i = 0; i < 64; i++) if (pred.v[i]) memcpy(lds + off.v[i], &q.v[i], 16);
    }
    static U32 lds_u8(const uint8_t *lds, const U32 &off) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = lds[off.v[i]]; return r; }
    static U32 lds_read32u(const uint8_t *lds, const U32 &off) { return lds_read32(lds, off); }
    template <int N>
    static U32 splat_byte_n(const U32 &x) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = ((x.v[i] >> (8 * N)) & 0xffu) * 0x01010101u; return r; }
    template <int N>
    static U32 slide_in_byte(const U32 &hi, const U32 &lo) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = (lo.v[i] >> 8) | (((hi.v[i] >> (8 * N)) & 0xffu) << 24); return r; }
    template <uint32_t SEL>
    static U32 perm(const U32 &hi, const U32 &lo) {            // selectors 0..7 (and 0x0C = the constant 0x00): all the bodies use
        V32 r;
        for (int i = 0; i < 64; i++) {
            const uint64_t src = ((uint64_t)hi.v[i] << 32) | lo.v[i];
            uint32_t o = 0;
            for (int k = 0; k < 4; k++) {
                const uint32_t s = (SEL >> (8 * k)) & 0xffu;
                o |= (s < 8u ? (uint32_t)((src >> (8 * s)) & 0xffu) : 0u) << (8 * k);
            }
            r.v[i] = o;
        }
        return r;
    }
    static U32 perm_sel(const U32 &hi, const U32 &lo, const U32 &sel) {        // selectors 0..7 only
        V32 r;
        for (int i = 0; i < 64; i++) {
            const uint64_t src = ((uint64_t)hi.v[i] << 32) | lo.v[i];
            uint32_t o = 0;
            for (int k = 0; k < 4; k++) o |= (uint32_t)((src >> (8 * ((sel.v[i] >> (8 * k)) & 7u))) & 0xffu) << (8 * k);
            r.v[i] = o;
        }
        return r;
    }
    static U32 shr_u(const U32 &x, uint32_t s) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = x.v[i] >> (s & 31u); return r; }
    static U32 alignbit_rt(const U32 &hi, const U32 &lo, uint32_t s) {
        V32 r; for (int i = 0; i < 64; i++) r.v[i] = (uint32_t)(((((uint64_t)hi.v[i]) << 32) | lo.v[i]) >> (s & 31u)); return r;
    }
    static void lds_read64(const uint8_t *lds, const U32 &off, U32 &lo, U32 &hi) {
        for (int i = 0; i < 64; i++) { memcpy(&lo.v[i], lds + off.v[i], 4); memcpy(&hi.v[i], lds + off.v[i] + 4, 4); }
    }
    static void lds_write16(uint8_t *lds, const U32 &off, const U32 &v) {
        for (int i = 0; i < 64; i++) { const uint16_t h = (uint16_t)v.v[i]; memcpy(lds + off.v[i], &h, 2); }
    }
    static void append_u32(uint32_t *list, uint32_t *counter, const U32 &v, const Bool &pred) {
        for (int i = 0; i < 64; i++) if (pred.v[i]) list[(*counter)++] = v.v[i];
    }
    static U32 bfi_k(uint32_t m, const U32 &a, const U32 &b) { return bfi(m, a, b); }
    static U32 and_or(const U32 &a, uint32_t m, const U32 &c) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = (a.v[i] & m) | c.v[i]; return r; }
    using Mask = VB;
    static Mask add_carry_mask(const U32 &a, const U32 &b, U32 &sum) {
        VB c;
        for (int i = 0; i < 64; i++) { const uint64_t t = (uint64_t)a.v[i] + b.v[i]; sum.v[i] = (uint32_t)t; c.v[i] = (t >> 32) != 0; }
        return c;
    }
    static U32 sel_mask(const Mask &m, const U32 &a, const U32 &b) { return sel(m, a, b); }
    template <int N>
    static U32 byte_eq_or(const U32 &x, const U32 &y, const Mask &c) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = (((x.v[i] >> (8 * N)) & 0xffu) == ((y.v[i] >> (8 * N)) & 0xffu) || c.v[i]) ? 1u : 0u; return r; }
    template <int N>
    static Bool byte_eq(const U32 &x, const U32 &y) { VB r; for (int i = 0; i < 64; i++) r.v[i] = ((x.v[i] >> (8 * N)) & 0xffu) == ((y.v[i] >> (8 * N)) & 0xffu); return r; }
    static U32 lds_read32(const uint8_t *lds, const U32 &off) { V32 r; for (int i = 0; i < 64; i++) memcpy(&r.v[i], lds + off.v[i], 4); return r; }
    static void lds_write32(uint8_t *lds, const U32 &off, const U32 &v) { for (int i = 0; i < 64; i++) memcpy(lds + off.v[i], &v.v[i], 4); }
    static void lds_write32p(uint8_t *lds, const U32 &off, const U32 &v, const Bool &pred) { for (int i = 0; i < 64; i++) if (pred.v[i]) memcpy(lds + off.v[i], &v.v[i], 4); }
    static void lds_or32(uint8_t *lds, const U32 &off, const U32 &v, const Bool &pred) {
        for (int i = 0; i < 64; i++) if (pred.v[i]) { uint32_t t; memcpy(&t, lds + off.v[i], 4); t |= v.v[i]; memcpy(lds + off.v[i], &t, 4); }
    }
    static void mem_fence() {}
    static uint32_t readlane(const U32 &x, uint32_t l) { return x.v[l & 63]; }
    static U32 writelane(const U32 &x, uint32_t v, uint32_t l) { V32 r = x; r.v[l & 63] = v; return r; }
    static U32 gload_u8(const Ptr &p, const Bool &pred) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = pred.v[i] ? *p.v[i] : 0u; return r; }
    static uint32_t wave_sum(const U32 &x) { uint32_t t = 0; for (int i = 0; i < 64; i++) t += x.v[i]; return t; }
    static void lds_wave_sync() {}
};

}  // namespace ta
