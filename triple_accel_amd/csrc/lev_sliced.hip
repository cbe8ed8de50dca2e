// lev_sliced.hip -- pair-sliced systolic band kernel: k-bounded unit-cost Levenshtein for fixed-length batches.
//
// Where lev_bits.hip gives every lane one pair and one BIT per band cell, this kernel turns the matrix around: a 32-bit
// register holds the same band cell of 32 different PAIRS (one "set"), and a lane owns a strip of R = 3 adjacent cells of
// the band window.  Every cell is the unit-cost difference cell of Myers (JACM 1999, sec. 3: the cell before it is
// vectorised along a column), evaluated with plain bitwise logic for 32 pairs at once:
//
//     zero = Eq | Mv_in | Mh_in                     (the diagonal step costs 0)
//     Pv_out = Mh_in | ~(zero | Ph_in)   Mv_out = zero & Ph_in        (vertical difference handed to the next column)
//     Ph_out = Mv_in | ~(zero | Pv_in)   Mh_out = zero & Pv_in        (horizontal difference handed to the next row)
//
// with Eq = AND over the 8 bit planes of (a-plane XNOR b-plane); 13 v_bitop3 per cell and 32 pairs.  The window is the one
// of lev_bits_body.h (window cell w of column j is row j - d_hi + w; band [min(0,delta) - t, max(0,delta) + t] widened
// downwards to whole strips; virtual rows above row 0 sit on the fixed point D = j - r and need no masking), and the result
// contract is the scalar path's: out = d <= k ? d : None (src/levenshtein.rs:539-541), d = |delta| + b_len - #(zero steps
// on the answer diagonal).
//
// Schedule.  A strip needs the horizontal difference of the strip above for the SAME column and the vertical difference
// of the strip below for the PREVIOUS column (the window slides), so strip c can run column j at time 2j + c at the
// earliest: every lane would idle every other step.  Each lane therefore carries two register banks; bank (n & 1) runs at
// step n, and a set lives in bank 0 of the even strips and bank 1 of the odd ones (the other set the other way round), so
// that the neighbours' results of step n - 1 are exactly what step n needs.  A group of S <= 15 strips sits in one DPP row
// of 16 lanes, four groups per wavefront: 256 pairs.  Lanes S..15 of every row are switched off (EXEC) in the main loop.
//
// Everything moves by DPP row shifts: the differences (row_shr / row_shl by one lane), the column character (its 8 planes
// travel down the strips, one lane per step) and the rows of `a` (a row serves three columns in a strip, then moves up to
// the strip above).  The row ends cost nothing: lane 0 has no lane above and lane S - 1's neighbour is switched off, so
// those lanes keep the DPP `old` operand -- which is where the next column character / the next row of `a` come in (read
// from LDS one step ahead) and where the band edges get their +1 differences (P is kept complemented so that the zero
// a bound_ctrl shift returns is +1).
//
// Data.  The strings arrive as bytes; the bit planes are made on the fly: per 32 columns every lane loads the same dword
// column of 32 pairs (32 loads), transposes the 32 x 32 bit matrix in registers (bit_transpose.h) and stores 4 positions
// x 8 planes into a 32-position LDS ring per set (single-buffered: only the two entry lanes of a row read it, in order).
// The loads of the next block are in flight while the current 64 steps run.  The answer strip logs its `zero` word per
// visit into LDS; one counting pass at the end turns the log into 32 distances per set.
//
// Algorithmic traffic is that of every pair kernel: a_len + b_len bytes read, 4 written per pair.
//
// STATUS (round 1): bit-exact (tests/test_gpu_lev_sliced.py) but NOT the default.  On cfg2 (1M x 256 B, k = 32) it issues
// 1.74e8 VALU instructions against 3.31e8 of lev_bits.hip, yet runs 0.72 ms against 0.53 ms: a wavefront keeps 256 pairs in
// flight and touches every 128-byte line of its strings in four 32-byte pieces one epoch apart, far beyond what L2 holds,
// so the fabric moves 3.1 GB for 516 MB of strings (0.41 ms with the loads taken out, 0.34 ms for the steps alone --
// DESIGN.md 3.9).  Opt in with TA_FORCE_SLICED=1.
#include "bit_transpose.h"
#include "ta_internal.h"

namespace ta {

namespace {

typedef uint32_t U32;
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;

constexpr U32 SL_SLOTS = 32;           // ring positions per set and string
constexpr U32 SL_SET_BYTES = 2048;     // a planes 0-3 | a planes 4-7 | b planes 0-3 | b planes 4-7, 32 x 16 B each
constexpr U32 SL_HI = 512, SL_B = 1024;
constexpr U32 SL_TMP_HI = 1024;        // start-up layout of a set's 2 KB: 64 positions of `a`, planes 0-3 | planes 4-7

// DPP row shifts by one lane inside a row of 16; a lane without a (live) source keeps `old` / reads 0
#define SL_SHR_OLD(old, x) (U32) __builtin_amdgcn_update_dpp((int)(old), (int)(x), 0x111, 0xf, 0xf, false)   // lane i <- i-1
#define SL_SHL_OLD(old, x) (U32) __builtin_amdgcn_update_dpp((int)(old), (int)(x), 0x101, 0xf, 0xf, false)   // lane i <- i+1
#define SL_SHR0(x) (U32) __builtin_amdgcn_update_dpp(0, (int)(x), 0x111, 0xf, 0xf, true)
#define SL_SHL0(x) (U32) __builtin_amdgcn_update_dpp(0, (int)(x), 0x101, 0xf, 0xf, true)
#define SL_B3(a, b, c, t) (U32) __builtin_amdgcn_bitop3_b32((a), (b), (c), (t))

struct P8 { uint4 lo, hi; };           // the 8 bit planes of one string position, 32 pairs each

struct Bank {
    P8 X[3];                // rows of `a` under the strip's three cells (rotating, see visit())
    P8 T;                   // column character of this bank's last visit
    U32 nPv[3], Mv[3];      // vertical differences this bank produced for its previous column (P complemented)
    U32 nPh, Mh;            // horizontal difference below the strip's last cell
};

__device__ __forceinline__ U32 neq8(const P8 &a, const P8 &b) {
    U32 ne = a.lo.x ^ b.lo.x;                                 // 0xF6: acc | (a ^ b)
    ne = SL_B3(ne, a.lo.y, b.lo.y, 0xF6); ne = SL_B3(ne, a.lo.z, b.lo.z, 0xF6); ne = SL_B3(ne, a.lo.w, b.lo.w, 0xF6);
    ne = SL_B3(ne, a.hi.x, b.hi.x, 0xF6); ne = SL_B3(ne, a.hi.y, b.hi.y, 0xF6); ne = SL_B3(ne, a.hi.z, b.hi.z, 0xF6);
    return SL_B3(ne, a.hi.w, b.hi.w, 0xF6);
}

__device__ __forceinline__ P8 shr_old(const P8 &old, const P8 &x) {
    P8 r;
    r.lo.x = SL_SHR_OLD(old.lo.x, x.lo.x); r.lo.y = SL_SHR_OLD(old.lo.y, x.lo.y); r.lo.z = SL_SHR_OLD(old.lo.z, x.lo.z); r.lo.w = SL_SHR_OLD(old.lo.w, x.lo.w);
    r.hi.x = SL_SHR_OLD(old.hi.x, x.hi.x); r.hi.y = SL_SHR_OLD(old.hi.y, x.hi.y); r.hi.z = SL_SHR_OLD(old.hi.z, x.hi.z); r.hi.w = SL_SHR_OLD(old.hi.w, x.hi.w);
    return r;
}
__device__ __forceinline__ P8 shl_old(const P8 &old, const P8 &x) {
    P8 r;
    r.lo.x = SL_SHL_OLD(old.lo.x, x.lo.x); r.lo.y = SL_SHL_OLD(old.lo.y, x.lo.y); r.lo.z = SL_SHL_OLD(old.lo.z, x.lo.z); r.lo.w = SL_SHL_OLD(old.lo.w, x.lo.w);
    r.hi.x = SL_SHL_OLD(old.hi.x, x.hi.x); r.hi.y = SL_SHL_OLD(old.hi.y, x.hi.y); r.hi.z = SL_SHL_OLD(old.hi.z, x.hi.z); r.hi.w = SL_SHL_OLD(old.hi.w, x.hi.w);
    return r;
}

__device__ __forceinline__ P8 lds_p8(const uint8_t *lds, U32 addr, U32 hi) {
    P8 r;
    r.lo = *(const uint4 *)(lds + addr);
    r.hi = *(const uint4 *)(lds + addr + hi);
    return r;
}

struct LaneConst {
    U32 vinit[3];                 // column-0 state of the three cells: ~0 on the virtual rows (dv = -1), 0 below (dv = +1);
                                  // the same word serves as nPv and as Mv
    U32 c;
    bool is_ans;
};

// the unit-cost difference cell, P inputs / outputs complemented
//   zero = ~ne | mv | mh;  nPv' = ~mh & (zero | ~nPh) (0x51);  Mv' = zero & ~nPh (0x30);  nPh' = ~mv & (zero | ~nPv);  Mh' = zero & ~nPv
#define SL_CELL(ne, npv, mv, nph, mh, Z, NPV, MV, NPH, MH)            \
    const U32 Z = SL_B3(ne, mv, mh, 0xEF);                            \
    const U32 NPV = SL_B3(Z, nph, mh, 0x51), MV = SL_B3(Z, nph, 0u, 0x30); \
    const U32 NPH = SL_B3(Z, npv, mv, 0x51), MH = SL_B3(Z, npv, 0u, 0x30);

// One visit of bank `me` at step n (the other bank ran step n - 1).  ROT: which register of X takes the entering row --
// the cells then sit on X[ROT+1], X[ROT+2], X[ROT]; NB: the register of the OTHER bank that held its cell-0 row at
// step n - 1.  The planes the entry lanes take in (row of `a` for lane S-1, column character for lane 0) are already in
// me.X[ROT] / me.T (the DPP `old` operand); once this visit has read other.T and other.X[NB] they are dead, and the LDS
// reads for the other bank's next visit land right there.
template <int ROT, int NB, bool PRO, int E_ANS>
__device__ __forceinline__ void visit(Bank &me, Bank &other, const uint8_t *lds, U32 next_addr, const LaneConst &L, U32 n, uint8_t *log_slot) {
    me.T = shr_old(me.T, other.T);                            // column character: from the strip above
    me.X[ROT] = shl_old(me.X[ROT], other.X[NB]);              // entering row: the strip below is done with it
    other.X[NB] = lds_p8(lds, next_addr, SL_HI);
    other.T = lds_p8(lds, next_addr + SL_B, SL_HI);
    const P8 &r0 = me.X[(ROT + 1) % 3], &r1 = me.X[(ROT + 2) % 3], &r2 = me.X[ROT];
    const U32 ne0 = neq8(r0, me.T), ne1 = neq8(r1, me.T), ne2 = neq8(r2, me.T);
    const U32 nph = SL_SHR0(other.nPh), mh = SL_SHR0(other.Mh);            // strip above, same column (row start: +1)
    const U32 npvb = SL_SHL0(other.nPv[0]), mvb = SL_SHL0(other.Mv[0]);    // strip below, previous column (row end: +1)
    SL_CELL(ne0, me.nPv[1], me.Mv[1], nph, mh, z0, npv0, mv0, nph0, mh0)
    SL_CELL(ne1, me.nPv[2], me.Mv[2], nph0, mh0, z1, npv1, mv1, nph1, mh1)
    SL_CELL(ne2, npvb, mvb, nph1, mh1, z2, npv2, mv2, nph2, mh2)
    me.nPh = nph2; me.Mh = mh2;
    me.nPv[0] = npv0; me.Mv[0] = mv0; me.nPv[1] = npv1; me.Mv[1] = mv1; me.nPv[2] = npv2; me.Mv[2] = mv2;
    if (PRO) {                                                // the strip has not reached column 1 yet: stay at column 0
        const bool notyet = n <= L.c;
#pragma unroll
        for (int e = 0; e < 3; e++) { me.nPv[e] = notyet ? L.vinit[e] : me.nPv[e]; me.Mv[e] = notyet ? L.vinit[e] : me.Mv[e]; }
    }
    if (L.is_ans) *(uint32_t *)log_slot = E_ANS == 0 ? z0 : E_ANS == 1 ? z1 : z2;
}

struct SlicedParams {
    const uint8_t *a, *b;
    uint64_t a_stride, b_stride;   // bytes between consecutive pairs
    uint64_t a_bytes, b_bytes;     // readable bytes behind a / b
    uint32_t alen, blen, n, k;
    uint32_t *out;
    int32_t dhi;                   // highest diagonal (j - i) of the band
    uint32_t S;                    // strips per group (odd, <= 15)
    uint32_t c_ans;                // strip holding the answer diagonal
    uint32_t dabs;                 // |b_len - a_len|
    uint32_t steps;                // time steps, a multiple of 64
};

// dword at byte offset `off` of a buffer of `total` readable bytes; bytes outside read as 0
__device__ __forceinline__ U32 load32_safe(const uint8_t *base, int64_t off, uint64_t total) {
    if (off >= 0 && (uint64_t)off + 4u <= total) return *(const u32_unaligned *)(base + off);
    U32 v = 0;
    for (int q = 0; q < 4; q++) {
        const int64_t o = off + q;
        if (o >= 0 && (uint64_t)o < total) v |= (U32)base[o] << (8 * q);
    }
    return v;
}

// 32 pairs x one dword column: lane (set rs, dword rt) loads pair p's dword at pos0 + 4 rt.  Blocks that stay inside the
// buffer (all but the first / last wavefront's) take the unchecked path.
__device__ __forceinline__ void load_block(const uint8_t *base, uint64_t stride, uint64_t total, uint64_t pair0, uint32_t n,
                                           int64_t pos0, U32 rs, U32 rt, U32 (&X)[32]) {
    const uint64_t first = pair0 + rs * 32u;
    const int64_t off0 = (int64_t)(first * stride) + pos0 + 4 * (int64_t)rt;
    const bool fast = first + 31u < n && off0 >= 0 && (uint64_t)off0 + 31u * stride + 4u <= total;
    if (__builtin_amdgcn_ballot_w64(!fast) == 0) {
        const uint8_t *ptr = base + off0;
#pragma unroll
        for (int p = 0; p < 32; p++) {
            X[p] = *(const u32_unaligned *)ptr;
            ptr += stride;
        }
    } else {
#pragma unroll 1
        for (int p = 0; p < 32; p++) {
            uint64_t pair = first + (uint32_t)p;
            if (pair >= n) pair = n - 1u;
            const U32 v = load32_safe(base, (int64_t)(pair * stride) + pos0 + 4 * (int64_t)rt, total);
#pragma unroll
            for (int q = 0; q < 32; q++) X[q] = q == p ? v : X[q];
        }
    }
}

// transpose and store 4 positions x 8 planes at slots (slot0 + 4 rt + 0..3) & mask of set rs; planes 4-7 sit `hi` bytes up
__device__ __forceinline__ void store_block(uint8_t *lds, U32 region, U32 hi, U32 slot0, U32 mask, U32 rs, U32 rt, U32 (&X)[32]) {
    bit_transpose32(X);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const U32 slot = (slot0 + 4u * rt + (U32)q) & mask;
        uint8_t *dst = lds + rs * SL_SET_BYTES + region + slot * 16u;
        *(uint4 *)dst = make_uint4(X[8 * q], X[8 * q + 1], X[8 * q + 2], X[8 * q + 3]);
        *(uint4 *)(dst + hi) = make_uint4(X[8 * q + 4], X[8 * q + 5], X[8 * q + 6], X[8 * q + 7]);
    }
}

__device__ __forceinline__ int floor_half(int x) { return x >> 1; }

template <int E_ANS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void lev_sliced_kernel(SlicedParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const U32 lane = threadIdx.x;
    const U32 S = P.S;
    const U32 g = lane >> 4, c = lane & 15u;
    const bool active = c < S;
    const U32 par = c & 1u;
    const U32 log_base = 8u * SL_SET_BYTES;
    LaneConst L;
    L.c = c;
    const int ci = (int)c, dhi = P.dhi;
#pragma unroll
    for (int e = 0; e < 3; e++) L.vinit[e] = (3 * ci + e - dhi) >= 1 ? 0u : ~0u;   // row of cell e at column 0
    L.is_ans = c == P.c_ans;
    // the answer strip logs one word per visit: 32 words per set and epoch, counted at the end of every epoch
    uint8_t *const log0 = lds + log_base + (2u * g + par) * 128u;                   // bank -> set: 2g + (bank ^ par)
    uint8_t *const log1 = lds + log_base + (2u * g + (par ^ 1u)) * 128u;
    const uint64_t pair0 = (uint64_t)blockIdx.x * 256u;

    // refill / counting role of this lane: set rs, dword column rt of a 32-position block; pairs rt, rt+8, rt+16, rt+24
    const U32 rs = lane >> 3, rt = lane & 7u;
    U32 XA[32], XB[32];
    // first log word that is column 1 of set rs: the answer strip holds it in bank beta = (rs ^ c_ans) & 1, whose visit v
    // is column v + floor((beta - 1 - c_ans) / 2) + 1
    const int cnt_v0 = -floor_half((int)((rs ^ P.c_ans) & 1u) - 1 - (int)P.c_ans);
    U32 cnt0 = 0, cnt1 = 0, cnt2 = 0, cnt3 = 0;

    // ---- start-up: the rows of `a` every strip holds before its first visits, and the row entering at step 0 ----
    // a-ring slot q holds a[q + a_sh]; b-ring slot q holds b[q]; the entry lanes consume slot floor((n - 1) / 2) at step n
    const int qoff = (int)(S - 1u) / 2;
    const int a_sh = 3 * (int)S - 1 - dhi - qoff;
    const int p0 = -dhi - 2;                                  // first position of the 64-position start-up window
    load_block(P.a, P.a_stride, P.a_bytes, pair0, P.n, p0, rs, rt, XA);
    store_block(lds, 0u, SL_TMP_HI, 0u, 63u, rs, rt, XA);
    load_block(P.a, P.a_stride, P.a_bytes, pair0, P.n, p0 + 32, rs, rt, XA);
    store_block(lds, 0u, SL_TMP_HI, 32u, 63u, rs, rt, XA);
    Bank b0, b1;
    {
        // bank beta starts as if it had just finished the column before its first one: cells of column floor((beta-1-c)/2)
        const U32 s0 = (2u * g + par) * SL_SET_BYTES, s1 = (2u * g + (par ^ 1u)) * SL_SET_BYTES;
        const int i0 = floor_half(0 - 1 - ci) - dhi - 1 + 3 * ci - p0, i1 = floor_half(1 - 1 - ci) - dhi - 1 + 3 * ci - p0;
#pragma unroll
        for (int e = 0; e < 3; e++) {
            b0.X[e] = lds_p8(lds, s0 + (U32)((i0 + e) & 63) * 16u, SL_TMP_HI);
            b1.X[e] = lds_p8(lds, s1 + (U32)((i1 + e) & 63) * 16u, SL_TMP_HI);
            b0.nPv[e] = b0.Mv[e] = b1.nPv[e] = b1.Mv[e] = L.vinit[e];
        }
        b0.T.lo = b0.T.hi = b1.T.lo = b1.T.hi = make_uint4(0, 0, 0, 0);
        b0.nPh = b0.Mh = b1.nPh = b1.Mh = 0u;
        // the row entering at step 0 (bank 0's X[0] is free: its cell-0 row left with the column before).  S is odd, so
        // lane S-1 and lane 0 hold set 2g in bank 0.
        const P8 ua = lds_p8(lds, (2u * g) * SL_SET_BYTES + (U32)((a_sh - 1 - p0) & 63) * 16u, SL_TMP_HI);
        if (c == S - 1u) b0.X[0] = ua;
    }
    load_block(P.a, P.a_stride, P.a_bytes, pair0, P.n, a_sh, rs, rt, XA);                     // a-block 0
    load_block(P.b, P.b_stride, P.b_bytes, pair0, P.n, 0, rs, rt, XB);                        // b-block 0

    const U32 ring_row = 2u * g * SL_SET_BYTES;               // entry lanes: bank beta <-> set 2g + beta
    const U32 epochs = P.steps / 64u;
#pragma unroll 1
    for (U32 E = 0; E < epochs; E++) {
        store_block(lds, 0u, SL_HI, 0u, SL_SLOTS - 1u, rs, rt, XA);                           // a-block E
        store_block(lds, SL_B, SL_HI, 0u, SL_SLOTS - 1u, rs, rt, XB);                         // b-block E
        load_block(P.a, P.a_stride, P.a_bytes, pair0, P.n, 32 * (int64_t)(E + 1u) + a_sh, rs, rt, XA);
        load_block(P.b, P.b_stride, P.b_bytes, pair0, P.n, 32 * (int64_t)(E + 1u), rs, rt, XB);
        if (active) {
            // step m = 64 E + 2 w (+1): the planes for step m + 1 (bank (m + 1) & 1, slot (m >> 1) & 31) are read while
            // step m runs; visit w & 31 of a bank logs word w & 31
            const U32 nE = 64u * E;
#define SL_STEP2(R, PRO, W)                                                                                              \
    visit<R, R, PRO, E_ANS>(b0, b1, lds, ring_row + SL_SET_BYTES + (W) * 16u, L, nE + 2u * (W), log0 + (W) * 4u);         \
    visit<R, (R + 1) % 3, PRO, E_ANS>(b1, b0, lds, ring_row + (W) * 16u, L, nE + 2u * (W) + 1u, log1 + (W) * 4u);
            // bank 0 at step 2v: its rotation is v % 3 and bank 1's last visit (v - 1) left its cell-0 row in X[v % 3];
            // bank 1 at step 2v + 1: rotation v % 3, bank 0's visit v left its cell-0 row in X[(v + 1) % 3]
            if (E == 0) {
#pragma unroll 1
                for (U32 w = 0; w < 30u; w += 3u) { SL_STEP2(0, true, w) SL_STEP2(1, true, w + 1u) SL_STEP2(2, true, w + 2u) }
                SL_STEP2(0, true, 30u) SL_STEP2(1, true, 31u)
            } else {
#pragma unroll 1
                for (U32 w = 0; w < 30u; w += 3u) { SL_STEP2(0, false, w) SL_STEP2(1, false, w + 1u) SL_STEP2(2, false, w + 2u) }
                SL_STEP2(0, false, 30u) SL_STEP2(1, false, 31u)
            }
#undef SL_STEP2
            // 32 visits per bank = 2 (mod 3): turn the row registers back to rotation 0
            { const P8 t = b0.X[0]; b0.X[0] = b0.X[2]; b0.X[2] = b0.X[1]; b0.X[1] = t; }
            { const P8 t = b1.X[0]; b1.X[0] = b1.X[2]; b1.X[2] = b1.X[1]; b1.X[1] = t; }
        }
        // count the epoch's log: word w of set rs is visit 32 E + w; pairs rt + 8 i ride bits rt + 8 i
        {
            const uint32_t *lg = (const uint32_t *)(lds + log_base + rs * 128u);
            const int vlo = cnt_v0 - 32 * (int)E, vhi = vlo + (int)P.blen;    // words [vlo, vhi) are columns 1..b_len
            U32 acc = 0;
            if (vlo <= 0 && vhi >= 32) {
#pragma unroll
                for (int w = 0; w < 32; w++) acc += (lg[w] >> rt) & 0x01010101u;
            } else {
#pragma unroll 4
                for (int w = 0; w < 32; w++) acc += (w >= vlo && w < vhi) ? ((lg[w] >> rt) & 0x01010101u) : 0u;
            }
            cnt0 += acc & 0xFFu; cnt1 += (acc >> 8) & 0xFFu; cnt2 += (acc >> 16) & 0xFFu; cnt3 += acc >> 24;
        }
    }
    {
        const uint64_t pr = pair0 + rs * 32u + rt;
        const U32 base = P.dabs + P.blen;
        const U32 d0 = base - cnt0, d1 = base - cnt1, d2 = base - cnt2, d3 = base - cnt3;
        if (pr < P.n) P.out[pr] = d0 <= P.k ? d0 : 0xFFFFFFFFu;
        if (pr + 8u < P.n) P.out[pr + 8u] = d1 <= P.k ? d1 : 0xFFFFFFFFu;
        if (pr + 16u < P.n) P.out[pr + 16u] = d2 <= P.k ? d2 : 0xFFFFFFFFu;
        if (pr + 24u < P.n) P.out[pr + 24u] = d3 <= P.k ? d3 : 0xFFFFFFFFu;
    }
}

}  // namespace

bool lev_sliced_applies(const StrView &a, const StrView &b, uint32_t unit_k, uint32_t *strips_out) {
    if (a.off || b.off) return false;                         // fixed-length (strided) batches only
    const LevSlicedPlan pl = lev_sliced_make_plan(a.len, b.len, unit_k);
    if (strips_out) *strips_out = pl.S;
    return pl.ok;
}

hipError_t lev_sliced_launch(const StrView &a, const StrView &b, uint32_t n, uint32_t k, uint32_t unit_k, uint32_t *out,
                             hipStream_t st, uint32_t *grid_out, uint32_t *lds_out, uint32_t *pairs_per_wave) {
    const LevSlicedPlan pl = lev_sliced_make_plan(a.len, b.len, unit_k);
    if (!pl.ok) return hipErrorInvalidValue;
    SlicedParams P;
    P.a = a.blob; P.b = b.blob; P.a_stride = a.stride; P.b_stride = b.stride;
    P.alen = (uint32_t)a.len; P.blen = (uint32_t)b.len; P.n = n; P.k = k; P.out = out;
    P.a_bytes = (uint64_t)(n - 1u) * a.stride + a.len; P.b_bytes = (uint64_t)(n - 1u) * b.stride + b.len;
    P.dabs = pl.dabs; P.dhi = pl.dhi; P.S = pl.S; P.c_ans = pl.c_ans; P.steps = pl.steps;
    const uint32_t lds = 8u * (SL_SET_BYTES + 128u);
    const uint32_t ppw = 256u, grid = (n + ppw - 1u) / ppw;
    if (grid_out) *grid_out = grid;
    if (lds_out) *lds_out = lds;
    if (pairs_per_wave) *pairs_per_wave = ppw;
    if (grid == 0) return hipSuccess;
#define SL_LAUNCH(E)                                                                                                      \
    {                                                                                                                     \
        (void)hipFuncSetAttribute((const void *)lev_sliced_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(lev_sliced_kernel<E>, dim3(grid), dim3(64), lds, st, P);                                       \
    }
    switch (pl.e_ans) {
        case 0: SL_LAUNCH(0) break;
        case 1: SL_LAUNCH(1) break;
        default: SL_LAUNCH(2) break;
    }
#undef SL_LAUNCH
    return hipGetLastError();
}

}  // namespace ta
