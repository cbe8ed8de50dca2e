// lev_band_body.h -- the banded anti-diagonal Levenshtein / restricted-Damerau kernel body.
//
// Replaces the reference's SIMD core levenshtein_simd_core_* (src/levenshtein.rs:829-1195)
// and the jewel layer under it (src/jewel.rs) for BATCHES of pairs; result contract is the
// scalar path's (src/levenshtein.rs:376-607): out = d if d <= k else None.
//
// Mapping.  A pair's band is the diagonals d = j - i in [-u, u]; diagonal
// index p = d + o with o = u|1 (odd), so dp(0,0) sits on an odd p.  L consecutive lanes of a
// wavefront own one pair; lane g owns the D diagonals p in [g*D, g*D+D), one DP cell per
// diagonal kept IN PLACE in a VGPR: cell (i,j) of diagonal p is overwritten by (i+1,j+1) two
// anti-diagonal steps later.  Step s = i + j updates the cells with p + s + o even ("even
// phase": q = p - g*D even, "odd phase": q odd); both neighbours of a cell on the previous
// anti-diagonal are the adjacent diagonals p-1 (cell (i,j-1)) and p+1 (cell (i-1,j)), i.e. the
// registers next door, or -- at a lane's edge -- the neighbouring lane's edge register, fetched
// with one DPP wave_shr:1 / wave_shl:1 per step.  64/L pairs advance in lock-step per wave.
//
// The two strings are consumed strictly sequentially (one new byte of `a` and of `b` per pair
// per iteration = 2 steps), so they are streamed HBM -> LDS in 64-byte ring chunks by coalesced
// 16-byte pieces, and enter the per-lane byte windows (packed 4 chars per VGPR, shifted with
// v_alignbyte) at the group's edge lanes; between lanes the windows shift through DPP too.
//
// Written against a wave policy W (wave.h) so the same text runs as HIP device code and as a
// 64-lane host emulation for tests.
#pragma once
#include "wave.h"

namespace ta {

constexpr uint32_t LEV_INF = 0x3FFFFFFFu;   // "unreachable"; real costs stay far below (n+m < 2^22)
constexpr uint32_t LEV_NEG = 0xC0000001u;   // the same in the score form: -LEV_INF as a signed number
// Streamed chunk = P.ch iterations (= bytes per string; 16, 32 or 64, lev_plan.h), ring = 2 chunks per (pair, string)
// slot in LDS, slot stride = ring + 4: an odd number of dwords, so the per-pair byte reads hit distinct banks
// (slot order: the PW rings of `a`, then the PW rings of `b`).
constexpr uint32_t lev_slot_bytes(uint32_t ch) { return 2u * ch + 4u; }

// the checkpoint-and-recompute batch traceback of the unit-cost families (lev_bits_trace_body.h)
struct LevBitsTraceParams {
    StrView a, b;
    const uint32_t *dist;        // distances of the batch (0xFFFFFFFF: None, no script)
    uint32_t n;                  // pairs
    uint32_t u;                  // unit_k of the pass: the band holds u + 1 (+ 2 with the transposition term) <= 33 diagonals
    uint32_t *ckpt;              // scratch: [wave][tile][word][lane]
    uint32_t ckpt_tiles;         // tiles per wavefront the scratch holds (>= ceil(longest column count / TILE))
    uint32_t *runs;              // [pair][runs_cap]: the runs of the script as the walk closes them -- LAST run first -- (edit type << 29) | count
    uint32_t runs_cap;           // >= 2 u + 2 (a script of cost <= u has at most 2 u + 1 runs) and >= the longest n + m where that is smaller
    uint32_t *n_runs;            // [pair]: runs of the script (0 for None)
    const uint32_t *subset = nullptr;   // optional: the order in which the pairs are taken (lane l of wavefront w: pair subset[64 w + l]) -- CSR batches in length order
    uint32_t packed_cap = 0;     // != 0: `runs` is the CALLER's packed script buffer, packed_cap words per pair -- the walk writes every run where it belongs, right-aligned in the
                                 // pair's slot (run r from the script's end at word (pair + 1) packed_cap - 1 - r): no run list in scratch, no reversal step
};

struct LevParams {
    StrView a, b;
    const uint32_t *subset;   // optional: indices of the pairs to process (exp search), else nullptr
    uint32_t *out;
    uint32_t n;               // pairs (or subset length)
    uint32_t k;               // threshold: out = d <= k ? d : NONE
    uint32_t mc, gc, sg, tc;  // EditCosts as u32 (src/levenshtein.rs:392-398)
    uint32_t u;               // band half-width in diagonals (>= every pair's unit_k, :760-763)
    uint32_t o;               // diagonal index of d = 0 (u | 1)
    uint32_t L;               // lanes per pair
    uint32_t PW;              // pairs per wave = 64 / L
    uint32_t lds_per_wave;    // bytes
    uint32_t Tw;              // warm-up iterations (>= L*D/2; padded so the streamed chunks start on 64-byte lines)
    uint32_t ch;              // bytes per string per streamed chunk
    uint32_t tune = 0;        // bit 2 (4): `subset` is ordered by exact column count (VLINE form); bit 0: chunk form of the bit-parallel band kernel's fetch also for fixed-length batches (set by the launcher)
                              // bit 1: early out of the bit-parallel band kernels (ta_set_option(TA_OPT_EARLY_OUT, 1))
    uint32_t q_table = 0, q_shift = 0;       // lev_bitsq: byte c of q_table = the symbol with code c, code = (byte >> q_shift) & 3
    uint32_t q_memb = 0, q_hi = 0, q_ns = 0; // lev_bitsqw (<= 32 symbols): code = (byte >> q_shift) & 31; bit c of q_memb: code c is a symbol; q_hi = mask of the
                                             // bits outside the code | their value in every symbol << 8; q_ns = number of symbols
    uint32_t *q_bad_count = nullptr, *q_bad_list = nullptr;   // lev_bitsq: the pairs that hold a byte outside the alphabet
    uint32_t *q_next_count = nullptr;        // lev_bitsq: the NEXT pass's counter, zeroed by this pass (two counters taken in turn)
    const uint32_t *n_dev = nullptr;         // bit-parallel band kernels: the number of pairs, read on the device (a list a kernel before wrote)
    uint32_t *ckpt = nullptr;                // stride-8 bit-parallel band kernel, CKPT instantiation: the column state every 16th column, [wave][tile][word][lane]
    uint32_t ckpt_tiles = 0;                 //   (what lev_bits_trace_body.h walks backwards from; tiles per wavefront the scratch holds)
    uint32_t *bnd = nullptr;  // lev_widebits: per wave 6 boundary lines of bnd_line u32 (strings spanning several stripes)
    uint64_t bnd_line = 0;
    uint64_t trace_cols = 0;  // lev_widebits TRACE: columns per stripe in P.trace (>= b_len + 64)
    uint32_t *trace;          // TRACE kernels: 2-bit argmin codes, word w of (iteration tau, phase, lane) at
                              // ((tau*2 + phase)*64 + lane)*LEV_TRACE_WORDS(D) + w, cell c in bits [2c, 2c+2)
    uint64_t trace_wave_words = 0;   // TRACE kernels on a batch: wavefront w's records start at trace + w * trace_wave_words
    uint32_t pair_base = 0;          // TRACE kernels on a batch: the launch's first pair (batches go in chunks that bound the records)
};

constexpr int lev_trace_words(int D) { return (D + 31) / 32; }   // 2 bits x D/2 cells per phase

// TRANS: 0 = no transposition, 1 = transposition as a dot4 penalty (needs 2*mc <= 255 + tc), 2 = as a select
// L1: one lane per pair (the band fits D diagonals: P.L == 1) -- no neighbour lane, so no DPP moves and no edge selects
// SCORE: the cells hold S(i,j) = gc (i+j) - dp(i,j) as SIGNED numbers and the recurrence takes maxima.  A gap step then costs
//   nothing (dp + gc at step s+1 is the same S), the substitution adds the byte 2 gc - mc [a != b] (one v_dot4 with a one-hot
//   multiplier of 1), and an affine cell is dot4, max3, one subtraction of sg, two max: 5 instructions against 7 (linear gaps:
//   2 against 2.5).  Needs 0 <= 2 gc - mc and 2 gc <= 255 (lev_score_form_applies, lev_plan.h); answers are identical -- the
//   map is monotone and exact in 32-bit integers (|S| < 2^31: lengths as for LEV_INF).
// LINE (with L1, fixed-length batches): the LINE form of the fetch -- every 128-byte line of a string requested once, whole, parked in
//   registers and handed to the LDS ring piece by piece (see run()); else the ring is filled chunk by chunk straight from memory.
template <class W, int D, bool AFFINE, int TRANS, bool TRACE = false, bool L1 = false, bool SCORE = false, bool LINE = false>
struct LevBand {
    static_assert(D % 2 == 0 && D >= 2, "D must be even");
    static_assert(!LINE || L1, "the line form of the fetch is the one-lane-per-pair layout's");
    using Q = typename W::Q;
    static_assert(!SCORE || (!TRACE && TRANS != 2), "the score form has no traceback and no select-form transposition");
    static constexpr int Dh = D / 2;                 // cells per lane per phase
    static constexpr int NW = (Dh + 2 + 3) / 4;      // packed window registers (Dh+2 bytes used)
    // the a-window holds a ^ 0x0C (XOR-ed once per 16 bytes on the way into LDS), so the byte test's operand a ^ b ^ 0x0C is
    // ONE v_xor per window register
    static constexpr uint32_t C12 = 0x0C0C0C0Cu;
    using U32 = typename W::U32;
    using Bool = typename W::Bool;
    using Ptr = typename W::Ptr;

    struct State {
        U32 reg[D];   // dp value of the newest cell on each diagonal
        U32 HA[D];    // cheapest way to LEAVE that cell by a gap along j (a_gap of the next cell): see phase()
        U32 HB[D];    // ... by a gap along i (b_gap); aliases HA when !AFFINE (unused)
        U32 PV[D];    // dp value one cell earlier on the diagonal = dp(i-2,j-2) for the next update (TRANS)
        U32 AW[NW];   // byte c+1 = a[i-1] for this lane's cell c (reversed window), byte 0 = intake
        U32 BW[NW];   // byte c+1 = b[j-1] for cell c, byte 0 = b[j-2] of cell 0, byte Dh+1 = intake
        U32 AWp[NW];  // TRANS: the a-window before its last advance = AW moved one cell down (a[i-2] under a[i-1])
        U32 BWp[NW];  // TRANS: the b-window before its last advance = BW moved one cell up   (b[j-2] under b[j-1])
    };

    // One anti-diagonal step for the cells q = 2c + PAR of every lane.
    template <int PAR>
    static TA_HD inline __attribute__((always_inline)) void phase(State &st, const LevParams &P, Bool is_g0, Bool is_gl,
                                                                  uint32_t tau, U32 lane) {
        const U32 INF = W::splat(SCORE ? LEV_NEG : LEV_INF);     // "unreachable" in the form the cells are held in
        U32 X[NW], Z[NW];
        U32 tcodes[lev_trace_words(D)];
        if (TRACE) {
#pragma unroll
            for (int w = 0; w < lev_trace_words(D); w++) tcodes[w] = W::splat(0);
        }
#pragma unroll
        for (int w = 0; w < NW; w++) X[w] = st.AW[w] ^ st.BW[w];
        if (TRANS) {
            // a[i-1]==b[j-2] && a[i-2]==b[j-1]  (src/levenshtein.rs:517-521) as one zero byte per cell
#pragma unroll
            for (int w = 0; w < NW; w++) {
                // b[j-2] and a[i-2] are exactly what the windows held before their last advance: no re-alignment needed
                // (both a-windows carry the 0x0C: taken off inside the OR, put back for the byte test -- three-input functions, v_bitop3)
                Z[w] = ((st.AW[w] ^ st.BWp[w]) ^ C12) | ((st.AWp[w] ^ st.BW[w]) ^ C12);
                if (TRANS == 1 && !SCORE)     // 1 per cell whose transposition test FAILS (non-zero byte; W::ne12: one v_perm_b32 byte test)
                    Z[w] = W::ne12(Z[w] ^ C12) & 0x01010101u;
                if (TRANS == 1 && SCORE)      // the byte 4 gc - tc per cell whose test PASSES, 0 where it fails (one v_bfi_b32 instead of the v_and_b32)
                    Z[w] = W::sel_bits(W::ne12(Z[w] ^ C12), W::splat(0), W::splat((4u * P.gc - P.tc) * 0x01010101u));
            }
        }
        // Linear gaps (!AFFINE): even-q cells are stored BIASED by +gc (they are only read as a gap source by odd
        // cells, or as their own diagonal predecessor), odd-q cells raw with HA = dp + 2gc for their even
        // neighbours -- one add per TWO cells instead of one per cell.
        U32 xl = INF, xr = INF;
        if (L1) {
            // every lane is its pair's first and last: both band edges
        } else if (PAR == 0) {
            xl = W::from_lower0((SCORE && !AFFINE) ? st.reg[D - 1] : st.HA[D - 1]);
            xl = W::sel(is_g0, INF, xl);            // band edge: nothing left of the pair's first diagonal
        } else {
            xr = W::from_upper0(AFFINE ? st.HB[0] : st.reg[0]);
            xr = W::sel(is_gl, INF, xr);
        }
        // per byte: 1 where a != b, four cells per VGPR (SWAR); each cell's substitution cost is then ONE
        // v_dot4_u32_u8 with a one-hot byte of mismatch_cost: reg + flag_byte * mc
        if (SCORE) {
            score_phase<PAR>(st, P, X, Z, xl, xr);
            return;
        }
#pragma unroll
        for (int w = 0; w < NW; w++)                          // 1 per nonzero byte: x ^ 0x0C is 12 exactly where x is 0, and one
            X[w] = W::ne12(X[w]) & 0x01010101u;                 // v_perm_b32 with all-ones sources maps 12 to 0x00, the rest to 0xFF
        // all substitution candidates first: a v_dot4 result needs 3 wait states before another VALU may read it,
        // so the mins below must not directly follow their own dot4
        U32 subv[Dh];
#pragma unroll
        for (int c = 0; c < Dh; c++) subv[c] = W::dot4_byte(X[(c + 1) >> 2], (c + 1) & 3, P.mc, st.reg[2 * c + PAR]);   // :471-475
        U32 tqv[TRANS == 1 ? Dh : 1];
        if (TRANS == 1) {
            // PV holds dp(i-2,j-2) + tc.  A failed test adds 255, which lifts the candidate above nv:
            // nv <= dp(i-1,j-1) + mc <= dp(i-2,j-2) + 2 mc <= dp(i-2,j-2) + tc + 255 (host guarantees the last step).
#pragma unroll
            for (int c = 0; c < Dh; c++) tqv[c] = W::dot4_byte(Z[(c + 1) >> 2], (c + 1) & 3, 255u, st.PV[2 * c + PAR]);   // :523-525 (<= : min)
        }
#pragma unroll
        for (int c = 0; c < Dh; c++) {
            const int q = 2 * c + PAR;
            const int byte = c + 1, w = byte >> 2;
            U32 sub = subv[c];
            const int ql = q > 0 ? q - 1 : 0, qr = q + 1 < D ? q + 1 : D - 1;
            U32 lft = (PAR == 0 && c == 0) ? xl : ((AFFINE || PAR == 0) ? st.HA[ql] : st.reg[ql]);          // a_gap  :476-483
            U32 rgt = (PAR == 1 && c == Dh - 1) ? xr : (AFFINE ? st.HB[qr] : (PAR == 0 ? st.HA[qr] : st.reg[qr]));   // b_gap :484-491
            U32 nv = W::umin3(sub, lft, rgt);                                     // :493-515
            U32 code = W::splat(0);
            if (TRACE) {   // argmin code with the scalar tie order: sub, then a_gap (<), then b_gap (<)   :493-515
                U32 m1 = W::umin(sub, lft);
                code = W::sel(rgt < m1, W::splat(2), W::sel(lft < sub, W::splat(1), W::splat(0)));
            }
            if (TRANS == 1) {
                st.PV[q] = st.reg[q] + P.tc;
                nv = W::umin(nv, tqv[c]);
            } else if (TRANS == 2) {
                U32 t = st.PV[q];
                st.PV[q] = st.reg[q] + P.tc;
                Bool tz = W::byte_of(Z[w], byte & 3) == 0u;
                if (TRACE) code = W::sel(tz & (t <= nv), W::splat(3), code);            // transpose wins ties (<=)
                nv = W::sel(tz, W::umin(nv, t), nv);
            }
            if (TRACE) tcodes[(2 * c) >> 5] = tcodes[(2 * c) >> 5] | (code << ((2 * c) & 31));
            st.reg[q] = nv;
            if (AFFINE) {
                U32 go = nv + (P.sg + P.gc);                                      // open a gap from this cell
                st.HA[q] = W::umin(go, lft + P.gc);                               // or extend the one that reached it
                st.HB[q] = W::umin(go, rgt + P.gc);
            } else if (PAR == 1) {
                st.HA[q] = nv + 2u * P.gc;
            }
        }
        if (TRACE) {
            static_assert(!TRACE || TRANS != 1, "trace kernels use the select form of the transposition");
#pragma unroll
            for (int w = 0; w < lev_trace_words(D); w++)
                W::store_u32(P.trace, (lane + (tau * 2u + (uint32_t)PAR) * 64u) * (uint32_t)lev_trace_words(D) + (uint32_t)w,
                             tcodes[w], lane == lane);
        }
    }

    // The same step on scores (SCORE): X = a ^ b per byte, Z = the transposition test's bytes (zero = passes).
    template <int PAR>
    static TA_HD inline __attribute__((always_inline)) void score_phase(State &st, const LevParams &P, U32 (&X)[NW], U32 (&Z)[NW], U32 xl, U32 xr) {
        // per byte: 2 gc where a == b, 2 gc - mc where not -- what S(i-1,j-1) gains on the way to step i+j   (:471-475)
        const U32 v_ne = W::splat((2u * P.gc - P.mc) * 0x01010101u), v_eq = W::splat(2u * P.gc * 0x01010101u);
#pragma unroll
        for (int w = 0; w < NW; w++) X[w] = W::sel_bits(W::ne12(X[w]), v_ne, v_eq);
        U32 subv[Dh];
#pragma unroll
        for (int c = 0; c < Dh; c++) subv[c] = W::dot4_byte(X[(c + 1) >> 2], (c + 1) & 3, 1u, st.reg[2 * c + PAR]);
        U32 tqv[TRANS == 1 ? Dh : 1];
        if (TRANS == 1) {
            // PV holds S(i-2,j-2) itself; a passed test adds the byte 4 gc - tc (0 <= it: tc/2 < gc; <= 255: lev_score_form_applies),
            // a failed one nothing -- and S(i-2,j-2) alone never beats nv: scores do not fall along a diagonal (every substitution
            // byte is >= 0), so nv >= S(i-1,j-1) + .. >= S(i-2,j-2)   (:523-525).  PV is a plain copy of the diagonal's previous
            // value: in the unrolled loop it is a register NAME, not an instruction.
#pragma unroll
            for (int c = 0; c < Dh; c++) tqv[c] = W::dot4_byte(Z[(c + 1) >> 2], (c + 1) & 3, 1u, st.PV[2 * c + PAR]);
        }
#pragma unroll
        for (int c = 0; c < Dh; c++) {
            const int q = 2 * c + PAR;
            const int ql = q > 0 ? q - 1 : 0, qr = q + 1 < D ? q + 1 : D - 1;
            U32 lft = (PAR == 0 && c == 0) ? xl : (AFFINE ? st.HA[ql] : st.reg[ql]);          // a_gap  :476-483
            U32 rgt = (PAR == 1 && c == Dh - 1) ? xr : (AFFINE ? st.HB[qr] : st.reg[qr]);      // b_gap  :484-491
            U32 nv = W::imax3(subv[c], lft, rgt);                                              // :493-515
            if (TRANS == 1) {
                st.PV[q] = st.reg[q];
                nv = W::imax(nv, tqv[c]);
            }
            st.reg[q] = nv;
            if (AFFINE) {
                U32 go = nv - P.sg;                    // open a gap from this cell
                // or extend the gap that reached the cell: free on scores.  (L1: beyond the band's edge there is no such gap)
                st.HA[q] = (L1 && PAR == 0 && c == 0) ? go : W::imax(go, lft);
                st.HB[q] = (L1 && PAR == 1 && c == Dh - 1) ? go : W::imax(go, rgt);
            }
        }
    }

    // a-window: every char moves one cell up (new row enters at cell 0) -- src/levenshtein.rs:1027-1031
    // (the new char is byte BI of a_in: the hot loop reads the ring four bytes at a time and the intake's v_perm picks the byte)
    template <int BI = 0>
    static TA_HD inline __attribute__((always_inline)) void advance_a(State &st, U32 a_in, Bool is_g0) {
        constexpr int sb = Dh;   // byte Dh = cell Dh-1 = the char the next lane needs
        constexpr uint32_t BIe = L1 ? BI : 0;
        if (TRANS) {
#pragma unroll
            for (int w = 0; w < NW; w++) st.AWp[w] = st.AW[w];
        }
        U32 t = a_in;
        if (!L1) {
            t = st.AW[sb >> 2] >> (8 * (sb & 3));
            t = W::from_lower0(t);
            t = W::sel(is_g0, BI ? a_in >> (8 * BI) : a_in, t);
        }
#pragma unroll
        for (int w = NW - 1; w >= 1; w--) st.AW[w] = W::template alignbyte<3>(st.AW[w], st.AW[w - 1]);
        st.AW[0] = W::template perm<0x0605000Cu | (BIe << 8)>(st.AW[0], t);      // bytes 1, 2 move up, the new char lands in byte 1, byte 0 = 0
    }
    // b-window: every char moves one cell down (new column enters at cell Dh-1) -- :1033-1037
    template <int BI = 0>
    static TA_HD inline __attribute__((always_inline)) void advance_b(State &st, U32 b_in, Bool is_gl) {
        constexpr uint32_t BIe = L1 ? BI : 0;
        if (TRANS) {
#pragma unroll
            for (int w = 0; w < NW; w++) st.BWp[w] = st.BW[w];
        }
        U32 t = b_in;
        if (!L1) {
            t = st.BW[0] >> 8;   // byte 1 = cell 0
            t = W::from_upper0(t);
            t = W::sel(is_gl, BI ? b_in >> (8 * BI) : b_in, t);
        }
        constexpr int ib = Dh + 1, ibm = ib & 3;     // the intake byte: byte ibm of the LAST window register (the bytes above it stay 0)
        static_assert((ib >> 2) == NW - 1, "intake byte in the last window register");
        if constexpr (ibm == 0) {
            // the last register holds nothing but the intake: it goes straight into the register below
#pragma unroll
            for (int w = 0; w < NW - 2; w++) st.BW[w] = W::template alignbyte<1>(st.BW[w + 1], st.BW[w]);
            st.BW[NW - 2] = W::template perm<0x00070605u | (BIe << 24)>(st.BW[NW - 2], t);
            static_assert(ibm != 0 || NW >= 2, "Dh + 1 = 0 mod 4 means at least two registers");
        } else {
#pragma unroll
            for (int w = 0; w < NW - 1; w++) st.BW[w] = W::template alignbyte<1>(st.BW[w + 1], st.BW[w]);
            // one v_perm: the bytes below the intake move down, the new char lands in byte ibm - 1, zeros above
            constexpr uint32_t s0 = ibm == 1 ? BIe : 0x05u, s1 = ibm == 1 ? 0x0Cu : (ibm == 2 ? BIe : 0x06u), s2 = ibm == 3 ? BIe : 0x0Cu;
            st.BW[NW - 1] = W::template perm<(0x0Cu << 24) | (s2 << 16) | (s1 << 8) | s0>(st.BW[NW - 1], t);
        }
    }

    // Stream chunk kc (iterations [kc*CH, kc*CH+CH)) of every pair's two strings into the LDS ring.  The 8 pieces
    // (2 strings x 4 x 16 B) of a pair are fetched by the pair's own L lanes, so no pointer ever crosses lanes.
    static TA_HD inline void load_chunk(uint8_t *lds, const LevParams &P, uint32_t kc, U32 grp, U32 g, Bool active,
                                  Ptr aptr, U32 alen, Ptr bptr, U32 blen, U32 ea, U32 eb) {
        const uint32_t CH = P.ch, PPS = CH / 16u, PIECES = 2u * PPS;
        for (uint32_t base = 0; base < PIECES; base += P.L) {
            U32 pc = g + base;                   // piece index within the pair: [0,PPS) = a, [PPS,2 PPS) = b
            Bool pred = active & (pc < PIECES);
            Bool isb = pc >= PPS;
            U32 piece = W::sel(isb, pc - PPS, pc);
            U32 len = W::sel(isb, blen, alen);
            U32 e = W::sel(isb, eb, ea);
            U32 y0 = piece * 16u + kc * CH;      // ring position (absolute)
            Bool ok = pred & (y0 >= e) & ((y0 - e) < len);
            U32 idx0 = W::sel(ok, y0 - e, W::splat(0));
            auto q = W::gload16(W::ptr_add(W::sel_ptr(isb, bptr, aptr), idx0), ok);
            q = W::qxor_v(q, W::sel(isb, W::splat(0), W::splat(C12)));
            U32 slot = grp + W::sel(isb, W::splat(P.PW), W::splat(0));   // all `a` rings, then all `b` rings
            W::lds_store16(lds, slot * lev_slot_bytes(CH) + (y0 & (2u * CH - 1u)), q, pred);
            // the ring's first four bytes again behind its end (the slot's 4 spare bytes): a 4-byte read may start at any ring byte
            W::lds_write32p(lds, slot * lev_slot_bytes(CH) + 2u * CH, W::qword(q, 0), pred & ((y0 & (2u * CH - 1u)) == 0u));
        }
    }

    static TA_HD inline void run(const LevParams &P0, uint32_t wave_index, uint8_t *lds) {
        LevParams P = P0;
        if (TRACE) P.trace = P0.trace + (uint64_t)wave_index * P0.trace_wave_words;     // this wavefront's records
        const U32 INF = W::splat(SCORE ? LEV_NEG : LEV_INF);
        const U32 lane = W::lane();
        const uint32_t L = P.L;
        const U32 grp = W::udiv(lane, L);
        const U32 g = lane - grp * L;
        const Bool active = grp < P.PW;
        const U32 slot_idx = grp + wave_index * P.PW;
        const Bool valid = active & (slot_idx < P.n);
        const U32 pair = P.subset ? W::load_u32(P.subset, slot_idx, valid, 0u) : (TRACE ? slot_idx + P.pair_base : slot_idx);
        const Bool is_g0 = (g == 0u), is_gl = (g == L - 1);

        Ptr aptr, bptr;
        U32 alen, blen;
        W::load_str(P.a, pair, valid, aptr, alen);
        W::load_str(P.b, pair, valid, bptr, blen);
        if (TRACE) {
            // the reference walks its matrix with the SHORTER string along the rows (src/levenshtein.rs:386-390): the argmin codes --
            // and with them the tie order of the script -- are those of that orientation (the walk relabels the gaps)
            const Bool sw = alen > blen;
            const Ptr pa = W::sel_ptr(sw, bptr, aptr), pb = W::sel_ptr(sw, aptr, bptr);
            const U32 la = W::sel(sw, blen, alen), lb = W::sel(sw, alen, blen);
            aptr = pa; bptr = pb; alen = la; blen = lb;
        }

        // the pair's band (lev_plan.h): diagonals [min(0,delta) - t, max(0,delta) + t], slot p = d + o, o odd
        const U32 s_ans = alen + blen;
        const U32 diff = W::sel(blen >= alen, blen - alen, alen - blen);
        const Bool inband = diff <= P.u;                       // else None (:426-428, :860-862)
        const U32 tband = W::sel(inband, (W::splat(P.u) - diff) >> 1, W::splat(0));
        const U32 o = W::sel(inband, (tband + W::sel(blen >= alen, W::splat(0), diff)) | 1u, W::splat(1));
        // answer cell (alen, blen): step s_ans, slot p_ans = o + blen - alen
        const U32 p_ans = W::sel(inband, (blen + o) - alen, W::splat(0));
        const U32 g_ans = W::udiv(p_ans, (uint32_t)D);
        const U32 q_ans = p_ans - g_ans * (uint32_t)D;

        const uint32_t Tw = P.Tw;                              // warm-up iterations: windows fill up (>= L*Dh)
        const U32 h = (o + 1u) >> 1;                           // <= (u + 2) / 2 <= L * Dh <= Tw
        // iteration t' = tau + Tw feeds a[t' - ca] into lane 0 and b[t' - cb] into lane L-1
        const U32 ca = W::splat(Tw) - h, cb = W::splat(Tw - L * Dh) + h;
        const U32 da = (W::splat(16u) - (ca & 15u)) & 15u, db = (W::splat(16u) - (cb & 15u)) & 15u;
        const U32 ea = ca + da, eb = cb + db;                  // multiples of 16: pieces never straddle index 0
        const uint32_t iters = Tw + W::wave_max((s_ans + 1u) >> 1);
        const U32 t_cap = W::sel(s_ans == 0u, W::splat(0xFFFFFFFFu), ((s_ans - 1u) >> 1) + Tw);

        State st;
#pragma unroll
        for (int q = 0; q < D; q++) {
            st.reg[q] = INF; st.HA[q] = INF;
            if (AFFINE) st.HB[q] = INF;
            if (TRANS) st.PV[q] = INF;
        }
#pragma unroll
        for (int w = 0; w < NW; w++) { st.AW[w] = W::splat(C12); st.BW[w] = W::splat(0); st.AWp[w] = W::splat(C12); st.BWp[w] = W::splat(0); }
        {   // seed dp(0,0) = 0 on diagonal p = o  (:450-452 row 0 then grows through the a_gap chain)
            const U32 gs = W::udiv(o, (uint32_t)D), qs = o - gs * (uint32_t)D;
            const Bool seed_lane = (g == gs);
#pragma unroll
            for (int q = 1; q < D; q += 2) {
                Bool hit = seed_lane & (qs == (uint32_t)q);
                st.reg[q] = W::sel(hit, W::splat(0), st.reg[q]);
                st.HA[q] = W::sel(hit, W::splat(SCORE ? 0u - P.sg : (AFFINE ? P.sg + P.gc : 2u * P.gc)), st.HA[q]);
                if (AFFINE) st.HB[q] = W::sel(hit, W::splat(SCORE ? 0u - P.sg : P.sg + P.gc), st.HB[q]);
            }
        }
        U32 ans = W::sel(s_ans == 0u, W::splat(0), INF);

        const uint32_t CH = P.ch, RMASK = 2u * CH - 1u;
        // (lanes beyond the last whole pair of the wavefront -- 64 % L of them -- read the first pair's rings: their values go nowhere,
        // but an address past the wavefront's LDS is not theirs to read; found by the AddressSanitizer build of the emulation)
        const U32 grp_r = W::sel(active, grp, W::splat(0));
        const U32 a_slot = grp_r * lev_slot_bytes(CH), b_slot = (grp_r + P.PW) * lev_slot_bytes(CH);

        // iterations before min(ca, cb) would only shift zeros into zero windows: start there
        const uint32_t hfar = W::wave_max(W::sel(active, W::sel(h + h >= L * Dh, h, W::splat(L * Dh) - h), W::splat(0)));
        const uint32_t tp0 = Tw - hfar, kc0 = tp0 / CH;
        // ---- LINE form (one lane per pair, fixed-length batch: one geometry for the wavefront).  The chunk form asks for every 128-byte line
        // of a 256-byte string in four 32-byte chunks, 32 iterations apart -- by then the line has left the L2: 3.6-3.9 x the strings'
        // bytes at the fabric side (profiles/r03).  Here a lane requests a whole line of its string in one burst (eight 16-byte loads),
        // parks it in registers (2 x 8 x 16 bytes per lane) and commits the two pieces of the next chunk to the ring where the chunk form
        // would load them; the commit of a line's last piece is followed by the burst for the next line.
        Q SA[8], SB[8];
        const Bool all_lanes = (lane == lane);
        const uint32_t alen_u = (uint32_t)P.a.len, blen_u = (uint32_t)P.b.len;
        uint32_t ea_u = 0, eb_u = 0;
        if (LINE) {                                            // the wavefront's one geometry, on the scalar unit (as above, per lane)
            const uint32_t diff_u = blen_u >= alen_u ? blen_u - alen_u : alen_u - blen_u;
            const bool inband_u = diff_u <= P.u;
            const uint32_t tb_u = inband_u ? (P.u - diff_u) >> 1 : 0u;
            const uint32_t o_u = inband_u ? ((tb_u + (blen_u >= alen_u ? 0u : diff_u)) | 1u) : 1u;
            const uint32_t h_u = (o_u + 1u) >> 1, ca_u = Tw - h_u, cb_u = Tw - L * Dh + h_u;
            ea_u = ca_u + ((16u - (ca_u & 15u)) & 15u); eb_u = cb_u + ((16u - (cb_u & 15u)) & 15u);
        }
        auto fetch_line = [&](Q (&S)[8], const Ptr &p, uint32_t len_u, int32_t m) __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const int32_t off = 128 * m + 16 * c;
                const Bool ok = (off >= 0 && (uint32_t)off < len_u) ? all_lanes : W::bfalse();
                S[c] = W::gload16(W::ptr_add(p, W::splat(off >= 0 ? (uint32_t)off : 0u)), ok);
            }
        };
        // ring position y0 (a multiple of 16) of one string: string piece sp = (y0 - e) / 16, zeros outside the string
        auto commit_piece = [&](Q (&S)[8], const Ptr &p, uint32_t len_u, uint32_t e_u, U32 slot, uint32_t y0, uint32_t x) __attribute__((always_inline)) {
            const int32_t sp = ((int32_t)y0 - (int32_t)e_u) >> 4;
            const bool inside = sp >= 0 && (uint32_t)(16 * sp) < len_u;
            const U32 dst = slot + (y0 & RMASK);
            const bool first = (y0 & RMASK) == 0u;              // the ring's first four bytes again behind its end
            switch (inside ? (sp & 7) : 8) {                     // wave-uniform: one of nine stores (case_tag LAST: see lev_bits_body.h, vput)
#define TA_BPUT(c) case c: { const Q q = W::qxor(S[c], x); W::lds_store16(lds, dst, q, all_lanes); if (first) W::lds_write32(lds, slot + 2u * CH, W::qword(q, 0)); W::template case_tag<c>(); } break;
                TA_BPUT(0) TA_BPUT(1) TA_BPUT(2) TA_BPUT(3) TA_BPUT(4) TA_BPUT(5) TA_BPUT(6) TA_BPUT(7)
#undef TA_BPUT
                default: { const Q q = W::qxor(W::qzero(), x); W::lds_store16(lds, dst, q, all_lanes); if (first) W::lds_write32(lds, slot + 2u * CH, W::qword(q, 0)); } break;
            }
            if (inside && (sp & 7) == 7) fetch_line(S, p, len_u, (sp >> 3) + 1);
        };
        auto commit_chunk = [&](uint32_t kc) __attribute__((always_inline)) {
            for (uint32_t y0 = kc * CH; y0 < kc * CH + CH; y0 += 16u) {
                commit_piece(SA, aptr, alen_u, ea_u, a_slot, y0, C12);
                commit_piece(SB, bptr, blen_u, eb_u, b_slot, y0, 0u);
            }
        };
        if (LINE) {
            const int32_t spa0 = ((int32_t)(kc0 * CH) - (int32_t)ea_u) >> 4, spb0 = ((int32_t)(kc0 * CH) - (int32_t)eb_u) >> 4;
            fetch_line(SA, aptr, alen_u, spa0 > 0 ? spa0 >> 3 : 0);
            fetch_line(SB, bptr, blen_u, spb0 > 0 ? spb0 >> 3 : 0);
            commit_chunk(kc0);
            commit_chunk(kc0 + 1);
        } else {
            load_chunk(lds, P, kc0, grp, g, active, aptr, alen, bptr, blen, ea, eb);
            load_chunk(lds, P, kc0 + 1, grp, g, active, aptr, alen, bptr, blen, ea, eb);
        }
        W::lds_wave_sync();

        for (uint32_t kc = kc0; kc * CH < iters; kc++) {
            if (kc > kc0) {
                if (LINE) commit_chunk(kc + 1);
                else load_chunk(lds, P, kc + 1, grp, g, active, aptr, alen, bptr, blen, ea, eb);
                W::lds_wave_sync();
            }
            const uint32_t t_lo = kc * CH;
            const uint32_t t_hi = (t_lo + CH < iters) ? t_lo + CH : iters;
            uint32_t tp = t_lo > tp0 ? t_lo : tp0;
            // warm-up part: only the char windows move
            for (; tp < t_hi && tp < Tw; tp++) {
                U32 a_in = W::lds_u8(lds, a_slot + ((da + tp) & RMASK));
                U32 b_in = W::lds_u8(lds, b_slot + ((db + tp) & RMASK));
                advance_b(st, b_in, is_gl);
                advance_a(st, a_in, is_g0);
            }
            // DP part: iteration tau = tp - Tw does steps s = 2 tau + 1 (even phase) and 2 tau + 2 (odd phase).  Four iterations per
            // pair of 4-byte ring reads while the chunk lasts (one address computation per four chars), single bytes for the rest.
            const uint32_t s_ans_u = alen_u + blen_u, t_cap_u = s_ans_u == 0u ? 0xFFFFFFFFu : ((s_ans_u - 1u) >> 1) + Tw;
            auto capture = [&](uint32_t t_now) {
                if (LINE && t_now != t_cap_u) return;        // (one geometry for the wavefront: a scalar compare, not a v_cmp + branch per iteration)
                Bool cap = (t_cap == t_now);
                if (__builtin_expect(W::any(cap), 0)) {      // the answer cell was written in this iteration (rare: keep it a branch)
                    U32 r = INF;
#pragma unroll
                    for (int q = 0; q < D; q++) r = W::sel(q_ans == (uint32_t)q, st.reg[q], r);
                    if (!AFFINE && !SCORE) r = r - W::sel((q_ans & 1u) == 0u, W::splat(P.gc), W::splat(0));   // un-bias an even-q cell
                    ans = W::sel(cap, r, ans);
                }
            };
            for (; tp + 4u <= t_hi; tp += 4u) {
                U32 a4 = W::lds_read32u(lds, a_slot + ((da + tp) & RMASK));
                U32 b4 = W::lds_read32u(lds, b_slot + ((db + tp) & RMASK));
#define TA_BAND_ITER(i) \
                phase<0>(st, P, is_g0, is_gl, tp + i - Tw, lane); \
                advance_b<i>(st, b4, is_gl); \
                phase<1>(st, P, is_g0, is_gl, tp + i - Tw, lane); \
                advance_a<i>(st, a4, is_g0); \
                capture(tp + i);
                TA_BAND_ITER(0) TA_BAND_ITER(1) TA_BAND_ITER(2) TA_BAND_ITER(3)
#undef TA_BAND_ITER
            }
            for (; tp < t_hi; tp++) {
                U32 a_in = W::lds_u8(lds, a_slot + ((da + tp) & RMASK));
                U32 b_in = W::lds_u8(lds, b_slot + ((db + tp) & RMASK));
                phase<0>(st, P, is_g0, is_gl, tp - Tw, lane);
                advance_b(st, b_in, is_gl);
                phase<1>(st, P, is_g0, is_gl, tp - Tw, lane);
                advance_a(st, a_in, is_g0);
                capture(tp);
            }
        }

        // the owning lane holds the distance; the pair's first lane writes the result
        U32 d = W::shfl(ans, grp * L + g_ans);
        if (SCORE) d = s_ans * P.gc - d;                       // dp = gc (n + m) - S; an unreachable cell comes out >= LEV_INF
        Bool some = inband & (d <= P.k) & (d < W::splat(LEV_INF));          // :539-541, :1166-1168
        U32 res = W::sel(some, d, W::splat(0xFFFFFFFFu));
        W::store_u32(P.out, pair, res, valid & is_g0);
    }
};

}  // namespace ta
