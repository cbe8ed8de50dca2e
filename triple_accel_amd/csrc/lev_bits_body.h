// lev_bits_body.h -- bit-parallel banded Levenshtein / restricted-Damerau for the two unit-cost families
// (LEVENSHTEIN_COSTS and RDAMERAU_COSTS, src/levenshtein.rs:79-91): what levenshtein(), rdamerau(),
// levenshtein_simd_k() and the levenshtein_exp loop run with (:1397-1526).
//
// Same result contract as lev_band_body.h -- d if d <= k else None (src/levenshtein.rs:539-541) -- but the
// cells of one column of the pair's band (lev_plan.h) are ONE bit each: the vertical differences of the DP
// matrix are in {-1, 0, +1} for unit costs, so a column is two bit-vectors (VP: +1, VN: -1) and a column step is
// ~10 bitwise ops per 32 cells (Myers 1999; Hyyro 2003, "A bit-vector algorithm for computing Levenshtein and
// Damerau edit distances", whose diagonal-band form is used here: the window slides one row down per column,
// so a DIAGONAL is a fixed bit position and a cell's value is its diagonal's start plus the steps on it,
//     D[r][j] on diagonal x = start(x) + sum over columns (1 - D0[bit of x]),
// D0 = "the cell equals its diagonal predecessor").  The diagonal that is followed is the window's TOP one (bit 0, the same
// in every lane: one v_alignbit per column shifts its D0 bit into a 32-column register, one v_bcnt per 32 columns counts them);
// the answer cell sits idx_ans rows further down its column and is reached over that column's vertical differences:
//     d = d_hi + b_len - zero steps on the top diagonal + popcount(VP & below) - popcount(VN & below).  What dominates is no longer the recurrence but building
// the match vector: the window keeps `a` XOR 0x0C, so after the XOR with the column character a byte is 12 exactly
// where the two agree; ONE v_perm_b32 with all-ones sources maps byte 12 to 0x00 and every other value to 0xFF (W::ne12),
// and v_dot4_i32_i8 with the weights -1, -2, ... -128 adds eight such 0 / -1 flags up to their bit mask: 3 instructions
// per dword of the window.
//
// One pair per lane (64 pairs per wavefront); the window is 4*NA bits wide (NA = packed dwords of `a` bytes
// under it; 4*NA - 3 in the static form below).  Rows outside [1, a_len] need no masking: above row 0 the virtual values D[r][j] = j + |r| satisfy
// the recurrence with or without spurious matches, and rows below a_len never feed the rows above them.
// Strings are streamed HBM -> registers -> LDS, 64 bytes per string per refill (see run()).
#pragma once
#include <type_traits>

#include "lev_band_body.h"

namespace ta {

// STATIC: the bytes of `a` under the window stay put for 4 columns (sub-column s reads window bit i from byte i + s and
// shifts the packed mismatch bits by s instead); the registers move a whole dword every 4th column.  Saves the NA
// v_alignbyte per column of the sliding form at the price of 3 window bits.
// LINE: the line form of the fetch (fixed-length batches, see run()); else the chunk form -- one form per instantiation, so that a
// kernel holds ONE copy of the column loop (two copies behind a wave-uniform branch cost 60 % more VGPRs and a wavefront per SIMD)
// S8: the STRIDE-8 form of the 33-diagonal window (NA = 8, bands of up to 33 diagonals: cfg2).  The 32 bytes of `a` under window
// bits 0..31 sit in 8 registers with register m holding the bytes of bits m, m + 8, m + 16, m + 24.  Then
//   * the mismatch flags of register m (0x00 / 0xFF per byte, v_xor + v_perm as everywhere) only need bit m of every byte taken
//     from register m's mask -- a tree of seven v_bfi_b32 over the eight masks (step8) -- to land on their window bits: no Horner shifts;
//   * the window moves one row down per column by RENAMING the registers (m <- m + 1) and pushing the new byte into the one
//     register that wraps around (one v_perm_b32); the loop is unrolled 8 columns so the names are fixed: no moves at all;
//   * the 33rd diagonal is the ONEBIT one (below): its byte is the one about to enter, its match a v_cmp on that byte.
// 39 VALU instructions per column instead of 48 (static form).
// VLINE: the line form for batches whose pairs have their OWN geometry and alignment (CSR batches): every 128-byte line of memory that
// holds bytes of a string is requested once, whole, by the pair's lane -- see run().
// CKPT (stride-8 form; fixed-length batches or CSR ones, with or without a subset list): the column state (VP, VN; with the transposition term PM', the bottom
// diagonal's PM', D0') goes to P.ckpt in front of every 16th column and behind the last one -- the forward sweep of the checkpoint-and-
// recompute traceback (lev_bits_trace_body.h) done by the distance pass itself.
template <class W, int NA, bool TRANS, bool STATIC = false, bool LINE = false, bool S8 = false, bool EARLY = false, bool VLINE = false, bool CKPT = false>
struct LevBits {
    static_assert(!CKPT || (S8 && !VLINE && !EARLY), "checkpoints live in the stride-8 form");
    static_assert(!(LINE && VLINE), "one fetch form per instantiation");
    static_assert(NA >= 1 && NA <= 32, "window of 4..128 diagonals");
    static_assert(!S8 || (NA == 8 && !STATIC), "the stride-8 form is its own window layout: 8 registers, 33 diagonals");
    static constexpr int WB = S8 ? 33 : STATIC ? 4 * NA - 3 : 4 * NA;   // window bits (diagonals)
    static constexpr int NW = (WB + 31) / 32;          // dwords per bit-vector (also holds the 4*NA packed mismatch bits)
    // ONEBIT: the window's last word holds ONE diagonal -- the band's bottom one (static windows of 33, 65, 97 bits: cfg2's 33).
    // That cell has no left neighbour, so its vertical step stays +1 for ever (VP = 1, VN = 0 reproduce themselves) and its
    // D0 is just  match | carry out of the words above;  it feeds the word above through D0 >> 1 and nothing else.  The
    // recurrence then runs on NWF = NW - 1 full words plus one v_cndmask: 6 instructions per column less on cfg2.
    static constexpr bool ONEBIT = (STATIC || S8) && (WB % 32) == 1 && NW > 1;
    static constexpr int NWF = ONEBIT ? NW - 1 : NW;   // words the recurrence keeps state for
    using U32 = typename W::U32;
    using Bool = typename W::Bool;
    using Ptr = typename W::Ptr;
    using Q = typename W::Q;
    // LDS bytes per pair: `a` 64 + 16 look-ahead + 4, `b` 64 + 4 (odd numbers of dwords); the stride-8 line form reads a dword per
    // 4 columns from at most two 16-byte pieces of `a` and one of `b`, so its rings are 3 and 2 pieces (+ 4 bytes of wrap copy)
    static constexpr bool SMALL_RINGS = S8 && LINE;
    static constexpr uint32_t BITS_SLOT_A = SMALL_RINGS ? 52 : 84;
    static constexpr uint32_t BITS_SLOT_B = SMALL_RINGS ? 36 : 68;
    // the sliding form takes `a` out of LDS a byte per column: its bytes are stored XOR 0x0C already (the static form reads a
    // dword per 4 columns and XORs that)
    static constexpr bool PREX = !STATIC && !S8;

    static constexpr uint32_t wmask(int q) { return (q == NW - 1 && (WB & 31)) ? ((1u << (WB & 31)) - 1u) : 0xFFFFFFFFu; }

    struct State {
        U32 VP[NW], VN[NW];     // vertical +1 / -1 differences of the previous column, at the current window's rows
        U32 AW[NA];             // byte i = a[row(i) - 1] ^ 0x0C, row(i) = j - d_hi + i: window bit i <-> byte i
        U32 PMp[NW], D0p[NW];   // TRANS: previous column's match vector and D0
        U32 acc;                // D0 of the window's top diagonal (bit 0), the last columns' bits from bit 31 down; zeros below them
        U32 rD0, rBot, rHP;     // step8<.., REC>: the column's D0 and its horizontal +1 steps HP (window bits 0..31), the bottom diagonal's D0 in bit 0 (lev_bits_trace_body.h)
    };

    // the window moves one row down: byte i <- byte i+1, the next byte of `a` enters on top
    static TA_HD inline __attribute__((always_inline)) void advance_a(State &st, U32 a_in) {
#pragma unroll
        for (int k = 0; k < NA - 1; k++) st.AW[k] = W::template alignbyte<1>(st.AW[k + 1], st.AW[k]);
        st.AW[NA - 1] = W::template alignbyte<1>(a_in, st.AW[NA - 1]);      // (a_in = a ^ 0x0C: PREX)
    }

    // One column: b_in = b[j-1].  D0 of the top diagonal is shifted into st.acc (CAP: a zero for lanes whose pair is finished;
    // the caller counts and clears st.acc at least every 32 columns).
    template <bool CAP, int S = 0>
    static TA_HD inline __attribute__((always_inline)) void column(State &st, U32 b_in, Bool live) {
        // STATIC: b_in holds the four column characters of the group, sub-column S takes byte S
        const U32 Bs = STATIC ? W::template splat_byte_n<S>(b_in) : W::splat_byte(b_in);
        U32 PM[NW], D0[NW], NE[NW];
#pragma unroll
        for (int q = 0; q < NW; q++) {
            // the window bytes carry `a` ^ 0x0C: a byte of x is 12 exactly where a == b, and W::ne12 (one v_perm_b32) turns
            // that into 0x00 / 0xFF = 0 / -1; signed weights -1, -2, .. -128 add eight flags up to their bit mask.  Byte
            // groups from the top down, Horner style: the accumulator of a group is the mask so far, shifted up.
            U32 ne = W::splat(0);
            bool first = true;
#pragma unroll
            for (int p = 3; p >= 0; p--) {
                const int k0 = 8 * q + 2 * p;
                if (k0 >= NA) continue;
                U32 acc = first ? W::sdot4_first(W::ne12(st.AW[k0] ^ Bs), W::splat(0xF8FCFEFFu))
                                : W::sdot4(W::ne12(st.AW[k0] ^ Bs), W::splat(0xF8FCFEFFu), ne << 8);
                first = false;
                if (k0 + 1 < NA) acc = W::sdot4(W::ne12(st.AW[k0 + 1] ^ Bs), W::splat(0x80C0E0F0u), acc);
                ne = acc;
            }
            NE[q] = ne;
        }
#pragma unroll
        for (int q = 0; q < NWF; q++) {                 // STATIC: window bit i is register byte i + S
            const U32 ne = (!STATIC || S == 0) ? NE[q] : ((q + 1 < NW) ? W::template alignbit<(S ? S : 1)>(NE[q + 1], NE[q]) : (NE[q] >> S));
            PM[q] = ~ne & wmask(q);
        }
        // ONEBIT: bit 0 = "the bottom diagonal's characters match" (the bits above it are not used)
        const U32 pm_bot = ONEBIT ? ~(S == 0 ? NE[NW - 1] : (NE[NW - 1] >> S)) : W::splat(0);
        // D0 = (((PM & VP) + VP) ^ VP) | PM | VN      (Hyyro 2003, eq. for the diagonal zero-difference vector)
        Bool carry = W::bfalse();
#pragma unroll
        for (int q = 0; q < NWF; q++) {
            U32 s;
            W::addc(PM[q] & st.VP[q], st.VP[q], carry, s, carry);
            D0[q] = ((s ^ st.VP[q]) | PM[q]) | st.VN[q];
        }
        if (TRANS) {
            // a[i-1] == b[j-2] && a[i-2] == b[j-1] (src/levenshtein.rs:517-521) and the diagonal step before was +1:
            // D0 |= ~D0_prev & (PM << 1) & (PM_prev >> 1)      (window indices: a diagonal keeps its bit; ONEBIT: the bottom
            // diagonal has nothing below it to swap with, its own match bit of the column before is st.PMp[NW - 1])
#pragma unroll
            for (int q = 0; q < NWF; q++) {
                const U32 pml = q ? W::template alignbit<31>(PM[q], PM[q - 1]) : (PM[q] << 1);
                const U32 pmr = (q + 1 < NW) ? W::template alignbit<1>(st.PMp[q + 1], st.PMp[q]) : (st.PMp[q] >> 1);
                D0[q] = D0[q] | (~st.D0p[q] & pml & pmr);
            }
        }
        if (!ONEBIT && (WB & 31)) D0[NW - 1] = D0[NW - 1] & wmask(NW - 1);
        // ONEBIT: the bottom diagonal's D0 = match | carry, in bit 0
        const U32 d0_bot = ONEBIT ? W::sel(carry, W::splat(0xFFFFFFFFu), pm_bot) : W::splat(0);
        st.acc = W::template alignbit<1>(CAP ? W::sel(live, D0[0], W::splat(0)) : D0[0], st.acc);
#pragma unroll
        for (int q = 0; q < NWF; q++) {
            const U32 HP = st.VN[q] | ~(D0[q] | st.VP[q]);
            const U32 HN = D0[q] & st.VP[q];
            const U32 D0s = (q + 1 < NWF) ? W::template alignbit<1>(D0[q + 1], D0[q])                     // next window's rows
                                          : (ONEBIT ? W::template alignbit<1>(d0_bot, D0[q]) : (D0[q] >> 1));
            st.VP[q] = HN | ~(D0s | HP);
            st.VN[q] = D0s & HP;
            if (TRANS) { st.PMp[q] = PM[q]; st.D0p[q] = D0[q]; }
        }
        if (TRANS && ONEBIT) st.PMp[NW - 1] = pm_bot;
    }

    // S8: iteration tp with C = tp % 8, warm-up or column.  Before it, register r holds the bytes of window bits
    // ((r - C) & 7) + 8 q; b_dw / a_raw = the dwords whose byte C & 3 is this iteration's column character / the byte of `a` that
    // enters (it is the bottom diagonal's byte now and the top byte of register C afterwards); a_x = a_raw ^ 0x0C0C0C0C.
    template <bool CAP, int C, bool COLUMN, bool REC = false>
    static TA_HD inline __attribute__((always_inline)) void step8(State &st, U32 b_dw, U32 a_raw, U32 a_x, Bool live) {
        if (COLUMN) {
            const U32 Bs = W::template splat_byte_n<(C & 3)>(b_dw);
            // bit m of every byte from the byte masks of register (C + m) & 7 (register C holds bits 0, 8, 16, 24): a tree of seven
            // v_bfi -- each takes the bits of its mask from one side and ALL the others from the other side, whose stray bits the
            // next level drops (a chain of v_and_or needs eight)
            U32 M[8];
#pragma unroll
            for (int m = 0; m < 8; m++) M[m] = W::ne12(st.AW[(C + m) & 7] ^ Bs);
            const U32 q01 = W::bfi_k(0x01010101u, M[0], M[1]), q23 = W::bfi_k(0x04040404u, M[2], M[3]);
            const U32 q45 = W::bfi_k(0x10101010u, M[4], M[5]), q67 = W::bfi_k(0x40404040u, M[6], M[7]);
            const U32 t = W::bfi_k(0x0F0F0F0Fu, W::bfi_k(0x03030303u, q01, q23), W::bfi_k(0x30303030u, q45, q67));
            const U32 PM = ~t;
            U32 sum, D0, d0_bot;
            if constexpr (TRANS) {
                const Bool m_bot = W::template byte_eq<(C & 3)>(a_raw, b_dw);
                Bool carry = W::bfalse();
                W::addc(PM & st.VP[0], st.VP[0], carry, sum, carry);
                D0 = ((sum ^ st.VP[0]) | PM) | st.VN[0];
                // as column(): the bottom diagonal's own match bit of the column before is st.PMp[NW - 1] bit 0
                const U32 pml = PM << 1, pmr = W::template alignbit<1>(st.PMp[NW - 1], st.PMp[0]);
                D0 = D0 | (~st.D0p[0] & pml & pmr);
                d0_bot = W::sel(carry | m_bot, W::splat(1), W::splat(0));          // ONEBIT: match | carry, in bit 0
                st.PMp[NW - 1] = W::sel(m_bot, W::splat(1), W::splat(0));
            } else {
                // the carry stays a wavefront mask (SGPR pair) from the addition to the bottom diagonal's match | carry: the byte
                // compare is one SDWA v_cmp, two VALU instructions for the 33rd diagonal instead of three
                const typename W::Mask cm = W::add_carry_mask(PM & st.VP[0], st.VP[0], sum);
                D0 = ((sum ^ st.VP[0]) | PM) | st.VN[0];
                d0_bot = W::template byte_eq_or<(C & 3)>(a_raw, b_dw, cm);
            }
            st.acc = W::template alignbit<1>(CAP ? W::sel(live, D0, W::splat(0)) : D0, st.acc);
            const U32 HP = st.VN[0] | ~(D0 | st.VP[0]);
            if (REC) { st.rD0 = D0; st.rBot = d0_bot; st.rHP = HP; }
            const U32 HN = D0 & st.VP[0];
            const U32 D0s = W::template alignbit<1>(d0_bot, D0);
            st.VP[0] = HN | ~(D0s | HP);
            st.VN[0] = D0s & HP;
            if (TRANS) { st.PMp[0] = PM; st.D0p[0] = D0; }
        }
        st.AW[C] = W::template slide_in_byte<(C & 3)>(a_x, st.AW[C]);
    }

    static TA_HD inline void run(const LevParams &P, uint32_t wave_index, uint8_t *lds) {
        const U32 lane = W::lane();
        const U32 slot_idx = lane + wave_index * 64u;
        const Bool in_batch = slot_idx < P.n;
        const U32 pair = P.subset ? W::load_u32(P.subset, slot_idx, in_batch, 0u) : slot_idx;

        Ptr aptr, bptr;
        U32 alen, blen;
        W::load_str(P.a, pair, in_batch, aptr, alen);     // rows
        W::load_str(P.b, pair, in_batch, bptr, blen);     // columns
        if constexpr (CKPT) {
            // the checkpoints are the trace kernel's forward sweep (lev_bits_trace_body.h): rows = the SHORTER string, pair by pair (the
            // distance is symmetric; fixed-length batches arrive with their views swapped by the launcher and no lane swaps here)
            const Bool sw = alen > blen;
            const Ptr tp_ = W::sel_ptr(sw, bptr, aptr);
            bptr = W::sel_ptr(sw, aptr, bptr); aptr = tp_;
            const U32 tl_ = W::sel(sw, blen, alen);
            blen = W::sel(sw, alen, blen); alen = tl_;
        }
        if constexpr (VLINE) {
            // The VLINE form runs ONE column count per pass: every event of the column loop is then wave-uniform (no capped blocks, the
            // way down taken once), only the rows' geometry and the alignments are per lane.  The launcher hands it batches ordered
            // by b's exact length (util_kernels.hip), so a wavefront is one pass -- two where a length class ends inside it.  Pairs
            // outside the band (None before any cell, src/levenshtein.rs:426-428; first in the order, any length) never start a pass.
            const U32 diff0 = W::sel(blen >= alen, blen - alen, alen - blen);
            const Bool out_of_band = in_batch & (diff0 > P.u);
            W::store_u32(P.out, pair, W::splat(0xFFFFFFFFu), out_of_band);
            Bool todo = in_batch & !out_of_band;
            while (W::any(todo)) {
                const uint32_t cols = W::first_u32(blen, todo);
                const Bool mine = todo & (blen == cols);
                run_pass(P, lds, lane, mine, pair, aptr, alen, bptr, blen, cols, wave_index);
                todo = todo & !mine;
            }
        } else {
            run_pass(P, lds, lane, in_batch, pair, aptr, alen, bptr, blen, 0u, wave_index);
        }
    }

    // `valid`: the lanes whose pair this pass answers; VLINE: all of them have blen == cols
    static TA_HD inline void run_pass(const LevParams &P, uint8_t *lds, const U32 &lane, const Bool &valid, const U32 &pair,
                                      const Ptr &aptr, const U32 &alen, const Ptr &bptr, const U32 &blen, uint32_t cols, uint32_t wave_index = 0) {
        const U32 grp = lane;
        const Bool active = (lane == lane);

        // the pair's band (lev_plan.h): diagonals d = j - i in [-nlo, d_hi]; window bit i <-> diagonal d_hi - i
        const U32 diff = W::sel(blen >= alen, blen - alen, alen - blen);
        const Bool inband = diff <= P.u;                       // else None (:426-428, :860-862)
        const U32 tband = W::sel(inband, (W::splat(P.u) - diff) >> 1, W::splat(0));
        const U32 nlo = W::sel(inband, tband + W::sel(blen >= alen, W::splat(0), diff) + (TRANS ? 1u : 0u), W::splat(0));
        const U32 dhi = W::splat((uint32_t)WB - 1u) - nlo;
        const U32 idx_ans = W::sel(inband, (dhi + alen) - blen, W::splat(0));   // row a_len at column b_len

        State st;
#pragma unroll
        for (int q = 0; q < NW; q++) {
            const uint32_t lo = 32u * (uint32_t)q;
            // column 0, D[r][0] = |r|: rows r = 1 - d_hi + i >= 1 step up (+1), rows <= 0 step down (-1)
            const U32 below = W::sel(dhi >= lo + 32u, W::splat(0xFFFFFFFFu),
                                     W::sel(dhi <= lo, W::splat(0), W::shlv(W::splat(1), dhi - lo) - 1u));
            st.VN[q] = below & wmask(q);
            st.VP[q] = ~below & wmask(q);
            st.PMp[q] = W::splat(0);
            st.D0p[q] = W::splat(wmask(q));
        }
#pragma unroll
        for (int k = 0; k < NA; k++) st.AW[k] = W::splat(0);
        st.acc = W::splat(0);
        U32 cnt = W::splat(0);
        uint32_t nacc = 0;                                     // columns whose bits (the top nacc of st.acc) are not counted yet, <= 32
        auto flush = [&]() { cnt = W::bcnt(st.acc >> (32u - nacc), cnt); nacc = 0; };      // (nacc >= 1)
        // the column just finished (st.VP / st.VN: bit i = the step from window row i to row i + 1): the way down from the top
        // diagonal's cell to row a_len, idx_ans steps
        auto way_down = [&]() -> U32 {
            U32 t = W::splat(0);
#pragma unroll
            for (int q = 0; q < NW; q++) {
                const uint32_t lo = 32u * (uint32_t)q;
                const U32 mb = W::sel(idx_ans >= lo + 32u, W::splat(0xFFFFFFFFu),
                                      W::sel(idx_ans <= lo, W::splat(0), W::shlv(W::splat(1), idx_ans - lo) - 1u));
                t = W::bcnt(st.VP[q] & mb, t) - W::bcnt(st.VN[q] & mb, W::splat(0));
            }
            return t;
        };
        U32 tail = way_down();                                 // (column 0: a pair without columns ends here)
        // EARLY OUT (P.tune bit 1; fixed-length batches in the stride-8 line form): a cell of the column just finished is its top
        // diagonal's cell plus the vertical steps above it, so no cell of the column is below  top - popcount(VN).  Once that
        // bound exceeds k in every pair of the wavefront no pair can end at or below k any more -- every alignment of cost <= k
        // stays inside the band and passes through this column -- and the wavefront stops: the answers are None either way
        // (src/levenshtein.rs:539-541).  Checked when the zero-step register is counted (every 24-32 columns).  Its own kernel
        // instantiation (EARLY), launched only under ta_set_option(TA_OPT_EARLY_OUT): the reference evaluates the whole band
        // whatever the data, and so do the default kernel and the benchmark.
        static_assert(!EARLY || (S8 && LINE), "the early out lives in the stride-8 line form");
        const bool early = EARLY && (P.tune & 2u) != 0u && P.k < 0x7FFFFFFFu;
        bool dead = false;

        // iteration tp inserts a[tp - ca] into the window and, from tp = T0 on, runs column tp - T0 + 1 with b[tp - T0]
        const uint32_t T0 = P.Tw;
        const U32 ca = W::splat(T0) - nlo, cb = W::splat(T0);
        const U32 da = (W::splat(16u) - (ca & 15u)) & 15u, db = W::splat(0);   // T0 is a multiple of 64 (lev_plan.h)
        const U32 ea = ca + da, eb = cb + db;
        const uint32_t tp0 = T0 - W::wave_max(W::sel(valid, nlo, W::splat(0)));
        const uint32_t iters = T0 + (VLINE ? cols : W::wave_max(blen));
        const U32 t_stop = blen + T0;                          // first iteration past the pair's last column

        // CKPT: the state in front of column 16 t + 1 (iteration T0 + 16 t) and behind the last column, [tile][word][lane]
        constexpr uint32_t CKW = TRANS ? 5u : 2u;
        uint32_t *ck = CKPT ? P.ckpt + (uint64_t)wave_index * P.ckpt_tiles * (CKW * 64u) : nullptr;
        auto save_ck = [&](uint32_t t) {
            uint32_t *c = ck + (uint64_t)t * (CKW * 64u);
            W::store_u32(c, lane, st.VP[0], active); W::store_u32(c + 64, lane, st.VN[0], active);
            if (TRANS) {
                W::store_u32(c + 128, lane, st.PMp[0], active); W::store_u32(c + 192, lane, st.PMp[NW - 1], active);
                W::store_u32(c + 256, lane, st.D0p[0], active);
            }
        };
        // ---- one span of iterations [tp, p_hi) on LDS-resident characters; addr_a(tp) / addr_b(tp) = LDS byte address of the
        // character(s) iteration tp needs (STATIC: the dword whose bytes are iterations tp..tp+3).
        // A pair whose last column has just run takes its way down from the state as it is now:
        auto finished = [&](Bool fin) { if (W::any(fin)) tail = W::sel(fin, way_down(), tail); };
        // (VLINE: the rings carry 8 bytes of wrap copy, so the second dword of an 8-iteration block is the first one's address + 4 --
        // a ds_read offset, not a second per-lane address computation)
        auto run_span = [&](uint32_t tp, uint32_t p_hi, auto addr_a, auto addr_b, auto uniform_tag) -> uint32_t {
            // UNI = line form = one geometry for the wavefront: every pair runs to the last column, none is capped, and the way
            // down is taken once after the last span
            constexpr bool UNI = decltype(uniform_tag)::value;
            if constexpr (S8) {
                // blocks of 8 iterations (tp a multiple of 8, like T0 and the span limits except the very end): two dwords of each
                // string, eight steps with their register names fixed.  One loop per kind of block -- warm-up, whole, capped /
                // cut short -- so that no loop body has paths to join (joins cost a copy per window register).
                for (; tp < p_hi && tp < T0; tp += 8u) {   // warm-up: rows 1..nlo slide in (whole blocks: T0 is a multiple of 64)
                    const U32 pa_w = addr_a(tp);
                    const U32 x0 = W::lds_read32u(lds, pa_w) ^ 0x0C0C0C0Cu, x1 = W::lds_read32u(lds, VLINE ? pa_w + 4u : addr_a(tp + 4u)) ^ 0x0C0C0C0Cu;
                    step8<false, 0, false>(st, x0, x0, x0, active); step8<false, 1, false>(st, x0, x0, x0, active);
                    step8<false, 2, false>(st, x0, x0, x0, active); step8<false, 3, false>(st, x0, x0, x0, active);
                    step8<false, 4, false>(st, x1, x1, x1, active); step8<false, 5, false>(st, x1, x1, x1, active);
                    step8<false, 6, false>(st, x1, x1, x1, active); step8<false, 7, false>(st, x1, x1, x1, active);
                }
                const bool cap = UNI ? false : W::any(t_stop < p_hi);
                if (!cap) {
                    for (; tp + 8u <= p_hi; tp += 8u) {    // the hot loop: whole blocks, every pair live
                        if (CKPT && ((tp - T0) & 15u) == 0u) save_ck((tp - T0) >> 4);
                        if (__builtin_expect(nacc > 24u, 0)) {
                            flush();
                            if (EARLY && UNI && early) {   // tp - T0 columns are done
                                const U32 top = (dhi + (tp - T0)) - cnt;
                                if (!W::any(valid & inband & (top <= W::bcnt(st.VN[0], W::splat(P.k))))) { dead = true; return tp; }
                            }
                        }
                        const U32 pa_h = addr_a(tp), pb_h = addr_b(tp);
                        const U32 r0 = W::lds_read32u(lds, pa_h), r1 = W::lds_read32u(lds, VLINE ? pa_h + 4u : addr_a(tp + 4u));
                        const U32 x0 = r0 ^ 0x0C0C0C0Cu, x1 = r1 ^ 0x0C0C0C0Cu;
                        const U32 b0 = W::lds_read32u(lds, pb_h), b1 = W::lds_read32u(lds, VLINE ? pb_h + 4u : addr_b(tp + 4u));
                        step8<false, 0, true>(st, b0, r0, x0, active); step8<false, 1, true>(st, b0, r0, x0, active);
                        step8<false, 2, true>(st, b0, r0, x0, active); step8<false, 3, true>(st, b0, r0, x0, active);
                        step8<false, 4, true>(st, b1, r1, x1, active); step8<false, 5, true>(st, b1, r1, x1, active);
                        step8<false, 6, true>(st, b1, r1, x1, active); step8<false, 7, true>(st, b1, r1, x1, active);
                        nacc += 8u;
                    }
                    if (!UNI && tp >= p_hi) finished(t_stop == p_hi);
                }
                for (; tp < p_hi; tp += 8u) {              // capped blocks, and the last one when the columns end inside it
                    if (CKPT && ((tp - T0) & 15u) == 0u) save_ck((tp - T0) >> 4);
                    if (nacc > 24u) flush();
                    const U32 pa_c = addr_a(tp), pb_c = addr_b(tp);
                    const U32 r0 = W::lds_read32u(lds, pa_c), r1 = W::lds_read32u(lds, VLINE ? pa_c + 4u : addr_a(tp + 4u));
                    const U32 x0 = r0 ^ 0x0C0C0C0Cu, x1 = r1 ^ 0x0C0C0C0Cu;
                    const U32 b0 = W::lds_read32u(lds, pb_c), b1 = W::lds_read32u(lds, VLINE ? pb_c + 4u : addr_b(tp + 4u));
                    const uint32_t left = p_hi - tp;       // >= 1 columns of this block run
                    Bool l = t_stop > tp, n;               // (a pair's last column: live now, not in the next one)
#define TA_STEP8(c, bw, rw, xw) if (left > (uint32_t)c) { n = t_stop > (tp + (uint32_t)c + 1u); step8<true, c, true>(st, bw, rw, xw, l); finished(l & !n); l = n; }
                    TA_STEP8(0, b0, r0, x0) TA_STEP8(1, b0, r0, x0) TA_STEP8(2, b0, r0, x0) TA_STEP8(3, b0, r0, x0)
                    TA_STEP8(4, b1, r1, x1) TA_STEP8(5, b1, r1, x1) TA_STEP8(6, b1, r1, x1) TA_STEP8(7, b1, r1, x1)
#undef TA_STEP8
                    nacc += left < 8u ? left : 8u;
                }
                return tp;
            }
            if (STATIC) {
                // groups of 4 iterations (tp a multiple of 4; T0 and the span limits are multiples of 4 except the very end);
                // the top diagonal's bits are counted when another group would not fit the register (every 8th group)
                const bool cap = UNI ? false : W::any(t_stop < p_hi);
                for (; tp < p_hi; tp += 4u) {
#pragma unroll
                    for (int k2 = 0; k2 < NA - 1; k2++) st.AW[k2] = st.AW[k2 + 1];
                    st.AW[NA - 1] = W::lds_read32u(lds, addr_a(tp)) ^ 0x0C0C0C0Cu;
                    if (tp < T0) continue;                 // warm-up: rows 1..nlo slide in
                    if (__builtin_expect(nacc > 28u, 0)) flush();
                    const U32 b0 = W::lds_read32u(lds, addr_b(tp)), b1 = b0, b2 = b0, b3 = b0;   // one dword: column s is byte s
                    if (!cap) {
                        column<false, 0>(st, b0, active);
                        if (tp + 1u < p_hi) column<false, 1>(st, b1, active);
                        if (tp + 2u < p_hi) column<false, 2>(st, b2, active);
                        if (tp + 3u < p_hi) column<false, 3>(st, b3, active);
                    } else {                               // (a pair's last column: live now, not in the next one)
                        const Bool l0 = t_stop > tp, l1 = t_stop > (tp + 1u), l2 = t_stop > (tp + 2u), l3 = t_stop > (tp + 3u);
                        column<true, 0>(st, b0, l0);
                        finished(l0 & !l1);
                        if (tp + 1u < p_hi) { column<true, 1>(st, b1, l1); finished(l1 & !l2); }
                        if (tp + 2u < p_hi) { column<true, 2>(st, b2, l2); finished(l2 & !l3); }
                        if (tp + 3u < p_hi) { column<true, 3>(st, b3, l3); finished(l3 & !(t_stop > (tp + 4u))); }
                    }
                    nacc += p_hi - tp < 4u ? p_hi - tp : 4u;
                }
                if (!UNI && !cap) finished(t_stop == p_hi);
                return tp;
            }
            for (; tp < p_hi && tp < T0; tp++)             // warm-up: rows 1..nlo slide in
                advance_a(st, W::lds_u8(lds, addr_a(tp)));
            if (UNI || !W::any(t_stop < p_hi)) {           // every pair still has columns up to the span's end
                for (; tp + 4u <= p_hi; tp += 4u) {        // four columns per address computation (the rings carry 4 bytes of wrap copy)
                    if (__builtin_expect(nacc > 28u, 0)) flush();
                    const U32 pa = addr_a(tp), pb = addr_b(tp);
#pragma unroll
                    for (uint32_t s4 = 0; s4 < 4u; s4++) {
                        advance_a(st, W::lds_u8(lds, pa + s4));
                        column<false, 0>(st, W::lds_u8(lds, pb + s4), active);
                    }
                    nacc += 4u;
                }
                for (; tp < p_hi; tp++) {
                    if (nacc > 31u) flush();
                    const U32 a_in = W::lds_u8(lds, addr_a(tp)), b_in = W::lds_u8(lds, addr_b(tp));
                    advance_a(st, a_in);
                    column<false, 0>(st, b_in, active);
                    nacc++;
                }
                if (!UNI) finished(t_stop == p_hi);
            } else {
                Bool live = t_stop > tp;
                for (; tp < p_hi; tp++) {
                    if (nacc > 31u) flush();
                    const U32 a_in = W::lds_u8(lds, addr_a(tp)), b_in = W::lds_u8(lds, addr_b(tp));
                    advance_a(st, a_in);
                    column<true, 0>(st, b_in, live);
                    nacc++;
                    const Bool next = t_stop > (tp + 1u);
                    finished(live & !next);                // a pair's last column: live now, not in the next one
                    live = next;
                }
            }
            return tp;
        };

        const U32 a_slot = grp * BITS_SLOT_A, b_slot = grp * BITS_SLOT_B + 64u * BITS_SLOT_A;
        if constexpr (VLINE) {
            // ---- VLINE form (CSR batches: every pair has its own lengths, band geometry AND alignment).  A string is read in
            // the units the memory system moves: the lane requests each 128-byte LINE that holds bytes of its string exactly once, whole
            // (eight 16-byte loads in one burst, parked in registers), so the fabric side of the L2 sees every such line once --
            // the chunk form's 64-byte pieces of unaligned strings, 64 columns apart, made it see them 3.5 times.
            //   * Stream.  Iteration t needs the byte at address  ptr - c + t  (c = ca for `a`: per lane, cb for `b`); its low 32 bits
            //     y = s + t  place it in the lane's stream of 16-byte-aligned PIECES: piece y >> 4, line y >> 7.
            //   * Ring.  LDS holds the last four pieces per string (64 bytes + an 8-byte copy of its start behind the end); ring byte
            //     = y & 63, a per-lane address.  Block u (iterations 16u .. 16u + 15) reads at most pieces (s >> 4) + u .. + 2 and
            //     the step in front of it commits piece  P(u) = (s >> 4) + u + 2  into the slot of the piece that died a block ago.
            //   * Registers.  The commit must name its source register in the instruction, the same in every lane, while P(u) & 7
            //     differs from lane to lane.  So the lanes are in eight CLASSES  kappa = ((s >> 4) + 2) & 7  and a burst of class
            //     kappa loads piece c of its line into register (c - kappa) & 7: step u then commits register u & 7 in EVERY lane.
            //     A lane's burst for its next line follows the commit of its piece 7 -- step u serves the class (7 - u) & 7, eight
            //     loads under that class's lane mask -- and its first piece is due at the next step, 16 iterations later.
            //   * Lines that hold no byte of the string are not requested; bytes of a line outside the string are whatever memory
            //     holds there (rows above the matrix and below a_len need no particular value -- see the file header -- and `b` is
            //     only read inside its columns).
            // per pair: a 64 + 8, b 64 + 8, an 8-byte dump (where the wrap copy of a piece that is not the ring's first goes: an
            // unconditional store to a selected address instead of a store under a lane mask -- no branch), 4 spare: 39 dwords (odd)
            constexpr uint32_t VSLOT = 156u, VB_OFF = 72u, VDUMP = 144u;
            const U32 va_slot = grp * VSLOT, vb_slot = va_slot + VB_OFF;
            uint32_t tp = S8 ? (tp0 & ~7u) : STATIC ? (tp0 & ~3u) : tp0;
            const uint32_t tb0 = tp & ~15u;
            const U32 s_a = W::ptr_lo32(aptr) - ca, s_b = W::ptr_lo32(bptr) - cb;
            const U32 e2a = (s_a >> 4) + 2u, e2b = (s_b >> 4) + 2u;
            // the line of iteration tb0's byte (address ptr - (c - tb0): c >= tb0), as a pointer and relative to the string's start.  (A
            // version that took the first three pieces of a string straight from memory -- one latency in front of the first column instead
            // of up to three -- was 0.7 % faster and asked for the first line twice: 6.55 M lines per launch instead of the floor's 4.3 M.)
            Ptr al = W::ptr_line(W::ptr_sub(aptr, ca - tb0)), bl = W::ptr_line(W::ptr_sub(bptr, cb - tb0));
            U32 rel_a = W::ptr_lo32(al) - W::ptr_lo32(aptr), rel_b = W::ptr_lo32(bl) - W::ptr_lo32(bptr);
            // a line [rel, rel + 128) holds bytes of the string [0, len)  <=>  rel + 127 < len + 127 (unsigned); nothing for lanes
            // without a pair or with an empty string
            const U32 lim_a = W::sel(valid & (alen > 0u), alen + 127u, W::splat(0)), lim_b = W::sel(valid & (blen > 0u), blen + 127u, W::splat(0));
            Q SA[8], SB[8];
#pragma unroll
            for (int c = 0; c < 8; c++) { SA[c] = W::qzero(); SB[c] = W::qzero(); }
            // the eight loads of class kappa's lanes: register j <- piece (j + kappa) & 7 of the line (W::gload_line_keep)
            auto burst_class = [&](Q (&S)[8], const Ptr &line, const Bool &pred, uint32_t kappa) __attribute__((always_inline)) { W::gload_line_keep(S, line, pred, kappa); };
            auto vput = [&](const Q (&S)[8], uint32_t reg, U32 dst, U32 mirror_at, uint32_t x) __attribute__((always_inline)) {
                // (case_tag, LAST in every case: the tails of eight cases that differ only in the index of S get sunk into one block that reads
                // S through a pointer phi -- and S, 64 registers, moves to scratch memory)
                switch (reg & 7u) {                                // wave-uniform: one of eight stores
#define TA_VPUT(c) case c: { const Q q = x ? W::qxor(S[c], x) : S[c]; W::lds_store16(lds, dst, q, active); \
                             W::lds_write32(lds, mirror_at, W::qword(q, 0)); W::lds_write32(lds, mirror_at + 4u, W::qword(q, 1)); W::template case_tag<c>(); } break;
                    TA_VPUT(0) TA_VPUT(1) TA_VPUT(2) TA_VPUT(3) TA_VPUT(4) TA_VPUT(5) TA_VPUT(6) TA_VPUT(7)
#undef TA_VPUT
                }
            };
            // step u: commit piece P(u) of both strings; the lanes whose piece was the last of its line request the next line
            auto vstep = [&](uint32_t u) __attribute__((always_inline)) {      // (not inlined, the parked registers live in scratch memory)
                W::wait_vm0();                                 // the bursts of the step before (16 iterations ago)
                const U32 pa = e2a + u, pb = e2b + u;
                vput(SA, u, va_slot + ((pa & 3u) << 4), W::sel((pa & 3u) == 0u, va_slot + 64u, va_slot + VDUMP), PREX ? 0x0C0C0C0Cu : 0u);
                vput(SB, u, vb_slot + ((pb & 3u) << 4), W::sel((pb & 3u) == 0u, vb_slot + 64u, va_slot + VDUMP), 0u);
                const Bool na = (pa & 7u) == 7u, nb = (pb & 7u) == 7u;
                al = W::sel_ptr(na, W::ptr_add(al, W::splat(128u)), al);  rel_a = W::sel(na, rel_a + 128u, rel_a);
                bl = W::sel_ptr(nb, W::ptr_add(bl, W::splat(128u)), bl);  rel_b = W::sel(nb, rel_b + 128u, rel_b);
                burst_class(SA, al, na & ((rel_a + 127u) < lim_a), 7u - (u & 7u));
                burst_class(SB, bl, nb & ((rel_b + 127u) < lim_b), 7u - (u & 7u));
            };
            // first lines: every class loads the line of iteration tb0's byte
            {
                const Bool oka = (rel_a + 127u) < lim_a, okb = (rel_b + 127u) < lim_b;
#pragma unroll
                for (uint32_t kappa = 0; kappa < 8u; kappa++) {
                    burst_class(SA, al, oka & ((e2a & 7u) == kappa), kappa);
                    burst_class(SB, bl, okb & ((e2b & 7u) == kappa), kappa);
                }
            }
            const uint32_t u0 = tb0 >> 4;
            vstep(u0 - 2u);                                        // pieces P(u0 - 2), P(u0 - 1): what block u0 reads besides P(u0)
            vstep(u0 - 1u);
            for (uint32_t tb = tb0; tb < iters; tb += 16u) {
                vstep(tb >> 4);
                W::lds_wave_sync();
                const uint32_t b_hi = tb + 16u < iters ? tb + 16u : iters;
                tp = run_span(tp, b_hi,
                              [&](uint32_t t) { return va_slot + ((s_a + t) & 63u); },
                              [&](uint32_t t) { return vb_slot + ((s_b + t) & 63u); }, std::true_type());
            }
            tail = way_down();                                  // every pair of the pass ended with the last column run
        } else if constexpr (LINE) {
            // ---- LINE form (fixed-length batches, with or without a subset list): every 128-byte line of a string is requested
            // ONCE, whole -- eight 16-byte loads of the lane's own pair in one burst, parked in registers (2 x 8 x 16 bytes per
            // lane) -- and handed to LDS piece by piece: LDS holds a ring of 5 pieces of `a` and 4 pieces of `b` per pair, each followed by
            // a 4-byte copy of its first dword for reads that run over the end: the same 84 + 68 bytes as the chunk form below.
            // The lengths are the batch's, so the geometry (ca, T0) is one number for the wavefront and every event below is
            // wave-uniform: per 16 iterations one piece of each string moves registers -> LDS (the slot of the piece that just
            // died), and the commit of a line's last piece is followed by the burst for the next line, whose first piece is
            // not due for another 16 iterations.  (The chunk form fetches half lines 64 iterations apart; by then the 4 MB L2
            // has dropped the line: 2.0x the string bytes at the L2's fabric side, measured with TCC_EA0_RDREQ_128B.)
            const uint32_t alen_u = (uint32_t)P.a.len, blen_u = (uint32_t)P.b.len;
            const uint32_t diff_u = blen_u >= alen_u ? blen_u - alen_u : alen_u - blen_u;
            const uint32_t nlo_u = diff_u <= P.u ? ((P.u - diff_u) >> 1) + (blen_u >= alen_u ? 0u : diff_u) + (TRANS ? 1u : 0u) : 0u;
            const int32_t ca_s = (int32_t)T0 - (int32_t)nlo_u;          // iteration tp inserts a[tp - ca_s], column uses b[tp - T0]
            constexpr int32_t RA = SMALL_RINGS ? 3 : 5, RB = SMALL_RINGS ? 2 : 4;
            Q SA[8], SB[8];
            auto fetch_a = [&](int32_t m) {
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const int32_t off = 128 * m + 16 * c;
                    // (every lane loads: a lane without a pair points at the batch's first pair -- load_str -- and its bytes go nowhere;
                    // a per-lane predicate would have every register zeroed before every burst)
                    const Bool ok = (off >= 0 && (uint32_t)off < alen_u) ? active : W::bfalse();
                    SA[c] = W::gload16(W::ptr_add(aptr, W::splat(off >= 0 ? (uint32_t)off : 0u)), ok);
                }
            };
            auto fetch_b = [&](int32_t m) {
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const int32_t off = 128 * m + 16 * c;
                    const Bool ok = (off >= 0 && (uint32_t)off < blen_u) ? active : W::bfalse();
                    SB[c] = W::gload16(W::ptr_add(bptr, W::splat(off >= 0 ? (uint32_t)off : 0u)), ok);
                }
            };
            auto put = [&](const Q (&S)[8], int32_t piece, U32 dst, uint32_t wrap_copy_at, uint32_t x) {     // wrap_copy_at: 0 = none
                switch (piece & 7) {                                   // wave-uniform: one of eight stores
#define TA_PUT(c) case c: { const Q q = x ? W::qxor(S[c], x) : S[c]; W::lds_store16(lds, dst, q, active); if (wrap_copy_at) W::lds_write32(lds, dst + wrap_copy_at, W::qword(q, 0)); } break;
                    TA_PUT(0) TA_PUT(1) TA_PUT(2) TA_PUT(3) TA_PUT(4) TA_PUT(5) TA_PUT(6) TA_PUT(7)
#undef TA_PUT
                }
            };
            auto fmod = [](int32_t x, int32_t m) -> uint32_t { const int32_t r = x % m; return (uint32_t)(r < 0 ? r + m : r); };
            auto commit_a = [&](int32_t piece) {
                const uint32_t slot = fmod(piece, RA);
                put(SA, piece, a_slot + 16u * slot, slot == 0u ? 16u * RA : 0u, PREX ? 0x0C0C0C0Cu : 0u);
                if ((piece & 7) == 7) fetch_a((piece >> 3) + 1);
            };
            auto commit_b = [&](int32_t piece) {
                const uint32_t slot = fmod(piece, RB);
                put(SB, piece, b_slot + 16u * slot, slot == 0u ? 16u * RB : 0u, 0u);
                if ((piece & 7) == 7) fetch_b((piece >> 3) + 1);
            };
            uint32_t tp = S8 ? (tp0 & ~7u) : STATIC ? (tp0 & ~3u) : tp0;
            const uint32_t tb0 = tp & ~15u;
            // pieces the first block reads: a string offset x lives in piece x >> 4 (arithmetic shift: offsets before the string
            // are pieces < 0, delivered as zeros)
            int32_t qa = ((int32_t)tb0 - ca_s) >> 4, qb = ((int32_t)tb0 - (int32_t)T0) >> 4;
            fetch_a(qa >> 3);
            fetch_b(qb >> 3);
            for (int32_t x = qa; x < qa + RA - 1; x++) commit_a(x);
            for (int32_t x = qb; x < qb + RB - 1; x++) commit_b(x);
            {
                for (uint32_t tb = tb0; tb < iters; tb += 16u, qa++, qb++) {
                    commit_a(qa + RA - 1);                                  // into the slot of piece qa - 1, which the last block finished
                    commit_b(qb + RB - 1);
                    W::lds_wave_sync();
                    const uint32_t b_hi = tb + 16u < iters ? tb + 16u : iters;
                    tp = run_span(tp, b_hi,
                                  [&](uint32_t t) { return a_slot + fmod((int32_t)t - ca_s, 16 * RA); },
                                  [&](uint32_t t) { return b_slot + fmod((int32_t)t - (int32_t)T0, 16 * RB); }, std::true_type());
                    if (dead) break;
                }
            }
            tail = way_down();                                  // every pair's last column was the last one run
        } else {
        // ---- CHUNK form (CSR batches: every pair has its own geometry).  Per (pair, string) LDS holds ONE 64-byte chunk [0,64)
        // plus the first 16 bytes of the next one [64,80): the read position of iteration tp is tp + d - 64 kc with d in [0,16)
        // (the 16-byte pieces sit on the string's own 16-byte grid), so it may run up to 15 bytes past the chunk.  The next chunk
        // waits in registers (8 x 16 bytes per lane, fetched a whole chunk ahead) and is committed when the current one is used up.
        // (`b` starts at iteration T0, a multiple of 64: its d is 0 and it needs no look-ahead bytes)
        Q S[8];
        // fixed-length batches that take this form (strings of up to one line): the geometry is one number per wavefront, so whether a
        // piece lies inside its string is a wave-uniform question and the loads need no per-lane predicate (which would have hipcc
        // zero the eight registers before every fetch); a lane without a pair reads the batch's first pair (load_str)
        const bool fixed = !P.a.off && !P.b.off;
        const uint32_t fx_alen = (uint32_t)P.a.len, fx_blen = (uint32_t)P.b.len;
        const uint32_t fx_diff = fx_blen >= fx_alen ? fx_blen - fx_alen : fx_alen - fx_blen;
        const uint32_t fx_nlo = fx_diff <= P.u ? ((P.u - fx_diff) >> 1) + (fx_blen >= fx_alen ? 0u : fx_diff) + (TRANS ? 1u : 0u) : 0u;
        const uint32_t fx_ca = T0 - fx_nlo, fx_ea = fx_ca + ((16u - (fx_ca & 15u)) & 15u), fx_eb = T0;
        auto fetch = [&](uint32_t kc) {
            if (fixed) {
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const uint32_t y0 = kc * 64u + 16u * (uint32_t)p;
                    const bool ina = fx_ea <= y0 && y0 - fx_ea < fx_alen, inb = fx_eb <= y0 && y0 - fx_eb < fx_blen;
                    S[p] = W::gload16(W::ptr_add(aptr, W::splat(ina ? y0 - fx_ea : 0u)), ina ? active : W::bfalse());
                    S[4 + p] = W::gload16(W::ptr_add(bptr, W::splat(inb ? y0 - fx_eb : 0u)), inb ? active : W::bfalse());
                }
                return;
            }
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const uint32_t y0 = kc * 64u + 16u * (uint32_t)p;
                const Bool oka = valid & (ea <= y0) & ((W::splat(y0) - ea) < alen);
                const Bool okb = valid & (eb <= y0) & ((W::splat(y0) - eb) < blen);
                S[p] = W::gload16(W::ptr_add(aptr, W::sel(oka, W::splat(y0) - ea, W::splat(0))), oka);
                S[4 + p] = W::gload16(W::ptr_add(bptr, W::sel(okb, W::splat(y0) - eb, W::splat(0))), okb);
            }
        };
        auto commit_main = [&]() {
#pragma unroll
            for (int p = 0; p < 4; p++) {
                W::lds_store16(lds, a_slot + 16u * p, PREX ? W::qxor(S[p], 0x0C0C0C0Cu) : S[p], active);
                W::lds_store16(lds, b_slot + 16u * p, S[4 + p], active);
            }
        };
        auto commit_look = [&]() { W::lds_store16(lds, a_slot + 64u, PREX ? W::qxor(S[0], 0x0C0C0C0Cu) : S[0], active); };
        const uint32_t kc0 = tp0 / 64u;
        fetch(kc0);
        commit_main();
        fetch(kc0 + 1);
        W::lds_wave_sync();

        for (uint32_t kc = kc0; kc * 64u < iters; kc++) {
            const uint32_t t_lo = kc * 64u;
            const uint32_t t_hi = (t_lo + 64u < iters) ? t_lo + 64u : iters;
            const U32 ra = a_slot + da - t_lo, rb = b_slot + db - t_lo;   // LDS address = r + tp
            uint32_t tp = t_lo > tp0 ? t_lo : tp0;
            if (STATIC) tp &= ~3u;                             // whole groups (the extra leading iterations slide zeros in)
            if (S8) tp &= ~7u;
            for (int part = 0; part < 2; part++) {
                // the last 16 iterations of a chunk may read into the look-ahead bytes: commit them first (the fetch
                // was issued at least 48 iterations ago)
                const uint32_t p_hi = part == 0 ? (t_lo + 48u < t_hi ? t_lo + 48u : t_hi) : t_hi;
                if (part == 1) { commit_look(); W::lds_wave_sync(); }
                tp = run_span(tp, p_hi, [&](uint32_t t) { return ra + t; }, [&](uint32_t t) { return rb + t; }, std::false_type());
            }
            if (t_hi < iters) {                                 // next chunk: registers -> LDS, then fetch the one after
                commit_main();
                fetch(kc + 2);
                W::lds_wave_sync();
            }
        }
        }

        if (CKPT) save_ck((W::wave_max(W::sel(valid, blen, W::splat(0))) + 15u) >> 4);     // the state behind the last column (whole tiles: the walk's first look)
        if (nacc) flush();
        const U32 d = (dhi + blen) - cnt + tail;               // the top diagonal starts at d_hi; + columns - zero-difference steps + way down
        const Bool some = inband & (d <= P.k) & (W::splat(dead ? 1u : 0u) == 0u);                 // :539-541, :1166-1168
        W::store_u32(P.out, pair, W::sel(some, d, W::splat(0xFFFFFFFFu)), valid);
    }
};

}  // namespace ta
