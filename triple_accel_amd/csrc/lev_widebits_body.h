// lev_widebits_body.h -- bit-parallel FULL-COLUMN Levenshtein / restricted-Damerau for long unit-cost pairs:
// levenshtein(), rdamerau() and the last rounds of levenshtein_exp (src/levenshtein.rs:1397-1526) on strings of
// a few hundred bytes to 4 KiB, where the band is (nearly) the whole matrix.
//
// One pair per wavefront.  The shorter string is laid along the rows; lane t owns RB = 32*NWL consecutive rows as
// NWL-dword bit-vectors of vertical differences (Pv: +1, Mv: -1; Myers 1999, in Hyyro's D0 form as in
// lev_bits_body.h).  The lanes run skewed -- at step s lane t computes column s - t + 1 -- so the only cross-lane
// traffic per step is what the block below needs from the block above for the same column, produced one step
// earlier: the top bits of the horizontal difference vectors (and of Hyyro's transposition term) plus the column's
// character, four DPP wave_shr:1 moves.
//
// The rows of a lane never change during a pair, so the match vector of a column is a table lookup instead of 64
// byte compares: per lane three small tables in LDS, indexed by bits 7..5, 4..2 and 1..0 of the column's character
// (Eq = A[c >> 5] & B[(c >> 2) & 7] & C[c & 3]), built once per pair with 3 ds_or_b32 per row.  A 9th A row stays
// zero: character code 256 = "no character", used for the virtual columns a lane sees before its first real one.
// Those virtual columns need no predication: with D[i][j] = i + |j| for j <= 0 the state Pv = ~0, Mv = 0 is a fixed
// point of the step when the block above reports a horizontal difference of -1 and nothing matches.
//
// Result: d = m + sum over lanes of (popcount(Pv) - popcount(Mv)) over rows <= n at column m;  out = d <= k ? d : None.
#pragma once
#include "lev_band_body.h"

namespace ta {

// TRACE (single pair, trace_on = true): every step also stores the lane's Pv, Mv and D0 of its column -- 3 bits per cell --
// to P.trace; the host walks them back from (n, m) with the scalar tie order (lev_trace_walk.h).
template <class W, int NWL, bool TRANS, bool TRACE = false>
struct LevWideBits {
    static_assert(NWL == 1 || NWL == 2, "32 or 64 rows per lane");
    static constexpr uint32_t RB = 32u * NWL;                    // rows per lane
    // three tables per lane, indexed by bits 7..5, 4..2 and 1..0 of the character: 9 (incl. the all-zero "no
    // character" row, c = 256) + 8 + 4 rows of 64 lanes x NWL dwords -- 10.5 KB per wavefront at 64 rows per lane
    static constexpr uint32_t ROW = 64u * NWL * 4u;              // bytes per table row
    static constexpr int SH = NWL == 2 ? 9 : 8;                  // log2(ROW)
    static constexpr uint32_t B_BASE = 9u * ROW, C_BASE = 17u * ROW;
    static constexpr uint32_t TABLE_ROWS = 21u;
    static constexpr uint32_t LDS_BYTES = TABLE_ROWS * ROW;
    using U32 = typename W::U32;
    using Bool = typename W::Bool;
    using Ptr = typename W::Ptr;

    struct State {
        U32 Pv[NWL], Mv[NWL], D0p[NWL], Eqp[NWL];
        U32 sP, sM, sX, sc;      // what the lane below reads next step: top dwords of Ph, Mh, X, and the character
        U32 rP, rM, rX;          // ... as received; lane 0 never receives and keeps the row-0 boundary it was given once
        U32 D0l[NWL];            // TRACE: D0 of the column just computed
    };

    // the two table rows of character c, NOT yet combined: the AND happens one step later, so the LDS reads of step
    // s + 1 stay in flight during the arithmetic of step s
    static TA_HD inline __attribute__((always_inline)) void lookup(const uint8_t *lds, U32 c, U32 lane_off, U32 (&T)[3 * NWL]) {
        const U32 ta = (c >> 5) * ROW + lane_off;
        const U32 tb = ((c >> 2) & 7u) * ROW + lane_off;
        const U32 tc = (c & 3u) * ROW + lane_off;
#pragma unroll
        for (int q = 0; q < NWL; q++) {
            T[q] = W::lds_read32(lds, ta + 4u * q);
            T[NWL + q] = W::lds_read32(lds, tb + (B_BASE + 4u * q));
            T[2 * NWL + q] = W::lds_read32(lds, tc + (C_BASE + 4u * q));
        }
    }

    // one column for every lane; rP/rM/rX = top dwords handed down by the lane above (row 0 boundary for lane 0)
    static TA_HD inline __attribute__((always_inline)) void step(State &st, const U32 (&Eq)[NWL], U32 c, U32 rP, U32 rM, U32 rX) {
        U32 D0[NWL], X[NWL], Ph[NWL], Mh[NWL];
        const U32 hN = rM >> 31;                                  // the row above stepped -1 along this column pair
        Bool carry = W::bfalse();
#pragma unroll
        for (int q = 0; q < NWL; q++) {
            const U32 e = q == 0 ? (Eq[0] | hN) : Eq[q];
            U32 s;
            W::addc(e & st.Pv[q], st.Pv[q], carry, s, carry);
            D0[q] = ((s ^ st.Pv[q]) | e) | st.Mv[q];
        }
        if (TRANS) {
            // D0 |= ((~D0_prev & Eq) << 1) & Eq_prev  (Hyyro 2003; src/levenshtein.rs:517-525), the shift crossing lanes
#pragma unroll
            for (int q = 0; q < NWL; q++) X[q] = ~st.D0p[q] & Eq[q];
#pragma unroll
            for (int q = 0; q < NWL; q++)
                D0[q] = D0[q] | (W::template alignbit<31>(X[q], q ? X[q - 1] : rX) & st.Eqp[q]);
        }
#pragma unroll
        for (int q = 0; q < NWL; q++) {
            Ph[q] = st.Mv[q] | ~(D0[q] | st.Pv[q]);
            Mh[q] = D0[q] & st.Pv[q];
        }
#pragma unroll
        for (int q = 0; q < NWL; q++) {
            const U32 Phs = W::template alignbit<31>(Ph[q], q ? Ph[q - 1] : rP);
            const U32 Mhs = W::template alignbit<31>(Mh[q], q ? Mh[q - 1] : rM);
            st.Pv[q] = Mhs | ~(D0[q] | Phs);
            st.Mv[q] = Phs & D0[q];
            if (TRANS) { st.D0p[q] = D0[q]; st.Eqp[q] = Eq[q]; }
            if (TRACE) st.D0l[q] = D0[q];
        }
        st.sP = Ph[NWL - 1]; st.sM = Mh[NWL - 1]; st.sc = c;
        if (TRANS) st.sX = X[NWL - 1];
    }

    // Per-stripe sweep parameters (wave-uniform).
    struct Sweep {
        uint32_t jlo, Cn, m;         // first column, number of columns, length of `b`
        const uint32_t *inP, *inM, *inX;   // top boundary written by the stripe above (nullptr: row 0 of the matrix)
        uint32_t plo, phi;           // columns the stripe above covered
        uint32_t *outP, *outM, *outX;      // this stripe's bottom boundary (nullptr: last stripe)
        uint32_t *trace;             // TRACE: this stripe's records, [column][lane][Pv.. Mv.. D0..] (3*NWL dwords)
    };

    // Step s of the skewed sweep: lane t computes column jlo + s - t.  TAIL (last stripe only): lanes t <= s - Cn hold
    // the last column and freeze.  BND: the stripe's top boundary comes from the stripe above instead of row 0.
    // OUT: lane 63's row is handed to the stripe below.
    template <bool TAIL, bool BND, bool OUT>
    static TA_HD inline __attribute__((always_inline)) void iter(State &st, const uint8_t *lds, U32 lane, U32 lane_off, Ptr bp,
                                                                const Sweep &Z, uint32_t s, U32 &cb, U32 (&hb)[3], const U32 &c,
                                                                const U32 (&T)[3 * NWL], U32 &c_out, U32 (&T_out)[3 * NWL]) {
        // (c, T) = this step's character and table rows; (c_out, T_out) receive the next step's -- two register sets
        // that the caller alternates, so the lookups stay in flight across the loop edge without copies
        U32 Eq[NWL];
#pragma unroll
        for (int q = 0; q < NWL; q++) Eq[q] = T[q] & T[NWL + q] & T[2 * NWL + q];
        // next step's character and match vector first: the LDS lookups overlap this step's arithmetic
        const uint32_t s1 = s + 1u;
        if ((s1 & 63u) == 0u && s1 < Z.Cn) {
            const U32 col = lane + (Z.jlo + s1);                              // 1-based column of lane e's byte
            cb = W::gload_u8(W::ptr_add(bp, col - 1u), col <= Z.m);
        }
        const uint32_t b_next = (s1 < Z.Cn) ? W::readlane(cb, s1 & 63u) : 256u;
        c_out = W::from_lower(c, W::splat(b_next));
        lookup(lds, c_out, lane_off, T_out);
        if (!BND) {
            st.rP = W::from_lower(st.sP, st.rP);      // lane 0 keeps 0x80000000: D[0][j] - D[0][j-1] = +1
            st.rM = W::from_lower(st.sM, st.rM);      // lane 0 keeps 0
            if (TRANS) st.rX = W::from_lower(st.sX, st.rX);
        } else {
            if ((s & 63u) == 0u) {                    // 64 columns of the boundary above, coalesced
                const U32 col = lane + (Z.jlo + s);
                const Bool in = (col >= Z.plo) & (col <= Z.phi);
                hb[0] = W::load_u32(Z.inP, col, in, 0x80000000u);             // outside the band above: a +1 step
                hb[1] = W::load_u32(Z.inM, col, in, 0u);
                hb[2] = TRANS ? W::load_u32(Z.inX, col, in, 0u) : W::splat(0);
            }
            st.rP = W::from_lower(st.sP, W::splat(W::readlane(hb[0], s & 63u)));
            st.rM = W::from_lower(st.sM, W::splat(W::readlane(hb[1], s & 63u)));
            if (TRANS) st.rX = W::from_lower(st.sX, W::splat(W::readlane(hb[2], s & 63u)));
        }
        const U32 rP = st.rP, rM = st.rM, rX = st.rX;
        if (!TAIL) {
            step(st, Eq, c, rP, rM, rX);
        } else {
            State nx = st;
            step(nx, Eq, c, rP, rM, rX);
            const Bool done = lane <= (s - Z.Cn);
#pragma unroll
            for (int q = 0; q < NWL; q++) {
                st.Pv[q] = W::sel(done, st.Pv[q], nx.Pv[q]); st.Mv[q] = W::sel(done, st.Mv[q], nx.Mv[q]);
                st.D0p[q] = nx.D0p[q]; st.Eqp[q] = nx.Eqp[q];
            }
            st.sP = nx.sP; st.sM = nx.sM; st.sX = nx.sX; st.sc = nx.sc;
#pragma unroll
            for (int q = 0; q < NWL; q++) st.D0l[q] = nx.D0l[q];
            if (TRACE) {                                  // (a lane that froze in an earlier step stores nothing)
                const Bool act = (lane <= s) & ((W::splat(s) - lane) < Z.Cn);
                const U32 rec = ((W::splat(Z.jlo + s) - lane) * 64u + lane) * (3u * NWL);
#pragma unroll
                for (int q = 0; q < NWL; q++) {
                    W::store_u32(Z.trace, rec + (uint32_t)q, nx.Pv[q], act);
                    W::store_u32(Z.trace, rec + (uint32_t)(NWL + q), nx.Mv[q], act);
                    W::store_u32(Z.trace, rec + (uint32_t)(2 * NWL + q), nx.D0l[q], act);
                }
            }
        }
        if (TRACE && !TAIL) {
            const Bool act = (lane <= s) & ((W::splat(s) - lane) < Z.Cn);
            const U32 rec = ((W::splat(Z.jlo + s) - lane) * 64u + lane) * (3u * NWL);
#pragma unroll
            for (int q = 0; q < NWL; q++) {
                W::store_u32(Z.trace, rec + (uint32_t)q, st.Pv[q], act);
                W::store_u32(Z.trace, rec + (uint32_t)(NWL + q), st.Mv[q], act);
                W::store_u32(Z.trace, rec + (uint32_t)(2 * NWL + q), st.D0l[q], act);
            }
        }
        if (OUT && s >= 63u) {                        // lane 63's row is the stripe's last: hand it to the stripe below
            const U32 col = W::splat(Z.jlo + (s - 63u));
            const Bool w = (lane == 63u) & (col <= Z.jlo + (Z.Cn - 1u));
            W::store_u32(Z.outP, col, st.sP, w);
            W::store_u32(Z.outM, col, st.sM, w);
            if (TRANS) W::store_u32(Z.outX, col, st.sX, w);
        }
    }

    // sum over columns [from, to] of the horizontal steps (+1 / 0 / -1) a boundary line recorded
    static TA_HD inline uint32_t sum_steps(const uint32_t *bP, const uint32_t *bM, uint32_t from, uint32_t to, U32 lane) {
        U32 acc = W::splat(0);
        for (uint64_t j0 = from; j0 <= to; j0 += 64u) {
            const U32 col = lane + (uint32_t)j0;
            const Bool in = col <= to;
            acc = acc + (W::load_u32(bP, col, in, 0u) >> 31) - (W::load_u32(bM, col, in, 0u) >> 31);
        }
        return W::wave_sum(acc);
    }

    // wave `wave_slot` of `nwaves` resident waves walks the pairs slot, slot + nwaves, ...
    static TA_HD inline void run(const LevParams &P, uint32_t wave_slot, uint32_t nwaves, uint8_t *lds) {
        const U32 lane = W::lane();
        const U32 lane_off = lane * (NWL * 4u);
        const Bool all = (lane == lane);
        constexpr uint32_t ROWS = 64u * RB;                      // rows per stripe
        uint32_t *lines = P.bnd ? P.bnd + (uint64_t)wave_slot * 6u * P.bnd_line : nullptr;
        for (uint32_t slot = wave_slot; slot < P.n; slot += nwaves) {
            const U32 pair = P.subset ? W::load_u32(P.subset, W::splat(slot), all, 0u) : W::splat(slot);
            Ptr xp, yp;
            U32 xl, yl;
            W::load_str(P.a, pair, all, xp, xl);
            W::load_str(P.b, pair, all, yp, yl);
            const uint32_t alen = W::readlane(xl, 0), blen = W::readlane(yl, 0);
            const bool swap = alen > blen;                       // rows <- the shorter string (distance is symmetric)
            const uint32_t n = swap ? blen : alen, m = swap ? alen : blen;
            const Ptr ap = swap ? yp : xp, bp = swap ? xp : yp;
            const Bool lane0 = (lane == 0u);
            uint32_t res;
            if (m - n > P.u) {
                res = 0xFFFFFFFFu;                               // :426-428, :860-862
            } else if (n == 0) {
                res = m <= P.k ? m : 0xFFFFFFFFu;                // one gap run (or two empty strings)
            } else {
                // the pair's band (lev_plan.h): row i visits columns [i - below, i + above]
                const uint32_t tband = (P.u - (m - n)) >> 1;
                const uint64_t below = tband, above = (uint64_t)tband + (m - n);
                const uint32_t stripes = (n + ROWS - 1u) / ROWS;
                uint32_t anchor = 0;                             // D[i0][jlo - 1]
                uint32_t plo = 1, phi = 0;
                uint32_t total = 0;
                for (uint32_t sq = 0; sq < stripes; sq++) {
                    const uint32_t i0 = sq * ROWS;
                    const uint32_t nrows = (n - i0 < ROWS) ? n - i0 : ROWS;
                    const bool last = (sq + 1u == stripes);
                    Sweep Z;
                    Z.m = m;
                    Z.jlo = ((uint64_t)i0 + 1u > below) ? (uint32_t)(i0 + 1u - below) : 1u;
                    const uint64_t hi64 = (uint64_t)i0 + nrows + above;
                    const uint32_t jhi = hi64 < m ? (uint32_t)hi64 : m;
                    Z.Cn = jhi - Z.jlo + 1u;                     // >= 1 because m - n <= u
                    uint32_t *rd = lines ? lines + (uint64_t)((sq + 1u) & 1u) * 3u * P.bnd_line : nullptr;   // written by stripe sq-1
                    uint32_t *wr = lines ? lines + (uint64_t)(sq & 1u) * 3u * P.bnd_line : nullptr;
                    Z.inP = sq ? rd : nullptr; Z.inM = sq ? rd + P.bnd_line : nullptr; Z.inX = sq ? rd + 2u * P.bnd_line : nullptr;
                    Z.outP = last ? nullptr : wr; Z.outM = last ? nullptr : wr + P.bnd_line; Z.outX = last ? nullptr : wr + 2u * P.bnd_line;
                    Z.plo = plo; Z.phi = phi;
                    Z.trace = TRACE ? P.trace + (uint64_t)sq * P.trace_cols * (64u * 3u * NWL) : nullptr;
                    if (sq) {                                    // D[i0][jlo-1]: down the stripe above at its column plo-1, then right
                        anchor += ROWS;
                        if (Z.jlo > plo) anchor += sum_steps(Z.inP, Z.inM, plo, Z.jlo - 1u, lane);
                    }

                    // ---- per-lane nibble tables of this lane's rows [i0 + lane*RB, +RB)
#pragma unroll 1
                    for (uint32_t e = 0; e < TABLE_ROWS; e++) {
#pragma unroll
                        for (int q = 0; q < NWL; q++) W::lds_write32(lds, lane_off + e * ROW + 4u * q, W::splat(0));
                    }
                    W::lds_wave_sync();
                    const U32 row0 = lane * RB + i0;
#pragma unroll 1
                    for (uint32_t r0 = 0; r0 < RB; r0 += 16) {
                        const U32 ia = row0 + r0;
                        auto piece = W::gload16(W::ptr_add(ap, ia), ia < n);   // blobs carry 16 bytes of slack (TA_BLOB_SLACK)
                        const U32 w4[4] = {W::qword(piece, 0), W::qword(piece, 1), W::qword(piece, 2), W::qword(piece, 3)};
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const U32 ch = W::byte_of(w4[r >> 2], r & 3);
                            const Bool ok = (ia + (uint32_t)r) < n;
                            const uint32_t bit = 1u << ((r0 + r) & 31u), qo = 4u * ((r0 + r) >> 5);
                            W::lds_or32(lds, (ch >> 5) * ROW + lane_off + qo, W::splat(bit), ok);
                            W::lds_or32(lds, ((ch >> 2) & 7u) * ROW + lane_off + B_BASE + qo, W::splat(bit), ok);
                            W::lds_or32(lds, (ch & 3u) * ROW + lane_off + C_BASE + qo, W::splat(bit), ok);
                        }
                    }
                    W::lds_wave_sync();

                    State st;
#pragma unroll
                    for (int q = 0; q < NWL; q++) {
                        st.Pv[q] = W::splat(0xFFFFFFFFu); st.Mv[q] = W::splat(0);   // column jlo-1: D grows by 1 per row (exact for jlo = 1)
                        st.D0p[q] = W::splat(0xFFFFFFFFu); st.Eqp[q] = W::splat(0);
                    }
                    st.sP = W::splat(0); st.sM = W::splat(0x80000000u); st.sX = W::splat(0); st.sc = W::splat(256);
                    st.rP = W::splat(0x80000000u); st.rM = W::splat(0); st.rX = W::splat(0);
                    const uint32_t t_last = (nrows - 1u) / RB;   // lane holding the stripe's last row
                    const uint32_t steps = Z.Cn + t_last;

                    const U32 col0 = lane + Z.jlo;
                    U32 cb = W::gload_u8(W::ptr_add(bp, col0 - 1u), col0 <= m);
                    U32 hb[3] = {W::splat(0x80000000u), W::splat(0), W::splat(0)};
                    U32 c = W::from_lower(st.sc, W::splat(W::readlane(cb, 0)));
                    U32 T[3 * NWL], T2[3 * NWL], c2;
                    lookup(lds, c, lane_off, T);
                    // one specialised loop per role of the stripe, so that a single-stripe pair (the common case) carries no
                    // boundary code at all
                    uint32_t s = 0;
#define TA_SWEEP(TAILV, BNDV, OUTV, LIMIT)                                                                               \
    for (; s + 1u < (LIMIT); s += 2u) {                                                                                  \
        iter<TAILV, BNDV, OUTV>(st, lds, lane, lane_off, bp, Z, s, cb, hb, c, T, c2, T2);                                \
        iter<TAILV, BNDV, OUTV>(st, lds, lane, lane_off, bp, Z, s + 1u, cb, hb, c2, T2, c, T);                           \
    }                                                                                                                    \
    if (s < (LIMIT)) {                                                                                                   \
        iter<TAILV, BNDV, OUTV>(st, lds, lane, lane_off, bp, Z, s, cb, hb, c, T, c2, T2);                                \
        c = c2;                                                                                                          \
        for (int q_ = 0; q_ < 3 * NWL; q_++) T[q_] = T2[q_];                                                             \
        s++;                                                                                                             \
    }
                    if (sq == 0 && last) { TA_SWEEP(false, false, false, Z.Cn); TA_SWEEP(true, false, false, steps); }
                    else if (sq == 0) { TA_SWEEP(false, false, true, steps); }
                    else if (!last) { TA_SWEEP(false, true, true, steps); }
                    else { TA_SWEEP(false, true, false, Z.Cn); TA_SWEEP(true, true, false, steps); }
#undef TA_SWEEP
                    if (last) {
                        // D[n][m] = D[i0][jlo-1] + steps right along row i0 to column m + steps down column m to row n
                        uint32_t right = m - (Z.jlo - 1u);       // row 0, or columns past the band above: +1 each
                        if (sq && phi >= Z.jlo) {
                            const uint32_t to = phi < m ? phi : m;
                            right = sum_steps(Z.inP, Z.inM, Z.jlo, to, lane) + (m - to);
                        }
                        U32 contrib = W::splat(0);
#pragma unroll
                        for (int q = 0; q < NWL; q++) {
                            const U32 first = lane * RB + 32u * (uint32_t)q;     // index of this dword's first row within the stripe
                            const U32 cnt = W::sel(first >= nrows, W::splat(0), W::sel(first + 32u <= nrows, W::splat(32), W::splat(nrows) - first));
                            const U32 msk = W::sel(cnt >= 32u, W::splat(0xFFFFFFFFu), W::shlv(W::splat(1), cnt) - 1u);
                            contrib = W::bcnt(st.Pv[q] & msk, contrib);
                            contrib = contrib - W::bcnt(st.Mv[q] & msk, W::splat(0));
                        }
                        total = anchor + right + W::wave_sum(contrib);
                    } else {
                        W::mem_fence();                          // lane 63's boundary stores -> this wave's loads in the next stripe
                    }
                    plo = Z.jlo; phi = jhi;
                }
                res = total <= P.k ? total : 0xFFFFFFFFu;        // :539-541
            }
            W::store_u32(P.out, pair, W::splat(res), lane0);
        }
    }

    // ------------------------------------------------------------------------------------------------------------
    // ONE huge pair spread over many wavefronts.  The sweep of stripe q is cut into tiles of CB steps; tile (q, kb)
    // needs the state tile (q, kb-1) left behind and the boundary line stripe q-1 has written up to the columns it
    // reads -- which, with the tile grid of stripe q shifted by off_q steps against stripe q-1 (the 63-step lane skew,
    // the 64-column read-ahead of the boundary chunks, and the band's column offset between the stripes), is complete
    // once tile (q-1, kb) has run.  So all tiles with q + kb = d are independent: the host launches d = 0, 1, 2, ...
    // on one stream (no flags, no spinning), every launch one wavefront per stripe.
    struct Huge {
        const uint8_t *ap, *bp;      // rows (the shorter string) / columns
        uint32_t n, m, u, k;
        uint32_t CB;                 // steps per tile, a multiple of 64
        uint32_t *lines;             // stripes x {HP, HN, transposition term} boundary lines of `line` u32 each
        uint64_t line;
        uint32_t *state;             // stripes x 64 lanes x 16 dwords
        uint32_t *out;
    };
    struct Stripe {
        uint32_t i0, nrows, jlo, jhi, Cn, steps, plo, phi;
        uint64_t off;                // steps by which this stripe's tile grid trails stripe 0's
    };
    static constexpr uint32_t HUGE_ROWS = 64u * RB;

    static TA_HD inline uint32_t huge_stripes(uint32_t n) { return (n + HUGE_ROWS - 1u) / HUGE_ROWS; }

    static TA_HD inline Stripe huge_stripe(uint32_t n, uint32_t m, uint32_t u, uint32_t q) {
        const uint32_t tband = (u - (m - n)) >> 1;
        const uint64_t below = tband, above = (uint64_t)tband + (m - n);
        Stripe S{};
        uint32_t plo = 1, phi = 0, jlo_prev = 1;
        uint64_t off = 0;
        for (uint32_t i = 0; i <= q; i++) {
            S.i0 = i * HUGE_ROWS;
            S.nrows = (n - S.i0 < HUGE_ROWS) ? n - S.i0 : HUGE_ROWS;
            S.jlo = ((uint64_t)S.i0 + 1u > below) ? (uint32_t)(S.i0 + 1u - below) : 1u;
            const uint64_t hi64 = (uint64_t)S.i0 + S.nrows + above;
            S.jhi = hi64 < m ? (uint32_t)hi64 : m;
            S.Cn = S.jhi - S.jlo + 1u;
            S.steps = S.Cn + (S.nrows - 1u) / RB;
            if (i) off = (off + (S.jlo - jlo_prev) + 128u + 63u) & ~(uint64_t)63;
            S.off = off; S.plo = plo; S.phi = phi;
            plo = S.jlo; phi = S.jhi; jlo_prev = S.jlo;
        }
        return S;
    }
    // number of launches: the last diagonal that holds a tile, plus one
    static TA_HD inline uint32_t huge_diagonals(uint32_t n, uint32_t m, uint32_t u, uint32_t CB) {
        const uint32_t stripes = huge_stripes(n);
        uint64_t dmax = 0;
        for (uint32_t q = 0; q < stripes; q++) {
            const Stripe S = huge_stripe(n, m, u, q);
            const uint64_t d = q + (S.off + S.steps - 1u) / CB;
            if (d > dmax) dmax = d;
        }
        return (uint32_t)(dmax + 1u);
    }

    static TA_HD inline void build_tables(uint8_t *lds, U32 lane, U32 lane_off, Ptr ap, uint32_t n, uint32_t i0) {
#pragma unroll 1
        for (uint32_t e = 0; e < TABLE_ROWS; e++) {
#pragma unroll
            for (int q = 0; q < NWL; q++) W::lds_write32(lds, lane_off + e * ROW + 4u * q, W::splat(0));
        }
        W::lds_wave_sync();
        const U32 row0 = lane * RB + i0;
#pragma unroll 1
        for (uint32_t r0 = 0; r0 < RB; r0 += 16) {
            const U32 ia = row0 + r0;
            auto piece = W::gload16(W::ptr_add(ap, ia), ia < n);
            const U32 w4[4] = {W::qword(piece, 0), W::qword(piece, 1), W::qword(piece, 2), W::qword(piece, 3)};
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const U32 ch = W::byte_of(w4[r >> 2], r & 3);
                const Bool ok = (ia + (uint32_t)r) < n;
                const uint32_t bit = 1u << ((r0 + r) & 31u), qo = 4u * ((r0 + r) >> 5);
                W::lds_or32(lds, (ch >> 5) * ROW + lane_off + qo, W::splat(bit), ok);
                W::lds_or32(lds, ((ch >> 2) & 7u) * ROW + lane_off + B_BASE + qo, W::splat(bit), ok);
                W::lds_or32(lds, (ch & 3u) * ROW + lane_off + C_BASE + qo, W::splat(bit), ok);
            }
        }
        W::lds_wave_sync();
    }

    // tile (q, d - q) of the huge pair; n >= 1, m - n <= u checked by the host
    static TA_HD inline void run_tile(const Huge &H, uint32_t q, uint32_t d, uint8_t *lds) {
        const uint32_t stripes = huge_stripes(H.n);
        if (q >= stripes || d < q) return;
        const uint32_t kb = d - q;
        const Stripe S = huge_stripe(H.n, H.m, H.u, q);
        const uint64_t lo = (uint64_t)kb * H.CB, hi = lo + H.CB;
        if (hi <= S.off) return;
        const uint32_t s_begin = lo > S.off ? (uint32_t)(lo - S.off) : 0u;
        if (s_begin >= S.steps) return;
        const uint32_t s_end = (hi - S.off < S.steps) ? (uint32_t)(hi - S.off) : S.steps;
        const bool last = (q + 1u == stripes);

        const U32 lane = W::lane();
        const U32 lane_off = lane * (NWL * 4u);
        const Bool all = (lane == lane);
        const Ptr ap = W::ptr_splat(H.ap), bp = W::ptr_splat(H.bp);
        Sweep Z;
        Z.m = H.m; Z.jlo = S.jlo; Z.Cn = S.Cn; Z.plo = S.plo; Z.phi = S.phi;
        uint32_t *rd = q ? H.lines + (uint64_t)(q - 1u) * 3u * H.line : nullptr;
        uint32_t *wr = last ? nullptr : H.lines + (uint64_t)q * 3u * H.line;
        Z.inP = rd; Z.inM = rd ? rd + H.line : nullptr; Z.inX = rd ? rd + 2u * H.line : nullptr;
        Z.outP = wr; Z.outM = wr ? wr + H.line : nullptr; Z.outX = wr ? wr + 2u * H.line : nullptr;
        Z.trace = nullptr;

        build_tables(lds, lane, lane_off, ap, H.n, S.i0);

        State st;
        U32 c;
        uint32_t *sv = H.state + (uint64_t)q * 64u * 16u;
        const U32 svi = lane * 16u;
        if (s_begin == 0) {
#pragma unroll
            for (int w = 0; w < NWL; w++) {
                st.Pv[w] = W::splat(0xFFFFFFFFu); st.Mv[w] = W::splat(0);
                st.D0p[w] = W::splat(0xFFFFFFFFu); st.Eqp[w] = W::splat(0);
            }
            st.sP = W::splat(0); st.sM = W::splat(0x80000000u); st.sX = W::splat(0); st.sc = W::splat(256);
            st.rP = W::splat(0x80000000u); st.rM = W::splat(0); st.rX = W::splat(0);
        } else {
#pragma unroll
            for (int w = 0; w < NWL; w++) {
                st.Pv[w] = W::load_u32(sv, svi + (uint32_t)w, all, 0u); st.Mv[w] = W::load_u32(sv, svi + 2u + (uint32_t)w, all, 0u);
                st.D0p[w] = W::load_u32(sv, svi + 4u + (uint32_t)w, all, 0u); st.Eqp[w] = W::load_u32(sv, svi + 6u + (uint32_t)w, all, 0u);
            }
            st.sP = W::load_u32(sv, svi + 8u, all, 0u); st.sM = W::load_u32(sv, svi + 9u, all, 0u); st.sX = W::load_u32(sv, svi + 10u, all, 0u);
            st.sc = W::load_u32(sv, svi + 11u, all, 0u); st.rP = W::load_u32(sv, svi + 12u, all, 0u); st.rM = W::load_u32(sv, svi + 13u, all, 0u);
            st.rX = W::load_u32(sv, svi + 14u, all, 0u);
        }
        const U32 colb = lane + (S.jlo + s_begin);                 // the 64 columns of b whose chunk holds step s_begin + 1
        U32 cb = W::gload_u8(W::ptr_add(bp, colb - 1u), colb <= H.m);
        if (s_begin == 0) c = W::from_lower(st.sc, W::splat(W::readlane(cb, 0)));
        else c = W::load_u32(sv, svi + 15u, all, 0u);
        U32 hb[3] = {W::splat(0x80000000u), W::splat(0), W::splat(0)};
        U32 T[3 * NWL], T2[3 * NWL], c2;
        lookup(lds, c, lane_off, T);

        uint32_t s = s_begin;
        const uint32_t lim_main = s_end < S.Cn ? s_end : S.Cn;     // steps below Cn: no lane has finished yet
#define TA_SWEEP(TAILV, BNDV, OUTV, LIMIT)                                                                               \
    for (; s + 1u < (LIMIT); s += 2u) {                                                                                  \
        iter<TAILV, BNDV, OUTV>(st, lds, lane, lane_off, bp, Z, s, cb, hb, c, T, c2, T2);                                \
        iter<TAILV, BNDV, OUTV>(st, lds, lane, lane_off, bp, Z, s + 1u, cb, hb, c2, T2, c, T);                           \
    }                                                                                                                    \
    if (s < (LIMIT)) {                                                                                                   \
        iter<TAILV, BNDV, OUTV>(st, lds, lane, lane_off, bp, Z, s, cb, hb, c, T, c2, T2);                                \
        c = c2;                                                                                                          \
        for (int q_ = 0; q_ < 3 * NWL; q_++) T[q_] = T2[q_];                                                             \
        s++;                                                                                                             \
    }
        if (q == 0 && last) { TA_SWEEP(false, false, false, lim_main); TA_SWEEP(true, false, false, s_end); }
        else if (q == 0) { TA_SWEEP(false, false, true, s_end); }
        else if (!last) { TA_SWEEP(false, true, true, s_end); }
        else { TA_SWEEP(false, true, false, lim_main); TA_SWEEP(true, true, false, s_end); }
#undef TA_SWEEP

        if (s_end < S.steps) {                                    // hand the sweep over to tile (q, kb + 1)
#pragma unroll
            for (int w = 0; w < NWL; w++) {
                W::store_u32(sv, svi + (uint32_t)w, st.Pv[w], all); W::store_u32(sv, svi + 2u + (uint32_t)w, st.Mv[w], all);
                W::store_u32(sv, svi + 4u + (uint32_t)w, st.D0p[w], all); W::store_u32(sv, svi + 6u + (uint32_t)w, st.Eqp[w], all);
            }
            W::store_u32(sv, svi + 8u, st.sP, all); W::store_u32(sv, svi + 9u, st.sM, all); W::store_u32(sv, svi + 10u, st.sX, all);
            W::store_u32(sv, svi + 11u, st.sc, all); W::store_u32(sv, svi + 12u, st.rP, all); W::store_u32(sv, svi + 13u, st.rM, all);
            W::store_u32(sv, svi + 14u, st.rX, all); W::store_u32(sv, svi + 15u, c, all);
        } else if (last) {
            // D[n][m] = D[i0][jlo-1] of the last stripe (re-anchored stripe by stripe) + steps right along row i0 to column m
            // + steps down column m to row n
            uint32_t anchor = 0;
            for (uint32_t i = 1; i <= q; i++) {
                const Stripe Si = huge_stripe(H.n, H.m, H.u, i);
                const uint32_t *lp = H.lines + (uint64_t)(i - 1u) * 3u * H.line;
                anchor += HUGE_ROWS;
                if (Si.jlo > Si.plo) anchor += sum_steps(lp, lp + H.line, Si.plo, Si.jlo - 1u, lane);
            }
            uint32_t right = H.m - (S.jlo - 1u);
            if (q && S.phi >= S.jlo) {
                const uint32_t to = S.phi < H.m ? S.phi : H.m;
                right = sum_steps(Z.inP, Z.inM, S.jlo, to, lane) + (H.m - to);
            }
            U32 contrib = W::splat(0);
#pragma unroll
            for (int w = 0; w < NWL; w++) {
                const U32 first = lane * RB + 32u * (uint32_t)w;
                const U32 cnt = W::sel(first >= S.nrows, W::splat(0), W::sel(first + 32u <= S.nrows, W::splat(32), W::splat(S.nrows) - first));
                const U32 msk = W::sel(cnt >= 32u, W::splat(0xFFFFFFFFu), W::shlv(W::splat(1), cnt) - 1u);
                contrib = W::bcnt(st.Pv[w] & msk, contrib);
                contrib = contrib - W::bcnt(st.Mv[w] & msk, W::splat(0));
            }
            const uint32_t total = anchor + right + W::wave_sum(contrib);
            W::store_u32(H.out, W::splat(0), W::splat(total <= H.k ? total : 0xFFFFFFFFu), lane == 0u);   // :539-541
        }
    }
};

}  // namespace ta
