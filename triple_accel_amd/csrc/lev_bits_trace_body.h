// lev_bits_trace_body.h -- batch tracebacks for the unit-cost families WITHOUT per-cell records: checkpoints + recomputation.
//
// levenshtein_simd_k_with_opts(.., trace_on = true, ..) (src/levenshtein.rs:714-720) returns the edit script of the path the scalar routine
// takes back from (n, m) -- its argmin order :493-532, its walk :561-606.  The DP band kernel's TRACE form stores a 2-bit argmin code per
// band cell (4.1 KB per 256-byte pair at k = 32: 4.1 GB per million pairs, written once and read once -- 21 x the strings).  For
// LEVENSHTEIN_COSTS / RDAMERAU_COSTS a band column is three bit-vectors (lev_bits_body.h), and from a cell's D0 bit and the vertical
// differences next to it the scalar routine's choice can be redone exactly (lev_trace_walk.h: up = V - v(i, j), diag = D0 ? V : V - 1,
// left = diag + v(i, j - 1); only DIFFERENCES of V enter the comparisons).  So one wavefront (a pair per lane, the stride-8 window of up to
// 33 diagonals) does, in ONE kernel:
//   F. a forward sweep over the columns that keeps the column state (VP, VN; with the transposition term PM', D0') at every TILE-th
//      column: 8 bytes per pair and checkpoint, in scratch memory;
//   B. tile by tile from the last one down: restore the tile's checkpoint, run its TILE columns again -- the same step8 -- and keep what
//      the walk needs of EVERY column of the tile in LDS (3 words per column and lane); then every lane walks its path through the tile,
//      from where it entered it to the tile's first column, and counts the RUNS of the script as it goes -- the characters are in LDS, so a
//      diagonal step is a Match or a Mismatch on the spot -- storing a run (edit type, count: one word) when it closes: last run first.
// The strings go straight from memory into a per-lane LDS slot, a tile's bytes at a time (a lane's 16-byte loads; every line of a string
// is touched a few times -- forwards, backwards -- instead of the records' 16 x).  A last step turns the pair's runs round and writes them as
// ta_edit records (lev_bits_trace.hip).
//
// The distances come from the distance kernel (the caller runs it first): a pair it answered None has no script, and the walk needs no
// absolute value.  Rows = the SHORTER string (the reference swaps, :386-390: the tie order depends on it); swap is per lane.
#pragma once
#include "lev_bits_body.h"

namespace ta {

// (LevBitsTraceParams: lev_band_body.h, next to LevParams)

// TILE: columns per checkpoint / per set of records in LDS; STILE: columns whose characters one fill of the string slots covers (a multiple
// of TILE: every 128-byte line of a string is then touched STILE / 16 times less often -- with one fill per TILE columns the kernel read
// 90 lines per 256-byte pair, 11.5 GB per million pairs at the L2's fabric side, and waited for them)
// HAVE_CKPT: the forward sweep was the distance pass's (LevBits<.., CKPT>: fixed-length batches; it left the checkpoints of tiles of 16
// columns and the state behind the last column in P.ckpt) -- phase F is skipped.
template <class W, bool TRANS, int TILE = 16, int STILE = 64, bool HAVE_CKPT = false>
struct LevBitsTrace {
    static_assert(!HAVE_CKPT || TILE == 16, "the distance pass checkpoints every 16th column");
    static_assert(TILE == 8 || TILE == 16 || TILE == 32, "tiles of whole 8-column blocks");
    static_assert(STILE % TILE == 0 && STILE >= TILE && STILE <= 128, "string tiles hold whole tiles");
    using K = LevBits<W, 8, TRANS, false, false, true>;
    using State = typename K::State;
    using U32 = typename W::U32;
    using Bool = typename W::Bool;
    using Ptr = typename W::Ptr;
    using Q = typename W::Q;
    static constexpr uint32_t T0 = 64;                                   // iteration of column 1 (a multiple of 8, >= the 32 warm-up iterations)
    static constexpr uint32_t CK_WORDS = TRANS ? 5 : 2;                  // VP, VN (, PM', bottom PM', D0')
    // per-lane string slot: a = PA pieces covering a-indices [a_lo, a_lo + 16 PA) -- the tile's bytes, the 32 before them (the window is
    // rebuilt from those) and 12 more (the walk reads the twelve characters up to a[i - 1] at once); b = PB pieces from 16 bytes before the tile
    static constexpr uint32_t PA = (44 + STILE + 15 + 15) / 16, PB = 1 + STILE / 16, RT = STILE / TILE;
    static constexpr uint32_t SLOT = 16 * (PA + PB) + 4;                 // bytes per lane (an odd number of dwords)
    // records, [word][lane]: pre-column VP / VN of columns 0 .. TILE (TILE = the column behind the tile), D0 of columns 0 .. TILE - 1,
    // one word of bottom-diagonal D0 bits (bit c = column c), and for the transposition test the D0 of the column in front of the tile
    static constexpr uint32_t R_VP = 0, R_VN = TILE + 1, R_D0 = 2 * (TILE + 1), R_BOT = R_D0 + TILE, R_D0P = R_BOT + 1, R_WORDS = R_D0P + 1;
    static constexpr uint32_t LDS_PER_WAVE = 64 * SLOT + 64 * 4 * R_WORDS;

    static TA_HD inline void run(const LevBitsTraceParams &P, uint32_t wave_index, uint8_t *lds) {
        const U32 lane = W::lane();
        const U32 slot_idx = lane + wave_index * 64u;
        const Bool in_batch = slot_idx < P.n;
        const U32 pair = W::sel(in_batch, slot_idx, W::splat(0));
        Ptr xp, yp;
        U32 n, m;
        Bool swapped = W::bfalse();
        {
            Ptr ap, bp;
            U32 al, bl;
            W::load_str(P.a, pair, in_batch, ap, al);
            W::load_str(P.b, pair, in_batch, bp, bl);
            swapped = al > bl;                                             // rows = the shorter string (:386-390)
            xp = W::sel_ptr(swapped, bp, ap); yp = W::sel_ptr(swapped, ap, bp);
            n = W::sel(swapped, bl, al); m = W::sel(swapped, al, bl);
        }
        const U32 dist = W::load_u32(P.dist, pair, in_batch, 0xFFFFFFFFu);
        const Bool some = in_batch & (dist != 0xFFFFFFFFu);                // (answered: inside the band, d <= k)
        // the pair's band (lev_plan.h, as lev_bits_body.h with n <= m): diagonals j - i in [-nlo, d_hi]; window bit x <-> diagonal d_hi - x
        const U32 diff = m - n;
        const U32 tband = W::sel(some, (W::splat(P.u) - diff) >> 1, W::splat(0));
        const U32 nlo = tband + (TRANS ? 1u : 0u);
        const U32 dhi = W::splat(32u) - nlo;
        const uint32_t cols = W::wave_max(W::sel(some, m, W::splat(0)));
        const uint32_t tiles = (cols + (uint32_t)TILE - 1u) / (uint32_t)TILE;
        uint8_t *rec = lds + 64u * SLOT;
        const U32 slot = lane * SLOT, rlane = lane * 4u;
        auto raddr = [&](uint32_t w) { return rlane + w * 256u; };                       // a wave-uniform record word
        auto raddr_v = [&](const U32 &w) { return rlane + (w << 8); };                   // a per-lane one
        uint32_t *ck = P.ckpt + (uint64_t)wave_index * P.ckpt_tiles * (CK_WORDS * 64u);

        // ---- the strings of tile t (iterations [tb, tb + TILE), tb = T0 + TILE t; iteration tp slides a[tp - T0 + nlo] in and runs column
        // tp - T0 + 1 with b[tp - T0]): pieces on the strings' own 16-byte grids, zeros outside the strings
        U32 a_lo = W::splat(0);                                            // a-index of the slot's first byte (may be "negative": two's complement)
        uint32_t b_lo = 0, loaded = 0xFFFFFFFFu;                           // b-index of the b slot's first byte; the string tile the slots hold
        // fetch: the pieces of string tile T into registers (in flight until commit needs them -- phase B asks for tile T - 1 as soon as tile T
        // sits in the slots: its latency hides behind RT tiles of work); commit: into the slots.  A piece outside its string is fetched from
        // the string's first bytes instead (readable: TA_BLOB_SLACK) and zeroed on its way into the slot -- no branch around the loads, so
        // they write the registers that wait for the commit.
        auto a_piece = [&](uint32_t T, uint32_t p, U32 &q0) {             // a-index of piece p ("negative" in front of the string: huge, not below n)
            q0 = (((W::splat((uint32_t)STILE * T) + nlo) - 44u) & ~15u) + 16u * p;
            return some & (q0 < n);
        };
        auto b_piece = [&](uint32_t T, uint32_t p, uint32_t &q0) {        // (the piece in front of the string: zeros)
            const bool front = T == 0u && p == 0u;
            q0 = front ? 0u : (uint32_t)STILE * T - 16u + 16u * p;
            return front ? W::bfalse() : (some & (W::splat(q0) < m));
        };
        auto fetch_strings = [&](uint32_t T, Q (&sa)[PA], Q (&sb)[PB]) {
#pragma unroll
            for (uint32_t p = 0; p < PA; p++) {
                U32 q0;
                const Bool ok = a_piece(T, p, q0);
                sa[p] = W::gload16_all(W::ptr_add(xp, W::sel(ok, q0, W::splat(0))));
            }
#pragma unroll
            for (uint32_t p = 0; p < PB; p++) {
                uint32_t q0;
                const Bool ok = b_piece(T, p, q0);
                sb[p] = W::gload16_all(W::ptr_add(yp, W::sel(ok, W::splat(q0), W::splat(0))));
            }
        };
        auto commit_strings = [&](uint32_t T, const Q (&sa)[PA], const Q (&sb)[PB]) {
            loaded = T;
            const Bool all = (lane == lane);
            a_lo = ((W::splat((uint32_t)STILE * T) + nlo) - 44u) & ~15u;
            b_lo = (uint32_t)STILE * T - 16u;
#pragma unroll
            for (uint32_t p = 0; p < PA; p++) {
                U32 q0;
                const Bool ok = a_piece(T, p, q0);
                W::lds_store16(lds, slot + 16u * p, W::qkeep(sa[p], ok), all);
            }
#pragma unroll
            for (uint32_t p = 0; p < PB; p++) {
                uint32_t q0;
                const Bool ok = b_piece(T, p, q0);
                W::lds_store16(lds, slot + 16u * (PA + p), W::qkeep(sb[p], ok), all);
            }
            W::lds_wave_sync();
        };
        auto load_strings = [&](uint32_t T) {                               // (phase F: fetch and commit on the spot)
            if (T == loaded) return;
            Q fa[PA], fb[PB];
            fetch_strings(T, fa, fb);
            commit_strings(T, fa, fb);
        };
        // LDS addresses of the bytes of iteration tp (a) / of b[tp - T0]
        auto a_addr = [&](uint32_t tp) { return slot + (((W::splat(tp - T0) + nlo) - a_lo)); };
        auto b_addr = [&](uint32_t tp) { return slot + 16u * PA + ((tp - T0) - b_lo); };

        State st;
        auto init_state = [&]() {
            // column 0, D[r][0] = |r|: rows r = 1 - d_hi + x >= 1 step up (+1), rows <= 0 step down (-1)
            const U32 below = W::sel(dhi >= 32u, W::splat(0xFFFFFFFFu), W::shlv(W::splat(1), dhi) - 1u);
            st.VN[0] = below; st.VP[0] = ~below;
            st.VN[1] = W::splat(0); st.VP[1] = W::splat(0);
            st.PMp[0] = W::splat(0); st.PMp[1] = W::splat(0);
            st.D0p[0] = W::splat(0xFFFFFFFFu); st.D0p[1] = W::splat(1);
            st.acc = W::splat(0);
        };
        // the window's bytes in front of iteration tb: the 32 iterations before it slide in (the slot holds them)
        auto rebuild_window = [&](uint32_t tb) {
            const Bool all = (lane == lane);
#pragma unroll
            for (int r = 0; r < 8; r++) st.AW[r] = W::splat(0);
            for (uint32_t tp = tb - 32u; tp < tb; tp += 8u) {
                const U32 pa = a_addr(tp);
                const U32 x0 = W::lds_read32u(lds, pa) ^ 0x0C0C0C0Cu, x1 = W::lds_read32u(lds, pa + 4u) ^ 0x0C0C0C0Cu;
                K::template step8<false, 0, false>(st, x0, x0, x0, all); K::template step8<false, 1, false>(st, x0, x0, x0, all);
                K::template step8<false, 2, false>(st, x0, x0, x0, all); K::template step8<false, 3, false>(st, x0, x0, x0, all);
                K::template step8<false, 4, false>(st, x1, x1, x1, all); K::template step8<false, 5, false>(st, x1, x1, x1, all);
                K::template step8<false, 6, false>(st, x1, x1, x1, all); K::template step8<false, 7, false>(st, x1, x1, x1, all);
            }
        };
        // the TILE columns of tile t; REC: every column's pre-state and D0 into the records
        auto run_tile = [&](uint32_t t, auto rec_tag) {
            constexpr bool REC = decltype(rec_tag)::value;
            const Bool all = (lane == lane);
            const uint32_t tb = T0 + (uint32_t)TILE * t;
            U32 bot = W::splat(0);
            for (uint32_t c = 0; c < (uint32_t)TILE; c += 8u) {
                const uint32_t tp = tb + c;
                const U32 pa = a_addr(tp), pb = W::splat(0) + b_addr(tp);
                const U32 r0 = W::lds_read32u(lds, pa), r1 = W::lds_read32u(lds, pa + 4u);
                const U32 x0 = r0 ^ 0x0C0C0C0Cu, x1 = r1 ^ 0x0C0C0C0Cu;
                const U32 b0 = W::lds_read32u(lds, pb), b1 = W::lds_read32u(lds, pb + 4u);
#define TA_TR_STEP(C_, bw, rw, xw)                                                                              \
                if (REC) { W::lds_write32(rec, raddr(R_VP + c + C_), st.VP[0]); W::lds_write32(rec, raddr(R_VN + c + C_), st.VN[0]); } \
                K::template step8<false, C_, true, REC>(st, bw, rw, xw, all);                                         \
                if (REC) { W::lds_write32(rec, raddr(R_D0 + c + C_), st.rD0); bot = bot | W::shlv(st.rBot & 1u, W::splat(c + C_)); }
                TA_TR_STEP(0, b0, r0, x0) TA_TR_STEP(1, b0, r0, x0) TA_TR_STEP(2, b0, r0, x0) TA_TR_STEP(3, b0, r0, x0)
                TA_TR_STEP(4, b1, r1, x1) TA_TR_STEP(5, b1, r1, x1) TA_TR_STEP(6, b1, r1, x1) TA_TR_STEP(7, b1, r1, x1)
#undef TA_TR_STEP
            }
            if (REC) W::lds_write32(rec, raddr(R_BOT), bot);
        };
        auto save_ckpt = [&](uint32_t t) {
            uint32_t *c = ck + (uint64_t)t * (CK_WORDS * 64u);
            W::store_u32(c, lane, st.VP[0], lane == lane); W::store_u32(c + 64, lane, st.VN[0], lane == lane);
            if (TRANS) {
                W::store_u32(c + 128, lane, st.PMp[0], lane == lane); W::store_u32(c + 192, lane, st.PMp[1], lane == lane);
                W::store_u32(c + 256, lane, st.D0p[0], lane == lane);
            }
        };
        // the checkpoint in front of tile t: asked for one tile ahead (phase B), taken into the state when the tile starts
        U32 ckv[CK_WORDS];
        auto fetch_ckpt = [&](uint32_t t) {
            const uint32_t *c = ck + (uint64_t)t * (CK_WORDS * 64u);
#pragma unroll
            for (uint32_t w = 0; w < CK_WORDS; w++) ckv[w] = W::load_u32(c + 64u * w, lane, lane == lane, 0u);
        };
        auto take_ckpt = [&]() {
            st.VP[0] = ckv[0]; st.VN[0] = ckv[1];
            if (TRANS) { st.PMp[0] = ckv[2]; st.PMp[1] = ckv[3]; st.D0p[0] = ckv[4]; }
        };

        // ---- F: forwards, a checkpoint in front of every tile (HAVE_CKPT: the distance pass did it; the state behind the last tile is its
        // last checkpoint)
        init_state();
        if (HAVE_CKPT) {
            fetch_ckpt(tiles); take_ckpt();
        } else {
            for (uint32_t t = 0; t < tiles; t++) {
                load_strings(t / RT);
                if (t == 0) rebuild_window(T0);
                save_ckpt(t);
                run_tile(t, std::false_type());
            }
        }
        // ---- B: backwards, tile by tile
        U32 i = W::sel(some, n, W::splat(0)), j = W::sel(some, m, W::splat(0));
        // the script's runs, last run first: r steps of edit e extend the open run or close it (one word to the pair's run list) and open another
        const U32 e_left = W::sel(swapped, W::splat(3), W::splat(2)), e_up = W::sel(swapped, W::splat(2), W::splat(3));   // AGap = 2, BGap = 3 (:561-606, relabelled under the swap)
        U32 cur = W::splat(7), cnt = W::splat(0), nruns = W::splat(0);
        auto note = [&](const U32 &e, const U32 &r, const Bool &on) {
            const Bool same = on & (e == cur);
            const Bool close = on & !same & (cur != 7u);
            W::store_u32(P.runs, pair * P.runs_cap + nruns, (cur << 29) | cnt, close & (nruns < P.runs_cap));
            nruns = W::sel(close, nruns + 1u, nruns);
            cnt = W::sel(same, cnt + r, W::sel(on, r, cnt));
            cur = W::sel(on, e, cur);
        };
        U32 nxt_vp = st.VP[0], nxt_vn = st.VN[0];                          // the pre-state of the column behind the last tile
        // string tiles from the last one down; the tile in front is in flight (registers) while this one's RT tiles are worked on.  It is
        // asked for unconditionally (tile 0 asks for itself again), so that the loads write the registers the next commit reads, and inside
        // the string tile's FIRST tile, behind the wait for that tile's checkpoint and in front of the request for the next one: every
        // later wait finds loads that have had a whole tile's work to arrive.
        Q sa[PA], sb[PB];
        uint32_t T = tiles > 0u ? (tiles - 1u) / RT : 0u, t = tiles;
        auto do_tile = [&](auto ahead_tag) {
            t--;
            const uint32_t tb = T0 + (uint32_t)TILE * t, j_lo = (uint32_t)TILE * t;      // the tile's columns: j_lo + 1 .. j_lo + TILE
            take_ckpt();
            if (decltype(ahead_tag)::value) fetch_strings(T > 0u ? T - 1u : 0u, sa, sb);
            if (t > 0u) fetch_ckpt(t - 1u);
            rebuild_window(tb);
            W::lds_write32(rec, raddr(R_D0P), TRANS ? st.D0p[0] : W::splat(0));
            W::lds_write32(rec, raddr(R_VP + TILE), nxt_vp); W::lds_write32(rec, raddr(R_VN + TILE), nxt_vn);
            nxt_vp = st.VP[0]; nxt_vn = st.VN[0];                          // (this tile's first pre-state is the tile before's "behind")
            run_tile(t, std::true_type());
            W::lds_wave_sync();
            Bool act = some & (j > j_lo) & (j <= j_lo + (uint32_t)TILE) & (i > 0u);
            if (P.runs_cap == 0u) act = W::bfalse();            // (a timing probe, TA_TRACE_SKIP_WALK=1: the recomputation without the walk -- no scripts)
            while (W::any(act)) {
                // One iteration = ONE general step at (i, j) and the run of matches behind it, on ONE round trip to LDS: the cell's records and
                // the twelve characters of each string up to x[i-1] / y[j-1] are requested together.  (Where x[i-1] == y[j-1] the scalar routine
                // takes the diagonal whatever the neighbours hold -- sub = diag is never above a_gap or b_gap: adjacent cells differ by at most
                // one -- and a transposition of four equal characters costs one more: a run of equal characters is a run of diagonal steps, up
                // to eight of them per iteration without a look at a record.)
                // (lanes that are not walking read the tile's first record and the slot's first bytes: every address stays inside the block)
                const U32 c = W::sel(act, (j - j_lo) - 1u, W::splat(0));   // column within the tile
                const U32 bi = W::sel(act, (i + dhi) - j, W::splat(1));    // window bit of row i at column j: 0 .. 32
                const U32 xo = slot + W::sel(act, (i - 1u) - a_lo, W::splat(12)), yo = slot + 16u * PA + W::sel(act, (j - 1u) - b_lo, W::splat(12));
                const U32 X2 = W::lds_read32u(lds, xo - 3u), X1 = W::lds_read32u(lds, xo - 7u), X0 = W::lds_read32u(lds, xo - 11u);    // x[i-4..i-1], x[i-8..i-5], x[i-12..i-9]
                const U32 Y2 = W::lds_read32u(lds, yo - 3u), Y1 = W::lds_read32u(lds, yo - 7u), Y0 = W::lds_read32u(lds, yo - 11u);
                const U32 d0w = W::lds_read32(rec, raddr_v(c + R_D0)), botw = W::lds_read32(rec, raddr(R_BOT));
                const U32 vp1 = W::lds_read32(rec, raddr_v(c + (R_VP + 1u))), vn1 = W::lds_read32(rec, raddr_v(c + (R_VN + 1u)));
                const U32 vp0 = W::lds_read32(rec, raddr_v(c + R_VP)), vn0 = W::lds_read32(rec, raddr_v(c + R_VN));
                const Bool first_col = c == 0u;
                const U32 d0m = TRANS ? W::lds_read32(rec, raddr_v(W::sel(first_col, W::splat(R_D0P), c + (R_D0 - 1u)))) : W::splat(0);
                const U32 d0 = W::sel(bi >= 32u, W::shrv(botw, c), W::shrv(d0w, bi)) & 1u;
                // v(i, j): the pre-state of column j + 1 at bit bi - 1 (bi = 0: the row above is outside the window)
                const Bool up_ok = bi >= 1u;
                const U32 sh1 = W::sel(up_ok, bi - 1u, W::splat(0));
                const U32 v_ij = (W::shrv(vp1, sh1) & 1u) - (W::shrv(vn1, sh1) & 1u);
                // v(i, j - 1): the pre-state of column j at bit bi (bi = 32: the bottom diagonal has no left neighbour)
                const Bool left_ok = bi <= 31u;
                const U32 sh0 = W::sel(left_ok, bi, W::splat(0));
                const U32 v_l = (W::shrv(vp0, sh0) & 1u) - (W::shrv(vn0, sh0) & 1u);
                // values relative to V = D[i][j], biased by BIAS so that they stay unsigned
                constexpr uint32_t BIAS = 8u, INF = 64u;
                const U32 diag = W::splat(BIAS) - (d0 ^ 1u);
                const U32 up = W::sel(up_ok, W::splat(BIAS) - v_ij, W::splat(INF));
                const U32 left = W::sel(left_ok, diag + v_l, W::splat(INF));
                const U32 x1 = X2 >> 24, y1 = Y2 >> 24;                     // x[i - 1], y[j - 1]
                const U32 sub = diag + W::sel(x1 == y1, W::splat(0), W::splat(1)), ag = left + 1u, bg = up + 1u;
                const U32 m1 = W::umin(sub, ag);
                U32 code = W::sel(bg < m1, W::splat(2), W::sel(ag < sub, W::splat(1), W::splat(0)));      // :493-515
                if (TRANS) {
                    const U32 x2 = (X2 >> 16) & 255u, y2 = (Y2 >> 16) & 255u;                               // x[i - 2], y[j - 2]
                    const Bool tt = (i > 1u) & (j > 1u) & (x1 == y2) & (x2 == y1);                           // :517-532
                    // D[i-2][j-2] = D0(i-1, j-1) ? diag : diag - 1; (i-1, j-1) is window bit bi of column j - 1
                    const U32 botm = W::sel(first_col, W::splat(0), W::shrv(botw, W::sel(first_col, W::splat(0), c - 1u)));
                    const U32 d0p = W::sel(bi >= 32u, botm, W::shrv(d0m, sh0)) & 1u;
                    const Bool dd_ok = (bi <= 31u) | !first_col;          // (the bottom bit of the column in front of the tile is not kept: a band-edge cell)
                    const U32 tval = W::sel(dd_ok, (diag - (d0p ^ 1u)) + 1u, W::splat(INF));
                    const U32 nv = W::umin(bg, m1);
                    code = W::sel(tt & (tval <= nv), W::splat(3), code);
                }
                note(W::sel(code == 0u, W::sel(x1 == y1, W::splat(0), W::splat(1)), W::sel(code == 1u, e_left, W::sel(code == 2u, e_up, W::splat(4)))), W::splat(1), act);
                const U32 two = W::sel(code == 3u, W::splat(2), W::splat(1));
                const U32 di = W::sel(code != 1u, two, W::splat(0)), dj = W::sel(code != 2u, two, W::splat(0));
                const U32 i1 = i - di, j1 = j - dj;
                // the run of matches behind the step: the strings' last eight characters in front of (i1, j1), out of the twelve that were read
                const U32 xh = W::sel(di == 0u, X2, W::alignbyte_v(X2, X1, W::splat(4) - di)), xl = W::sel(di == 0u, X1, W::alignbyte_v(X1, X0, W::splat(4) - di));
                const U32 yh = W::sel(dj == 0u, Y2, W::alignbyte_v(Y2, Y1, W::splat(4) - dj)), yl = W::sel(dj == 0u, Y1, W::alignbyte_v(Y1, Y0, W::splat(4) - dj));
                const U32 dh = xh ^ yh, dl = xl ^ yl;
                U32 r = W::sel(dh == 0u, W::splat(4) + (W::clz(dl) >> 3), W::clz(dh) >> 3);      // equal characters from x[i1-1] / y[j1-1] downwards
                r = W::umin(r, W::sel(j1 > j_lo, W::umin(i1, j1 - j_lo), W::splat(0)));
                const Bool fast = act & (r > 0u);
                note(W::splat(0), r, fast);                                    // r Matches
                i = W::sel(act, i1 - W::sel(fast, r, W::splat(0)), i);
                j = W::sel(act, j1 - W::sel(fast, r, W::splat(0)), j);
                act = act & (j > j_lo) & (j <= j_lo + (uint32_t)TILE) & (i > 0u) & (i <= n);
            }
        };
        if (tiles > 0u) { fetch_ckpt(tiles - 1u); fetch_strings(T, sa, sb); }
        while (t > 0u) {
            commit_strings(T, sa, sb);
            const uint32_t t_lo = T * RT;
            do_tile(std::true_type());
            while (t > t_lo) do_tile(std::false_type());
            T = T > 0u ? T - 1u : 0u;
        }
        // the borders: row 0 (j steps left) and column 0 (i steps up), each one run
        note(e_left, j, some & (i == 0u) & (j > 0u) & (j <= m));
        note(e_up, i, some & (j == 0u) & (i > 0u) & (i <= n));
        note(W::splat(7), W::splat(0), some & (cur != 7u));                // close the last run
        W::store_u32(P.n_runs, pair, W::sel(some, nruns, W::splat(0)), in_batch);
    }

};

}  // namespace ta
