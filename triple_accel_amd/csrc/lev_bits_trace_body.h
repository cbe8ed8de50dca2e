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
//   B. tile by tile from the last one down: restore the tile's checkpoint (asked for one tile ahead), run its TILE columns again -- the
//      same step8 -- and keep what the walk needs of EVERY column of the tile in LDS (3 words per column and lane); then every lane walks
//      its path through the tile, from where it entered it to the tile's first column, and counts the RUNS of the script as it goes -- the
//      tile's characters are in LDS too, so a diagonal step is a Match or a Mismatch on the spot and a run of equal characters is taken
//      eight at a time -- storing a run (edit type, count: one word) when it closes: last run first.
// The strings of STILE columns sit in REGISTERS (a lane's 16-byte loads on the string's own grid, through LDS once to land on the lane's
// window offset; every line of a string is touched a few times -- forwards, backwards -- instead of the records' 16 x): the recomputation
// reads them at compile-time register indices, and LDS holds per lane only the records and ONE tile's characters for the walk -- 19.3 KB per
// wavefront, 8 wavefronts per CU (with the string tile itself in LDS: 27.3 KB, 6; the kernel's time follows the residency one to one).
// A last step turns the pair's runs round and writes them as ta_edit records (lev_bits_trace.hip).
//
// The distances come from the distance kernel (the caller runs it first): a pair it answered None has no script, and the walk needs no
// absolute value.  Rows = the SHORTER string (the reference swaps, :386-390: the tie order depends on it); swap is per lane.
#pragma once
#include "lev_bits_body.h"

namespace ta {

// (LevBitsTraceParams: lev_band_body.h, next to LevParams)

// TILE: columns per checkpoint / per set of records in LDS; STILE: columns whose characters one fetch of the strings covers (a multiple
// of TILE: every 128-byte line of a string is then touched STILE / 16 times less often -- with one fetch per TILE columns the kernel read
// 90 lines per 256-byte pair, 11.5 GB per million pairs at the L2's fabric side, and waited for them)
// HAVE_CKPT: the forward sweep was the distance pass's (LevBits<.., CKPT>, rows = the shorter string as here; it left the checkpoints of tiles of 16
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
    // The strings of a string tile (STILE columns) live in REGISTERS: AD dwords of a -- a-indices [first, first + STILE + 48), first =
    // STILE T + nlo - 44: the tile's bytes, the 32 in front of them (the window is rebuilt from those) and 12 more (the walk reads the twelve
    // characters up to a[i - 1] at once) -- and PB pieces of b from 16 bytes in front of the tile.  Every column's bytes then sit at a
    // compile-time register index (the tile's number q inside the string tile is a template constant).  a's pieces are fetched on the
    // string's own 16-byte grid and pass through LDS once (BOUNCE bytes per lane, in the records' space) to land on `first`, which is per lane.
    static constexpr uint32_t PA = (44 + STILE + 15 + 15) / 16, PB = 1 + STILE / 16, RT = STILE / TILE;
    static constexpr uint32_t AD = (STILE + 48) / 4, BD = 4 * PB, BOUNCE = 16 * PA + 4;
    static_assert(15 + 4 * AD <= 16 * PA, "the pieces cover the registers' bytes");
    // the walk's characters of ONE tile, per lane in LDS: XW dwords of a from first + TILE q, YW dwords of b from STILE T - 16 + TILE q
    static constexpr uint32_t XW = (TILE + 48) / 4, YW = (TILE + 16) / 4;
    static constexpr uint32_t WSLOT = 4 * (XW + YW) + (((XW + YW) % 2u) == 0u ? 4u : 0u);   // bytes per lane (an odd number of dwords)
    // records, [word][lane]: D0 and HP (the horizontal +1 steps) of columns 0 .. TILE - 1, one word of bottom-diagonal D0 bits (bit c =
    // column c), and for the transposition test the D0 of the column in front of the tile.  Two words per column are all the walk needs:
    // with V = D[i][j] the scalar routine's candidates are sub = D[i-1][j-1] + (x != y), a_gap = D[i][j-1] + 1, b_gap = D[i-1][j] + 1 and
    // V is their minimum, so  D0 = 0 (V = diag + 1)  or  x == y (then D0 = 1, sub = V)  =>  the diagonal, whatever the gaps cost (ties go to
    // it, :493-515);  D0 = 1 and x != y (sub = V + 1)  =>  a gap: the left one iff D[i][j-1] = V - 1, i.e. HP, else the upper one (left wins
    // their tie);  the transposition (:517-532, taken on `<=`) iff its characters fit and D[i-2][j-2] + 1 = V, i.e. not both D0(i, j) and
    // D0(i-1, j-1).  (Round 5 kept the vertical differences VP / VN of every column too: 52 words per lane and tile instead of 34 -- 19.7 KB
    // of LDS per wavefront, 8 wavefronts per CU; now 15.1 KB and 10.)
    static constexpr uint32_t R_D0 = 0, R_HP = TILE, R_BOT = 2 * TILE, R_D0P = R_BOT + 1, R_WORDS = R_D0P + 1;
    static constexpr uint32_t REC_BYTES = 64 * 4 * R_WORDS > 64 * BOUNCE ? 64 * 4 * R_WORDS : 64 * BOUNCE;
    static constexpr uint32_t LDS_PER_WAVE = 64 * WSLOT + REC_BYTES;
    template <uint32_t N> using IC = std::integral_constant<uint32_t, N>;
    // f(IC<q>) for the run-time q < RT: the tile bodies exist once per q
    template <class F> static TA_HD inline void for_q(uint32_t q, F &&f) {
        if (RT > 7u && q == 7u) f(IC<(RT > 7u ? 7u : 0u)>());
        else if (RT > 6u && q == 6u) f(IC<(RT > 6u ? 6u : 0u)>());
        else if (RT > 5u && q == 5u) f(IC<(RT > 5u ? 5u : 0u)>());
        else if (RT > 4u && q == 4u) f(IC<(RT > 4u ? 4u : 0u)>());
        else if (RT > 3u && q == 3u) f(IC<(RT > 3u ? 3u : 0u)>());
        else if (RT > 2u && q == 2u) f(IC<(RT > 2u ? 2u : 0u)>());
        else if (RT > 1u && q == 1u) f(IC<(RT > 1u ? 1u : 0u)>());
        else f(IC<0u>());
    }

    // (runs_out: the lane's run count, as stored to P.n_runs -- the caller's last step needs no reload)
    static TA_HD inline void run(const LevBitsTraceParams &P, uint32_t wave_index, uint8_t *lds, U32 *runs_out = nullptr) {
        const U32 lane = W::lane();
        const U32 slot_idx = lane + wave_index * 64u;
        const Bool in_batch = slot_idx < P.n;
        const U32 pair = P.subset ? W::load_u32(P.subset, slot_idx, in_batch, 0u) : W::sel(in_batch, slot_idx, W::splat(0));
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
        uint8_t *rec = lds + 64u * WSLOT;
        const U32 wlane = lane * WSLOT, rlane = lane * 4u;
        auto raddr = [&](uint32_t w) { return rlane + w * 256u; };                       // a wave-uniform record word
        auto raddr_v = [&](const U32 &w) { return rlane + (w << 8); };                   // a per-lane one
        uint32_t *ck = P.ckpt + (uint64_t)wave_index * P.ckpt_tiles * (CK_WORDS * 64u);

        // ---- the strings of tile t (iterations [tb, tb + TILE), tb = T0 + TILE t; iteration tp slides a[tp - T0 + nlo] in and runs column
        // tp - T0 + 1 with b[tp - T0]): zeros outside the strings.  A piece outside its string is fetched from the string's first bytes
        // instead (readable: TA_BLOB_SLACK) and zeroed afterwards -- no branch around the loads.
        U32 A[AD];                                                         // a-indices first + 4 d ..
        U32 Bw[BD];                                                        // b-indices STILE T - 16 + 4 d ..
        U32 first = W::splat(0);                                           // (may be "negative": two's complement)
        auto load_strings = [&](uint32_t T) {
            const Bool all = (lane == lane);
            first = (W::splat((uint32_t)STILE * T) + nlo) - 44u;
            const U32 a_lo = first & ~15u, bounce = lane * BOUNCE;
#pragma unroll
            for (uint32_t p = 0; p < PA; p++) {
                const U32 q0 = a_lo + 16u * p;                             // a-index of the piece ("negative" in front of the string: huge, not below n)
                const Bool ok = some & (q0 < n);
                W::lds_store16(rec, bounce + 16u * p, W::qkeep(W::gload16_all(W::ptr_add(xp, W::sel(ok, q0, W::splat(0)))), ok), all);
            }
#pragma unroll
            for (uint32_t p = 0; p < PB; p++) {
                const bool front = T == 0u && p == 0u;                     // the piece in front of the string: zeros
                const uint32_t q0 = front ? 0u : (uint32_t)STILE * T - 16u + 16u * p;
                const Bool ok = front ? W::bfalse() : (some & (W::splat(q0) < m));
                const Q piece = W::qkeep(W::gload16_all(W::ptr_add(yp, W::sel(ok, W::splat(q0), W::splat(0)))), ok);
                Bw[4u * p] = W::qword(piece, 0); Bw[4u * p + 1u] = W::qword(piece, 1); Bw[4u * p + 2u] = W::qword(piece, 2); Bw[4u * p + 3u] = W::qword(piece, 3);
            }
            W::lds_wave_sync();
            const U32 src = bounce + (first & 15u);
#pragma unroll
            for (uint32_t d = 0; d < AD; d++) A[d] = W::lds_read32u(rec, src + 4u * d);
            W::lds_wave_sync();
        };

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
        // the window's bytes in front of tile q's first iteration: the 32 iterations before it slide in
        auto rebuild_window = [&](auto qc) {
            constexpr uint32_t q = decltype(qc)::value;
            const Bool all = (lane == lane);
#pragma unroll
            for (int r = 0; r < 8; r++) st.AW[r] = W::splat(0);
#pragma unroll
            for (uint32_t k = 0; k < 4u; k++) {                            // a-bytes at first + 12 + TILE q + 8 k
                // (opaque: what is computed from the registers must not be hoisted out of the tile loop, tile by tile for every q -- it was, at 450 live values)
                const U32 x0 = W::opaque(A[3u + (uint32_t)TILE * q / 4u + 2u * k]) ^ 0x0C0C0C0Cu, x1 = W::opaque(A[4u + (uint32_t)TILE * q / 4u + 2u * k]) ^ 0x0C0C0C0Cu;
                K::template step8<false, 0, false>(st, x0, x0, x0, all); K::template step8<false, 1, false>(st, x0, x0, x0, all);
                K::template step8<false, 2, false>(st, x0, x0, x0, all); K::template step8<false, 3, false>(st, x0, x0, x0, all);
                K::template step8<false, 4, false>(st, x1, x1, x1, all); K::template step8<false, 5, false>(st, x1, x1, x1, all);
                K::template step8<false, 6, false>(st, x1, x1, x1, all); K::template step8<false, 7, false>(st, x1, x1, x1, all);
            }
        };
        // the TILE columns of tile q of the string tile; REC: every column's pre-state and D0 into the records
        auto run_tile = [&](auto qc, auto rec_tag) {
            constexpr uint32_t q = decltype(qc)::value;
            constexpr bool REC = decltype(rec_tag)::value;
            const Bool all = (lane == lane);
            U32 bot = W::splat(0);
#define TA_TR_STEP(C_, bw, rw, xw)                                                                              \
                K::template step8<false, C_, true, REC>(st, bw, rw, xw, all);                                         \
                if (REC) { W::lds_write32(rec, raddr(R_D0 + c + C_), st.rD0); W::lds_write32(rec, raddr(R_HP + c + C_), st.rHP); bot = bot | W::shlv(st.rBot & 1u, W::splat(c + C_)); }
#define TA_TR_BLOCK(CB_)                                                                                         \
            if ((uint32_t)TILE > CB_) {                                                                          \
                constexpr uint32_t c = CB_, ia = 11u + ((uint32_t)TILE * q + c) / 4u, ib = 4u + ((uint32_t)TILE * q + c) / 4u;   \
                constexpr uint32_t ja = ia < AD - 1u ? ia : AD - 2u, jb = ib < BD - 1u ? ib : BD - 2u;                         \
                const U32 r0 = W::opaque(A[ja]), r1 = W::opaque(A[ja + 1u]); /* a-bytes at first + 44 + TILE q + c */             \
                const U32 x0 = r0 ^ 0x0C0C0C0Cu, x1 = r1 ^ 0x0C0C0C0Cu;                                           \
                const U32 b0 = W::opaque(Bw[jb]), b1 = W::opaque(Bw[jb + 1u]); /* b-bytes at STILE T + TILE q + c */              \
                TA_TR_STEP(0, b0, r0, x0) TA_TR_STEP(1, b0, r0, x0) TA_TR_STEP(2, b0, r0, x0) TA_TR_STEP(3, b0, r0, x0)          \
                TA_TR_STEP(4, b1, r1, x1) TA_TR_STEP(5, b1, r1, x1) TA_TR_STEP(6, b1, r1, x1) TA_TR_STEP(7, b1, r1, x1)          \
            }
            TA_TR_BLOCK(0u) TA_TR_BLOCK(8u) TA_TR_BLOCK(16u) TA_TR_BLOCK(24u)
#undef TA_TR_BLOCK
#undef TA_TR_STEP
            if (REC) W::lds_write32(rec, raddr(R_BOT), bot);
        };
        // the walk's characters of tile q into the lanes' slots
        auto stage_walk = [&](auto qc) {
            constexpr uint32_t q = decltype(qc)::value, d0 = (uint32_t)TILE * q / 4u;
#pragma unroll
            for (uint32_t w = 0; w < XW; w++) W::lds_write32(lds, wlane + 4u * w, A[d0 + w < AD ? d0 + w : AD - 1u]);
#pragma unroll
            for (uint32_t w = 0; w < YW; w++) W::lds_write32(lds, wlane + 4u * (XW + w), Bw[d0 + w < BD ? d0 + w : BD - 1u]);
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
            if constexpr (TRANS) { st.PMp[0] = ckv[2]; st.PMp[1] = ckv[3]; st.D0p[0] = ckv[4]; }
        };

        // ---- F: forwards, a checkpoint in front of every tile (HAVE_CKPT: the distance pass did it; the state behind the last tile is its
        // last checkpoint)
        init_state();
        if (!HAVE_CKPT) {
            for (uint32_t t = 0; t < tiles; t++) {
                if (t % RT == 0u) load_strings(t / RT);
                for_q(t % RT, [&](auto qc) {
                    if (t == 0u) rebuild_window(qc);
                    save_ckpt(t);
                    run_tile(qc, std::false_type());
                });
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
            // (packed form: the run goes straight into the caller's buffer, the script's last run in the last word of the pair's slot)
            const U32 at = P.packed_cap ? (pair + 1u) * P.packed_cap - 1u - nruns : pair * P.runs_cap + nruns;
            W::store_u32(P.runs, at, (cur << 29) | cnt, close & (nruns < (P.packed_cap ? P.packed_cap : P.runs_cap)));
            nruns = W::sel(close, nruns + 1u, nruns);
            cnt = W::sel(same, cnt + r, W::sel(on, r, cnt));
            cur = W::sel(on, e, cur);
        };
        uint32_t t = tiles;
        auto do_tile = [&](uint32_t q) {                                    // tile t = RT T + q
            const uint32_t j_lo = (uint32_t)TILE * t;                       // the tile's columns: j_lo + 1 .. j_lo + TILE
            take_ckpt();
            if (t > 0u) fetch_ckpt(t - 1u);
            for_q(q, [&](auto qc) {
                rebuild_window(qc);
                if (TRANS) W::lds_write32(rec, raddr(R_D0P), st.D0p[0]);
                stage_walk(qc);
                run_tile(qc, std::true_type());
            });
            // the wait for the next checkpoint HERE, where only it and the walk before this one's stores are in flight: at the next tile's
            // start it would also wait for the stores of the walk below (one counter for loads and stores, in order)
#pragma unroll
            for (uint32_t w = 0; w < CK_WORDS; w++) ckv[w] = W::opaque(ckv[w]);
            const U32 xbase = first + (uint32_t)TILE * q;                   // a-index of the walk slot's first byte
            const uint32_t ybase = j_lo - 16u;                              // b-index (mod 2^32)
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
                const U32 xo = wlane + W::sel(act, (i - 1u) - xbase, W::splat(12)), yo = wlane + 4u * XW + W::sel(act, (j - 1u) - ybase, W::splat(12));
                const U32 X2 = W::lds_read32u(lds, xo - 3u), X1 = W::lds_read32u(lds, xo - 7u), X0 = W::lds_read32u(lds, xo - 11u);    // x[i-4..i-1], x[i-8..i-5], x[i-12..i-9]
                const U32 Y2 = W::lds_read32u(lds, yo - 3u), Y1 = W::lds_read32u(lds, yo - 7u), Y0 = W::lds_read32u(lds, yo - 11u);
                const U32 d0w = W::lds_read32(rec, raddr_v(c + R_D0)), hpw = W::lds_read32(rec, raddr_v(c + R_HP)), botw = W::lds_read32(rec, raddr(R_BOT));
                const Bool first_col = c == 0u;
                const U32 d0m = TRANS ? W::lds_read32(rec, raddr_v(W::sel(first_col, W::splat(R_D0P), c + (R_D0 - 1u)))) : W::splat(0);
                const Bool bottom = bi >= 32u;                             // the 33rd diagonal: no left neighbour inside the band
                const Bool top = bi == 0u;                                 // the band's first diagonal: no upper neighbour
                const U32 sh0 = W::sel(bottom, W::splat(0), bi);
                const Bool d0 = (W::sel(bottom, W::shrv(botw, c), W::shrv(d0w, sh0)) & 1u) != 0u;
                const Bool hp = (!bottom) & ((W::shrv(hpw, sh0) & 1u) != 0u);
                const U32 x1 = X2 >> 24, y1 = Y2 >> 24;                     // x[i - 1], y[j - 1]
                const Bool eq = x1 == y1;
                // :493-515 (see R_D0 above): the diagonal unless D0 and a mismatch; then left iff HP, else up (a gap that would leave the band
                // cannot be the minimum: the diagonal stands, as the scalar routine's INF makes it)
                const Bool gap = d0 & (!eq);
                U32 code = W::sel(gap & hp, W::splat(1), W::sel(gap & (!top), W::splat(2), W::splat(0)));
                if (TRANS) {
                    const U32 x2 = (X2 >> 16) & 255u, y2 = (Y2 >> 16) & 255u;                               // x[i - 2], y[j - 2]
                    const Bool tt = (i > 1u) & (j > 1u) & (x1 == y2) & (x2 == y1);                           // :517-532
                    // D0(i-1, j-1): window bit bi of column j - 1 (the bottom bit of the column in front of the tile is not kept: a band-edge cell)
                    const U32 botm = W::sel(first_col, W::splat(0), W::shrv(botw, W::sel(first_col, W::splat(0), c - 1u)));
                    const Bool d0p = (W::sel(bottom, botm, W::shrv(d0m, sh0)) & 1u) != 0u;
                    const Bool dd_ok = (!bottom) | (!first_col);
                    code = W::sel(tt & dd_ok & (!(d0 & d0p)), W::splat(3), code);                              // D[i-2][j-2] + 1 = V
                }
                note(W::sel(code == 0u, W::sel(eq, W::splat(0), W::splat(1)), W::sel(code == 1u, e_left, W::sel(code == 2u, e_up, W::splat(4)))), W::splat(1), act);
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
        if (tiles > 0u) fetch_ckpt(tiles - 1u);
        while (t > 0u) {                                                    // string tiles from the last one down
            const uint32_t T = (t - 1u) / RT;
            load_strings(T);
            while (t > T * RT) { t--; do_tile(t - T * RT); }
        }
        // the borders: row 0 (j steps left) and column 0 (i steps up), each one run
        note(e_left, j, some & (i == 0u) & (j > 0u) & (j <= m));
        note(e_up, i, some & (j == 0u) & (i > 0u) & (i <= n));
        note(W::splat(7), W::splat(0), some & (cur != 7u));                // close the last run
        W::store_u32(P.n_runs, pair, W::sel(some, nruns, W::splat(0)), in_batch);
        if (runs_out) *runs_out = W::sel(some, nruns, W::splat(0));
    }

};

}  // namespace ta
