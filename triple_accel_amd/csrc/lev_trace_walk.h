// lev_trace_walk.h -- host walk over the records of the row-blocked bit-parallel TRACE kernel (lev_widebits_body.h).
//
// The kernel leaves 3 bits per visited cell (i, j): Pv / Mv = the vertical difference D[i][j] - D[i-1][j] is +1 / -1,
// D0 = D[i][j] == D[i-1][j-1].  From a cell's value these give the three predecessors' values
//     up   = D[i-1][j]   = V - (Pv - Mv)
//     diag = D[i-1][j-1] = D0 ? V : V - 1
//     left = D[i][j-1]   = diag + (Pv - Mv)(i, j-1)
// so the walk from (n, m) can redo the scalar path's choice at every cell with its tie order -- substitution first,
// a_gap only if smaller, b_gap only if smaller, transposition if not larger (src/levenshtein.rs:493-532) -- and emits
// the same run-length script (:561-606).  Unit costs only (the kernel's cost families).  Plain C++: the product calls
// it after copying the records to the host, the tests call it on the emulation's records.
#pragma once
#include <stdint.h>

namespace ta {

struct WbTrace {
    const uint32_t *rec;      // [stripe][column 0..trace_cols)[lane 0..63][Pv x nwl, Mv x nwl, D0 x nwl]
    uint64_t trace_cols;
    uint32_t nwl;             // dwords per lane per vector (rows per lane = 32 * nwl)
    uint32_t n, m, u;         // rows, columns, unit_k of the pass (the band the kernel visited)
};

struct WbCell {
    int v;                    // Pv - Mv
    bool d0;
    bool ok;                  // false: the kernel did not visit this cell (outside the band's columns for its stripe)
};

static inline WbCell wb_cell(const WbTrace &T, uint64_t i, uint64_t j) {   // 1 <= i <= n, 1 <= j <= m
    const uint32_t rb = 32u * T.nwl, rows = 64u * rb;
    const uint64_t q = (i - 1) / rows, r = (i - 1) % rows, lane = r / rb, bit = r % rb;
    // columns the stripe visited (lev_widebits_body.h: the pair's band, lev_plan.h)
    const uint64_t tband = (T.u - (T.m - T.n)) >> 1, below = tband, above = tband + (T.m - T.n);
    const uint64_t i0 = q * rows, nrows = (T.n - i0 < rows) ? T.n - i0 : rows;
    const uint64_t jlo = (i0 + 1 > below) ? i0 + 1 - below : 1, jhi = (i0 + nrows + above < T.m) ? i0 + nrows + above : T.m;
    WbCell c{0, false, false};
    if (j < jlo || j > jhi) return c;
    const uint32_t *p = T.rec + ((q * T.trace_cols + j) * 64u + lane) * (3u * T.nwl);
    const uint32_t w = (uint32_t)(bit >> 5), b = (uint32_t)(bit & 31);
    c.v = (int)((p[w] >> b) & 1u) - (int)((p[T.nwl + w] >> b) & 1u);
    c.d0 = ((p[2 * T.nwl + w] >> b) & 1u) != 0;
    c.ok = true;
    return c;
}

// x = rows (length n), y = columns (length m), d = D[n][m].  emit(code) is called for every step of the walk from
// (n, m) to (0, 0): 0 = (i-1, j-1) [match or mismatch], 1 = (i, j-1), 2 = (i-1, j), 3 = (i-2, j-2) [transposition].
// Returns false if the records contradict themselves (never on a correct kernel).
template <class Emit>
static inline bool wb_trace_walk(const WbTrace &T, const uint8_t *x, const uint8_t *y, uint32_t d, bool trans, Emit emit) {
    const int64_t INF = (int64_t)1 << 40;
    uint64_t i = T.n, j = T.m;
    int64_t V = d;
    while (i > 0 || j > 0) {
        if (i == 0) { emit(1); j--; V--; continue; }            // row 0: D[0][j] = j
        if (j == 0) { emit(2); i--; V--; continue; }            // column 0: D[i][0] = i
        const WbCell c = wb_cell(T, i, j);
        if (!c.ok) return false;
        const int64_t diag = c.d0 ? V : V - 1;
        int64_t up = INF, left = INF;
        if (i == 1) up = (int64_t)j;                             // D[0][j]
        else if (wb_cell(T, i - 1, j).ok) up = V - c.v;
        if (j == 1) left = (int64_t)i;                           // D[i][0]
        else { const WbCell l = wb_cell(T, i, j - 1); if (l.ok) left = diag + l.v; }
        const int64_t sub = diag + (x[i - 1] != y[j - 1] ? 1 : 0), ag = left + 1, bg = up + 1;
        const int64_t m1 = sub < ag ? sub : ag;
        int code = (bg < m1) ? 2 : ((ag < sub) ? 1 : 0);        // :493-515
        int64_t nv = bg < m1 ? bg : m1;
        int64_t tval = INF;
        if (trans && i > 1 && j > 1 && x[i - 1] == y[j - 2] && x[i - 2] == y[j - 1]) {   // :517-532
            int64_t dd;                                          // D[i-2][j-2]
            if (i == 2) dd = (int64_t)j - 2;
            else if (j == 2) dd = (int64_t)i - 2;
            else { const WbCell p = wb_cell(T, i - 1, j - 1); dd = p.ok ? (p.d0 ? diag : diag - 1) : INF; }
            tval = dd + 1;
            if (tval <= nv) { code = 3; nv = tval; }
        }
        if (nv != V) return false;
        emit(code);
        switch (code) {
            case 0: i--; j--; V = diag; break;
            case 1: j--; V = left; break;
            case 2: i--; V = up; break;
            default: i -= 2; j -= 2; V = tval - 1; break;
        }
    }
    return V == 0;
}

}  // namespace ta
