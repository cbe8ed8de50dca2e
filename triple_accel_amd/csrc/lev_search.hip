// lev_search.hip -- gfx950 kernels for levenshtein_search / hamming_search over a haystack shard in HBM.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdlib.h>

#include "ham_search_body.h"
#include "ham_swar_body.h"
#include "ham_bits_body.h"
#include "ham_phase_body.h"
#include "lev_filter_body.h"
#include "lev_search_body.h"
#include "lev_search_wave_body.h"
#include "ta_internal.h"

namespace ta {

template <int N, bool TRANS, bool PACKED>
__global__ __launch_bounds__(256) void lev_search_kernel(SearchParams P) {
    const uint64_t tile = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t emit_begin = tile * P.tile;
    if (emit_begin >= P.hay_len) return;
    uint64_t emit_end = emit_begin + P.tile;
    if (emit_end > P.hay_len) emit_end = P.hay_len;
    const uint64_t col_begin = emit_begin > P.halo ? emit_begin - P.halo : 0;
    SearchCosts C{P.k, P.mc, P.gc, P.sg, P.tc, P.anchored};
    ta_match *hits = P.hits;
    unsigned long long *count = P.count;
    const uint64_t base = P.base, emit_from = P.emit_from, cap = P.cap;
    auto emit = [=](uint64_t end, uint32_t len, uint32_t cost) {
        const uint64_t gend = base + end;
        if (gend <= emit_from) return;
        unsigned long long idx = atomicAdd(count, 1ull);
        if (idx < cap) hits[idx] = ta_match{gend - len, gend, cost, 0u};
    };
    if (PACKED) lev_search_tile_packed<N, TRANS>(P.hay, P.needle, P.needle_len, C, col_begin, emit_begin, emit_end, emit);
    else lev_search_tile<N, TRANS>(P.hay, P.needle, P.needle_len, C, col_begin, emit_begin, emit_end, emit);
}

// long needles: the column lives in HBM scratch, element-major so that a wavefront's accesses coalesce
__global__ __launch_bounds__(256) void lev_search_mem_kernel(SearchParams P, uint64_t tiles) {
    const uint64_t tile = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= tiles) return;
    const uint64_t emit_begin = tile * P.tile;
    if (emit_begin >= P.hay_len) return;
    uint64_t emit_end = emit_begin + P.tile;
    if (emit_end > P.hay_len) emit_end = P.hay_len;
    const uint64_t col_begin = emit_begin > P.halo ? emit_begin - P.halo : 0;
    SearchCosts C{P.k, P.mc, P.gc, P.sg, P.tc, P.anchored};
    ta_match *hits = P.hits;
    unsigned long long *count = P.count;
    const uint64_t base = P.base, emit_from = P.emit_from, cap = P.cap;
    lev_search_tile_mem(P.hay, P.needle_dev, P.needle_len, C, P.tc != 0, P.col_scratch + tile, tiles,
                        col_begin, emit_begin, emit_end,
                        [=](uint64_t end, uint32_t len, uint32_t cost) {
                            const uint64_t gend = base + end;
                            if (gend <= emit_from) return;
                            unsigned long long idx = atomicAdd(count, 1ull);
                            if (idx < cap) hits[idx] = ta_match{gend - len, gend, cost, 0u};
                        });
}

template <int N>
static hipError_t launch_n(const SearchParams &P, bool trans, bool packed, uint32_t grid, hipStream_t s) {
    if (packed) {
        if (trans) hipLaunchKernelGGL((lev_search_kernel<N, true, true>), dim3(grid), dim3(256), 0, s, P);
        else hipLaunchKernelGGL((lev_search_kernel<N, false, true>), dim3(grid), dim3(256), 0, s, P);
    } else if constexpr (N % 8 == 0) {
        if (trans) hipLaunchKernelGGL((lev_search_kernel<N, true, false>), dim3(grid), dim3(256), 0, s, P);
        else hipLaunchKernelGGL((lev_search_kernel<N, false, false>), dim3(grid), dim3(256), 0, s, P);
    }
    return hipGetLastError();
}

// packed: cost and length in one VGPR (see lev_search_tile_packed for the validity conditions, checked by the caller)
hipError_t lev_search_launch(const SearchParams &P, bool packed, bool trans, hipStream_t s) {
    if (P.hay_len == 0) return hipSuccess;
    set_last_kernel_name("lev_search%s_kernel", P.needle_len <= 32 ? "" : "_mem");
    const uint64_t tiles = (P.hay_len + P.tile - 1) / P.tile;
    const uint32_t grid = (uint32_t)((tiles + 255) / 256);
    const uint32_t n = P.needle_len;
    if (packed && n <= 32) {              // one instantiation per needle length: straight-line column code
        switch (n) {
#define TA_N(x) case x: return launch_n<x>(P, trans, true, grid, s);
            TA_N(1) TA_N(2) TA_N(3) TA_N(4) TA_N(5) TA_N(6) TA_N(7) TA_N(8) TA_N(9) TA_N(10) TA_N(11) TA_N(12)
            TA_N(13) TA_N(14) TA_N(15) TA_N(16) TA_N(17) TA_N(18) TA_N(19) TA_N(20) TA_N(21) TA_N(22) TA_N(23) TA_N(24)
            TA_N(25) TA_N(26) TA_N(27) TA_N(28) TA_N(29) TA_N(30) TA_N(31) TA_N(32)
#undef TA_N
        }
    }
    if (n <= 8) return launch_n<8>(P, trans, false, grid, s);
    if (n <= 16) return launch_n<16>(P, trans, false, grid, s);
    if (n <= 24) return launch_n<24>(P, trans, false, grid, s);
    if (n <= 32) return launch_n<32>(P, trans, false, grid, s);
    hipLaunchKernelGGL(lev_search_mem_kernel, dim3(grid), dim3(256), 0, s, P, tiles);
    return hipGetLastError();
}

// ---- candidate filter + exact kernel on the flagged blocks (unit-cost families, needle <= 32 bytes)

// One lane scans P.tile end positions (a multiple of 64) after P.halo bytes of left context and appends the index of
// every 64-column block that holds a cost <= k to `list` (lev_filter_body.h).  The 256-entry match table of the
// needle lives in LDS; the haystack is read 16 bytes per lane per load.
// REPL: the match table is kept 64 times -- row c (256 bytes) holds peq[c] once per lane of a wavefront, lane l reads dword l of
// the row -- so the 64 lookups of a wavefront fall on 64 different addresses of which no two in a 32-lane group share a bank
// (ds_read_b32: bank = dword index mod 32), whatever the haystack bytes are; one flat 256-entry table indexed by a random byte
// per lane costs ~4.3 extra LDS cycles per lookup (SQ_LDS_BANK_CONFLICT 72.6 M per GiB, profiles/r02).  The address is built
// by ONE v_perm_b32 per byte -- [lane*4 | c << 8] -- where the flat table needs one SDWA shift: the VALU count is unchanged.
template <bool TRANS, bool REPL, bool EXACT = false>
__global__ __launch_bounds__(REPL ? 512 : 256) void lev_filter_kernel(SearchParams P, uint32_t *list, uint32_t list_cap, unsigned int *list_count) {
    __shared__ __attribute__((aligned(16))) uint32_t peq[REPL ? 256 * 64 : 256];
    if (REPL) {                                    // 512 threads: two per row, half a row each (two workgroups = 16 wavefronts per CU)
        const uint32_t m = lev_filter_peq(P.needle, P.needle_len, threadIdx.x >> 1);
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 *row = (u32x4 *)(peq + (threadIdx.x >> 1) * 64 + (threadIdx.x & 1u) * 32);
#pragma unroll
        for (int q = 0; q < 8; q++) row[q] = u32x4{m, m, m, m};
    } else {
        peq[threadIdx.x] = lev_filter_peq(P.needle, P.needle_len, threadIdx.x);
    }
    __syncthreads();
    const uint32_t lane_off = (threadIdx.x & 63u) * 4u;
    // table entry of byte b (0..3) of dword v
    auto lookup = [&](uint32_t v, int b) -> uint32_t {
        if (REPL) {
            const uint32_t a = __builtin_amdgcn_perm(v, lane_off, 0x0C0C0000u | ((4u + (uint32_t)b) << 8));   // lane*4 | byte b << 8
            return *(const uint32_t *)((const uint8_t *)peq + a);
        }
        return peq[(v >> (8 * b)) & 0xffu];
    };
    const uint64_t tile = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t emit_begin = tile * P.tile;
    if (emit_begin >= P.hay_len) return;
    uint64_t emit_end = emit_begin + P.tile;
    if (emit_end > P.hay_len) emit_end = P.hay_len;
    const uint64_t col_begin = emit_begin > P.halo ? emit_begin - P.halo : 0;
    const uint32_t k = P.k;
    const uint8_t *hay = P.hay;
    FilterState st;
    lev_filter_reset(st, P.needle_len);
    for (uint64_t i = col_begin; i < emit_begin; i++) lev_filter_step<TRANS>(st, lookup(hay[i], 0));   // left context
    auto flag = [&](uint64_t col) {
        const unsigned int idx = atomicAdd(list_count, 1u);
        if (idx < list_cap) list[idx] = (uint32_t)(col / FILTER_BLOCK);
    };
    typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(1)));
    uint64_t i = emit_begin;
    const uint64_t full_end = emit_begin + ((emit_end - emit_begin) & ~(uint64_t)(FILTER_BLOCK - 1));
    // a whole 128-byte LINE per lane is requested at once (eight 16-byte loads in one burst, tiles start on 128-byte
    // multiples), one line ahead: every line of the haystack crosses the L2's fabric side once (two 64-byte bursts 64 columns
    // apart measured 1.31x: the second half had left the 4 MB L2 by the time its lane came back for it)
    constexpr uint32_t LINE = 2 * FILTER_BLOCK;
    u32x4u nxt[8];
#pragma unroll
    for (int q = 0; q < 8; q++) nxt[q] = (i + 16u * q < emit_end) ? *(const u32x4u *)(hay + i + 16u * q) : u32x4u{0, 0, 0, 0};
    while (i < full_end) {                                        // whole 64-column blocks, two per line
        u32x4u cur[8];
#pragma unroll
        for (int q = 0; q < 8; q++) cur[q] = nxt[q];
#pragma unroll
        for (int q = 0; q < 8; q++)                               // blobs carry 16 bytes of slack
            if (i + LINE + 16u * q < emit_end) nxt[q] = *(const u32x4u *)(hay + i + LINE + 16u * q);
#pragma unroll 1
        for (int blk = 0; blk < 2 && i < full_end; blk++) {
            bool any = false;
            if (EXACT) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const u32x4u v = blk == 0 ? cur[q] : cur[4 + q];
#pragma unroll
                    for (int b = 0; b < 16; b++) any |= lev_filter_step<TRANS>(st, lookup(v[b >> 2], b & 3)) <= k;
                }
            } else {            // the score settled per 32 columns: a lower bound of the block's smallest cost (lev_filter_fold32)
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    uint32_t MH = 0;
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const u32x4u v = blk == 0 ? cur[2 * half + q] : cur[4 + 2 * half + q];
#pragma unroll
                        for (int b = 0; b < 16; b++) lev_filter_step_h<TRANS>(st, lookup(v[b >> 2], b & 3), MH);
                    }
                    any |= lev_filter_fold32(st, MH, k);
                }
            }
            if (any) flag(i);
            i += FILTER_BLOCK;
        }
    }
    if (i < emit_end) {                                           // the shard's last, partial block
        bool any = false;
        for (uint64_t j = i; j < emit_end; j++) any |= lev_filter_step<TRANS>(st, lookup(hay[j], 0)) <= k;
        if (any) flag(i);
    }
}

// needles of 33..512 bytes: NWF-dword vectors, match table (256 x NWF dwords) in LDS built from the device copy of the needle
template <int NWF, bool TRANS>
__global__ __launch_bounds__(256) void lev_filter_kernel_n(SearchParams P, uint32_t *list, uint32_t list_cap, unsigned int *list_count) {
    __shared__ uint32_t peq[256 * NWF];
    for (int w = 0; w < NWF; w++) peq[threadIdx.x * NWF + w] = lev_filter_peq_word(P.needle_dev, P.needle_len, NWF, threadIdx.x, w);
    __syncthreads();
    const uint64_t tile = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t emit_begin = tile * P.tile;
    if (emit_begin >= P.hay_len) return;
    uint64_t emit_end = emit_begin + P.tile;
    if (emit_end > P.hay_len) emit_end = P.hay_len;
    const uint64_t col_begin = emit_begin > P.halo ? emit_begin - P.halo : 0;
    const uint32_t k = P.k;
    const uint8_t *hay = P.hay;
    FilterStateN<NWF> st;
    lev_filter_reset_n<NWF>(st, P.needle_len);
    auto stepc = [&](uint32_t c) -> uint32_t {
        uint32_t Eq[NWF];
#pragma unroll
        for (int w = 0; w < NWF; w++) Eq[w] = peq[c * NWF + w];
        return lev_filter_step_n<NWF, TRANS>(st, Eq);
    };
    auto flag = [&](uint64_t col) {
        const unsigned int idx = atomicAdd(list_count, 1u);
        if (idx < list_cap) list[idx] = (uint32_t)(col / FILTER_BLOCK);
    };
    for (uint64_t i = col_begin; i < emit_begin; i++) stepc(hay[i]);               // left context
    typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(1)));
    uint64_t i = emit_begin;
    const uint64_t full_end = emit_begin + ((emit_end - emit_begin) & ~(uint64_t)(FILTER_BLOCK - 1));
    u32x4u nxt[4];
#pragma unroll
    for (int q = 0; q < 4; q++) nxt[q] = (i + 16u * q < emit_end) ? *(const u32x4u *)(hay + i + 16u * q) : u32x4u{0, 0, 0, 0};
    while (i < full_end) {                                        // whole 64-column blocks, requested one block ahead
        u32x4u cur[4];
#pragma unroll
        for (int q = 0; q < 4; q++) cur[q] = nxt[q];
#pragma unroll
        for (int q = 0; q < 4; q++)                               // blobs carry 16 bytes of slack
            if (i + FILTER_BLOCK + 16u * q < emit_end) nxt[q] = *(const u32x4u *)(hay + i + FILTER_BLOCK + 16u * q);
        bool any = false;
#pragma unroll 1
        for (int q = 0; q < 4; q++) {
#pragma unroll
            for (int b = 0; b < 16; b++) any |= stepc((cur[q][b >> 2] >> (8 * (b & 3))) & 0xffu) <= k;
        }
        if (any) flag(i);
        i += FILTER_BLOCK;
    }
    if (i < emit_end) {                                           // the shard's last, partial block
        bool any = false;
        for (uint64_t j = i; j < emit_end; j++) any |= stepc(hay[j]) <= k;
        if (any) flag(i);
    }
}

template <int NWF>
static hipError_t launch_filter_n(const SearchParams &P, bool trans, uint32_t grid, uint32_t *list, uint32_t list_cap,
                                  unsigned int *list_count, hipStream_t s) {
    if (trans) hipLaunchKernelGGL((lev_filter_kernel_n<NWF, true>), dim3(grid), dim3(256), 0, s, P, list, list_cap, list_count);
    else hipLaunchKernelGGL((lev_filter_kernel_n<NWF, false>), dim3(grid), dim3(256), 0, s, P, list, list_cap, list_count);
    return hipGetLastError();
}

hipError_t lev_filter_launch(const SearchParams &P, bool trans, uint32_t *list, uint32_t list_cap, unsigned int *list_count,
                             hipStream_t s) {
    if (P.hay_len == 0) return hipSuccess;
    const uint64_t tiles = (P.hay_len + P.tile - 1) / P.tile;
    const uint32_t grid = (uint32_t)((tiles + 255) / 256);
    set_last_kernel_name("lev_filter_kernel%s", P.needle_len <= 32 ? "" : "_n");
    switch ((P.needle_len + 31u) / 32u) {
        case 1:
            if (!env_str("TA_FILTER_FLAT")) {
                const uint32_t g2 = (uint32_t)((tiles + 511) / 512);
                if (env_str("TA_FILTER_EXACT")) {   // A/B: the score per column (flags exactly the blocks that hold a hit)
                    if (trans) hipLaunchKernelGGL((lev_filter_kernel<true, true, true>), dim3(g2), dim3(512), 0, s, P, list, list_cap, list_count);
                    else hipLaunchKernelGGL((lev_filter_kernel<false, true, true>), dim3(g2), dim3(512), 0, s, P, list, list_cap, list_count);
                    return hipGetLastError();
                }
                if (trans) hipLaunchKernelGGL((lev_filter_kernel<true, true>), dim3(g2), dim3(512), 0, s, P, list, list_cap, list_count);
                else hipLaunchKernelGGL((lev_filter_kernel<false, true>), dim3(g2), dim3(512), 0, s, P, list, list_cap, list_count);
                return hipGetLastError();
            }
            // A/B (TA_FILTER_FLAT=1): the flat 256-entry table (bank conflicts on random bytes), 256 threads per workgroup
            if (trans) hipLaunchKernelGGL((lev_filter_kernel<true, false, true>), dim3(grid), dim3(256), 0, s, P, list, list_cap, list_count);
            else hipLaunchKernelGGL((lev_filter_kernel<false, false, true>), dim3(grid), dim3(256), 0, s, P, list, list_cap, list_count);
            return hipGetLastError();
        case 2: return launch_filter_n<2>(P, trans, grid, list, list_cap, list_count, s);
        case 3: return launch_filter_n<3>(P, trans, grid, list, list_cap, list_count, s);
        case 4: return launch_filter_n<4>(P, trans, grid, list, list_cap, list_count, s);
        case 5: return launch_filter_n<5>(P, trans, grid, list, list_cap, list_count, s);
        case 6: return launch_filter_n<6>(P, trans, grid, list, list_cap, list_count, s);
        case 7: return launch_filter_n<7>(P, trans, grid, list, list_cap, list_count, s);
        case 8: return launch_filter_n<8>(P, trans, grid, list, list_cap, list_count, s);
        case 9: case 10: case 11: case 12: return launch_filter_n<12>(P, trans, grid, list, list_cap, list_count, s);     // needles of up to 384 bytes
        case 13: case 14: case 15: case 16: return launch_filter_n<16>(P, trans, grid, list, list_cap, list_count, s);    // up to 512 (round 5; the table: 16 KB of LDS)
        default: return hipErrorInvalidValue;
    }
}

// the memory-backed exact kernel (needles > 32 bytes) on the flagged blocks
__global__ __launch_bounds__(64) void lev_search_mem_list_kernel(SearchParams P, const uint32_t *list, uint32_t n_list) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_list) return;
    const uint64_t emit_begin = (uint64_t)list[t] * FILTER_BLOCK;
    uint64_t emit_end = emit_begin + FILTER_BLOCK;
    if (emit_end > P.hay_len) emit_end = P.hay_len;
    const uint64_t col_begin = emit_begin > P.halo ? emit_begin - P.halo : 0;
    SearchCosts C{P.k, P.mc, P.gc, P.sg, P.tc, P.anchored};
    ta_match *hits = P.hits;
    unsigned long long *count = P.count;
    const uint64_t base = P.base, emit_from = P.emit_from, cap = P.cap;
    lev_search_tile_mem(P.hay, P.needle_dev, P.needle_len, C, P.tc != 0, P.col_scratch + t, n_list, col_begin, emit_begin, emit_end,
                        [=](uint64_t end, uint32_t len, uint32_t cost) {
                            const uint64_t gend = base + end;
                            if (gend <= emit_from) return;
                            unsigned long long idx = atomicAdd(count, 1ull);
                            if (idx < cap) hits[idx] = ta_match{gend - len, gend, cost, 0u};
                        });
}

// ---- the exact (cost, length) kernel on the flagged 64-column blocks: ONE WAVEFRONT per block (lev_search_wave_body.h)
//
// Persistent grid (SRCH_WAVE_GRID workgroups of 4 wavefronts): the number of flagged blocks is read ON THE DEVICE -- the
// host never waits between the filter and this kernel.  The last workgroup to finish writes the report (hit count, and for
// a Best pass the hits with the smallest k) into host-mapped pinned memory: one stream synchronisation tells the host
// everything, no copy.  When the filter flagged so many blocks that the lane-per-tile kernel over everything is cheaper,
// nothing is searched and the report says so.
//
// No device-wide fence anywhere: an agent-scope release on gfx950 is an L2 write-back (buffer_wbl2), and one per wavefront
// made this kernel 300 us long.  What crosses workgroups goes through agent-scope atomics instead: the hit cursor, the
// Best passes' hit records and block slots (stored with agent-scope atomic stores = write-through), the done counter.
// Best passes (only the hits with the smallest k can survive the Best fold): no running minimum, no second cursor -- a block
// leaves ONE 16-byte slot {its best cost, its hit count, where its hits start} (agent-scope atomic stores = write-through, like
// its hit records), the last workgroup takes the minimum over the n_list slots and picks the hits of that cost out of the few
// blocks that reach it.  (A running atomicMax plus a candidate cursor cost every block two more serial round trips: 45 us
// against 31 us for the All-mode kernel, profiles/r03/ab_search.md.)
constexpr uint32_t SRCH_WAVE_GRID = 512;
constexpr uint32_t SRCH_WAVE_DENSE_MUL = 5;          // a block costs ~5x its columns in the lane-per-tile kernel's currency

static __device__ __forceinline__ void store_match_agent(ta_match *dst, uint64_t start, uint64_t end, uint32_t k) {
    unsigned long long *w = (unsigned long long *)dst;
    __hip_atomic_store(w, (unsigned long long)start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(w + 1, (unsigned long long)end, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(w + 2, (unsigned long long)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool TRANS, bool BEST>
__global__ __launch_bounds__(256) void lev_search_wave_kernel(SearchParams P, const uint32_t *list, uint32_t cap_list, SearchCtl *ctl,
                                                              SearchSlot *slots, uint8_t *report, uint32_t done_groups) {
    __shared__ uint32_t s_last, s_sel;
    __shared__ __attribute__((aligned(16))) uint32_t s_row[4][64];        // per wavefront: the last row's values of a block's emitted columns
    (void)store_match_agent;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6)), n_waves = gridDim.x * 4u;   // wave-uniform: scalar loop control
    const uint32_t t_enter = (uint32_t)__builtin_amdgcn_s_memrealtime();
    const uint32_t n_list = ctl->n_list;                                   // written by the filter kernel before this launch
    const uint64_t per_block = (uint64_t)FILTER_BLOCK + P.halo + P.needle_len;
    const bool dense = n_list > cap_list || (uint64_t)n_list * per_block * SRCH_WAVE_DENSE_MUL >= P.hay_len;
    if (!dense) {
        // the needle, one byte per lane, straight from the kernarg segment (P is the first argument)
        const uint8_t *needle = (const uint8_t *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(SearchParams, needle);
        const SearchCosts C{P.k, P.mc, P.gc, P.sg, P.tc, 0u};
        ta_match *hits = P.hits;
        const uint64_t base = P.base, emit_from = P.emit_from, cap = P.cap;
        for (uint32_t t = wave; t < n_list; t += n_waves) {
            const uint64_t emit_begin = (uint64_t)list[t] * FILTER_BLOCK;
            uint64_t emit_end = emit_begin + FILTER_BLOCK;
            if (emit_end > P.hay_len) emit_end = P.hay_len;
            const uint64_t col_begin = emit_begin > P.halo ? emit_begin - P.halo : 0;
            if (BEST && t < SEARCH_SLOT_CAP && lane == 0)                  // "no hits" until the block says otherwise (no fill needed)
                __hip_atomic_store((unsigned long long *)(slots + t), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lev_search_block_wave<DevWave, TRANS>(P.hay, needle, P.needle_len, C, col_begin, emit_begin, emit_end, (uint8_t *)s_row[threadIdx.x >> 6],
                [=](bool hit, uint32_t key, uint32_t col) {               // lane t: emitted column t of the block
                    const uint64_t gend = base + col_begin + col + 1;
                    const uint32_t cost = key >> 16, len = 0xFFFFu - (key & 0xFFFFu);
                    const bool mine = hit && gend > emit_from;
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(mine);
                    if (m == 0) return;
                    const uint32_t cnt = (uint32_t)__builtin_popcountll(m), rank = (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                    const uint32_t leader = (uint32_t)__builtin_ctzll(m);
                    unsigned long long idx0 = 0;
                    if (lane == leader) idx0 = atomicAdd(&ctl->count, (unsigned long long)cnt);
                    idx0 = ((unsigned long long)__builtin_amdgcn_readlane((uint32_t)(idx0 >> 32), leader) << 32) |
                           (unsigned long long)__builtin_amdgcn_readlane((uint32_t)idx0, leader);
                    if (mine && idx0 + rank < cap) {
                        if (BEST) store_match_agent(hits + idx0 + rank, gend - len, gend, cost);     // the last workgroup may read it
                        else hits[idx0 + rank] = ta_match{gend - len, gend, cost, 0u};
                    }
                    if (BEST && t < SEARCH_SLOT_CAP) {
                        const uint32_t inv = DevWave::wave_max(mine ? 0xFFFFFFFFu - cost : 0u);      // 0xFFFFFFFF - the block's best cost
                        if (lane == leader) {
                            unsigned long long *sw = (unsigned long long *)(slots + t);
                            __hip_atomic_store(sw, ((unsigned long long)cnt << 32) | (0xFFFFFFFFu - inv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(sw + 1, idx0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                });
        }
    }
    if (wave == 0 && lane == 0) {
        __hip_atomic_store(&ctl->pad[0], t_enter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ctl->pad[1], (uint32_t)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // last workgroup out writes the report.  Every store that another workgroup reads later was an agent-scope atomic store and this
    // wavefront's own memory operations are waited for (s_waitcnt 0); on top of that:
    // Bit 31 of done_groups (the default; TA_SRCH_NO_FENCE=1 clears it): thread 0 of every workgroup issues ONE agent-scope release fence in
    // front of its counter bump and the last workgroup an acquire fence behind it -- the release / acquire pair the language memory model
    // asks for, per workgroup, not per wavefront (per wavefront it made the kernel 300 us long; per workgroup it costs 2-6 us per pass).
    const bool fenced = (done_groups & 0x80000000u) != 0u;
    done_groups &= 0x7FFFFFFFu;
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {       // two levels: the last workgroup of a group bumps the groups' counter
        if (fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const uint32_t g = blockIdx.x % done_groups, members = gridDim.x / done_groups + (g < gridDim.x % done_groups ? 1u : 0u);
        uint32_t last = 0;
        if (__hip_atomic_fetch_add(&ctl->done[g * 16u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1u)
            last = __hip_atomic_fetch_add(&ctl->done2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == done_groups - 1u ? 1u : 0u;
        s_last = last;
    }
    s_sel = 0;
    __syncthreads();
    if (!s_last) return;
    if (fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const uint32_t t_report = (uint32_t)__builtin_amdgcn_s_memrealtime();
    const unsigned long long count = __hip_atomic_load(&ctl->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    SearchReport *rep = (SearchReport *)report;
    uint32_t sel_state = 0, min_k = 0xFFFFFFFFu, n_cand_seen = 0;
    if (BEST && !dense) {
        if (count <= P.cap && n_list <= SEARCH_SLOT_CAP) {
            __shared__ uint32_t s_min;
            if (threadIdx.x == 0) s_min = 0xFFFFFFFFu;
            __syncthreads();
            const unsigned long long *sw = (const unsigned long long *)slots;
            uint32_t mine_min = 0xFFFFFFFFu;
            for (uint32_t i0 = 0; i0 < n_list; i0 += 1024u) {          // four loads in flight per thread: other workgroups wrote the
                unsigned long long w[4];                                 // slots, they are read past the L1 / this XCD's L2
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t i = i0 + 256u * (uint32_t)j + threadIdx.x;
                    w[j] = i < n_list ? __hip_atomic_load(sw + 2 * (uint64_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                }
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if ((uint32_t)(w[j] >> 32) != 0u && (uint32_t)w[j] < mine_min) mine_min = (uint32_t)w[j];
            }
            if (mine_min != 0xFFFFFFFFu) atomicMin(&s_min, mine_min);
            __syncthreads();
            min_k = s_min;
            ta_match *sel = (ta_match *)(report + sizeof(SearchReport));
            const unsigned long long *hw = (const unsigned long long *)P.hits;
            if (min_k != 0xFFFFFFFFu) {
                for (uint32_t i = threadIdx.x; i < n_list; i += 256u) {
                    const unsigned long long w = __hip_atomic_load(sw + 2 * (uint64_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t cnt = (uint32_t)(w >> 32);
                    if (cnt == 0u || (uint32_t)w != min_k) continue;
                    const unsigned long long idx0 = __hip_atomic_load(sw + 2 * (uint64_t)i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (uint32_t q = 0; q < cnt; q++) {
                        const unsigned long long w2 = __hip_atomic_load(hw + 3 * (idx0 + q) + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((uint32_t)w2 != min_k) continue;
                        const unsigned long long w0 = __hip_atomic_load(hw + 3 * (idx0 + q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned long long w1 = __hip_atomic_load(hw + 3 * (idx0 + q) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const uint32_t pos = atomicAdd(&s_sel, 1u);
                        if (pos < SEARCH_REPORT_SEL) sel[pos] = ta_match{w0, w1, (uint32_t)w2, 0u};
                    }
                }
            }
            __syncthreads();
            sel_state = s_sel <= SEARCH_REPORT_SEL ? 1u : 2u;
            n_cand_seen = n_list;
        } else {
            sel_state = 2u;
        }
    }
    if (threadIdx.x == 0) {
        rep->count = count; rep->n_list = n_list; rep->dense = dense ? 1u : 0u;
        rep->sel_count = s_sel; rep->sel_state = sel_state; rep->min_k = min_k; rep->n_slots = n_cand_seen;
        rep->t[0] = __hip_atomic_load(&ctl->pad[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        rep->t[1] = __hip_atomic_load(&ctl->pad[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        rep->t[2] = t_report; rep->t[3] = (uint32_t)__builtin_amdgcn_s_memrealtime();
    }
}

hipError_t lev_search_wave_launch(const SearchParams &P, bool trans, bool best, const uint32_t *list, uint32_t cap_list,
                                  SearchCtl *ctl, SearchSlot *slots, uint8_t *report_dev, hipStream_t s) {
    if (P.needle_len == 0 || P.needle_len > 64 || FILTER_BLOCK + P.halo > SRCH_WAVE_MAX_COLS) return hipErrorInvalidValue;
    uint32_t g = SRCH_WAVE_GRID, done_groups = SEARCH_DONE_GROUPS;
    if (const char *e = env_str("TA_SRCH_GRID")) { const int v = atoi(e); if (v >= 1 && v <= 4096) g = (uint32_t)v; }
    if (const char *e = env_str("TA_SRCH_DONE_GROUPS")) { const int v = atoi(e); if (v >= 1 && v <= (int)SEARCH_DONE_GROUPS) done_groups = (uint32_t)v; }
    if (done_groups > g) done_groups = g;
    if (!env_str("TA_SRCH_NO_FENCE")) done_groups |= 0x80000000u;    // (2-6 us of a 0.49 ms pass: profiles/r04/raw/probe_search_fence.txt)
    const dim3 grid(g), block(256);
    if (trans) {
        if (best) hipLaunchKernelGGL((lev_search_wave_kernel<true, true>), grid, block, 0, s, P, list, cap_list, ctl, slots, report_dev, done_groups);
        else hipLaunchKernelGGL((lev_search_wave_kernel<true, false>), grid, block, 0, s, P, list, cap_list, ctl, slots, report_dev, done_groups);
    } else {
        if (best) hipLaunchKernelGGL((lev_search_wave_kernel<false, true>), grid, block, 0, s, P, list, cap_list, ctl, slots, report_dev, done_groups);
        else hipLaunchKernelGGL((lev_search_wave_kernel<false, false>), grid, block, 0, s, P, list, cap_list, ctl, slots, report_dev, done_groups);
    }
    return hipGetLastError();
}

// needles beyond the wavefront kernel's 64 rows: the memory-backed column, one lane per flagged block
hipError_t lev_search_list_launch(const SearchParams &P, bool /*trans*/, const uint32_t *list, uint32_t n_list, hipStream_t s) {
    if (n_list == 0) return hipSuccess;
    if (P.needle_len <= 32) return hipErrorInvalidValue;
    hipLaunchKernelGGL(lev_search_mem_list_kernel, dim3((n_list + 63) / 64), dim3(64), 0, s, P, list, n_list);
    return hipGetLastError();
}

// hamming_search: one lane per aligned group of 4 consecutive haystack offsets.  The lane loads the aligned dwords
// covering its 4 windows once (coalesced), re-aligns them per offset with v_alignbyte, and counts non-zero bytes of
// window ^ needle with the SWAR test + v_bcnt accumulate.  Replaces hamming_search_simd_core_*
// (src/hamming.rs:481-552); contract: mismatches(offset) <= k for every offset 0..=h-n.
__global__ __launch_bounds__(256) void hamming_search_kernel(SearchParams P, uint32_t delta) {
    const uint64_t grp = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // aligned dword index of the group's first byte
    const uint32_t n = P.needle_len;
    const uint64_t h = P.hay_len;
    const uint64_t last = h - n;                                            // last valid offset
    const uint64_t byte0 = grp * 4;                                         // in the aligned-down address space
    if (byte0 > last + delta) return;
    const uint32_t *hw = (const uint32_t *)(P.hay - delta) + grp;
    const uint32_t nwords = (n + 3) >> 2;
    const uint32_t tail_mask = (n & 3) ? (0xFFFFFFFFu >> (8 * (4 - (n & 3)))) : 0xFFFFFFFFu;
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    uint32_t lo = hw[0];
    for (uint32_t w = 0; w < nwords; w++) {
        const uint32_t hi = hw[w + 1];
        uint32_t nd;                                                        // needle word w (wave-uniform)
        {
            typedef uint32_t u32u __attribute__((aligned(1)));
            nd = *(const u32u *)(P.needle_dev + 4 * w);                     // needle_dev carries 16 bytes of slack
        }
        // byte test with one v_perm_b32 (wave.h ne12): the needle word carries ^ 0x0C, a window byte then XORs to 12 exactly
        // where it matches; bytes past the needle's end are forced to 12.  Every mismatching byte adds 8 one-bits.
        const uint32_t m = (w + 1 == nwords) ? tail_mask : 0xFFFFFFFFu;
        const uint32_t nd12 = nd ^ 0x0C0C0C0Cu, pad = 0x0C0C0C0Cu & ~m;
        auto nz = [&](uint32_t win) -> uint32_t {
            return __builtin_amdgcn_perm(0xFFFFFFFFu, 0xFFFFFFFFu, ((win ^ nd12) & m) | pad);
        };
        c0 += __builtin_popcount(nz(lo));
        c1 += __builtin_popcount(nz(__builtin_amdgcn_alignbyte(hi, lo, 1)));
        c2 += __builtin_popcount(nz(__builtin_amdgcn_alignbyte(hi, lo, 2)));
        c3 += __builtin_popcount(nz(__builtin_amdgcn_alignbyte(hi, lo, 3)));
        lo = hi;
    }
    const uint32_t cnt[4] = {c0 >> 3, c1 >> 3, c2 >> 3, c3 >> 3};
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint64_t b = byte0 + r;
        if (b < delta) continue;
        const uint64_t pos = b - delta;
        if (pos > last) continue;
        if (cnt[r] <= P.k) {
            unsigned long long idx = atomicAdd(P.count, 1ull);
            if (idx < P.cap) P.hits[idx] = ta_match{P.base + pos, P.base + pos + n, cnt[r], 0u};
        }
    }
}

// needles of up to 64 bytes, SWAR form (ham_swar_body.h): one lane per 16 consecutive offsets, three 16-byte loads per lane in flight
// together, the windows of a lane shifted once and shared by its offsets.  The NUL-byte scan of the SIMD contract (src/lib.rs:237-243)
// rides along: every haystack dword is the first of exactly one lane's four.  delta = hay & 15 (the loads are 16-byte aligned).
template <int NW>
__global__ __launch_bounds__(256) void hamming_search_swar16_kernel(SearchParams P, uint32_t delta, uint32_t *nul_flag) {
    const uint64_t lane_id = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint64_t byte0 = lane_id * 16u;                       // in the address space of hay - delta
    const uint32_t n = P.needle_len, k = P.k;
    const uint64_t h = P.hay_len, last = h - n, end = (uint64_t)delta + h;   // bytes [delta, end) are the haystack; 16 bytes of slack follow
    if (byte0 >= end) return;                                   // (the lanes behind the last offset still check their bytes for NUL)
    uint32_t nd12[NW], tail_mask, tail_pad;
    ham_swar_needle<NW>(P.needle, n, nd12, tail_mask, tail_pad);
    typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(16)));
    const u32x4a *src = (const u32x4a *)(P.hay - delta + byte0);
    constexpr int NQ = (4 + NW + 3) / 4;
    u32x4a q[NQ];
#pragma unroll
    for (int t = 0; t < NQ; t++) q[t] = (byte0 + 16u * (uint64_t)t <= end) ? src[t] : u32x4a{0u, 0u, 0u, 0u};
    uint32_t w[4 + NW];
#pragma unroll
    for (int i = 0; i < 4 + NW; i++) w[i] = q[i >> 2][i & 3];
    if (nul_flag) {                                             // bytes byte0 .. byte0 + 15 are this lane's to check
        bool z;
        if (byte0 >= delta && byte0 + 16u <= end) {
            z = (ham_ne12(w[0] ^ 0x0C0C0C0Cu) & ham_ne12(w[1] ^ 0x0C0C0C0Cu) & ham_ne12(w[2] ^ 0x0C0C0C0Cu) & ham_ne12(w[3] ^ 0x0C0C0C0Cu)) != 0xFFFFFFFFu;
        } else {
            z = false;
            for (uint32_t b = 0; b < 16u; b++) {
                const uint64_t x = byte0 + b;
                if (x >= delta && x < end) z |= ((w[b >> 2] >> (8u * (b & 3u))) & 0xFFu) == 0u;
            }
        }
        if (z) atomicOr(nul_flag, 1u);
    }
    uint32_t cnt[16];
    ham_swar_lane<NW>(w, nd12, tail_mask, tail_pad, cnt);
    // the counts are 8 x the mismatches: one minimum over the sixteen (v_min3) against 8 k + 7 decides whether the lane reports at all
    const uint32_t thr = k >= 0x1FFFFFFFu ? 0xFFFFFFFFu : 8u * k + 7u;
    auto min3 = [](uint32_t a, uint32_t b, uint32_t c) { const uint32_t m = a < b ? a : b; return m < c ? m : c; };
    const uint32_t m0 = min3(cnt[0], cnt[1], cnt[2]), m1 = min3(cnt[3], cnt[4], cnt[5]), m2 = min3(cnt[6], cnt[7], cnt[8]),
                   m3 = min3(cnt[9], cnt[10], cnt[11]), m4 = min3(cnt[12], cnt[13], cnt[14]);
    if (min3(min3(m0, m1, m2), min3(m3, m4, cnt[15]), 0xFFFFFFFFu) > thr) return;
#pragma unroll 1
    for (uint32_t o = 0; o < 16u; o++) {
        const uint64_t x = byte0 + o;
        if (cnt[o] > thr || x < delta || x - delta > last) continue;
        const uint64_t pos = x - delta;
        unsigned long long idx = atomicAdd(P.count, 1ull);
        if (idx < P.cap) P.hits[idx] = ta_match{P.base + pos, P.base + pos + n, cnt[o] >> 3, 0u};
    }
}

// needles of 9..32 bytes with a small k, bit-sliced counters (ham_bits_body.h): one lane per tile of P.tile bytes (a multiple of 128), the
// 256-entry Mis table once per lane in LDS (64 KB, 512 threads: two workgroups per CU), a whole 128-byte line per lane per burst one line
// ahead.  The NUL-byte scan rides along (the lane checks the bytes of its own tile).
template <int B>
__global__ __launch_bounds__(512) void hamming_search_bits_kernel(SearchParams P, uint32_t *nul_flag) {
    __shared__ __attribute__((aligned(16))) uint32_t mis[256 * 64];
    {
        const uint32_t m = ham_bits_mis(P.needle, P.needle_len, threadIdx.x >> 1);
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 *row = (u32x4 *)(mis + (threadIdx.x >> 1) * 64 + (threadIdx.x & 1u) * 32);
#pragma unroll
        for (int q = 0; q < 8; q++) row[q] = u32x4{m, m, m, m};
    }
    __syncthreads();
    const uint32_t lane_off = (threadIdx.x & 63u) * 4u;
    auto lookup = [&](uint32_t v, int b) -> uint32_t {
        const uint32_t a = __builtin_amdgcn_perm(v, lane_off, 0x0C0C0000u | ((4u + (uint32_t)b) << 8));   // lane*4 | byte b << 8
        return *(const uint32_t *)((const uint8_t *)mis + a);
    };
    const uint32_t n = P.needle_len, k = P.k;
    const uint64_t h = P.hay_len, last = h - n;
    const uint64_t tile = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t b0 = tile * P.tile;                             // this lane reports the alignments that END at bytes [b0, b1)
    if (b0 >= h) return;
    const uint64_t b1 = b0 + P.tile < h ? b0 + P.tile : h;
    const uint8_t *hay = P.hay;
    HamBitsState<B> st;
    ham_bits_reset<B>(st, n);
    uint32_t bias[B];
    ham_bits_bias<B>(k, n, bias);
    for (uint64_t i = b0 > n - 1u ? b0 - (n - 1u) : 0; i < b0; i++) ham_bits_step<B>(st, lookup(hay[i], 0), bias);   // the n - 1 bytes in front
    auto report = [&](uint64_t x) {                                // the alignment ending at byte x has at most k mismatches
        if (x < n - 1u) return;                                    // (no alignment ends there: the verdict bits of a haystack's first bytes)
        const uint64_t pos = x - (n - 1u);
        if (pos > last) return;
        uint32_t cnt = 0;
        for (uint32_t j = 0; j < n; j++) cnt += hay[pos + j] != P.needle[j];           // (n <= 32: the kernarg copy)
        const unsigned long long idx = atomicAdd(P.count, 1ull);
        if (idx < P.cap) P.hits[idx] = ta_match{P.base + pos, P.base + pos + n, cnt, 0u};
    };
    typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(1)));
    uint64_t i = b0;
    const uint64_t full_end = b0 + ((b1 - b0) & ~(uint64_t)127);
    uint32_t nz = 0xFFFFFFFFu;                                     // AND of the byte tests: 0xFF per byte that is not NUL
    u32x4u nxt[8];
#pragma unroll
    for (int q = 0; q < 8; q++) nxt[q] = (i + 16u * q < b1) ? *(const u32x4u *)(hay + i + 16u * q) : u32x4u{0, 0, 0, 0};
    while (i < full_end) {                                         // whole lines
        u32x4u cur[8];
#pragma unroll
        for (int q = 0; q < 8; q++) cur[q] = nxt[q];
#pragma unroll
        for (int q = 0; q < 8; q++)                                // blobs carry 16 bytes of slack
            if (i + 128u + 16u * q < b1) nxt[q] = *(const u32x4u *)(hay + i + 128u + 16u * q);
#pragma unroll 1
        for (int part = 0; part < 4; part++) {                     // 32 bytes = 32 verdicts per register
            uint32_t acc = 0;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const u32x4u v = part == 0 ? cur[q] : part == 1 ? cur[2 + q] : part == 2 ? cur[4 + q] : cur[6 + q];
#pragma unroll
                for (int b = 0; b < 16; b++) {
                    const uint32_t ov = ham_bits_step<B>(st, lookup(v[b >> 2], b & 3), bias);
                    acc = __builtin_amdgcn_alignbit(acc, ov, 31);   // (acc << 1) | (ov >> 31)
                }
                if (nul_flag) nz &= __builtin_amdgcn_perm(0xFFFFFFFFu, 0xFFFFFFFFu, v[0] ^ 0x0C0C0C0Cu) & __builtin_amdgcn_perm(0xFFFFFFFFu, 0xFFFFFFFFu, v[1] ^ 0x0C0C0C0Cu) &
                                    __builtin_amdgcn_perm(0xFFFFFFFFu, 0xFFFFFFFFu, v[2] ^ 0x0C0C0C0Cu) & __builtin_amdgcn_perm(0xFFFFFFFFu, 0xFFFFFFFFu, v[3] ^ 0x0C0C0C0Cu);
            }
            if (acc != 0xFFFFFFFFu)                                // bit 31 - t = the verdict of byte i + 32 part + t
                for (uint32_t t = 0; t < 32u; t++)
                    if (!((acc >> (31u - t)) & 1u)) report(i + 32u * (uint32_t)part + t);
        }
        i += 128;
    }
    for (; i < b1; i++) {                                          // the tile's last, partial line
        const uint32_t c = hay[i];
        if (!(ham_bits_step<B>(st, lookup(c, 0), bias) >> 31)) report(i);
        if (nul_flag && c == 0u) nz = 0;
    }
    if (nul_flag && nz != 0xFFFFFFFFu) atomicOr(nul_flag, 1u);
}

// Bit-sliced counters over a subset of the needle's positions, Q phases per dword (ham_phase_body.h): any needle length; one lane per tile of
// P.tile bytes (a multiple of 128), the 256-entry phase-0 Mis table once per lane in LDS (64 KB, 512 threads), a whole 128-byte line per
// lane per burst one line ahead.  A 16-byte piece is 16 / Q steps whose verdict words are AND-ed: only a piece that holds a candidate looks
// at them one by one, and a candidate is recounted over the whole needle.  The NUL-byte scan rides along.
template <int Q, int B>
__global__ __launch_bounds__(512) void hamming_search_phase_kernel(SearchParams P, HamPhaseGeom G, uint32_t *nul_flag) {
    __shared__ __attribute__((aligned(16))) uint32_t mis[256 * 64];
    {
        const uint32_t m = ham_phase_mis(P.needle_len <= 64u ? P.needle : P.needle_dev, G, threadIdx.x >> 1);   // (up to 64 bytes: the kernarg copy, no upload)
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 *row = (u32x4 *)(mis + (threadIdx.x >> 1) * 64 + (threadIdx.x & 1u) * 32);
#pragma unroll
        for (int q = 0; q < 8; q++) row[q] = u32x4{m, m, m, m};
    }
    __syncthreads();
    constexpr uint32_t W = 32u / Q, STEPS = 16u / Q;
    const uint32_t lane_off = (threadIdx.x & 63u) * 4u;
    auto lookup = [&](uint32_t v, int b) -> uint32_t {
        const uint32_t a = __builtin_amdgcn_perm(v, lane_off, 0x0C0C0000u | ((4u + (uint32_t)b) << 8));   // lane*4 | byte b << 8
        return *(const uint32_t *)((const uint8_t *)mis + a);
    };
    const uint32_t n = P.needle_len, k = P.k, keep = G.keep, span = G.span;
    const uint64_t h = P.hay_len, last = h - n;
    const uint64_t tile = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t b0 = tile * P.tile;                             // this lane reports the alignments whose VERDICT byte is in [b0, b1)
    if (b0 >= h) return;
    const uint64_t b1 = b0 + P.tile < h ? b0 + P.tile : h;
    const uint8_t *hay = P.hay;
    HamBitsState<B> st;
    ham_phase_reset<B>(st, G);
    uint32_t bias[B];
    ham_phase_bias<B>(k, G, bias);
    auto bytes_step = [&](uint64_t i) -> uint32_t {                // one step from single bytes (warm-up, the last partial line)
        uint32_t m = 0;
#pragma unroll
        for (int r = Q - 1; r >= 0; r--) {
            const uint32_t c = i + (uint32_t)r < h ? hay[i + (uint32_t)r] : 0u;
            const uint32_t t = lookup(c, 0);
            m = Q == 1 ? t : ((m << (W & 31u)) | t);
        }
        return ham_phase_step<B, (Q > 1)>(st, m, bias, keep);
    };
    for (uint64_t i = b0 > span ? b0 - span : 0; i < b0; i += Q) bytes_step(i);   // the span bytes in front (b0 and span are multiples of Q)
    auto verify = [&](uint64_t x) {                                // the alignment whose verdict byte is x passed the filter
        if (x < span) return;                                      // (no alignment ends there: the verdict bits of a haystack's first bytes)
        const uint64_t pos = x - span;
        if (pos > last) return;
        uint32_t cnt = 0;
        const uint8_t *nd = n <= 64u ? P.needle : P.needle_dev;
        for (uint32_t j = 0; j < n; j++) cnt += hay[pos + j] != nd[j];
        if (cnt > k) return;
        const unsigned long long idx = atomicAdd(P.count, 1ull);
        if (idx < P.cap) P.hits[idx] = ta_match{P.base + pos, P.base + pos + n, cnt, 0u};
    };
    typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(1)));
    uint64_t i = b0;
    const uint64_t full_end = b0 + ((b1 - b0) & ~(uint64_t)127);
    uint32_t nz = 0xFFFFFFFFu;                                     // AND of the byte tests: 0xFF per byte that is not NUL
    u32x4u nxt[8];
#pragma unroll
    for (int q = 0; q < 8; q++) nxt[q] = (i + 16u * q < b1) ? *(const u32x4u *)(hay + i + 16u * q) : u32x4u{0, 0, 0, 0};
    while (i < full_end) {                                         // whole lines
        u32x4u cur[8];
#pragma unroll
        for (int q = 0; q < 8; q++) cur[q] = nxt[q];
#pragma unroll
        for (int q = 0; q < 8; q++)                                // blobs carry 16 bytes of slack
            if (i + 128u + 16u * q < b1) nxt[q] = *(const u32x4u *)(hay + i + 128u + 16u * q);
#pragma unroll 1
        for (int part = 0; part < 4; part++) {
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const u32x4u v = part == 0 ? cur[q] : part == 1 ? cur[2 + q] : part == 2 ? cur[4 + q] : cur[6 + q];
                // windows of at most 8 steps: their verdict words are kept (8 registers) and AND-ed; only a window that holds a candidate
                // looks at them one by one
                constexpr uint32_t WIN = STEPS > 8u ? 8u : STEPS;
#pragma unroll
                for (uint32_t w0 = 0; w0 < STEPS; w0 += WIN) {
                    uint32_t ov[WIN];
#pragma unroll
                    for (uint32_t s = 0; s < WIN; s++) {
                        uint32_t m = 0;
#pragma unroll
                        for (int r = Q - 1; r >= 0; r--) {
                            const uint32_t b = Q * (w0 + s) + (uint32_t)r;
                            const uint32_t t = lookup(v[b >> 2], b & 3);
                            m = Q == 1 ? t : ((m << (W & 31u)) | t);
                        }
                        ov[s] = ham_phase_step<B, (Q > 1)>(st, m, bias, keep);
                    }
                    uint32_t all = ov[0];
#pragma unroll
                    for (uint32_t s = 1; s < WIN; s++) all &= ov[s];
                    if ((all | keep) != 0xFFFFFFFFu) {              // a phase top that is clear: a candidate somewhere in this window
                        uint32_t cm = 0;                           // bit Q s + r: the verdict of byte Q (w0 + s) + r of the piece
#pragma unroll
                        for (uint32_t s = 0; s < WIN; s++) {
                            uint32_t o = ov[s];
                            asm volatile("" : "+v"(o));             // (keeps the sorting of the verdicts inside the rare branch: hipcc hoists it otherwise)
#pragma unroll
                            for (uint32_t r = 0; r < (uint32_t)Q; r++)
                                cm |= ((~o >> (r * W + W - 1u)) & 1u) << (Q * s + r);
                        }
                        const uint64_t first = i + 32u * (uint32_t)part + 16u * (uint32_t)q + Q * w0;
                        while (cm) {
                            const uint32_t t = (uint32_t)__builtin_ctz(cm);
                            cm &= cm - 1u;
                            verify(first + t);
                        }
                    }
                }
                if (nul_flag) nz &= __builtin_amdgcn_perm(0xFFFFFFFFu, 0xFFFFFFFFu, v[0] ^ 0x0C0C0C0Cu) & __builtin_amdgcn_perm(0xFFFFFFFFu, 0xFFFFFFFFu, v[1] ^ 0x0C0C0C0Cu) &
                                    __builtin_amdgcn_perm(0xFFFFFFFFu, 0xFFFFFFFFu, v[2] ^ 0x0C0C0C0Cu) & __builtin_amdgcn_perm(0xFFFFFFFFu, 0xFFFFFFFFu, v[3] ^ 0x0C0C0C0Cu);
            }
        }
        i += 128;
    }
    for (; i < b1; i += Q) {                                       // the tile's last, partial line, step by step
        const uint32_t ov = bytes_step(i);
#pragma unroll
        for (uint32_t r = 0; r < (uint32_t)Q; r++) {
            if (!((ov >> (r * W + W - 1u)) & 1u)) verify(i + r);
            if (nul_flag && i + r < h && hay[i + r] == 0u) nz = 0;
        }
    }
    if (nul_flag && nz != 0xFFFFFFFFu) atomicOr(nul_flag, 1u);
}

// needles of up to 32 bytes: shift-add scan (ham_search_body.h), one lane per tile of P.tile offsets, table in LDS,
// haystack requested 64 bytes per lane one block ahead
template <int NWS>
__global__ __launch_bounds__(256) void hamming_search_sa_kernel(SearchParams P) {
    __shared__ uint32_t tab[256 * NWS];
    for (int w = 0; w < NWS; w++) tab[threadIdx.x * NWS + w] = ham_sa_table_word(P.needle, P.needle_len, threadIdx.x, w);
    __syncthreads();
    const uint32_t n = P.needle_len, k = P.k;
    const uint64_t offsets = P.hay_len - n + 1;
    const uint64_t ob = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * P.tile;
    if (ob >= offsets) return;
    const uint64_t oe = ob + P.tile < offsets ? ob + P.tile : offsets;
    const uint8_t *hay = P.hay;
    HamSaState<NWS> st;
#pragma unroll
    for (int w = 0; w < NWS; w++) st.S[w] = 0;
    auto stepc = [&](uint32_t c) -> uint32_t {
        uint32_t Tc[NWS];
#pragma unroll
        for (int w = 0; w < NWS; w++) Tc[w] = tab[c * NWS + w];
        return ham_sa_step<NWS>(st, Tc, n);
    };
    auto emit = [&](uint64_t p, uint32_t cnt) {
        unsigned long long idx = atomicAdd(P.count, 1ull);
        if (idx < P.cap) P.hits[idx] = ta_match{P.base + p, P.base + p + n, cnt, 0u};
    };
    uint64_t i = ob;
    for (; i < ob + (n - 1); i++) stepc(hay[i]);                  // the first n-1 bytes complete no offset
    const uint64_t end = oe + (n - 1);                            // one past the last byte this tile reads
    typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(1)));
    const uint64_t full_end = i + ((end - i) & ~(uint64_t)63);
    u32x4u nxt[4];
#pragma unroll
    for (int q = 0; q < 4; q++) nxt[q] = (i + 16u * q < end) ? *(const u32x4u *)(hay + i + 16u * q) : u32x4u{0, 0, 0, 0};
    while (i < full_end) {
        u32x4u cur[4];
#pragma unroll
        for (int q = 0; q < 4; q++) cur[q] = nxt[q];
#pragma unroll
        for (int q = 0; q < 4; q++)                               // blobs carry 16 bytes of slack
            if (i + 64u + 16u * q < end) nxt[q] = *(const u32x4u *)(hay + i + 64u + 16u * q);
#pragma unroll 1
        for (int q = 0; q < 4; q++) {
#pragma unroll
            for (int b = 0; b < 16; b++) {
                const uint32_t cnt = stepc((cur[q][b >> 2] >> (8 * (b & 3))) & 0xffu);
                if (cnt <= k) emit(i + 16u * q + b - (n - 1), cnt);
            }
        }
        i += 64;
    }
    for (; i < end; i++) {
        const uint32_t cnt = stepc(hay[i]);
        if (cnt <= k) emit(i - (n - 1), cnt);
    }
}

// nul_flag: where the kernel reports a zero byte in the haystack (nullptr: no check wanted); *nul_done = the launch did the check
hipError_t hamming_search_launch(const SearchParams &P0, hipStream_t s, uint32_t *nul_flag, bool *nul_done) {
    SearchParams P = P0;
    if (nul_done) *nul_done = false;
    if (P.hay_len < P.needle_len || P.needle_len == 0) return hipSuccess;
    // needles of 9..32 bytes whose k needs few counter bits: bit-sliced counters, 3 B + 3 instructions per byte against the SWAR form's
    // 3 per needle dword + 3 (TA_HAMMING_SEARCH_NO_BITS=1 keeps the SWAR form)
    // bit-sliced counters over a subset of the needle's positions, Q phases per dword (ham_phase_body.h): any needle length, where the
    // subset is selective for k and the form costs fewer instructions per byte than the SWAR form (TA_HAMMING_SEARCH_NO_PHASE=1: without)
    {
        uint32_t Q = 0, L = 0; int B = 0;
        const uint32_t swar_x4 = 4u * (3u * ((P.needle_len + 3u) / 4u) + 4u);
        if ((P.needle_dev || P.needle_len <= 64u) && ham_phase_plan(P.needle_len, P.k, Q, L, B) && (P.needle_len > 64u || ham_phase_cost_x4(Q, B) < swar_x4) &&
            !env_str("TA_HAMMING_SEARCH_SWAR") && !env_str("TA_HAMMING_SEARCH_SA") && !env_str("TA_HAMMING_SEARCH_NO_BITS") &&
            !env_str("TA_HAMMING_SEARCH_NO_PHASE")) {
            if (const char *fq = env_str("TA_HAMMING_PHASE_Q")) {                    // tests: force fewer phases (1 or 2) where the plan allows them
                const uint32_t q = (uint32_t)atoi(fq);
                if ((q == 1u || q == 2u) && q < Q) {
                    uint32_t l = (P.needle_len + q - 1u) / q;
                    if (l > 32u / q) l = 32u / q;
                    Q = q; L = l;
                }
            }
            const HamPhaseGeom G = ham_phase_geom(Q, L);
            uint64_t tile = (P.hay_len + 262143) / 262144;            // two sets of resident lanes
            tile = (tile + 127) & ~(uint64_t)127;
            if (tile < 256) tile = 256;
            P.tile = (uint32_t)(tile > 0x7FFFFF80ull ? 0x7FFFFF80ull : tile);
            const uint64_t lanes = (P.hay_len + P.tile - 1) / P.tile;
            const dim3 grid((uint32_t)((lanes + 511) / 512)), block(512);
            if (nul_done) *nul_done = nul_flag != nullptr;
            set_last_kernel_name("hamming_search_phase_kernel<%u,%d>", Q, B);
#define TA_HP(QQ, BB) hipLaunchKernelGGL((hamming_search_phase_kernel<QQ, BB>), grid, block, 0, s, P, G, nul_flag)
#define TA_HPB(QQ) switch (B) { case 1: TA_HP(QQ, 1); break; case 2: TA_HP(QQ, 2); break; case 3: TA_HP(QQ, 3); break; case 4: TA_HP(QQ, 4); break; default: TA_HP(QQ, 5); break; }
            if (Q == 4u) { TA_HPB(4) } else if (Q == 2u) { TA_HPB(2) } else { TA_HPB(1) }
#undef TA_HPB
#undef TA_HP
            return hipGetLastError();
        }
    }
    const int planes = ham_bits_planes(P.k);
    if (P.needle_len >= 9 && P.needle_len <= 32 && planes && P.k < P.needle_len && 3 * planes + 3 < 3 * (int)((P.needle_len + 3) / 4) + 1 &&
        !env_str("TA_HAMMING_SEARCH_SWAR") && !env_str("TA_HAMMING_SEARCH_SA") && !env_str("TA_HAMMING_SEARCH_NO_BITS")) {
        uint64_t tile = (P.hay_len + 262143) / 262144;            // two sets of resident lanes
        tile = (tile + 127) & ~(uint64_t)127;
        if (tile < 256) tile = 256;
        P.tile = (uint32_t)(tile > 0x7FFFFF80ull ? 0x7FFFFF80ull : tile);
        const uint64_t lanes = (P.hay_len + P.tile - 1) / P.tile;
        const dim3 grid((uint32_t)((lanes + 511) / 512)), block(512);
        if (nul_done) *nul_done = nul_flag != nullptr;
        set_last_kernel_name("hamming_search_bits_kernel<%d>", planes);
        switch (planes) {
            case 1: hipLaunchKernelGGL(hamming_search_bits_kernel<1>, grid, block, 0, s, P, nul_flag); break;
            case 2: hipLaunchKernelGGL(hamming_search_bits_kernel<2>, grid, block, 0, s, P, nul_flag); break;
            case 3: hipLaunchKernelGGL(hamming_search_bits_kernel<3>, grid, block, 0, s, P, nul_flag); break;
            case 4: hipLaunchKernelGGL(hamming_search_bits_kernel<4>, grid, block, 0, s, P, nul_flag); break;
            default: hipLaunchKernelGGL(hamming_search_bits_kernel<5>, grid, block, 0, s, P, nul_flag); break;
        }
        return hipGetLastError();
    }
    if (P.needle_len <= 64 && !env_str("TA_HAMMING_SEARCH_SWAR") && !env_str("TA_HAMMING_SEARCH_SA")) {
        const uint32_t delta = (uint32_t)((uintptr_t)P.hay & 15u);
        const uint64_t lanes = (P.hay_len + delta + 15) / 16;
        const dim3 grid((uint32_t)((lanes + 255) / 256)), block(256);
        if (nul_done) *nul_done = nul_flag != nullptr;
        set_last_kernel_name("hamming_search_swar16_kernel<%u>", (P.needle_len + 3) / 4);
        switch ((P.needle_len + 3) / 4) {
#define TA_HS(NW) case NW: hipLaunchKernelGGL(hamming_search_swar16_kernel<NW>, grid, block, 0, s, P, delta, nul_flag); break;
            TA_HS(1) TA_HS(2) TA_HS(3) TA_HS(4) TA_HS(5) TA_HS(6) TA_HS(7) TA_HS(8)
            TA_HS(9) TA_HS(10) TA_HS(11) TA_HS(12) TA_HS(13) TA_HS(14) TA_HS(15) TA_HS(16)
#undef TA_HS
        }
        return hipGetLastError();
    }
    if (P.needle_len <= 32 && !env_str("TA_HAMMING_SEARCH_SWAR")) {
        const uint64_t offsets = P.hay_len - P.needle_len + 1;
        uint64_t tile = (offsets + 262143) / 262144;              // two sets of resident lanes
        if (tile < 8ull * P.needle_len) tile = 8ull * P.needle_len;
        if (tile < 64) tile = 64;
        P.tile = (uint32_t)(tile > 0x7FFFFFFFull ? 0x7FFFFFFFull : tile);
        const uint64_t lanes = (offsets + P.tile - 1) / P.tile;
        const uint32_t grid = (uint32_t)((lanes + 255) / 256);
        const uint32_t nws = (P.needle_len + 3) / 4;
        set_last_kernel_name("hamming_search_sa_kernel<%u>", nws <= 2 ? 2u : nws <= 4 ? 4u : 8u);
        if (nws <= 2) hipLaunchKernelGGL(hamming_search_sa_kernel<2>, dim3(grid), dim3(256), 0, s, P);
        else if (nws <= 4) hipLaunchKernelGGL(hamming_search_sa_kernel<4>, dim3(grid), dim3(256), 0, s, P);
        else hipLaunchKernelGGL(hamming_search_sa_kernel<8>, dim3(grid), dim3(256), 0, s, P);
        return hipGetLastError();
    }
    const uint32_t delta = (uint32_t)((uintptr_t)P.hay & 3u);
    const uint64_t groups = (P.hay_len - P.needle_len + delta) / 4 + 1;
    set_last_kernel_name("hamming_search_kernel");
    hipLaunchKernelGGL(hamming_search_kernel, dim3((uint32_t)((groups + 255) / 256)), dim3(256), 0, s, P, delta);
    return hipGetLastError();
}

// A search's outcome into host-mapped pinned memory (one 256-thread block behind the search kernel on its stream): the hit count, the
// NUL-byte flag and -- when they are few enough for the box -- the hit records themselves: the host has everything after ONE stream
// synchronisation, no copy (a count brought back by hipMemcpyAsync into pageable memory + a device-side sort and copy of the hits cost
// a hamming_search call 0.11 ms beyond its kernel).
__global__ void search_report_copy_kernel(const unsigned long long *count, const uint32_t *nul_flag, const ta_match *hits, uint64_t cap, uint8_t *box) {
    SearchReport *rep = (SearchReport *)box;
    const unsigned long long c = *count;
    const uint64_t have = c < cap ? c : cap;
    const bool fits = have <= SEARCH_REPORT_SEL;
    if (threadIdx.x == 0) {
        rep->count = c;
        rep->dense = nul_flag ? *nul_flag : 0u;        // (this report: a NUL byte in the haystack)
        rep->sel_count = fits ? (uint32_t)have : 0u;
        rep->sel_state = fits ? 1u : 2u;               // 1: the records follow this header; 2: too many, they are in the caller's device buffer
    }
    if (!fits) return;
    uint64_t *dst = (uint64_t *)(box + sizeof(SearchReport));
    const uint64_t *src = (const uint64_t *)hits;
    for (uint64_t w = threadIdx.x; w < have * 3u; w += blockDim.x) dst[w] = src[w];
}
hipError_t search_report_copy_launch(const unsigned long long *count, const uint32_t *nul_flag, const ta_match *hits, uint64_t cap, uint8_t *box, hipStream_t s) {
    hipLaunchKernelGGL(search_report_copy_kernel, dim3(1), dim3(256), 0, s, count, nul_flag, hits, cap, box);
    return hipGetLastError();
}

// check_no_null_bytes (src/lib.rs:237-243) over device memory: *flag != 0 iff a zero byte exists
__global__ void has_zero_byte_kernel(const uint8_t *p, uint64_t len, uint32_t *flag) {
    uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    bool z = false;
    if (i + 16 <= len) {
        typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(1)));
        u32x4u v = *(const u32x4u *)(p + i);
#pragma unroll
        for (int w = 0; w < 4; w++) {
            uint32_t x = v[w];
            z |= ((((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u) != 0x80808080u;
        }
    } else {
        for (uint64_t t = i; t < len; t++) z |= (p[t] == 0);
    }
    if (z) atomicOr(flag, 1u);
}
hipError_t has_zero_byte_launch(const uint8_t *p, uint64_t len, uint32_t *flag, hipStream_t s) {
    if (len == 0) return hipSuccess;
    uint64_t threads = (len + 15) / 16;
    hipLaunchKernelGGL(has_zero_byte_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, s, p, len, flag);
    return hipGetLastError();
}

}  // namespace ta
