// lev_search.hip -- gfx950 kernels for levenshtein_search / hamming_search over a haystack shard in HBM.
#include <hip/hip_runtime.h>

#include "lev_search_body.h"
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

// hamming_search: one lane per haystack offset, needle (kernarg) compared 4 bytes at a time.
// Replaces hamming_search_simd_core_* (src/hamming.rs:481-552); contract: mismatches(offset) <= k.
__global__ __launch_bounds__(256) void hamming_search_kernel(SearchParams P) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = P.needle_len;
    if (i + n > P.hay_len) return;
    const uint8_t *h = P.hay + i;
    uint32_t cnt = 0;
    uint32_t j = 0;
    typedef uint32_t u32u __attribute__((aligned(1)));
    for (; j + 4 <= n; j += 4) {
        uint32_t x = *(const u32u *)(h + j) ^ *(const u32u *)(P.needle_dev + j);
        cnt += __builtin_popcount((((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u);
    }
    for (; j < n; j++) cnt += (h[j] != P.needle_dev[j]);
    if (cnt <= P.k) {
        unsigned long long idx = atomicAdd(P.count, 1ull);
        if (idx < P.cap) P.hits[idx] = ta_match{P.base + i, P.base + i + n, cnt, 0u};
    }
}

hipError_t hamming_search_launch(const SearchParams &P, hipStream_t s) {
    if (P.hay_len < P.needle_len || P.needle_len == 0) return hipSuccess;
    const uint64_t positions = P.hay_len - P.needle_len + 1;
    hipLaunchKernelGGL(hamming_search_kernel, dim3((uint32_t)((positions + 255) / 256)), dim3(256), 0, s, P);
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
