// util_kernels.hip -- Hamming batch kernel (HBM-bound) and the small helper kernels of the batch API.
#include <hip/hip_runtime.h>

#include "ta_internal.h"

namespace ta {

__device__ __forceinline__ void dev_str(const StrView &s, uint32_t i, const uint8_t *&p, uint64_t &len) {
    if (s.off) {
        uint64_t o0 = s.off[i], o1 = s.off[i + 1];
        p = s.blob + o0;
        len = o1 - o0;
    } else {
        p = s.blob + (uint64_t)i * s.stride;
        len = s.len;
    }
}

// number of nonzero bytes in x (a ^ b): the mismatches of 4 positions
__device__ __forceinline__ uint32_t nz_bytes(uint32_t x) {
    uint32_t t = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;
    return __builtin_popcount(t);
}

// hamming(a, b) for a batch: G = 2^g lanes per pair (G*16 >= the longest string, at most one wavefront),
// 64/G pairs per wavefront, 16 B per lane per trip, coalesced.
// Replaces hamming_simd_parallel / Avx::count_mismatches (src/hamming.rs:317, src/jewel.rs:2320-2365);
// result contract hamming_naive (src/hamming.rs:36-47): mismatching positions, None on length mismatch.
__global__ __launch_bounds__(256) void hamming_batch_kernel(StrView a, StrView b, uint32_t n, uint32_t *out, uint32_t G) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t ppw = 64u / G;
    const uint32_t pair = wave * ppw + lane / G;
    const uint32_t g = lane & (G - 1);
    const bool valid = pair < n;
    const uint8_t *pa = a.blob, *pb = b.blob;
    uint64_t la = 0, lb = 0;
    if (valid) { dev_str(a, pair, pa, la); dev_str(b, pair, pb, lb); }
    const bool same = (la == lb);
    typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(1)));
    uint32_t cnt = 0;
    if (valid && same) {
        const uint64_t full = la & ~(uint64_t)15;
        for (uint64_t i = (uint64_t)g * 16; i < full; i += (uint64_t)G * 16) {
            u32x4u x = *(const u32x4u *)(pa + i);
            u32x4u y = *(const u32x4u *)(pb + i);
            cnt += nz_bytes(x.x ^ y.x) + nz_bytes(x.y ^ y.y) + nz_bytes(x.z ^ y.z) + nz_bytes(x.w ^ y.w);
        }
        for (uint64_t i = full + g; i < la; i += G) cnt += (pa[i] != pb[i]);
    }
    for (uint32_t m = G >> 1; m >= 1; m >>= 1) cnt += __shfl_xor(cnt, m, 64);
    if (valid && g == 0) out[pair] = same ? cnt : 0xFFFFFFFFu;                    // assert!(len == b.len())  src/hamming.rs:38
}

hipError_t hamming_batch_launch(const StrView &a, const StrView &b, uint32_t n, uint32_t *out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    // strided batches know their length; CSR batches use a full wavefront per pair
    uint64_t len = a.off ? 1024 : a.len;
    uint32_t G = 1;
    while (G < 64 && (uint64_t)G * 16 < len) G <<= 1;
    const uint32_t ppw = 64 / G, waves = (n + ppw - 1) / ppw;
    set_last_kernel_name("hamming_batch_kernel");
    hipLaunchKernelGGL(hamming_batch_kernel, dim3((waves + 3) / 4), dim3(256), 0, s, a, b, n, out, G);
    return hipGetLastError();
}

__global__ void strings_maxlen_kernel(StrView s, uint32_t n, uint32_t *out_max) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t v = 0;
    if (i < n) v = (uint32_t)(s.off[i + 1] - s.off[i]);
    for (int m = 32; m >= 1; m >>= 1) {
        uint32_t y = __shfl_xor(v, m, 64);
        v = v > y ? v : y;
    }
    if ((threadIdx.x & 63u) == 0 && v) atomicMax(out_max, v);
}
hipError_t strings_maxlen_launch(const StrView &s, uint32_t n, uint32_t *out_max, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(strings_maxlen_kernel, dim3((n + 255) / 256), dim3(256), 0, st, s, n, out_max);
    return hipGetLastError();
}

// exp search: collect the pairs whose result is still None into the next round's subset.  n_in_dev (optional): the input list's length as
// a kernel before this one left it on the device (n_in is then its upper bound: the grid).  One atomic per wavefront: the lanes that keep
// their pair take consecutive places in lane order.
__device__ __forceinline__ void compact_append(bool keep, uint32_t pair, uint32_t *list_out, uint32_t *count) {
    const unsigned long long mask = __ballot(keep);
    if (!mask) return;
    const uint32_t lane = threadIdx.x & 63u, leader = (uint32_t)__ffsll((long long)mask) - 1u;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(mask));
    base = (uint32_t)__shfl((int)base, (int)leader);
    if (keep) list_out[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = pair;
}
__global__ void compact_none_kernel(const uint32_t *out, const uint32_t *subset_in, uint32_t n_in, const uint32_t *n_in_dev,
                                    uint32_t *subset_out, uint32_t *count) {
    if (n_in_dev) n_in = *n_in_dev;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < n_in;
    const uint32_t pair = in ? (subset_in ? subset_in[i] : i) : 0u;
    compact_append(in && out[pair] == 0xFFFFFFFFu, pair, subset_out, count);
}
// n words set to v, as a kernel: calls that may be captured into a graph use it instead of hipMemsetAsync -- a 4-byte memset NODE of a
// captured call was not ordered with the kernel nodes around it on gfx950 / ROCm 7.2 (the unit-cost pre-pass under bench.py's graph:
// a stale counter, a memory fault; fine with AMD_SERIALIZE_KERNEL=3 and with this kernel)
__global__ void fill_u32_kernel(uint32_t *p, uint32_t v, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
hipError_t fill_u32_launch(uint32_t *p, uint32_t v, uint32_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(fill_u32_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, p, v, n);
    return hipGetLastError();
}
// the complement: the pairs a pass ANSWERED (the unit-cost pre-pass of weighted batches, TA_OPT_UNIT_PREFILTER: its survivors)
__global__ void compact_some_kernel(const uint32_t *out, const uint32_t *subset_in, uint32_t n_in, uint32_t *subset_out, uint32_t *count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < n_in;
    const uint32_t pair = in ? (subset_in ? subset_in[i] : i) : 0u;
    compact_append(in && out[pair] != 0xFFFFFFFFu, pair, subset_out, count);
}
hipError_t compact_some_launch(const uint32_t *out, const uint32_t *subset_in, uint32_t n_in, uint32_t *subset_out, uint32_t *count, hipStream_t st) {
    if (n_in == 0) return hipSuccess;
    hipLaunchKernelGGL(compact_some_kernel, dim3((n_in + 255) / 256), dim3(256), 0, st, out, subset_in, n_in, subset_out, count);
    return hipGetLastError();
}
hipError_t compact_none_launch(const uint32_t *out, const uint32_t *subset_in, uint32_t n_in, const uint32_t *n_in_dev, uint32_t *subset_out,
                               uint32_t *count, hipStream_t st) {
    if (n_in == 0) return hipSuccess;
    hipLaunchKernelGGL(compact_none_kernel, dim3((n_in + 255) / 256), dim3(256), 0, st, out, subset_in, n_in, n_in_dev,
                       subset_out, count);
    return hipGetLastError();
}

// costs that are the unit costs times g (lev_unit_scale, lev_plan.h): the pass ran on unit costs with k / g, its answers times g
__global__ void scale_results_kernel(uint32_t *out, const uint32_t *list, uint32_t n, const uint32_t *n_dev, uint32_t g) {
    if (n_dev) n = *n_dev;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t pair = list ? list[i] : i;
    const uint32_t d = out[pair];
    if (d != 0xFFFFFFFFu) out[pair] = d * g;
}
hipError_t scale_results_launch(uint32_t *out, const uint32_t *list, uint32_t n, const uint32_t *n_dev, uint32_t g, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(scale_results_kernel, dim3((n + 255) / 256), dim3(256), 0, st, out, list, n, n_dev, g);
    return hipGetLastError();
}

// ta_levenshtein_trace_batch_packed on the routes whose walk writes ta_edit records: pair i's first min(n_edits[i], cap_in) records -> one word
// each ((edit << 29) | count), right-aligned in its slot of cap_out words (a script of more than cap_out runs keeps its last ones)
__global__ void pack_edits_kernel(const ta_edit *edits, const uint32_t *n_edits, uint32_t n, uint64_t cap_in, uint32_t *packed, uint64_t cap_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t have = n_edits[i] < cap_in ? n_edits[i] : cap_in, keep = have < cap_out ? have : cap_out;
    const ta_edit *src = edits + (uint64_t)i * cap_in + (have - keep);
    uint32_t *dst = packed + ((uint64_t)i + 1u) * cap_out - keep;
    for (uint64_t t = 0; t < keep; t++) dst[t] = (src[t].edit << 29) | (uint32_t)(src[t].count & 0x1FFFFFFFu);
}
hipError_t pack_edits_launch(const ta_edit *edits, const uint32_t *n_edits, uint32_t n, uint64_t cap_in, uint32_t *packed, uint64_t cap_out, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_edits_kernel, dim3((n + 255) / 256), dim3(256), 0, st, edits, n_edits, n, cap_in, packed, cap_out);
    return hipGetLastError();
}

// exp search with a lower bound: next round's work list = unresolved pairs whose bound admits the next threshold
__global__ void compact_bound_kernel(const uint32_t *out, const uint32_t *bound, uint32_t k, const uint32_t *list_in, uint32_t n_in,
                                     const uint32_t *n_in_dev, uint32_t *list_out, uint32_t *count) {
    if (n_in_dev) n_in = *n_in_dev;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < n_in;
    const uint32_t pair = in ? (list_in ? list_in[i] : i) : 0u;
    compact_append(in && out[pair] == 0xFFFFFFFFu && bound[pair] <= k, pair, list_out, count);
}
hipError_t compact_bound_launch(const uint32_t *out, const uint32_t *bound, uint32_t k, const uint32_t *list_in, uint32_t n_in,
                                const uint32_t *n_in_dev, uint32_t *list_out, uint32_t *count, hipStream_t st) {
    if (n_in == 0) return hipSuccess;
    hipLaunchKernelGGL(compact_bound_kernel, dim3((n_in + 255) / 256), dim3(256), 0, st, out, bound, k, list_in, n_in, n_in_dev, list_out, count);
    return hipGetLastError();
}

// ---- ragged batches: the pairs ordered by length ("bucket by length first if lengths vary so waves are uniform").
// The band kernels run a wavefront to the longest of its pairs: 64 pairs of 8 different lengths cost the longest one's columns plus
// the capped blocks at the end (+8 % on the ragged cfg2 batch, profiles/r04/ab_ragged.md).  The pairs of a CSR batch are therefore
// taken in the order of a counting sort on
//     key = 0 for a pair outside the band (|len_a - len_b| > unit_k: None before any cell, src/levenshtein.rs:426-428), else
//     LEN_BINS - 1 - (w >> shift),   w = len_b (LEN_BY_COLUMNS: the bit-parallel kernels run one column per byte of b) or
//                                     w = len_a + len_b (LEN_BY_STEPS: the DP band kernel runs one step per anti-diagonal)
// with shift = 3 (8-byte classes) or the smallest one that fits LEN_BINS bins, and the LONGEST pairs first: the launch's tail is then made
// of the shortest wavefronts (the ragged cfg2 batch: 0.227 ms against 0.243 shortest-first, profiles/r04/ab_ragged.md).  `exact`
// (the VLINE fetch form asks for it) takes shift 0 where that fits: every wavefront's pairs then run EXACTLY the same number of columns;
// it is not the default because eight times as many non-empty bins are eight times as many global atomics in the histogram and
// scatter kernels (26.4 us against 23.0 per million pairs) and the chunk-form kernel gains nothing from it.  Three small launches:
// histogram (per-block counters in LDS, one global atomic per non-empty bin and block), exclusive scan of <= LEN_BINS bins,
// scatter (a block reserves its share of every bin with one atomic, its pairs take their places through LDS counters).  The
// order inside a bin is whatever the atomics make it -- the results are per pair and do not depend on it.
constexpr uint32_t LEN_BINS = 1024, LEN_PAIRS_PER_BLOCK = 2048, LEN_PPT = LEN_PAIRS_PER_BLOCK / 256;
// every bin has LEN_SUB counters (a block uses the one of its index mod LEN_SUB): hundreds of blocks bumping the same ~30 addresses
// serialise at the L2 (18 us per million pairs with one counter per bin, 15.7 with 8)
constexpr uint32_t LEN_SUB = 32, LEN_ASC = 0x80000000u, LEN_BY_STEPS = 0x40000000u;     // counters laid out [sub][bin]: the scan reads them coalesced
__device__ __forceinline__ uint32_t len_key(const StrView &a, const StrView &b, uint32_t pair, uint32_t u, uint32_t shift) {
    const uint64_t la = a.off ? a.off[pair + 1] - a.off[pair] : a.len, lb = b.off ? b.off[pair + 1] - b.off[pair] : b.len;
    const uint64_t mx = la > lb ? la : lb, mn = la > lb ? lb : la;
    if (mx - mn > u) return 0u;
    const uint64_t w = (shift & LEN_BY_STEPS) ? la + lb : lb;
    const uint64_t key = 1u + (w >> (shift & 31u));
    const uint32_t kk = key < LEN_BINS ? (uint32_t)key : LEN_BINS - 1u;
    return (shift & LEN_ASC) ? kk : LEN_BINS - kk;            // the longest pairs first (keys 1..LEN_BINS-1 mirrored); LEN_ASC: A/B only
}
// the block's pairs and their keys, all loads in flight before the first LDS atomic (a load -> atomic chain per pair made
// the kernels latency-bound: 13.5 us per million pairs each)
__device__ __forceinline__ void len_load_keys(const StrView &a, const StrView &b, const uint32_t *subset_in, uint32_t n, uint32_t u, uint32_t shift,
                                              uint32_t (&pair)[LEN_PPT], uint32_t (&key)[LEN_PPT]) {
    const uint32_t base = blockIdx.x * LEN_PAIRS_PER_BLOCK;
#pragma unroll
    for (uint32_t q = 0; q < LEN_PPT; q++) {
        const uint32_t i = base + threadIdx.x + 256u * q;
        pair[q] = i < n ? (subset_in ? subset_in[i] : i) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (uint32_t q = 0; q < LEN_PPT; q++) key[q] = pair[q] != 0xFFFFFFFFu ? len_key(a, b, pair[q], u, shift) : 0u;
}
__global__ __launch_bounds__(256) void len_hist_kernel(StrView a, StrView b, const uint32_t *subset_in, uint32_t n, uint32_t u, uint32_t shift, uint32_t *hist) {
    __shared__ uint32_t h[LEN_BINS];
    for (uint32_t i = threadIdx.x; i < LEN_BINS; i += 256u) h[i] = 0;
    uint32_t pair[LEN_PPT], key[LEN_PPT];
    len_load_keys(a, b, subset_in, n, u, shift, pair, key);
    __syncthreads();
#pragma unroll
    for (uint32_t q = 0; q < LEN_PPT; q++)
        if (pair[q] != 0xFFFFFFFFu) atomicAdd(&h[key[q]], 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < LEN_BINS; i += 256u)
        if (h[i]) atomicAdd(&hist[(blockIdx.x % LEN_SUB) * LEN_BINS + i], h[i]);
}
// hist -> exclusive prefix sums in `cursor`; hist itself goes back to zero (the next call's starting state: no fill per call)
__global__ __launch_bounds__(LEN_BINS) void len_scan_kernel(uint32_t *hist, uint32_t *cursor) {
    __shared__ uint32_t s[LEN_BINS];
    const uint32_t t = threadIdx.x;
    uint32_t part[LEN_SUB], tot = 0;
#pragma unroll
    for (uint32_t q = 0; q < LEN_SUB; q++) { part[q] = hist[q * LEN_BINS + t]; hist[q * LEN_BINS + t] = 0; tot += part[q]; }
    s[t] = tot;
    __syncthreads();
    for (uint32_t d = 1; d < LEN_BINS; d <<= 1) {
        const uint32_t v = t >= d ? s[t - d] : 0u;
        __syncthreads();
        s[t] += v;
        __syncthreads();
    }
    uint32_t run = t ? s[t - 1] : 0u;
#pragma unroll
    for (uint32_t q = 0; q < LEN_SUB; q++) { cursor[q * LEN_BINS + t] = run; run += part[q]; }
}
__global__ __launch_bounds__(256) void len_scatter_kernel(StrView a, StrView b, const uint32_t *subset_in, uint32_t n, uint32_t u, uint32_t shift,
                                                          uint32_t *cursor, uint32_t *subset_out) {
    __shared__ uint32_t h[LEN_BINS];
    for (uint32_t i = threadIdx.x; i < LEN_BINS; i += 256u) h[i] = 0;
    uint32_t pair[LEN_PPT], key[LEN_PPT], rank[LEN_PPT];
    len_load_keys(a, b, subset_in, n, u, shift, pair, key);
    __syncthreads();
#pragma unroll
    for (uint32_t q = 0; q < LEN_PPT; q++)
        if (pair[q] != 0xFFFFFFFFu) rank[q] = atomicAdd(&h[key[q]], 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < LEN_BINS; i += 256u)
        if (h[i]) h[i] = atomicAdd(&cursor[(blockIdx.x % LEN_SUB) * LEN_BINS + i], h[i]);     // the block's first place in bin i
    __syncthreads();
#pragma unroll
    for (uint32_t q = 0; q < LEN_PPT; q++)
        if (pair[q] != 0xFFFFFFFFu) subset_out[h[key[q]] + rank[q]] = pair[q];
}
// bins: 2 * LEN_BINS * LEN_SUB u32 (256 KiB) of device scratch whose first half is ZERO on entry (and again on exit: zero it once, when it is
// allocated); subset_out: n u32.  max_len = the batch's longest string.
// by_steps: the key counts anti-diagonal steps (len_a + len_b: the DP band kernel) instead of columns (len_b: the bit-parallel kernels)
hipError_t length_order_launch(const StrView &a, const StrView &b, const uint32_t *subset_in, uint32_t n, uint32_t u, uint64_t max_len,
                               bool by_steps, uint32_t *bins, uint32_t *subset_out, hipStream_t st, bool exact, bool *exact_columns) {
    if (exact_columns) *exact_columns = false;
    if (n == 0) return hipSuccess;
    const uint64_t w_max = by_steps ? 2 * max_len : max_len;
    uint32_t shift = exact ? 0u : 3u;                                 // 8-byte (or exact) classes while LEN_BINS - 2 of them cover the longest pair
    while ((w_max >> shift) + 2 > LEN_BINS) shift++;
    if (exact_columns) *exact_columns = !by_steps && shift == 0;
    if (by_steps) shift |= LEN_BY_STEPS;
    if (env_int("TA_ORDER_ASC")) shift |= LEN_ASC;
    const uint32_t blocks = (n + LEN_PAIRS_PER_BLOCK - 1) / LEN_PAIRS_PER_BLOCK;
    hipLaunchKernelGGL(len_hist_kernel, dim3(blocks), dim3(256), 0, st, a, b, subset_in, n, u, shift, bins);
    hipLaunchKernelGGL(len_scan_kernel, dim3(1), dim3(LEN_BINS), 0, st, bins, bins + LEN_BINS * LEN_SUB);
    hipLaunchKernelGGL(len_scatter_kernel, dim3(blocks), dim3(256), 0, st, a, b, subset_in, n, u, shift, bins + LEN_BINS * LEN_SUB, subset_out);
    return hipGetLastError();
}

// Bag (multiset) lower bound of the edit cost of every pair: with ex_a / ex_b the bytes `a` / `b` has in excess of the other
// (per byte value), any script needs max(ex_a, ex_b) edits that change the bags -- a mismatch fixes one excess on each
// side, a gap one; a transposition none -- so with s mismatches  cost >= s mismatch + (max(0, ex_a - s) + max(0, ex_b - s)) gap,
// piecewise linear in s with its minimum at s = 0, min(ex) or max(ex).
// levenshtein_exp uses it to skip the doubling rounds (src/levenshtein.rs:1445-1454) a pair cannot finish in.
// One wavefront per pair, one signed counter per byte value in LDS (+1 for a, -1 for b).
__global__ __launch_bounds__(256) void bag_bound_kernel(StrView a, StrView b, uint32_t n, uint32_t mc, uint32_t gc, uint32_t *bound) {
    __shared__ int table[4][256];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    int *t = table[w];
    for (uint32_t pair = blockIdx.x * 4u + w; pair < n; pair += gridDim.x * 4u) {
        const uint8_t *pa, *pb;
        uint64_t la, lb;
        dev_str(a, pair, pa, la);
        dev_str(b, pair, pb, lb);
        for (uint32_t i = lane; i < 256u; i += 64u) t[i] = 0;
        typedef uint32_t u32u __attribute__((aligned(1)));
        const uint64_t fa = la & ~(uint64_t)3, fb = lb & ~(uint64_t)3;
        for (uint64_t i = (uint64_t)lane * 4; i < fa; i += 256) {
            const uint32_t x = *(const u32u *)(pa + i);
            atomicAdd(&t[x & 255u], 1); atomicAdd(&t[(x >> 8) & 255u], 1); atomicAdd(&t[(x >> 16) & 255u], 1); atomicAdd(&t[x >> 24], 1);
        }
        for (uint64_t i = fa + lane; i < la; i += 64) atomicAdd(&t[pa[i]], 1);
        for (uint64_t i = (uint64_t)lane * 4; i < fb; i += 256) {
            const uint32_t x = *(const u32u *)(pb + i);
            atomicAdd(&t[x & 255u], -1); atomicAdd(&t[(x >> 8) & 255u], -1); atomicAdd(&t[(x >> 16) & 255u], -1); atomicAdd(&t[x >> 24], -1);
        }
        for (uint64_t i = fb + lane; i < lb; i += 64) atomicAdd(&t[pb[i]], -1);
        uint32_t ea = 0, eb = 0;
        for (uint32_t i = lane; i < 256u; i += 64u) {
            const int v = t[i];
            if (v > 0) ea += (uint32_t)v; else eb += (uint32_t)(-v);
        }
        for (uint32_t m = 32; m >= 1; m >>= 1) { ea += __shfl_xor(ea, m, 64); eb += __shfl_xor(eb, m, 64); }
        if (lane == 0) {
            const uint64_t lo = ea < eb ? ea : eb, hi = ea < eb ? eb : ea;
            const uint64_t c0 = (uint64_t)(ea + (uint64_t)eb) * gc, c1 = lo * mc + (hi - lo) * gc, c2 = hi * mc;
            const uint64_t c01 = c0 < c1 ? c0 : c1, c = c01 < c2 ? c01 : c2;
            bound[pair] = c > 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)c;
        }
    }
}
hipError_t bag_bound_launch(const StrView &a, const StrView &b, uint32_t n, uint32_t mc, uint32_t gc, uint32_t *bound, hipStream_t st) {
    if (n == 0) return hipSuccess;
    uint32_t blocks = (n + 3u) / 4u;
    if (blocks > 256u * 32u) blocks = 256u * 32u;
    hipLaunchKernelGGL(bag_bound_kernel, dim3(blocks), dim3(256), 0, st, a, b, n, mc, gc, bound);
    return hipGetLastError();
}

// Best-mode reduction of a shard's All-mode hits on the device: the smallest k, then the hits that have it.  (Only those can
// survive ta_search_fold_best: its final filter keeps k == the running minimum, every hit with the minimum passes the
// running test, and the overlap rule only ever compares a survivor with the survivor before it.)
__global__ void hits_min_k_kernel(const ta_match *hits, uint64_t n, uint32_t *min_k) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t v = i < n ? hits[i].k : 0xFFFFFFFFu;
    for (int m = 32; m >= 1; m >>= 1) {
        uint32_t y = __shfl_xor(v, m, 64);
        v = v < y ? v : y;
    }
    if ((threadIdx.x & 63u) == 0 && v != 0xFFFFFFFFu) atomicMin(min_k, v);
}
__global__ void hits_select_k_kernel(const ta_match *hits, uint64_t n, const uint32_t *min_k, ta_match *out, uint32_t cap, uint32_t *count) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const ta_match h = hits[i];
    if (h.k != *min_k) return;
    uint32_t at = atomicAdd(count, 1u);
    if (at < cap) out[at] = h;
}
hipError_t hits_best_launch(const ta_match *hits, uint64_t n, uint32_t *min_k /*device, preset to ~0*/, ta_match *out, uint32_t cap,
                            uint32_t *count /*device, pre-zeroed*/, hipStream_t st) {
    if (n == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(hits_min_k_kernel, dim3(blocks), dim3(256), 0, st, hits, n, min_k);
    hipLaunchKernelGGL(hits_select_k_kernel, dim3(blocks), dim3(256), 0, st, hits, n, min_k, out, cap, count);
    return hipGetLastError();
}

}  // namespace ta
