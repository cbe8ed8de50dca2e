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

// exp search: collect the pairs whose result is still None into the next round's subset
__global__ void compact_none_kernel(const uint32_t *out, const uint32_t *subset_in, uint32_t n_in,
                                    uint32_t *subset_out, uint32_t *count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_in) return;
    uint32_t pair = subset_in ? subset_in[i] : i;
    if (out[pair] == 0xFFFFFFFFu) subset_out[atomicAdd(count, 1u)] = pair;
}
hipError_t compact_none_launch(const uint32_t *out, const uint32_t *subset_in, uint32_t n_in, uint32_t *subset_out,
                               uint32_t *count, hipStream_t st) {
    if (n_in == 0) return hipSuccess;
    hipLaunchKernelGGL(compact_none_kernel, dim3((n_in + 255) / 256), dim3(256), 0, st, out, subset_in, n_in,
                       subset_out, count);
    return hipGetLastError();
}

}  // namespace ta
