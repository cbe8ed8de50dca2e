// bit_transpose.h -- 32 x 32 bit-matrix transpose in registers: Y[m] bit p = X[p] bit m.
// Five butterfly stages (distance 16, 8, 4, 2, 1); each exchanges the off-diagonal blocks of word pairs (i, i + s):
//   X[i]   keeps its bits with (bit & s) == 0 and takes those of X[i+s] shifted up by s,
//   X[i+s] keeps its bits with (bit & s) != 0 and takes those of X[i] shifted down by s.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define TA_BT_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define TA_BT_HD inline
#endif

namespace ta {

template <int S>
TA_BT_HD void bit_transpose_stage(uint32_t (&x)[32]) {
    constexpr uint32_t M = S == 16 ? 0x0000FFFFu : S == 8 ? 0x00FF00FFu : S == 4 ? 0x0F0F0F0Fu : S == 2 ? 0x33333333u : 0x55555555u;
#pragma unroll
    for (int i = 0; i < 32; i++) {
        if (i & S) continue;
        const uint32_t lo = x[i], hi = x[i + S];
        x[i] = (lo & M) | ((hi << S) & ~M);
        x[i + S] = (hi & ~M) | ((lo >> S) & M);
    }
}

TA_BT_HD void bit_transpose32(uint32_t (&x)[32]) {
    bit_transpose_stage<16>(x);
    bit_transpose_stage<8>(x);
    bit_transpose_stage<4>(x);
    bit_transpose_stage<2>(x);
    bit_transpose_stage<1>(x);
}

}  // namespace ta
