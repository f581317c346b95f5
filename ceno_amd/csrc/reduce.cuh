// wave64 shuffle + LDS block reduction of extension-field accumulators (modular sums).
#pragma once
#include "gl64.cuh"

namespace red {

__device__ __forceinline__ uint64_t shfl_down64(uint64_t v, int delta) {
    // two 32-bit DPP/permute moves
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_down(lo, delta, 64);
    hi = __shfl_down(hi, delta, 64);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ gl::E2 wave_sum(gl::E2 v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        gl::E2 o{shfl_down64(v.c0, off), shfl_down64(v.c1, off)};
        v = v + o;
    }
    return v;  // valid in lane 0
}

// Sum D accumulators over a block of NT threads (NT multiple of 64, <= 1024).
// Result valid in thread 0.  `smem` must hold (NT/64) * D E2 values.
template <int D, int NT>
__device__ __forceinline__ void block_sum(gl::E2 (&acc)[D], gl::E2* smem) {
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < D; t++) acc[t] = wave_sum(acc[t]);
    if (NW > 1) {
        if (lane == 0) {
#pragma unroll
            for (int t = 0; t < D; t++) smem[wave * D + t] = acc[t];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int t = 0; t < D; t++) {
                gl::E2 s = smem[t];
                for (int w = 1; w < NW; w++) s = s + smem[w * D + t];
                acc[t] = s;
            }
        }
    }
}

}  // namespace red
