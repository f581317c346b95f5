// wave64 DPP + LDS block reduction of extension-field accumulators (modular sums).
//
// Intra-row (16 lanes) steps use DPP lane permutes (VALU, no LDS round trip): quad_perm [1,0,3,2],
// quad_perm [2,3,0,1], row_half_mirror, row_mirror — after them every lane of a row holds the row total;
// the four row totals are combined through v_readlane.  (ds_bpermute-based __shfl_down cost ~5 us for
// three extension accumulators in the latency-critical tail rounds.)
#pragma once
#include "gl64.hpp"

namespace red {

template <int CTRL>
__device__ __forceinline__ uint64_t dpp64(uint64_t v) {
    int lo = (int)(uint32_t)v, hi = (int)(uint32_t)(v >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
template <int CTRL>
__device__ __forceinline__ gl::E2 dpp_e2(gl::E2 v) {
    return gl::E2{dpp64<CTRL>(v.c0), dpp64<CTRL>(v.c1)};
}
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int lane) {
    uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}

// sum over the 64 lanes of a wave; the result is valid in EVERY lane
__device__ __forceinline__ gl::E2 wave_sum(gl::E2 v) {
    v = v + dpp_e2<0xB1>(v);   // quad_perm [1,0,3,2]
    v = v + dpp_e2<0x4E>(v);   // quad_perm [2,3,0,1]
    v = v + dpp_e2<0x141>(v);  // row_half_mirror
    v = v + dpp_e2<0x140>(v);  // row_mirror
    gl::E2 r0{readlane64(v.c0, 0), readlane64(v.c1, 0)};
    gl::E2 r1{readlane64(v.c0, 16), readlane64(v.c1, 16)};
    gl::E2 r2{readlane64(v.c0, 32), readlane64(v.c1, 32)};
    gl::E2 r3{readlane64(v.c0, 48), readlane64(v.c1, 48)};
    return (r0 + r1) + (r2 + r3);
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
