// Streaming-bandwidth ceilings for the access patterns of the dense sumcheck kernel (not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_bw.hip -o tools/ubench_bw
// Question (round-2 verdict): the guide measures 6.29 TB/s for a float4 copy, the fold rounds of k_dense run at 4.4-5.0 TB/s and
// this repository's own "same pattern without arithmetic" ceiling was 5.4 TB/s.  Which part of the gap is the access pattern
// (64 B per lane = four dwordx4 loads whose lanes are 64 B apart), which the 2:1 read/write mix, which the three table streams?
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef unsigned long long u2 __attribute__((ext_vector_type(2)));
template <bool NT> __device__ __forceinline__ u2 ld(const u2* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(u2* p, u2 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// 1:1 copy, 16 B per lane, lane-contiguous (the guide's float4 copy)
template <bool NTL, bool NTS>
__global__ void __launch_bounds__(256) k_copy(const u2* __restrict__ in, u2* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) st<NTS>(out + i, ld<NTL>(in + i));
}
__global__ void __launch_bounds__(256) k_read(const u2* __restrict__ in, size_t n, u2* sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    u2 acc = {0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) acc ^= in[i];
    if (acc.x == 0x1234567) sink[0] = acc;
}
// fold-shaped, K tables: read 4 elements, write 2.  LAYOUT 0 = k_dense today: lane p owns elements 4p..4p+3 (64 B per lane).
// LAYOUT 1 = lane-contiguous: a wave reads four 1 KB rows (element 64k + lane), neighbours exchange through DPP, the two outputs of
// a lane pair go out as two 512 B runs per store instruction.
template <int K, int LAYOUT, bool NTL, bool NTS>
__global__ void __launch_bounds__(256) k_fold(const u2* const* __restrict__ ins, u2* const* __restrict__ outs, size_t pairs /* of outputs: n/4 */) {
    const size_t stride = (size_t)gridDim.x * 256;
    if (LAYOUT == 0) {
        for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < pairs; p += stride) {
#pragma unroll
            for (int m = 0; m < K; m++) {
                const u2* q = ins[m] + 4 * p;
                u2 a = ld<NTL>(q), b = ld<NTL>(q + 1), c = ld<NTL>(q + 2), d = ld<NTL>(q + 3);
                st<NTS>(outs[m] + 2 * p, a ^ b);
                st<NTS>(outs[m] + 2 * p + 1, c ^ d);
            }
        }
    } else {
        const int lane = threadIdx.x & 63;
        const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = stride >> 6;
        for (size_t w = wave; w < pairs / 64; w += nwaves) {  // a wave owns 256 inputs = 64 lane-pairs-of-pairs
#pragma unroll
            for (int m = 0; m < K; m++) {
                const u2* q = ins[m] + w * 256 + lane;
                u2 a0 = ld<NTL>(q), a1 = ld<NTL>(q + 64), a2 = ld<NTL>(q + 128), a3 = ld<NTL>(q + 192);
                // even lane folds rows 0,1, odd lane rows 2,3: send the partner what it needs
                const bool odd = lane & 1;
                u2 s0 = odd ? a0 : a2, s1 = odd ? a1 : a3, r0, r1;
                r0.x = __shfl_xor(s0.x, 1); r0.y = __shfl_xor(s0.y, 1);
                r1.x = __shfl_xor(s1.x, 1); r1.y = __shfl_xor(s1.y, 1);
                u2 f0 = odd ? (r0 ^ a2) : (a0 ^ r0), f1 = odd ? (r1 ^ a3) : (a1 ^ r1);
                // outputs: folded index 32k + lane/2, k = 0,1 (even lane) or 2,3 (odd lane)
                u2* o = outs[m] + w * 128 + (lane >> 1) + (odd ? 64 : 0);
                st<NTS>(o, f0);
                st<NTS>(o + 32, f1);
            }
        }
    }
}

int main(int argc, char** argv) {
    const size_t n = (size_t)1 << 26;  // per table: 2^26 x 16 B = 1 GiB (round 1 of the nv=26 sumcheck reads exactly this per table)
    const int KMAX = 3;
    u2 *in[KMAX], *out[KMAX], *sink;
    for (int m = 0; m < KMAX; m++) {
        CK(hipMalloc(&in[m], n * 16));
        CK(hipMalloc(&out[m], n * 16));
        CK(hipMemset(in[m], m + 1, n * 16));
    }
    CK(hipMalloc(&sink, 64));
    const u2 **d_in;
    u2** d_out;
    CK(hipMalloc(&d_in, sizeof(void*) * KMAX));
    CK(hipMalloc(&d_out, sizeof(void*) * KMAX));
    CK(hipMemcpy(d_in, in, sizeof(void*) * KMAX, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_out, out, sizeof(void*) * KMAX, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time = [&](auto&& f) {
        f();
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 4; i++) f();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms / 4 < best ? ms / 4 : best;
        }
        return best;
    };
    printf("{\n");
    for (int blocks : {1024, 2048, 4096}) {
        const double gb1 = n * 16.0 / 1e9;
        float tr = time([&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, in[0], n, sink); });
        float tc = time([&] { hipLaunchKernelGGL((k_copy<false, false>), dim3(blocks), dim3(256), 0, 0, in[0], out[0], n); });
        float tcn = time([&] { hipLaunchKernelGGL((k_copy<true, true>), dim3(blocks), dim3(256), 0, 0, in[0], out[0], n); });
        float tcs = time([&] { hipLaunchKernelGGL((k_copy<false, true>), dim3(blocks), dim3(256), 0, 0, in[0], out[0], n); });
        printf(" \"blocks_%d\": {\"read_GBps\": %.0f, \"copy_GBps\": %.0f, \"copy_nt_GBps\": %.0f, \"copy_nt_store_GBps\": %.0f,\n", blocks, gb1 / tr * 1e3, 2 * gb1 / tc * 1e3,
               2 * gb1 / tcn * 1e3, 2 * gb1 / tcs * 1e3);
        auto fold = [&](auto kern, int K) {
            float t = time([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_in, d_out, n / 4); });
            return K * gb1 * 1.5 / t * 1e3;
        };
        printf("   \"fold_K1_lane64B\": %.0f, \"fold_K1_rows\": %.0f, \"fold_K3_lane64B\": %.0f, \"fold_K3_rows\": %.0f,\n", fold(k_fold<1, 0, false, false>, 1),
               fold(k_fold<1, 1, false, false>, 1), fold(k_fold<3, 0, false, false>, 3), fold(k_fold<3, 1, false, false>, 3));
        printf("   \"fold_K3_lane64B_ntstore\": %.0f, \"fold_K3_lane64B_ntboth\": %.0f, \"fold_K3_rows_ntstore\": %.0f, \"fold_K3_rows_ntboth\": %.0f, \"fold_K3_rows_ntload\": %.0f}%s\n",
               fold(k_fold<3, 0, false, true>, 3), fold(k_fold<3, 0, true, true>, 3), fold(k_fold<3, 1, false, true>, 3), fold(k_fold<3, 1, true, true>, 3),
               fold(k_fold<3, 1, true, false>, 3), blocks == 4096 ? "" : ",");
    }
    printf("}\n");
    return 0;
}
