// Micro-benchmark of the extension-field multiply (GoldilocksExt2) and multiply-accumulate forms: throughput per variant,
// results cross-checked.   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_e2.hip -o tools/ubench_e2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../ceno_amd/csrc/gl64.hpp"
using namespace gl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// 64x64 -> 128 with no zero-extension moves: the two cross products are chained through the 64-bit addend of
// v_mad_u64_u32 (its carry lands in an SGPR pair), the limbs are then assembled with one add and three add-with-carry
__device__ __forceinline__ L4 mul_wide2(uint64_t a, uint64_t b) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    uint64_t lo, hi, m, cy, d0, d1, d2;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(lo), "=s"(d0) : "v"(a0), "v"(b0));
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(hi), "=s"(d1) : "v"(a1), "v"(b1));
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(m), "=s"(d2) : "v"(a0), "v"(b1));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(m), "=s"(cy) : "v"(a1), "v"(b0));
    uint32_t c;
    L4 r;
    r.w0 = (uint32_t)lo;
    r.w1 = addc32((uint32_t)(lo >> 32), (uint32_t)m, 0u, c);
    r.w2 = addc32((uint32_t)hi, (uint32_t)(m >> 32), c, c);
    uint32_t w3 = addc32((uint32_t)(hi >> 32), 0u, c, c);
    asm("s_nop 1\n\tv_addc_co_u32 %0, vcc, 0, %0, %1" : "+v"(w3) : "s"(cy) : "vcc");
    r.w3 = w3;
    return r;
}
__device__ __forceinline__ uint64_t mul_add2_v2(uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
    const L4 p = mul_wide2(a, b), q = mul_wide2(c, d);
    uint32_t cy;
    const uint32_t s0 = addc32(p.w0, q.w0, 0u, cy);
    const uint32_t s1 = addc32(p.w1, q.w1, cy, cy);
    const uint32_t s2 = addc32(p.w2, q.w2, cy, cy);
    const uint32_t s3 = addc32(p.w3, q.w3, cy, cy);
    return reduce_limbs(s0, s1, s2, s3, cy);
}
__device__ __forceinline__ E2 e2_mul_v2(E2 a, E2 b) {
    const uint64_t a1w = mul_small(a.c1, (uint32_t)W);
    return E2{mul_add2_v2(a.c0, b.c0, a1w, b.c1), mul_add2_v2(a.c0, b.c1, a.c1, b.c0)};
}
__device__ __forceinline__ void e2acc_mac_v2(E2Acc& acc, E2 a, E2 b) {
    acc5_add(acc.s00, mul_wide2(a.c0, b.c0));
    acc5_add(acc.s11, mul_wide2(a.c1, b.c1));
    acc5_add(acc.s01, mul_wide2(a.c0, b.c1));
    acc5_add(acc.s01, mul_wide2(a.c1, b.c0));
}

// Column accumulators fed through the 64-bit addend of v_mad_u64_u32 (the addition is free), one carry count per column: per wide product
// 4 multiply-adds + 4 add-with-carry and NO zero-extension moves or 160-bit adds — the form with the fewest instructions per product
// (8 against 4 + 3 moves + 1 + 5), at the price of 9 registers per base accumulator instead of 5.
struct ColAcc {
    uint64_t c0, c1, c2;  // columns of weight 2^0, 2^32, 2^64
    uint32_t t0, t1, t2;  // how often each column wrapped
};
__device__ __forceinline__ void colacc_mac(ColAcc& A, uint64_t a, uint64_t b) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    uint64_t k0, k1, k2, k3;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(A.c0), "=s"(k0) : "v"(a0), "v"(b0));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(A.c1), "=s"(k1) : "v"(a0), "v"(b1));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(A.c2), "=s"(k3) : "v"(a1), "v"(b1));
    asm("v_addc_co_u32 %0, vcc, 0, %0, %1" : "+v"(A.t0) : "s"(k0) : "vcc");
    asm("v_addc_co_u32 %0, vcc, 0, %0, %1" : "+v"(A.t1) : "s"(k1) : "vcc");
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(A.c1), "=s"(k2) : "v"(a1), "v"(b0));
    asm("v_addc_co_u32 %0, vcc, 0, %0, %1" : "+v"(A.t2) : "s"(k3) : "vcc");
    asm("s_nop 0\n\tv_addc_co_u32 %0, vcc, 0, %0, %1" : "+v"(A.t1) : "s"(k2) : "vcc");
}
__device__ __forceinline__ uint64_t colacc_reduce(const ColAcc& A) {
    // value = c0 + t0 2^64 + 2^32 (c1 + t1 2^64) + 2^64 (c2 + t2 2^64): once per thread, written plainly
    uint64_t r = reduce128(A.c0, (uint64_t)A.t0);
    r = add(r, mul(reduce128(A.c1, (uint64_t)A.t1), (uint64_t)1 << 32));
    r = add(r, mul(reduce128(A.c2, (uint64_t)A.t2), EPS));
    return r;
}
struct E2ColAcc {
    ColAcc s00, s11, s01;
};
__device__ __forceinline__ void e2col_mac(E2ColAcc& acc, E2 a, E2 b) {
    colacc_mac(acc.s00, a.c0, b.c0);
    colacc_mac(acc.s11, a.c1, b.c1);
    colacc_mac(acc.s01, a.c0, b.c1);
    colacc_mac(acc.s01, a.c1, b.c0);
}
__device__ __forceinline__ E2 e2col_reduce(const E2ColAcc& acc) {
    return E2{add(colacc_reduce(acc.s00), mul_small(colacc_reduce(acc.s11), (uint32_t)W)), colacc_reduce(acc.s01)};
}
__global__ void __launch_bounds__(256) k_mac_col(uint64_t* out, int iters, uint64_t seed) {
    E2 x[4], y;
    for (int t = 0; t < 4; t++) x[t] = E2{splitmix_gl(seed, threadIdx.x * 8 + t), splitmix_gl(seed + 1, blockIdx.x * 8 + t)};
    y = E2{splitmix_gl(seed + 2, threadIdx.x), splitmix_gl(seed + 3, blockIdx.x)};
    const E2 nd{splitmix_gl(seed + 4, threadIdx.x), 7};
    E2ColAcc acc[4];
    for (int t = 0; t < 4; t++) acc[t] = E2ColAcc{};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            e2col_mac(acc[t], x[t], y);
            y = y - nd;
        }
    }
    uint64_t h = 0;
    for (int t = 0; t < 4; t++) {
        E2 v = e2col_reduce(acc[t]);
        h ^= v.c0 ^ (v.c1 * 3);
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = h;
}

template <int V>
__global__ void __launch_bounds__(256) k_mul(uint64_t* out, int iters, uint64_t seed) {
    E2 x[4], y;
    for (int t = 0; t < 4; t++) x[t] = E2{splitmix_gl(seed, threadIdx.x * 8 + t), splitmix_gl(seed + 1, blockIdx.x * 8 + t)};
    y = E2{splitmix_gl(seed + 2, threadIdx.x), splitmix_gl(seed + 3, blockIdx.x)};
    const E2 nd{splitmix_gl(seed + 4, threadIdx.x), 7};
    for (int i = 0; i < iters; i++) {
        // the shape of a sumcheck factor multiply: 4 evaluation points, the operand steps by a subtraction
#pragma unroll
        for (int t = 0; t < 4; t++) {
            x[t] = V == 0 ? x[t] * y : e2_mul_v2(x[t], y);
            y = y - nd;
        }
    }
    uint64_t h = 0;
    for (int t = 0; t < 4; t++) h ^= x[t].c0 ^ (x[t].c1 * 3);
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = h;
}
template <int V>
__global__ void __launch_bounds__(256) k_mac(uint64_t* out, int iters, uint64_t seed) {
    E2 x[4], y;
    for (int t = 0; t < 4; t++) x[t] = E2{splitmix_gl(seed, threadIdx.x * 8 + t), splitmix_gl(seed + 1, blockIdx.x * 8 + t)};
    y = E2{splitmix_gl(seed + 2, threadIdx.x), splitmix_gl(seed + 3, blockIdx.x)};
    const E2 nd{splitmix_gl(seed + 4, threadIdx.x), 7};
    E2Acc acc[4];
    for (int t = 0; t < 4; t++) acc[t] = e2acc_zero();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (V == 0) e2acc_mac(acc[t], x[t], y);
            else e2acc_mac_v2(acc[t], x[t], y);
            y = y - nd;
        }
    }
    uint64_t h = 0;
    for (int t = 0; t < 4; t++) {
        E2 v = e2acc_reduce(acc[t]);
        h ^= v.c0 ^ (v.c1 * 3);
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = h;
}
int main() {
    uint64_t* o; size_t n = 2048 * 256;
    CK(hipMalloc(&o, 5 * n * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto&& f) { f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int i = 0; i < 5; i++) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 5; };
    int iters = 500;
    float t0 = time([&] { hipLaunchKernelGGL(k_mul<0>, dim3(2048), dim3(256), 0, 0, o, iters, 12345ull); });
    float t1 = time([&] { hipLaunchKernelGGL(k_mul<1>, dim3(2048), dim3(256), 0, 0, o + n, iters, 12345ull); });
    float t2 = time([&] { hipLaunchKernelGGL(k_mac<0>, dim3(2048), dim3(256), 0, 0, o + 2 * n, iters, 12345ull); });
    float t3 = time([&] { hipLaunchKernelGGL(k_mac<1>, dim3(2048), dim3(256), 0, 0, o + 3 * n, iters, 12345ull); });
    float t4 = time([&] { hipLaunchKernelGGL(k_mac_col, dim3(2048), dim3(256), 0, 0, o + 4 * n, iters, 12345ull); });
    double ops = 2048.0 * 256 * iters * 4;
    printf("ext mul + step: shipped %.3e /s | mad-chained mul_wide %.3e /s\n", ops / (t0 * 1e-3), ops / (t1 * 1e-3));
    printf("ext mac + step: shipped %.3e /s | mad-chained mul_wide %.3e /s | column accumulators through the mad addend %.3e /s\n", ops / (t2 * 1e-3),
           ops / (t3 * 1e-3), ops / (t4 * 1e-3));
    uint64_t* h = (uint64_t*)malloc(5 * n * 8);
    CK(hipMemcpy(h, o, 5 * n * 8, hipMemcpyDeviceToHost));
    size_t bad1 = 0, bad3 = 0, bad4 = 0;
    for (size_t i = 0; i < n; i++) { bad1 += h[i] != h[n + i]; bad3 += h[2 * n + i] != h[3 * n + i]; bad4 += h[2 * n + i] != h[4 * n + i]; }
    printf("mismatches: mul %zu, mac %zu, mac (column accumulators) %zu\n", bad1, bad3, bad4);
    return 0;
}
