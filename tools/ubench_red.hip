// Micro-benchmark of Goldilocks reduction variants (throughput of a*b mod p with a non-canonical 64-bit result).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_red.hip -o tools/ubench_red
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../ceno_amd/csrc/gl64.hpp"
using namespace gl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint64_t red_v2(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    uint64_t lo = join(w0, w1), t0, r;
    bool b = __builtin_sub_overflow(lo, (uint64_t)w3, &t0);
    t0 -= b ? EPS : 0;
    uint64_t t1 = ((uint64_t)w2 << 32) - w2;
    bool c = __builtin_add_overflow(t0, t1, &r);
    r += c ? EPS : 0;
    return r;
}
__device__ __forceinline__ uint64_t red_asm(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    uint32_t r0, r1, m, t1l, t1h;
    asm volatile(
        "v_sub_co_u32 %0, vcc, %5, %8\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %6, vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32 %2, 0, -1, vcc\n\t"
        "v_sub_co_u32 %0, vcc, %0, %2\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_sub_co_u32 %3, vcc, 0, %7\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %4, vcc, 0, %7, vcc\n\t"
        "v_add_co_u32 %0, vcc, %0, %3\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %1, vcc, %1, %4, vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32 %2, 0, -1, vcc\n\t"
        "v_add_co_u32 %0, vcc, %0, %2\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        : "=&v"(r0), "=&v"(r1), "=&v"(m), "=&v"(t1l), "=&v"(t1h)
        : "v"(w0), "v"(w1), "v"(w2), "v"(w3)
        : "vcc");
    return join(r0, r1);
}
// mad-based: t = w2 * EPS + lo in ONE v_mad_u64_u32 (carry out in an SGPR pair), then - w3, then (cy - bw) * EPS
__device__ __forceinline__ uint64_t red_w(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    uint64_t t = join(w0, w1), cy;
    asm volatile("v_mad_u64_u32 %0, %1, %2, -1, %0\n\ts_nop 1" : "+v"(t), "=s"(cy) : "v"(w2));
    uint32_t r0 = (uint32_t)t, r1 = (uint32_t)(t >> 32), a, b;
    asm volatile(
        "v_sub_co_u32 %0, vcc, %0, %5\n\t"
        "v_cndmask_b32 %2, 0, -1, %4\n\t"
        "s_nop 0\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32 %3, 0, -1, vcc\n\t"
        "v_add_co_u32 %0, vcc, %0, %2\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_sub_co_u32 %0, vcc, %0, %3\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %1, vcc\n\t"
        : "+v"(r0), "+v"(r1), "=&v"(a), "=&v"(b)
        : "s"(cy), "v"(w3)
        : "vcc");
    return join(r0, r1);
}
template <int V>
__device__ __forceinline__ uint64_t mulv(uint64_t a, uint64_t b) {
    L4 p = mul_wide(a, b);
    if (V == 0) return reduce_limbs_nc(p.w0, p.w1, p.w2, p.w3, 0u);
    if (V == 2) return red_v2(p.w0, p.w1, p.w2, p.w3);
    if (V == 4) return red_w(p.w0, p.w1, p.w2, p.w3);
    if (V == 5) return reduce128_ncm(p.w0, p.w1, p.w2, p.w3);
    return red_asm(p.w0, p.w1, p.w2, p.w3);
}
template <int V>
__global__ void __launch_bounds__(256) k_alu(uint64_t* out, int iters, uint64_t seed) {
    uint64_t a = seed + threadIdx.x, b = seed ^ (blockIdx.x * 977 + 5), c = a * 3 + 1, d = b * 5 + 7;
    for (int i = 0; i < iters; i++) {
        a = mulv<V>(a, b); b = mulv<V>(b, a); c = mulv<V>(c, d); d = mulv<V>(d, c);
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = canon(a) ^ canon(b) ^ canon(c) ^ canon(d);
}
int main() {
    uint64_t* o; size_t n = 2048 * 256;
    CK(hipMalloc(&o, 5 * n * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto&& f) { f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int i = 0; i < 5; i++) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 5; };
    int iters = 2000;
    float t0 = time([&] { hipLaunchKernelGGL(k_alu<0>, dim3(2048), dim3(256), 0, 0, o, iters, 12345ull); });
    float t2 = time([&] { hipLaunchKernelGGL(k_alu<2>, dim3(2048), dim3(256), 0, 0, o + n, iters, 12345ull); });
    float t3 = time([&] { hipLaunchKernelGGL(k_alu<3>, dim3(2048), dim3(256), 0, 0, o + 2 * n, iters, 12345ull); });
    float t4 = time([&] { hipLaunchKernelGGL(k_alu<4>, dim3(2048), dim3(256), 0, 0, o + 3 * n, iters, 12345ull); });
    float t5 = time([&] { hipLaunchKernelGGL(k_alu<5>, dim3(2048), dim3(256), 0, 0, o + 4 * n, iters, 12345ull); });
    double ops = 2048.0 * 256 * iters * 4;
    printf("merged-correction (8 VALU + 2 SALU) reduction %.3e /s\n", ops / (t5 * 1e-3));
    printf("mad-based asm reduction %.3e /s\n", ops / (t4 * 1e-3));
    printf("base mul nc: limb chain %.3e /s | 64-bit overflow builtins %.3e /s | asm reduction %.3e /s\n", ops / (t0 * 1e-3), ops / (t2 * 1e-3), ops / (t3 * 1e-3));
    uint64_t* h = (uint64_t*)malloc(5 * n * 8);
    CK(hipMemcpy(h, o, 5 * n * 8, hipMemcpyDeviceToHost));
    size_t bad2 = 0, bad3 = 0, bad4 = 0, bad5 = 0;
    for (size_t i = 0; i < n; i++) { bad2 += h[i] != h[n + i]; bad3 += h[i] != h[2 * n + i]; bad4 += h[i] != h[3 * n + i]; bad5 += h[i] != h[4 * n + i]; }
    printf("mismatches vs limb chain: builtins %zu, asm %zu, mad-asm %zu, merged %zu\n", bad2, bad3, bad4, bad5);
    return 0;
}
