// Micro-benchmarks used to choose access patterns and field-arithmetic variants (not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../ceno_amd/csrc/gl64.hpp"
using namespace gl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// read patterns: each thread consumes PER_T E2 elements per iteration
template <int MODE>
__global__ void __launch_bounds__(256) k_read(const E2* __restrict__ in, size_t n, uint64_t* out) {
    size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nthreads = (size_t)gridDim.x * 256;
    uint64_t acc = 0;
    if (MODE == 0) {  // 16 B per lane, lane-contiguous
        for (size_t i = tid; i < n; i += nthreads) { E2 v = in[i]; acc ^= v.c0 + v.c1; }
    } else if (MODE == 1) {  // 32 B per lane contiguous (2 loads)
        for (size_t i = tid; i < n / 2; i += nthreads) { E2 a = in[2 * i], b = in[2 * i + 1]; acc ^= a.c0 + a.c1 + b.c0 + b.c1; }
    } else if (MODE == 2) {  // 64 B per lane contiguous (4 loads)
        for (size_t i = tid; i < n / 4; i += nthreads) {
            E2 a = in[4 * i], b = in[4 * i + 1], c = in[4 * i + 2], d = in[4 * i + 3];
            acc ^= a.c0 + a.c1 + b.c0 + b.c1 + c.c0 + c.c1 + d.c0 + d.c1;
        }
    } else if (MODE == 3) {  // 4 lane-contiguous loads per iteration (wave reads 4 KB as 4 x 1 KB rows)
        size_t wave = tid >> 6, lane = tid & 63, nwaves = nthreads >> 6;
        for (size_t w = wave; w < n / 256; w += nwaves) {
            const E2* p = in + w * 256 + lane;
            E2 a = p[0], b = p[64], c = p[128], d = p[192];
            acc ^= a.c0 + a.c1 + b.c0 + b.c1 + c.c0 + c.c1 + d.c0 + d.c1;
        }
    }
    if (acc == 0x1234567) out[0] = acc;
}

// fold-like: read 64 B per lane, write 32 B per lane
__global__ void __launch_bounds__(256) k_fold_like(const E2* __restrict__ in, E2* __restrict__ outp, size_t pairs) {
    size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nthreads = (size_t)gridDim.x * 256;
    for (size_t p = tid; p < pairs; p += nthreads) {
        E2 a = in[4 * p], b = in[4 * p + 1], c = in[4 * p + 2], d = in[4 * p + 3];
        outp[2 * p] = E2{a.c0 ^ b.c0, a.c1 ^ b.c1};
        outp[2 * p + 1] = E2{c.c0 ^ d.c0, c.c1 ^ d.c1};
    }
}

// fold-like variants: V=0 plain, 1 nontemporal stores, 2 nontemporal loads+stores, 3 nontemporal loads only
template <int V>
__global__ void __launch_bounds__(256) k_fold_nt(const E2* __restrict__ in, E2* __restrict__ outp, size_t pairs) {
    size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nthreads = (size_t)gridDim.x * 256;
    typedef unsigned long long u2 __attribute__((ext_vector_type(2)));
    const u2* vin = reinterpret_cast<const u2*>(in);
    u2* vout = reinterpret_cast<u2*>(outp);
    for (size_t p = tid; p < pairs; p += nthreads) {
        u2 a, b, c, d;
        if (V == 2 || V == 3) { a = __builtin_nontemporal_load(vin + 4 * p); b = __builtin_nontemporal_load(vin + 4 * p + 1); c = __builtin_nontemporal_load(vin + 4 * p + 2); d = __builtin_nontemporal_load(vin + 4 * p + 3); }
        else { a = vin[4 * p]; b = vin[4 * p + 1]; c = vin[4 * p + 2]; d = vin[4 * p + 3]; }
        u2 x = a ^ b, y = c ^ d;
        if (V == 1 || V == 2) { __builtin_nontemporal_store(x, vout + 2 * p); __builtin_nontemporal_store(y, vout + 2 * p + 1); }
        else { vout[2 * p] = x; vout[2 * p + 1] = y; }
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) k_alu(E2* out, int iters, E2 seed) {
    E2 a = E2{seed.c0 + threadIdx.x, seed.c1 + blockIdx.x}, b = E2{seed.c1 ^ threadIdx.x, seed.c0};
    E2 c = a, d = b;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { a = a * b; b = b * a; c = c * d; d = d * c; }            // ext mults (Karatsuba)
        else if (MODE == 1) { a.c0 = mul(a.c0, b.c0); b.c0 = mul(b.c0, a.c0); c.c0 = mul(c.c0, d.c0); d.c0 = mul(d.c0, c.c0); }  // base mults
        else if (MODE == 2) { a = a + b; b = b - a; c = c + d; d = d - c; }          // ext adds
        else if (MODE == 3) { a = e2_mul_ref(a, b); b = e2_mul_ref(b, a); c = e2_mul_ref(c, d); d = e2_mul_ref(d, c); }  // old Karatsuba
        else if (MODE == 4) { a.c0 = mul_ref(a.c0, b.c0); b.c0 = mul_ref(b.c0, a.c0); c.c0 = mul_ref(c.c0, d.c0); d.c0 = mul_ref(d.c0, c.c0); }
        else if (MODE == 5) {  // raw 32x32+64 multiply-add chain
            a.c0 = (uint64_t)(uint32_t)a.c0 * (uint32_t)b.c0 + a.c1; b.c0 = (uint64_t)(uint32_t)b.c0 * (uint32_t)a.c0 + b.c1;
            c.c0 = (uint64_t)(uint32_t)c.c0 * (uint32_t)d.c0 + c.c1; d.c0 = (uint64_t)(uint32_t)d.c0 * (uint32_t)c.c0 + d.c1;
        }
    }
    if ((a.c0 ^ b.c0 ^ c.c0 ^ d.c0) == 0x1234567) out[0] = a + b + c + d;
}

int main() {
    size_t n = (size_t)1 << 27;  // 2 GB of E2
    E2 *in, *outp; uint64_t* o;
    CK(hipMalloc(&in, n * sizeof(E2))); CK(hipMalloc(&outp, n / 2 * sizeof(E2))); CK(hipMalloc(&o, 64));
    CK(hipMemset(in, 1, n * sizeof(E2)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto&& f) { f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int i = 0; i < 5; i++) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 5; };
    for (int blocks : {1024, 2048, 4096, 8192}) {
        float t0 = time([&] { hipLaunchKernelGGL(k_read<0>, dim3(blocks), dim3(256), 0, 0, in, n, o); });
        float t1 = time([&] { hipLaunchKernelGGL(k_read<1>, dim3(blocks), dim3(256), 0, 0, in, n, o); });
        float t2 = time([&] { hipLaunchKernelGGL(k_read<2>, dim3(blocks), dim3(256), 0, 0, in, n, o); });
        float t3 = time([&] { hipLaunchKernelGGL(k_read<3>, dim3(blocks), dim3(256), 0, 0, in, n, o); });
        float t4 = time([&] { hipLaunchKernelGGL(k_fold_like, dim3(blocks), dim3(256), 0, 0, in, outp, n / 4); });
        double gb = n * 16.0 / 1e9;
        printf("blocks=%5d  read16B/lane %.0f GB/s | 32B/lane %.0f | 64B/lane %.0f | 4x1KB rows %.0f | fold-like(r+w) %.0f GB/s\n", blocks,
               gb / t0 * 1e3, gb / t1 * 1e3, gb / t2 * 1e3, gb / t3 * 1e3, gb * 1.5 / t4 * 1e3);
    }
    for (int blocks : {1024, 1536, 2048}) {
        float f0 = time([&] { hipLaunchKernelGGL(k_fold_nt<0>, dim3(blocks), dim3(256), 0, 0, in, outp, n / 4); });
        float f1 = time([&] { hipLaunchKernelGGL(k_fold_nt<1>, dim3(blocks), dim3(256), 0, 0, in, outp, n / 4); });
        float f2 = time([&] { hipLaunchKernelGGL(k_fold_nt<2>, dim3(blocks), dim3(256), 0, 0, in, outp, n / 4); });
        float f3 = time([&] { hipLaunchKernelGGL(k_fold_nt<3>, dim3(blocks), dim3(256), 0, 0, in, outp, n / 4); });
        double gb = n * 16.0 * 1.5 / 1e9;
        printf("fold-like blocks=%d: plain %.0f | nt-store %.0f | nt-load+store %.0f | nt-load %.0f GB/s\n", blocks, gb / f0 * 1e3, gb / f1 * 1e3, gb / f2 * 1e3, gb / f3 * 1e3);
    }
    int iters = 2000;
    const char* names[] = {"ext mul (lazy schoolbook)", "base mul (limb)", "ext add/sub", "ext mul (old karatsuba)", "base mul (old)", "mad_u64_u32 chain"};
    for (int mode = 0; mode < 6; mode++) {
        float t = time([&] {
            if (mode == 0) hipLaunchKernelGGL(k_alu<0>, dim3(2048), dim3(256), 0, 0, outp, iters, E2{3, 5});
            if (mode == 1) hipLaunchKernelGGL(k_alu<1>, dim3(2048), dim3(256), 0, 0, outp, iters, E2{3, 5});
            if (mode == 2) hipLaunchKernelGGL(k_alu<2>, dim3(2048), dim3(256), 0, 0, outp, iters, E2{3, 5});
            if (mode == 3) hipLaunchKernelGGL(k_alu<3>, dim3(2048), dim3(256), 0, 0, outp, iters, E2{3, 5});
            if (mode == 4) hipLaunchKernelGGL(k_alu<4>, dim3(2048), dim3(256), 0, 0, outp, iters, E2{3, 5});
            if (mode == 5) hipLaunchKernelGGL(k_alu<5>, dim3(2048), dim3(256), 0, 0, outp, iters, E2{3, 5});
        });
        double ops = 2048.0 * 256 * iters * 4;
        printf("alu mode %d (%s): %.3e ops/s\n", mode, names[mode], ops / (t * 1e-3));
    }
    return 0;
}
