// Does a latency-bound single-wave kernel run faster when the rest of the chip is busy (clock / power state)?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_clock.hip -o tools/ubench_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../ceno_amd/csrc/gl64.hpp"
using namespace gl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void k_chain(uint64_t* out, int iters, uint64_t seed) {
    uint64_t a = seed + threadIdx.x, b = seed * 7 + 3;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < iters; i++) a = mul_nc(a, b);  // one dependent chain
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = a; out[1] = t1 - t0; }
}
__global__ void __launch_bounds__(256) k_busy(uint64_t* buf, size_t n, int reps) {
    uint64_t acc = 0;
    for (int r = 0; r < reps; r++)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += mul_nc(buf[i], acc | 1);
    if (acc == 0x1234567) buf[0] = acc;
}
int main() {
    uint64_t *o, *buf; size_t n = (size_t)1 << 26;
    CK(hipMalloc(&o, 64)); CK(hipMalloc(&buf, n * 8)); CK(hipMemset(buf, 1, n * 8));
    hipStream_t s1, s2; int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, lo));
    uint64_t h[2]; const int iters = 20000;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, s1, o, iters, 5ull); CK(hipStreamSynchronize(s1));
        CK(hipMemcpy(h, o, 16, hipMemcpyDeviceToHost));
        printf("idle chip : %.1f ns per dependent multiplication\n", h[1] * 10.0 / iters);
    }
    hipLaunchKernelGGL(k_busy, dim3(2040), dim3(256), 0, s2, buf, n, 40);  // leaves a few CUs' worth of slots
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, s1, o, iters, 5ull); CK(hipStreamSynchronize(s1));
        CK(hipMemcpy(h, o, 16, hipMemcpyDeviceToHost));
        printf("busy chip : %.1f ns per dependent multiplication\n", h[1] * 10.0 / iters);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
