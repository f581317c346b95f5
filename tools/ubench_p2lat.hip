// Could the Fiat-Shamir challenger live on the device?  Latency of ONE Poseidon2 permutation (width 8) on gfx950 — the
// permutation a duplex challenger runs twice between a round message of degree 3 and its challenge (6 message words + 2 label
// words at rate 4) — as a dependent chain, one lane per permutation and eight lanes per permutation (the form the Merkle tree
// tops use), against the host permutation of the transcript (ceno_amd/host/transcript.cpp p2host, measured by
// tools/ubench_p2lat.py on the same box).  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/ubench_p2lat.hip -o tools/ubench_p2lat
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "../ceno_amd/csrc/gl64.hpp"
#include "../ceno_amd/csrc/poseidon2.hpp"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void __launch_bounds__(64) k_chain1(uint64_t* out, int iters, const p2::Params* pp) {
    __shared__ p2::Params sp;
    for (int i = threadIdx.x; i < (int)(sizeof(p2::Params) / 8); i += 64) reinterpret_cast<uint64_t*>(&sp)[i] = reinterpret_cast<const uint64_t*>(pp)[i];
    __syncthreads();
    uint64_t s[8];
    for (int k = 0; k < 8; k++) s[k] = threadIdx.x + k;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < iters; i++) p2::permute(s, sp);
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = s[0]; out[1] = t1 - t0; }
}
__global__ void __launch_bounds__(64) k_chain8(uint64_t* out, int iters, const p2::Params* pp) {
    __shared__ p2::Params sp;
    for (int i = threadIdx.x; i < (int)(sizeof(p2::Params) / 8); i += 64) reinterpret_cast<uint64_t*>(&sp)[i] = reinterpret_cast<const uint64_t*>(pp)[i];
    __syncthreads();
    uint64_t x = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < iters; i++) x = p2::permute_lanes8(x, sp);
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = x; out[1] = t1 - t0; }
}
int main() {
    uint64_t* o;
    p2::Params h, *d;
    p2::default_params(h);
    CK(hipMalloc(&o, 64));
    CK(hipMalloc(&d, sizeof(h)));
    CK(hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice));
    uint64_t r[2];
    const int iters = 2000;
    double best1 = 1e30, best8 = 1e30;
    for (int rep = 0; rep < 4; rep++) {
        hipLaunchKernelGGL(k_chain1, dim3(1), dim3(64), 0, 0, o, iters, d);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(r, o, 16, hipMemcpyDeviceToHost));
        if (r[1] * 10.0 / iters < best1) best1 = r[1] * 10.0 / iters;  // wall_clock64 ticks at 100 MHz
        hipLaunchKernelGGL(k_chain8, dim3(1), dim3(64), 0, 0, o, iters, d);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(r, o, 16, hipMemcpyDeviceToHost));
        if (r[1] * 10.0 / iters < best8) best8 = r[1] * 10.0 / iters;
    }
    printf("{\"device_permutation_us_one_lane\": %.2f, \"device_permutation_us_eight_lanes\": %.2f}\n", best1 / 1e3, best8 / 1e3);
    return 0;
}
