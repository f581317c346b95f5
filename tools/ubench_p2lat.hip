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
#include "../ceno_amd/csrc/reduce.hpp"
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
// One ROUND of a persistent tail kernel with the challenger INSIDE it (round-4 verdict item 6): a single resident workgroup of 256 threads
// (k_tail's shape) does, per round, (1) the block sum of the three evaluation points, (2) the duplex challenger on 8 lanes of wave 0 — absorb
// the 6 message words + 2 label words at rate 4 = two dependent permutations, squeeze the challenge, hand it over through LDS — and (3) the
// fold of nine LDS-resident tables of 128 pairs with it (a tower layer's tables).  Phase times by wall_clock64 (100 MHz), summed over the rounds.
__global__ void __launch_bounds__(256) k_tail_round_inkernel(uint64_t* out, int rounds, const p2::Params* pp) {
    __shared__ p2::Params sp;
    __shared__ gl::E2 smem[4 * 3];
    __shared__ gl::E2 tab[9][256];
    __shared__ uint64_t s_ch[2];
    for (int i = threadIdx.x; i < (int)(sizeof(p2::Params) / 8); i += 256) reinterpret_cast<uint64_t*>(&sp)[i] = reinterpret_cast<const uint64_t*>(pp)[i];
    for (int t = 0; t < 9; t++) tab[t][threadIdx.x] = gl::E2{(uint64_t)threadIdx.x * 7 + t, (uint64_t)t + 1};
    __syncthreads();
    const unsigned lane = threadIdx.x & 63;
    uint64_t state = lane & 7;  // lane i of every group of 8 holds state word i
    unsigned long long t_sum = 0, t_hash = 0, t_fold = 0;
    gl::E2 acc[3] = {gl::E2{threadIdx.x + 1ull, 2}, gl::E2{3, threadIdx.x + 5ull}, gl::E2{7, 9}};
    for (int r = 0; r < rounds; r++) {
        const unsigned long long a = wall_clock64();
        gl::E2 m[3] = {acc[0], acc[1], acc[2]};
        red::block_sum<3, 256>(m, smem);
        __syncthreads();
        const unsigned long long b = wall_clock64();
        if (threadIdx.x < 64) {
            // message words live in thread 0 after the block sum: broadcast to the 8 challenger lanes by readlane
            uint64_t w[8];
            for (int e = 0; e < 3; e++) {
                w[2 * e] = red::readlane64(m[e].c0, 0);
                w[2 * e + 1] = red::readlane64(m[e].c1, 0);
            }
            w[6] = 0x496e7465726e616cull;  // label words ("Internal round" packs into two field elements)
            w[7] = 0x20726f756e64ull;
            const unsigned i = lane & 7;
            if (i < 4) state = gl::add(state, w[i] % gl::P);
            state = p2::permute_lanes8(state, sp);
            if (i < 4) state = gl::add(state, w[4 + i] % gl::P);
            state = p2::permute_lanes8(state, sp);
            if (threadIdx.x < 2) s_ch[threadIdx.x] = state;
        }
        __syncthreads();
        const unsigned long long c = wall_clock64();
        const gl::E2 ch{s_ch[0], s_ch[1]};
        if (threadIdx.x < 128)
            for (int t = 0; t < 9; t++) {
                const gl::E2 lo = tab[t][2 * threadIdx.x], hi = tab[t][2 * threadIdx.x + 1];
                const gl::E2 v = lo + ch * (hi - lo);
                __syncthreads();
                tab[t][threadIdx.x] = v;
                tab[t][threadIdx.x + 128] = v + gl::E2{1, 0};
                acc[t % 3] = acc[t % 3] + v;
            }
        else
            for (int t = 0; t < 9; t++) __syncthreads();
        __syncthreads();
        const unsigned long long d = wall_clock64();
        t_sum += b - a;
        t_hash += c - b;
        t_fold += d - c;
    }
    if (threadIdx.x == 0) {
        out[0] = acc[0].c0 ^ state;
        out[1] = t_sum;
        out[2] = t_hash;
        out[3] = t_fold;
    }
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
    uint64_t q[4];
    double ph[3] = {1e30, 1e30, 1e30};
    const int rounds = 500;
    for (int rep = 0; rep < 4; rep++) {
        hipLaunchKernelGGL(k_tail_round_inkernel, dim3(1), dim3(256), 0, 0, o, rounds, d);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(q, o, 32, hipMemcpyDeviceToHost));
        for (int k = 0; k < 3; k++) ph[k] = q[1 + k] * 10.0 / rounds / 1e3 < ph[k] ? q[1 + k] * 10.0 / rounds / 1e3 : ph[k];
    }
    printf("{\"device_permutation_us_one_lane\": %.2f, \"device_permutation_us_eight_lanes\": %.2f, \"in_kernel_round_us\": {\"block_sum\": %.2f, "
           "\"duplex_challenger_two_permutations_8_lanes\": %.2f, \"fold_nine_tables_of_128_pairs\": %.2f, \"total\": %.2f}}\n",
           best1 / 1e3, best8 / 1e3, ph[0], ph[1], ph[2], ph[0] + ph[1] + ph[2]);
    return 0;
}
