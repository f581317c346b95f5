// How does the latency of "launch a small kernel, wait for its result on the host" scale with the number of host threads that do
// it concurrently, each on its own stream?  (Not part of the library; the question behind the lane scaling of DESIGN.md §4: eight
// chip-proof lanes are slower than four even without persistent kernels.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_lanes.hip -o tools/ubench_lanes -lpthread
// Modes:  sync = hipStreamSynchronize after every launch;  flag = the kernel writes a sequence number into pinned host memory and the
// host spins on it (no runtime call on the wait side);  chain = 16 launches back to back, then one flag wait.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_work(unsigned long long* flag, unsigned long long seq, int spin) {
    unsigned long long x = seq;
    for (int i = 0; i < spin; i++) x = x * 6364136223846793005ull + 1442695040888963407ull;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (x == 0x1234567) seq++;
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int main(int argc, char** argv) {
    const int iters = 2000;
    printf("{\n");
    const char* modes[] = {"sync", "flag", "chain16"};
    for (int mode = 0; mode < 3; mode++) {
        for (int blocks : {1, 64}) {
            printf(" \"%s_blocks%d_us_per_wait\": {", modes[mode], blocks);
            for (int T : {1, 2, 4, 8, 16}) {
                std::vector<hipStream_t> st(T);
                std::vector<unsigned long long*> flags(T);
                for (int t = 0; t < T; t++) {
                    CK(hipStreamCreateWithFlags(&st[t], hipStreamNonBlocking));
                    CK(hipHostMalloc((void**)&flags[t], 64, hipHostMallocMapped));
                    *flags[t] = 0;
                }
                std::atomic<int> ready{0};
                std::vector<double> us(T);
                auto worker = [&](int t) {
                    CK(hipSetDevice(0));
                    volatile unsigned long long* f = flags[t];
                    unsigned long long seq = 0;
                    auto once = [&] {
                        if (mode == 2) {
                            for (int j = 0; j < 16; j++) hipLaunchKernelGGL(k_work, dim3(blocks), dim3(256), 0, st[t], flags[t], ++seq, 200);
                        } else {
                            hipLaunchKernelGGL(k_work, dim3(blocks), dim3(256), 0, st[t], flags[t], ++seq, 200);
                        }
                        if (mode == 0) CK(hipStreamSynchronize(st[t]));
                        else
                            while (*f != seq) {}
                    };
                    for (int i = 0; i < 50; i++) once();
                    ready.fetch_add(1);
                    while (ready.load() < T) {}
                    const auto t0 = std::chrono::steady_clock::now();
                    for (int i = 0; i < iters; i++) once();
                    us[t] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
                };
                std::vector<std::thread> th;
                for (int t = 0; t < T; t++) th.emplace_back(worker, t);
                for (auto& x : th) x.join();
                double worst = 0;
                for (double u : us) worst = u > worst ? u : worst;
                printf("\"%d\": %.2f%s", T, worst, T == 16 ? "" : ", ");
                for (int t = 0; t < T; t++) {
                    CK(hipStreamDestroy(st[t]));
                    CK(hipHostFree(flags[t]));
                }
            }
            printf("}%s\n", (mode == 2 && blocks == 64) ? "" : ",");
        }
    }
    printf("}\n");
    return 0;
}
