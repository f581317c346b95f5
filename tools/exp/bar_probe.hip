// Experiment: is fine-grained DEVICE memory directly writable by the host on this box (large BAR)?  If so the
// challenge mailbox of the pipelined sumcheck could live in HBM (the device polls locally, the host posts one PCIe write).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
__global__ void k_read(const volatile uint64_t* p, uint64_t* out) { *out = *p; }
__global__ void k_poll(const volatile uint64_t* p, uint64_t want, uint64_t* out) {
    uint64_t n = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != want && n < (1ull << 26)) n++;
    *out = n;
}
int main() {
    uint64_t* p = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&p, 4096, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags: %s ptr=%p\n", hipGetErrorString(e), (void*)p);
    if (e != hipSuccess) return 1;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) == hipSuccess) printf("type=%d device=%d hostPointer=%p devicePointer=%p\n", (int)at.type, at.device, at.hostPointer, at.devicePointer);
    uint64_t* out = nullptr;
    hipHostMalloc((void**)&out, 64);
    hipMemset(p, 0, 4096);
    hipDeviceSynchronize();
    printf("host write...\n");
    fflush(stdout);
    *(volatile uint64_t*)p = 42;  // SIGSEGV here if VRAM is not host mapped
    printf("host write done\n");
    k_read<<<1, 1>>>(p, out);
    hipDeviceSynchronize();
    printf("device read back %llu\n", (unsigned long long)*out);
    // latency: device polls, host writes
    for (int rep = 0; rep < 3; rep++) {
        *(volatile uint64_t*)p = 0;
        k_poll<<<1, 1>>>(p, 1000 + rep, out);
        auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 200) {}
        auto t1 = std::chrono::steady_clock::now();
        *(volatile uint64_t*)p = 1000 + rep;
        hipDeviceSynchronize();
        auto t2 = std::chrono::steady_clock::now();
        printf("poll iterations %llu, host write -> kernel exit + sync %.1f us\n", (unsigned long long)*out,
               std::chrono::duration<double, std::micro>(t2 - t1).count());
    }
    return 0;
}
