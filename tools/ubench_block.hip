// Which runtime calls wait for a kernel that is itself waiting for the host?  (Not part of the library.)  Thread A runs a kernel that
// spins on a host flag; thread B calls hipMalloc / hipFree / hipHostMalloc / hipHostFree / hipStreamCreate; the flag is set after
// 200 ms.  A call that returns only after ~200 ms synchronises with the device.  Then: does a LAUNCH on a third stream get through
// while B sits in hipFree?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_block.hip -o tools/ubench_block -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void k_wait(volatile unsigned long long* flag) { while (*flag == 0) __builtin_amdgcn_s_sleep(10); }
__global__ void k_nop(unsigned long long* p) { if (p) *p = 1; }
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t sa, sc;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    unsigned long long* flag;
    CK(hipHostMalloc((void**)&flag, 64, hipHostMallocMapped));
    void* pre = nullptr;
    CK(hipMalloc(&pre, 1 << 20));
    void* preh = nullptr;
    CK(hipHostMalloc(&preh, 1 << 20, 0));
    const char* names[] = {"hipMalloc(64MB)", "hipFree", "hipHostMalloc(1MB)", "hipHostFree", "hipStreamCreate", "hipMemcpyAsync H2D 4KB pageable + no sync", "hipMallocAsync/none"};
    printf("{\n");
    for (int test = 0; test < 6; test++) {
        *flag = 0;
        hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, sa, flag);
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        std::atomic<double> t_call{-1}, t_launch{-1};
        const double t0 = now_ms();
        std::thread B([&] {
            CK(hipSetDevice(0));
            const double a = now_ms();
            void* p = nullptr;
            hipStream_t s2;
            static char hostbuf[4096];
            switch (test) {
            case 0: CK(hipMalloc(&p, 64 << 20)); break;
            case 1: CK(hipFree(pre)); pre = nullptr; break;
            case 2: CK(hipHostMalloc(&p, 1 << 20, 0)); break;
            case 3: CK(hipHostFree(preh)); preh = nullptr; break;
            case 4: CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); break;
            case 5: { void* d; CK(hipMalloc(&d, 4096)); CK(hipMemcpyAsync(d, hostbuf, 4096, hipMemcpyHostToDevice, sc)); } break;
            }
            t_call = now_ms() - a;
        });
        std::thread C([&] {  // a launch on a third stream 50 ms into the wait
            CK(hipSetDevice(0));
            std::this_thread::sleep_for(std::chrono::milliseconds(50));
            const double a = now_ms();
            hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, sc, (unsigned long long*)nullptr);
            CK(hipStreamSynchronize(sc));
            t_launch = now_ms() - a;
        });
        std::this_thread::sleep_for(std::chrono::milliseconds(200));
        *flag = 1;
        B.join();
        C.join();
        CK(hipStreamSynchronize(sa));
        printf(" \"%s\": {\"call_ms\": %.2f, \"launch_plus_sync_on_another_stream_ms\": %.2f}%s\n", names[test], t_call.load(), t_launch.load(), test == 5 ? "" : ",");
        (void)t0;
    }
    printf("}\n");
    return 0;
}
