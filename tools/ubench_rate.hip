// Issue rate of the integer VALU instructions the field arithmetic is made of (wave64, 8 waves per SIMD, independent chains).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_rate.hip -o tools/ubench_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define REP8(x) x x x x x x x x
template <int OP>
__global__ void __launch_bounds__(256) k(uint64_t* out, int iters, uint64_t seed) {
    uint64_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x, c = a ^ 0x55, d = b ^ 0x77, e = a + 9, f = b + 11, g = a * 5, h = b * 7;
    uint32_t x = (uint32_t)a | 1, y = (uint32_t)b | 3;
    for (int i = 0; i < iters; i++) {
        if (OP == 0) {  // v_add_u32 (reference: full rate)
            asm volatile(REP8("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n")
                         : "+v"(((uint32_t*)&a)[0]), "+v"(((uint32_t*)&b)[0]), "+v"(((uint32_t*)&c)[0]), "+v"(((uint32_t*)&d)[0]), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x));
        } else if (OP == 1) {  // v_lshl_add_u64
            asm volatile(REP8("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %5\n v_lshl_add_u64 %2, %2, 0, %6\n v_lshl_add_u64 %3, %3, 0, %7\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));
        } else if (OP == 2) {  // v_mad_u64_u32
            asm volatile(REP8("v_mad_u64_u32 %0, s[20:21], %4, %5, %0\n v_mad_u64_u32 %1, s[20:21], %4, %5, %1\n v_mad_u64_u32 %2, s[20:21], %4, %5, %2\n v_mad_u64_u32 %3, s[20:21], %4, %5, %3\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y) : "s20", "s21");
        } else if (OP == 3) {  // v_cmp_lt_u64 (writes an SGPR pair)
            asm volatile(REP8("v_cmp_lt_u64 s[20:21], %0, %1\n v_cmp_lt_u64 s[22:23], %1, %2\n v_cmp_lt_u64 s[24:25], %2, %3\n v_cmp_lt_u64 s[26:27], %3, %0\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        } else if (OP == 4) {  // v_add_co_u32 (VCC-free form: SGPR pair carry out)
            asm volatile(REP8("v_add_co_u32 %0, s[20:21], %0, %4\n v_add_co_u32 %1, s[22:23], %1, %4\n v_add_co_u32 %2, s[24:25], %2, %4\n v_add_co_u32 %3, s[26:27], %3, %4\n")
                         : "+v"(((uint32_t*)&a)[0]), "+v"(((uint32_t*)&b)[0]), "+v"(((uint32_t*)&c)[0]), "+v"(((uint32_t*)&d)[0]) : "v"(x) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        } else if (OP == 5) {  // v_cndmask_b32 with an SGPR-pair mask
            asm volatile(REP8("v_cndmask_b32 %0, %0, %4, s[20:21]\n v_cndmask_b32 %1, %1, %4, s[20:21]\n v_cndmask_b32 %2, %2, %4, s[20:21]\n v_cndmask_b32 %3, %3, %4, s[20:21]\n")
                         : "+v"(((uint32_t*)&a)[0]), "+v"(((uint32_t*)&b)[0]), "+v"(((uint32_t*)&c)[0]), "+v"(((uint32_t*)&d)[0]) : "v"(x) : "s20", "s21");
        } else if (OP == 6) {  // v_mul_lo_u32
            asm volatile(REP8("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n")
                         : "+v"(((uint32_t*)&a)[0]), "+v"(((uint32_t*)&b)[0]), "+v"(((uint32_t*)&c)[0]), "+v"(((uint32_t*)&d)[0]) : "v"(x));
        } else if (OP == 8) {  // v_add_co_u32 e32 (VCC carry out)
            asm volatile(REP8("v_add_co_u32_e32 %0, vcc, %0, %4\n v_add_co_u32_e32 %1, vcc, %1, %4\n v_add_co_u32_e32 %2, vcc, %2, %4\n v_add_co_u32_e32 %3, vcc, %3, %4\n")
                         : "+v"(((uint32_t*)&a)[0]), "+v"(((uint32_t*)&b)[0]), "+v"(((uint32_t*)&c)[0]), "+v"(((uint32_t*)&d)[0]) : "v"(x) : "vcc");
        } else if (OP == 9) {  // v_addc_co_u32 e32 (VCC in and out; VCC written by the previous one: hazard handled by interleaving 4 chains?)
            asm volatile(REP8("v_addc_co_u32_e32 %0, vcc, %0, %4, vcc\n v_addc_co_u32_e32 %1, vcc, %1, %4, vcc\n v_addc_co_u32_e32 %2, vcc, %2, %4, vcc\n v_addc_co_u32_e32 %3, vcc, %3, %4, vcc\n")
                         : "+v"(((uint32_t*)&a)[0]), "+v"(((uint32_t*)&b)[0]), "+v"(((uint32_t*)&c)[0]), "+v"(((uint32_t*)&d)[0]) : "v"(x) : "vcc");
        } else if (OP == 10) {  // v_cndmask_b32 e32 (VCC mask)
            asm volatile(REP8("v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cndmask_b32_e32 %1, %1, %4, vcc\n v_cndmask_b32_e32 %2, %2, %4, vcc\n v_cndmask_b32_e32 %3, %3, %4, vcc\n")
                         : "+v"(((uint32_t*)&a)[0]), "+v"(((uint32_t*)&b)[0]), "+v"(((uint32_t*)&c)[0]), "+v"(((uint32_t*)&d)[0]) : "v"(x) : "vcc");
        } else if (OP == 11) {  // v_cmp_lt_u32 e32 (VCC out)
            asm volatile(REP8("v_cmp_lt_u32_e32 vcc, %0, %1\n v_cmp_lt_u32_e32 vcc, %1, %2\n v_cmp_lt_u32_e32 vcc, %2, %3\n v_cmp_lt_u32_e32 vcc, %3, %0\n")
                         : "+v"(((uint32_t*)&a)[0]), "+v"(((uint32_t*)&b)[0]), "+v"(((uint32_t*)&c)[0]), "+v"(((uint32_t*)&d)[0]) : : "vcc");
        } else if (OP == 12) {  // v_cmp_lt_u64 e32 (VCC out)
            asm volatile(REP8("v_cmp_lt_u64_e32 vcc, %0, %1\n v_cmp_lt_u64_e32 vcc, %1, %2\n v_cmp_lt_u64_e32 vcc, %2, %3\n v_cmp_lt_u64_e32 vcc, %3, %0\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");
        } else if (OP == 13) {  // v_add3_u32 (VOP3, no carry)
            asm volatile(REP8("v_add3_u32 %0, %0, %4, %1\n v_add3_u32 %1, %1, %4, %2\n v_add3_u32 %2, %2, %4, %3\n v_add3_u32 %3, %3, %4, %0\n")
                         : "+v"(((uint32_t*)&a)[0]), "+v"(((uint32_t*)&b)[0]), "+v"(((uint32_t*)&c)[0]), "+v"(((uint32_t*)&d)[0]) : "v"(x));
        } else if (OP == 14) {  // v_mul_hi_u32
            asm volatile(REP8("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n")
                         : "+v"(((uint32_t*)&a)[0]), "+v"(((uint32_t*)&b)[0]), "+v"(((uint32_t*)&c)[0]), "+v"(((uint32_t*)&d)[0]) : "v"(x));
        } else if (OP == 15) {  // v_mad_u32_u24
            asm volatile(REP8("v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0\n")
                         : "+v"(((uint32_t*)&a)[0]), "+v"(((uint32_t*)&b)[0]), "+v"(((uint32_t*)&c)[0]), "+v"(((uint32_t*)&d)[0]) : "v"(x));
        } else if (OP == 7) {  // v_mov_b32
            asm volatile(REP8("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4\n")
                         : "+v"(((uint32_t*)&a)[0]), "+v"(((uint32_t*)&b)[0]), "+v"(((uint32_t*)&c)[0]), "+v"(((uint32_t*)&d)[0]) : "v"(x));
        }
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}
int main() {
    uint64_t* o; CK(hipMalloc(&o, 2048 * 256 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto&& f) { f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int i = 0; i < 3; i++) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 3; };
    int iters = 4000;
    const char* names[] = {"v_add_u32", "v_lshl_add_u64", "v_mad_u64_u32", "v_cmp_lt_u64", "v_add_co_u32 (sgpr carry)", "v_cndmask_b32 (sgpr mask)", "v_mul_lo_u32", "v_mov_b32", "v_add_co_u32_e32 (vcc)", "v_addc_co_u32_e32 (vcc)", "v_cndmask_b32_e32 (vcc)", "v_cmp_lt_u32_e32 (vcc)", "v_cmp_lt_u64_e32 (vcc)", "v_add3_u32", "v_mul_hi_u32", "v_mad_u32_u24"};
    float t[16];
    t[0] = time([&] { hipLaunchKernelGGL(k<0>, dim3(2048), dim3(256), 0, 0, o, iters, 3ull); });
    t[1] = time([&] { hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, o, iters, 3ull); });
    t[2] = time([&] { hipLaunchKernelGGL(k<2>, dim3(2048), dim3(256), 0, 0, o, iters, 3ull); });
    t[3] = time([&] { hipLaunchKernelGGL(k<3>, dim3(2048), dim3(256), 0, 0, o, iters, 3ull); });
    t[4] = time([&] { hipLaunchKernelGGL(k<4>, dim3(2048), dim3(256), 0, 0, o, iters, 3ull); });
    t[5] = time([&] { hipLaunchKernelGGL(k<5>, dim3(2048), dim3(256), 0, 0, o, iters, 3ull); });
    t[6] = time([&] { hipLaunchKernelGGL(k<6>, dim3(2048), dim3(256), 0, 0, o, iters, 3ull); });
    t[7] = time([&] { hipLaunchKernelGGL(k<7>, dim3(2048), dim3(256), 0, 0, o, iters, 3ull); });
    t[8] = time([&] { hipLaunchKernelGGL(k<8>, dim3(2048), dim3(256), 0, 0, o, iters, 3ull); });
    t[9] = time([&] { hipLaunchKernelGGL(k<9>, dim3(2048), dim3(256), 0, 0, o, iters, 3ull); });
    t[10] = time([&] { hipLaunchKernelGGL(k<10>, dim3(2048), dim3(256), 0, 0, o, iters, 3ull); });
    t[11] = time([&] { hipLaunchKernelGGL(k<11>, dim3(2048), dim3(256), 0, 0, o, iters, 3ull); });
    t[12] = time([&] { hipLaunchKernelGGL(k<12>, dim3(2048), dim3(256), 0, 0, o, iters, 3ull); });
    t[13] = time([&] { hipLaunchKernelGGL(k<13>, dim3(2048), dim3(256), 0, 0, o, iters, 3ull); });
    t[14] = time([&] { hipLaunchKernelGGL(k<14>, dim3(2048), dim3(256), 0, 0, o, iters, 3ull); });
    t[15] = time([&] { hipLaunchKernelGGL(k<15>, dim3(2048), dim3(256), 0, 0, o, iters, 3ull); });
    double insts = 2048.0 * 4 /*waves per block*/ * iters * 32;  // wave-instructions
    for (int i = 0; i < 16; i++)
        printf("%-28s %.3e wave-instr/s  = %.2f cycles per wave-instr per SIMD at 2.4 GHz (1024 SIMDs)\n", names[i], insts / (t[i] * 1e-3),
               1024.0 * 2.4e9 / (insts / (t[i] * 1e-3)));
    return 0;
}
