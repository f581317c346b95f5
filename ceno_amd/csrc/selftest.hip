// Device-side self-test hooks of the field arithmetic (test infrastructure of libceno_hip.so: tests/test_gpu_field.py feeds
// chosen limb patterns — both conditional corrections of the reduction, their cancellation, every carry path of the wide
// product — and compares with Python integers).  Not on any product path.
#include "common.hpp"

using namespace gl;

// in: n records of 5 limbs (w0..w3 32-bit, c < 16) ; out: 2 words per record = canon(reduce128_nc(w0..w3)), reduce_limbs(w0..w3,c)
__global__ void k_selftest_reduce(const uint32_t* __restrict__ in, size_t n, uint64_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* w = in + 5 * i;
    const uint64_t r = canon(reduce128_nc(w[0], w[1], w[2], w[3])), rm = canon(reduce128_ncm(w[0], w[1], w[2], w[3]));
    out[2 * i] = r == rm ? r : ~0ull;  // both forms of the 128-bit reduction must agree
    out[2 * i + 1] = reduce_limbs(w[0], w[1], w[2], w[3], w[4]);
}
// in: n pairs (a, b) of ARBITRARY 64-bit words; out: 6 words per pair = mul(a', b'), canon(mul_nc(a, b)), add(a', b'), sub(a', b'),
// canon(add_nc(a, b')), mul_add(a, b', a')  with a' = canon(a), b' = canon(b)
__global__ void k_selftest_ops(const uint64_t* __restrict__ in, size_t n, uint64_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t a = in[2 * i], b = in[2 * i + 1], ac = canon(a), bc = canon(b);
    uint64_t* o = out + 6 * i;
    o[0] = mul(ac, bc);
    o[1] = canon(mul_nc(a, b));
    o[2] = add(ac, bc);
    o[3] = sub(ac, bc);
    o[4] = canon(add_nc(a, bc));
    o[5] = mul_add(a, bc, ac);
}
// in: n pairs of ext elements (4 canonical words); out: 2 ext per pair = a * b, e2acc-reduced (a * b + b * a + a * a)
__global__ void k_selftest_ext(const uint64_t* __restrict__ in, size_t n, uint64_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const E2 a{in[4 * i], in[4 * i + 1]}, b{in[4 * i + 2], in[4 * i + 3]};
    const E2 p = a * b;
    E2Acc acc = e2acc_zero();
    e2acc_mac(acc, a, b);
    e2acc_mac(acc, b, a);
    e2acc_mac(acc, a, a);
    const E2 q = e2acc_reduce(acc);
    out[4 * i] = p.c0; out[4 * i + 1] = p.c1; out[4 * i + 2] = q.c0; out[4 * i + 3] = q.c1;
}

extern "C" int ceno_hip_selftest_field(ceno_hip_ctx* ctx, int which, const void* host_in, size_t n, uint64_t* host_out) {
    CHECK_ARG(ctx, host_in && host_out && n > 0 && which >= 0 && which <= 2, "bad selftest arguments");
    hipStream_t st = ctx_stream(ctx, nullptr);
    const size_t in_bytes = which == 0 ? n * 5 * 4 : which == 1 ? n * 16 : n * 32;
    const size_t out_words = which == 0 ? 2 * n : which == 1 ? 6 * n : 4 * n;
    void *din = nullptr, *dout = nullptr;
    TRY(ctx_alloc(ctx, in_bytes, &din));
    if (int rc = ctx_alloc(ctx, out_words * 8, &dout)) { ctx_free(ctx, din); return rc; }
    hipError_t e = hipMemcpyAsync(din, host_in, in_bytes, hipMemcpyHostToDevice, st);
    const unsigned g = (unsigned)((n + 255) / 256);
    if (e == hipSuccess) {
        if (which == 0) k_selftest_reduce<<<g, 256, 0, st>>>((const uint32_t*)din, n, (uint64_t*)dout);
        if (which == 1) k_selftest_ops<<<g, 256, 0, st>>>((const uint64_t*)din, n, (uint64_t*)dout);
        if (which == 2) k_selftest_ext<<<g, 256, 0, st>>>((const uint64_t*)din, n, (uint64_t*)dout);
        e = hipMemcpyAsync(host_out, dout, out_words * 8, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    ctx_free(ctx, din);
    ctx_free(ctx, dout);
    if (e != hipSuccess) return ctx_fail(ctx, CENO_HIP_ERR_HIP, "selftest: %s", hipGetErrorString(e));
    return 0;
}
