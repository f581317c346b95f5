// Fused large rounds of a tower layer's sumcheck with the eq factor taken out of the evaluation points (sumcheck_tower.hpp).
#include "sumcheck_tower.hpp"

using namespace gl;

namespace {

// MODE 0: round 0, three values (q(1), leading coefficient, q(0): no claim is known)   MODE 1: round 0 under a known claim, two values
// MODE 2: fold with the previous challenge, write, two values
template <int NP, int NL, int MODE>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(MODE == 0 ? 3 : 4, 8))) k_tower(const MleSlot* __restrict__ slots, TowerCoef coef, size_t pairs, Epilogue ep) {
    constexpr bool R0 = MODE < 2;
    constexpr int D = MODE == 0 ? 3 : 2;
    constexpr int K = 1 + 2 * NP + 4 * NL;
    __shared__ E2 smem[(NT / 64) * D];
    __shared__ unsigned long long s_chal[3];
    __shared__ int s_flag;
    E2 r = e2_zero();
    if (ep.wait_seq != 0) {
        if (!read_challenge(ep, r, s_chal)) return;  // pipeline aborted / timed out: leave everything untouched
    }
    const E2Pre rp = e2_pre(r);
    const E2* in[K];
    E2* out[K];
#pragma unroll
    for (int m = 0; m < K; m++) {
        in[m] = reinterpret_cast<const E2*>(slots[m].in);
        out[m] = reinterpret_cast<E2*>(slots[m].out);
    }
    E2 acc[D];  // reduced: three unreduced 160-bit totals next to the three of a pair do not fit 128 registers
#pragma unroll
    for (int t = 0; t < D; t++) acc[t] = e2_zero();
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t p = (size_t)blockIdx.x * NT + threadIdx.x; p < pairs; p += stride) {
        auto load = [&](int m, E2& lo, E2& hi) {
            if (R0) {
                const E2* q = in[m] + 2 * p;
                lo = q[0];
                hi = q[1];
            } else {
                const E2* q = in[m] + 4 * p;
                const E2 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3];
                lo = a0 + e2_mul_pre(rp, a1 - a0);
                hi = a2 + e2_mul_pre(rp, a3 - a2);
                E2* o = out[m] + 2 * p;
                o[0] = lo;
                o[1] = hi;
            }
        };
        E2 elo, ehi;
        load(0, elo, ehi);
        const E2 w = elo + ehi;  // P_i * E_i[y]: the eq table's pair is (1 - rt_i, rt_i) times it
        E2Acc g[D];
#pragma unroll
        for (int t = 0; t < D; t++) g[t] = e2acc_zero();
#pragma unroll
        for (int i = 0; i < NP; i++) {
            E2 alo, ahi, blo, bhi;
            load(1 + 2 * i, alo, ahi);
            load(2 + 2 * i, blo, bhi);
            e2acc_mac(g[0], coef.prod[i], ahi * bhi);
            e2acc_mac(g[1], coef.prod[i], (ahi - alo) * (bhi - blo));
            if (D == 3) e2acc_mac(g[2], coef.prod[i], alo * blo);
        }
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int b = 1 + 2 * NP + 4 * k;
            E2 p1l, p1h, p2l, p2h, q1l, q1h, q2l, q2h;
            load(b, p1l, p1h);
            load(b + 1, p2l, p2h);
            load(b + 2, q1l, q1h);
            load(b + 3, q2l, q2h);
            auto point = [&](E2Acc& acc, E2 p1, E2 p2, E2 q1, E2 q2) {
                E2Acc x = e2acc_zero();  // p1 q2 + p2 q1: one reduction for the two products
                e2acc_mac(x, p1, q2);
                e2acc_mac(x, p2, q1);
                e2acc_mac(acc, coef.logup[k][0], e2acc_reduce(x));
                e2acc_mac(acc, coef.logup[k][1], q1 * q2);
            };
            point(g[0], p1h, p2h, q1h, q2h);
            point(g[1], p1h - p1l, p2h - p2l, q1h - q1l, q2h - q2l);
            if (D == 3) point(g[2], p1l, p2l, q1l, q2l);
        }
#pragma unroll
        for (int t = 0; t < D; t++) acc[t] = acc[t] + w * e2acc_reduce(g[t]);
    }
    epilogue<D, NT>(acc, ep, smem, &s_flag);
}

// (grids stay within the workgroups that are resident at once: round 0 without a claim holds 3 per CU at 164 VGPRs, see resident_grid)
template <int NP, int NL>
void launch_npnl(ceno_hip_ctx* ctx, int mode, const MleSlot* slots, const TowerCoef& coef, size_t pairs, const Epilogue& ep, unsigned grid, hipStream_t st) {
    if (mode == 0) hipLaunchKernelGGL((k_tower<NP, NL, 0>), dim3(resident_grid(ctx, k_tower<NP, NL, 0>, NT, 0, grid)), dim3(NT), 0, st, slots, coef, pairs, ep);
    else if (mode == 1) hipLaunchKernelGGL((k_tower<NP, NL, 1>), dim3(resident_grid(ctx, k_tower<NP, NL, 1>, NT, 0, grid)), dim3(NT), 0, st, slots, coef, pairs, ep);
    else hipLaunchKernelGGL((k_tower<NP, NL, 2>), dim3(resident_grid(ctx, k_tower<NP, NL, 2>, NT, 0, grid)), dim3(NT), 0, st, slots, coef, pairs, ep);
}

}  // namespace

bool tower_fast_shape(int n_prod, int n_logup) {
    return n_prod >= 0 && n_logup >= 0 && n_prod <= TOWER_FAST_MAX_PROD && n_logup <= TOWER_FAST_MAX_LOGUP && n_prod + n_logup >= 1;
}

void launch_tower_round(ceno_hip_ctx* ctx, int n_prod, int n_logup, int mode, const MleSlot* slots, const TowerCoef& coef, size_t pairs, const Epilogue& ep,
                        unsigned grid, hipStream_t st) {
#define CASE(P, L) \
    case (P) * 4 + (L): launch_npnl<P, L>(ctx, mode, slots, coef, pairs, ep, grid, st); break;
    switch (n_prod * 4 + n_logup) {
        CASE(1, 0) CASE(2, 0) CASE(3, 0)
        CASE(0, 1) CASE(1, 1) CASE(2, 1) CASE(3, 1)
        CASE(0, 2) CASE(1, 2) CASE(2, 2) CASE(3, 2)
    default: break;  // tower_fast_shape() said no: the caller never gets here
    }
#undef CASE
}
