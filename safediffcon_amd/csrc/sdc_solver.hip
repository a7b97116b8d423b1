// Evaluation rollout for the 1D Burgers' task: the explicit finite-difference solver the
// reference calls right after sampling to score a control sequence --
//   burgers_numeric_solve_free   1D/data/generate_burgers.py:207-299   (via control_trajectories, 1D/utils/metrics.py:42-65)
// 10 000 Euler steps of   u_i += dt * ( -1/2 * (u^2_{i+1} - u^2_{i-1}) / (2 dx) + visc * (u_{i-1} - 2 u_i + u_{i+1}) / dx^2 + f_i )
// with zero Dirichlet ghosts, the force switching every steps/Nt steps.  The reference spends ~8 torch launches per
// step; here one workgroup per trajectory keeps u in LDS (ping-pong, one barrier per step) for the whole rollout.
// Arithmetic is written with explicit round-to-nearest mul/add (no FMA contraction) in the reference's operation
// order, so results match the fp32 CPU run bit for bit.
#include "sdc_common.h"

namespace {

__global__ void burgers_rollout_kernel(const float* __restrict__ u0, const float* __restrict__ f, float* __restrict__ traj,
                                       int s, int Nt, int steps, int rec, float dt, float ct, float d0, float d1, float d2) {
    extern __shared__ float sh[];
    float* buf0 = sh;
    float* buf1 = sh + (s + 2);
    const int n = blockIdx.x;
    for (int i = threadIdx.x; i < s + 2; i += blockDim.x) { buf0[i] = 0.f; buf1[i] = 0.f; }
    __syncthreads();
    const float* fn = f + (int64_t)n * Nt * s;
    float* tn = traj + (int64_t)n * (Nt + 1) * s;
    for (int i = threadIdx.x; i < s; i += blockDim.x) {
        const float v = u0[(int64_t)n * s + i];
        buf0[i + 1] = v;
        tn[i] = v;
    }
    __syncthreads();
    const float nct = -ct;
    int fidx = -1, c = 0;
    for (int j = 0; j < steps; ++j) {
        if (j % rec == 0) ++fidx;
        const float* cur = (j & 1) ? buf1 : buf0;
        float* nxt = (j & 1) ? buf0 : buf1;
        const bool record = ((j + 1) % rec == 0) && c < Nt;
        for (int i = threadIdx.x; i < s; i += blockDim.x) {
            const float um = cur[i], uc = cur[i + 1], up = cur[i + 2];
            const float fv = fidx < Nt ? fn[(int64_t)fidx * s + i] : 0.f;
            // einsum('nsi,si->ns'): products summed in index order
            const float tr = __fadd_rn(__fmul_rn(__fmul_rn(um, um), nct), __fmul_rn(__fmul_rn(up, up), ct));
            const float df = __fadd_rn(__fadd_rn(__fmul_rn(um, d0), __fmul_rn(uc, d1)), __fmul_rn(up, d2));
            const float rhs = __fadd_rn(__fadd_rn(__fmul_rn(-0.5f, tr), df), fv);
            const float un = __fadd_rn(uc, __fmul_rn(dt, rhs));
            nxt[i + 1] = un;
            if (record) tn[(int64_t)(c + 1) * s + i] = un;
        }
        if (record) ++c;
        __syncthreads();
    }
}

}  // namespace

extern "C" int sdc_burgers_rollout(const float* u0, const float* f, float* traj, int N, int s, int Nt, int steps,
                                   int record_every, float dt, float coef_transport, float d0, float d1, float d2,
                                   void* stream) {
    SDC_REQUIRE(u0 && f && traj, SDC_ENULL, "sdc_burgers_rollout: null pointer");
    SDC_REQUIRE(N > 0 && s > 0 && Nt > 0 && steps > 0 && record_every > 0, SDC_EINVAL, "sdc_burgers_rollout: bad sizes");
    SDC_REQUIRE((size_t)(s + 2) * 2 * sizeof(float) <= 64 * 1024, SDC_EINVAL, "sdc_burgers_rollout: grid too large for LDS");
    int threads = ((s + 63) / 64) * 64;
    if (threads > 1024) threads = 1024;
    hipLaunchKernelGGL(burgers_rollout_kernel, dim3(N), dim3(threads), (size_t)(s + 2) * 2 * sizeof(float),
                       sdc::as_stream(stream), u0, f, traj, s, Nt, steps, record_every, dt, coef_transport, d0, d1, d2);
    return sdc::check_launch("sdc_burgers_rollout");
}
