// Tokamak score check: the KSTAR surrogate rollout the reference runs on every sampled control
// sequence --
//   KSTARSolver.simulate / predict_0d / control     tokamak/kstar_solver.py:163-428
//   kstar_v220505 / kstar_nn / bpw_nn               tokamak/common/model_structure.py:69-152   (Keras LSTM(100,100) + Dense nets)
//   control_trajectories                            tokamak/utils/metrics.py:60-85             (serial over samples there)
// The reference makes 122 single-sample Keras predict() calls per trajectory, one trajectory after the other.  Here one
// workgroup carries NS trajectories through all 122 rows: thread t < 400 owns gate column t of the LSTM kernels (held in
// registers for the ten window rows of a call, serving the NS samples), the 10-row input window, the hidden / cell states and the small dense nets live in
// LDS, and the integer quantisation of the actions (f2i / i2f) and the output de-normalisation run in fp64 like the Python.
//
// Per step (kstar_solver.py:236-350):  control: inputs <- f2i(clip(action));  window rows shift (inputs part), new last row;
// y = mean_models(LSTM(window) * ystd + ymean);  window outputs part shifts, last row <- y;  (bp, wmhd) = bpw net on
// [bn, Ip, Bt, (In+Out)/2, (Out-In)/2, Elon, UpTri, LoTri];  H factors;  row = [bn, bp, h89, h98, q95, q0, li, wmhd].
// Networks run in fp32 (Keras predict casts its input to float32), BatchNormalization folded to x * inv + off by the host.
#include "sdc_common.h"
#include "../../include/sdc.h"

namespace {

constexpr int KT = 512;                  // threads per workgroup
constexpr int T = SDC_KSTAR_SEQ, NIN = SDC_KSTAR_NIN, U = SDC_KSTAR_UNITS, G = 4 * SDC_KSTAR_UNITS;
constexpr int MAXW = 256;                // widest dense layer

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// floats of one LSTM network's parameters, in the order the kernel walks them
constexpr int64_t LSTM_FLOATS = 2 * NIN + (int64_t)NIN * G + (int64_t)U * G + G + 2 * U + (int64_t)U * G + (int64_t)U * G + G;

template <int NS>
struct Lds {
    float win[NS][T][NIN];      // the raw float32 window (what Keras sees after its cast)
    float xb[NS][T][NIN];       // after the first BatchNormalization
    float seq[NS][T][U];        // first LSTM's outputs after the second BatchNormalization
    float h[NS][U], c[NS][U];
    float z[NS][G];
    float va[NS][MAXW], vb[NS][MAXW];
    double inp[NS][15];
    double y[NS][4];            // [bn, q95, q0, li]
    double yacc[NS][4];
    double bpw[NS][2];
};

// one BatchNormalization -> Dense stack on the NS vectors in s.va; the layers ping-pong between s.va and s.vb and the buffer
// holding the result is returned
template <int NS>
__device__ const float (*mlp(const SdcKstarMlp& d, int net, Lds<NS>& s))[MAXW] {
    const float* p = d.params + (int64_t)net * d.stride;
    float (*cur)[MAXW] = s.va;
    for (int l = 0; l < d.nlayers; ++l) {
        const int ni = d.width[l], no = d.width[l + 1];
        const float* inv = p; const float* off = p + ni; const float* K = p + 2 * ni; const float* b = K + (int64_t)ni * no;
        float (*dst)[MAXW] = (cur == s.va) ? s.vb : s.va;
        for (int e = threadIdx.x; e < NS * no; e += KT) {
            const int sm = e / no, j = e - sm * no;
            float acc = b[j];
            for (int k = 0; k < ni; ++k) acc += (cur[sm][k] * inv[k] + off[k]) * K[(int64_t)k * no + j];
            dst[sm][j] = d.act[l] ? sigmoidf(acc) : acc;
        }
        __syncthreads();
        cur = dst;
        p = b + no;
    }
    return cur;
}

// one LSTM layer over the T window rows.  FIRST: input = s.xb (NIN wide), output sequence -> s.seq through the next
// BatchNormalization; otherwise input = s.seq (U wide) and only the final state is kept.  Thread col < 400 holds its column of
// the recurrent kernel (and of the input kernel) in registers for the ten rows: the weights are read once per layer call.
template <int NS, bool FIRST>
__device__ void lstm_layer(Lds<NS>& s, const float* __restrict__ K, const float* __restrict__ R, const float* __restrict__ b,
                           const float* __restrict__ inv_next, const float* __restrict__ off_next) {
    constexpr int NI = FIRST ? NIN : U;
    constexpr bool KREG = FIRST || NS == 1;              // the input kernel's column in registers too (register budget: 256)
    const int col = threadIdx.x < G ? threadIdx.x : G - 1;
    float wk[KREG ? NI : 1], wr[U];
    if (KREG) {
#pragma unroll
        for (int k = 0; k < NI; ++k) wk[k] = K[k * G + col];
    }
#pragma unroll
    for (int k = 0; k < U; ++k) wr[k] = R[k * G + col];
    const float bias = b[col];
    for (int e = threadIdx.x; e < NS * U; e += KT) { (&s.h[0][0])[e] = 0.f; (&s.c[0][0])[e] = 0.f; }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        // even / odd input index into two partial sums, the same order whichever way the input kernel is held
        float acc0[NS], acc1[NS];
#pragma unroll
        for (int sm = 0; sm < NS; ++sm) { acc0[sm] = bias; acc1[sm] = 0.f; }
        if (!KREG) {
#pragma unroll 5
            for (int k = 0; k < NI; k += 2) {
                const float w0 = K[k * G + col], w1 = K[(k + 1) * G + col];
#pragma unroll
                for (int sm = 0; sm < NS; ++sm) { acc0[sm] += s.seq[sm][t][k] * w0; acc1[sm] += s.seq[sm][t][k + 1] * w1; }
            }
        }
        float acc[NS];
#pragma unroll
        for (int sm = 0; sm < NS; ++sm) {
            float a0 = acc0[sm], a1 = acc1[sm];
            if (KREG) {
                const float* x = FIRST ? &s.xb[sm][t][0] : &s.seq[sm][t][0];
#pragma unroll
                for (int k = 0; k + 1 < NI; k += 2) { a0 += x[k] * wk[k]; a1 += x[k + 1] * wk[k + 1]; }
            }
            const float* hh = &s.h[sm][0];
#pragma unroll
            for (int k = 0; k < U; k += 2) { a0 += hh[k] * wr[k]; a1 += hh[k + 1] * wr[k + 1]; }
            acc[sm] = a0 + a1;
        }
        if (threadIdx.x < G) {
#pragma unroll
            for (int sm = 0; sm < NS; ++sm) s.z[sm][col] = acc[sm];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < NS * U; e += KT) {
            const int sm = e / U, u = e - sm * U;
            const float zi = s.z[sm][u], zf = s.z[sm][U + u], zc = s.z[sm][2 * U + u], zo = s.z[sm][3 * U + u];
            const float cn = sigmoidf(zf) * s.c[sm][u] + sigmoidf(zi) * tanhf(zc);
            const float hn = sigmoidf(zo) * tanhf(cn);
            s.c[sm][u] = cn;
            s.h[sm][u] = hn;
            if (FIRST) s.seq[sm][t][u] = hn * inv_next[u] + off_next[u];
        }
        __syncthreads();
    }
}

// rows 4..16 of the window from the current inputs (kstar_solver.py:218-233, :241-258), as Keras' float32 cast sees them
__device__ __forceinline__ float window_value(const double* inp, int j) {
    //                   Ip Bt GW Elon UpTri LoTri InMid OutMid Pnb1a Pnb1b Pnb1c Pec2 InMid
    const int src[13] = {0, 1, 2, 12, 13, 14, 10, 11, 3, 4, 5, 6, 10};
    double v = inp[src[j]];
    if (j == 11) v += inp[7];
    if (j == 12) v = v > 1.265 + 1.e-4 ? 1.0 : 0.0;
    return (float)v;
}

template <int NS>
__global__ __launch_bounds__(KT) void kstar_rollout_kernel(const SdcKstarModel m, const float* __restrict__ actions, int64_t sb,
                                                          int64_t st, int64_t sc, double* __restrict__ out,
                                                          const double* __restrict__ y0, int B, int nsteps) {
    __shared__ Lds<NS> s;
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * NS;

    // the (beta_p, W_mhd) net's input [bn, Ip, Bt, (In+Out)/2, (Out-In)/2, Elon, UpTri, LoTri] -> s.va
    auto bpw_input = [&]() {
        if (tid < NS * 8) {
            const int sm = tid >> 3, j = tid & 7;
            const double* in = s.inp[sm];
            double v = j == 0 ? s.y[sm][0] : in[j == 1 ? 0 : (j == 2 ? 1 : (j + 7))];
            if (j == 3) v = 0.5 * (in[10] + in[11]);
            if (j == 4) v = 0.5 * (in[11] - in[10]);
            s.va[sm][j] = (float)v;
        }
        __syncthreads();
    };
    // (bp, wmhd) net + H factors + the output row (kstar_solver.py:270-350)
    auto emit_row = [&](int row) {
        bpw_input();
        if (tid < NS * 2) s.bpw[tid >> 1][tid & 1] = 0.0;
        __syncthreads();
        for (int net = 0; net < m.n_bpw; ++net) {
            const float (*r)[MAXW] = mlp<NS>(m.bpw, net, s);
            if (tid < NS * 2) s.bpw[tid >> 1][tid & 1] += (double)r[tid >> 1][tid & 1] * m.bpw_ystd[tid & 1] + m.bpw_ymean[tid & 1];
            __syncthreads();
            if (net + 1 < m.n_bpw) bpw_input();                    // the layers have overwritten the input in s.va
        }
        if (tid < NS && b0 + tid < B) {
            const int sm = tid;
            const double* in = s.inp[sm];
            const double bp = s.bpw[sm][0] / m.n_bpw, wmhd = s.bpw[sm][1] / m.n_bpw;
            const double ip = in[0], bt = in[1], fgw = in[2];
            double ptot = in[3] + in[4] + in[5] + in[6] + in[7];
            ptot = ptot > 1.e-1 ? ptot : 1.e-1;
            const double rin = in[10], rout = in[11], k = in[12];
            const double rgeo = 0.5 * (rin + rout), amin = 0.5 * (rout - rin);
            const double ne = fgw * 10 * (ip / (3.141592653589793 * (amin * amin)));
            const double tau89 = 0.038 * pow(ip, 0.85) * pow(bt, 0.2) * pow(ne, 0.1) * pow(ptot, -0.5) * pow(rgeo, 1.5) * pow(k, 0.5) *
                                 pow(amin / rgeo, 0.3) * pow(2.0, 0.5);
            const double tau98 = 0.0562 * pow(ip, 0.93) * pow(bt, 0.15) * pow(ne, 0.41) * pow(ptot, -0.69) * pow(rgeo, 1.97) * pow(k, 0.78) *
                                 pow(amin / rgeo, 0.58) * pow(2.0, 0.19);
            double* o = out + ((int64_t)(b0 + sm) * (nsteps + 1) + row) * 8;
            o[0] = s.y[sm][0]; o[1] = bp; o[2] = 1.e-6 * wmhd / ptot / tau89; o[3] = 1.e-6 * wmhd / ptot / tau98;
            o[4] = s.y[sm][1]; o[5] = s.y[sm][2]; o[6] = s.y[sm][3]; o[7] = wmhd;
        }
        __syncthreads();
    };

    // ---- steady-state row (predict_0d(steady=True), :171-234): the same for every trajectory, y0 comes from kstar_steady_kernel
    if (tid < NS * 15) s.inp[tid / 15][tid % 15] = m.inputs0[tid % 15];
    if (tid < NS * 4) s.y[tid >> 2][tid & 3] = y0[tid & 3];
    __syncthreads();
    for (int e = tid; e < NS * T * NIN; e += KT) {
        const int sm = e / (T * NIN), j = e % NIN;
        (&s.win[0][0][0])[e] = j < 4 ? (float)s.y[sm][j] : (j < 17 ? window_value(s.inp[sm], j - 4) : (float)m.year_in);
    }
    __syncthreads();
    emit_row(0);

    for (int step = 1; step <= nsteps; ++step) {
        // control(actions[step - 1]), :352-378
        if (tid < NS * 9) {
            const int sm = tid / 9, i = tid - sm * 9;
            const int dst[9] = {0, 3, 4, 5, 12, 13, 14, 10, 11};
            const int b = b0 + sm < B ? b0 + sm : B - 1;
            double a = (double)actions[(int64_t)b * sb + (int64_t)(step - 1) * st + (int64_t)i * sc];
            a = a < m.low_action[i] ? m.low_action[i] : (a > m.high_action[i] ? m.high_action[i] : a);      // np.clip (NaN passes through)
            const double q = (double)(long long)(a * m.scale);                                              // int(): toward zero
            s.inp[sm][dst[i]] = q / m.scale;
        }
        __syncthreads();
        // window: inputs part up one row, new last row
        float keep = 0.f;
        const int e0 = tid;                                        // NS * 9 * 14 <= 512 elements
        const bool mover = e0 < NS * (T - 1) * (NIN - 4);
        int msm = 0, mr = 0, mj = 0;
        if (mover) { msm = e0 / ((T - 1) * (NIN - 4)); const int r = e0 % ((T - 1) * (NIN - 4)); mr = r / (NIN - 4); mj = 4 + r % (NIN - 4); keep = s.win[msm][mr + 1][mj]; }
        __syncthreads();
        if (mover) s.win[msm][mr][mj] = keep;
        if (tid < NS * 13) s.win[tid / 13][T - 1][4 + tid % 13] = window_value(s.inp[tid / 13], tid % 13);
        if (tid < NS * 4) s.yacc[tid >> 2][tid & 3] = 0.0;
        __syncthreads();
        // y = mean over the ensemble of LSTM(window) * ystd + ymean  (model_structure.py:113-117)
        for (int net = 0; net < m.n_lstm; ++net) {
            const float* p = m.lstm + (int64_t)net * m.lstm_stride;
            const float* inv0 = p; const float* off0 = p + NIN; const float* K0 = p + 2 * NIN; const float* R0 = K0 + NIN * G;
            const float* bb0 = R0 + U * G; const float* inv1 = bb0 + G; const float* off1 = inv1 + U; const float* K1 = off1 + U;
            const float* R1 = K1 + U * G; const float* bb1 = R1 + U * G;
            for (int e = tid; e < NS * T * NIN; e += KT) { const int j = e % NIN; (&s.xb[0][0][0])[e] = (&s.win[0][0][0])[e] * inv0[j] + off0[j]; }
            __syncthreads();
            lstm_layer<NS, true>(s, K0, R0, bb0, inv1, off1);
            lstm_layer<NS, false>(s, K1, R1, bb1, nullptr, nullptr);
            for (int e = tid; e < NS * U; e += KT) s.va[e / U][e % U] = s.h[e / U][e % U];
            __syncthreads();
            const float (*r)[MAXW] = mlp<NS>(m.head, net, s);
            if (tid < NS * 4) s.yacc[tid >> 2][tid & 3] += (double)r[tid >> 2][tid & 3] * m.lstm_ystd[tid & 3] + m.lstm_ymean[tid & 3];
            __syncthreads();
        }
        // outputs part up one row, last row <- y
        float keepy = 0.f;
        const bool ymover = tid < NS * (T - 1) * 4;
        if (ymover) keepy = s.win[tid / ((T - 1) * 4)][(tid % ((T - 1) * 4)) / 4 + 1][tid & 3];
        __syncthreads();
        if (ymover) s.win[tid / ((T - 1) * 4)][(tid % ((T - 1) * 4)) / 4][tid & 3] = keepy;
        if (tid < NS * 4) {
            const double y = s.yacc[tid >> 2][tid & 3] / m.n_lstm;
            s.y[tid >> 2][tid & 3] = y;
            s.win[tid >> 2][T - 1][tid & 3] = (float)y;
        }
        __syncthreads();
        emit_row(step);
    }
}

// the steady-state network on the initial inputs (kstar_nn, n_models = 1): y0[4] = prediction * ystd + ymean
__global__ __launch_bounds__(KT) void kstar_steady_kernel(const SdcKstarModel m, double* __restrict__ y0) {
    __shared__ Lds<1> s;
    const int tid = threadIdx.x;
    if (tid < 17) {
        //                   Ip Bt Pnb1a Pnb1b Pnb1c Pec2 Pec3 Zec2 Zec3 InMid OutMid Elon UpTri LoTri InMid GW
        const int src[16] = {0, 1, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 10, 2};
        const double* in = m.inputs0;
        double v = tid < 16 ? in[src[tid]] : m.year_in;
        if (tid == 9) v = 0.5 * (in[10] + in[11]);
        if (tid == 10) v = 0.5 * (in[11] - in[10]);
        if (tid == 14) v = v > 1.265 + 1.e-4 ? 1.0 : 0.0;
        s.va[0][tid] = (float)v;
    }
    __syncthreads();
    const float (*r)[MAXW] = mlp<1>(m.steady, 0, s);
    if (tid < 4) y0[tid] = (double)r[0][tid] * m.nn_ystd[tid] + m.nn_ymean[tid];
}

bool mlp_ok(const SdcKstarMlp& d, int n_in, int n_out) {
    if (!d.params || d.nlayers < 1 || d.nlayers > 6 || d.width[0] != n_in || d.width[d.nlayers] != n_out) return false;
    int64_t need = 0;
    for (int l = 0; l <= d.nlayers; ++l)
        if (d.width[l] < 1 || d.width[l] > MAXW) return false;
    for (int l = 0; l < d.nlayers; ++l) need += 2 * d.width[l] + (int64_t)d.width[l] * d.width[l + 1] + d.width[l + 1];
    return d.stride >= need;                              // the stride covers one network
}

}  // namespace

extern "C" size_t sdc_kstar_lstm_floats(void) { return (size_t)LSTM_FLOATS; }

extern "C" int sdc_kstar_rollout(const SdcKstarModel* mp, const float* actions, int64_t act_b_stride, int64_t act_t_stride,
                                 int64_t act_c_stride, double* out, double* work, int B, int nsteps, void* stream) {
    SDC_REQUIRE(mp && actions && out && work, SDC_ENULL, "sdc_kstar_rollout: null pointer");
    const SdcKstarModel& m = *mp;
    SDC_REQUIRE(B > 0 && nsteps > 0, SDC_EINVAL, "sdc_kstar_rollout: bad sizes");
    SDC_REQUIRE(m.lstm && m.n_lstm >= 1 && m.n_bpw >= 1 && m.lstm_stride >= LSTM_FLOATS, SDC_EINVAL,
                "sdc_kstar_rollout: LSTM parameters missing or stride below sdc_kstar_lstm_floats()");
    SDC_REQUIRE(mlp_ok(m.head, SDC_KSTAR_UNITS, 4), SDC_EINVAL, "sdc_kstar_rollout: head must map 100 -> 4 in 1..6 layers of width <= 256");
    SDC_REQUIRE(mlp_ok(m.steady, 17, 4), SDC_EINVAL, "sdc_kstar_rollout: steady-state net must map 17 -> 4");
    SDC_REQUIRE(mlp_ok(m.bpw, 8, 2), SDC_EINVAL, "sdc_kstar_rollout: (beta_p, W_mhd) net must map 8 -> 2");
    SDC_REQUIRE(m.scale > 0.0, SDC_EINVAL, "sdc_kstar_rollout: scale must be positive");
    hipStream_t s = sdc::as_stream(stream);
    hipLaunchKernelGGL(kstar_steady_kernel, dim3(1), dim3(KT), 0, s, m, work);
    if (B <= 256)
        hipLaunchKernelGGL(kstar_rollout_kernel<1>, dim3((unsigned)B), dim3(KT), 0, s, m, actions, act_b_stride, act_t_stride, act_c_stride,
                           out, (const double*)work, B, nsteps);
    else if (B <= 512)
        hipLaunchKernelGGL(kstar_rollout_kernel<2>, dim3((unsigned)((B + 1) / 2)), dim3(KT), 0, s, m, actions, act_b_stride, act_t_stride,
                           act_c_stride, out, (const double*)work, B, nsteps);
    else
        hipLaunchKernelGGL(kstar_rollout_kernel<4>, dim3((unsigned)((B + 3) / 4)), dim3(KT), 0, s, m, actions, act_b_stride, act_t_stride,
                           act_c_stride, out, (const double*)work, B, nsteps);
    return sdc::check_launch("sdc_kstar_rollout");
}
