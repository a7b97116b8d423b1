// Fused DDPM reverse step: eps -> x0 -> closed-form guidance -> eps' -> x0' -> clamp -> posterior mean
// -> + sigma*z -> conditioning writes, one pass over the state (16 B/element with explicit noise,
// 12 B/element with in-kernel Philox noise).  Also: per-sample guidance reductions, conformal
// scores/weights, Philox N(0,1) fill, conditioning-only pass and the device-side step counter.
//
// Layouts (per sample):  burgers (C=3, H, W)   tokamak (C=12, L)   smoke (F, C=7, H, W)  [frame-major]
// gpar (device float[8]) holds the guidance constants so a captured graph survives a new quantile Q:
//   burgers: {w_score, u_bound^2, Q, SCALER=10}
//   tokamak: {w_obj, w_safe, guidance_scaler, safety_threshold, Q}
//   smoke  : {w_safe, safe_bound, Q, standard_fixed_ratio}
// gscal (device float[4*B]) = {hinge-active flag, arg-extremum flat index, extremum, 1/ties} from sdc_guide_reduce.
#include "sdc_common.h"

namespace {

constexpr int NT = 256;

__constant__ float TOK_SCALER[12] = {2, 7, 2, 1, 2, 2, 2, 2, 1, 1, 2, 3};   // tokamak/utils/common.py:16
__constant__ float SMOKE_RESCALER[7] = {2, 19, 20, 17, 20, 1, 1};           // 2d/ddpm/data_2d.py:38

// ------------------------------------------------------------------ Philox4x32-10 + Box-Muller
__device__ __forceinline__ void philox_round(uint32_t c[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ float4 philox_normal4(uint64_t seed, uint32_t draw, uint64_t idx) {
    uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), draw, 0x5DCu};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const float s = 2.3283064365386963e-10f;   // 2^-32
    const float u0 = ((float)c[0] + 0.5f) * s, u1 = ((float)c[1] + 0.5f) * s;
    const float u2 = ((float)c[2] + 0.5f) * s, u3 = ((float)c[3] + 0.5f) * s;
    const float r0 = sqrtf(-2.0f * __logf(fminf(fmaxf(u0, 1e-12f), 1.0f)));
    const float r1 = sqrtf(-2.0f * __logf(fminf(fmaxf(u2, 1e-12f), 1.0f)));
    float s0, c0, s1, c1;
    __sincosf(6.283185307179586f * u1, &s0, &c0);
    __sincosf(6.283185307179586f * u3, &s1, &c1);
    return make_float4(r0 * c0, r0 * s0, r1 * c1, r1 * s1);
}

// ------------------------------------------------------------------ block reductions
__device__ float block_sum(float v, float* sh) {
    v = sdc::wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < NT / 64; ++w) t += sh[w];
    return t;
}

// extremum with index: sign = +1 -> max, -1 -> min
__device__ void block_argext(float v, int idx, float sign, float* shv, int* shi, float& outv, int& outi) {
    float key = v * sign;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ok = __shfl_xor(key, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ok > key || (ok == key && oi < idx)) { key = ok; idx = oi; }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { shv[threadIdx.x >> 6] = key; shi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    float bk = shv[0]; int bi = shi[0];
    for (int w = 1; w < NT / 64; ++w)
        if (shv[w] > bk || (shv[w] == bk && shi[w] < bi)) { bk = shv[w]; bi = shi[w]; }
    outv = bk * sign; outi = bi;
}

// coefficient row of the current step.  DDPM (row index = t): {a, b, c1, c2, sigma, k}.
// DDIM (row index = step number): {a, b, sqrt(alpha_next), c, sigma, k, last}: 1D/model/diffusion.py:500-510.
struct Coef { float a, b, c1, c2, sig, k, last; };
__device__ __forceinline__ Coef load_coef(const float* coef, const int32_t* t_dev) {
    const float* c = coef + (int64_t)(*t_dev) * 8;
    return Coef{c[0], c[1], c[2], c[3], c[4], c[5], c[6]};
}

// safety functional f(state) per sample, evaluated on v(idx) (a callable giving the element value)
//   burgers: 10 * mean|amax over (c=2, h<11)      (1D/utils/guidance.py:66-70)
//   tokamak: min_t<nt 7*x[1,t]                    (tokamak/utils/metrics.py:144-151)
//   smoke  : mean_hw R6*x[F-1,6]                  (2d/inference_2d.py:183)
// `ties` = how many elements attain the extremum (amax / amin modes): torch's amax/amin backward splits the
// gradient evenly between tied elements, which happens for real once x0 is clipped to [-1, 1] (DDIM).
template <typename V>
__device__ void safety_functional(const SdcStepDesc& d, V v, float* shf, int* shi, float& val, int& arg, int& ties,
                                  float& raw) {
    arg = 0; ties = 1; raw = 0.f;
    if (d.model == SDC_MODEL_BURGERS) {
        const int H = d.d1, W = d.d2;
        const int n = 11 * W;
        const int base = 2 * H * W;
        if (!d.use_max) {
            float s = 0.f;
            for (int i = threadIdx.x; i < n; i += NT) s += v(base + i);
            val = 10.0f * block_sum(s, shf) / (float)n;
        } else {
            float m = -INFINITY; int mi = 0x7fffffff;
            for (int i = threadIdx.x; i < n; i += NT) {
                const float x = v(base + i);
                if (x > m) { m = x; mi = base + i; }
            }
            block_argext(m, mi, 1.f, shf, shi, val, arg);
            raw = val;
            float cnt = 0.f;
            for (int i = threadIdx.x; i < n; i += NT) cnt += (v(base + i) == raw) ? 1.f : 0.f;
            ties = (int)(block_sum(cnt, shf) + 0.5f);
            val *= 10.0f;
        }
    } else if (d.model == SDC_MODEL_TOKAMAK) {
        const int L = d.d1, nt = d.cond_idx;
        float m = INFINITY; int mi = 0x7fffffff;
        for (int i = threadIdx.x; i < nt; i += NT) {
            const float x = v(L + i);
            if (x < m) { m = x; mi = L + i; }
        }
        block_argext(m, mi, -1.f, shf, shi, val, arg);
        raw = val;
        float cnt = 0.f;
        for (int i = threadIdx.x; i < nt; i += NT) cnt += (v(L + i) == raw) ? 1.f : 0.f;
        ties = (int)(block_sum(cnt, shf) + 0.5f);
        val *= TOK_SCALER[1];
    } else {
        const int F = d.d0, C = d.d1, HW = d.d2 * d.d3;
        const int base = ((F - 1) * C + 6) * HW;
        float s = 0.f;
        for (int i = threadIdx.x; i < HW; i += NT) s += v(base + i);
        val = SMOKE_RESCALER[6] * block_sum(s, shf) / (float)HW;
    }
}

// hinge argument: > 0  <=> guidance / weight hinge active
__device__ __forceinline__ float hinge_arg(const SdcStepDesc& d, const float* gp, float f) {
    if (d.model == SDC_MODEL_BURGERS) return f + gp[2] - gp[1];
    if (d.model == SDC_MODEL_TOKAMAK) return gp[3] - f + gp[4];
    return f + gp[2] - gp[1];
}

__global__ __launch_bounds__(NT) void guide_reduce_kernel(const SdcStepDesc d, const float* __restrict__ x,
                                                          const float* __restrict__ eps, const float* __restrict__ coef,
                                                          const int32_t* __restrict__ t_dev, const float* __restrict__ gpar,
                                                          float* __restrict__ gscal) {
    __shared__ float shf[NT / 64];
    __shared__ int shi[NT / 64];
    const int b = blockIdx.x;
    const int64_t per = (int64_t)d.d0 * d.d1 * d.d2 * d.d3;
    const Coef c = load_coef(coef, t_dev);
    const float* xb = x + b * per;
    const float* eb = eps + b * per;
    float f, raw; int arg, ties;
    const bool ddim = d.ddim != 0;
    safety_functional(d, [&](int i) {
        const float v = c.a * xb[i] - c.b * eb[i];
        return ddim ? fminf(fmaxf(v, -1.0f), 1.0f) : v;
    }, shf, shi, f, arg, ties, raw);
    if (threadIdx.x == 0) {
        gscal[b * 4] = hinge_arg(d, gpar, f) > 0.f ? 1.0f : 0.0f;
        gscal[b * 4 + 1] = __int_as_float(arg);
        gscal[b * 4 + 2] = raw;                          // the extremum itself (for tie detection)
        gscal[b * 4 + 3] = 1.0f / (float)ties;
    }
}

// closed-form dJ/dx0 for element `i` of sample b (i = flat per-sample index), given x0 there
__device__ __forceinline__ float guide_grad(const SdcStepDesc& d, const float* gp, const float* gscal,
                                            const float* target, int b, int i, float x0) {
    const float active = gscal[b * 4];
    const int arg = __float_as_int(gscal[b * 4 + 1]);
    const float ext = gscal[b * 4 + 2], share = gscal[b * 4 + 3];
    // unique extremum: the index decides (robust to re-computation); ties (clipped values): value equality is exact
    const bool hit = share == 1.0f ? (i == arg) : (x0 == ext);
    if (d.model == SDC_MODEL_BURGERS) {
        const int H = d.d1, W = d.d2;
        const int c = i / (H * W), h = (i / W) % H;
        if (c != 2 || h >= 11) return 0.f;
        if (!d.use_max) return active * gp[0] * 10.0f / (float)(11 * W);
        return hit ? active * gp[0] * 10.0f * share : 0.f;
    }
    if (d.model == SDC_MODEL_TOKAMAK) {
        const int L = d.d1, nt = d.cond_idx;
        const int c = i / L, t = i % L;
        if (t >= nt) return 0.f;
        if (c == 0 || c == 2) {
            const float S = TOK_SCALER[c];
            const float tg = target[((int64_t)b * 3 + c) * nt + t];
            return gp[2] * gp[0] * 2.0f * (S * x0 - tg) * S / (float)nt;
        }
        if (c == 1 && hit) return -gp[2] * gp[1] * TOK_SCALER[1] * active * share;
        return 0.f;
    }
    const int C = d.d1, HW = d.d2 * d.d3, F = d.d0;
    const int f = i / (C * HW), c = (i / HW) % C;
    if (c == 5) return -(1.0f - gp[0]) * SMOKE_RESCALER[5] / (float)(F * HW);
    if (c == 6 && f == F - 1) return gp[0] * SMOKE_RESCALER[6] / (float)HW * active;
    return 0.f;
}

// conditioning writes for element i of sample b; returns true and sets v when the element is imposed
__device__ __forceinline__ bool cond_value(const SdcStepDesc& d, const float* c0, const float* c1, const float* c2,
                                           int b, int i, float& v) {
    if (d.model == SDC_MODEL_BURGERS) {
        const int H = d.d1, W = d.d2, ci = d.cond_idx;
        const int c = i / (H * W), h = (i / W) % H, w = i % W;
        if (d.pad_zero && ((c == 0 && h > ci) || (c >= 1 && h >= ci))) { v = 0.f; return true; }
        if (c == 1 && d.has_wgt) { v = c2[((int64_t)b * H + h) * W + w]; return true; }
        if (c == 0 && h == ci) { v = c1[(int64_t)b * W + w]; return true; }
        if (c == 0 && h == 0) { v = c0[(int64_t)b * W + w]; return true; }
        return false;
    }
    if (d.model == SDC_MODEL_TOKAMAK) {
        const int L = d.d1, nt = d.cond_idx;
        const int c = i / L, t = i % L;
        if (d.has_wgt && c >= 3) { v = c2[((int64_t)b * 9 + (c - 3)) * L + t]; return true; }   // tokamak/...:411,453
        if (d.pad_zero && ((c < 3 && t >= nt) || (c >= 3 && t >= nt - 1))) { v = 0.f; return true; }
        if ((c == 0 || c == 2) && t < nt) { v = c1[((int64_t)b * 2 + (c >> 1)) * nt + t]; return true; }
        if (c < 3 && t == 0) { v = c0[b * 3 + c]; return true; }
        return false;
    }
    const int F = d.d0, C = d.d1, HW = d.d2 * d.d3;
    const int f = i / (C * HW), c = (i / HW) % C, p = i % HW;
    if (d.has_wgt && (c == 3 || c == 4)) { v = c1[(((int64_t)b * F + f) * 2 + (c - 3)) * HW + p]; return true; }
    if (d.impose == 2) return false;          // control only (end of the smoke DDIM loop, 2d/ddpm/diffusion_2d.py:400-401)
    if (f == 0 && c == 0) { v = c0[(int64_t)b * HW + p]; return true; }
    return false;
}

struct StepArgs {
    SdcStepDesc d;
    const float* x; const float* eps; const float* gext; const float* coef;
    const int32_t* t_dev; const int32_t* draw_dev; const float* noise; int64_t noise_stride;
    const float* gpar; const float* gscal; const float* target;
    const float* c0; const float* c1; const float* c2;
    float* xout; float* x0out;
    int64_t per;      // elements per sample
    int64_t nvec;     // total float4 vectors
};

__global__ __launch_bounds__(NT) void step_update_kernel(const StepArgs a) {
    const SdcStepDesc& d = a.d;
    const Coef c = load_coef(a.coef, a.t_dev);
    const int draw = (a.draw_dev ? *a.draw_dev : 0) + d.skip_draws;   // calibration branch discards a draw first
    for (int64_t v = (int64_t)blockIdx.x * NT + threadIdx.x; v < a.nvec; v += (int64_t)gridDim.x * NT) {
        const int64_t e0 = v * 4;
        const int b = (int)(e0 / a.per);
        const int i0 = (int)(e0 - (int64_t)b * a.per);
        const float4 xv = *reinterpret_cast<const float4*>(a.x + e0);
        const float4 ev = *reinterpret_cast<const float4*>(a.eps + e0);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
        const float es[4] = {ev.x, ev.y, ev.z, ev.w};
        float gs[4] = {0.f, 0.f, 0.f, 0.f};
        if (d.guide == 2) {
            const float4 gv = *reinterpret_cast<const float4*>(a.gext + e0);
            gs[0] = gv.x; gs[1] = gv.y; gs[2] = gv.z; gs[3] = gv.w;
        }
        if (d.guide == 3) {   // x0 only (caller evaluates an arbitrary nablaJ on it); DDIM hands over the clipped x0
            float q[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                q[j] = c.a * xs[j] - c.b * es[j];
                if (d.ddim) q[j] = fminf(fmaxf(q[j], -1.0f), 1.0f);
            }
            *reinterpret_cast<float4*>(a.x0out + e0) = make_float4(q[0], q[1], q[2], q[3]);
            continue;
        }
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c.sig != 0.f) {
            if (a.noise) z = *reinterpret_cast<const float4*>(a.noise + (int64_t)draw * a.noise_stride + e0);
            else z = philox_normal4(d.seed, (uint32_t)draw, (uint64_t)v);
        }
        const float zs[4] = {z.x, z.y, z.z, z.w};
        float out[4], x0c[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float e = es[j];
            const bool ddim = d.ddim != 0;
            if (d.guide == 1) {
                float x0 = c.a * xs[j] - c.b * e;
                if (ddim) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
                e = e + guide_grad(d, a.gpar, a.gscal, a.target, b, i0 + j, x0) * c.k;
            } else if (d.guide == 2) {
                e = e + gs[j] * c.k;
            }
            float x0 = c.a * xs[j] - c.b * e;
            if (d.clip || ddim) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
            x0c[j] = x0;
            float o;
            if (!ddim) {
                o = c.c1 * x0 + c.c2 * xs[j] + c.sig * zs[j];
            } else if (c.last != 0.f) {
                o = x0;                                              // time_next < 0: img = x_start
            } else {
                const float er = (c.a * xs[j] - x0) / c.b;           // rederive_pred_noise (:272-273)
                o = x0 * c.c1 + c.c2 * er + c.sig * zs[j];           // x0 sqrt(a_next) + c eps + sigma z (:508-510)
            }
            if (d.impose) {
                float cv;
                if (cond_value(d, a.c0, a.c1, a.c2, b, i0 + j, cv)) o = cv;
            }
            out[j] = o;
        }
        *reinterpret_cast<float4*>(a.xout + e0) = make_float4(out[0], out[1], out[2], out[3]);
        if (a.x0out) *reinterpret_cast<float4*>(a.x0out + e0) = make_float4(x0c[0], x0c[1], x0c[2], x0c[3]);
    }
}

__global__ __launch_bounds__(NT) void impose_kernel(const SdcStepDesc d, float* __restrict__ x, const float* c0,
                                                    const float* c1, const float* c2, int64_t per, int64_t n) {
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < n; e += (int64_t)gridDim.x * NT) {
        const int b = (int)(e / per);
        float cv;
        if (cond_value(d, c0, c1, c2, b, (int)(e - (int64_t)b * per), cv)) x[e] = cv;
    }
}

__global__ __launch_bounds__(NT) void randn_kernel(float* __restrict__ x, int64_t nvec, uint64_t seed,
                                                   const int32_t* __restrict__ draw_dev) {
    const int draw = draw_dev ? *draw_dev : 0;
    for (int64_t v = (int64_t)blockIdx.x * NT + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * NT)
        *reinterpret_cast<float4*>(x + v * 4) = philox_normal4(seed, (uint32_t)draw, (uint64_t)v);
}

__global__ void advance_table_kernel(int32_t* idx_dev, int32_t* t_dev, const int32_t* ttab, int32_t* draw_dev, int ddraw) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int i = *idx_dev + 1;
        *idx_dev = i;
        *t_dev = ttab[i];
        if (draw_dev) *draw_dev += ddraw;
    }
}

__global__ void advance_kernel(int32_t* t_dev, int dt, int32_t* draw_dev, int ddraw) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (t_dev) *t_dev += dt;
        if (draw_dev) *draw_dev += ddraw;
    }
}

// conformal score + importance weight, one workgroup per calibration sample
__global__ __launch_bounds__(NT) void conformal_kernel(const SdcStepDesc d, const float* __restrict__ pred,
                                                       const float* __restrict__ truth, const float* __restrict__ target,
                                                       const float* __restrict__ gpar, float* __restrict__ score,
                                                       float* __restrict__ weight) {
    __shared__ float shf[NT / 64];
    __shared__ int shi[NT / 64];
    const int b = blockIdx.x;
    const int64_t per = (int64_t)d.d0 * d.d1 * d.d2 * d.d3;
    const float* pb = pred + b * per;
    const float* tb = truth + b * per;
    float fp, ft, raw; int arg, ties;
    safety_functional(d, [&](int i) { return pb[i]; }, shf, shi, fp, arg, ties, raw);
    safety_functional(d, [&](int i) { return tb[i]; }, shf, shi, ft, arg, ties, raw);
    float J, sc;
    if (d.model == SDC_MODEL_BURGERS) {
        sc = fabsf(fp - ft);
        J = gpar[0] * fmaxf(hinge_arg(d, gpar, ft), 0.f);
    } else if (d.model == SDC_MODEL_TOKAMAK) {
        sc = fabsf(fp - ft);
        const int L = d.d1, nt = d.cond_idx;
        float s = 0.f;
        for (int i = threadIdx.x; i < 2 * nt; i += NT) {
            const int c = (i / nt) * 2, t = i % nt;
            const float dv = TOK_SCALER[c] * tb[c * L + t] - target[((int64_t)b * 3 + c) * nt + t];
            s += dv * dv;
        }
        const float obj = block_sum(s, shf) / (float)nt;
        J = gpar[2] * (gpar[0] * obj + gpar[1] * fmaxf(hinge_arg(d, gpar, ft), 0.f));
    } else {
        const int F = d.d0, C = d.d1, HW = d.d2 * d.d3;
        sc = fabsf(fp - SMOKE_RESCALER[6] * tb[((int64_t)(F - 1) * C + 6) * HW]);
        float s = 0.f;
        for (int i = threadIdx.x; i < F * HW; i += NT) s += tb[((int64_t)(i / HW) * C + 5) * HW + i % HW];
        const float succ = SMOKE_RESCALER[5] * block_sum(s, shf) / (float)(F * HW);
        J = gpar[3] * (-(1.0f - gpar[0]) * succ + gpar[0] * fmaxf(hinge_arg(d, gpar, ft), 0.f));
    }
    if (threadIdx.x == 0) {
        score[b] = sc;
        weight[b] = expf(-J);
    }
}

int check_desc(const SdcStepDesc& d, const char* who) {
    SDC_REQUIRE(d.model >= 0 && d.model <= 2, SDC_EINVAL, "%s: unknown model %d", who, d.model);
    SDC_REQUIRE(d.B > 0 && d.d0 > 0 && d.d1 > 0 && d.d2 > 0 && d.d3 > 0, SDC_EINVAL, "%s: bad dims", who);
    const int inner = d.model == SDC_MODEL_BURGERS ? d.d2 : d.model == SDC_MODEL_TOKAMAK ? d.d1 : d.d3;
    SDC_REQUIRE(inner % 4 == 0, SDC_EINVAL, "%s: innermost extent %d must be a multiple of 4", who, inner);
    if (d.model == SDC_MODEL_BURGERS)
        SDC_REQUIRE(d.d0 == 3 && d.d1 >= 11 && d.cond_idx >= 0 && d.cond_idx < d.d1, SDC_EINVAL,
                    "%s: burgers expects (3, >=11, W), cond_idx < H", who);
    if (d.model == SDC_MODEL_TOKAMAK)
        SDC_REQUIRE(d.d0 == 12 && d.cond_idx >= 1 && d.cond_idx <= d.d1, SDC_EINVAL, "%s: tokamak expects (12, L), 1 <= nt <= L", who);
    if (d.model == SDC_MODEL_SMOKE) SDC_REQUIRE(d.d1 == 7, SDC_EINVAL, "%s: smoke expects (F, 7, H, W)", who);
    const int64_t per = (int64_t)d.d0 * d.d1 * d.d2 * d.d3;
    SDC_REQUIRE(per < (1ll << 31), SDC_EINVAL, "%s: sample too large", who);
    return SDC_OK;
}

int grid_for(int64_t work) {
    int64_t g = (work + NT - 1) / NT;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" {

int sdc_guide_reduce(const SdcStepDesc* d, const float* x, const float* eps, const float* coef, const int32_t* t_dev,
                     const float* gpar, float* gscal, void* stream) {
    SDC_REQUIRE(d && x && eps && coef && t_dev && gpar && gscal, SDC_ENULL, "sdc_guide_reduce: null pointer");
    if (int rc = check_desc(*d, "sdc_guide_reduce")) return rc;
    hipLaunchKernelGGL(guide_reduce_kernel, dim3(d->B), dim3(NT), 0, sdc::as_stream(stream), *d, x, eps, coef, t_dev,
                       gpar, gscal);
    return sdc::check_launch("sdc_guide_reduce");
}

int sdc_step_update(const SdcStepDesc* d, const float* x, const float* eps, const float* gext, const float* coef,
                    const int32_t* t_dev, const int32_t* draw_dev, const float* noise, int64_t noise_stride,
                    const float* gpar, const float* gscal, const float* target, const float* c0, const float* c1,
                    const float* c2, float* xout, float* x0out, void* stream) {
    SDC_REQUIRE(d && x && eps && coef && t_dev, SDC_ENULL, "sdc_step_update: null pointer");
    if (int rc = check_desc(*d, "sdc_step_update")) return rc;
    SDC_REQUIRE(d->guide >= 0 && d->guide <= 3, SDC_EINVAL, "sdc_step_update: bad guide mode");
    SDC_REQUIRE(d->guide != 1 || (gpar && gscal), SDC_ENULL, "sdc_step_update: built-in guidance needs gpar and gscal");
    SDC_REQUIRE(d->guide != 1 || d->model != SDC_MODEL_TOKAMAK || target, SDC_ENULL, "sdc_step_update: tokamak guidance needs target");
    SDC_REQUIRE(d->guide != 2 || gext, SDC_ENULL, "sdc_step_update: guide=2 needs gext");
    SDC_REQUIRE(d->guide == 3 ? x0out != nullptr : xout != nullptr, SDC_ENULL, "sdc_step_update: output pointer is null");
    if (d->impose) {
        SDC_REQUIRE(c0, SDC_ENULL, "sdc_step_update: impose needs c0");
        SDC_REQUIRE(d->model == SDC_MODEL_SMOKE || c1, SDC_ENULL, "sdc_step_update: impose needs c1");
        SDC_REQUIRE(!d->has_wgt || (d->model == SDC_MODEL_SMOKE ? c1 : c2), SDC_ENULL, "sdc_step_update: has_wgt needs its tensor");
        SDC_REQUIRE(d->impose != 2 || d->model == SDC_MODEL_SMOKE, SDC_EINVAL, "sdc_step_update: impose=2 is smoke-only");
    }
    const uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(eps) | reinterpret_cast<uintptr_t>(xout) |
                         reinterpret_cast<uintptr_t>(x0out) | reinterpret_cast<uintptr_t>(gext) | reinterpret_cast<uintptr_t>(noise);
    SDC_REQUIRE(al % 16 == 0, SDC_EALIGN, "sdc_step_update: tensors must be 16-byte aligned");
    StepArgs a;
    a.d = *d; a.x = x; a.eps = eps; a.gext = gext; a.coef = coef; a.t_dev = t_dev; a.draw_dev = draw_dev;
    a.noise = noise; a.noise_stride = noise_stride; a.gpar = gpar; a.gscal = gscal; a.target = target;
    a.c0 = c0; a.c1 = c1; a.c2 = c2; a.xout = xout; a.x0out = x0out;
    a.per = (int64_t)d->d0 * d->d1 * d->d2 * d->d3;
    a.nvec = a.per * d->B / 4;
    hipLaunchKernelGGL(step_update_kernel, dim3(grid_for(a.nvec)), dim3(NT), 0, sdc::as_stream(stream), a);
    return sdc::check_launch("sdc_step_update");
}

int sdc_impose(const SdcStepDesc* d, float* x, const float* c0, const float* c1, const float* c2, void* stream) {
    SDC_REQUIRE(d && x && c0, SDC_ENULL, "sdc_impose: null pointer");
    if (int rc = check_desc(*d, "sdc_impose")) return rc;
    const int64_t per = (int64_t)d->d0 * d->d1 * d->d2 * d->d3;
    const int64_t n = per * d->B;
    hipLaunchKernelGGL(impose_kernel, dim3(grid_for(n)), dim3(NT), 0, sdc::as_stream(stream), *d, x, c0, c1, c2, per, n);
    return sdc::check_launch("sdc_impose");
}

int sdc_randn(float* x, int64_t n, uint64_t seed, const int32_t* draw_dev, void* stream) {
    SDC_REQUIRE(x, SDC_ENULL, "sdc_randn: null pointer");
    SDC_REQUIRE(n > 0 && n % 4 == 0, SDC_EINVAL, "sdc_randn: n must be a positive multiple of 4");
    SDC_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0, SDC_EALIGN, "sdc_randn: x must be 16-byte aligned");
    hipLaunchKernelGGL(randn_kernel, dim3(grid_for(n / 4)), dim3(NT), 0, sdc::as_stream(stream), x, n / 4, seed, draw_dev);
    return sdc::check_launch("sdc_randn");
}

int sdc_advance(int32_t* t_dev, int dt, int32_t* draw_dev, int ddraw, void* stream) {
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(64), 0, sdc::as_stream(stream), t_dev, dt, draw_dev, ddraw);
    return sdc::check_launch("sdc_advance");
}

int sdc_advance_table(int32_t* idx_dev, int32_t* t_dev, const int32_t* ttab, int32_t* draw_dev, int ddraw, void* stream) {
    SDC_REQUIRE(idx_dev && t_dev && ttab, SDC_ENULL, "sdc_advance_table: null pointer");
    hipLaunchKernelGGL(advance_table_kernel, dim3(1), dim3(64), 0, sdc::as_stream(stream), idx_dev, t_dev, ttab, draw_dev,
                       ddraw);
    return sdc::check_launch("sdc_advance_table");
}

int sdc_conformal_score(const SdcStepDesc* d, const float* pred, const float* truth, const float* target,
                        const float* gpar, float* score, float* weight, void* stream) {
    SDC_REQUIRE(d && pred && truth && gpar && score && weight, SDC_ENULL, "sdc_conformal_score: null pointer");
    if (int rc = check_desc(*d, "sdc_conformal_score")) return rc;
    SDC_REQUIRE(d->model != SDC_MODEL_TOKAMAK || target, SDC_ENULL, "sdc_conformal_score: tokamak needs target");
    hipLaunchKernelGGL(conformal_kernel, dim3(d->B), dim3(NT), 0, sdc::as_stream(stream), *d, pred, truth, target, gpar,
                       score, weight);
    return sdc::check_launch("sdc_conformal_score");
}

}  // extern "C"
