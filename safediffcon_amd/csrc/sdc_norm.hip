// GroupNorm (+ time scale/shift + SiLU + residual), channel LayerNorm / RMSNorm, SiLU / GELU.
// All HBM-bound: one read + one write per element, 16-byte accesses where the extent allows.
#include "sdc_common.h"

namespace {

constexpr int NT = 256;

// ---------------------------------------------------------------- GroupNorm statistics
// One workgroup per (b, g, split): the group's (C/G)*S elements are contiguous.  fp64 accumulation
// (sum, sum of squares) so that var = E[x^2] - mean^2 keeps fp32-level accuracy for any mean/std
// ratio; partials of all splits are combined by the last-arriving split via plain fp64 slots
// + a second tiny kernel (deterministic order, no atomics).
// stats != null (one split per group): thread 0 goes on to the (mean, rstd) pair with gn_finalize_kernel's arithmetic -- the same
// bits, one launch less (the fine-tuning forward of the 1-D nets: 38 GroupNorms per step, each a few microseconds of work).
__global__ __launch_bounds__(NT) void gn_partial_kernel(const float* __restrict__ x, double* __restrict__ part,
                                                        int64_t n_per_group, int nsplit, float* __restrict__ stats = nullptr,
                                                        double inv_n = 0.0, float eps = 0.f) {
    const int grp = blockIdx.x / nsplit;
    const int sp = blockIdx.x % nsplit;
    const int64_t chunk = (n_per_group + nsplit - 1) / nsplit;
    const int64_t lo = (int64_t)sp * chunk;
    const int64_t hi = lo + chunk < n_per_group ? lo + chunk : n_per_group;
    const float* base = x + (int64_t)grp * n_per_group;
    double s = 0.0, q = 0.0;
    if (((n_per_group | chunk) & 3) == 0) {
        const float4* b4 = reinterpret_cast<const float4*>(base);
        for (int64_t i = lo / 4 + threadIdx.x; i < hi / 4; i += NT) {
            const float4 v = b4[i];
            s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
            q += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        }
    } else {
        for (int64_t i = lo + threadIdx.x; i < hi; i += NT) {
            const double v = base[i];
            s += v;
            q += v * v;
        }
    }
    __shared__ double sh[2][NT / 64];
    s = sdc::wave_sum(s);
    q = sdc::wave_sum(q);
    if ((threadIdx.x & 63) == 0) {
        sh[0][threadIdx.x >> 6] = s;
        sh[1][threadIdx.x >> 6] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0, tq = 0;
        for (int w = 0; w < NT / 64; ++w) { ts += sh[0][w]; tq += sh[1][w]; }
        part[(int64_t)blockIdx.x * 2] = ts;
        part[(int64_t)blockIdx.x * 2 + 1] = tq;
        if (stats) {
            const double mean = ts * inv_n;
            double var = tq * inv_n - mean * mean;
            if (var < 0) var = 0;
            stats[blockIdx.x * 2] = (float)mean;
            stats[blockIdx.x * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
        }
    }
}

__global__ void gn_finalize_kernel(const double* __restrict__ part, float* __restrict__ stats, int ngroups, int nsplit,
                                   double inv_n, float eps) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    double s = 0, q = 0;
    for (int i = 0; i < nsplit; ++i) {
        s += part[((int64_t)g * nsplit + i) * 2];
        q += part[((int64_t)g * nsplit + i) * 2 + 1];
    }
    const double mean = s * inv_n;
    double var = q * inv_n - mean * mean;
    if (var < 0) var = 0;
    stats[g * 2] = (float)mean;
    stats[g * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// the same with one WAVE per group (sdc_gn_finalize with many parts per group: the conv-epilogue sums of the smoke net's upper levels
// are 128-590 pairs per (sample, group), and one thread walking them is a chain of dependent 8-byte loads: 22 us per launch, 30
// launches per step).  Lane l sums parts l, l + 64, ... in order, then the lanes are summed in a fixed butterfly: deterministic and
// independent of the batch, like the serial form (the two orders differ by ~1e-16 relative in fp64, below the fp32 result's rounding).
__global__ __launch_bounds__(256) void gn_finalize_wave_kernel(const double* __restrict__ part, float* __restrict__ stats, int ngroups,
                                                               int nsplit, double inv_n, float eps) {
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (g >= ngroups) return;
    double s = 0, q = 0;
    for (int i = lane; i < nsplit; i += 64) {
        s += part[((int64_t)g * nsplit + i) * 2];
        q += part[((int64_t)g * nsplit + i) * 2 + 1];
    }
    s = sdc::wave_sum(s);
    q = sdc::wave_sum(q);
    if (lane == 0) {
        const double mean = s * inv_n;
        double var = q * inv_n - mean * mean;
        if (var < 0) var = 0;
        stats[g * 2] = (float)mean;
        stats[g * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// ---------------------------------------------------------------- GroupNorm apply
// grid.x = b*C + c (one channel row), grid.y walks 4-element vectors of that row.
// STREAM (tensors far beyond the 256 MB Infinity Cache, i.e. the smoke net's upper levels): one vector per thread, no
// grid-stride loop, nontemporal loads and stores -- tools/stream_probe.hip: 6.5-6.7 TB/s on 1 GiB tensors against 4.6-5.4 for
// the looping / cached forms.  Tensors that fit the cache keep the cached form: their consumer finds them there.
template <bool VEC, bool STREAM = false>
__global__ __launch_bounds__(NT) void gn_apply_kernel(const float* x, const float* __restrict__ stats,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ ss, const int32_t* __restrict__ t_dev,
                                                      int64_t ss_t_stride, int64_t ss_b_stride, int64_t ss_off,
                                                      const float* res, float* y, int C,
                                                      int G, int64_t S) {
    const int bc = blockIdx.x;
    typedef float nf4 __attribute__((ext_vector_type(4)));
    nf4 sv = {0.f, 0.f, 0.f, 0.f}, sr = {0.f, 0.f, 0.f, 0.f};
    const int64_t si = (int64_t)blockIdx.y * NT + threadIdx.x;            // STREAM: gridDim.y * NT >= S / 4
    if constexpr (VEC && STREAM) {                                        // the stream loads travel while the row constants are fetched
        if (si < (S >> 2)) {
            sv = __builtin_nontemporal_load(reinterpret_cast<const nf4*>(x + (int64_t)bc * S) + si);
            if (res) sr = __builtin_nontemporal_load(reinterpret_cast<const nf4*>(res + (int64_t)bc * S) + si);
        }
    }
    const int b = bc / C, c = bc - b * C;
    const int g = c / (C / G);
    const float mean = stats[(b * G + g) * 2], rstd = stats[(b * G + g) * 2 + 1];
    float mul = rstd * gamma[c];
    float add = beta[c] - mean * mul;
    if (ss) {
        const int64_t row = (t_dev ? (int64_t)(*t_dev) : 0) * ss_t_stride + (int64_t)b * ss_b_stride + ss_off;
        const float sc = ss[row + c] + 1.0f, sh = ss[row + C + c];
        mul *= sc;
        add = add * sc + sh;
    }
    const int64_t base = (int64_t)bc * S;
    if constexpr (VEC) {
        const int64_t nv = S >> 2;
        const float4* x4 = reinterpret_cast<const float4*>(x + base);
        const float4* r4 = res ? reinterpret_cast<const float4*>(res + base) : nullptr;
        float4* y4 = reinterpret_cast<float4*>(y + base);
        if constexpr (STREAM) {
            if (si < nv) {
                nf4 v;
                v.x = sdc::silu_f(sv.x * mul + add);
                v.y = sdc::silu_f(sv.y * mul + add);
                v.z = sdc::silu_f(sv.z * mul + add);
                v.w = sdc::silu_f(sv.w * mul + add);
                if (r4) v += sr;
                __builtin_nontemporal_store(v, reinterpret_cast<nf4*>(y4) + si);
            }
            return;
        }
        for (int64_t i = (int64_t)blockIdx.y * NT + threadIdx.x; i < nv; i += (int64_t)gridDim.y * NT) {
            float4 v = x4[i];
            v.x = sdc::silu_f(v.x * mul + add);
            v.y = sdc::silu_f(v.y * mul + add);
            v.z = sdc::silu_f(v.z * mul + add);
            v.w = sdc::silu_f(v.w * mul + add);
            if (r4) {
                const float4 r = r4[i];
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            y4[i] = v;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.y * NT + threadIdx.x; i < S; i += (int64_t)gridDim.y * NT) {
            float v = sdc::silu_f(x[base + i] * mul + add);
            if (res) v += res[base + i];
            y[base + i] = v;
        }
    }
}

// Short rows (S < 1024, e.g. the tokamak U-Net: 16..128 time steps per channel): one workgroup per (b, c)
// row would leave most lanes idle, so threads walk a flat vector index and look the row constants up.
__global__ __launch_bounds__(NT) void gn_apply_flat_kernel(const float* x, const float* __restrict__ stats,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ ss, const int32_t* __restrict__ t_dev,
                                                           int64_t ss_t_stride, int64_t ss_b_stride, int64_t ss_off,
                                                           const float* res, float* y, int C, int G, int nv_row,
                                                           int64_t nv_total) {
    const int64_t trow = (ss && t_dev) ? (int64_t)(*t_dev) * ss_t_stride : 0;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const float4* r4 = res ? reinterpret_cast<const float4*>(res) : nullptr;
    float4* y4 = reinterpret_cast<float4*>(y);
    for (int64_t v = (int64_t)blockIdx.x * NT + threadIdx.x; v < nv_total; v += (int64_t)gridDim.x * NT) {
        const int bc = (int)(v / nv_row);
        const int b = bc / C, c = bc - b * C;
        const int g = c / (C / G);
        const float mean = stats[(b * G + g) * 2], rstd = stats[(b * G + g) * 2 + 1];
        float mul = rstd * gamma[c];
        float add = beta[c] - mean * mul;
        if (ss) {
            const int64_t row = trow + (int64_t)b * ss_b_stride + ss_off;
            const float sc = ss[row + c] + 1.0f, sh = ss[row + C + c];
            mul *= sc;
            add = add * sc + sh;
        }
        float4 q = x4[v];
        q.x = sdc::silu_f(q.x * mul + add);
        q.y = sdc::silu_f(q.y * mul + add);
        q.z = sdc::silu_f(q.z * mul + add);
        q.w = sdc::silu_f(q.w * mul + add);
        if (r4) {
            const float4 r = r4[v];
            q.x += r.x; q.y += r.y; q.z += r.z; q.w += r.w;
        }
        y4[v] = q;
    }
}

// ---------------------------------------------------------------- GroupNorm, statistics + apply in ONE launch
// Small groups (the deep levels of the 1-D / tokamak U-Nets: (C/G) * S <= 32768 elements, 128 KB): one workgroup per
// (b, g) sums the group exactly like gn_partial_kernel (one split) + gn_finalize_kernel, then applies
// (scale + 1, shift), SiLU and the residual to its own channels -- the second read of the group is L2-hot.  Three launches
// (partial, finalize, apply) become one.  fp64 statistics like the three-launch path, summed in another order: the two agree
// to ~2e-6 of the output, not bit for bit -- which is why sdc_gn_fused_ok decides on the group's size alone, never on the batch.
constexpr int NTF = 1024;            // one workgroup per group: 16 waves keep the two sweeps short
__global__ __launch_bounds__(NTF) void gn_fused_kernel(const float* x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ ss, const int32_t* __restrict__ t_dev,
                                                      int64_t ss_t_stride, int64_t ss_b_stride, int64_t ss_off, const float* res,
                                                      float* y, int C, int G, int64_t S, float eps) {
    extern __shared__ float coef[];                    // [cpg] mul, [cpg] add
    const int grp = blockIdx.x;
    const int b = grp / G, g = grp - b * G;
    const int cpg = C / G;
    const int64_t n = (int64_t)cpg * S;
    const float* base = x + (int64_t)grp * n;
    double s = 0.0, q = 0.0;
    if ((n & 3) == 0) {
        const float4* b4 = reinterpret_cast<const float4*>(base);
        for (int64_t i = threadIdx.x; i < n / 4; i += NTF) {
            const float4 v = b4[i];
            s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
            q += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        }
    } else {
        for (int64_t i = threadIdx.x; i < n; i += NTF) {
            const double v = base[i];
            s += v;
            q += v * v;
        }
    }
    __shared__ double sh[2][NTF / 64];
    __shared__ float st2[2];
    s = sdc::wave_sum(s);
    q = sdc::wave_sum(q);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0, tq = 0;
        for (int w = 0; w < NTF / 64; ++w) { ts += sh[0][w]; tq += sh[1][w]; }
        const double inv_n = 1.0 / (double)n;
        const double mean = ts * inv_n;
        double var = tq * inv_n - mean * mean;
        if (var < 0) var = 0;
        st2[0] = (float)mean;
        st2[1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const float mean = st2[0], rstd = st2[1];
    const int64_t row = ss ? (t_dev ? (int64_t)(*t_dev) : 0) * ss_t_stride + (int64_t)b * ss_b_stride + ss_off : 0;
    for (int cc = threadIdx.x; cc < cpg; cc += NTF) {
        const int c = g * cpg + cc;
        float mul = rstd * gamma[c];
        float add = beta[c] - mean * mul;
        if (ss) {
            const float sc = ss[row + c] + 1.0f, shf = ss[row + C + c];
            mul *= sc;
            add = add * sc + shf;
        }
        coef[cc] = mul;
        coef[cpg + cc] = add;
    }
    __syncthreads();
    const float* rb = res ? res + (int64_t)grp * n : nullptr;
    float* yb = y + (int64_t)grp * n;
    if ((S & 3) == 0) {
        const float4* x4 = reinterpret_cast<const float4*>(base);
        const float4* r4 = reinterpret_cast<const float4*>(rb);
        float4* y4 = reinterpret_cast<float4*>(yb);
        const int nv_row = (int)(S >> 2);
        for (int64_t v = threadIdx.x; v < n / 4; v += NTF) {
            const int cc = (int)(v / nv_row);
            const float mul = coef[cc], add = coef[cpg + cc];
            float4 w = x4[v];
            w.x = sdc::silu_f(w.x * mul + add);
            w.y = sdc::silu_f(w.y * mul + add);
            w.z = sdc::silu_f(w.z * mul + add);
            w.w = sdc::silu_f(w.w * mul + add);
            if (rb) { const float4 r = r4[v]; w.x += r.x; w.y += r.y; w.z += r.z; w.w += r.w; }
            y4[v] = w;
        }
    } else {
        for (int64_t i = threadIdx.x; i < n; i += NTF) {
            const int cc = (int)(i / S);
            float v = sdc::silu_f(base[i] * coef[cc] + coef[cpg + cc]);
            if (rb) v += rb[i];
            yb[i] = v;
        }
    }
}

// ---------------------------------------------------------------- channel LayerNorm / RMSNorm
// 64 positions x 4 channel slices per workgroup; positions are the contiguous axis so every
// channel row is read in 256-byte wave-wide segments.  Two passes over C (second pass is L2-hot).
// PL position lanes x (NT/PL) channel slices per workgroup.  CACHE: C <= CPT*NSL, every thread keeps its channel
// values in registers, so x is read from HBM exactly once (otherwise the later passes re-read it L2-hot).
template <int PL, bool CACHE>
__global__ __launch_bounds__(NT) void chan_norm_kernel(const float* x, const float* __restrict__ g,
                                                       const float* res, float* y, int C,
                                                       int64_t S, int mode, float eps) {
    constexpr int NSL = NT / PL;
    constexpr int CPT = 32;
    const int lane = threadIdx.x % PL;
    const int slice = threadIdx.x / PL;
    const int b = blockIdx.y;
    const int64_t pos = (int64_t)blockIdx.x * PL + lane;
    const bool ok = pos < S;
    const int64_t base = (int64_t)b * C * S + pos;
    float vc[CACHE ? CPT : 1];
    float s = 0.f, q = 0.f;
    if constexpr (CACHE) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = slice + i * NSL;
            vc[i] = (ok && c < C) ? x[base + (int64_t)c * S] : 0.f;
            s += vc[i];
            q += vc[i] * vc[i];
        }
    } else if (ok) {
        for (int c = slice; c < C; c += NSL) {
            const float v = x[base + (int64_t)c * S];
            s += v;
            q += v * v;
        }
    }
    __shared__ float sh[2][NSL][PL];
    sh[0][slice][lane] = s;
    sh[1][slice][lane] = q;
    __syncthreads();
    s = 0.f; q = 0.f;
#pragma unroll
    for (int i = 0; i < NSL; ++i) { s += sh[0][i][lane]; q += sh[1][i][lane]; }
    float mean, mul;
    if (mode == 0) {
        mean = s / C;
        // second, centred pass for the variance keeps LN exact when |mean| >> std
        float q2 = 0.f;
        if constexpr (CACHE) {
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                const float dv = (slice + i * NSL < C) ? vc[i] - mean : 0.f;
                q2 += dv * dv;
            }
        } else if (ok) {
            for (int c = slice; c < C; c += NSL) {
                const float dv = x[base + (int64_t)c * S] - mean;
                q2 += dv * dv;
            }
        }
        __syncthreads();
        sh[1][slice][lane] = q2;
        __syncthreads();
        float var = 0.f;
#pragma unroll
        for (int i = 0; i < NSL; ++i) var += sh[1][i][lane];
        mul = 1.0f / sqrtf(var / C + eps);
    } else {
        mean = 0.f;
        mul = sqrtf((float)C) / fmaxf(sqrtf(q), 1e-12f);
    }
    if (!ok) return;
    if constexpr (CACHE) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = slice + i * NSL;
            if (c < C) {
                const int64_t o = base + (int64_t)c * S;
                float v = (vc[i] - mean) * mul * g[c];
                if (res) v += res[o];
                y[o] = v;
            }
        }
    } else {
        for (int c = slice; c < C; c += NSL) {
            const int64_t o = base + (int64_t)c * S;
            float v = (x[o] - mean) * mul * g[c];
            if (res) v += res[o];
            y[o] = v;
        }
    }
}

// Long rows whose length is a multiple of 4 (the smoke net's temporal / spatial attention pre-norms, 1 GB tensors): PLV position
// lanes of FOUR adjacent positions each, NT / PLV channel slices, the thread's <= 32 channels x 4 positions cached in registers:
// x is read from HBM once with 16-byte nontemporal loads in runs of 16 PLV bytes per channel row, y written the same way
// (the scalar form above moved 3.3 TB/s at C = 128 / 256).  Same arithmetic per position as chan_norm_kernel<., true>; with
// PLV = 64 (C <= 128) also the same summation order.
template <int PLV>
__global__ __launch_bounds__(NT) void chan_norm_vec_kernel(const float* x, const float* __restrict__ g, const float* res, float* y, int C,
                                                           int64_t S, int mode, float eps) {
    constexpr int NSL = NT / PLV;
    constexpr int CPT = 32;
    typedef float nf4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x % PLV, slice = threadIdx.x / PLV;
    const int b = blockIdx.y;
    const int64_t S4 = S >> 2;
    const int64_t pos = (int64_t)blockIdx.x * PLV + lane;
    const bool ok = pos < S4;
    const int64_t base = (int64_t)b * C * S4 + pos;
    const nf4* x4 = reinterpret_cast<const nf4*>(x);
    const nf4 zero = {0.f, 0.f, 0.f, 0.f};
    nf4 vc[CPT];
    nf4 s = zero, q = zero;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = slice + i * NSL;
        vc[i] = (ok && c < C) ? __builtin_nontemporal_load(x4 + base + (int64_t)c * S4) : zero;
    }
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        s += vc[i];
        q += vc[i] * vc[i];
    }
    __shared__ nf4 sh[2][NSL][PLV];
    sh[0][slice][lane] = s;
    sh[1][slice][lane] = q;
    __syncthreads();
    s = zero; q = zero;
#pragma unroll
    for (int i = 0; i < NSL; ++i) { s += sh[0][i][lane]; q += sh[1][i][lane]; }
    nf4 mean, mul;
    if (mode == 0) {
        mean = s / (float)C;
        nf4 q2 = zero;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const nf4 dv = (slice + i * NSL < C) ? vc[i] - mean : zero;
            q2 += dv * dv;
        }
        __syncthreads();
        sh[1][slice][lane] = q2;
        __syncthreads();
        nf4 var = zero;
#pragma unroll
        for (int i = 0; i < NSL; ++i) var += sh[1][i][lane];
        var = var / (float)C + eps;
        mul.x = 1.0f / sqrtf(var.x); mul.y = 1.0f / sqrtf(var.y); mul.z = 1.0f / sqrtf(var.z); mul.w = 1.0f / sqrtf(var.w);
    } else {
        mean = zero;
        const float rc = sqrtf((float)C);
        mul.x = rc / fmaxf(sqrtf(q.x), 1e-12f); mul.y = rc / fmaxf(sqrtf(q.y), 1e-12f);
        mul.z = rc / fmaxf(sqrtf(q.z), 1e-12f); mul.w = rc / fmaxf(sqrtf(q.w), 1e-12f);
    }
    if (!ok) return;
    const nf4* r4 = reinterpret_cast<const nf4*>(res);
    nf4* y4 = reinterpret_cast<nf4*>(y);
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = slice + i * NSL;
        if (c < C) {
            const int64_t o = base + (int64_t)c * S4;
            nf4 v = (vc[i] - mean) * mul * g[c];
            if (res) v += __builtin_nontemporal_load(r4 + o);
            __builtin_nontemporal_store(v, y4 + o);
        }
    }
}

__global__ __launch_bounds__(NT) void act_kernel(const float* x, float* y, int64_t n,
                                                 int kind) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const float v = x[i];
        y[i] = kind == 0 ? v / (1.0f + expf(-v)) : 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    }
}

// scratch for GroupNorm partials: (b*G) * nsplit * 2 doubles.  Kept tiny and owned by the caller is
// the ABI rule, so the partial buffer is carved from the *stats* allocation: the caller passes stats
// with room for ngroups*2 floats + ngroups*nsplit*2 doubles (see safediffcon_amd/engine.py).
constexpr int MAX_SPLIT = 16;

// ---------------------------------------------------------------- GroupNorm apply + SiLU + residual INSIDE the output conv
// The last ResnetBlock of a U-Net feeds only `final_conv`, a 1x1(x1) conv to a handful of channels (smoke: Conv3d(64, 7, 1),
// conv3d.py:468-471; Burgers / tokamak: 1D/model/unet.py:376-378): instead of one HBM pass that writes the normalised tensor and a
// conv that reads it back, one streaming kernel reads the raw conv output h and the residual, forms v = SiLU(GN(h)) + res per
// element and accumulates the CO outputs of its four positions in registers: 4 (2 C + CO) bytes per position instead of 4 (4 C + CO).
// grid = (position vectors, samples); a thread owns four adjacent positions; the channel constants and the weight row of a channel
// are wave-uniform (scalar loads).  Output through strides (the smoke net writes eps frame-major).
// (held to 128 registers = four waves per SIMD: left alone the compiler keeps 32 loads in flight in 394 registers and runs one
// workgroup per CU -- 1.2 TB/s.  The channel constants and weight rows come from LDS (one broadcast read each): as scalar loads
// they serialised the loop on s_waitcnt.)
template <int CO>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 8))) void gn_pw_out_kernel(
    const float* __restrict__ h, const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ res, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y, int C, int G, int Cout,
    int64_t S, int64_t plane, int64_t ys0, int64_t ys1, int64_t ys2) {
    typedef float nf4 __attribute__((ext_vector_type(4)));
    extern __shared__ float gpw_lds[];                    // [C][2 + CO]: mul, add, w[0..CO)
    constexpr int RW = 2 + CO;
    const int b = blockIdx.y;
    const int cpg = C / G;
    for (int i = threadIdx.x; i < C * RW; i += NT) {
        const int c = i / RW, k = i - c * RW;
        const int g = c / cpg;
        const float mean = stats[(b * G + g) * 2], rstd = stats[(b * G + g) * 2 + 1];
        const float mul = rstd * gamma[c];
        gpw_lds[i] = k == 0 ? mul : (k == 1 ? beta[c] - mean * mul : (k - 2 < Cout ? w[(int64_t)(k - 2) * C + c] : 0.0f));
    }
    __syncthreads();
    const int64_t si = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (si >= (S >> 2)) return;
    const int64_t cs4 = S >> 2;
    const nf4* h4 = reinterpret_cast<const nf4*>(h + (int64_t)b * C * S) + si;
    const nf4* r4 = reinterpret_cast<const nf4*>((res ? res : h) + (int64_t)b * C * S) + si;
    const float rk = res ? 1.0f : 0.0f;
    nf4 acc[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) {
        const float bv = (bias && o < Cout) ? bias[o] : 0.0f;
        acc[o] = nf4{bv, bv, bv, bv};
    }
    constexpr int UN = 4;                                 // channels in flight per thread: 8 x 16-byte loads (C % 4 == 0: host check)
#pragma unroll 1
    for (int c0 = 0; c0 < C; c0 += UN) {
        nf4 hv[UN], rv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            hv[u] = __builtin_nontemporal_load(h4 + (int64_t)(c0 + u) * cs4);
            rv[u] = __builtin_nontemporal_load(r4 + (int64_t)(c0 + u) * cs4);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const float* row = gpw_lds + (c0 + u) * RW;
            const float mul = row[0], add = row[1];
            nf4 v;
            v.x = fmaf(rk, rv[u].x, sdc::silu_f(fmaf(hv[u].x, mul, add)));
            v.y = fmaf(rk, rv[u].y, sdc::silu_f(fmaf(hv[u].y, mul, add)));
            v.z = fmaf(rk, rv[u].z, sdc::silu_f(fmaf(hv[u].z, mul, add)));
            v.w = fmaf(rk, rv[u].w, sdc::silu_f(fmaf(hv[u].w, mul, add)));
#pragma unroll
            for (int o = 0; o < CO; ++o) {
                const float wv = row[2 + o];
                acc[o].x = fmaf(wv, v.x, acc[o].x); acc[o].y = fmaf(wv, v.y, acc[o].y);
                acc[o].z = fmaf(wv, v.z, acc[o].z); acc[o].w = fmaf(wv, v.w, acc[o].w);
            }
        }
    }
    const int64_t s0 = si << 2;
    const int64_t d = s0 / plane, hw = s0 - d * plane;    // the four positions lie in one (H, W) plane: plane % 4 == 0 (host check)
    float* yb = y + (int64_t)b * ys0 + d * ys2 + hw;
#pragma unroll
    for (int o = 0; o < CO; ++o)
        if (o < Cout) *reinterpret_cast<nf4*>(yb + (int64_t)o * ys1) = acc[o];
}

}  // namespace

extern "C" size_t sdc_gn_stats_bytes(int B, int G) {
    const size_t ngroups = (size_t)B * G;
    return ((ngroups * 2 * sizeof(float) + 15) & ~(size_t)15) + ngroups * MAX_SPLIT * 2 * sizeof(double);
}

extern "C" int sdc_gn_stats(const float* x, float* stats, int B, int C, int G, int64_t S, float eps, void* stream) {
    SDC_REQUIRE(x && stats, SDC_ENULL, "sdc_gn_stats: null pointer");
    SDC_REQUIRE(B > 0 && C > 0 && G > 0 && C % G == 0 && S > 0, SDC_EINVAL, "sdc_gn_stats: bad shape B=%d C=%d G=%d", B, C, G);
    const int ngroups = B * G;
    const int64_t n = (int64_t)(C / G) * S;
    // enough workgroups to fill 256 CUs a few times over, but >= 16K elements per split
    int nsplit = 1;
    while (nsplit < MAX_SPLIT && ngroups * nsplit < 1024 && n / (nsplit * 2) >= 16384) nsplit *= 2;
    // partials live right after the float stats, 16-byte aligned (caller allocates sdc_gn_stats_floats())
    const size_t off = (((size_t)ngroups * 2 * sizeof(float)) + 15) & ~(size_t)15;
    double* part = reinterpret_cast<double*>(reinterpret_cast<char*>(stats) + off);
    hipStream_t s = sdc::as_stream(stream);
    if (nsplit == 1) {
        hipLaunchKernelGGL(gn_partial_kernel, dim3(ngroups), dim3(NT), 0, s, x, part, n, 1, stats, 1.0 / (double)n, eps);
        return sdc::check_launch("sdc_gn_stats");
    }
    hipLaunchKernelGGL(gn_partial_kernel, dim3(ngroups * nsplit), dim3(NT), 0, s, x, part, n, nsplit, (float*)nullptr, 0.0, 0.f);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((ngroups + 63) / 64), dim3(64), 0, s, part, stats, ngroups, nsplit,
                       1.0 / (double)n, eps);
    return sdc::check_launch("sdc_gn_stats");
}

extern "C" int sdc_gn_finalize(const double* parts, float* stats, int B, int G, int nparts, int64_t n_per_group, float eps,
                               void* stream) {
    SDC_REQUIRE(parts && stats, SDC_ENULL, "sdc_gn_finalize: null pointer");
    SDC_REQUIRE(B > 0 && G > 0 && nparts > 0 && n_per_group > 0, SDC_EINVAL, "sdc_gn_finalize: bad shape");
    const int ngroups = B * G;
    // (the form is chosen by the parts per group alone -- a property of the sample's geometry -- never by the batch)
    if (nparts >= 32)
        hipLaunchKernelGGL(gn_finalize_wave_kernel, dim3((ngroups + 3) / 4), dim3(256), 0, sdc::as_stream(stream), parts, stats, ngroups,
                           nparts, 1.0 / (double)n_per_group, eps);
    else
        hipLaunchKernelGGL(gn_finalize_kernel, dim3((ngroups + 63) / 64), dim3(64), 0, sdc::as_stream(stream), parts, stats, ngroups,
                           nparts, 1.0 / (double)n_per_group, eps);
    return sdc::check_launch("sdc_gn_finalize");
}

extern "C" int sdc_gn_pointwise_out(const float* h, const float* stats, const float* gamma, const float* beta, const float* residual,
                                    const float* w, const float* bias, float* y, int B, int C, int G, int Cout, int64_t S,
                                    int64_t plane, int64_t ys0, int64_t ys1, int64_t ys2, void* stream) {
    SDC_REQUIRE(h && stats && gamma && beta && w && y, SDC_ENULL, "sdc_gn_pointwise_out: null pointer");
    SDC_REQUIRE(B > 0 && C > 0 && G > 0 && C % G == 0 && C % 4 == 0 && C <= 2048 && S > 0 && Cout > 0 && Cout <= 16, SDC_EINVAL,
                "sdc_gn_pointwise_out: bad shape B=%d C=%d G=%d Cout=%d (C %% 4 == 0, C <= 2048, Cout <= 16)", B, C, G, Cout);
    SDC_REQUIRE(plane > 0 && S % plane == 0 && plane % 4 == 0 && ys0 % 4 == 0 && ys1 % 4 == 0 && ys2 % 4 == 0, SDC_EINVAL,
                "sdc_gn_pointwise_out: the (H, W) plane and the output strides must be multiples of 4 floats");
    SDC_REQUIRE((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(y)) % 16 == 0,
                SDC_EALIGN, "sdc_gn_pointwise_out: h, residual and y must be 16-byte aligned");
    SDC_REQUIRE(B <= 65535 && (S / 4 + NT - 1) / NT < (1ll << 31), SDC_EINVAL, "sdc_gn_pointwise_out: grid too large");
    hipStream_t s = sdc::as_stream(stream);
    dim3 grid((unsigned)((S / 4 + NT - 1) / NT), (unsigned)B);
#define SDC_GNPW(COV) hipLaunchKernelGGL(gn_pw_out_kernel<COV>, grid, dim3(NT), (size_t)C * (2 + COV) * sizeof(float), s, h, stats, gamma, beta, \
                                         residual, w, bias, y, C, G, Cout, S, plane, ys0, ys1, ys2)
    if (Cout <= 4) SDC_GNPW(4);
    else if (Cout <= 8) SDC_GNPW(8);
    else if (Cout <= 12) SDC_GNPW(12);
    else SDC_GNPW(16);
#undef SDC_GNPW
    return sdc::check_launch("sdc_gn_pointwise_out");
}

extern "C" int sdc_gn_fused_ok(int B, int C, int G, int64_t S) {
    if (B <= 0 || C <= 0 || G <= 0 || C % G || S <= 0) return 0;
    const int64_t n = (int64_t)(C / G) * S;
    // one workgroup per group: worth it while the groups are small (L2-hot second read).  The choice depends on the GROUP
    // only, never on the batch: the fused kernel and the three-launch path sum in different orders (they agree to ~2e-6), so a
    // batch-dependent switch would make a trajectory's rounding depend on the batch it rides in (cf. sdc_chan_norm).
    return n <= 32768 && (C / G) <= 8192;
}

extern "C" int sdc_gn_fused(const float* x, const float* gamma, const float* beta, const float* ss, const int32_t* t_dev,
                            int64_t ss_t_stride, int64_t ss_b_stride, int64_t ss_off, const float* residual, float* y, int B, int C,
                            int G, int64_t S, float eps, void* stream) {
    SDC_REQUIRE(x && gamma && beta && y, SDC_ENULL, "sdc_gn_fused: null pointer");
    SDC_REQUIRE(sdc_gn_fused_ok(B, C, G, S), SDC_EINVAL, "sdc_gn_fused: group too large or too few groups (sdc_gn_fused_ok)");
    // 16-byte loads are taken by the statistics pass when (C/G)*S % 4 == 0 and by the apply pass when S % 4 == 0
    SDC_REQUIRE((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) % 16 == 0 ||
                    ((S & 3) != 0 && (((int64_t)(C / G) * S) & 3) != 0), SDC_EINVAL, "sdc_gn_fused: 16-byte aligned tensors required");
    const size_t lds = (size_t)2 * (C / G) * sizeof(float);
    hipLaunchKernelGGL(gn_fused_kernel, dim3((unsigned)(B * G)), dim3(NTF), lds, sdc::as_stream(stream), x, gamma, beta, ss, t_dev,
                       ss_t_stride, ss_b_stride, ss_off, residual, y, C, G, S, eps);
    return sdc::check_launch("sdc_gn_fused");
}

extern "C" int sdc_gn_apply(const float* x, const float* stats, const float* gamma, const float* beta, const float* ss,
                            const int32_t* t_dev, int64_t ss_t_stride, int64_t ss_b_stride, int64_t ss_off,
                            const float* residual, float* y, int B, int C, int G, int64_t S, void* stream) {
    SDC_REQUIRE(x && stats && gamma && beta && y, SDC_ENULL, "sdc_gn_apply: null pointer");
    SDC_REQUIRE(B > 0 && C > 0 && G > 0 && C % G == 0 && S > 0, SDC_EINVAL, "sdc_gn_apply: bad shape");
    hipStream_t s = sdc::as_stream(stream);
    const bool vec = (S % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                                       reinterpret_cast<uintptr_t>(residual)) % 16 == 0);
    const int64_t work = vec ? S / 4 : S;
    int gx = (int)((work + NT - 1) / NT);
    if (gx > 64) gx = 64;
    dim3 grid(B * C, gx);
    if (vec && S < 1024) {
        const int64_t nvt = (int64_t)B * C * (S / 4);
        int64_t blocks = (nvt + NT - 1) / NT;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(gn_apply_flat_kernel, dim3((unsigned)blocks), dim3(NT), 0, s, x, stats, gamma, beta, ss, t_dev,
                           ss_t_stride, ss_b_stride, ss_off, residual, y, C, G, (int)(S / 4), nvt);
    } else if (vec && (int64_t)B * C * S * 4 >= (512ll << 20) && (work + NT - 1) / NT < 65536) {
        grid.y = (unsigned)((work + NT - 1) / NT);
        hipLaunchKernelGGL((gn_apply_kernel<true, true>), grid, dim3(NT), 0, s, x, stats, gamma, beta, ss, t_dev, ss_t_stride,
                           ss_b_stride, ss_off, residual, y, C, G, S);
    } else if (vec)
        hipLaunchKernelGGL(gn_apply_kernel<true>, grid, dim3(NT), 0, s, x, stats, gamma, beta, ss, t_dev, ss_t_stride,
                           ss_b_stride, ss_off, residual, y, C, G, S);
    else
        hipLaunchKernelGGL(gn_apply_kernel<false>, grid, dim3(NT), 0, s, x, stats, gamma, beta, ss, t_dev, ss_t_stride,
                           ss_b_stride, ss_off, residual, y, C, G, S);
    return sdc::check_launch("sdc_gn_apply");
}

// Wide-channel form (C too large for the register cache): the (C x PL) tile of PL positions is parked in LDS by the
// statistics pass, so x is read from HBM once and the deep, short levels (tokamak: C up to 2048 at 16 positions) do not
// pay two or three latency-bound sweeps over global memory.  Same thread layout: PL position lanes x NT/PL channel slices.
template <int PL>
__global__ __launch_bounds__(NT) void chan_norm_lds_kernel(const float* x, const float* __restrict__ g, const float* res, float* y,
                                                           int C, int64_t S, int mode, float eps) {
    constexpr int NSL = NT / PL;
    extern __shared__ float tile[];                 // [C][PL] then [2][NSL][PL] reduction scratch
    float* sh = tile + (size_t)C * PL;
    const int lane = threadIdx.x % PL, slice = threadIdx.x / PL;
    const int b = blockIdx.y;
    const int64_t pos = (int64_t)blockIdx.x * PL + lane;
    const bool ok = pos < S;
    const int64_t base = (int64_t)b * C * S + pos;
    float s = 0.f, q = 0.f;
    for (int c = slice; c < C; c += NSL) {
        const float v = ok ? x[base + (int64_t)c * S] : 0.f;
        tile[c * PL + lane] = v;
        s += v;
        q += v * v;
    }
    sh[slice * PL + lane] = s;
    sh[(NSL + slice) * PL + lane] = q;
    __syncthreads();
    s = 0.f; q = 0.f;
#pragma unroll
    for (int i = 0; i < NSL; ++i) { s += sh[i * PL + lane]; q += sh[(NSL + i) * PL + lane]; }
    float mean, mul;
    if (mode == 0) {
        mean = s / C;
        float q2 = 0.f;
        for (int c = slice; c < C; c += NSL) { const float dv = tile[c * PL + lane] - mean; q2 += dv * dv; }
        __syncthreads();
        sh[(NSL + slice) * PL + lane] = q2;
        __syncthreads();
        float var = 0.f;
#pragma unroll
        for (int i = 0; i < NSL; ++i) var += sh[(NSL + i) * PL + lane];
        mul = 1.0f / sqrtf(var / C + eps);
    } else {
        mean = 0.f;
        mul = sqrtf((float)C) / fmaxf(sqrtf(q), 1e-12f);
    }
    if (!ok) return;
    for (int c = slice; c < C; c += NSL) {
        const int64_t o = base + (int64_t)c * S;
        float v = (tile[c * PL + lane] - mean) * mul * g[c];
        if (res) v += res[o];
        y[o] = v;
    }
}

extern "C" int sdc_chan_norm(const float* x, const float* g, const float* residual, float* y, int B, int C, int64_t S,
                             int mode, float eps, void* stream) {
    SDC_REQUIRE(x && g && y, SDC_ENULL, "sdc_chan_norm: null pointer");
    SDC_REQUIRE(B > 0 && C > 0 && S > 0 && (mode == 0 || mode == 1), SDC_EINVAL, "sdc_chan_norm: bad arguments");
    SDC_REQUIRE(B < 65536, SDC_EINVAL, "sdc_chan_norm: B too large for grid.y");
    auto lds_bytes = [&](int pl) { return sizeof(float) * ((size_t)C * pl + 2 * (NT / pl) * pl); };
    static std::atomic<uint64_t> attr{0};
    SDC_LDS_OPTIN(attr, chan_norm_lds_kernel<16>, 144 * 1024, "sdc_chan_norm");
    // 64-position tiles for long rows only; shorter rows take the 16-lane form (more, smaller workgroups: those launches
    // are latency-bound, not bandwidth-bound).  The choice depends on the row length alone, never on the batch, so a
    // trajectory's rounding does not depend on how many others share the launch.
    if (S >= 1024 && S % 4 == 0 && C <= 512) {
        // (row length and width alone decide, never the batch -- and never the buffer address: the vector form sums in another
        // order for C > 128, so a fallback on misaligned pointers would make a trajectory's rounding depend on where the pool
        // placed its buffers; rows of S % 4 == 0 floats must therefore start 16-byte aligned)
        const bool al16 = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 15) == 0;
        SDC_REQUIRE(al16, SDC_EALIGN, "sdc_chan_norm: x, y and residual must be 16-byte aligned when S >= 1024, S %% 4 == 0 and C <= 512");
        const int64_t S4 = S / 4;
        if (C <= 128)
            hipLaunchKernelGGL((chan_norm_vec_kernel<64>), dim3((unsigned)((S4 + 63) / 64), B), dim3(NT), 0, sdc::as_stream(stream), x, g, residual, y, C, S, mode, eps);
        else if (C <= 256)
            hipLaunchKernelGGL((chan_norm_vec_kernel<32>), dim3((unsigned)((S4 + 31) / 32), B), dim3(NT), 0, sdc::as_stream(stream), x, g, residual, y, C, S, mode, eps);
        else
            hipLaunchKernelGGL((chan_norm_vec_kernel<16>), dim3((unsigned)((S4 + 15) / 16), B), dim3(NT), 0, sdc::as_stream(stream), x, g, residual, y, C, S, mode, eps);
    } else if (S >= 1024) {
        dim3 grid((unsigned)((S + 63) / 64), B);
        if (C <= 128)
            hipLaunchKernelGGL((chan_norm_kernel<64, true>), grid, dim3(NT), 0, sdc::as_stream(stream), x, g, residual, y, C, S, mode, eps);
        else
            hipLaunchKernelGGL((chan_norm_kernel<64, false>), grid, dim3(NT), 0, sdc::as_stream(stream), x, g, residual, y, C, S, mode, eps);
    } else {   // few positions, many channels (deep levels): 16 position lanes x 16 channel slices
        dim3 grid((unsigned)((S + 15) / 16), B);
        if (C <= 512)
            hipLaunchKernelGGL((chan_norm_kernel<16, true>), grid, dim3(NT), 0, sdc::as_stream(stream), x, g, residual, y, C, S, mode, eps);
        else if (C <= 2048)
            hipLaunchKernelGGL((chan_norm_lds_kernel<16>), grid, dim3(NT), lds_bytes(16), sdc::as_stream(stream), x, g, residual, y, C, S, mode, eps);
        else
            hipLaunchKernelGGL((chan_norm_kernel<16, false>), grid, dim3(NT), 0, sdc::as_stream(stream), x, g, residual, y, C, S, mode, eps);
    }
    return sdc::check_launch("sdc_chan_norm");
}

extern "C" int sdc_act(const float* x, float* y, int64_t n, int kind, void* stream) {
    SDC_REQUIRE(x && y, SDC_ENULL, "sdc_act: null pointer");
    SDC_REQUIRE(n > 0 && (kind == 0 || kind == 1), SDC_EINVAL, "sdc_act: bad arguments");
    int64_t blocks = (n + NT - 1) / NT;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(act_kernel, dim3((unsigned)blocks), dim3(NT), 0, sdc::as_stream(stream), x, y, n, kind);
    return sdc::check_launch("sdc_act");
}
