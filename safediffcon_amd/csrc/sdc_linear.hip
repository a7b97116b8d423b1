// nn.Linear over a batch of rows -- the time MLP and the per-block scale/shift MLPs of the fine-tuning path
// (1D/model/unet.py:300-305 `time_mlp`, :158-162 ResnetBlock.mlp; tokamak/model/unet.py the same; conv3d.py:212-216): forward,
// data gradient and weight / bias gradient.
//
// In the samplers these layers are a per-timestep lookup table; in a fine-tuning step they run forward and backward once per
// ResnetBlock: 64 rows against a weight of up to 4096 x 1024 (tokamak, dim 256: 16 MB).  That is a weight stream, not a GEMM:
// the conv kernels' 64-channel tiles put it on 64 workgroups that each walk 1024 taps (90 us per launch, 190 GB/s; 4 ms of a 28 ms
// C3 step).  Here every kernel is shaped by the weight matrix: 16 of its rows (or columns) per workgroup, the contraction split
// over the four waves and summed through LDS in a fixed order, each element of W read once with the rows of x / gy (L2-resident)
// as the other MFMA operand.  fp32 `v_mfma_f32_16x16x4_f32`; the four contraction steps of a 16-byte load go to four MFMAs
// (both operands index the contraction the same way, so any consistent order is the same sum).  No atomics: results are
// reproducible bit for bit.
#include "sdc_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NT = 256;

struct LinArgs {
    const float* x;     // [R][xs]  rows (forward: inputs, K wide; dgrad: gy, M wide; wgrad: gy)
    const float* w;     // [M][K]   nn.Linear weight (wgrad: the second operand x [R][K])
    const float* b;     // [M] or null
    float* y;           // forward [R][ys] (M wide); dgrad gx [R][ys] (K wide); wgrad gw [M][K]
    float* yb;          // wgrad: gbias [M] or null
    int R, K, M;
    int64_t xs, ys, ws; // row strides in floats (ws: wgrad's x)
};

__device__ __forceinline__ f32x4 ld4(const float* p, bool ok) { return ok ? *reinterpret_cast<const f32x4*>(p) : f32x4{0.f, 0.f, 0.f, 0.f}; }

// sum of the four waves' accumulators in wave order, result in wave 0
__device__ __forceinline__ void wave_reduce4(f32x4 (&acc)[4], float* red, int wave, int lane) {
    if (wave)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) *reinterpret_cast<f32x4*>(red + (((wave - 1) * 4 + rt) * 64 + lane) * 4) = acc[rt];
    __syncthreads();
    if (wave == 0)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int q = 0; q < 3; ++q) acc[rt] += *reinterpret_cast<const f32x4*>(red + ((q * 4 + rt) * 64 + lane) * 4);
}

// y[r][m] = b[m] + sum_k x[r][k] W[m][k].  Workgroup = weight rows m0 .. m0+15, wave = a quarter of K (in chunks of 64).
__global__ __launch_bounds__(NT) void lin_fwd_kernel(const LinArgs a) {
    __shared__ __attribute__((aligned(16))) float red[3 * 4 * 64 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * 16;
    const int kq = ((a.K + 255) / 256) * 64, kb = wave * kq, ke = min(a.K, kb + kq);
    const bool mrow = m0 + i < a.M;
    const float* wp = a.w + (int64_t)(m0 + i) * a.K + 4 * g;
    for (int r0 = 0; r0 < a.R; r0 += 64) {
        f32x4 acc[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k = kb; k < ke; k += 64) {
            f32x4 w4[4], x4[4][4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int kk = k + 16 * s + 4 * g;
                const bool kin = kk < ke;
                w4[s] = ld4(wp + k + 16 * s, mrow && kin);
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    const int r = r0 + rt * 16 + i;
                    x4[s][rt] = ld4(a.x + (int64_t)r * a.xs + kk, kin && r < a.R);
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[s][j], x4[s][rt][j], acc[rt], 0, 0, 0);
        }
        wave_reduce4(acc, red, wave, lane);
        if (wave == 0) {
            const int m = m0 + 4 * g;           // rows m .. m+3 of the accumulator tile, column i = row r of the batch
            if (m < a.M) {
                const f32x4 bb = a.b ? *reinterpret_cast<const f32x4*>(a.b + m) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    const int r = r0 + rt * 16 + i;
                    if (r < a.R) *reinterpret_cast<f32x4*>(a.y + (int64_t)r * a.ys + m) = acc[rt] + bb;
                }
            }
        }
        __syncthreads();
    }
}

// gx[r][k] = sum_m gy[r][m] W[m][k].  Workgroup = weight columns k0 .. k0+15, wave = a quarter of M (in chunks of 64).
__global__ __launch_bounds__(NT) void lin_dgrad_kernel(const LinArgs a) {
    __shared__ __attribute__((aligned(16))) float red[3 * 4 * 64 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    const int k0 = blockIdx.x * 16;
    const int mq = ((a.M + 255) / 256) * 64, mb = wave * mq, me = min(a.M, mb + mq);
    const bool kcol = k0 + i < a.K;
    const float* wp = a.w + k0 + i;
    for (int r0 = 0; r0 < a.R; r0 += 64) {
        f32x4 acc[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int m = mb; m < me; m += 64) {
            float w1[4][4];
            f32x4 g4[4][4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int mm = m + 16 * s + 4 * g;
                const bool min_ = mm < me;
#pragma unroll
                for (int j = 0; j < 4; ++j) w1[s][j] = (min_ && kcol) ? wp[(int64_t)(mm + j) * a.K] : 0.f;
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    const int r = r0 + rt * 16 + i;
                    g4[s][rt] = ld4(a.x + (int64_t)r * a.xs + mm, min_ && r < a.R);
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[s][j], g4[s][rt][j], acc[rt], 0, 0, 0);
        }
        wave_reduce4(acc, red, wave, lane);
        if (wave == 0) {
            const int k = k0 + 4 * g;
            if (k < a.K)
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    const int r = r0 + rt * 16 + i;
                    if (r < a.R) *reinterpret_cast<f32x4*>(a.y + (int64_t)r * a.ys + k) = acc[rt];
                }
        }
        __syncthreads();
    }
}

// gw[m][k] = sum_r gy[r][m] x[r][k];  gb[m] = sum_r gy[r][m].  Workgroup = rows m0 .. m0+15 x 256 columns, wave = 64 columns.
__global__ __launch_bounds__(NT) void lin_wgrad_kernel(const LinArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * 16, k0 = blockIdx.y * 256 + wave * 64;
    const bool mcol = m0 + i < a.M;
    f32x4 acc[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) acc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    for (int r = 0; r < a.R; r += 16) {
        float ga[4], xb[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rr = r + 4 * g + j;
            const bool rin = rr < a.R;
            ga[j] = (rin && mcol) ? a.x[(int64_t)rr * a.xs + m0 + i] : 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const int k = k0 + kt * 16 + i;
                xb[kt][j] = (rin && k < a.K) ? a.w[(int64_t)rr * a.ws + k] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bsum += ga[j];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) acc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[j], xb[kt][j], acc[kt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        const int k = k0 + kt * 16 + i;
        if (k < a.K)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int m = m0 + 4 * g + v;
                if (m < a.M) a.y[(int64_t)m * a.K + k] = acc[kt][v];
            }
    }
    if (a.yb && blockIdx.y == 0 && wave == 0) {          // the four lane groups hold the rows r = 4 g + j (mod 16) of column m0 + i
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        if (g == 0 && mcol) a.yb[m0 + i] = bsum;
    }
}

int lin_check(const char* what, const void* p0, const void* p1, const void* p2, int R, int K, int M, int64_t s0, int64_t s1) {
    SDC_REQUIRE(p0 && p1 && p2, SDC_ENULL, "%s: null pointer", what);
    SDC_REQUIRE(R > 0 && K > 0 && M > 0, SDC_EINVAL, "%s: empty problem (%d rows, %d -> %d)", what, R, K, M);
    SDC_REQUIRE(K % 4 == 0 && M % 4 == 0 && s0 % 4 == 0 && s1 % 4 == 0, SDC_EINVAL,
                "%s: feature counts and row strides must be multiples of 4 (got %d -> %d, strides %lld, %lld)", what, K, M, (long long)s0,
                (long long)s1);
    SDC_REQUIRE((reinterpret_cast<uintptr_t>(p0) | reinterpret_cast<uintptr_t>(p1) | reinterpret_cast<uintptr_t>(p2)) % 16 == 0, SDC_EINVAL,
                "%s: pointers must be 16-byte aligned", what);
    return SDC_OK;
}

}  // namespace

extern "C" int sdc_linear(const float* x, const float* w, const float* bias, float* y, int rows, int K, int M, int64_t x_stride,
                          int64_t y_stride, void* stream) {
    if (int rc = lin_check("sdc_linear", x, w, y, rows, K, M, x_stride, y_stride)) return rc;
    SDC_REQUIRE(!bias || reinterpret_cast<uintptr_t>(bias) % 16 == 0, SDC_EINVAL, "sdc_linear: bias must be 16-byte aligned");
    SDC_REQUIRE(x_stride >= K && y_stride >= M, SDC_EINVAL, "sdc_linear: row strides shorter than the rows");
    LinArgs a{x, w, bias, y, nullptr, rows, K, M, x_stride, y_stride, 0};
    hipLaunchKernelGGL(lin_fwd_kernel, dim3((unsigned)((M + 15) / 16)), dim3(NT), 0, sdc::as_stream(stream), a);
    return sdc::check_launch("sdc_linear");
}

extern "C" int sdc_linear_dgrad(const float* gy, const float* w, float* gx, int rows, int K, int M, int64_t gy_stride, int64_t gx_stride,
                                void* stream) {
    if (int rc = lin_check("sdc_linear_dgrad", gy, w, gx, rows, K, M, gy_stride, gx_stride)) return rc;
    SDC_REQUIRE(gy_stride >= M && gx_stride >= K, SDC_EINVAL, "sdc_linear_dgrad: row strides shorter than the rows");
    LinArgs a{gy, w, nullptr, gx, nullptr, rows, K, M, gy_stride, gx_stride, 0};
    hipLaunchKernelGGL(lin_dgrad_kernel, dim3((unsigned)((K + 15) / 16)), dim3(NT), 0, sdc::as_stream(stream), a);
    return sdc::check_launch("sdc_linear_dgrad");
}

extern "C" int sdc_linear_wgrad(const float* gy, const float* x, float* gw, float* gbias, int rows, int K, int M, int64_t gy_stride,
                                int64_t x_stride, void* stream) {
    if (int rc = lin_check("sdc_linear_wgrad", gy, x, gw, rows, K, M, gy_stride, x_stride)) return rc;
    SDC_REQUIRE(gy_stride >= M && x_stride >= K, SDC_EINVAL, "sdc_linear_wgrad: row strides shorter than the rows");
    LinArgs a{gy, x, nullptr, gw, gbias, rows, K, M, gy_stride, 0, x_stride};
    hipLaunchKernelGGL(lin_wgrad_kernel, dim3((unsigned)((M + 15) / 16), (unsigned)((K + 255) / 256)), dim3(NT), 0,
                       sdc::as_stream(stream), a);
    return sdc::check_launch("sdc_linear_wgrad");
}
