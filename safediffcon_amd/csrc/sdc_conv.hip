// Implicit-GEMM N-d convolution on the gfx950 fp32 matrix cores.
//
//   D[co][p] = sum_k Wp[k][co] * X[k][p],   k = tap*Cin + ci,  p = (b, od, oh, ow) flattened
//
// A = weights (M = Cout), B = input gathered on the fly (N = output positions), both staged
// through double-buffered LDS; v_mfma_f32_32x32x2_f32 does exact fp32 FMA chains, so the result
// is what a k-ordered fmaf loop would give (parity mode, SURVEY 8a: "all arithmetic is fp32").
// The gather folds in: zero padding, stride, two concatenated inputs (skip connections),
// virtual nearest-neighbour upsampling (Upsample+conv) and zero insertion (ConvTranspose3d),
// and arbitrary element strides (the smoke tensor is stored frame-major).
//
// Tile: BM x BN outputs per 256-thread workgroup, BK = 16; wave64 tiles of 32x32 accumulators.
// fp32 MFMA is the bound (64 cycles / instruction / SIMD = 64 FLOP/clk/SIMD): one A and one B
// VGPR feed each MFMA, so plain ds_read_b32 of conflict-free [k][m] / [k][n] LDS rows is enough;
// global loads for chunk c+1 are issued before the MFMAs of chunk c and written to the other LDS
// buffer afterwards (one barrier per chunk).
#include "sdc_common.h"

namespace {

constexpr int BK = 16;
constexpr int NT = 256;

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
    SdcConvDesc d;
    const float* x0;
    const float* x1;
    const float* wp;
    const float* bias;
    const float* res;
    float* y;
    int Ntot;      // B*oD*oH*oW
    int Ktot;      // taps*Cin
    int Cin;
    int lgD, lgH, lgW;
};

template <int BM, int BN, int WM, int WN, bool FAST>
__global__ __launch_bounds__(NT) void conv_kernel(const ConvArgs a) {
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int BROWS = BK * BN / NT;   // B-tile elements per thread
    constexpr int BSTEP = NT / BN;
    constexpr int AROWS = BK * BM / NT;
    constexpr int ASTEP = NT / BM;
    static_assert(BROWS >= 1 && AROWS >= 1, "tile too small");

    __shared__ float As[2][BK][BM];
    __shared__ float Bs[2][BK][BN];

    const SdcConvDesc& d = a.d;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int n0 = blockIdx.x * BN;
    const int m0 = blockIdx.y * BM;

    // ---- B loader: this thread's output position
    const int bj = tid % BN;
    const int brow0 = tid / BN;
    const int p = n0 + bj;
    const bool pvalid = p < a.Ntot;
    int ow = 0, oh = 0, od = 0, ob = 0;
    if (pvalid) {
        int r = p;
        ow = r % d.oW; r /= d.oW;
        oh = r % d.oH; r /= d.oH;
        od = r % d.oD; ob = r / d.oD;
    }
    const int vd0 = od * d.sD - d.pD, vh0 = oh * d.sH - d.pH, vw0 = ow * d.sW - d.pW;
    const int mD = d.up_mode ? ((1 << a.lgD) - 1) : 0;
    const int mH = d.up_mode ? ((1 << a.lgH) - 1) : 0;
    const int mW = d.up_mode ? ((1 << a.lgW) - 1) : 0;

    // ---- A loader
    const int aco = tid % BM;
    const int arow0 = tid / BM;
    const bool acov = (m0 + aco) < d.Cout;

    float breg[BROWS], areg[AROWS];

    auto spatial = [&](int tap, bool& ok, int& id, int& ih, int& iw) {
        const int kw = tap % d.kW;
        const int t2 = tap / d.kW;
        const int kh = t2 % d.kH;
        const int kd = t2 / d.kH;
        const int vd = vd0 + kd, vh = vh0 + kh, vw = vw0 + kw;
        id = vd >> a.lgD; ih = vh >> a.lgH; iw = vw >> a.lgW;
        ok = pvalid && vd >= 0 && vh >= 0 && vw >= 0 && id < d.iD && ih < d.iH && iw < d.iW &&
             ((vd & mD) | (vh & mH) | (vw & mW)) == 0;
    };

    auto load_chunk = [&](int kc) {
        const int kbase = kc * BK;
        // weights
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            const int k = kbase + arow0 + i * ASTEP;
            areg[i] = (acov && k < a.Ktot) ? a.wp[(int64_t)k * d.Cout + m0 + aco] : 0.0f;
        }
        if constexpr (FAST) {
            // whole chunk shares one tap and one input tensor
            const int tap = kbase / a.Cin;
            const int ci0 = kbase - tap * a.Cin;
            bool ok; int id, ih, iw;
            spatial(tap, ok, id, ih, iw);
            const float* src; int64_t sc; int64_t off;
            if (ci0 < d.Cin0) {
                src = a.x0; sc = d.x0s[1];
                off = ob * d.x0s[0] + (int64_t)ci0 * sc + id * d.x0s[2] + ih * d.x0s[3] + iw * d.x0s[4];
            } else {
                src = a.x1; sc = d.x1s[1];
                off = ob * d.x1s[0] + (int64_t)(ci0 - d.Cin0) * sc + id * d.x1s[2] + ih * d.x1s[3] + iw * d.x1s[4];
            }
            if (!ok) off = 0;
            off += (int64_t)brow0 * sc;
#pragma unroll
            for (int i = 0; i < BROWS; ++i) {
                const float v = src[off + (int64_t)(i * BSTEP) * sc];
                breg[i] = ok ? v : 0.0f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < BROWS; ++i) {
                const int k = kbase + brow0 + i * BSTEP;
                float v = 0.0f;
                if (k < a.Ktot) {
                    const int tap = k / a.Cin;
                    const int ci = k - tap * a.Cin;
                    bool ok; int id, ih, iw;
                    spatial(tap, ok, id, ih, iw);
                    if (ok) {
                        if (ci < d.Cin0)
                            v = a.x0[ob * d.x0s[0] + ci * d.x0s[1] + id * d.x0s[2] + ih * d.x0s[3] + iw * d.x0s[4]];
                        else
                            v = a.x1[ob * d.x1s[0] + (ci - d.Cin0) * d.x1s[1] + id * d.x1s[2] + ih * d.x1s[3] +
                                     iw * d.x1s[4]];
                    }
                }
                breg[i] = v;
            }
        }
    };

    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AROWS; ++i) As[buf][arow0 + i * ASTEP][aco] = areg[i];
#pragma unroll
        for (int i = 0; i < BROWS; ++i) Bs[buf][brow0 + i * BSTEP][bj] = breg[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nchunks = (a.Ktot + BK - 1) / BK;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    const int l31 = lane & 31, lh = lane >> 5;
    const int am = wm * (TM * 32) + l31;
    const int bn = wn * (TN * 32) + l31;

    for (int kc = 0; kc < nchunks; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nchunks) load_chunk(kc + 1);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = As[buf][kk + lh][am + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = Bs[buf][kk + lh][bn + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kc + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: D rows (co) live in registers, columns (positions) on lanes -> coalesced along W
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int pp = n0 + wn * (TN * 32) + j * 32 + l31;
        if (pp >= a.Ntot) continue;
        int r = pp;
        const int qw = r % d.oW; r /= d.oW;
        const int qh = r % d.oH; r /= d.oH;
        const int qd = r % d.oD; const int qb = r / d.oD;
        const int64_t yoff = qb * d.ys[0] + qd * d.ys[2] + qh * d.ys[3] + qw * d.ys[4];
        const int64_t roff = a.res ? (qb * d.rs[0] + qd * d.rs[2] + qh * d.rs[3] + qw * d.rs[4]) : 0;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int co = m0 + wm * (TM * 32) + i * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * lh;
                if (co < d.Cout) {
                    float v = acc[i][j][rr];
                    if (a.bias) v += a.bias[co];
                    if (a.res) v += a.res[roff + co * d.rs[1]];
                    a.y[yoff + co * d.ys[1]] = v;
                }
            }
        }
    }
}

int ilog2_exact(int v) {
    if (v == 1) return 0;
    if (v == 2) return 1;
    if (v == 4) return 2;
    return -1;
}

template <int BM, int BN, int WM, int WN>
void launch(const ConvArgs& a, bool fast, hipStream_t s) {
    dim3 grid((a.Ntot + BN - 1) / BN, (a.d.Cout + BM - 1) / BM);
    if (fast)
        hipLaunchKernelGGL((conv_kernel<BM, BN, WM, WN, true>), grid, dim3(NT), 0, s, a);
    else
        hipLaunchKernelGGL((conv_kernel<BM, BN, WM, WN, false>), grid, dim3(NT), 0, s, a);
}

}  // namespace

extern "C" int sdc_conv(const SdcConvDesc* dp, const float* x0, const float* x1, const float* wp, const float* bias,
                        const float* residual, float* y, void* stream) {
    SDC_REQUIRE(dp && x0 && wp && y, SDC_ENULL, "sdc_conv: null pointer");
    const SdcConvDesc& d = *dp;
    SDC_REQUIRE(d.B > 0 && d.Cin0 > 0 && d.Cin1 >= 0 && d.Cout > 0, SDC_EINVAL, "sdc_conv: bad channel/batch counts");
    SDC_REQUIRE(d.Cin1 == 0 || x1, SDC_ENULL, "sdc_conv: Cin1 > 0 but x1 is null");
    SDC_REQUIRE(d.kD > 0 && d.kH > 0 && d.kW > 0 && d.sD > 0 && d.sH > 0 && d.sW > 0, SDC_EINVAL,
                "sdc_conv: bad kernel/stride");
    SDC_REQUIRE(d.precision == 0, SDC_EINVAL, "sdc_conv: only precision 0 (exact fp32 MFMA) is implemented");
    ConvArgs a;
    a.d = d;
    a.lgD = ilog2_exact(d.uD); a.lgH = ilog2_exact(d.uH); a.lgW = ilog2_exact(d.uW);
    SDC_REQUIRE(a.lgD >= 0 && a.lgH >= 0 && a.lgW >= 0, SDC_EINVAL, "sdc_conv: upsample factors must be 1, 2 or 4");
    // output size must agree with what the gather will produce
    auto osz = [](int i, int u, int mode, int k, int s, int p) {
        const int v = mode ? (i - 1) * u + 1 : i * u;
        return (v + 2 * p - k) / s + 1;
    };
    SDC_REQUIRE(d.oD == osz(d.iD, d.uD, d.up_mode, d.kD, d.sD, d.pD) && d.oH == osz(d.iH, d.uH, d.up_mode, d.kH, d.sH, d.pH) &&
                    d.oW == osz(d.iW, d.uW, d.up_mode, d.kW, d.sW, d.pW),
                SDC_EINVAL, "sdc_conv: output size (%d,%d,%d) inconsistent with input/kernel/stride/pad", d.oD, d.oH, d.oW);
    const int64_t ntot = (int64_t)d.B * d.oD * d.oH * d.oW;
    SDC_REQUIRE(ntot < (1ll << 31), SDC_EINVAL, "sdc_conv: too many output positions");
    a.x0 = x0; a.x1 = x1; a.wp = wp; a.bias = bias; a.res = residual; a.y = y;
    a.Ntot = (int)ntot;
    a.Cin = d.Cin0 + d.Cin1;
    a.Ktot = d.kD * d.kH * d.kW * a.Cin;
    const bool fast = (d.Cin0 % BK == 0) && (d.Cin1 % BK == 0);
    hipStream_t s = sdc::as_stream(stream);
    if (d.Cout > 64 && a.Ntot >= 128 * 256)
        launch<128, 128, 2, 2>(a, fast, s);
    else if (d.Cout > 32)
        launch<64, 128, 2, 2>(a, fast, s);
    else
        launch<32, 128, 1, 4>(a, fast, s);
    return sdc::check_launch("sdc_conv");
}
