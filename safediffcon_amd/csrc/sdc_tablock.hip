// Fused temporal-attention block of the smoke U-Net at width 64 (Residual(PreNorm(dim, EinopsToAndFrom(Attention)))):
//
//   y = x + Wo . softmax( rot(s Wq xn) rot(Wk xn)^T + relpos ) (Wv xn),   xn = channel LayerNorm(x) * gamma
//
// conv3d.py:262-275 (EinopsToAndFrom 'b c f h w' <-> 'b (h w) f c'), :277-353 (Attention: bias-free qkv / out Linear,
// heads 4 x 32, q*scale, rotary on q and k, relative-position bias, softmax over keys), :165-184 (PreNorm, LayerNorm),
// :383,:402-405 (the wrapping).  A sequence is the 32 frames of one pixel; tokens are strided by H*W.
//
// The unfused chain (norm -> 1x1 C->384 -> attention core -> 1x1 128->C + residual) writes xn, q/k/v and the 128-
// channel attention output to HBM and reads them back: 14 x the bytes of x on top of x itself.  Here one workgroup
// owns 8 adjacent pixels (32-byte runs of the channel-major tensor, all 32 frames, all 4 heads):
//   1. x tile (64 ch x 256 tokens) -> per-token LayerNorm (two threads per token) -> LDS
//   2. per head, per wave (one pixel each, 8 waves = two per SIMD): q, k, v^T = W_h xn on the matrix cores (K = 64, weight
//      fragments fetched one 32-step chain ahead into rotating register sets), q scaled, rotary on the
//      accumulators, S^T = K Q^T (+ bias), softmax, O^T = V P, y += Wo_h O^T -- every product's accumulator registers
//      are the next product's operand fragments, so nothing but xn is read from LDS and there is no barrier in the loop
//   3. y (64 ch x 32 frames per wave) through LDS -> + x -> stores in 32-byte runs.
// x is read twice (the second time for the residual, L2/MALL-warm), y written once.  fp32 MFMA throughout.
#include "sdc_common.h"
#include <cstdlib>

namespace {

constexpr int NT = 512;                  // 8 waves: one pixel each, two waves per SIMD
constexpr int C = 64;
constexpr int NS = 8;                    // pixels per workgroup
constexpr int XP = NS * 33;              // LDS pitch of one channel row: [pixel][33 frames]
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct TaArgs {
    const float* x; const float* g; const float* wqkv; const float* wo; const float* rot; const float* bias;
    float* y;
    int inner;               // pixels per outer index (H*W)
    float eps;
    int64_t so, sc, st;      // element (o, c, pixel i, frame f) at o*so + c*sc + f*st + i
#ifdef SDC_KERNEL_EXPERIMENTS
    int dbg;                 // kernel experiments (SDC_TA_DBG; WRONG RESULTS): 1 no head loop, 2 no weight fetches, 4 no softmax / rotary
#define TA_DBG(a) ((a).dbg)
#else
#define TA_DBG(a) 0         // the shipping library has no result-changing switches (build with -DSDC_KERNEL_EXPERIMENTS for them)
#endif
};

__device__ __forceinline__ int crow(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

__global__ __launch_bounds__(NT) void ta_block_kernel(const TaArgs a) {
    extern __shared__ float lds[];
    float* const xs = lds;                        // [64][XP]   xn, later the y image
    float* const biasT = xs + C * XP;             // [4][32][33]  [head][key][query]
    float* const rotc = biasT + 4 * 32 * 33;      // [32][16]
    float* const rots = rotc + 32 * 16;           // [32][16]
    float* const red = rots + 32 * 16;            // [2][2][256] LayerNorm partials
    float* const wl = red + 1024;                 // [2][8192] weights of one head: Wq | Wk | Wv ([64 c][32 d] each) | Wo ([32 d][64 co])

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    // XCD-aware numbering: the 4 pixel groups sharing a 128-byte line stay on one XCD (speed only)
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int seq0 = bid * NS;
    const int o = seq0 / a.inner, i0 = seq0 - o * a.inner;
    const float* xb = a.x + o * a.so + i0;
    float* yb = a.y + o * a.so + i0;

    // ---- 1. x tile + LayerNorm over the 64 channels of a token (pixel hw, frame f): two threads per token, 32 channels each
    // (every global read of the prologue -- x tile, head 0's weights, bias and rotary tables -- is issued before the first
    // is used: one memory round trip instead of three)
    typedef float nfloat4 __attribute__((ext_vector_type(4)));
    nfloat4 w0reg[4];
    float biasreg[8], rotreg[2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        w0reg[i] = *reinterpret_cast<const nfloat4*>(a.wqkv + (int64_t)(tid >> 3) * 384 + i * 128 + (tid & 7) * 4);
    w0reg[3] = *reinterpret_cast<const nfloat4*>(a.wo + (int64_t)(tid >> 4) * C + (tid & 15) * 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) biasreg[i] = a.bias ? a.bias[tid + i * NT] : 0.0f;
    {
        const float2 rr = a.rot ? *reinterpret_cast<const float2*>(a.rot + tid * 2) : make_float2(1.0f, 0.0f);
        rotreg[0] = rr.x; rotreg[1] = rr.y;
    }
    {
        const int tok = tid & 255, half = tid >> 8, hw = tok & 7, f = tok >> 3;
        const float* xt = xb + (int64_t)f * a.st + hw + (int64_t)(half * 32) * a.sc;
        float v[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) v[c] = xt[(int64_t)c * a.sc];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 32; ++c) s += v[c];
        red[half * 256 + tok] = s;
        __syncthreads();
        const float mean = (red[tok] + red[256 + tok]) * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < 32; ++c) { v[c] -= mean; q += v[c] * v[c]; }
        red[512 + half * 256 + tok] = q;
        __syncthreads();
        const float rstd = rsqrtf((red[512 + tok] + red[768 + tok]) * (1.0f / C) + a.eps);
#pragma unroll
        for (int c = 0; c < 32; ++c) xs[(half * 32 + c) * XP + hw * 33 + f] = v[c] * rstd * a.g[half * 32 + c];
    }
    {   // head 0's weights, the transposed bias table, the rotary table
#pragma unroll
        for (int i = 0; i < 3; ++i) *reinterpret_cast<nfloat4*>(wl + i * 2048 + (tid >> 3) * 32 + (tid & 7) * 4) = w0reg[i];
        *reinterpret_cast<nfloat4*>(wl + 6144 + (tid >> 4) * 64 + (tid & 15) * 4) = w0reg[3];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + i * NT, h = e >> 10, q = (e >> 5) & 31, kk = e & 31;
            biasT[(h * 32 + kk) * 33 + q] = biasreg[i];
        }
        rotc[tid] = rotreg[0];
        rots[tid] = rotreg[1];
    }
    __syncthreads();

    // ---- 2. heads; this wave owns pixel `wave`
    const int hw = wave;
    f32x16 yacc[2];                                // row tiles of y (32 channels each)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) yacc[i][r] = 0.f;
    const float scale = 0.17677669529663687f;

    // The weights of a head (32 KB) are staged in LDS once per workgroup with 16-byte loads -- four per thread -- and shared by
    // the 8 waves; per-wave dword fetches of the fragments (512 per lane) cost 12 % of the kernel.  Two buffers: the next
    // head's weights are fetched before the head's chains and parked after them, one barrier per head.
    nfloat4 wreg[4];
    auto fetch_head = [&](int head) {
        if (TA_DBG(a) & 2) return;
#pragma unroll
        for (int i = 0; i < 3; ++i) {                    // q, k, v: rows c = 0..63, 32 columns of this head
            const int rem = tid, c = rem >> 3, c4 = (rem & 7) * 4;
            wreg[i] = *reinterpret_cast<const nfloat4*>(a.wqkv + (int64_t)c * 384 + i * 128 + head * 32 + c4);
        }
        {                                                // Wo rows d = 0..31 of this head, 64 columns
            const int dd = tid >> 4, c4 = (tid & 15) * 4;
            wreg[3] = *reinterpret_cast<const nfloat4*>(a.wo + (int64_t)(head * 32 + dd) * C + c4);
        }
    };
    auto park_head = [&](int buf) {
        float* w = wl + buf * 8192;
#pragma unroll
        for (int i = 0; i < 3; ++i) *reinterpret_cast<nfloat4*>(w + i * 2048 + (tid >> 3) * 32 + (tid & 7) * 4) = wreg[i];
        *reinterpret_cast<nfloat4*>(w + 6144 + (tid >> 4) * 64 + (tid & 15) * 4) = wreg[3];
    };
    // q / k [32 d][32 f]: A = W[c][d = l31] from LDS, B = xn[c][f]
    auto project = [&](const float* w, f32x16& acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 32; ++ks)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[(2 * ks + lh) * 32 + l31], xs[(2 * ks + lh) * XP + hw * 33 + l31], acc, 0, 0, 0);
    };
    // V is formed transposed (operands swapped: [frame rows][d columns]) so that its accumulator registers are the A
    // fragments of O^T = V P as they stand
    auto project_t = [&](const float* w, f32x16& acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 32; ++ks)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[(2 * ks + lh) * XP + hw * 33 + l31], w[(2 * ks + lh) * 32 + l31], acc, 0, 0, 0);
    };
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);   // the later-dispatched half loses every arbitration against its SIMD partner otherwise
    for (int head = 0; head < ((TA_DBG(a) & 1) ? 0 : 4); ++head) {
        const float* wh = wl + (head & 1) * 8192;
        __syncthreads();                           // this head's weights are in LDS; every wave is done with the other buffer
        if (head < 3) fetch_head(head + 1);
        f32x16 pr[3];
        project(wh, pr[0]);
        project(wh + 2048, pr[1]);
        project_t(wh + 4096, pr[2]);
        if (head < 3) park_head((head + 1) & 1);
        // q * scale, rotary on (d = 2m, 2m+1) pairs = registers (r, r+1) for even r; frame = l31
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const int m = crow(r, lh) >> 1;
            const float cs = rotc[l31 * 16 + m], sn = rots[l31 * 16 + m];
            const float q0 = pr[0][r] * scale, q1 = pr[0][r + 1] * scale;
            pr[0][r] = q0 * cs - q1 * sn; pr[0][r + 1] = q1 * cs + q0 * sn;
            const float k0 = pr[1][r], k1 = pr[1][r + 1];
            pr[1][r] = k0 * cs - k1 * sn; pr[1][r + 1] = k1 * cs + k0 * sn;
        }
        // The products chain through registers: with the contraction index walked in accumulator-row order (step r covers
        // rows crow(r, 0) | crow(r, 1) on the two half-waves), register r of one product's accumulator IS the operand
        // fragment of the next -- K, Q, V^T, S^T and O^T never visit LDS.
        // S^T[key][query] = sum_d K[d][key] Q[d][query]: A = K registers, B = Q registers (two half chains)
        f32x16 acc, acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pr[1][r], pr[0][r], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(pr[1][r + 1], pr[0][r + 1], acc2, 0, 0, 0);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[r] += acc2[r] + biasT[(head * 32 + crow(r, lh)) * 33 + l31];
            mx = fmaxf(mx, acc[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = expf(acc[r] - mx); sum += acc[r]; }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] *= inv;
        // O^T[d][query] = sum_key V^T[key][d] P[key][query]: A = V^T registers, B = P registers
        f32x16 oacc, oacc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { oacc[r] = 0.f; oacc2[r] = 0.f; }
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(pr[2][r], acc[r], oacc, 0, 0, 0);
            oacc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(pr[2][r + 1], acc[r + 1], oacc2, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[r] += oacc2[r];
        // y[co][f] += sum_d Wo[co][head*32 + d] O^T[d][f]:  A = Wo fragments (w1, d in accumulator-row order), B = O^T registers
#pragma unroll
        for (int r = 0; r < 16; ++r) {             // Wo rows d in accumulator-row order
            const float* wor = wh + 6144 + crow(r, lh) * 64 + l31;
            yacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wor[0], oacc[r], yacc[0], 0, 0, 0);
            yacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wor[32], oacc[r], yacc[1], 0, 0, 0);
        }
    }
    __syncthreads();                               // every wave is done reading xn
    // ---- 3. y image [co][pixel][f] over xn, then + x and 32-byte-run stores
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) xs[(i * 32 + crow(r, lh)) * XP + hw * 33 + l31] = yacc[i][r];
    __syncthreads();
    {
        const int tok = tid & 255, half = tid >> 8, pw = tok & 7, f = tok >> 3;
        const int64_t off = (int64_t)f * a.st + pw + (int64_t)(half * 32) * a.sc;
        float xr[32];                              // the residual: 32 loads in flight at once (in batches of 8 they cost four round trips)
#pragma unroll
        for (int c = 0; c < 32; ++c) xr[c] = xb[off + (int64_t)c * a.sc];
#pragma unroll
        for (int c = 0; c < 32; ++c) yb[off + (int64_t)c * a.sc] = xs[(half * 32 + c) * XP + pw * 33 + f] + xr[c];
    }
}

}  // namespace

extern "C" int sdc_tattn_block(const float* x, const float* g_pre, const float* wqkv, const float* wo, const float* rot,
                               const float* bias, float* y, int outer, int inner, int Cc, int ntok, int64_t so, int64_t sc,
                               int64_t st, float eps, void* stream) {
    SDC_REQUIRE(x && g_pre && wqkv && wo && y, SDC_ENULL, "sdc_tattn_block: null pointer");
    SDC_REQUIRE(Cc == 64 && ntok == 32, SDC_EINVAL, "sdc_tattn_block: dim 64 and 32 frames only (got %d, %d)", Cc, ntok);
    SDC_REQUIRE(outer > 0 && inner > 0 && inner % NS == 0, SDC_EINVAL, "sdc_tattn_block: pixels per image must be a multiple of 8");
    const int64_t nblk = (int64_t)outer * inner / NS;
    SDC_REQUIRE(nblk < (1ll << 31), SDC_EINVAL, "sdc_tattn_block: too many sequences");
    TaArgs a;
    a.x = x; a.g = g_pre; a.wqkv = wqkv; a.wo = wo; a.rot = rot; a.bias = bias; a.y = y;
    a.inner = inner; a.eps = eps; a.so = so; a.sc = sc; a.st = st;
#ifdef SDC_KERNEL_EXPERIMENTS
    static const int dbg = getenv("SDC_TA_DBG") ? atoi(getenv("SDC_TA_DBG")) : 0;
    a.dbg = dbg;
#endif
    // the prologue reads the weights with 16-byte and the rotary table with 8-byte vector loads
    SDC_REQUIRE(reinterpret_cast<uintptr_t>(wqkv) % 16 == 0 && reinterpret_cast<uintptr_t>(wo) % 16 == 0 &&
                (!rot || reinterpret_cast<uintptr_t>(rot) % 8 == 0), SDC_EINVAL,
                "sdc_tattn_block: wqkv / wo must be 16-byte aligned and rot 8-byte aligned");
    const size_t ldsb = sizeof(float) * (size_t)(C * XP + 4 * 32 * 33 + 2 * 32 * 16 + 1024 + 2 * 8192);
    static std::atomic<uint64_t> attr{0};
    SDC_LDS_OPTIN(attr, ta_block_kernel, 160 * 1024, "sdc_tattn_block");
    hipLaunchKernelGGL(ta_block_kernel, dim3((unsigned)nblk), dim3(NT), ldsb, sdc::as_stream(stream), a);
    return sdc::check_launch("sdc_tattn_block");
}
