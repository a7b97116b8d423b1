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
// Workgroups are persistent (one per CU) and walk the pixel groups: tables staged once, the next group's x and head 0's
// weights requested while the current group is on the matrix cores.
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
    int nblk;                // pixel groups of NS
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

// wave-uniform base (SGPR pair) + 32-bit per-lane byte offset: `global_load_dword v, v_off, s[base:base+1]` -- one VGPR of
// address for all 32 channel rows of a token instead of a 64-bit pointer per row
typedef __attribute__((address_space(1))) float* gptr_t;
typedef __attribute__((address_space(1))) char* gcptr_t;
__device__ __forceinline__ gptr_t uni(const float* p) {
    const uint64_t u = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return (gptr_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ float ldu(gptr_t base, uint32_t byte_off) { return *(gptr_t)((gcptr_t)base + byte_off); }
__device__ __forceinline__ void stu(gptr_t base, uint32_t byte_off, float v) { *(gptr_t)((gcptr_t)base + byte_off) = v; }

// packed rotary step: x = (x0, x1), cs = (cos, sin) -> (x0 cos - x1 sin, x1 cos + x0 sin) in two VALU instructions
typedef float ta2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ta2 ta_rot(ta2 x, ta2 cs) {
    ta2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(x), "v"(cs));                       // (x0 c, x1 c)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(x), "v"(cs), "v"(t));   // (-x1 s + x0 c, x0 s + x1 c)
    return r;
}

__global__ __launch_bounds__(NT) void ta_block_kernel(const TaArgs a) {
    extern __shared__ float lds[];
    float* const xs = lds;                        // [64][XP]   xn, later the y image
    float* const biasT = xs + C * XP;             // [4][32][33]  [head][key][query]
    float* const rotcs = biasT + 4 * 32 * 33;     // [16 m][32 frames][2]: (cos, sin) pairs, one conflict-free 8-byte read per rotation
    float* const red = rotcs + 2 * 32 * 16;       // [2][2][256] LayerNorm partials
    float* const wl = red + 1024;                 // [2][8192] weights of one head: Wq | Wk | Wv ([64 c][32 d] each) | Wo ([32 d][64 co])

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    // Persistent workgroups (one per CU): the bias / rotary tables are staged once, head 0's weights and the x tile of the
    // NEXT pixel group travel while the current one is on the matrix cores, so a tile's only exposed memory latency is the
    // first one's.  XCD-aware walk: workgroup w runs on XCD w & 7 (round-robin dispatch); each XCD takes a contiguous
    // eighth of the pixel groups and its workgroups interleave over it, so the 4 groups sharing a 128-byte line are in
    // flight on one XCD at about the same time (speed only).
    const int nblk = a.nblk, nwg = gridDim.x, w = blockIdx.x;
    const bool xcd = ((nblk & 7) == 0) && ((nwg & 7) == 0);
    const int per = xcd ? (nblk >> 3) : nblk;             // tiles of this workgroup's share
    const int base = xcd ? (w & 7) * per : 0;
    const int first = xcd ? (w >> 3) : w, stride = xcd ? (nwg >> 3) : nwg;
    typedef float nfloat4 __attribute__((ext_vector_type(4)));

    const int tok = tid & 255, half = tid >> 8, pw = tok & 7, f = tok >> 3;
    const uint32_t toff = (uint32_t)(((int64_t)f * a.st + pw + (int64_t)(half * 32) * a.sc) * 4);   // this thread's token (byte
                                                   // offset inside one outer index: < 2^32, host check); 32 channels from here
    auto tile_ptr = [&](int li) -> int64_t {
        const int seq0 = (base + li) * NS;
        const int o = seq0 / a.inner, i0 = seq0 - o * a.inner;
        return (int64_t)o * a.so + i0;
    };

    // ---- once per workgroup: head 0's weights, the transposed bias table, the rotary table
    nfloat4 wreg[4];
    auto fetch_head = [&](int head) {
        if (TA_DBG(a) & 2) return;
#pragma unroll
        for (int i = 0; i < 3; ++i) {                    // q, k, v: rows c = 0..63, 32 columns of this head
            const int c = tid >> 3, c4 = (tid & 7) * 4;
            wreg[i] = *reinterpret_cast<const nfloat4*>(a.wqkv + (int64_t)c * 384 + i * 128 + head * 32 + c4);
        }
        {                                                // Wo rows d = 0..31 of this head, 64 columns
            const int dd = tid >> 4, c4 = (tid & 15) * 4;
            wreg[3] = *reinterpret_cast<const nfloat4*>(a.wo + (int64_t)(head * 32 + dd) * C + c4);
        }
    };
    auto park_head = [&](int buf) {
        float* wb = wl + buf * 8192;
#pragma unroll
        for (int i = 0; i < 3; ++i) *reinterpret_cast<nfloat4*>(wb + i * 2048 + (tid >> 3) * 32 + (tid & 7) * 4) = wreg[i];
        *reinterpret_cast<nfloat4*>(wb + 6144 + (tid >> 4) * 64 + (tid & 15) * 4) = wreg[3];
    };
    float v[32];                                   // raw x of the tile about to be normalised (this thread's token, 32 channels)
    if (first < per) {
        const float* xt = a.x + tile_ptr(first);
#pragma unroll
        for (int c = 0; c < 32; ++c) v[c] = ldu(uni(xt + (int64_t)c * a.sc), toff);
    }
    {
        fetch_head(0);
        float biasreg[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) biasreg[i] = a.bias ? a.bias[tid + i * NT] : 0.0f;
        const float2 rr = a.rot ? *reinterpret_cast<const float2*>(a.rot + tid * 2) : make_float2(1.0f, 0.0f);
        park_head(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + i * NT, h = e >> 10, q = (e >> 5) & 31, kk = e & 31;
            // (bias / scale: the q * scale of the reference is applied to the finished scores, inside the softmax's exponent --
            // S = scale (q k^T + bias / scale) -- instead of to the 32 x 32 q tile of every head: 16 multiplications per head less)
            biasT[(h * 32 + kk) * 33 + q] = biasreg[i] * 5.65685424949238f;
        }
        *reinterpret_cast<float2*>(rotcs + (((tid & 15) * 32) + (tid >> 4)) * 2) = rr;      // tid = frame * 16 + m
    }
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);   // the later-dispatched half loses every arbitration against its SIMD partner otherwise
    const int hw = wave;                           // this wave's pixel of the tile

    // q, k [32 d][32 f]: A = W[c][d = l31] from LDS, B = xn[c][f]; V is formed transposed (operands swapped: [frame rows]
    // [d columns]) so that its accumulator registers are the A fragments of O^T = V P as they stand.  The three products
    // share the xn fragment and run as three independent accumulator chains, their operands fetched one k-step ahead
    // (4 LDS reads feed 3 MFMAs; one chain after the other took 6 reads per 3 and waited for each)
    auto project3 = [&](const float* wm, f32x16 (&pr)[3]) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) pr[i][r] = 0.f;
        const float* wp = wm + lh * 32 + l31;
        const float* xp = xs + lh * XP + hw * 33 + l31;
        float xn_c = xp[0], wq_c = wp[0], wk_c = wp[2048], wv_c = wp[4096];
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) {
            float xn_n = 0.f, wq_n = 0.f, wk_n = 0.f, wv_n = 0.f;
            if (ks + 1 < 32) {
                xn_n = xp[(2 * ks + 2) * XP];
                wq_n = wp[(2 * ks + 2) * 32];
                wk_n = wp[2048 + (2 * ks + 2) * 32];
                wv_n = wp[4096 + (2 * ks + 2) * 32];
            }
            pr[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq_c, xn_c, pr[0], 0, 0, 0);
            pr[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wk_c, xn_c, pr[1], 0, 0, 0);
            pr[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(xn_c, wv_c, pr[2], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);      // next step's four LDS reads ...
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);      // ... then this step's three MFMAs
            xn_c = xn_n; wq_c = wq_n; wk_c = wk_n; wv_c = wv_n;
        }
    };

    for (int li = first; li < per; li += stride) {
        const int64_t tp = tile_ptr(li);
        const bool more = li + stride < per;
        // ---- 1. LayerNorm over the 64 channels of a token (pixel pw, frame f): two threads per token, 32 channels each
        {
            float sm = 0.f;
#pragma unroll
            for (int c = 0; c < 32; ++c) sm += v[c];
            red[half * 256 + tok] = sm;
            __syncthreads();
            const float mean = (red[tok] + red[256 + tok]) * (1.0f / C);
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < 32; ++c) { v[c] -= mean; q += v[c] * v[c]; }
            red[512 + half * 256 + tok] = q;
            __syncthreads();
            const float rstd = rsqrtf((red[512 + tok] + red[768 + tok]) * (1.0f / C) + a.eps);
#pragma unroll
            for (int c = 0; c < 32; ++c) xs[(half * 32 + c) * XP + pw * 33 + f] = v[c] * rstd * a.g[half * 32 + c];
        }
        // ---- 2. heads; this wave owns pixel `wave`
        f32x16 yacc[2];                                // row tiles of y (32 channels each)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) yacc[i][r] = 0.f;
#pragma unroll
        for (int head = 0; head < ((TA_DBG(a) & 1) ? 0 : 4); ++head) {
            const float* wh = wl + (head & 1) * 8192;
            __syncthreads();                           // xn / this head's weights are in LDS; every wave is done with the other buffer
            if (head < 3 || more) fetch_head((head + 1) & 3);      // (head 3 fetches head 0 for the next tile)
            f32x16 pr[3];
            project3(wh, pr);
            if (head < 3 || more) park_head((head + 1) & 1);
            // rotary on (d = 2m, 2m+1) pairs = registers (r, r+1) for even r; frame = l31.  Two packed operations per pair:
            // (x0, x1) -> (x0 c, x1 c) -> (x0 c - x1 s, x1 c + x0 s); the (cos, sin) pair is one 8-byte LDS read.  (The q * scale
            // of the reference rides on the softmax exponent below.)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int m = crow(r, lh) >> 1;
                const ta2 cs = *reinterpret_cast<const ta2*>(rotcs + (m * 32 + l31) * 2);
                ta2 q = {pr[0][r], pr[0][r + 1]}, k = {pr[1][r], pr[1][r + 1]};
                q = ta_rot(q, cs);
                k = ta_rot(k, cs);
                pr[0][r] = q.x; pr[0][r + 1] = q.y;
                pr[1][r] = k.x; pr[1][r + 1] = k.y;
            }
            // The products chain through registers: with the contraction index walked in accumulator-row order (step r covers
            // rows crow(r, 0) | crow(r, 1) on the two half-waves), register r of one product's accumulator IS the operand
            // fragment of the next -- K, Q, V^T, S^T and O^T never visit LDS.
            // S^T[key][query] / scale = bias / scale + sum_d K[d][key] Q[d][query]: A = K registers, B = Q registers; ONE chain that
            // starts from the bias (a dependent fp32 MFMA issues every 64 cycles, its latency: two half chains and their sum bought
            // nothing and cost 48 VALU instructions per head)
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = biasT[(head * 32 + crow(r, lh)) * 33 + l31];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pr[1][r], pr[0][r], acc, 0, 0, 0);
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            // p = exp(scale (acc - mx)) = exp2(acc * (scale log2 e) - mx * (scale log2 e)): one fma + v_exp per element
            const float SL = 0.17677669529663687f * 1.4426950408889634f;
            const float nmx = -mx * SL;
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[r] = __builtin_amdgcn_exp2f(fmaf(acc[r], SL, nmx)); sum += acc[r]; }
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] *= inv;
            // O^T[d][query] = sum_key V^T[key][d] P[key][query]: A = V^T registers, B = P registers (one chain)
            f32x16 oacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(pr[2][r], acc[r], oacc, 0, 0, 0);
            // y[co][f] += sum_d Wo[co][head*32 + d] O^T[d][f]:  A = Wo fragments (w1, d in accumulator-row order), B = O^T registers
#pragma unroll
            for (int r = 0; r < 16; ++r) {             // Wo rows d in accumulator-row order
                const float* wor = wh + 6144 + crow(r, lh) * 64 + l31;
                yacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wor[0], oacc[r], yacc[0], 0, 0, 0);
                yacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wor[32], oacc[r], yacc[1], 0, 0, 0);
            }
        }
        // ---- 3. the residual (L2-warm second read of x) and the next tile's x are requested before the y image forms
        float xr[32];
        {
            const float* xt = a.x + tp;
#pragma unroll
            for (int c = 0; c < 32; ++c) xr[c] = ldu(uni(xt + (int64_t)c * a.sc), toff);
        }
        {   // (unconditional: a conditional load would keep the old v alive across the whole head loop; the last tile re-reads itself)
            const float* xt = a.x + tile_ptr(more ? li + stride : li);
#pragma unroll
            for (int c = 0; c < 32; ++c) v[c] = ldu(uni(xt + (int64_t)c * a.sc), toff);
        }
        __syncthreads();                               // every wave is done reading xn
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) xs[(i * 32 + crow(r, lh)) * XP + hw * 33 + l31] = yacc[i][r];
        __syncthreads();
        {
            const float* yt = a.y + tp;
#pragma unroll
            for (int c = 0; c < 32; ++c) stu(uni(yt + (int64_t)c * a.sc), toff, xs[(half * 32 + c) * XP + pw * 33 + f] + xr[c]);
        }
        __syncthreads();                               // the y image is consumed: xs is free for the next tile's xn
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
    SDC_REQUIRE(((int64_t)31 * st + inner + (int64_t)63 * sc) * 4 < (1ll << 32) && so >= 0 && sc >= 0 && st >= 0, SDC_EINVAL,
                "sdc_tattn_block: one outer index must span less than 4 GB (32-bit lane offsets)");
    TaArgs a;
    a.x = x; a.g = g_pre; a.wqkv = wqkv; a.wo = wo; a.rot = rot; a.bias = bias; a.y = y;
    a.inner = inner; a.nblk = (int)nblk; a.eps = eps; a.so = so; a.sc = sc; a.st = st;
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
    // one persistent workgroup per CU (157 KB of LDS: one fits), at most one per pixel group
    // (the CU count of a device is queried once, like the LDS opt-in: no runtime query per launch, no guessed grid)
    static std::atomic<int> ncu_of[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { sdc::set_error("sdc_tattn_block: hipGetDevice failed"); return SDC_EHIP; }
    int ncu = ncu_of[dev].load(std::memory_order_acquire);
    if (ncu <= 0) {
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) {
            sdc::set_error("sdc_tattn_block: cannot read the CU count of device %d", dev);
            return SDC_EHIP;
        }
        ncu_of[dev].store(ncu, std::memory_order_release);
    }
    const unsigned grid = (unsigned)(nblk < ncu ? nblk : ncu);
    hipLaunchKernelGGL(ta_block_kernel, dim3(grid), dim3(NT), ldsb, sdc::as_stream(stream), a);
    return sdc::check_launch("sdc_tattn_block");
}
