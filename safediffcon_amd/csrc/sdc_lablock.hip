// Fused LinearAttention block (Residual(PreNorm(dim, LinearAttention(dim))), heads 4 x dim_head 32, dim C in {64,128}):
//
//   y = x + post( Wo . LA(pre(x)) + bo ),   LA: q = softmax_d(Wq xn) * 32^-0.5, k = softmax_n(Wk xn), v = Wv xn,
//                                               ctx = k v^T (32x32 per head), out = ctx^T q
// 1D/model/unet.py:182-222 (+ PreNorm :64-71, LayerNorm :53-63), tokamak/model/unet.py:186-222 (RMSNorm :45-51),
// conv3d.py:232-258 (SpatialLinearAttention, PreNorm :176-184).
//
// The unfused chain writes xn, q/k/v (3 x 128 channels), out (128 channels) and the projected y to HBM and reads them
// back -- ~27 x the bytes of x per block at C = 64.  Here x is read three times and y written once:
//   pass 1  la_blk_ctx : x tile -> channel norm -> K^T = xn^T Wk^T on the matrix cores -> online softmax over tokens
//                        (running max per d, rescaled accumulators) -> M^T[c][d] += sum_tok xn[c][tok] p[tok][d]
//                        (ctx = M Wv^T / rowsum: V is never formed per token)
//   mid     la_blk_mid : merge the token splits (log-sum-exp), ctx = M Wv^T / sum, T[co][h,d] = sum_e Wo[co][h,e] ctx_h[d][e]
//   pass 2  la_blk_out : x tile -> channel norm -> Q = Wq xn -> softmax over d, * scale -> y = T q + bo -> channel
//                        norm (LayerNorm / RMSNorm / none) -> + x -> store
// All products are v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate).  Tokens contiguous, n % 64 == 0.
#include "sdc_common.h"

namespace {

constexpr int NT = 256;
constexpr int TT = 64;            // tokens per tile
constexpr int XP = TT + 1;        // LDS pitch of a token row (odd: conflict-free when lanes walk channels)
constexpr int HID = 128;
constexpr float LOG2E = 1.4426950408889634f;
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct LaArgs {
    const float* x;
    const float* g_pre;
    const float* wqkv;      // packed [C][384]  (q | k | v, each heads*32)
    const float* wo;        // packed [128][C]
    const float* bo;        // [C] or null
    const float* g_post;    // [C] or null
    float* part;            // [nseq][nsplit][4][32*(C+2)]
    float* tt;              // [nseq][128][C]
    float* y;
    // GroupNorm + SiLU (+ residual) of the producing ResnetBlock applied on load (sdc_linattn_block_gn): x is then the RAW conv
    // output, gn_stats (mean, rstd) per (outer index, group), gn_res the residual branch (same strides as x) or null
    const float* gn_stats; const float* gn_gamma; const float* gn_beta; const float* gn_res;
    float* gn_hout;         // pass 1 stores h here (x's strides); pass 2 then reads it as its x
    int gn_G;
    int inner, nsplit, tiles_per_split, ntiles, tiles_per_blk;
    int pre_mode, post_mode;
    float eps;
    int64_t so, sc, si;
};

// x tile (C channels x 64 tokens) -> channel norm over C per token -> xs[c][tok].  Thread (tok = tid & 63, group =
// tid >> 6) owns C/4 channels of one token; the per-token statistics are combined across the 4 groups through `red`.
// mode 0: (x - mean) * rsqrt(var + eps) * g (two-pass variance); mode 1: x / max(||x||, 1e-12) * g * sqrt(C).
// fetch_tile only issues the loads (the next tile's are issued while the current one is on the matrix cores);
// norm_tile consumes them; `xr`, when given, keeps the raw tile for the residual add.
// wave-uniform base (SGPR pair) + one 32-bit per-lane byte offset: the channel row of an access is wave-uniform (a wave owns whole
// rows: grp = tid >> 6), only the token is per lane -- left to the compiler every access formed a 64-bit address per lane
// (17 v_lshl_add_u64 + 20-66 v_add_u32 per tile in these kernels, where a VALU instruction costs matrix-pipe time)
typedef __attribute__((address_space(1))) float* la_gptr;
typedef __attribute__((address_space(1))) char* la_gcptr;
__device__ __forceinline__ la_gptr la_uni(const float* p) {
    const uint64_t u = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return (la_gptr)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ float la_ld(la_gptr base, uint32_t byte_off) { return *(la_gptr)((la_gcptr)base + byte_off); }
__device__ __forceinline__ void la_st(la_gptr base, uint32_t byte_off, float v) { *(la_gptr)((la_gcptr)base + byte_off) = v; }

// max(a, b, c) in one instruction (fmaxf(fmaxf()) came out as v_max_f32 pairs plus a canonicalising v_max x, x per input: 45 for 32)
__device__ __forceinline__ float la_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

template <int C>
__device__ __forceinline__ void fetch_tile(const float* __restrict__ xb, int64_t sc, int tid, float (&v)[C / 4]) {
    constexpr int CG = C / 4;
    const int tok = tid & 63, grp = __builtin_amdgcn_readfirstlane(tid >> 6);
    la_gptr rp = la_uni(xb + (int64_t)(grp * CG) * sc);
    // (the lane offset passes through an empty asm before every access: left alone its zero-extension is hoisted out of the loop as a
    // 64-bit register pair and every load forms its address with a v_lshl_add_u64 -- 16 per tile here -- instead of taking the
    // scalar base + 32-bit offset form)
    uint32_t off = (uint32_t)tok * 4u;
#pragma unroll
    for (int k = 0; k < CG; ++k) {
        asm volatile("" : "+v"(off));
        v[k] = la_ld(rp, off);
        rp += sc;
        asm volatile("" : "+s"(rp));
    }
}

// GroupNorm-on-load form (template GN; the producing ResnetBlock's second GroupNorm + SiLU + residual add, conv3d.py:189-230,
// never written to HBM): gn_apply_tile turns the raw conv tile into the block input h = SiLU(x * mul[c] + add[c]) + r with the
// same expressions as gn_apply_kernel (sdc_norm.hip), coefficients from LDS.
template <int C>
__device__ __forceinline__ void gn_apply_tile(float (&v)[C / 4], const float* __restrict__ rb, int64_t sc, const float* __restrict__ gcoef,
                                              int tid) {
    constexpr int CG = C / 4;
    const int tok = tid & 63, grp = tid >> 6;
    float rv[CG];                                   // the residual tile is requested first: it travels under the SiLUs
    if (rb) {
        la_gptr rp = la_uni(rb + (int64_t)(__builtin_amdgcn_readfirstlane(grp) * CG) * sc);
        uint32_t off = (uint32_t)tok * 4u;
#pragma unroll
        for (int k = 0; k < CG; ++k) {
            asm volatile("" : "+v"(off));
            rv[k] = la_ld(rp, off);
            rp += sc;
            asm volatile("" : "+s"(rp));
        }
    } else {
#pragma unroll
        for (int k = 0; k < CG; ++k) rv[k] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < CG; ++k) {
        const float mul = gcoef[grp * CG + k], add = gcoef[C + grp * CG + k];
        v[k] = sdc::silu_f(v[k] * mul + add) + rv[k];
    }
}
// (mul, add) per channel of outer index o into LDS: mul = rstd * gamma, add = beta - mean * mul
template <int C>
__device__ __forceinline__ void gn_coef_fill(const float* __restrict__ stats, const float* __restrict__ gamma,
                                             const float* __restrict__ beta, int G, int o, float* __restrict__ gcoef, int tid) {
    for (int c = tid; c < C; c += NT) {
        const int g = c / (C / G);
        const float mean = stats[(o * G + g) * 2], rstd = stats[(o * G + g) * 2 + 1];
        const float mul = rstd * gamma[c];
        gcoef[c] = mul;
        gcoef[C + c] = beta[c] - mean * mul;
    }
}

template <int C>
__device__ __forceinline__ void norm_tile(float (&v)[C / 4], const float* __restrict__ g, int mode, float eps,
                                          float* __restrict__ xs, float* __restrict__ xr, float* __restrict__ red, int tid) {
    constexpr int CG = C / 4;
    const int tok = tid & 63, grp = tid >> 6;
    if (xr) {
#pragma unroll
        for (int k = 0; k < CG; ++k) xr[(grp * CG + k) * XP + tok] = v[k];
    }
    // the element-wise chains run on register pairs (v_pk_add / v_pk_fma / v_pk_mul: half the instructions; CG is even)
    typedef float la_f2 __attribute__((ext_vector_type(2)));
    la_f2 s2 = {0.f, 0.f};
    if (mode == 0) {
#pragma unroll
        for (int k = 0; k < CG; k += 2) s2 += la_f2{v[k], v[k + 1]};
        red[grp * TT + tok] = s2.x + s2.y;
        __syncthreads();
        const float mean = (red[tok] + red[TT + tok] + red[2 * TT + tok] + red[3 * TT + tok]) * (1.0f / C);
        const la_f2 m2 = {mean, mean};
        la_f2 q2 = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < CG; k += 2) {
            const la_f2 d2 = la_f2{v[k], v[k + 1]} - m2;
            v[k] = d2.x; v[k + 1] = d2.y;
            q2 += d2 * d2;
        }
        red[(4 + grp) * TT + tok] = q2.x + q2.y;
        __syncthreads();
        const float var = (red[4 * TT + tok] + red[5 * TT + tok] + red[6 * TT + tok] + red[7 * TT + tok]) * (1.0f / C);
        const float rstd = rsqrtf(var + eps);
        const la_f2 r2 = {rstd, rstd};
#pragma unroll
        for (int k = 0; k < CG; k += 2) {
            const la_f2 o2 = la_f2{v[k], v[k + 1]} * r2 * la_f2{g[grp * CG + k], g[grp * CG + k + 1]};
            xs[(grp * CG + k) * XP + tok] = o2.x;
            xs[(grp * CG + k + 1) * XP + tok] = o2.y;
        }
    } else {
#pragma unroll
        for (int k = 0; k < CG; k += 2) { const la_f2 d2 = {v[k], v[k + 1]}; s2 += d2 * d2; }
        red[grp * TT + tok] = s2.x + s2.y;
        __syncthreads();
        const float nrm = sqrtf(red[tok] + red[TT + tok] + red[2 * TT + tok] + red[3 * TT + tok]);
        const float f = sqrtf((float)C) / fmaxf(nrm, 1e-12f);
        const la_f2 f2 = {f, f};
#pragma unroll
        for (int k = 0; k < CG; k += 2) {
            const la_f2 o2 = la_f2{v[k], v[k + 1]} * f2 * la_f2{g[grp * CG + k], g[grp * CG + k + 1]};
            xs[(grp * CG + k) * XP + tok] = o2.x;
            xs[(grp * CG + k + 1) * XP + tok] = o2.y;
        }
    }
    __syncthreads();
}

// 32 x 64 projection tile of one head: acc[j][r] = (W_h xn)[row(r, lh)][j*32 + l31], W fragments in registers
template <int C>
__device__ __forceinline__ void project(const float (&wreg)[C / 2], const float* __restrict__ xs, int l31, int lh, f32x16 (&acc)[2]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < C / 2; ++ks) {
        const float b0 = xs[(2 * ks + lh) * XP + l31], b1 = xs[(2 * ks + lh) * XP + 32 + l31];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[ks], b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[ks], b1, acc[1], 0, 0, 0);
    }
}

// ------------------------------------------------------------------ pass 1: per (split, sequence); wave = head
// Everything is kept transposed so that the reductions run over registers and the products chain without LDS:
//   K^T[tok][d] = xn^T Wk^T        (operands swapped: rows = tokens on registers, columns = d on lanes)
//   softmax statistics per d       = per lane: max / sum over the 32 accumulator registers + one cross-half shuffle
//   M^T[c][d] += sum_tok xn[c][tok] p[tok][d]:  A = xn from LDS, B = the P registers as they stand (the contraction
//   index tok is walked in accumulator-row order); the online rescale factor is per d = per lane, one multiply per register.
template <int C, bool GN>
__global__ __launch_bounds__(NT, (C == 64 ? 2 : 1)) SDC_NO_DS_MERGE void la_blk_ctx(const LaArgs a) {
    constexpr int NCT = C / 32;                     // row tiles of M^T (channels)
    extern __shared__ float lds[];                  // C = 128 needs 67 KB: dynamic
    float* const xs = lds;                          // [C][XP]
    float* const red = lds + (C + HID) * XP;        // [8][TT]  (same offsets as pass 2's layout)
    float* const gcoef = red + 8 * TT;              // [2][C]  (GroupNorm-on-load form only)
    const int tid = threadIdx.x, lane = tid & 63, head = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int split = blockIdx.x, seq = blockIdx.y;
    const int o = seq / a.inner, i = seq - o * a.inner;
    const float* xseq = a.x + o * a.so + i * a.si;
    const float* rseq = (GN && a.gn_res) ? a.gn_res + o * a.so + i * a.si : nullptr;
    if (GN) gn_coef_fill<C>(a.gn_stats, a.gn_gamma, a.gn_beta, a.gn_G, o, gcoef, tid);

    float wreg[C / 2];                              // Wk_h[d = l31][c = 2ks + lh]
#pragma unroll
    for (int ks = 0; ks < C / 2; ++ks) wreg[ks] = a.wqkv[(int64_t)(2 * ks + lh) * (3 * HID) + HID + head * 32 + l31];

    f32x16 macc[NCT];                               // M^T tile t: rows c = t*32 + crow, columns d = l31
#pragma unroll
    for (int t = 0; t < NCT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) macc[t][r] = 0.f;
    float mrun = -INFINITY, psum = 0.f;             // per d (this lane; the two half-waves hold different token rows)

    const int t0 = split * a.tiles_per_split;
    const int t1 = min(t0 + a.tiles_per_split, a.ntiles);
    float xv[C / 4];
    if (t0 < t1) fetch_tile<C>(xseq + (int64_t)t0 * TT, a.sc, tid, xv);
    if (GN) __syncthreads();                        // coefficient table complete
    for (int tile = t0; tile < t1; ++tile) {
        if (GN) {
            // h once, here: pass 2 reads it back (from the block's own output buffer, tile by tile in place) instead of paying
            // the SiLUs a second time -- measured with both passes normalising on load: +4.4 ms in these kernels for 3.5 ms
            // of sdc_gn_apply saved at C4
            gn_apply_tile<C>(xv, rseq ? rseq + (int64_t)tile * TT : nullptr, a.sc, gcoef, tid);
            float* hb = a.gn_hout + o * a.so + i * a.si + (int64_t)tile * TT;
            constexpr int CG = C / 4;
            la_gptr hp = la_uni(hb + (int64_t)(__builtin_amdgcn_readfirstlane(tid >> 6) * CG) * a.sc);
            uint32_t hoff = (uint32_t)(tid & 63) * 4u;
#pragma unroll
            for (int k = 0; k < CG; ++k) {
                asm volatile("" : "+v"(hoff));
                la_st(hp, hoff, xv[k]);
                hp += a.sc;
                asm volatile("" : "+s"(hp));
            }
        }
        norm_tile<C>(xv, a.g_pre, a.pre_mode, a.eps, xs, nullptr, red, tid);
        if (tile + 1 < t1) fetch_tile<C>(xseq + (int64_t)(tile + 1) * TT, a.sc, tid, xv);
        // K^T tiles j = 0, 1: rows tok = j*32 + crow(r, lh), columns d = l31
        f32x16 kacc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { kacc[0][r] = 0.f; kacc[1][r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < C / 2; ++ks) {
            const float a0 = xs[(2 * ks + lh) * XP + l31], a1 = xs[(2 * ks + lh) * XP + 32 + l31];
            kacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, wreg[ks], kacc[0], 0, 0, 0);
            kacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, wreg[ks], kacc[1], 0, 0, 0);
        }
        // online softmax over tokens, per d
        float tmax = kacc[0][0];
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = la_max3(tmax, kacc[0][r], kacc[1][r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(mrun, tmax);
        const float f = __expf(mrun - mnew);
        mrun = mnew;
        float ps = 0.f;
        // exp(k - m) as exp2(k log2e - m log2e): one multiply-add in front of the v_exp instead of a subtraction and a multiplication
        const float nml = -mnew * LOG2E;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            kacc[0][r] = __builtin_amdgcn_exp2f(fmaf(kacc[0][r], LOG2E, nml));
            kacc[1][r] = __builtin_amdgcn_exp2f(fmaf(kacc[1][r], LOG2E, nml));
            ps += kacc[0][r] + kacc[1][r];
        }
        psum = psum * f + ps;
        // the running maximum settles after a few tiles: when no lane's moved, f is exactly 1 everywhere and the rescaling of the
        // NCT x 16 accumulator registers (a multiplication by 1: the same bits) is skipped -- a wave-uniform branch
        if (__builtin_amdgcn_ballot_w64(f != 1.0f)) {
#pragma unroll
            for (int t = 0; t < NCT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) macc[t][r] *= f;
        }
        // M^T[c][d] += sum_tok xn[c][tok] p[tok][d]; step (j, r) covers tokens j*32 + crow(r, 0) | crow(r, 1)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tok = j * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
                for (int t = 0; t < NCT; ++t)
                    macc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[(t * 32 + l31) * XP + tok], kacc[j][r], macc[t], 0, 0, 0);
            }
        __syncthreads();                             // xs is rewritten by the next tile
    }
    // partial result of this split: M[32][C], m[32], s[32]
    float* pp = a.part + (((int64_t)seq * a.nsplit + split) * 4 + head) * (32 * (C + 2));
#pragma unroll
    for (int t = 0; t < NCT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) pp[l31 * C + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] = macc[t][r];
    const float stot = psum + __shfl_xor(psum, 32, 64);
    if (lh == 0) { pp[32 * C + l31] = mrun; pp[32 * C + 32 + l31] = stot; }
}

// ------------------------------------------------------------------ mid: per (sequence, head)
template <int C>
__global__ __launch_bounds__(NT) void la_blk_mid(const LaArgs a) {
    __shared__ float Ms[32][C + 1];
    __shared__ float cs[32][33];
    __shared__ float fsp[8][32];                    // per split: exp(m_s - m) ; nsplit <= 8
    __shared__ float rs[32];
    const int tid = threadIdx.x;
    const int head = blockIdx.x, seq = blockIdx.y;
    const float* pb = a.part + ((int64_t)seq * a.nsplit * 4 + head) * (32 * (C + 2));
    const int64_t sstride = (int64_t)4 * 32 * (C + 2);
    if (tid < 32) {
        float m = -INFINITY;
        for (int s = 0; s < a.nsplit; ++s) m = fmaxf(m, pb[s * sstride + 32 * C + tid]);
        float tot = 0.f;
        for (int s = 0; s < a.nsplit; ++s) {
            const float f = __expf(pb[s * sstride + 32 * C + tid] - m);
            fsp[s][tid] = f;
            tot += f * pb[s * sstride + 32 * C + 32 + tid];
        }
        rs[tid] = 1.0f / tot;
    }
    __syncthreads();
    for (int e = tid; e < 32 * C; e += NT) {
        const int d = e / C, c = e - d * C;
        float v = 0.f;
        for (int s = 0; s < a.nsplit; ++s) v += fsp[s][d] * pb[s * sstride + e];
        Ms[d][c] = v;
    }
    __syncthreads();
    // ctx[d][e] = sum_c M[d][c] Wv[e][c] / rowsum[d]
    for (int q = tid; q < 32 * 32; q += NT) {
        const int d = q >> 5, e = q & 31;
        float v = 0.f;
        for (int c = 0; c < C; ++c) v += Ms[d][c] * a.wqkv[(int64_t)c * (3 * HID) + 2 * HID + head * 32 + e];
        cs[d][e] = v * rs[d];
    }
    __syncthreads();
    // Tt[h*32 + d][co] = sum_e Wo[(h*32 + e)][co] ctx[d][e]
    float* tb = a.tt + (int64_t)seq * HID * C + (int64_t)head * 32 * C;
    for (int q = tid; q < 32 * C; q += NT) {
        const int d = q / C, co = q - d * C;
        float v = 0.f;
#pragma unroll 8
        for (int e = 0; e < 32; ++e) v += a.wo[(int64_t)(head * 32 + e) * C + co] * cs[d][e];
        tb[d * C + co] = v;
    }
}

// ------------------------------------------------------------------ pass 2: per (tile group, sequence)
template <int C>
__global__ __launch_bounds__(NT, (C == 64 ? 2 : 1)) SDC_NO_DS_MERGE void la_blk_out(const LaArgs a) {
    constexpr int NRT = C / 32;                     // row tiles of y (channels)
    constexpr int TPW = NRT / 2;                    // y tiles per wave (one row tile, TPW column tiles)
    extern __shared__ float lds[];
    float* const xs = lds;                          // [C][XP]
    float* const qs = lds + C * XP;                 // [128][XP]
    float* const red = qs + HID * XP;               // [8][TT]
    float* const xr = red + 8 * TT;                 // [C][XP] raw x tile (residual)
    float* const bg = xr + C * XP;                  // [2][C] bias | post gain
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int seq = blockIdx.y;
    const int o = seq / a.inner, i = seq - o * a.inner;
    const float* xseq = a.x + o * a.so + i * a.si;
    float* yseq = a.y + o * a.so + i * a.si;
    const int rt = (NRT == 2) ? (wave & 1) : wave;  // this wave's row tile of y
    const int ct0 = (NRT == 2) ? (wave >> 1) : 0;   // its first column tile

    float wreg[C / 2];                              // Wq_h[d = l31][c = 2ks + lh], head = wave
#pragma unroll
    for (int ks = 0; ks < C / 2; ++ks) wreg[ks] = a.wqkv[(int64_t)(2 * ks + lh) * (3 * HID) + wave * 32 + l31];
    float treg[HID / 2];                            // T[co = rt*32 + l31][hd = 2ks + lh]
    const float* tb = a.tt + (int64_t)seq * HID * C;
#pragma unroll
    for (int ks = 0; ks < HID / 2; ++ks) treg[ks] = tb[(2 * ks + lh) * C + rt * 32 + l31];
    for (int c = tid; c < C; c += NT) {
        bg[c] = a.bo ? a.bo[c] : 0.f;
        bg[C + c] = a.g_post ? a.g_post[c] : 1.f;
    }

    const int t0 = blockIdx.x * a.tiles_per_blk;
    const int t1 = min(t0 + a.tiles_per_blk, a.ntiles);
    float xv[C / 4];
    if (t0 < t1) fetch_tile<C>(xseq + (int64_t)t0 * TT, a.sc, tid, xv);
    for (int tile = t0; tile < t1; ++tile) {
        norm_tile<C>(xv, a.g_pre, a.pre_mode, a.eps, xs, xr, red, tid);
        if (tile + 1 < t1) fetch_tile<C>(xseq + (int64_t)(tile + 1) * TT, a.sc, tid, xv);
        f32x16 qacc[2];
        project<C>(wreg, xs, l31, lh, qacc);
        // softmax over d (the 32 rows of the head) per token, times dim_head^-0.5
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float m = fmaxf(qacc[j][0], qacc[j][1]);
#pragma unroll
            for (int r = 2; r < 16; r += 2) m = la_max3(m, qacc[j][r], qacc[j][r + 1]);
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            typedef float la_f2 __attribute__((ext_vector_type(2)));
            const float nml = -m * LOG2E;
            const la_f2 l2 = {LOG2E, LOG2E}, n2 = {nml, nml};
            la_f2 s2 = {0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {      // (register pairs: packed multiply-add in front of the two v_exp, packed sum)
                const la_f2 e2 = la_f2{qacc[j][r], qacc[j][r + 1]} * l2 + n2;
                const la_f2 p2 = {__builtin_amdgcn_exp2f(e2.x), __builtin_amdgcn_exp2f(e2.y)};
                qacc[j][r] = p2.x; qacc[j][r + 1] = p2.y;
                s2 += p2;
            }
            float s = s2.x + s2.y;
            s += __shfl_xor(s, 32, 64);
            const float f = 0.17677669529663687f * __builtin_amdgcn_rcpf(s);      // (reciprocal instruction: the IEEE division is ten more)
            const la_f2 f2 = {f, f};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const la_f2 o2 = la_f2{qacc[j][r], qacc[j][r + 1]} * f2;
                qs[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * XP + j * 32 + l31] = o2.x;
                qs[(wave * 32 + ((r + 1) & 3) + 8 * ((r + 1) >> 2) + 4 * lh) * XP + j * 32 + l31] = o2.y;
            }
        }
        __syncthreads();
        // y[co][tok] = sum_hd T[co][hd] q[hd][tok]
        f32x16 yacc[TPW];
#pragma unroll
        for (int u = 0; u < TPW; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) yacc[u][r] = bg[rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
#pragma unroll
        for (int ks = 0; ks < HID / 2; ++ks)
#pragma unroll
            for (int u = 0; u < TPW; ++u)
                yacc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(treg[ks], qs[(2 * ks + lh) * XP + (ct0 + u) * 32 + l31], yacc[u], 0, 0, 0);
        // channel norm over the C rows of each token column: 16 registers x 2 half-waves x NRT row tiles (waves)
        if (a.post_mode >= 0) {
            float st[TPW];
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                float s = 0.f;
                if (a.post_mode == 0) {              // (uniform branch: a per-element select computed both forms, 16 v_cndmask per tile)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s += yacc[u][r];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s += yacc[u][r] * yacc[u][r];
                }
                s += __shfl_xor(s, 32, 64);
                if (lh == 0) red[rt * TT + (ct0 + u) * 32 + l31] = s;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < NRT; ++q) s += red[q * TT + (ct0 + u) * 32 + l31];
                st[u] = s;
            }
            if (a.post_mode == 0) {
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    typedef float la_f2 __attribute__((ext_vector_type(2)));
                    const float mean = st[u] * (1.0f / C);
                    const la_f2 m2 = {mean, mean};
                    la_f2 q2 = {0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const la_f2 d2 = la_f2{yacc[u][r], yacc[u][r + 1]} - m2;
                        yacc[u][r] = d2.x; yacc[u][r + 1] = d2.y;
                        q2 += d2 * d2;
                    }
                    float q = q2.x + q2.y;
                    q += __shfl_xor(q, 32, 64);
                    if (lh == 0) red[(4 + rt) * TT + (ct0 + u) * 32 + l31] = q;
                }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    float q = 0.f;
#pragma unroll
                    for (int w = 0; w < NRT; ++w) q += red[(4 + w) * TT + (ct0 + u) * 32 + l31];
                    const float rstd = rsqrtf(q * (1.0f / C) + a.eps);
                    typedef float la_f2 __attribute__((ext_vector_type(2)));
                    const la_f2 r2 = {rstd, rstd};
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const la_f2 o2 = la_f2{yacc[u][r], yacc[u][r + 1]} * r2 *
                                         la_f2{bg[C + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh], bg[C + rt * 32 + ((r + 1) & 3) + 8 * ((r + 1) >> 2) + 4 * lh]};
                        yacc[u][r] = o2.x; yacc[u][r + 1] = o2.y;
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    const float f = sqrtf((float)C) / fmaxf(sqrtf(st[u]), 1e-12f);
#pragma unroll
                    for (int r = 0; r < 16; ++r) yacc[u][r] = yacc[u][r] * f * bg[C + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
                }
            }
        }
        // + x, store (tokens on lanes: coalesced rows)
        // (rows co = rt*32 + (r & 3) + 8 (r >> 2) from a scalar base, the half-wave's 4 rows and the column as one lane offset:
        // host check `4 sc` elements < 2^30)
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const int col = (ct0 + u) * 32 + l31;
            uint32_t loff = (uint32_t)((int64_t)(4 * lh) * a.sc + col) * 4u;
            // running scalar row base: rows rt*32 + (r & 3) + 8 (r >> 2) are +1, +1, +1, +5 channel strides apart (formed per row from
            // (cob, tile) the base cost ~6 scalar instructions per store, 190 per tile at width 64 -- a wave issues one instruction
            // per four cycles whatever its kind)
            la_gptr yrow = la_uni(yseq + (int64_t)(rt * 32) * a.sc + (int64_t)tile * TT);
            const float* xrow = xr + (rt * 32 + 4 * lh) * XP + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                asm volatile("" : "+v"(loff));
                la_st(yrow, loff, yacc[u][r] + xrow[((r & 3) + 8 * (r >> 2)) * XP]);
                yrow += ((r & 3) == 3 ? 5 : 1) * a.sc;
                asm volatile("" : "+s"(yrow));
            }
        }
        __syncthreads();                             // xs / qs / red are rewritten by the next tile
    }
}

// token splits of pass 1: a function of the sequence length only, so a trajectory's result does not depend on how
// many others share the launch (batch-invariant summation order)
int pick_nsplit(int64_t /*nseq*/, int ntiles) {
    int ns = ntiles / 8;
    if (ns > 8) ns = 8;
    return ns < 1 ? 1 : ns;
}

}  // namespace

extern "C" size_t sdc_linattn_block_bytes(int outer, int inner, int C, int64_t n) {
    if (outer <= 0 || inner <= 0 || n <= 0 || (C != 64 && C != 128)) return 0;
    const int64_t nseq = (int64_t)outer * inner;
    const int ns = pick_nsplit(nseq, (int)(n / TT));
    return sizeof(float) * (size_t)(nseq * ns * 4 * 32 * (C + 2) + nseq * HID * C);
}

namespace {
int linattn_block_impl(const float* x, const float* gn_stats, const float* gn_gamma, const float* gn_beta, int gn_G,
                       const float* gn_res, const float* g_pre, const float* wqkv, const float* wo, const float* bo,
                       const float* g_post, float* work, float* y, int outer, int inner, int C, int64_t n,
                       int64_t so, int64_t sc, int64_t si, int pre_mode, int post_mode, float eps, void* stream) {
    SDC_REQUIRE(x && g_pre && wqkv && wo && work && y, SDC_ENULL, "sdc_linattn_block: null pointer");
    SDC_REQUIRE(C == 64 || C == 128, SDC_EINVAL, "sdc_linattn_block: dim must be 64 or 128 (got %d)", C);
    SDC_REQUIRE(outer > 0 && inner > 0 && n > 0 && n % TT == 0, SDC_EINVAL, "sdc_linattn_block: tokens must be a multiple of 64");
    SDC_REQUIRE((pre_mode == 0 || pre_mode == 1) && post_mode >= -1 && post_mode <= 1, SDC_EINVAL, "sdc_linattn_block: bad norm mode");
    SDC_REQUIRE(post_mode < 0 || g_post, SDC_ENULL, "sdc_linattn_block: post norm needs its gain");
    const int64_t nseq = (int64_t)outer * inner;
    SDC_REQUIRE(nseq < 65536, SDC_EINVAL, "sdc_linattn_block: outer*inner must be < 65536");
    SDC_REQUIRE(sc > 0 && sc < (1ll << 27), SDC_EINVAL, "sdc_linattn_block: channel stride must stay below 2^27 elements (32-bit lane offsets)");
    LaArgs a;
    a.x = x; a.g_pre = g_pre; a.wqkv = wqkv; a.wo = wo; a.bo = bo; a.g_post = g_post; a.y = y;
    a.gn_stats = gn_stats; a.gn_gamma = gn_gamma; a.gn_beta = gn_beta; a.gn_res = gn_res; a.gn_G = gn_G; a.gn_hout = nullptr;
    a.inner = inner; a.ntiles = (int)(n / TT);
    a.nsplit = pick_nsplit(nseq, a.ntiles);
    a.tiles_per_split = (a.ntiles + a.nsplit - 1) / a.nsplit;
    a.nsplit = (a.ntiles + a.tiles_per_split - 1) / a.tiles_per_split;     // no empty splits
    {
        // keep the layout sdc_linattn_block_bytes promised: partials first (sized for the unclamped split count)
        const int ns0 = pick_nsplit(nseq, a.ntiles);
        a.part = work;
        a.tt = work + nseq * ns0 * 4 * 32 * (C + 2);
    }
    int tpb = (int)((nseq * a.ntiles + 1023) / 1024);
    a.tiles_per_blk = tpb < 1 ? 1 : (tpb > 8 ? 8 : tpb);
    a.pre_mode = pre_mode; a.post_mode = post_mode; a.eps = eps;
    a.so = so; a.sc = sc; a.si = si;
    hipStream_t s = sdc::as_stream(stream);
    const dim3 g1((unsigned)a.nsplit, (unsigned)nseq), gm(4, (unsigned)nseq),
        g2((unsigned)((a.ntiles + a.tiles_per_blk - 1) / a.tiles_per_blk), (unsigned)nseq);
    const size_t ldsb = sizeof(float) * (size_t)((C + HID) * XP + 8 * TT + 2 * C);  // pass 1 (+ GroupNorm coefficients)
    const size_t ldsb2 = sizeof(float) * (size_t)((C + HID) * XP + 8 * TT + C * XP + 2 * C);          // pass 2: + raw tile, bias, gain
    static std::atomic<uint64_t> attr0{0}, attr1{0}, attr2{0};
    static std::atomic<uint64_t> attr3{0};
    SDC_LDS_OPTIN(attr0, (la_blk_ctx<128, false>), 96 * 1024, "sdc_linattn_block");
    SDC_LDS_OPTIN(attr1, la_blk_out<128>, 128 * 1024, "sdc_linattn_block");
    SDC_LDS_OPTIN(attr2, la_blk_out<64>, 96 * 1024, "sdc_linattn_block");
    SDC_LDS_OPTIN(attr3, (la_blk_ctx<128, true>), 96 * 1024, "sdc_linattn_block");
    const bool gn = gn_stats != nullptr;
    LaArgs a2 = a;                                  // pass 2 of the GroupNorm-on-load form: x = the h pass 1 left in y
    if (gn) { a.gn_hout = y; a2.x = y; a2.gn_stats = nullptr; }
    if (C == 64) {
        if (gn) hipLaunchKernelGGL((la_blk_ctx<64, true>), g1, dim3(NT), ldsb, s, a);
        else hipLaunchKernelGGL((la_blk_ctx<64, false>), g1, dim3(NT), ldsb, s, a);
        hipLaunchKernelGGL(la_blk_mid<64>, gm, dim3(NT), 0, s, a);
        hipLaunchKernelGGL(la_blk_out<64>, g2, dim3(NT), ldsb2, s, a2);
    } else {
        if (gn) hipLaunchKernelGGL((la_blk_ctx<128, true>), g1, dim3(NT), ldsb, s, a);
        else hipLaunchKernelGGL((la_blk_ctx<128, false>), g1, dim3(NT), ldsb, s, a);
        hipLaunchKernelGGL(la_blk_mid<128>, gm, dim3(NT), 0, s, a);
        hipLaunchKernelGGL(la_blk_out<128>, g2, dim3(NT), ldsb2, s, a2);
    }
    return sdc::check_launch("sdc_linattn_block");
}
}  // namespace

extern "C" int sdc_linattn_block(const float* x, const float* g_pre, const float* wqkv, const float* wo, const float* bo,
                                 const float* g_post, float* work, float* y, int outer, int inner, int C, int64_t n,
                                 int64_t so, int64_t sc, int64_t si, int pre_mode, int post_mode, float eps, void* stream) {
    return linattn_block_impl(x, nullptr, nullptr, nullptr, 0, nullptr, g_pre, wqkv, wo, bo, g_post, work, y, outer, inner, C, n, so, sc,
                              si, pre_mode, post_mode, eps, stream);
}

extern "C" int sdc_linattn_block_gn(const float* x_raw, const float* gn_stats, const float* gn_gamma, const float* gn_beta, int gn_G,
                                    const float* gn_residual, const float* g_pre, const float* wqkv, const float* wo, const float* bo,
                                    const float* g_post, float* work, float* y, int outer, int inner, int C, int64_t n,
                                    int64_t so, int64_t sc, int64_t si, int pre_mode, int post_mode, float eps, void* stream) {
    SDC_REQUIRE(gn_stats && gn_gamma && gn_beta, SDC_ENULL, "sdc_linattn_block_gn: null GroupNorm pointer");
    SDC_REQUIRE(gn_G > 0 && (C == 64 || C == 128) && C % gn_G == 0, SDC_EINVAL, "sdc_linattn_block_gn: groups must divide the channels");
    return linattn_block_impl(x_raw, gn_stats, gn_gamma, gn_beta, gn_G, gn_residual, g_pre, wqkv, wo, bo, g_post, work, y, outer, inner,
                              C, n, so, sc, si, pre_mode, post_mode, eps, stream);
}
