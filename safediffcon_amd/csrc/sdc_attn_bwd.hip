// Backward of the attention cores (fine-tuning path), heads x dim_head 32, fp32:
//   * sdc_attn_bwd    softmax attention (mid Attention 1D/model/unet.py:224-258, smoke mid spatial attention and temporal
//                     attention with rotary + relative-position bias, conv3d.py:277-353): dq, dk, dv and the bias gradient
//   * sdc_linattn_bwd linear attention (1D/model/unet.py:182-222, conv3d.py:232-258): dq, dk, dv
// Same tensor conventions as the forward cores (csrc/sdc_attn.hip): q, k, v are channel ranges of the channel-major conv
// output, addressed through strides; the gradients are written in the same layout.  These are correctness-first fp32 VALU
// kernels: the cores hold ~1 % of a U-Net's FLOPs, their projections run on the MFMA conv kernels.
#include "sdc_common.h"

namespace {

constexpr int NT = 256;
constexpr int DH = 32;
constexpr float SCALE = 0.17677669529663687f;       // dim_head^-0.5

// ------------------------------------------------------------------------------------------------ softmax attention
// P = softmax(q' k'^T + bias), O = P V with q' = rot(q scale), k' = rot(k).  Given dO:
//   D_i = sum_j P_ij dP_ij (= dO_i . O_i),  dP_ij = dO_i . v_j,  dS_ij = P_ij (dP_ij - D_i)
//   dq'_i = sum_j dS_ij k'_j,  dk'_j = sum_i dS_ij q'_i,  dv_j = sum_i P_ij dO_i,  dbias[h][i][j] = sum over sequences dS_ij
// A workgroup owns nseq sequences of one head (one thread per (sequence, token), nseq * ntok <= 256), q', k', v, dO in LDS.
// Phase A: thread = query i (row statistics, D_i, dq_i);  phase B: thread = key j (dk_j, dv_j, its column of dS).  No
// atomics: every output element has one owner.  Workgroups loop over sequence groups (fixed assignment) so that the bias
// gradient needs one partial table per workgroup; the partials are summed in a fixed order afterwards.
struct AttnBwdArgs {
    const float* qkv; const float* dout; const float* rot; const float* bias;
    float* dqkv; float* dbias_part;
    int outer, inner, heads, ntok, nseq, tok_contig, ngrp, nslots;
    int64_t so, sc, si, st, oso, osc, osi, ost;
    int ls, ld, lj;   // LDS strides for (seq, d, tok)
};

__global__ __launch_bounds__(NT) void attn_bwd_kernel(const AttnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ntok = a.ntok, nseq = a.nseq;
    const int ksz = a.tok_contig ? nseq * a.ls : DH * a.ld;
    float* Qs = lds;
    float* Ks = lds + ksz;
    float* Vs = lds + 2 * ksz;
    float* Gs = lds + 3 * ksz;
    float* Mx = lds + 4 * ksz;            // [256] row maxima, 1 / row sums, D_i
    float* Li = Mx + NT;
    float* Dd = Li + NT;
    const int head = blockIdx.x % a.heads;
    const int slot = blockIdx.x / a.heads;
    const int nseq_tot = a.outer * a.inner;
    const int tid = threadIdx.x;
    const int nthr = nseq * ntok;
    int sq, ti;
    if (a.tok_contig) { ti = tid % ntok; sq = tid / ntok; }
    else { sq = tid % nseq; ti = tid / nseq; }
    const bool owner = tid < nthr;
    const bool want_bias = a.bias != nullptr && a.dbias_part != nullptr;      // (ntok <= 32, host check)
    float* Db = Dd + NT;                  // [nseq][query][key] sums of dS over this workgroup's groups (bias gradient)
    if (want_bias) {
        for (int e = tid; e < nseq * ntok * ntok; e += NT) Db[e] = 0.0f;
    }

    auto seq_base = [&](int s) -> int64_t { const int o = s / a.inner, i = s - o * a.inner; return o * a.so + i * a.si; };
    auto seq_obase = [&](int s) -> int64_t { const int o = s / a.inner, i = s - o * a.inner; return o * a.oso + i * a.osi; };
    const int64_t qoff = (int64_t)(head * DH) * a.sc;
    const int64_t koff = (int64_t)(a.heads * DH + head * DH) * a.sc;
    const int64_t voff = (int64_t)(2 * a.heads * DH + head * DH) * a.sc;
    const int64_t goff = (int64_t)(head * DH) * a.osc;

    for (int grp = slot; grp < a.ngrp; grp += a.nslots) {
        const int seq0 = grp * nseq;
        // ---- stage q (scaled), k, v, dO
        const int total = nseq * DH * ntok;
        for (int e = tid; e < total; e += NT) {
            int s_, d, j;
            if (a.tok_contig) { j = e % ntok; d = (e / ntok) % DH; s_ = e / (ntok * DH); }
            else { s_ = e % nseq; j = (e / nseq) % ntok; d = e / (nseq * ntok); }
            float qv = 0.f, kv = 0.f, vv = 0.f, gv = 0.f;
            if (seq0 + s_ < nseq_tot) {
                const int64_t g = seq_base(seq0 + s_) + (int64_t)d * a.sc + (int64_t)j * a.st;
                qv = a.qkv[g + qoff] * SCALE;
                kv = a.qkv[g + koff];
                vv = a.qkv[g + voff];
                gv = a.dout[seq_obase(seq0 + s_) + goff + (int64_t)d * a.osc + (int64_t)j * a.ost];
            }
            const int li = s_ * a.ls + d * a.ld + j * a.lj;
            Qs[li] = qv; Ks[li] = kv; Vs[li] = vv; Gs[li] = gv;
        }
        __syncthreads();
        if (a.rot) {
            // rotate q and k pairs in place: (x0, x1) -> (x0 c - x1 s, x1 c + x0 s), angle = tok * freq[pair]
            const int npair = nseq * (DH / 2) * ntok;
            for (int e = tid; e < npair; e += NT) {
                int s_, m, j;
                if (a.tok_contig) { j = e % ntok; m = (e / ntok) % (DH / 2); s_ = e / (ntok * (DH / 2)); }
                else { s_ = e % nseq; j = (e / nseq) % ntok; m = e / (nseq * ntok); }
                const float c = a.rot[(j * (DH / 2) + m) * 2], s = a.rot[(j * (DH / 2) + m) * 2 + 1];
                const int l0 = s_ * a.ls + (2 * m) * a.ld + j * a.lj, l1 = l0 + a.ld;
                float x0 = Ks[l0], x1 = Ks[l1];
                Ks[l0] = x0 * c - x1 * s; Ks[l1] = x1 * c + x0 * s;
                x0 = Qs[l0]; x1 = Qs[l1];
                Qs[l0] = x0 * c - x1 * s; Qs[l1] = x1 * c + x0 * s;
            }
            __syncthreads();
        }
        const bool live = owner && seq0 + sq < nseq_tot;
        const float* Kq = Ks + sq * a.ls;
        const float* Vq = Vs + sq * a.ls;
        const float* Qq = Qs + sq * a.ls;
        const float* Gq = Gs + sq * a.ls;
        // ---- phase A: this thread's query row
        if (live) {
            float q[DH], g[DH];
#pragma unroll
            for (int d = 0; d < DH; ++d) { q[d] = Qq[d * a.ld + ti * a.lj]; g[d] = Gq[d * a.ld + ti * a.lj]; }
            const float* brow = a.bias ? a.bias + ((int64_t)head * ntok + ti) * ntok : nullptr;
            float mx = -INFINITY;
            for (int j = 0; j < ntok; ++j) {
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < DH; ++d) s += q[d] * Kq[d * a.ld + j * a.lj];
                if (brow) s += brow[j];
                mx = fmaxf(mx, s);
            }
            float l = 0.f, dn = 0.f, a1[DH], a2[DH];
#pragma unroll
            for (int d = 0; d < DH; ++d) { a1[d] = 0.f; a2[d] = 0.f; }
            for (int j = 0; j < ntok; ++j) {
                float s = 0.f, dp = 0.f;
                float kk[DH];
#pragma unroll
                for (int d = 0; d < DH; ++d) { kk[d] = Kq[d * a.ld + j * a.lj]; s += q[d] * kk[d]; dp += g[d] * Vq[d * a.ld + j * a.lj]; }
                if (brow) s += brow[j];
                const float e = sdc::softmax_exp(s - mx);
                l += e;
                dn += e * dp;
                const float edp = e * dp;
#pragma unroll
                for (int d = 0; d < DH; ++d) { a1[d] += edp * kk[d]; a2[d] += e * kk[d]; }
            }
            const float inv = 1.0f / l, D = dn * inv;
            Mx[tid] = mx; Li[tid] = inv; Dd[tid] = D;
            float dq[DH];
#pragma unroll
            for (int d = 0; d < DH; ++d) dq[d] = (a1[d] - D * a2[d]) * inv;
            if (a.rot) {
#pragma unroll
                for (int m = 0; m < DH / 2; ++m) {       // transposed rotation
                    const float c = a.rot[(ti * (DH / 2) + m) * 2], s = a.rot[(ti * (DH / 2) + m) * 2 + 1];
                    const float y0 = dq[2 * m], y1 = dq[2 * m + 1];
                    dq[2 * m] = y0 * c + y1 * s;
                    dq[2 * m + 1] = y1 * c - y0 * s;
                }
            }
            float* ob = a.dqkv + seq_base(seq0 + sq) + qoff + (int64_t)ti * a.st;
#pragma unroll
            for (int d = 0; d < DH; ++d) ob[(int64_t)d * a.sc] = dq[d] * SCALE;
        }
        __syncthreads();
        // ---- phase B: this thread's key column
        if (live) {
            float k[DH], v[DH], dk[DH], dv[DH];
#pragma unroll
            for (int d = 0; d < DH; ++d) { k[d] = Kq[d * a.ld + ti * a.lj]; v[d] = Vq[d * a.ld + ti * a.lj]; dk[d] = 0.f; dv[d] = 0.f; }
            const int sbase = a.tok_contig ? sq * ntok : sq;            // tid of (sq, token i) = sbase + i * sstep
            const int sstep = a.tok_contig ? 1 : nseq;
            float* dbcol = Db + sq * ntok * ntok + ti;                     // this thread's column: element i at dbcol[i * ntok]
            for (int i = 0; i < ntok; ++i) {
                float s = 0.f, dp = 0.f, qq[DH], gg[DH];
#pragma unroll
                for (int d = 0; d < DH; ++d) { qq[d] = Qq[d * a.ld + i * a.lj]; gg[d] = Gq[d * a.ld + i * a.lj]; s += qq[d] * k[d]; dp += gg[d] * v[d]; }
                if (a.bias) s += a.bias[((int64_t)head * ntok + i) * ntok + ti];
                const int r = sbase + i * sstep;
                const float p = sdc::softmax_exp(s - Mx[r]) * Li[r];
                const float ds = p * (dp - Dd[r]);
#pragma unroll
                for (int d = 0; d < DH; ++d) { dk[d] += ds * qq[d]; dv[d] += p * gg[d]; }
                if (want_bias) dbcol[i * ntok] += ds;
            }
            if (a.rot) {
#pragma unroll
                for (int m = 0; m < DH / 2; ++m) {
                    const float c = a.rot[(ti * (DH / 2) + m) * 2], s = a.rot[(ti * (DH / 2) + m) * 2 + 1];
                    const float y0 = dk[2 * m], y1 = dk[2 * m + 1];
                    dk[2 * m] = y0 * c + y1 * s;
                    dk[2 * m + 1] = y1 * c - y0 * s;
                }
            }
            float* kb = a.dqkv + seq_base(seq0 + sq) + koff + (int64_t)ti * a.st;
            float* vb = a.dqkv + seq_base(seq0 + sq) + voff + (int64_t)ti * a.st;
#pragma unroll
            for (int d = 0; d < DH; ++d) { kb[(int64_t)d * a.sc] = dk[d]; vb[(int64_t)d * a.sc] = dv[d]; }
        }
        __syncthreads();
    }
    if (want_bias) {
        // Db[sq][i][j]: sum over this workgroup's sequences in a fixed order -> one partial table
        __syncthreads();
        float* out = a.dbias_part + ((int64_t)slot * a.heads + head) * ntok * ntok;
        for (int e = tid; e < ntok * ntok; e += NT) {
            float sacc = 0.f;
            for (int s_ = 0; s_ < nseq; ++s_) sacc += Db[s_ * ntok * ntok + e];
            out[e] = sacc;
        }
    }
}

// ------------------------------------------------------------------------------------------------ temporal attention, MFMA
// 32 frames x dim_head 32: every product of the backward is one 32 x 32 x 32 tile = 16 v_mfma_f32_32x32x2_f32.  A workgroup owns
// 8 adjacent pixels (32-byte runs of the channel-major tensors) of one head, each wave two of them.  Per (pixel, head):
//   orientation 1 (lanes = queries):  S^T = K' Q'^T,  dP^T = V dO^T  -> row statistics, D, dS^T in registers -> dQ' = dS K'
//                                     (the dS^T registers are the A operand as they stand)
//   orientation 2 (lanes = keys):     S = Q' K'^T,  dP = dO V^T  -> P, dS (statistics of orientation 1 through LDS)
//                                     -> dK' = dS^T Q',  dV = P^T dO
// 112 MFMAs per (pixel, head); q', k', v, dO live in LDS as [d][frame] images (pitch 33: conflict-free along either index); the
// gradients go back through the same images so that global stores are 32-byte runs.  Workgroups loop over pixel groups (fixed
// assignment): the bias gradient is accumulated in registers and leaves as one partial table per workgroup.
constexpr int TB_NS = 8;                  // pixels per workgroup
constexpr int TB_P = 33;                  // image pitch
constexpr int TB_IMG = 32 * TB_P;         // floats per [d][f] image

__global__ __launch_bounds__(NT) void tattn_bwd_kernel(const AttnBwdArgs a) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const img = lds;                                   // [4 tensors][8 pixels][32 d][33]
    float* const biasT = lds + 4 * TB_NS * TB_IMG;            // [key][query]
    float* const biasN = biasT + 32 * TB_P;                   // [query][key]
    float* const rotc = biasN + 32 * TB_P;                    // [frame][16]
    float* const rots = rotc + 32 * 16;
    float* const stat = rots + 32 * 16;                       // [wave][3][32]: row max, 1 / row sum, D
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int head = blockIdx.x % a.heads;
    const int slot = blockIdx.x / a.heads;
    auto IMG = [&](int t, int px) -> float* { return img + (t * TB_NS + px) * TB_IMG; };
    auto row_of = [&](int r) -> int { return 8 * (r >> 2) + 4 * lh + (r & 3); };

    for (int e = tid; e < 32 * 32; e += NT) {
        const int q = e >> 5, kk = e & 31;
        const float bv = a.bias ? a.bias[((int64_t)head * 32 + q) * 32 + kk] : 0.0f;
        biasT[kk * TB_P + q] = bv;
        biasN[q * TB_P + kk] = bv;
    }
    for (int e = tid; e < 32 * 16; e += NT) {
        rotc[e] = a.rot ? a.rot[e * 2] : 1.0f;
        rots[e] = a.rot ? a.rot[e * 2 + 1] : 0.0f;
    }
    float dbacc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dbacc[r] = 0.0f;
    float* const st_m = stat + wave * 96, * const st_l = st_m + 32, * const st_d = st_m + 64;

    for (int grp = slot; grp < a.ngrp; grp += a.nslots) {
        const int seq0 = grp * TB_NS;
        const int o = seq0 / a.inner, i0 = seq0 - o * a.inner;
        const float* qb = a.qkv + o * a.so + i0 + (int64_t)(head * DH) * a.sc;
        const float* kb = qb + (int64_t)(a.heads * DH) * a.sc;
        const float* vb = kb + (int64_t)(a.heads * DH) * a.sc;
        const float* gb = a.dout + o * a.oso + i0 + (int64_t)(head * DH) * a.osc;
        __syncthreads();                                      // the previous group's stores have read the images
        // ---- stage q (scaled), k, v, dO: element e = tid + 256 it -> (d = e >> 8, f = (e >> 3) & 31, pixel = e & 7)
#pragma unroll 4
        for (int it = 0; it < 32; ++it) {
            const int e = tid + it * NT;
            const int hw = e & 7, f = (e >> 3) & 31, d = e >> 8;
            const int64_t g = (int64_t)d * a.sc + (int64_t)f * a.st + hw;
            const int li = d * TB_P + f;
            IMG(0, hw)[li] = qb[g] * SCALE;
            IMG(1, hw)[li] = kb[g];
            IMG(2, hw)[li] = vb[g];
            IMG(3, hw)[li] = gb[(int64_t)d * a.osc + (int64_t)f * a.ost + hw];
        }
        __syncthreads();
        if (a.rot) {
            // rotate q and k pairs in place: (x0, x1) -> (x0 c - x1 s, x1 c + x0 s), angle = frame * freq[pair]
            for (int e = tid; e < TB_NS * 16 * 32; e += NT) {
                const int f = e & 31, m = (e >> 5) & 15, px = e >> 9;
                const float c = rotc[f * 16 + m], sn = rots[f * 16 + m];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float* p0 = IMG(t, px) + (2 * m) * TB_P + f;
                    const float x0 = p0[0], x1 = p0[TB_P];
                    p0[0] = x0 * c - x1 * sn;
                    p0[TB_P] = x1 * c + x0 * sn;
                }
            }
            __syncthreads();
        }
        // ---- every wave: its two pixels
        for (int pp = 0; pp < 2; ++pp) {
            const int px = wave * 2 + pp;
            float* Qi = IMG(0, px);
            float* Ki = IMG(1, px);
            float* Vi = IMG(2, px);
            float* Gi = IMG(3, px);
            // orientation 1: lanes = queries
            f32x16 sT, dpT;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sT[r] = 0.f; dpT[r] = 0.f; }
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {
                const int li = (2 * s2 + lh) * TB_P + l31;
                sT = __builtin_amdgcn_mfma_f32_32x32x2f32(Ki[li], Qi[li], sT, 0, 0, 0);
                dpT = __builtin_amdgcn_mfma_f32_32x32x2f32(Vi[li], Gi[li], dpT, 0, 0, 0);
            }
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sT[r] += biasT[row_of(r) * TB_P + l31]; mx = fmaxf(mx, sT[r]); }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sT[r] = sdc::softmax_exp(sT[r] - mx); sum += sT[r]; }
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.0f / sum;
            float D = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sT[r] *= inv; D += sT[r] * dpT[r]; }
            D += __shfl_xor(D, 32, 64);
            if (lh == 0) { st_m[l31] = mx; st_l[l31] = inv; st_d[l31] = D; }
#pragma unroll
            for (int r = 0; r < 16; ++r) { sT[r] = sT[r] * (dpT[r] - D); dbacc[r] += sT[r]; }        // dS^T[key][query]
            // dQ'[q][d] = sum_key dS[q][key] K'[key][d]
            f32x16 dq;
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[r] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) dq = __builtin_amdgcn_mfma_f32_32x32x2f32(sT[r], Ki[l31 * TB_P + row_of(r)], dq, 0, 0, 0);
            // orientation 2: lanes = keys
            f32x16 sN, dpN;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sN[r] = 0.f; dpN[r] = 0.f; }
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {
                const int li = (2 * s2 + lh) * TB_P + l31;
                sN = __builtin_amdgcn_mfma_f32_32x32x2f32(Qi[li], Ki[li], sN, 0, 0, 0);
                dpN = __builtin_amdgcn_mfma_f32_32x32x2f32(Gi[li], Vi[li], dpN, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = row_of(r);
                const float p = sdc::softmax_exp(sN[r] + biasN[q * TB_P + l31] - st_m[q]) * st_l[q];
                sN[r] = p;                                        // P[q][key]
                dpN[r] = p * (dpN[r] - st_d[q]);                  // dS[q][key]
            }
            f32x16 dk, dv;
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int li = l31 * TB_P + row_of(r);
                dk = __builtin_amdgcn_mfma_f32_32x32x2f32(dpN[r], Qi[li], dk, 0, 0, 0);
                dv = __builtin_amdgcn_mfma_f32_32x32x2f32(sN[r], Gi[li], dv, 0, 0, 0);
            }
            // transposed rotation of dq', dk' (pairs = adjacent lanes d, d ^ 1), the scale of q; results replace the images
            const int m = l31 >> 1;
            const bool odd = l31 & 1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = row_of(r);
                const float c = rotc[f * 16 + m], sn = rots[f * 16 + m];
                const float pq = __shfl_xor(dq[r], 1, 64), pk = __shfl_xor(dk[r], 1, 64);
                const float rq = dq[r] * c + (odd ? -pq : pq) * sn;
                const float rk = dk[r] * c + (odd ? -pk : pk) * sn;
                Qi[l31 * TB_P + f] = rq * SCALE;
                Ki[l31 * TB_P + f] = rk;
                Vi[l31 * TB_P + f] = dv[r];
            }
        }
        __syncthreads();
        // ---- gradients back to the channel-major tensor
        float* dqb = a.dqkv + o * a.so + i0 + (int64_t)(head * DH) * a.sc;
        float* dkb = dqb + (int64_t)(a.heads * DH) * a.sc;
        float* dvb = dkb + (int64_t)(a.heads * DH) * a.sc;
#pragma unroll 4
        for (int it = 0; it < 32; ++it) {
            const int e = tid + it * NT;
            const int hw = e & 7, f = (e >> 3) & 31, d = e >> 8;
            const int64_t g = (int64_t)d * a.sc + (int64_t)f * a.st + hw;
            const int li = d * TB_P + f;
            dqb[g] = IMG(0, hw)[li];
            dkb[g] = IMG(1, hw)[li];
            dvb[g] = IMG(2, hw)[li];
        }
    }
    if (a.bias != nullptr && a.dbias_part != nullptr) {
        __syncthreads();
        float* tmp = lds;                                     // [4 waves][32 query][32 key]
#pragma unroll
        for (int r = 0; r < 16; ++r) tmp[(wave * 32 + l31) * 32 + row_of(r)] = dbacc[r];       // lane = query, row = key
        __syncthreads();
        float* out = a.dbias_part + ((int64_t)slot * a.heads + head) * 32 * 32;
        for (int e = tid; e < 32 * 32; e += NT) out[e] = (tmp[e] + tmp[1024 + e]) + (tmp[2048 + e] + tmp[3072 + e]);
    }
}

// out[i] = sum_k part[k][i] in a fixed order: 1024 threads = 16 groups x 64 elements, group g adds the slots k = g (mod 16) (loads
// four deep), group 0 adds the sixteen partial sums in order -- a thread walking a thousand slots alone is a chain of a thousand
// dependent loads
constexpr int SR_G = 16;
__global__ __launch_bounds__(64 * SR_G) void sum_rows_kernel(const float* __restrict__ part, float* __restrict__ out, int n, int nsplit) {
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    float s = 0.0f;
    if (i < n) {
        int k = g;
        for (; k + 3 * SR_G < nsplit; k += 4 * SR_G) {
            const float v0 = part[(int64_t)k * n + i], v1 = part[(int64_t)(k + SR_G) * n + i];
            const float v2 = part[(int64_t)(k + 2 * SR_G) * n + i], v3 = part[(int64_t)(k + 3 * SR_G) * n + i];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; k < nsplit; k += SR_G) s += part[(int64_t)k * n + i];
    }
    __shared__ float sh[SR_G][64];
    sh[g][lane] = s;
    __syncthreads();
    if (g == 0 && i < n) {
        float t = sh[0][lane];
        for (int q = 1; q < SR_G; ++q) t += sh[q][lane];
        out[i] = t;
    }
}

// ------------------------------------------------------------------------------------------------ linear attention
// forward:  ks[d][n] = softmax_n(k[d][:]),  ctx[d][e] = sum_n ks[d][n] v[e][n],  qs[d][n] = softmax_d(q[:][n]) scale,
//           out[e][n] = sum_d ctx[d][e] qs[d][n]
// backward: dctx[d][e] = sum_n qs[d][n] do[e][n];  dqs[d][n] = sum_e ctx[d][e] do[e][n];  dv[e][n] = sum_d dctx[d][e] ks[d][n];
//           dks[d][n] = sum_e dctx[d][e] v[e][n];  dq = qsm (dqs scale - sum_d qsm dqs scale);  dk = ks (dks - sum_n ks dks)
// Tokens contiguous.  Four kernels:
//   la_bwd_red    one workgroup per (256-token chunk, sequence, head): the chunk's k row maxima, sums of exp(k - max), unnormalised
//                 ctx and dctx (64-token tiles; thread = 4 (d, e) pairs)                            -> cpart[blk][chunk][2112]
//   la_bwd_merge  per (sequence, head): the chunks merged in a fixed order (maxima, rescaled sums)  -> scratch[blk][2112]
//   la_bwd_tok    thread = token: dq, dv, ks dks (parked in dk), partial T_d = sum ks dks per tile  -> tpart[blk][tile][32]
//   la_bwd_fin    T_d (fixed order over the tiles), dk = ks dks - ks T_d
struct LaBwdArgs {
    const float* qkv; const float* dout; float* dqkv; float* scratch; float* cpart;
    int inner, heads, ntile, nchunk;
    int64_t n, so, sc, si, oso, osc, osi;
};
constexpr int LA_SCR = 32 + 32 + 1024 + 1024;        // rmax, rinv (chunks: sum), ctx, dctx per (sequence, head)
constexpr int LA_CHUNK = 256;

__global__ __launch_bounds__(NT) void la_bwd_red_kernel(const LaBwdArgs a) {
    const int blk = blockIdx.y;
    const int head = blk % a.heads;
    const int seq = blk / a.heads;
    const int o = seq / a.inner, i = seq - o * a.inner;
    const int64_t c0 = (int64_t)blockIdx.x * LA_CHUNK;
    const int64_t c1 = c0 + LA_CHUNK < a.n ? c0 + LA_CHUNK : a.n;
    const float* qb = a.qkv + o * a.so + i * a.si + (int64_t)(head * DH) * a.sc;
    const float* kb = qb + (int64_t)(a.heads * DH) * a.sc;
    const float* vb = kb + (int64_t)(a.heads * DH) * a.sc;
    const float* gb = a.dout + o * a.oso + i * a.osi + (int64_t)(head * DH) * a.osc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ float rmax[DH];
    __shared__ float t_ek[DH][65], t_v[DH][65], t_qs[DH][65], t_g[DH][65];
    __shared__ float qpm[4][64], qps[4][64];
    // ---- k row maxima of the chunk (wave w owns rows 8w .. 8w+7)
    {
        float m[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) m[r] = -INFINITY;
        for (int64_t j = c0 + lane; j < c1; j += 64) {
#pragma unroll
            for (int r = 0; r < 8; ++r) m[r] = fmaxf(m[r], kb[(int64_t)(wave * 8 + r) * a.sc + j]);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float mm = sdc::wave_max(m[r]);
            if (lane == 0) rmax[wave * 8 + r] = mm;
        }
    }
    __syncthreads();
    // ---- sums of exp(k - max), unnormalised ctx, dctx.  thread -> pairs (d = tid >> 3, e = 4 (tid & 7) .. + 3)
    const int pd = tid >> 3, pe = 4 * (tid & 7);
    float cu[4] = {0.f, 0.f, 0.f, 0.f}, dc[4] = {0.f, 0.f, 0.f, 0.f};
    float psum[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) psum[it] = 0.f;
    for (int64_t t0 = c0; t0 < c1; t0 += 64) {
        const int64_t j = t0 + lane;
        const bool live = j < c1;
        // q softmax over d of token j: every wave holds 8 of the 32 rows (row = wave + 4 it), maxima and sums meet in LDS
        float qv[8], pm = -INFINITY;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            qv[it] = live ? qb[(int64_t)(wave + 4 * it) * a.sc + j] : 0.f;
            pm = fmaxf(pm, qv[it]);
        }
        qpm[wave][lane] = pm;
        __syncthreads();
        const float qmx = fmaxf(fmaxf(qpm[0][lane], qpm[1][lane]), fmaxf(qpm[2][lane], qpm[3][lane]));
        float ps = 0.f;
#pragma unroll
        for (int it = 0; it < 8; ++it) { qv[it] = expf(qv[it] - qmx); ps += qv[it]; }
        qps[wave][lane] = ps;
        __syncthreads();
        const float qinv = SCALE / ((qps[0][lane] + qps[1][lane]) + (qps[2][lane] + qps[3][lane]));
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = wave + 4 * it;
            float ek = 0.f, vv = 0.f, qs = 0.f, gg = 0.f;
            if (live) {
                ek = expf(kb[(int64_t)row * a.sc + j] - rmax[row]);
                vv = vb[(int64_t)row * a.sc + j];
                qs = qv[it] * qinv;
                gg = gb[(int64_t)row * a.osc + j];
            }
            t_ek[row][lane] = ek; t_v[row][lane] = vv; t_qs[row][lane] = qs; t_g[row][lane] = gg;
            psum[it] += ek;
        }
        __syncthreads();
        for (int c = 0; c < 64; ++c) {
            const float ekd = t_ek[pd][c], qsd = t_qs[pd][c];
#pragma unroll
            for (int u = 0; u < 4; ++u) { cu[u] += ekd * t_v[pe + u][c]; dc[u] += qsd * t_g[pe + u][c]; }
        }
        __syncthreads();
    }
    float* cp = a.cpart + ((int64_t)blk * a.nchunk + blockIdx.x) * LA_SCR;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const float t = sdc::wave_sum(psum[it]);
        if (lane == 0) cp[32 + wave + 4 * it] = t;
    }
    if (tid < DH) cp[tid] = rmax[tid];
#pragma unroll
    for (int u = 0; u < 4; ++u) { cp[64 + pd * 32 + pe + u] = cu[u]; cp[64 + 1024 + pd * 32 + pe + u] = dc[u]; }
}

__global__ __launch_bounds__(NT) void la_bwd_merge_kernel(const LaBwdArgs a) {
    const int blk = blockIdx.x, tid = threadIdx.x;
    const float* cp = a.cpart + (int64_t)blk * a.nchunk * LA_SCR;
    __shared__ float rmax[DH], rinv[DH];
    if (tid < DH) {
        float m = -INFINITY;
        for (int c = 0; c < a.nchunk; ++c) m = fmaxf(m, cp[(int64_t)c * LA_SCR + tid]);
        float s = 0.f;
        for (int c = 0; c < a.nchunk; ++c) s += cp[(int64_t)c * LA_SCR + 32 + tid] * expf(cp[(int64_t)c * LA_SCR + tid] - m);
        rmax[tid] = m; rinv[tid] = 1.0f / s;
    }
    __syncthreads();
    const int pd = tid >> 3, pe = 4 * (tid & 7);
    float cu[4] = {0.f, 0.f, 0.f, 0.f}, dc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < a.nchunk; ++c) {
        const float* q = cp + (int64_t)c * LA_SCR;
        const float f = expf(q[pd] - rmax[pd]);
#pragma unroll
        for (int u = 0; u < 4; ++u) { cu[u] += q[64 + pd * 32 + pe + u] * f; dc[u] += q[64 + 1024 + pd * 32 + pe + u]; }
    }
    float* scr = a.scratch + (int64_t)blk * LA_SCR;
    if (tid < DH) { scr[tid] = rmax[tid]; scr[32 + tid] = rinv[tid]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { scr[64 + pd * 32 + pe + u] = cu[u] * rinv[pd]; scr[64 + 1024 + pd * 32 + pe + u] = dc[u]; }
}

__global__ __launch_bounds__(NT) void la_bwd_tok_kernel(const LaBwdArgs a, float* __restrict__ tpart) {
    const int blk = blockIdx.y;
    const int head = blk % a.heads;
    const int seq = blk / a.heads;
    const int o = seq / a.inner, i = seq - o * a.inner;
    const int64_t n = a.n;
    const float* qb = a.qkv + o * a.so + i * a.si + (int64_t)(head * DH) * a.sc;
    const float* kb = qb + (int64_t)(a.heads * DH) * a.sc;
    const float* vb = kb + (int64_t)(a.heads * DH) * a.sc;
    const float* gb = a.dout + o * a.oso + i * a.osi + (int64_t)(head * DH) * a.osc;
    float* dqb = a.dqkv + o * a.so + i * a.si + (int64_t)(head * DH) * a.sc;
    float* dkb = dqb + (int64_t)(a.heads * DH) * a.sc;
    float* dvb = dkb + (int64_t)(a.heads * DH) * a.sc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ float rmax[DH], rinv[DH];
    __shared__ float ctx[DH][DH + 1], dctx[DH][DH + 1];
    __shared__ float red[NT / 64][DH];
    {
        const float* scr = a.scratch + (int64_t)blk * LA_SCR;
        if (tid < DH) { rmax[tid] = scr[tid]; rinv[tid] = scr[32 + tid]; }
        for (int e = tid; e < 1024; e += NT) { ctx[e >> 5][e & 31] = scr[64 + e]; dctx[e >> 5][e & 31] = scr[64 + 1024 + e]; }
    }
    __syncthreads();
    float tp[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) tp[d] = 0.f;
    const int64_t j = (int64_t)blockIdx.x * NT + tid;
    if (j < n) {
        float g[DH], ks[DH], v[DH];
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            g[d] = gb[(int64_t)d * a.osc + j];
            v[d] = vb[(int64_t)d * a.sc + j];
            ks[d] = expf(kb[(int64_t)d * a.sc + j] - rmax[d]) * rinv[d];
        }
        float dv[DH];
#pragma unroll
        for (int e = 0; e < DH; ++e) dv[e] = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            float dks = 0.f;
#pragma unroll
            for (int e = 0; e < DH; ++e) { dv[e] += dctx[d][e] * ks[d]; dks += dctx[d][e] * v[e]; }
            tp[d] = ks[d] * dks;
            dkb[(int64_t)d * a.sc + j] = tp[d];               // ks dks, finished by la_bwd_fin
        }
#pragma unroll
        for (int e = 0; e < DH; ++e) dvb[(int64_t)e * a.sc + j] = dv[e];
        float qsm[DH], qmx = -INFINITY, qsum = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) { qsm[d] = qb[(int64_t)d * a.sc + j]; qmx = fmaxf(qmx, qsm[d]); }
#pragma unroll
        for (int d = 0; d < DH; ++d) { qsm[d] = expf(qsm[d] - qmx); qsum += qsm[d]; }
        const float qi = 1.0f / qsum;
        float dqs[DH], dot = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            float sacc = 0.f;
#pragma unroll
            for (int e = 0; e < DH; ++e) sacc += ctx[d][e] * g[e];
            qsm[d] *= qi;
            dqs[d] = sacc * SCALE;
            dot += qsm[d] * dqs[d];
        }
#pragma unroll
        for (int d = 0; d < DH; ++d) dqb[(int64_t)d * a.sc + j] = qsm[d] * (dqs[d] - dot);
    }
    // this tile's part of T_d = sum_n ks dks: wave sums, then the four waves in a fixed order
#pragma unroll
    for (int d = 0; d < DH; ++d) {
        const float t = sdc::wave_sum(tp[d]);
        if (lane == 0) red[wave][d] = t;
    }
    __syncthreads();
    if (tid < DH) tpart[((int64_t)blk * a.ntile + blockIdx.x) * DH + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

__global__ __launch_bounds__(NT) void la_bwd_fin_kernel(const LaBwdArgs a, const float* __restrict__ tpart) {
    const int blk = blockIdx.y;
    const int head = blk % a.heads;
    const int seq = blk / a.heads;
    const int o = seq / a.inner, i = seq - o * a.inner;
    const float* kb = a.qkv + o * a.so + i * a.si + (int64_t)(a.heads * DH + head * DH) * a.sc;
    float* dkb = a.dqkv + o * a.so + i * a.si + (int64_t)(a.heads * DH + head * DH) * a.sc;
    __shared__ float Td[DH], rmax[DH], rinv[DH];
    const int tid = threadIdx.x;
    if (tid < DH) {
        float t = 0.f;
        for (int k = 0; k < a.ntile; ++k) t += tpart[((int64_t)blk * a.ntile + k) * DH + tid];
        Td[tid] = t;
        const float* scr = a.scratch + (int64_t)blk * LA_SCR;
        rmax[tid] = scr[tid];
        rinv[tid] = scr[32 + tid];
    }
    __syncthreads();
    const int64_t j = (int64_t)blockIdx.x * NT + tid;
    if (j >= a.n) return;
#pragma unroll
    for (int d = 0; d < DH; ++d) {
        const float ks = expf(kb[(int64_t)d * a.sc + j] - rmax[d]) * rinv[d];
        dkb[(int64_t)d * a.sc + j] -= ks * Td[d];
    }
}

}  // namespace

extern "C" size_t sdc_attn_bwd_bytes(int outer, int inner, int heads, int ntok) {
    (void)outer; (void)inner;
    return (size_t)1024 * heads * ntok * ntok * sizeof(float);
}

extern "C" int sdc_attn_bwd(const float* qkv, const float* dout, const float* rot, const float* bias, float* dqkv, float* dbias,
                            void* work, int outer, int inner, int heads, int ntok, int64_t q_so, int64_t q_sc, int64_t q_si,
                            int64_t q_st, int64_t o_so, int64_t o_sc, int64_t o_si, int64_t o_st, void* stream) {
    SDC_REQUIRE(qkv && dout && dqkv, SDC_ENULL, "sdc_attn_bwd: null pointer");
    SDC_REQUIRE(outer > 0 && inner > 0 && heads > 0 && ntok > 0 && ntok <= 256, SDC_EINVAL, "sdc_attn_bwd: bad shape (ntok=%d, max 256)", ntok);
    SDC_REQUIRE(!bias || !dbias || (ntok <= 32 && work), SDC_EINVAL, "sdc_attn_bwd: the bias gradient is built for ntok <= 32 and needs the workspace");
    AttnBwdArgs a;
    a.qkv = qkv; a.dout = dout; a.rot = rot; a.bias = bias; a.dqkv = dqkv;
    a.dbias_part = (bias && dbias) ? static_cast<float*>(work) : nullptr;
    a.outer = outer; a.inner = inner; a.heads = heads; a.ntok = ntok;
    a.so = q_so; a.sc = q_sc; a.si = q_si; a.st = q_st;
    a.oso = o_so; a.osc = o_sc; a.osi = o_si; a.ost = o_st;
    a.tok_contig = (q_st == 1);
    hipStream_t s = sdc::as_stream(stream);
    if (!a.tok_contig && ntok == 32 && inner % TB_NS == 0 && q_si == 1 && o_si == 1) {
        // temporal attention of the smoke net: the MFMA kernel
        a.nseq = TB_NS; a.ls = a.ld = a.lj = 0;
        a.ngrp = outer * inner / TB_NS;
        a.nslots = a.ngrp < 1024 ? a.ngrp : 1024;
        const size_t ldsb = (size_t)(4 * TB_NS * TB_IMG + 2 * 32 * TB_P + 2 * 32 * 16 + 4 * 96) * sizeof(float);
        static std::atomic<uint64_t> attr_t{0};
        SDC_LDS_OPTIN(attr_t, tattn_bwd_kernel, 160 * 1024, "sdc_attn_bwd[mfma]");
        hipLaunchKernelGGL(tattn_bwd_kernel, dim3((unsigned)(a.nslots * heads)), dim3(NT), ldsb, s, a);
        if (a.dbias_part) {
            const int nb = heads * ntok * ntok;
            hipLaunchKernelGGL(sum_rows_kernel, dim3((nb + 63) / 64), dim3(64 * SR_G), 0, s, a.dbias_part, dbias, nb, a.nslots);
        }
        return sdc::check_launch("sdc_attn_bwd[mfma]");
    }
    int nseq = NT / ntok;
    if (nseq < 1) nseq = 1;
    const int nseq_tot = outer * inner;
    if (a.tok_contig) {
        if (nseq > nseq_tot) nseq = nseq_tot;
        a.ls = DH * ntok + 1; a.ld = ntok; a.lj = 1;
    } else {
        if (nseq > inner) nseq = inner;
        while (inner % nseq) --nseq;
        a.ls = 1; a.ld = ntok * nseq; a.lj = nseq;
    }
    if (a.dbias_part && nseq > 4) {          // room for the [nseq][ntok][ntok] bias-gradient table
        nseq = 4;
        if (!a.tok_contig) { while (inner % nseq) --nseq; a.ld = ntok * nseq; a.lj = nseq; }
    }
    a.nseq = nseq;
    const size_t ksz = a.tok_contig ? (size_t)nseq * a.ls : (size_t)DH * a.ld;
    const size_t lds_bytes = (4 * ksz + 3 * NT + (a.dbias_part ? (size_t)nseq * ntok * ntok : 0)) * sizeof(float);
    SDC_REQUIRE(lds_bytes <= 160 * 1024, SDC_EINVAL, "sdc_attn_bwd: LDS footprint %zu too large", lds_bytes);
    a.ngrp = (nseq_tot + nseq - 1) / nseq;
    a.nslots = a.ngrp < 1024 ? a.ngrp : 1024;
    static std::atomic<uint64_t> attr{0};
    SDC_LDS_OPTIN(attr, attn_bwd_kernel, 160 * 1024, "sdc_attn_bwd");
    hipLaunchKernelGGL(attn_bwd_kernel, dim3((unsigned)(a.nslots * heads)), dim3(NT), lds_bytes, s, a);
    if (a.dbias_part) {
        const int nb = heads * ntok * ntok;
        hipLaunchKernelGGL(sum_rows_kernel, dim3((nb + 63) / 64), dim3(64 * SR_G), 0, s, a.dbias_part, dbias, nb, a.nslots);
    }
    return sdc::check_launch("sdc_attn_bwd");
}

extern "C" size_t sdc_linattn_bwd_bytes(int outer, int inner, int heads, int64_t n) {
    const size_t nblk = (size_t)outer * inner * heads, ntile = (size_t)((n + NT - 1) / NT), nchunk = (size_t)((n + LA_CHUNK - 1) / LA_CHUNK);
    return nblk * (LA_SCR + ntile * DH + nchunk * LA_SCR) * sizeof(float);
}

extern "C" int sdc_linattn_bwd(const float* qkv, const float* dout, float* dqkv, void* work, int outer, int inner, int heads, int64_t n,
                               int64_t q_so, int64_t q_sc, int64_t q_si, int64_t o_so, int64_t o_sc, int64_t o_si, void* stream) {
    SDC_REQUIRE(qkv && dout && dqkv && work, SDC_ENULL, "sdc_linattn_bwd: null pointer");
    SDC_REQUIRE(outer > 0 && inner > 0 && heads > 0 && n > 0, SDC_EINVAL, "sdc_linattn_bwd: bad shape");
    const int64_t nblk = (int64_t)outer * inner * heads;
    SDC_REQUIRE(nblk < 65536, SDC_EINVAL, "sdc_linattn_bwd: outer*inner*heads must be < 65536");
    LaBwdArgs a;
    a.qkv = qkv; a.dout = dout; a.dqkv = dqkv; a.inner = inner; a.heads = heads; a.n = n;
    a.so = q_so; a.sc = q_sc; a.si = q_si; a.oso = o_so; a.osc = o_sc; a.osi = o_si;
    a.ntile = (int)((n + NT - 1) / NT);
    a.scratch = static_cast<float*>(work);
    a.nchunk = (int)((n + LA_CHUNK - 1) / LA_CHUNK);
    float* tpart = a.scratch + nblk * LA_SCR;
    a.cpart = tpart + nblk * a.ntile * DH;
    hipStream_t s = sdc::as_stream(stream);
    hipLaunchKernelGGL(la_bwd_red_kernel, dim3((unsigned)a.nchunk, (unsigned)nblk), dim3(NT), 0, s, a);
    hipLaunchKernelGGL(la_bwd_merge_kernel, dim3((unsigned)nblk), dim3(NT), 0, s, a);
    hipLaunchKernelGGL(la_bwd_tok_kernel, dim3((unsigned)a.ntile, (unsigned)nblk), dim3(NT), 0, s, a, tpart);
    hipLaunchKernelGGL(la_bwd_fin_kernel, dim3((unsigned)a.ntile, (unsigned)nblk), dim3(NT), 0, s, a, (const float*)tpart);
    return sdc::check_launch("sdc_linattn_bwd");
}
