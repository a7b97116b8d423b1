// Attention cores (heads x dim_head 32, fp32):
//   * linear attention   (LinearAttention / SpatialLinearAttention)
//   * softmax attention  (mid Attention, spatial mid attention, temporal attention with rotary + rel-pos bias)
// Both read q,k,v straight out of the channel-major conv output through strides, so no permute /
// rearrange copy is ever materialised ('b (h c) x y -> b h c (x y)', 'b c f h w -> b (h w) f c').
#include "sdc_common.h"

namespace {

constexpr int NT = 256;
constexpr int DH = 32;

// ------------------------------------------------------------------ linear attention: context
// one workgroup per (sequence, head): ctx[d][e] = sum_n softmax_n(k[d,:])[n] * v[e,n]
__global__ __launch_bounds__(NT) void la_ctx_kernel(const float* __restrict__ qkv, float* __restrict__ ctx, int inner,
                                                    int heads, int64_t n, int64_t so, int64_t sc, int64_t si) {
    const int head = blockIdx.x % heads;
    const int seq = blockIdx.x / heads;
    const int o = seq / inner, i = seq - o * inner;
    const float* kb = qkv + o * so + i * si + (int64_t)(heads * DH + head * DH) * sc;
    const float* vb = qkv + o * so + i * si + (int64_t)(2 * heads * DH + head * DH) * sc;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    __shared__ float rmax[DH], rinv[DH];
    __shared__ float pk[DH][65], vv[DH][65];

    // row statistics: each wave owns 8 rows
    for (int r = 0; r < 8; ++r) {
        const int d = wave * 8 + r;
        const float* row = kb + (int64_t)d * sc;
        float m = -INFINITY;
        for (int64_t j = lane; j < n; j += 64) m = fmaxf(m, row[j]);
        m = sdc::wave_max(m);
        float s = 0.f;
        for (int64_t j = lane; j < n; j += 64) s += expf(row[j] - m);
        s = sdc::wave_sum(s);
        if (lane == 0) { rmax[d] = m; rinv[d] = 1.0f / s; }
    }
    __syncthreads();

    const int d = threadIdx.x >> 3;
    const int e0 = (threadIdx.x & 7) * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t t0 = 0; t0 < n; t0 += 64) {
        // stage exp(k - max) and v tiles: thread -> (row = wave + 4*it, col = lane)
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = wave + 4 * it;
            const int64_t j = t0 + lane;
            float kvv = 0.f, vvv = 0.f;
            if (j < n) {
                kvv = expf(kb[(int64_t)row * sc + j] - rmax[row]);
                vvv = vb[(int64_t)row * sc + j];
            }
            pk[row][lane] = kvv;
            vv[row][lane] = vvv;
        }
        __syncthreads();
#pragma unroll 8
        for (int j = 0; j < 64; ++j) {
            const float p = pk[d][j];
            acc[0] += p * vv[e0 + 0][j];
            acc[1] += p * vv[e0 + 1][j];
            acc[2] += p * vv[e0 + 2][j];
            acc[3] += p * vv[e0 + 3][j];
        }
        __syncthreads();
    }
    float* out = ctx + (int64_t)blockIdx.x * DH * DH + d * DH + e0;
    const float inv = rinv[d];
    out[0] = acc[0] * inv; out[1] = acc[1] * inv; out[2] = acc[2] * inv; out[3] = acc[3] * inv;
}

// ------------------------------------------------------------------ linear attention: output
// thread per token: q softmax over d, out[e] = sum_d ctx[d][e] * q[d] * scale
__global__ __launch_bounds__(NT) void la_out_kernel(const float* __restrict__ qkv, const float* __restrict__ ctx,
                                                    float* __restrict__ out, int inner, int heads, int64_t n,
                                                    int64_t so, int64_t sc, int64_t si, int64_t oso, int64_t osc,
                                                    int64_t osi) {
    const int head = blockIdx.y % heads;
    const int seq = blockIdx.y / heads;
    const int o = seq / inner, i = seq - o * inner;
    __shared__ float cs[DH][DH];
    for (int e = threadIdx.x; e < DH * DH; e += NT) cs[e / DH][e % DH] = ctx[(int64_t)blockIdx.y * DH * DH + e];
    __syncthreads();
    const int64_t j = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (j >= n) return;
    const float* qb = qkv + o * so + i * si + (int64_t)(head * DH) * sc + j;
    float q[DH];
    float m = -INFINITY;
#pragma unroll
    for (int d = 0; d < DH; ++d) { q[d] = qb[(int64_t)d * sc]; m = fmaxf(m, q[d]); }
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) { q[d] = expf(q[d] - m); s += q[d]; }
    const float f = 0.17677669529663687f / s;   // dim_head^-0.5 / sum
    float acc[DH];
#pragma unroll
    for (int e = 0; e < DH; ++e) acc[e] = 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) {
        const float qd = q[d] * f;
#pragma unroll
        for (int e = 0; e < DH; e += 4) {
            const float4 c4 = *reinterpret_cast<const float4*>(&cs[d][e]);
            acc[e] += c4.x * qd; acc[e + 1] += c4.y * qd; acc[e + 2] += c4.z * qd; acc[e + 3] += c4.w * qd;
        }
    }
    float* ob = out + o * oso + i * osi + (int64_t)(head * DH) * osc + j;
#pragma unroll
    for (int e = 0; e < DH; ++e) ob[(int64_t)e * osc] = acc[e];
}

// ------------------------------------------------------------------ softmax attention
// A workgroup owns `nseq` sequences of one head (nseq*ntok <= 256 threads, one query per thread).
// K and V of those sequences sit in LDS; the LDS index order follows the memory order of the
// sequences so that both the global loads and the LDS reads stay conflict-free:
//   tokens contiguous in memory (st == 1): thread = seq*ntok + tok, K[seq][d][tok]
//   tokens strided (temporal attention)  : thread = tok*nseq + seq, K[d][tok][seq] (seq = adjacent pixels)
struct AttnArgs {
    const float* qkv; float* out; const float* rot; const float* bias;
    int outer, inner, heads, ntok, nseq, tok_contig;
    int64_t so, sc, si, st, oso, osc, osi, ost;
    int ls, ld, lj;   // LDS strides for (seq, d, tok)
};

__global__ __launch_bounds__(NT) void attn_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ntok = a.ntok, nseq = a.nseq;
    const int ksz = a.tok_contig ? nseq * a.ls : DH * a.ld;
    float* Ks = lds;
    float* Vs = lds + ksz;
    const int head = blockIdx.x % a.heads;
    const int grp = blockIdx.x / a.heads;
    const int seq0 = grp * nseq;
    const int nseq_tot = a.outer * a.inner;
    const int tid = threadIdx.x;
    const int nthr = nseq * ntok;

    auto seq_base = [&](int s) -> int64_t {
        const int o = s / a.inner, i = s - o * a.inner;
        return o * a.so + i * a.si;
    };
    auto seq_obase = [&](int s) -> int64_t {
        const int o = s / a.inner, i = s - o * a.inner;
        return o * a.oso + i * a.osi;
    };

    // ---- stage K, V
    const int total = nseq * DH * ntok;
    const int64_t koff = (int64_t)(a.heads * DH + head * DH) * a.sc;
    const int64_t voff = (int64_t)(2 * a.heads * DH + head * DH) * a.sc;
    for (int e = tid; e < total; e += NT) {
        int sq, d, j;
        if (a.tok_contig) { j = e % ntok; d = (e / ntok) % DH; sq = e / (ntok * DH); }
        else { sq = e % nseq; j = (e / nseq) % ntok; d = e / (nseq * ntok); }
        float kv = 0.f, vv = 0.f;
        if (seq0 + sq < nseq_tot) {
            const int64_t g = seq_base(seq0 + sq) + (int64_t)d * a.sc + (int64_t)j * a.st;
            kv = a.qkv[g + koff];
            vv = a.qkv[g + voff];
        }
        const int li = sq * a.ls + d * a.ld + j * a.lj;
        Ks[li] = kv;
        Vs[li] = vv;
    }
    __syncthreads();
    if (a.rot) {
        // rotate K pairs in place: (x0,x1) -> (x0 c - x1 s, x1 c + x0 s), angle = tok * freq[pair]
        const int npair = nseq * (DH / 2) * ntok;
        for (int e = tid; e < npair; e += NT) {
            int sq, m, j;
            if (a.tok_contig) { j = e % ntok; m = (e / ntok) % (DH / 2); sq = e / (ntok * (DH / 2)); }
            else { sq = e % nseq; j = (e / nseq) % ntok; m = e / (nseq * ntok); }
            const float c = a.rot[(j * (DH / 2) + m) * 2], s = a.rot[(j * (DH / 2) + m) * 2 + 1];
            const int l0 = sq * a.ls + (2 * m) * a.ld + j * a.lj, l1 = l0 + a.ld;
            const float x0 = Ks[l0], x1 = Ks[l1];
            Ks[l0] = x0 * c - x1 * s;
            Ks[l1] = x1 * c + x0 * s;
        }
        __syncthreads();
    }
    if (tid >= nthr) return;
    int sq, ti;
    if (a.tok_contig) { ti = tid % ntok; sq = tid / ntok; }
    else { sq = tid % nseq; ti = tid / nseq; }
    if (seq0 + sq >= nseq_tot) return;

    // ---- this thread's query
    const float scale = 0.17677669529663687f;
    float q[DH];
    {
        const int64_t g = seq_base(seq0 + sq) + (int64_t)(head * DH) * a.sc + (int64_t)ti * a.st;
#pragma unroll
        for (int d = 0; d < DH; ++d) q[d] = a.qkv[g + (int64_t)d * a.sc] * scale;
        if (a.rot) {
#pragma unroll
            for (int m = 0; m < DH / 2; ++m) {
                const float c = a.rot[(ti * (DH / 2) + m) * 2], s = a.rot[(ti * (DH / 2) + m) * 2 + 1];
                const float x0 = q[2 * m], x1 = q[2 * m + 1];
                q[2 * m] = x0 * c - x1 * s;
                q[2 * m + 1] = x1 * c + x0 * s;
            }
        }
    }
    const float* brow = a.bias ? a.bias + ((int64_t)head * ntok + ti) * ntok : nullptr;
    const float* Kq = Ks + sq * a.ls;
    const float* Vq = Vs + sq * a.ls;

    float mx = -INFINITY;
    for (int j = 0; j < ntok; ++j) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) s += q[d] * Kq[d * a.ld + j * a.lj];
        if (brow) s += brow[j];
        mx = fmaxf(mx, s);
    }
    float l = 0.f;
    float o[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) o[d] = 0.f;
    for (int j = 0; j < ntok; ++j) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) s += q[d] * Kq[d * a.ld + j * a.lj];
        if (brow) s += brow[j];
        const float p = expf(s - mx);
        l += p;
#pragma unroll
        for (int d = 0; d < DH; ++d) o[d] += p * Vq[d * a.ld + j * a.lj];
    }
    const float inv = 1.0f / l;
    const int64_t og = seq_obase(seq0 + sq) + (int64_t)(head * DH) * a.osc + (int64_t)ti * a.ost;
#pragma unroll
    for (int d = 0; d < DH; ++d) a.out[og + (int64_t)d * a.osc] = o[d] * inv;
}

}  // namespace

extern "C" int sdc_linattn(const float* qkv, float* ctx, float* out, int outer, int inner, int heads, int64_t n,
                           int64_t q_so, int64_t q_sc, int64_t q_si, int64_t o_so, int64_t o_sc, int64_t o_si,
                           void* stream) {
    SDC_REQUIRE(qkv && ctx && out, SDC_ENULL, "sdc_linattn: null pointer");
    SDC_REQUIRE(outer > 0 && inner > 0 && heads > 0 && n > 0, SDC_EINVAL, "sdc_linattn: bad shape");
    const int64_t nblk = (int64_t)outer * inner * heads;
    SDC_REQUIRE(nblk < 65536 * 16, SDC_EINVAL, "sdc_linattn: too many sequences");
    hipStream_t s = sdc::as_stream(stream);
    hipLaunchKernelGGL(la_ctx_kernel, dim3((unsigned)nblk), dim3(NT), 0, s, qkv, ctx, inner, heads, n, q_so, q_sc, q_si);
    SDC_REQUIRE(nblk < 65536, SDC_EINVAL, "sdc_linattn: outer*inner*heads must be < 65536");
    hipLaunchKernelGGL(la_out_kernel, dim3((unsigned)((n + NT - 1) / NT), (unsigned)nblk), dim3(NT), 0, s, qkv, ctx, out,
                       inner, heads, n, q_so, q_sc, q_si, o_so, o_sc, o_si);
    return sdc::check_launch("sdc_linattn");
}

extern "C" int sdc_attn(const float* qkv, float* out, const float* rot, const float* bias, int outer, int inner,
                        int heads, int ntok, int64_t q_so, int64_t q_sc, int64_t q_si, int64_t q_st, int64_t o_so,
                        int64_t o_sc, int64_t o_si, int64_t o_st, void* stream) {
    SDC_REQUIRE(qkv && out, SDC_ENULL, "sdc_attn: null pointer");
    SDC_REQUIRE(outer > 0 && inner > 0 && heads > 0 && ntok > 0 && ntok <= 256, SDC_EINVAL,
                "sdc_attn: bad shape (ntok=%d, max 256)", ntok);
    AttnArgs a;
    a.qkv = qkv; a.out = out; a.rot = rot; a.bias = bias;
    a.outer = outer; a.inner = inner; a.heads = heads; a.ntok = ntok;
    a.so = q_so; a.sc = q_sc; a.si = q_si; a.st = q_st;
    a.oso = o_so; a.osc = o_sc; a.osi = o_si; a.ost = o_st;
    a.tok_contig = (q_st == 1);
    int nseq = NT / ntok;
    if (nseq < 1) nseq = 1;
    const int nseq_tot = outer * inner;
    if (a.tok_contig) {
        if (nseq > nseq_tot) nseq = nseq_tot;
        a.ls = DH * ntok + 1; a.ld = ntok; a.lj = 1;
    } else {
        // sequences must be adjacent in memory along `inner` for the strided form to coalesce
        if (nseq > inner) nseq = inner;
        while (inner % nseq) --nseq;     // never straddle an `outer` boundary
        a.ls = 1; a.ld = ntok * nseq; a.lj = nseq;
    }
    a.nseq = nseq;
    const size_t ksz = a.tok_contig ? (size_t)nseq * a.ls : (size_t)DH * a.ld;
    const size_t lds_bytes = 2 * ksz * sizeof(float);
    SDC_REQUIRE(lds_bytes <= 160 * 1024, SDC_EINVAL, "sdc_attn: LDS footprint %zu too large", lds_bytes);
    const int ngrp = (nseq_tot + nseq - 1) / nseq;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(attn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL(attn_kernel, dim3((unsigned)(ngrp * heads)), dim3(NT), lds_bytes, sdc::as_stream(stream), a);
    return sdc::check_launch("sdc_attn");
}
