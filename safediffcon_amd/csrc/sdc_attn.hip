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

    // row maxima: each wave owns 8 rows and sweeps them together (8 independent loads in flight per step);
    // the softmax denominators come for free from the tile pass below
    {
        float m[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) m[r] = -INFINITY;
        for (int64_t j = lane; j < n; j += 64) {
#pragma unroll
            for (int r = 0; r < 8; ++r) m[r] = fmaxf(m[r], kb[(int64_t)(wave * 8 + r) * sc + j]);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float mm = sdc::wave_max(m[r]);
            if (lane == 0) rmax[wave * 8 + r] = mm;
        }
    }
    __syncthreads();
    float psum[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) psum[it] = 0.f;

    // ctx = P . V^T on the matrix cores: per 64-token tile every wave takes 16 tokens = 8 k-steps of
    // v_mfma_f32_32x32x2_f32 (A[d][n] = exp(k - max), B[n][e] = v), partial tiles are summed through LDS at the end
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int l31 = lane & 31, lh = lane >> 5;
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
            psum[it] += kvv;
        }
        __syncthreads();
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
            const int col = wave * 16 + 2 * s8 + lh;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pk[l31][col], vv[l31][col], acc, 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {             // softmax denominators: row (wave + 4*it) was staged by this wave only
        const float t = sdc::wave_sum(psum[it]);
        if (lane == 0) rinv[wave + 4 * it] = 1.0f / t;
    }
    // lane (e = l31, half lh) holds d = (r&3) + 8*(r>>2) + 4*lh; reduce the four waves' partial contexts
    float* red = &pk[0][0];                      // 32 x 65 floats are enough for one 32x32 partial per pass
    float tot[4] = {0.f, 0.f, 0.f, 0.f};
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + l31] = acc[r];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = threadIdx.x * 4 + i;   // o = d*32 + e
            tot[i] += red[(o >> 5) * 33 + (o & 31)];
        }
        __syncthreads();
    }
    float* out = ctx + (int64_t)blockIdx.x * DH * DH + threadIdx.x * 4;
    const float inv = rinv[threadIdx.x >> 3];
    out[0] = tot[0] * inv; out[1] = tot[1] * inv; out[2] = tot[2] * inv; out[3] = tot[3] * inv;
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
        const float p = sdc::softmax_exp(s - mx);
        l += p;
#pragma unroll
        for (int d = 0; d < DH; ++d) o[d] += p * Vq[d * a.ld + j * a.lj];
    }
    const float inv = 1.0f / l;
    const int64_t og = seq_obase(seq0 + sq) + (int64_t)(head * DH) * a.osc + (int64_t)ti * a.ost;
#pragma unroll
    for (int d = 0; d < DH; ++d) a.out[og + (int64_t)d * a.osc] = o[d] * inv;
}

// ------------------------------------------------------------------ temporal attention on the matrix cores
// 32 frames x dim_head 32 is exactly one v_mfma_f32_32x32x2_f32 tile per product:
//   S^T[key][query] = K . Q^T   (16 k-steps over d)      -> scores of one query sit in ONE lane's registers
//   O[query][d]     = P^T . V   (16 k-steps over keys)   -> the P registers are the A operand as they stand
// A workgroup owns 8 adjacent pixels (32-byte runs of the channel-major tensor) of one head; Q and K are staged
// through LDS transposed to [pixel][d][frame], V to [pixel][frame][d] (loaded early, kept in registers while
// Q.K^T runs, written over the Q tile afterwards); rotary + rel-pos bias + softmax run on the accumulators; the
// output goes back through LDS so that global stores are 32-byte runs again.  Workgroups are numbered so that the
// 4 pixel groups sharing a 128-byte line land on the same XCD (speed only).
constexpr int TA_NS = 8;                 // sequences (adjacent pixels) per workgroup, at least (16 where the image allows)

// wave-uniform base (SGPR pair) + one 32-bit per-lane byte offset for all 32 channel rows of a thread's element
typedef __attribute__((address_space(1))) float* ta_gptr;
typedef __attribute__((address_space(1))) char* ta_gcptr;
__device__ __forceinline__ ta_gptr ta_uni(const float* p) {
    const uint64_t u = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return (ta_gptr)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ float ta_ld(ta_gptr base, uint32_t byte_off) { return *(ta_gptr)((ta_gcptr)base + byte_off); }
__device__ __forceinline__ void ta_st(ta_gptr base, uint32_t byte_off, float v) { *(ta_gptr)((ta_gcptr)base + byte_off) = v; }

// NS = 8 pixels per workgroup (4 waves, 32-byte runs, two workgroups per CU) or 16 (8 waves, 64-byte runs = one whole HBM
// burst per row, one workgroup per CU): the same LDS bytes and waves per CU, half the partly used bursts.
template <int NS>
__global__ __launch_bounds__(NS * 32, NS == 8 ? 2 : 1) void tattn_kernel(const AttnArgs a, const int tiles_per_wg) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    constexpr int NT = NS * 32;                                       // (shadows the file's 256)
    constexpr int TA_SQ = NS == 8 ? 32 * 32 + 8 : 32 * 32 + 2;        // pixel pitch of the [d][f] image: 8 hw + f (NS = 8) / 2 hw + f (16) spreads a store over the banks
    constexpr int TA_SV = NS == 8 ? 32 * 33 + 8 : 32 * 33 + 2;        // pixel pitch of the [f][d] image (33-float frame pitch)
    extern __shared__ float ta_lds[];
    float* const Ks = ta_lds;                                         // [NS][TA_SQ]
    float* const QVs = Ks + NS * TA_SQ;                               // Q ([d][f], TA_SQ pitch) first, then V / O ([f][d], TA_SV pitch)
    float (*const biasT)[33] = reinterpret_cast<float (*)[33]>(QVs + NS * TA_SV);       // [key][query] of this head
    float (*const rotc)[33] = biasT + 32;                             // [m][frame]: a wave reads one m, 32 frames
    float (*const rots)[33] = rotc + 16;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    // A workgroup walks tiles_per_wg tiles, gridDim.x apart (the host picks gridDim.x as a multiple of 8 heads, so all of them
    // belong to one head: bias / rotary tables are staged once) and requests tile t+1's Q and K while tile t is on the matrix
    // cores -- two workgroups per CU (LDS) did not cover each other's load phases: 2.9 TB/s.
    // XCD-aware numbering of the virtual tile id: consecutive logical ids stay on one XCD.
    const int nblk = gridDim.x * tiles_per_wg;
    // (a workgroup's tiles are gridDim.x apart on purpose: the four pixel groups that share a 128-byte line must be in flight on
    // one XCD at the same time -- walking them one after the other in one workgroup fetched every line four times: 2.6 -> 4.3 ms)
    auto tile_of = [&](int t) { const int v = blockIdx.x + t * gridDim.x; return ((nblk & 7) == 0) ? (v & 7) * (nblk >> 3) + (v >> 3) : v; };
    const int head = tile_of(0) % a.heads;
    const float scale = 0.17677669529663687f;
    // element e = tid + NT*it of a tile -> (d = it, f = (e / NS) & 31, hw = e % NS): channel row it from a scalar base, the
    // thread's (frame, pixel) as ONE byte offset (host check: a frame stride x 32 frames stays below 2^32 bytes)
    float kreg[32], qreg[32], vreg[32];
    const uint32_t toff = (uint32_t)(((int64_t)((tid / NS) & 31) * a.st + (tid % NS)) * 4);
    const uint32_t toffo = (uint32_t)(((int64_t)((tid / NS) & 31) * a.ost + (tid % NS)) * 4);
    auto tile_base = [&](int bid, const float*& qb, float*& ob) {
        const int seq0 = (bid / a.heads) * NS;
        const int o = seq0 / a.inner, i0 = seq0 - o * a.inner;
        qb = a.qkv + o * a.so + i0 + (int64_t)(head * DH) * a.sc;
        ob = a.out + o * a.oso + i0 + (int64_t)(head * DH) * a.osc;
    };
    auto load_qk = [&](const float* qb) {
        const float* kb = qb + (int64_t)(a.heads * DH) * a.sc;
        // (running scalar bases: 64 base pairs computed up front do not fit the SGPR file and were spilled into VGPR lanes)
        ta_gptr kp = ta_uni(kb), qp = ta_uni(qb);
        const int64_t step = a.sc;
#pragma unroll
        for (int it = 0; it < 32; ++it) {
            kreg[it] = ta_ld(kp, toff);
            qreg[it] = ta_ld(qp, toff);
            kp += step; qp += step;
            asm volatile("" : "+s"(kp), "+s"(qp));
        }
    };
    for (int e = tid; e < 32 * 32; e += NT) {
        const int q = e >> 5, kk = e & 31;
        biasT[kk][q] = a.bias ? a.bias[((int64_t)head * 32 + q) * 32 + kk] : 0.0f;
    }
    if (a.rot)
        for (int e = tid; e < 32 * 16; e += NT) {
            rotc[e & 15][e >> 4] = a.rot[e * 2];
            rots[e & 15][e >> 4] = a.rot[e * 2 + 1];
        }
    const float* qb; float* ob;
    tile_base(tile_of(0), qb, ob);
    load_qk(qb);
    for (int t = 0; t < tiles_per_wg; ++t) {
        // ---- stage Q (scaled) and K of this tile from the registers; V is only needed after the softmax: issue its loads now
#pragma unroll
        for (int it = 0; it < 32; ++it) {
            const int e = tid + it * NT;
            const int hw = e % NS, f = (e / NS) & 31, d = e / (NS * 32);
            Ks[hw * TA_SQ + d * 32 + f] = kreg[it];
            QVs[hw * TA_SQ + d * 32 + f] = qreg[it] * scale;
        }
        {
            const float* vb = qb + (int64_t)(2 * a.heads * DH) * a.sc;
            ta_gptr vp = ta_uni(vb);
            const int64_t step = a.sc;
#pragma unroll
            for (int it = 0; it < 32; ++it) {
                vreg[it] = ta_ld(vp, toff);
                vp += step;
                asm volatile("" : "+s"(vp));
            }
        }
        float* const ob_cur = ob;
        if (t + 1 < tiles_per_wg) {                      // the next tile's Q and K travel under this tile's products
            tile_base(tile_of(t + 1), qb, ob);
            load_qk(qb);
        }
        __syncthreads();
        if (a.rot) {
            // rotate (d = 2m, 2m+1) pairs of Q and K in place; angle = frame * freq[m]
            for (int e = tid; e < NS * 16 * 32; e += NT) {
                const int f = e & 31, m = (e >> 5) & 15, hw = e >> 9;
                const float c = rotc[m][f], sn = rots[m][f];
                const int l0 = hw * TA_SQ + (2 * m) * 32 + f, l1 = l0 + 32;
                float x0 = Ks[l0], x1 = Ks[l1];
                Ks[l0] = x0 * c - x1 * sn; Ks[l1] = x1 * c + x0 * sn;
                x0 = QVs[l0]; x1 = QVs[l1];
                QVs[l0] = x0 * c - x1 * sn; QVs[l1] = x1 * c + x0 * sn;
            }
            __syncthreads();
        }

        // ---- S^T = K . Q^T, softmax over keys; wave w owns pixels 2w and 2w+1
        f32x16 p[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int hw = wave * 2 + u;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float af = Ks[hw * TA_SQ + (2 * s + lh) * 32 + l31];      // A[key][d]
                const float bf = QVs[hw * TA_SQ + (2 * s + lh) * 32 + l31];     // B[d][query]
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc, 0, 0, 0);
            }
            // lane (query = l31, half lh) holds keys (r&3) + 8*(r>>2) + 4*lh
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[r] += biasT[(r & 3) + 8 * (r >> 2) + 4 * lh][l31];
                mx = fmaxf(mx, acc[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[r] = sdc::softmax_exp(acc[r] - mx); sum += acc[r]; }
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] *= inv;
            p[u] = acc;
        }
        __syncthreads();                              // every wave is done with the Q tile
#pragma unroll
        for (int it = 0; it < 32; ++it) {
            const int e = tid + it * NT;
            const int hw = e % NS, f = (e / NS) & 31, d = e / (NS * 32);
            QVs[hw * TA_SV + f * 33 + d] = vreg[it];
        }
        __syncthreads();

        // ---- O = P^T . V : k-step r pairs key (r&3)+8*(r>>2) [half 0] with that key + 4 [half 1] -- exactly register r
        f32x16 oacc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int hw = wave * 2 + u;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float bf = QVs[hw * TA_SV + key * 33 + l31];              // B[key][d]
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p[u][r], bf, acc, 0, 0, 0);   // A[query][key] = P^T
            }
            oacc[u] = acc;
        }
        __syncthreads();                              // V no longer needed: reuse its tile for the output image
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int hw = wave * 2 + u;
            // lane (d = l31, half lh) holds queries (r&3) + 8*(r>>2) + 4*lh
#pragma unroll
            for (int r = 0; r < 16; ++r) QVs[hw * TA_SV + ((r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + l31] = oacc[u][r];
        }
        __syncthreads();
        ta_gptr op = ta_uni(ob_cur);
#pragma unroll
        for (int it = 0; it < 32; ++it) {
            const int e = tid + it * NT;
            const int hw = e % NS, f = (e / NS) & 31, d = e / (NS * 32);
            ta_st(op, toffo, QVs[hw * TA_SV + f * 33 + d]);
            op += a.osc;
            asm volatile("" : "+s"(op));
        }
        __syncthreads();                              // the next tile's staging writes Ks / QVs
    }
}


// ------------------------------------------------------------------ softmax attention over 256 contiguous tokens on the matrix cores
// The smoke net's mid spatial attention (conv3d.py:450-452,548: 16x16 = 256 tokens, heads 4 x 32, no rotary / bias).  One
// workgroup = one (sequence, head): K [d][tok] and V [d][tok] (pitch 257: the O product reads V down a column) are staged in
// LDS; wave w owns the query blocks w and w + 4 (32 queries each).  Per query block, S^T[key][query] = K . Q^T over the 8 key
// blocks stays in registers (8 x 16 accumulators: the 256 scores of one query sit in two lanes), softmax runs on them, and the
// P registers are the B operands of O^T[d][query] = V . P as they stand.  2048 MFMAs per workgroup instead of 2 x 256 x 32
// scalar FMA loops per query (attn_kernel computed Q.K^T twice): 13.9 -> ~70 TFLOP/s.
constexpr int A2_VP = 257;
__global__ __launch_bounds__(NT) void attn256_kernel(const AttnArgs a) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const Ks = lds;                       // [32 d][256 tok]
    float* const Vs = lds + 32 * 256;            // [32 d][257]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int head = blockIdx.x % a.heads, sq = blockIdx.x / a.heads;
    const int o = sq / a.inner, i = sq - o * a.inner;
    const float* qb = a.qkv + o * a.so + i * a.si + (int64_t)(head * DH) * a.sc;
    const float* kb = qb + (int64_t)(a.heads * DH) * a.sc;
    const float* vb = kb + (int64_t)(a.heads * DH) * a.sc;
    typedef float nfloat4 __attribute__((ext_vector_type(4)));
    // stage K and V: 32 rows x 256 tokens each, 16-byte loads (8 per thread per tensor)
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int f = tid + it * NT;               // float4 index: row d = f >> 6, 4 tokens at (f & 63) * 4
        const int d = f >> 6, t4 = (f & 63) * 4;
        const nfloat4 kv = *reinterpret_cast<const nfloat4*>(kb + (int64_t)d * a.sc + t4);
        const nfloat4 vv = *reinterpret_cast<const nfloat4*>(vb + (int64_t)d * a.sc + t4);
        *reinterpret_cast<nfloat4*>(Ks + d * 256 + t4) = kv;
        float* vp = Vs + d * A2_VP + t4;
        vp[0] = vv.x; vp[1] = vv.y; vp[2] = vv.z; vp[3] = vv.w;
    }
    __syncthreads();
    const float scale = 0.17677669529663687f;
    float* ob = a.out + o * a.oso + i * a.osi + (int64_t)(head * DH) * a.osc;
    for (int u = 0; u < 2; ++u) {
        const int q0 = (wave + 4 * u) * 32;
        // (the K / V fragment reads below do not depend on u: left to itself the compiler hoists all 256 of them out of this
        // loop and keeps them in scratch -- 80 spilled registers; an offset it cannot see through keeps them where they are used)
        int koff = lh * 256 + l31, voff = l31 * A2_VP + 4 * lh;
        asm volatile("" : "+v"(koff), "+v"(voff));
        // B fragments of Q^T: B[d = 2s + lh][query = l31]
        float qf[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) qf[s] = qb[(int64_t)(2 * s + lh) * a.sc + q0 + l31] * scale;
        f32x16 sc[8];
#pragma unroll
        for (int kbk = 0; kbk < 8; ++kbk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[kbk][r] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s)           // A[key = l31][d = 2s + lh]
                sc[kbk] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[koff + (2 * s) * 256 + kbk * 32], qf[s], sc[kbk], 0, 0, 0);
        }
        // lane (query = l31, half lh) holds keys kbk*32 + (r&3) + 8*(r>>2) + 4*lh
        float mx = -INFINITY;
#pragma unroll
        for (int kbk = 0; kbk < 8; ++kbk)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[kbk][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kbk = 0; kbk < 8; ++kbk)
#pragma unroll
            for (int r = 0; r < 16; ++r) { sc[kbk][r] = sdc::softmax_exp(sc[kbk][r] - mx); sum += sc[kbk][r]; }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        // O^T[d][query] = sum_key V[d][key] P[key][query]: A[d = l31][key], B = P registers (k-step r of block kbk)
        f32x16 oacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
#pragma unroll
        for (int kbk = 0; kbk < 8; ++kbk)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[voff + kbk * 32 + (r & 3) + 8 * (r >> 2)], sc[kbk][r], oacc, 0, 0, 0);
        // lane (query = l31, half lh) holds d = (r&3) + 8*(r>>2) + 4*lh: 128-byte runs along the tokens
#pragma unroll
        for (int r = 0; r < 16; ++r) ob[(int64_t)((r & 3) + 8 * (r >> 2) + 4 * lh) * a.osc + q0 + l31] = oacc[r] * inv;
    }
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
    if (!a.tok_contig && ntok == 32 && inner % TA_NS == 0 && q_si == 1 && o_si == 1 && q_st > 0 && o_st > 0 && q_st < (1ll << 24) && o_st < (1ll << 24)) {     // (frame stride x 32 frames x 4 bytes < 2^32: 32-bit lane offsets)
        const int ns = inner % 16 == 0 ? 16 : 8;
        const int nblk = (outer * inner / ns) * heads;
        // tiles per workgroup: the largest of 4 / 2 / 1 that leaves a grid of a multiple of 8 heads (one head per workgroup under
        // the XCD numbering) and still >= 4 workgroups per CU
        int tpw = 1;
        for (int t = 4; t > 1; t >>= 1)
            if (nblk % (t * 8 * heads) == 0 && nblk / t >= 1024) { tpw = t; break; }
        const size_t ldsb = sizeof(float) * (size_t)(ns * ((ns == 8 ? 1032 : 1026) + (ns == 8 ? 1064 : 1058)) + 64 * 33);
        if (ns == 16) {
            static std::atomic<uint64_t> attr_t{0};
            SDC_LDS_OPTIN(attr_t, tattn_kernel<16>, 160 * 1024, "sdc_attn[mfma]");
            hipLaunchKernelGGL(tattn_kernel<16>, dim3((unsigned)(nblk / tpw)), dim3(512), ldsb, sdc::as_stream(stream), a, tpw);
        } else {
            static std::atomic<uint64_t> attr_t8{0};
            SDC_LDS_OPTIN(attr_t8, tattn_kernel<8>, 160 * 1024, "sdc_attn[mfma]");
            hipLaunchKernelGGL(tattn_kernel<8>, dim3((unsigned)(nblk / tpw)), dim3(256), ldsb, sdc::as_stream(stream), a, tpw);
        }
        return sdc::check_launch("sdc_attn[mfma]");
    }
    if (a.tok_contig && ntok == 256 && !rot && !bias && o_st == 1 && q_sc % 4 == 0 && q_so % 4 == 0 && q_si % 4 == 0 &&
        reinterpret_cast<uintptr_t>(qkv) % 16 == 0) {
        const size_t ldsb = sizeof(float) * (size_t)(32 * 256 + 32 * A2_VP);
        static std::atomic<uint64_t> attr2{0};
        SDC_LDS_OPTIN(attr2, attn256_kernel, 160 * 1024, "sdc_attn[mfma 256]");
        hipLaunchKernelGGL(attn256_kernel, dim3((unsigned)(outer * inner * heads)), dim3(NT), ldsb, sdc::as_stream(stream), a);
        return sdc::check_launch("sdc_attn[mfma 256]");
    }
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
    static std::atomic<uint64_t> attr_done{0};
    SDC_LDS_OPTIN(attr_done, attn_kernel, 160 * 1024, "sdc_attn");
    hipLaunchKernelGGL(attn_kernel, dim3((unsigned)(ngrp * heads)), dim3(NT), lds_bytes, sdc::as_stream(stream), a);
    return sdc::check_launch("sdc_attn");
}
