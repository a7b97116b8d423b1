// Backward (VJP) kernels of the fine-tuning path (p_losses + loss.backward() of the three U-Nets,
// 1D/model/diffusion.py:638-733, 2d/ddpm/diffusion_2d.py:434-452; callers 1D/inference/inference_ft.py:183-187,
// tokamak/inference/pipeline.py:238-263, 2d/inference_2d.py:267-279).
//
//   * conv data gradients need no kernel of their own: they are convolutions with flipped / transposed taps and run on the
//     forward kernels (sdc_conv, Winograd forms included) with re-packed weights;
//   * sdc_conv_wgrad: the weight (and bias) gradient of every conv / Linear / transposed conv, an fp32-MFMA GEMM over the
//     positions,  dW[m][n][tap] = sum_{b,pos} G[b][m][pos] X[b][n][pos*s - p + tap];
//   * sdc_gn_silu_bwd: GroupNorm -> (scale + 1, shift) -> SiLU (+ residual) backward, two HBM passes;
//   * sdc_chan_norm_bwd, sdc_act_bwd, sdc_sumpool2: channel LayerNorm / RMSNorm, SiLU / GELU and nearest-upsample VJPs.
#include "sdc_common.h"

namespace {

constexpr int NT = 256;
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------ weight gradient
// GEMM  C[m][n] (per tap) = sum_k A[m][k] B[k][n]  with k = output positions: A = G rows (k contiguous), B = X rows shifted
// by the tap.  Workgroup = 4 waves (2 x 2), 64 m x 64 n x the KW taps of one (kd, kh); the K walk covers the output rows
// (b, od, oh) of this split in chunks of BKP = 64 / 32 / 16 positions along W.  Per chunk the G tile [64][BKP] and the X tile
// [64][(BKP - 1) SW + KW]
// (one input row segment: it serves all KW taps) go through LDS; a wave issues KW MFMAs per k pair from one ds_read_b32 of A
// and KW of B.  Chunk c+1 is fetched to registers while chunk c computes (two LDS buffers, one barrier per chunk).
// Partial results of the splits are written to the workspace and summed in a fixed order (deterministic, no atomics).
struct WgradArgs {
    SdcWgradDesc d;
    const float* g;
    const float* x;
    float* part;            // [nsplit][M][N][kD][kH][kW]
    float* bpart;           // [nsplit][M] or null
    int rows_per_split;     // output rows (b, od, oh) per split
    int nsplit, Mt, Nt;
    int lgD, lgH, lgW;      // log2 of the nearest-upsample factors of X
    int nch;                // wgrad_mkh_kernel: column chunks of a row (grid.z); the partial copies are [nsplit * nch]
};

template <int KW, int SW, int BKP, bool FOLD = false>
__global__ __launch_bounds__(NT) void wgrad_kernel(const WgradArgs a) {
    // FOLD (stem convs: N * KW <= 64, e.g. 7 input channels x 7 taps): the GEMM column is the pair (input channel, tap kw), so a
    // k pair costs ONE MFMA per wave instead of KW mostly-empty ones
    // BKP = positions per chunk (64 where the row length allows: 96 MFMAs per chunk and wave at KW = 3 cover the latency of
    // the next chunk's global loads; 16 for the short rows of the deep levels)
    constexpr int SPAN = (BKP - 1) * SW + KW;     // input columns under a chunk
    constexpr int AP = BKP + 1;                   // LDS pitches (odd: conflict-free column reads)
    constexpr int BP = SPAN | 1;
    constexpr int NBL = (64 * SPAN + NT - 1) / NT;        // X elements per thread and chunk
    constexpr int GPT = BKP / 4;                  // G positions per thread (thread = row tid >> 2, quarter tid & 3)
    extern __shared__ __attribute__((aligned(16))) float wlds[];
    float* const As = wlds;                       // [2][64 * AP]
    float* const Bs = wlds + 2 * 64 * AP;         // [2][64 * BP]
    const SdcWgradDesc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
    // tile: (m tile, n tile, kd, kh)
    int tb = blockIdx.x;
    const int kh = tb % d.kH; tb /= d.kH;
    const int kd = tb % d.kD; tb /= d.kD;
    const int nt = tb % a.Nt;
    const int mt = tb / a.Nt;
    const int m0 = mt * 64, n0 = nt * 64;
    const int split = blockIdx.y;
    const int R = d.B * d.oD * d.oH;
    const int r_lo = split * a.rows_per_split;
    const int r_hi = min(R, r_lo + a.rows_per_split);
    const int chunks_per_row = (d.oW + BKP - 1) / BKP;      // a ragged last chunk is zero-filled

    // G fetch: thread -> (m = tid >> 2, positions GPT (tid & 3) .. + GPT - 1)
    const int gm = tid >> 2, gq = tid & 3;
    const bool gm_ok = m0 + gm < d.M;
    const int64_t g_moff = (int64_t)(gm_ok ? m0 + gm : 0) * d.gs[1];
    // (16-byte loads need every row start 16-byte aligned: the strides; a thread's run starts at a multiple of 4 positions)
    const bool gvec = d.gs[4] == 1 && ((d.gs[0] | d.gs[1] | d.gs[2] | d.gs[3]) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.g) & 15) == 0;
    // X fetch: element e = tid + NT i -> (n = e / SPAN, j = e % SPAN)
    int xn[NBL], xj[NBL];
    int64_t x_noff[NBL];
    bool xn_ok[NBL];
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
        const int e = tid + NT * i;
        xn[i] = e / SPAN;
        xj[i] = e - xn[i] * SPAN;
        xn_ok[i] = e < 64 * SPAN && n0 + xn[i] < d.N;
        x_noff[i] = (int64_t)(xn_ok[i] ? n0 + xn[i] : 0) * d.xs[1];
    }
    const int iWu = d.iW << a.lgW, iHu = d.iH << a.lgH, iDu = d.iD << a.lgD;

    constexpr int NACC = FOLD ? 1 : KW;
    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    // FOLD: this lane's column n' = wn 32 + l31 -> (channel fci, tap fkw)
    const int fcol = wn * 32 + l31, fci = fcol / KW, fkw = fcol - fci * KW;
    const bool fok = fci < d.N;
    float bsum = 0.0f;
    const bool want_bias = a.bpart != nullptr && nt == 0 && kd == 0 && kh == 0;

    float greg[GPT], xreg[NBL];
    // chunk walk: (row r, chunk c); rows whose input row (id, ih) falls outside X contribute nothing to this (kd, kh) -- but
    // still feed the bias sum
    int r = r_lo, c = 0;
    auto fetch = [&](bool& valid_out) {
        // decompose the row (uniform)
        const int oh = r % d.oH;
        const int q = r / d.oH;
        const int od = q % d.oD;
        const int b = q / d.oD;
        const int idu = od * d.sD - d.pD + kd, ihu = oh * d.sH - d.pH + kh;
        const bool rv = idu >= 0 && idu < iDu && ihu >= 0 && ihu < iHu;
        valid_out = rv;
        const int ow0 = c * BKP;
        const int p0 = ow0 + GPT * gq;                       // this thread's first position
        const float* gp = a.g + (int64_t)b * d.gs[0] + (int64_t)od * d.gs[2] + (int64_t)oh * d.gs[3] + g_moff + (int64_t)p0 * d.gs[4];
        if (rv || want_bias) {
            const int left = gm_ok ? d.oW - p0 : 0;          // positions of this thread inside the row
            if (gvec && left >= GPT) {
#pragma unroll
                for (int i = 0; i < GPT / 4; ++i) {
                    const float4 v = reinterpret_cast<const float4*>(gp)[i];
                    greg[4 * i] = v.x; greg[4 * i + 1] = v.y; greg[4 * i + 2] = v.z; greg[4 * i + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int i = 0; i < GPT; ++i) greg[i] = i < left ? gp[(int64_t)i * d.gs[4]] : 0.0f;
            }
        }
        if (rv) {
            const float* xp = a.x + (int64_t)b * d.xs[0] + (int64_t)(idu >> a.lgD) * d.xs[2] + (int64_t)(ihu >> a.lgH) * d.xs[3];
            const int iw0 = ow0 * SW - d.pW;
#pragma unroll
            for (int i = 0; i < NBL; ++i) {
                const int iwu = iw0 + xj[i];
                const bool ok = xn_ok[i] && iwu >= 0 && iwu < iWu;
                xreg[i] = ok ? xp[x_noff[i] + (int64_t)(iwu >> a.lgW) * d.xs[4]] : 0.0f;
            }
        }
    };
    auto park = [&](int buf, bool valid) {
        if (want_bias) {
#pragma unroll
            for (int i = 0; i < GPT; ++i) bsum += greg[i];
        }
        if (!valid) return;
        float* Ab = As + buf * (64 * AP);
        float* Bb = Bs + buf * (64 * BP);
#pragma unroll
        for (int i = 0; i < GPT; ++i) Ab[gm * AP + GPT * gq + i] = greg[i];
#pragma unroll
        for (int i = 0; i < NBL; ++i)
            if (tid + NT * i < 64 * SPAN) Bb[xn[i] * BP + xj[i]] = xreg[i];
    };
    auto advance = [&]() { if (++c == chunks_per_row) { c = 0; ++r; } };

    bool v_cur = false, v_nxt = false;
    if (r < r_hi) {
        fetch(v_cur);
        park(0, v_cur);
        advance();
    }
    __syncthreads();
    int buf = 0;
    while (true) {
        const bool more = r < r_hi;
        if (more) fetch(v_nxt);
        if (v_cur) {
            const float* Ab = As + buf * (64 * AP) + (wm * 32 + l31) * AP + lh;
            if constexpr (FOLD) {
                const float* Bb = Bs + buf * (64 * BP) + (fok ? fci : 0) * BP + fkw + lh * SW;
#pragma unroll 8
                for (int kk = 0; kk < BKP / 2; ++kk)
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ab[2 * kk], fok ? Bb[2 * kk * SW] : 0.0f, acc[0], 0, 0, 0);
            } else {
                const float* Bb = Bs + buf * (64 * BP) + (wn * 32 + l31) * BP + lh * SW;
#pragma unroll 8
                for (int kk = 0; kk < BKP / 2; ++kk) {
                    const float av = Ab[2 * kk];
#pragma unroll
                    for (int t = 0; t < KW; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, Bb[2 * kk * SW + t], acc[t], 0, 0, 0);
                }
            }
        }
        if (!more) break;
        park(buf ^ 1, v_nxt);
        advance();
        __syncthreads();
        buf ^= 1;
        v_cur = v_nxt;
    }

    // ---- partial sums of this split: part[split][m][n][kd][kh][kw]
    const int taps = d.kD * d.kH * KW;
    float* P = a.part + (int64_t)split * d.M * d.N * taps;
    const int n = n0 + wn * 32 + l31;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
        const int m = m0 + wm * 32 + 8 * (rr >> 2) + 4 * lh + (rr & 3);
        if constexpr (FOLD) {
            if (m < d.M && fok) P[((int64_t)m * d.N + fci) * taps + (kd * d.kH + kh) * KW + fkw] = acc[0][rr];
        } else if (m < d.M && n < d.N) {
            float* o = P + ((int64_t)m * d.N + n) * taps + (kd * d.kH + kh) * KW;
#pragma unroll
            for (int t = 0; t < KW; ++t) o[t] = acc[t][rr];
        }
    }
    if (want_bias) {
        // the four threads of a G row sit in adjacent lanes
        bsum += __shfl_xor(bsum, 1, 64);
        bsum += __shfl_xor(bsum, 2, 64);
        if (gq == 0 && gm_ok) a.bpart[(int64_t)split * d.M + m0 + gm] = bsum;
    }
}

template <int KW, int SW, int BKP, bool FOLD = false>
int launch_wgrad(const WgradArgs& a, dim3 grid, hipStream_t s) {
    constexpr int SPAN = (BKP - 1) * SW + KW;
    const size_t lds = (size_t)2 * 64 * ((BKP + 1) + (SPAN | 1)) * sizeof(float);
    static std::atomic<uint64_t> attr{0};
    SDC_LDS_OPTIN(attr, (wgrad_kernel<KW, SW, BKP, FOLD>), 160 * 1024, "sdc_conv_wgrad");
    hipLaunchKernelGGL((wgrad_kernel<KW, SW, BKP, FOLD>), grid, dim3(NT), lds, s, a);
    return SDC_OK;
}

template <int KW, int SW>
int launch_wgrad_bkp(const WgradArgs& a, dim3 grid, hipStream_t s) {
    if constexpr (KW == 7) {
        if (a.d.N * KW <= 64) {                 // stem convs (7 or 3 input channels)
            if (a.d.oW % 64 == 0) return launch_wgrad<KW, SW, 64, true>(a, grid, s);
            if (a.d.oW % 32 == 0) return launch_wgrad<KW, SW, 32, true>(a, grid, s);
            return launch_wgrad<KW, SW, 16, true>(a, grid, s);
        }
    }
    if (a.d.oW % 64 == 0) return launch_wgrad<KW, SW, 64>(a, grid, s);
    if (a.d.oW % 32 == 0) return launch_wgrad<KW, SW, 32>(a, grid, s);
    return launch_wgrad<KW, SW, 16>(a, grid, s);
}

// KH x KW (x kD) stride-1 'same' convs (3 x 3, and the 7 x 7 stems in the tap-folded form): one workgroup takes all KH row taps of
// a (kd, m tile, n tile) together.  Output row oh needs the input rows oh - pH .. oh + pH: walking the rows of a (b, od) block in
// order, every G row and every X row is fetched ONCE and kept in an LDS ring (slot = row & (size - 1)) while it serves KH output
// rows -- 1/KH of the fabric reads of the per-tap kernel above, which at 64 channels (one m and one n tile: nothing else
// amortises the loads) is what bounds it; at the stem (7 input channels) a step now holds 7 x 32 MFMAs per wave instead of 32.
// Step t of a segment [oh0, oh1) of a block (t = oh0 - pH .. oh1 - 1 + pH): park X row t and G row t - pH, run the MFMAs of
// output row t - pH.
template <int KH, int KW, int BKP, bool FOLD>
__global__ __launch_bounds__(NT) void wgrad_mkh_kernel(const WgradArgs a) {
    constexpr int PH = KH / 2;
    constexpr int XS = KH == 3 ? 4 : 8;           // X ring: rows t - 2 PH .. t in use while row t + 1 is parked
    constexpr int XC = FOLD ? 16 : 64;            // X channels per slot (FOLD: N KW <= 64, so N <= 9)
    constexpr int SPAN = BKP - 1 + KW, AP = BKP + 1, BP = SPAN | 1;
    constexpr int NBL = (XC * SPAN + NT - 1) / NT, GPT = BKP / 4;
    constexpr int ASZ = 64 * AP, BSZ = XC * BP;
    // WINO (the 3 x 3 form): the three kw taps of a row tap as a Winograd F(3,2) correlation over pairs of positions -- the weight
    // gradient of F(2,3):  dW = G^T [ (A dY) (.) (B^T d) ]  summed over the tiles, dY = (y0, y1) the two gradients of a tile, d its
    // four inputs;  A dY = (y0, y0 + y1, y0 - y1, -y1),  B^T d = (d0 - d2, d1 + d2, d2 - d1, d1 - d3),
    // dW = (M0 + (M1 + M2)/2, (M1 - M2)/2, (M1 + M2)/2 + M3).  Four MFMAs per tile pair instead of six (2/3 of the matrix work);
    // the operands are formed from the raw LDS rows as they are read (one add per MFMA), the 4 -> 3 reduction once at the end.
    // (rows of 16 positions keep the direct form: four tile pairs per step leave the per-step costs uncovered, measured 11 % slower)
    constexpr bool WINO = !FOLD && KW == 3 && BKP >= 32;
    constexpr int NACC = FOLD ? KH : (WINO ? KH * 4 : KH * KW);
    extern __shared__ __attribute__((aligned(16))) float wlds[];
    float* const As = wlds;                       // [4][ASZ]
    float* const Bs = wlds + 4 * ASZ;             // [XS][BSZ]
    const SdcWgradDesc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
    int tb = blockIdx.x;
    const int kd = tb % d.kD; tb /= d.kD;
    const int nt = tb % a.Nt;
    const int mt = tb / a.Nt;
    const int m0 = mt * 64, n0 = nt * 64;
    const int split = blockIdx.y;
    const int c0 = (int)blockIdx.z * BKP;         // rows longer than one chunk (BKP = 64): chunk blockIdx.z is one more partial copy
    const int R = d.B * d.oD * d.oH;
    const int r_lo = split * a.rows_per_split;
    const int r_hi = min(R, r_lo + a.rows_per_split);

    const int gm = tid >> 2, gq = tid & 3;
    const bool gm_ok = m0 + gm < d.M;
    const int64_t g_moff = (int64_t)(gm_ok ? m0 + gm : 0) * d.gs[1];
    const bool gvec = d.gs[4] == 1 && ((d.gs[0] | d.gs[1] | d.gs[2] | d.gs[3]) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.g) & 15) == 0;
    int xn[NBL], xj[NBL];
    int64_t x_noff[NBL];
    bool xn_ok[NBL];
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
        const int e = tid + NT * i;
        xn[i] = e / SPAN;
        xj[i] = e - xn[i] * SPAN;
        xn_ok[i] = e < XC * SPAN && n0 + xn[i] < d.N;
        x_noff[i] = (int64_t)(xn_ok[i] ? n0 + xn[i] : 0) * d.xs[1];
    }
    const int iWu = d.iW << a.lgW, iDu = d.iD << a.lgD;

    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    // FOLD: this lane's column n' = wn 32 + l31 -> (channel fci, tap fkw)
    const int fcol = wn * 32 + l31, fci = fcol / KW, fkw = fcol - fci * KW;
    const bool fok = fci < d.N;
    float bsum = 0.0f;
    const bool want_bias = a.bpart != nullptr && nt == 0 && kd == 0;

    // walk: (block blk = (b, od), segment [oh0, oh1) of it inside this split, step t)
    struct Step { int blk, oh0, oh1, t; };
    auto open_block = [&](int blk, Step& st) {
        st.blk = blk;
        st.oh0 = max(r_lo - blk * d.oH, 0);
        st.oh1 = min(r_hi - blk * d.oH, d.oH);
        st.t = st.oh0 - PH;
    };
    auto next_step = [&](Step& st) -> bool {            // false when the split is done
        if (st.t < st.oh1 - 1 + PH) { ++st.t; return true; }
        if ((st.blk + 1) * d.oH >= r_hi) return false;
        open_block(st.blk + 1, st);
        return true;
    };
    auto blk_valid = [&](const Step& st, int& b, int& od, int& idu) {
        od = st.blk % d.oD; b = st.blk / d.oD;
        idu = od * d.sD - d.pD + kd;
        return idu >= 0 && idu < iDu;
    };

    float greg[GPT], xreg[NBL];
    auto fetch = [&](const Step& st) {
        int b, od, idu;
        const bool bv = blk_valid(st, b, od, idu);
        const int grow = st.t - PH;
        const bool gneed = grow >= st.oh0 && grow < st.oh1 && (bv || want_bias);
        const bool xneed = bv && st.t >= 0 && st.t < d.iH;
        if (gneed) {
            const int p0 = c0 + GPT * gq;
            const float* gp = a.g + (int64_t)b * d.gs[0] + (int64_t)od * d.gs[2] + (int64_t)grow * d.gs[3] + g_moff + (int64_t)p0 * d.gs[4];
            const int left = gm_ok ? d.oW - p0 : 0;
            if (gvec && left >= GPT) {
#pragma unroll
                for (int i = 0; i < GPT / 4; ++i) {
                    const float4 v = reinterpret_cast<const float4*>(gp)[i];
                    greg[4 * i] = v.x; greg[4 * i + 1] = v.y; greg[4 * i + 2] = v.z; greg[4 * i + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int i = 0; i < GPT; ++i) greg[i] = i < left ? gp[(int64_t)i * d.gs[4]] : 0.0f;
            }
        }
        if (xneed) {
            const float* xp = a.x + (int64_t)b * d.xs[0] + (int64_t)(idu >> a.lgD) * d.xs[2] + (int64_t)st.t * d.xs[3];
#pragma unroll
            for (int i = 0; i < NBL; ++i) {
                const int iwu = c0 + xj[i] - d.pW;
                const bool ok = xn_ok[i] && iwu >= 0 && iwu < iWu;
                xreg[i] = ok ? xp[x_noff[i] + (int64_t)(iwu >> a.lgW) * d.xs[4]] : 0.0f;
            }
        }
    };
    auto park = [&](const Step& st) {
        int b, od, idu;
        const bool bv = blk_valid(st, b, od, idu);
        const int grow = st.t - PH;
        const bool gneed = grow >= st.oh0 && grow < st.oh1 && (bv || want_bias);
        const bool xneed = bv && st.t >= 0 && st.t < d.iH;
        if (gneed) {
            if (want_bias) {
#pragma unroll
                for (int i = 0; i < GPT; ++i) bsum += greg[i];
            }
            float* Ab = As + (grow & 3) * ASZ;
#pragma unroll
            for (int i = 0; i < GPT; ++i) Ab[gm * AP + GPT * gq + i] = greg[i];
        }
        if (xneed) {
            float* Bb = Bs + (st.t & (XS - 1)) * BSZ;
#pragma unroll
            for (int i = 0; i < NBL; ++i)
                if (tid + NT * i < XC * SPAN) Bb[xn[i] * BP + xj[i]] = xreg[i];
        }
    };
    auto compute = [&](const Step& st) {
        int b, od, idu;
        const int oh = st.t - PH;
        if (!(oh >= st.oh0 && oh < st.oh1) || !blk_valid(st, b, od, idu)) return;
        const float* Ab = As + (oh & 3) * ASZ + (wm * 32 + l31) * AP + lh;
        const int boff = FOLD ? (fok ? fci : 0) * BP + fkw + lh : (wn * 32 + l31) * BP + lh;
        bool v[KH];
        const float* Bk[KH];
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
            const int ih = oh - PH + kh;
            v[kh] = ih >= 0 && ih < d.iH;
            Bk[kh] = Bs + (ih & (XS - 1)) * BSZ + boff;
        }
        if constexpr (WINO) {
            // lane (l31, lh) feeds tile 2 kk + lh: positions 4 kk + 2 lh, + 1 of the G row; inputs from the same index on (the X
            // row starts one column left of position 0)
#pragma unroll 2
            for (int kk = 0; kk < BKP / 4; ++kk) {
                const float y0 = Ab[4 * kk + lh], y1 = Ab[4 * kk + lh + 1];      // Ab already holds + lh: index 4 kk + 2 lh
                const float a1 = y0 + y1, a2 = y0 - y1, a3 = -y1;
#pragma unroll
                for (int kh = 0; kh < KH; ++kh) {
                    if (!v[kh]) continue;
                    const float* x = Bk[kh] + 4 * kk + lh;                       // Bk holds + lh as well
                    const float d0 = x[0], d1 = x[1], d2 = x[2], d3 = x[3];
                    acc[kh * 4 + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(y0, d0 - d2, acc[kh * 4 + 0], 0, 0, 0);
                    acc[kh * 4 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, d1 + d2, acc[kh * 4 + 1], 0, 0, 0);
                    acc[kh * 4 + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, d2 - d1, acc[kh * 4 + 2], 0, 0, 0);
                    acc[kh * 4 + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, d1 - d3, acc[kh * 4 + 3], 0, 0, 0);
                }
            }
        } else {
#pragma unroll 4
        for (int kk = 0; kk < BKP / 2; ++kk) {
            const float av = Ab[2 * kk];
#pragma unroll
            for (int kh = 0; kh < KH; ++kh) {
                if (!v[kh]) continue;
                if constexpr (FOLD) {
                    acc[kh] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, fok ? Bk[kh][2 * kk] : 0.0f, acc[kh], 0, 0, 0);
                } else {
#pragma unroll
                    for (int t = 0; t < KW; ++t)
                        acc[kh * KW + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, Bk[kh][2 * kk + t], acc[kh * KW + t], 0, 0, 0);
                }
            }
        }
        }
    };

    if (r_lo < r_hi) {
        Step cur, nxt;
        open_block(r_lo / d.oH, cur);
        fetch(cur);
        park(cur);
        __syncthreads();
        while (true) {
            nxt = cur;
            const bool more = next_step(nxt);
            if (more) fetch(nxt);
            compute(cur);
            if (!more) break;
            if (nxt.blk != cur.blk) __syncthreads();     // a new block's first slot may be one this step still reads
            park(nxt);
            __syncthreads();
            cur = nxt;
        }
    }

    const int taps = d.kD * KH * KW;
    const int copy = split * a.nch + (int)blockIdx.z;
    float* P = a.part + (int64_t)copy * d.M * d.N * taps;
    const int n = n0 + wn * 32 + l31;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
        const int m = m0 + wm * 32 + 8 * (rr >> 2) + 4 * lh + (rr & 3);
        if constexpr (FOLD) {
            if (m < d.M && fok) {
                float* o = P + ((int64_t)m * d.N + fci) * taps + kd * KH * KW + fkw;
#pragma unroll
                for (int kh = 0; kh < KH; ++kh) o[kh * KW] = acc[kh][rr];
            }
        } else if (m < d.M && n < d.N) {
            float* o = P + ((int64_t)m * d.N + n) * taps + kd * KH * KW;
            if constexpr (WINO) {
#pragma unroll
                for (int kh = 0; kh < KH; ++kh) {
                    const float m0v = acc[kh * 4][rr], m1v = acc[kh * 4 + 1][rr], m2v = acc[kh * 4 + 2][rr], m3v = acc[kh * 4 + 3][rr];
                    const float hs = 0.5f * (m1v + m2v);
                    o[kh * 3] = m0v + hs; o[kh * 3 + 1] = 0.5f * (m1v - m2v); o[kh * 3 + 2] = hs + m3v;
                }
            } else {
#pragma unroll
                for (int t = 0; t < KH * KW; ++t) o[t] = acc[t][rr];
            }
        }
    }
    if (want_bias) {
        bsum += __shfl_xor(bsum, 1, 64);
        bsum += __shfl_xor(bsum, 2, 64);
        if (gq == 0 && gm_ok) a.bpart[(int64_t)copy * d.M + m0 + gm] = bsum;
    }
}

// the merged-kh form applies to square stride-1 'same' taps whose rows are one chunk: 3 x 3, and 7 x 7 with N * 7 <= 64 (stems)
bool wgrad_mkh_ok(const SdcWgradDesc& d) {
    const bool k3 = d.kW == 3 && d.kH == 3 && d.pH == 1 && d.pW == 1;
    const bool k7 = d.kW == 7 && d.kH == 7 && d.pH == 3 && d.N * 7 <= 64;
    return (k3 || k7) && d.sW == 1 && d.sH == 1 && d.uH == 1 && d.iH == d.oH &&
           (d.oW == 16 || d.oW == 32 || d.oW == 64 || (k3 && d.oW > 64 && d.oW % 64 == 0 && d.oW <= 1024));
}
// rows longer than 64 positions go through the merged-kh kernel in chunks of 64 columns: each chunk is one more partial copy
int wgrad_chunks(const SdcWgradDesc& d) { return wgrad_mkh_ok(d) && d.oW > 64 ? d.oW / 64 : 1; }

template <int KH, int KW, int BKP, bool FOLD>
int launch_wgrad_mkh(const WgradArgs& a, dim3 grid, hipStream_t s) {
    constexpr int XS = KH == 3 ? 4 : 8, XC = FOLD ? 16 : 64;
    const size_t lds = ((size_t)4 * 64 * (BKP + 1) + (size_t)XS * XC * ((BKP - 1 + KW) | 1)) * sizeof(float);
    static std::atomic<uint64_t> attr{0};
    SDC_LDS_OPTIN(attr, (wgrad_mkh_kernel<KH, KW, BKP, FOLD>), 160 * 1024, "sdc_conv_wgrad");
    hipLaunchKernelGGL((wgrad_mkh_kernel<KH, KW, BKP, FOLD>), grid, dim3(NT), lds, s, a);
    return SDC_OK;
}

template <int KH, int KW, bool FOLD>
int launch_wgrad_mkh_bkp(const WgradArgs& a, dim3 grid, hipStream_t s) {
    if (a.d.oW % 64 == 0) return launch_wgrad_mkh<KH, KW, 64, FOLD>(a, grid, s);
    if (a.d.oW == 32) return launch_wgrad_mkh<KH, KW, 32, FOLD>(a, grid, s);
    return launch_wgrad_mkh<KH, KW, 16, FOLD>(a, grid, s);
}

// out[i] = sum_s part[s][i] in a fixed order.  KG = 1: a thread per element, splits in sequence (loads four deep).  KG = 4 (many
// splits over a small gradient): wave g of the workgroup sums the splits s = g (mod 4) of 64 elements, wave 0 adds the four
// partial sums in order -- the chain of dependent HBM latencies is what this reduction costs, not its bytes.
// A second table (the bias partials of the same weight gradient) rides in the same launch: workgroups nb1 .. take it.
template <int KG>
__global__ __launch_bounds__(NT) void sum_splits_kernel(const float* __restrict__ part, float* __restrict__ out, int64_t n, int nsplit,
                                                        const float* __restrict__ part2, float* __restrict__ out2, int64_t n2, unsigned nb1) {
    const int g = KG == 1 ? 0 : (int)(threadIdx.x >> 6);
    unsigned blk = blockIdx.x;
    if (blk >= nb1) { blk -= nb1; part = part2; out = out2; n = n2; }
    const int64_t i = KG == 1 ? (int64_t)blk * NT + threadIdx.x : (int64_t)blk * 64 + (threadIdx.x & 63);
    float s = 0.0f;
    if (i < n) {
        int k = g;
        for (; k + 7 * KG < nsplit; k += 8 * KG) {             // eight loads in flight, added in split order (as below)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(k + u * KG) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k + 3 * KG < nsplit; k += 4 * KG) {
            const float v0 = part[(int64_t)k * n + i], v1 = part[(int64_t)(k + KG) * n + i];
            const float v2 = part[(int64_t)(k + 2 * KG) * n + i], v3 = part[(int64_t)(k + 3 * KG) * n + i];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; k < nsplit; k += KG) s += part[(int64_t)k * n + i];
    }
    if (KG == 1) {
        if (i < n) out[i] = s;
    } else {
        __shared__ float sh[KG][64];
        sh[g][threadIdx.x & 63] = s;
        __syncthreads();
        if (g == 0 && i < n) {
            float t = sh[0][threadIdx.x];
            for (int q = 1; q < KG; ++q) t += sh[q][threadIdx.x];
            out[i] = t;
        }
    }
}

void launch_sum_splits(const float* part, float* out, int64_t n, int nsplit, hipStream_t s, const float* part2 = nullptr,
                       float* out2 = nullptr, int64_t n2 = 0) {
    const int per = nsplit >= 8 ? 64 : NT;
    const unsigned nb1 = (unsigned)((n + per - 1) / per), nb2 = part2 ? (unsigned)((n2 + per - 1) / per) : 0u;
    if (nsplit >= 8)
        hipLaunchKernelGGL(sum_splits_kernel<4>, dim3(nb1 + nb2), dim3(NT), 0, s, part, out, n, nsplit, part2, out2, n2, nb1);
    else
        hipLaunchKernelGGL(sum_splits_kernel<1>, dim3(nb1 + nb2), dim3(NT), 0, s, part, out, n, nsplit, part2, out2, n2, nb1);
}

int wgrad_splits(const SdcWgradDesc& d, int* rows_per_split) {
    const int Mt = (d.M + 63) / 64, Nt = (d.N + 63) / 64;
    const int nch = wgrad_chunks(d);
    const int64_t tiles = (int64_t)Mt * Nt * d.kD * (wgrad_mkh_ok(d) ? 1 : d.kH) * nch;
    const int R = d.B * d.oD * d.oH;
    // splits: ~3 workgroups per CU in flight (a layer with few tiles and long rows is otherwise a handful of workgroups); every
    // split writes, and the reduction re-reads, a full copy of the gradient, at HBM rate: cheap beside idle CUs, bounded at 256 MB
    // (the merged-kh kernel on rows of >= 64 positions holds 135 KB of LDS -- one workgroup per CU: one round of 256 workgroups,
    // rather than three rounds of short ones that each pay the ring's fill and a partial copy; rows of 32: two per CU)
    const int64_t target = wgrad_mkh_ok(d) && d.kH == 3 ? (d.oW >= 64 ? 256 : (d.oW == 32 ? 512 : 768)) : 768;
    int64_t want = (target + tiles - 1) / tiles;
    const int64_t nw = (int64_t)d.M * d.N * d.kD * d.kH * d.kW;
    const int64_t cap = (64ll << 20) / (nw * nch);
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    if (want > R) want = R;
    const int rps = (int)((R + want - 1) / want);
    *rows_per_split = rps;
    return (R + rps - 1) / rps;
}

int ilog2_12(int v) { return v == 1 ? 0 : (v == 2 ? 1 : -1); }

}  // namespace

extern "C" size_t sdc_conv_wgrad_bytes(const SdcWgradDesc* dp) {
    if (!dp) return 0;
    int rps = 0;
    const int ns = wgrad_splits(*dp, &rps);
    const size_t taps = (size_t)dp->kD * dp->kH * dp->kW;
    const size_t nc = (size_t)ns * wgrad_chunks(*dp);
    return (nc * dp->M * dp->N * taps + nc * dp->M) * sizeof(float);
}

extern "C" int sdc_conv_wgrad(const SdcWgradDesc* dp, const float* g, const float* x, float* dw, float* dbias, void* work,
                              size_t work_bytes, void* stream) {
    SDC_REQUIRE(dp && g && x && dw && work, SDC_ENULL, "sdc_conv_wgrad: null pointer");
    const SdcWgradDesc& d = *dp;
    SDC_REQUIRE(d.B > 0 && d.M > 0 && d.N > 0 && d.oD > 0 && d.oH > 0 && d.oW > 0 && d.iD > 0 && d.iH > 0 && d.iW > 0, SDC_EINVAL,
                "sdc_conv_wgrad: bad sizes");
    SDC_REQUIRE(work_bytes >= sdc_conv_wgrad_bytes(dp), SDC_EINVAL, "sdc_conv_wgrad: workspace too small");
    SDC_REQUIRE((int64_t)d.B * d.oD * d.oH < (1ll << 31), SDC_EINVAL, "sdc_conv_wgrad: too many output rows");
    WgradArgs a;
    a.d = d; a.g = g; a.x = x;
    a.lgD = ilog2_12(d.uD); a.lgH = ilog2_12(d.uH); a.lgW = ilog2_12(d.uW);
    SDC_REQUIRE(a.lgD >= 0 && a.lgH >= 0 && a.lgW >= 0, SDC_EINVAL, "sdc_conv_wgrad: upsample factors must be 1 or 2");
    a.nsplit = wgrad_splits(d, &a.rows_per_split);
    a.Mt = (d.M + 63) / 64; a.Nt = (d.N + 63) / 64;
    const int64_t nw = (int64_t)d.M * d.N * d.kD * d.kH * d.kW;
    // one split: the workgroups write the gradient itself (no partial copy, no reduction pass)
    a.nch = wgrad_chunks(d);
    const int ncopy = a.nsplit * a.nch;
    a.part = ncopy == 1 ? dw : static_cast<float*>(work);
    a.bpart = dbias ? (ncopy == 1 ? dbias : static_cast<float*>(work) + (int64_t)ncopy * nw) : nullptr;
    const bool mkh = wgrad_mkh_ok(d);
    const int64_t tiles = (int64_t)a.Mt * a.Nt * d.kD * (mkh ? 1 : d.kH);
    SDC_REQUIRE(tiles < (1ll << 31) && a.nsplit < 65536, SDC_EINVAL, "sdc_conv_wgrad: grid too large");
    dim3 grid((unsigned)tiles, (unsigned)a.nsplit, (unsigned)a.nch);
    hipStream_t s = sdc::as_stream(stream);
    int lrc;
    if (mkh) lrc = d.kW == 3 ? launch_wgrad_mkh_bkp<3, 3, false>(a, grid, s) : launch_wgrad_mkh_bkp<7, 7, true>(a, grid, s);
    else if (d.kW == 1 && d.sW == 1) lrc = launch_wgrad_bkp<1, 1>(a, grid, s);
    else if (d.kW == 3 && d.sW == 1) lrc = launch_wgrad_bkp<3, 1>(a, grid, s);
    else if (d.kW == 7 && d.sW == 1) lrc = launch_wgrad_bkp<7, 1>(a, grid, s);
    else if (d.kW == 4 && d.sW == 2) lrc = launch_wgrad_bkp<4, 2>(a, grid, s);
    else if (d.kW == 2 && d.sW == 2) lrc = launch_wgrad_bkp<2, 2>(a, grid, s);
    else {
        sdc::set_error("sdc_conv_wgrad: tap / stride combination (kW %d, sW %d) not built (1/1, 3/1, 7/1, 4/2, 2/2)", d.kW, d.sW);
        return SDC_EINVAL;
    }
    if (lrc) return lrc;
    int rc = sdc::check_launch("sdc_conv_wgrad");
    if (rc || ncopy == 1) return rc;
    {
        launch_sum_splits(a.part, dw, nw, ncopy, s, dbias ? a.bpart : nullptr, dbias, (int64_t)d.M);
    }
    return sdc::check_launch("sdc_conv_wgrad[reduce]");
}

// ------------------------------------------------------------------------------------------------ GroupNorm + SiLU backward
// forward (sdc_gn_apply):  xh = (h - mean) rstd;  u = xh gamma + beta;  v = u (1 + sc) + sh;  y = silu(v) (+ res)
// pass 1 (per (b, c) row): A1 = sum gy silu'(v),  A2 = sum gy silu'(v) xh                        -> rows[b][c] = (A1, A2)
// pass 2 (per (b, c) row): k = gamma (1 + sc);  m1 = sum_{c in group} k A1 / n,  m2 = sum k A2 / n;
//                          gh = rstd (gy silu'(v) k - m1 - xh m2)
// The parameter gradients are small sums over rows[]:  d gamma[c] = sum_b (1 + sc) A2,  d beta[c] = sum_b (1 + sc) A1,
// d sc[b][c] = gamma A2 + beta A1,  d sh[b][c] = A1  (formed by the caller from the B x C x 2 table).
__device__ __forceinline__ float dsilu(float v) {
    const float s = 1.0f / (1.0f + __expf(-v));
    return s * (1.0f + v * (1.0f - s));
}

// rows kernel: a workgroup owns RPB (b, c) rows -- one for long rows (all 256 threads sweep it), four (one per wave) for the short
// rows of the 1-D nets (S < 1024: a workgroup per 16-element row would be 130 k nearly empty workgroups)
template <int RPB>
__global__ __launch_bounds__(NT) void gn_bwd_rows_kernel(const float* __restrict__ h, const float* __restrict__ gy,
                                                        const float* __restrict__ stats, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ ss,
                                                        int64_t ss_b_stride, float* __restrict__ rows, int nrows, int C, int G, int64_t S) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bc = RPB == 1 ? blockIdx.x : blockIdx.x * RPB + wave;
    const bool live = bc < nrows;
    double a1 = 0.0, a2 = 0.0;
    if (live) {
        const int b = bc / C, c = bc - b * C;
        const int g = c / (C / G);
        const float mean = stats[(b * G + g) * 2], rstd = stats[(b * G + g) * 2 + 1];
        float sc = 1.0f, sh = 0.0f;
        if (ss) { sc = ss[(int64_t)b * ss_b_stride + c] + 1.0f; sh = ss[(int64_t)b * ss_b_stride + C + c]; }
        const float ga = gamma[c], be = beta[c];
        const int64_t base = (int64_t)bc * S;
        const int i0 = RPB == 1 ? threadIdx.x : lane, step = RPB == 1 ? NT : 64;
        for (int64_t i = i0; i < S; i += step) {
            const float xh = (h[base + i] - mean) * rstd;
            const float v = (xh * ga + be) * sc + sh;
            const float gv = gy[base + i] * dsilu(v);
            a1 += (double)gv;
            a2 += (double)gv * (double)xh;
        }
    }
    a1 = sdc::wave_sum(a1);
    a2 = sdc::wave_sum(a2);
    if (RPB == 1) {
        __shared__ double shm[2][NT / 64];
        if (lane == 0) { shm[0][wave] = a1; shm[1][wave] = a2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double t1 = 0, t2 = 0;
            for (int w = 0; w < NT / 64; ++w) { t1 += shm[0][w]; t2 += shm[1][w]; }
            rows[(int64_t)bc * 2] = (float)t1;
            rows[(int64_t)bc * 2 + 1] = (float)t2;
        }
    } else if (live && lane == 0) {
        rows[(int64_t)bc * 2] = (float)a1;
        rows[(int64_t)bc * 2 + 1] = (float)a2;
    }
}

// long rows with few of them (3-D levels at a small batch: 256 rows of 131 072 elements): the row is cut into ysplit pieces, one
// workgroup each, 16-byte loads; the pieces' fp64 sums are added in order by gn_bwd_rows_finish_kernel
__global__ __launch_bounds__(NT) void gn_bwd_rows_part_kernel(const float* __restrict__ h, const float* __restrict__ gy,
                                                             const float* __restrict__ stats, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ ss,
                                                             int64_t ss_b_stride, double* __restrict__ part, int C, int G, int64_t S,
                                                             int64_t piece) {
    const int bc = blockIdx.x, y = blockIdx.y;
    const int b = bc / C, c = bc - b * C;
    const int g = c / (C / G);
    const float mean = stats[(b * G + g) * 2], rstd = stats[(b * G + g) * 2 + 1];
    float sc = 1.0f, sh = 0.0f;
    if (ss) { sc = ss[(int64_t)b * ss_b_stride + c] + 1.0f; sh = ss[(int64_t)b * ss_b_stride + C + c]; }
    const float ga = gamma[c], be = beta[c];
    const int64_t lo = (int64_t)y * piece, hi = lo + piece < S ? lo + piece : S;       // piece is a multiple of 4, S too
    const float4* h4 = reinterpret_cast<const float4*>(h + (int64_t)bc * S);
    const float4* g4 = reinterpret_cast<const float4*>(gy + (int64_t)bc * S);
    double a1 = 0.0, a2 = 0.0;
    for (int64_t i = lo / 4 + threadIdx.x; i < hi / 4; i += NT) {
        const float4 hv = h4[i], gv4 = g4[i];
        const float hh[4] = {hv.x, hv.y, hv.z, hv.w}, gg[4] = {gv4.x, gv4.y, gv4.z, gv4.w};
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float xh = (hh[q] - mean) * rstd;
            const float v = (xh * ga + be) * sc + sh;
            const float gv = gg[q] * dsilu(v);
            s1 += gv;
            s2 += gv * xh;
        }
        a1 += (double)s1;
        a2 += (double)s2;
    }
    a1 = sdc::wave_sum(a1);
    a2 = sdc::wave_sum(a2);
    __shared__ double shm[2][NT / 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { shm[0][wave] = a1; shm[1][wave] = a2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t1 = 0, t2 = 0;
        for (int w = 0; w < NT / 64; ++w) { t1 += shm[0][w]; t2 += shm[1][w]; }
        part[((int64_t)bc * gridDim.y + y) * 2] = t1;
        part[((int64_t)bc * gridDim.y + y) * 2 + 1] = t2;
    }
}

__global__ __launch_bounds__(NT) void gn_bwd_rows_finish_kernel(const double* __restrict__ part, float* __restrict__ rows, int nrows, int ysplit) {
    const int bc = blockIdx.x * NT + threadIdx.x;
    if (bc >= nrows) return;
    double t1 = 0, t2 = 0;
    for (int y = 0; y < ysplit; ++y) { t1 += part[((int64_t)bc * ysplit + y) * 2]; t2 += part[((int64_t)bc * ysplit + y) * 2 + 1]; }
    rows[(int64_t)bc * 2] = (float)t1;
    rows[(int64_t)bc * 2 + 1] = (float)t2;
}

// pieces per row for the kernel above (1: the row kernels below take the row whole)
int gn_bwd_ysplit(int nrows, int64_t S) {
    if (S < 16384 || S % 4 != 0 || nrows >= 1024) return 1;
    int64_t y = (1024 + nrows - 1) / nrows;
    if (y > S / 4096) y = S / 4096;
    return (int)(y < 1 ? 1 : (y > 64 ? 64 : y));
}

// group means of (k A1, k A2), k = gamma (1 + sc): one workgroup per (b, g) -> gstat[b][g] = (m1, m2)
__device__ __forceinline__ void gn_bwd_group_body(const int bg, const float* __restrict__ rows, const float* __restrict__ gamma,
                                                  const float* __restrict__ ss, int64_t ss_b_stride, float* __restrict__ gstat,
                                                  int C, int G, int64_t S) {
    const int b = bg / G, g = bg - b * G;
    const int cpg = C / G;
    double s1 = 0.0, s2 = 0.0;
    for (int cc = threadIdx.x; cc < cpg; cc += NT) {
        const int ch = g * cpg + cc;
        const float k = gamma[ch] * (ss ? ss[(int64_t)b * ss_b_stride + ch] + 1.0f : 1.0f);
        s1 += (double)k * rows[((int64_t)b * C + ch) * 2];
        s2 += (double)k * rows[((int64_t)b * C + ch) * 2 + 1];
    }
    __shared__ double shm[2][NT / 64];
    s1 = sdc::wave_sum(s1);
    s2 = sdc::wave_sum(s2);
    if ((threadIdx.x & 63) == 0) { shm[0][threadIdx.x >> 6] = s1; shm[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t1 = 0, t2 = 0;
        for (int w = 0; w < NT / 64; ++w) { t1 += shm[0][w]; t2 += shm[1][w]; }
        const double inv = 1.0 / ((double)cpg * (double)S);
        gstat[bg * 2] = (float)(t1 * inv);
        gstat[bg * 2 + 1] = (float)(t2 * inv);
    }
}

// apply: gh = rstd (gy silu'(v) k - m1 - xh m2).  FLAT: threads walk a flat element index and look the row constants up
// (short rows); otherwise grid.x = (b, c) row, grid.y walks the row.
template <bool FLAT>
__global__ __launch_bounds__(NT) void gn_bwd_apply_kernel(const float* __restrict__ h, const float* __restrict__ gy,
                                                         const float* __restrict__ stats, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ ss,
                                                         int64_t ss_b_stride, const float* __restrict__ gstat, float* __restrict__ gh,
                                                         int C, int G, int64_t S, int64_t total) {
    auto row_consts = [&](int bc, float& mean, float& rstd, float& ga, float& be, float& sc, float& sh, float& m1, float& m2) {
        const int b = bc / C, c = bc - b * C;
        const int g = c / (C / G);
        mean = stats[(b * G + g) * 2]; rstd = stats[(b * G + g) * 2 + 1];
        m1 = gstat[(b * G + g) * 2]; m2 = gstat[(b * G + g) * 2 + 1];
        sc = 1.0f; sh = 0.0f;
        if (ss) { sc = ss[(int64_t)b * ss_b_stride + c] + 1.0f; sh = ss[(int64_t)b * ss_b_stride + C + c]; }
        ga = gamma[c]; be = beta[c];
    };
    float mean, rstd, ga, be, sc, sh, m1, m2;
    if (FLAT) {
        for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
            row_consts((int)(i / S), mean, rstd, ga, be, sc, sh, m1, m2);
            const float xh = (h[i] - mean) * rstd;
            const float v = (xh * ga + be) * sc + sh;
            gh[i] = rstd * (gy[i] * dsilu(v) * (ga * sc) - m1 - xh * m2);
        }
    } else {
        const int bc = blockIdx.x;
        row_consts(bc, mean, rstd, ga, be, sc, sh, m1, m2);
        const float k = ga * sc;
        const int64_t base = (int64_t)bc * S;
        for (int64_t i = (int64_t)blockIdx.y * NT + threadIdx.x; i < S; i += (int64_t)gridDim.y * NT) {
            const float xh = (h[base + i] - mean) * rstd;
            const float v = (xh * ga + be) * sc + sh;
            gh[base + i] = rstd * (gy[base + i] * dsilu(v) * k - m1 - xh * m2);
        }
    }
}

// ------------------------------------------------------------------------------------------------ channel norm backward
// forward (sdc_chan_norm): LayerNorm over channels (mode 0): y = (x - mean) rsqrt(var + eps) g;  RMSNorm (mode 1):
// y = x / max(||x||, 1e-12) g sqrt(C).  Thread layout of the forward kernel: PL position lanes x NT / PL channel slices per
// workgroup (positions are the contiguous axis: every channel row is read in coalesced segments), channel sums through LDS.
// The kernel also leaves (mean, scale) of every position in `pstat` for the gain-gradient kernel.
template <int PL>
__global__ __launch_bounds__(NT) void chan_norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                          const float* __restrict__ g, float* __restrict__ gx,
                                                          float* __restrict__ pstat, int C, int64_t S, int mode, float eps) {
    constexpr int NSL = NT / PL;
    const int lane = threadIdx.x % PL, slice = threadIdx.x / PL;
    const int b = blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * PL + lane;
    const bool ok = p < S;
    const int64_t base = (int64_t)b * C * S + (ok ? p : 0);
    __shared__ float sh[3][NSL][PL];
    auto reduce3 = [&](float& a0, float& a1, float& a2) {
        sh[0][slice][lane] = a0; sh[1][slice][lane] = a1; sh[2][slice][lane] = a2;
        __syncthreads();
        a0 = a1 = a2 = 0.f;
#pragma unroll
        for (int i = 0; i < NSL; ++i) { a0 += sh[0][i][lane]; a1 += sh[1][i][lane]; a2 += sh[2][i][lane]; }
        __syncthreads();
    };
    // pass 1: sum x, sum x^2
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (ok)
        for (int c = slice; c < C; c += NSL) { const float v = x[base + (int64_t)c * S]; s1 += v; s2 += v * v; }
    reduce3(s1, s2, s3);
    float mean = 0.f, scale;
    if (mode == 0) {
        mean = s1 / (float)C;
        float q2 = 0.f, z0 = 0.f, z1 = 0.f;               // centred second pass, like the forward kernel
        if (ok)
            for (int c = slice; c < C; c += NSL) { const float dv = x[base + (int64_t)c * S] - mean; q2 += dv * dv; }
        reduce3(q2, z0, z1);
        scale = rsqrtf(q2 / (float)C + eps);
    } else {
        scale = 1.0f / fmaxf(sqrtf(s2), 1e-12f);          // (the sqrt(C) factor is applied below)
    }
    // pass 2: sums of gy g and gy g xhat
    float t1 = 0.f, t2 = 0.f, t3 = 0.f;
    if (ok)
        for (int c = slice; c < C; c += NSL) {
            const float gg = gy[base + (int64_t)c * S] * g[c];
            const float xv = x[base + (int64_t)c * S];
            t1 += gg;
            t2 += gg * (mode == 0 ? (xv - mean) * scale : xv);
        }
    reduce3(t1, t2, t3);
    if (!ok) return;
    if (slice == 0) {
        pstat[((int64_t)b * S + p) * 2] = mean;
        pstat[((int64_t)b * S + p) * 2 + 1] = mode == 0 ? scale : scale * sqrtf((float)C);
    }
    if (mode == 0) {
        const float m1 = t1 / (float)C, m2 = t2 / (float)C;
        for (int c = slice; c < C; c += NSL) {
            const int64_t o = base + (int64_t)c * S;
            const float xh = (x[o] - mean) * scale;
            gx[o] = scale * (gy[o] * g[c] - m1 - xh * m2);
        }
    } else {
        const float sq = sqrtf((float)C);
        const float k = sqrtf(s2) > 1e-12f ? t2 * scale * scale : 0.f;
        for (int c = slice; c < C; c += NSL) {
            const int64_t o = base + (int64_t)c * S;
            gx[o] = sq * scale * (gy[o] * g[c] - x[o] * k);
        }
    }
}

// d g[c] = sum_{b, p} gy[b][c][p] (x[b][c][p] - mean[b][p]) scale[b][p]: one workgroup per (channel, batch slab), fixed order
__global__ __launch_bounds__(NT) void chan_norm_gaing_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                            const float* __restrict__ pstat, float* __restrict__ gpart, int B, int C,
                                                            int64_t S, int nslab) {
    const int c = blockIdx.x, slab = blockIdx.y;
    const int b_lo = (int)((int64_t)B * slab / nslab), b_hi = (int)((int64_t)B * (slab + 1) / nslab);
    float acc = 0.f;
    for (int b = b_lo; b < b_hi; ++b) {
        const float* xb = x + ((int64_t)b * C + c) * S;
        const float* yb = gy + ((int64_t)b * C + c) * S;
        const float* st = pstat + (int64_t)b * S * 2;
        for (int64_t p = threadIdx.x; p < S; p += NT) acc += yb[p] * (xb[p] - st[2 * p]) * st[2 * p + 1];
    }
    __shared__ float red[NT / 64];
    acc = sdc::wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) gpart[(int64_t)c * B + slab] = (red[0] + red[1]) + (red[2] + red[3]);      // [C][B]
    // the entries of the slabs that do not exist: zeroed here, by slab 0's workgroup (a hipMemsetAsync in front of this launch
    // was not reliably replayed when the call was recorded by a stream capture -- safediffcon_amd/train_graph.py: the gain
    // gradients of a replayed fine-tuning step then summed whatever the graph's memory pool had left there)
    if (slab == 0)
        for (int j = nslab + threadIdx.x; j < B; j += NT) gpart[(int64_t)c * B + j] = 0.0f;
}

// ------------------------------------------------------------------------------------------------ small element-wise VJPs
// kind 0: SiLU, 1: GELU (exact erf):  gx = gy * f'(x)
__global__ __launch_bounds__(NT) void act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gx,
                                                    int64_t n, int kind) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const float v = x[i];
        float dv;
        if (kind == 0) dv = dsilu(v);
        else dv = 0.5f * (1.0f + erff(v * 0.70710678118654752f)) + v * 0.3989422804014327f * __expf(-0.5f * v * v);
        gx[i] = gy[i] * dv;
    }
}

// VJP of nearest x2 upsampling along H and W (fh, fw in {1, 2}): gx[r][h][w] = sum of the fh x fw block of g
__global__ __launch_bounds__(NT) void sumpool_kernel(const float* __restrict__ g, float* __restrict__ gx, int64_t rows, int H, int W,
                                                    int fh, int fw) {
    const int64_t n = rows * H * W;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const int w = (int)(i % W);
        const int64_t q = i / W;
        const int hh = (int)(q % H);
        const int64_t r = q / H;
        const float* p = g + (r * (H * fh) + (int64_t)hh * fh) * (W * fw) + (int64_t)w * fw;
        float s = 0.0f;
        for (int a = 0; a < fh; ++a)
            for (int bb = 0; bb < fw; ++bb) s += p[(int64_t)a * (W * fw) + bb];
        gx[i] = s;
    }
}

// parameter gradients from the row table: a workgroup owns 64 channels, wave g the samples b = g (mod 4) (fixed order, the four
// partial sums added in order):  dgamma[c] = sum_b (1 + sc) A2, dbeta[c] = sum_b (1 + sc) A1, dss[b] = [gamma A2 + beta A1 (C) | A1 (C)]
__device__ __forceinline__ void gn_bwd_param_body(const int blk, const float* __restrict__ rows, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, const float* __restrict__ ss,
                                                  int64_t ss_b_stride, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                  float* __restrict__ dss, int B, int C) {
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blk * 64 + lane;
    const bool live = c < C;
    float sg = 0.0f, sb = 0.0f;
    if (live) {
        const float ga = gamma[c], be = beta[c];
        for (int b = g; b < B; b += NT / 64) {
            const float2 r = *reinterpret_cast<const float2*>(rows + ((int64_t)b * C + c) * 2);
            const float k = ss ? ss[(int64_t)b * ss_b_stride + c] + 1.0f : 1.0f;
            sg += k * r.y;
            sb += k * r.x;
            if (dss) { dss[(int64_t)b * 2 * C + c] = ga * r.y + be * r.x; dss[(int64_t)b * 2 * C + C + c] = r.x; }
        }
    }
    __shared__ float sh[2][NT / 64][64];
    sh[0][g][lane] = sg; sh[1][g][lane] = sb;
    __syncthreads();
    if (g == 0 && live) {
        dgamma[c] = (sh[0][0][lane] + sh[0][1][lane]) + (sh[0][2][lane] + sh[0][3][lane]);
        dbeta[c] = (sh[1][0][lane] + sh[1][1][lane]) + (sh[1][2][lane] + sh[1][3][lane]);
    }
}

// both small passes over the row table in ONE launch (neither depends on the other): workgroups 0 .. B G - 1 form the group means,
// the rest the parameter gradients -- in a fine-tuning step of the 1-D nets a launch costs more than either pass
__global__ __launch_bounds__(NT) void gn_bwd_group_param_kernel(const float* __restrict__ rows, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, const float* __restrict__ ss,
                                                               int64_t ss_b_stride, float* __restrict__ gstat, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, float* __restrict__ dss, int B, int C, int G,
                                                               int64_t S) {
    const int nbg = B * G;
    if ((int)blockIdx.x < nbg) gn_bwd_group_body((int)blockIdx.x, rows, gamma, ss, ss_b_stride, gstat, C, G, S);
    else gn_bwd_param_body((int)blockIdx.x - nbg, rows, gamma, beta, ss, ss_b_stride, dgamma, dbeta, dss, B, C);
}

extern "C" size_t sdc_gn_silu_bwd_floats(int B, int C, int G, int64_t S) {
    if (B <= 0 || C <= 0 || G <= 0) return 0;
    const size_t base = (((size_t)B * C + (size_t)B * G) * 2 + 1) / 2 * 2;
    return base + (size_t)B * C * gn_bwd_ysplit(B * C, S) * 2 * 2;                // fp64 pieces, counted in floats
}

extern "C" int sdc_gn_silu_bwd(const float* h, const float* gy, const float* stats, const float* gamma, const float* beta,
                               const float* ss, int64_t ss_b_stride, float* rows, float* gh, float* dgamma, float* dbeta, float* dss,
                               int B, int C, int G, int64_t S, void* stream) {
    SDC_REQUIRE(h && gy && stats && gamma && beta && rows && gh, SDC_ENULL, "sdc_gn_silu_bwd: null pointer");
    SDC_REQUIRE((dgamma != nullptr) == (dbeta != nullptr) && (!dss || (dgamma && ss)), SDC_EINVAL,
                "sdc_gn_silu_bwd: dgamma and dbeta come together, dss with them and with ss");
    SDC_REQUIRE(B > 0 && C > 0 && G > 0 && C % G == 0 && S > 0, SDC_EINVAL, "sdc_gn_silu_bwd: bad sizes");
    SDC_REQUIRE((int64_t)B * C < (1ll << 31), SDC_EINVAL, "sdc_gn_silu_bwd: too many rows");
    hipStream_t s = sdc::as_stream(stream);
    const int nrows = B * C;
    float* gstat = rows + (int64_t)nrows * 2;             // rows = [B][C][2] row sums, then [B][G][2] group means, then the pieces
    const int ysplit = gn_bwd_ysplit(nrows, S);
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(gy)) & 15) == 0;
    if (ysplit > 1 && vec_ok) {
        const size_t off = (((size_t)(nrows + B * G) * 2 + 1) / 2) * 2;         // doubles start on an 8-byte boundary of the float array
        double* part = reinterpret_cast<double*>(rows + off);
        SDC_REQUIRE((reinterpret_cast<uintptr_t>(part) & 7) == 0, SDC_EINVAL, "sdc_gn_silu_bwd: rows must be 8-byte aligned");
        int64_t piece = ((S + ysplit - 1) / ysplit + 3) / 4 * 4;
        hipLaunchKernelGGL(gn_bwd_rows_part_kernel, dim3((unsigned)nrows, (unsigned)ysplit), dim3(NT), 0, s, h, gy, stats, gamma, beta, ss,
                           ss_b_stride, part, C, G, S, piece);
        hipLaunchKernelGGL(gn_bwd_rows_finish_kernel, dim3((unsigned)((nrows + NT - 1) / NT)), dim3(NT), 0, s, (const double*)part, rows, nrows, ysplit);
    } else if (S >= 1024)
        hipLaunchKernelGGL(gn_bwd_rows_kernel<1>, dim3((unsigned)nrows), dim3(NT), 0, s, h, gy, stats, gamma, beta, ss, ss_b_stride, rows, nrows, C, G, S);
    else
        hipLaunchKernelGGL(gn_bwd_rows_kernel<4>, dim3((unsigned)((nrows + 3) / 4)), dim3(NT), 0, s, h, gy, stats, gamma, beta, ss, ss_b_stride, rows, nrows, C, G, S);
    hipLaunchKernelGGL(gn_bwd_group_param_kernel, dim3((unsigned)(B * G + (dgamma ? (C + 63) / 64 : 0))), dim3(NT), 0, s, rows, gamma, beta,
                       ss, ss_b_stride, gstat, dgamma, dbeta, dss, B, C, G, S);
    const int64_t total = (int64_t)nrows * S;
    if (S >= 1024) {
        int ysplit = (int)((S + NT * 8 - 1) / (NT * 8));
        if (ysplit > 64) ysplit = 64;
        hipLaunchKernelGGL(gn_bwd_apply_kernel<false>, dim3((unsigned)nrows, (unsigned)ysplit), dim3(NT), 0, s, h, gy, stats, gamma, beta, ss,
                           ss_b_stride, gstat, gh, C, G, S, total);
    } else {
        const int blocks = (int)((total + NT - 1) / NT < 8192 ? (total + NT - 1) / NT : 8192);
        hipLaunchKernelGGL(gn_bwd_apply_kernel<true>, dim3((unsigned)blocks), dim3(NT), 0, s, h, gy, stats, gamma, beta, ss, ss_b_stride,
                           gstat, gh, C, G, S, total);
    }
    return sdc::check_launch("sdc_gn_silu_bwd");
}

// gain-gradient partials per channel (the caller sums them) followed by the (mean, scale) table of every position
static int cn_slabs(int B, int C) {
    int ns = 1;
    while (ns < B && (int64_t)C * ns < 1024) ns *= 2;      // enough workgroups to fill the chip
    return ns < B ? ns : B;
}

extern "C" size_t sdc_chan_norm_bwd_parts(int B, int64_t S) { (void)S; return (size_t)B; }

extern "C" size_t sdc_chan_norm_bwd_bytes(int B, int C, int64_t S) {
    return ((size_t)C * B + (size_t)B * S * 2) * sizeof(float);
}

extern "C" int sdc_chan_norm_bwd(const float* x, const float* gy, const float* g, float* gx, float* gpart, int B, int C, int64_t S,
                                 int mode, float eps, void* stream) {
    SDC_REQUIRE(x && gy && g && gx && gpart, SDC_ENULL, "sdc_chan_norm_bwd: null pointer");
    SDC_REQUIRE(B > 0 && B < 65536 && C > 0 && S > 0 && (mode == 0 || mode == 1), SDC_EINVAL, "sdc_chan_norm_bwd: bad arguments");
    hipStream_t s = sdc::as_stream(stream);
    float* pstat = gpart + (size_t)C * B;                  // gpart = sdc_chan_norm_bwd_bytes(B, C, S): [C][B] partials, then the table
    if (S >= 1024) {
        dim3 grid((unsigned)((S + 63) / 64), (unsigned)B);
        hipLaunchKernelGGL(chan_norm_bwd_kernel<64>, grid, dim3(NT), 0, s, x, gy, g, gx, pstat, C, S, mode, eps);
    } else {
        dim3 grid((unsigned)((S + 15) / 16), (unsigned)B);
        hipLaunchKernelGGL(chan_norm_bwd_kernel<16>, grid, dim3(NT), 0, s, x, gy, g, gx, pstat, C, S, mode, eps);
    }
    const int ns = cn_slabs(B, C);
    // partial layout [C][B]: the entries of slabs beyond ns are zeroed by the kernel itself
    hipLaunchKernelGGL(chan_norm_gaing_kernel, dim3((unsigned)C, (unsigned)ns), dim3(NT), 0, s, x, gy, pstat, gpart, B, C, S, ns);
    return sdc::check_launch("sdc_chan_norm_bwd");
}

extern "C" int sdc_act_bwd(const float* x, const float* gy, float* gx, int64_t n, int kind, void* stream) {
    SDC_REQUIRE(x && gy && gx, SDC_ENULL, "sdc_act_bwd: null pointer");
    SDC_REQUIRE(n >= 0 && (kind == 0 || kind == 1), SDC_EINVAL, "sdc_act_bwd: bad arguments");
    if (n == 0) return SDC_OK;
    const int blocks = (int)((n + NT - 1) / NT < 8192 ? (n + NT - 1) / NT : 8192);
    hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks), dim3(NT), 0, sdc::as_stream(stream), x, gy, gx, n, kind);
    return sdc::check_launch("sdc_act_bwd");
}

extern "C" int sdc_sumpool2(const float* g, float* gx, int64_t rows, int H, int W, int fh, int fw, void* stream) {
    SDC_REQUIRE(g && gx, SDC_ENULL, "sdc_sumpool2: null pointer");
    SDC_REQUIRE(rows > 0 && H > 0 && W > 0 && (fh == 1 || fh == 2) && (fw == 1 || fw == 2), SDC_EINVAL, "sdc_sumpool2: bad arguments");
    const int64_t n = rows * H * W;
    const int blocks = (int)((n + NT - 1) / NT < 8192 ? (n + NT - 1) / NT : 8192);
    hipLaunchKernelGGL(sumpool_kernel, dim3(blocks), dim3(NT), 0, sdc::as_stream(stream), g, gx, rows, H, W, fh, fw);
    return sdc::check_launch("sdc_sumpool2");
}

// ------------------------------------------------------------------------------------------------ weight packing
// Kernel layout of an nn.Conv weight (include/sdc.h, SdcConvDesc.precision) in ONE launch: Wp[(kd, kh, kw, ci)][co], then for
// 3-wide taps the Winograd F(2,3) taps Wg[(kd kH + kh) 4 + xi][ci][co], for 3x3 taps the F(2x2,3x3) taps Wg2[kd][ci][co][j 4 + xi]
// and for 3x3x3 taps the F(2x2x2,3x3x3) taps Wg3[jd][ci][co][j 4 + xi] -- every transformed tap summed in fp64 and rounded once,
// like the host packing (engine.pack_conv_weight; equal to it to the last bit or two of fp32, the fp64 sums are ordered differently).  flip != 0 packs the data-gradient weight instead: w'[co'][ci'][k] =
// w[ci'][co'][K - 1 - k] (transposed channels, flipped taps).  A fine-tuning step re-packs every conv twice; as torch ops that
// was ~12 launches per conv.
namespace {
__device__ __forceinline__ double wino_g(int row, int tap) {
    // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
    if (row == 0) return tap == 0 ? 1.0 : 0.0;
    if (row == 3) return tap == 2 ? 1.0 : 0.0;
    return (row == 2 && tap == 1) ? -0.5 : 0.5;
}

struct PackArgs {
    const float* w; float* out;
    int Cout, Cin, kD, kH, kW, flip;       // logical (packed) channel counts
    int64_t n0, n1, n2, n3, n4;            // floats of the five sections (n4: F(4,3) taps of a 1-D conv, after the others)
};

// One workgroup stages a (co_t co) x (ci_t ci) x taps block of the weight through LDS -- read along the source's contiguous axis
// ([co][ci, taps] rows, or [ci][co, taps] rows when flipped), written along co, the packed layouts' contiguous axis -- and emits
// that block of every section.
__device__ __forceinline__ void pack_weight_tile(const PackArgs& a, const int co_sh, const int ci_sh, const unsigned tap_magic,
                                                 const int bx, const int by, float* tile) {
    // tile: [co_t][ci_t * taps + 1], element (co, ci, logical tap)
    const int co_t = 1 << co_sh, ci_t = 1 << ci_sh;
    const int taps = a.kD * a.kH * a.kW;
    const int pitch = ci_t * taps + 1;
    const int co0 = bx << co_sh, ci0 = by << ci_sh;
    const int nco = a.Cout - co0 < co_t ? a.Cout - co0 : co_t;
    const int nci = a.Cin - ci0 < ci_t ? a.Cin - ci0 : ci_t;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (!a.flip) {
        const int len = nci * taps;
        for (int r = wave; r < nco; r += NT / 64) {
            const float* src = a.w + ((int64_t)(co0 + r) * a.Cin + ci0) * taps;
            for (int c = lane; c < len; c += 64) tile[r * pitch + c] = src[c];
        }
    } else {                                              // w[ci][co][taps], taps reversed
        const int len = nco * taps;
        for (int r = wave; r < nci; r += NT / 64) {
            const float* src = a.w + ((int64_t)(ci0 + r) * a.Cout + co0) * taps;
            for (int c = lane; c < len; c += 64) {
                const int co = (int)(((unsigned)c * tap_magic) >> 24), t = c - co * taps;     // c / taps (exact: c * taps < 2^24)
                tile[co * pitch + r * taps + (taps - 1 - t)] = src[c];
            }
        }
    }
    __syncthreads();
    auto wv = [&](int co, int ci, int kd, int kh, int kw) -> double { return (double)tile[co * pitch + ci * taps + (kd * a.kH + kh) * a.kW + kw]; };
    const int cell = co_t * ci_t;
    // Wp[(kd, kh, kw, ci)][co]
    for (int tap = 0; tap < taps; ++tap)
        for (int e = threadIdx.x; e < cell; e += NT) {
            const int co = e & (co_t - 1), ci = e >> co_sh;
            if (co < nco && ci < nci) a.out[((int64_t)tap * a.Cin + ci0 + ci) * a.Cout + co0 + co] = tile[co * pitch + ci * taps + tap];
        }
    if (a.n1) {                                           // Wg[(kd kH + kh) 4 + xi][ci][co]
        float* o = a.out + a.n0;
        for (int kd = 0; kd < a.kD; ++kd)
            for (int kh = 0; kh < a.kH; ++kh)
                for (int xi = 0; xi < 4; ++xi)
                    for (int e = threadIdx.x; e < cell; e += NT) {
                        const int co = e & (co_t - 1), ci = e >> co_sh;
                        if (co >= nco || ci >= nci) continue;
                        double v = 0.0;
                        for (int kw = 0; kw < 3; ++kw) v += wino_g(xi, kw) * wv(co, ci, kd, kh, kw);
                        o[((int64_t)((kd * a.kH + kh) * 4 + xi) * a.Cin + ci0 + ci) * a.Cout + co0 + co] = (float)v;
                    }
    }
    if (a.n4) {                                           // Wg43[xi][ci][co], xi < 6: F(4,3) taps of a 1-D conv (kD = kH = 1)
        float* o = a.out + a.n0 + a.n1 + a.n2 + a.n3;
        for (int xi = 0; xi < 6; ++xi)
            for (int e = threadIdx.x; e < cell; e += NT) {
                const int co = e & (co_t - 1), ci = e >> co_sh;
                if (co >= nco || ci >= nci) continue;
                const double g0 = wv(co, ci, 0, 0, 0), g1 = wv(co, ci, 0, 0, 1), g2 = wv(co, ci, 0, 0, 2);
                double v;
                if (xi == 0) v = g0 / 4.0;
                else if (xi == 1) v = -((g0 + g2) + g1) / 6.0;
                else if (xi == 2) v = -((g0 + g2) - g1) / 6.0;
                else if (xi == 3) v = (g0 / 4.0 + g2) / 6.0 + g1 / 12.0;
                else if (xi == 4) v = (g0 / 4.0 + g2) / 6.0 - g1 / 12.0;
                else v = g2;
                o[((int64_t)xi * a.Cin + ci0 + ci) * a.Cout + co0 + co] = (float)v;
            }
    }
    // the 2-D / 3-D Winograd taps: a thread forms the (kd, kh)-weighted sums of the three kw taps once (rows 0 and 3 of G pick one
    // tap, rows 1 and 2 weigh all three by +-1/2) and writes the four xi taps of its (jd, j) as one 16-byte store
    auto g_lo = [](int r) { return r == 3 ? 2 : 0; };
    auto g_hi = [](int r) { return r == 0 ? 0 : 2; };
    auto g_cf = [](int r, int t) -> double { return (r == 0 || r == 3) ? 1.0 : ((r == 2 && t == 1) ? -0.5 : 0.5); };
    auto emit4 = [](float* o, double s0, double s1, double s2) {
        const float4 v = make_float4((float)s0, (float)(0.5 * ((s0 + s2) + s1)), (float)(0.5 * ((s0 + s2) - s1)), (float)s2);
        if ((reinterpret_cast<uintptr_t>(o) & 15) == 0) *reinterpret_cast<float4*>(o) = v;       // odd channel products shift the sections
        else { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
    };
    if (a.n2) {                                           // Wg2[kd][ci][co][j 4 + xi]
        float* o = a.out + a.n0 + a.n1;
        for (int kd = 0; kd < a.kD; ++kd)
            for (int e = threadIdx.x; e < cell * 4; e += NT) {
                const int j = e & 3, co = (e >> 2) & (co_t - 1), ci = e >> (2 + co_sh);
                if (co >= nco || ci >= nci) continue;
                double s0 = 0.0, s1 = 0.0, s2 = 0.0;
                for (int kh = g_lo(j); kh <= g_hi(j); ++kh) {
                    const double c = g_cf(j, kh);
                    s0 += c * wv(co, ci, kd, kh, 0); s1 += c * wv(co, ci, kd, kh, 1); s2 += c * wv(co, ci, kd, kh, 2);
                }
                emit4(o + (((int64_t)kd * a.Cin + ci0 + ci) * a.Cout + co0 + co) * 16 + j * 4, s0, s1, s2);
            }
    }
    if (a.n3) {                                           // Wg3[jd][ci][co][j 4 + xi]
        float* o = a.out + a.n0 + a.n1 + a.n2;
        for (int jd = 0; jd < 4; ++jd)
            for (int e = threadIdx.x; e < cell * 4; e += NT) {
                const int j = e & 3, co = (e >> 2) & (co_t - 1), ci = e >> (2 + co_sh);
                if (co >= nco || ci >= nci) continue;
                double s0 = 0.0, s1 = 0.0, s2 = 0.0;
                for (int kd = g_lo(jd); kd <= g_hi(jd); ++kd)
                    for (int kh = g_lo(j); kh <= g_hi(j); ++kh) {
                        const double c = g_cf(jd, kd) * g_cf(j, kh);
                        s0 += c * wv(co, ci, kd, kh, 0); s1 += c * wv(co, ci, kd, kh, 1); s2 += c * wv(co, ci, kd, kh, 2);
                    }
                emit4(o + (((int64_t)jd * a.Cin + ci0 + ci) * a.Cout + co0 + co) * 16 + j * 4, s0, s1, s2);
            }
    }
}

__global__ __launch_bounds__(NT) void pack_weight_kernel(const PackArgs a, const int co_sh, const int ci_sh, const unsigned tap_magic) {
    extern __shared__ float tile[];
    pack_weight_tile(a, co_sh, ci_sh, tap_magic, blockIdx.x, blockIdx.y, tile);
}

// Every conv weight of a net in ONE launch (a fine-tuning step packs each conv twice, forward and data-gradient form: 187
// launches of ~13 us on the tokamak net).  Workgroup b belongs to the item with block0 <= b < block0 + grid_x * grid_y
// (binary search over the table, which lives in device memory and is built once per net: sdc_pack_batch_plan).
__global__ __launch_bounds__(NT) void pack_weight_batch_kernel(const SdcPackItem* __restrict__ items, const int n) {
    extern __shared__ float tile[];
    const int b = blockIdx.x;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].block0 <= b) lo = mid; else hi = mid - 1;
    }
    const SdcPackItem& it = items[lo];
    PackArgs a;
    a.w = it.w; a.out = it.out; a.Cout = it.Cout; a.Cin = it.Cin; a.kD = it.kD; a.kH = it.kH; a.kW = it.kW; a.flip = it.flip;
    a.n0 = it.n[0]; a.n1 = it.n[1]; a.n2 = it.n[2]; a.n3 = it.n[3]; a.n4 = it.n[4];
    const int r = b - it.block0;
    pack_weight_tile(a, it.co_sh, it.ci_sh, it.tap_magic, r % it.grid_x, r / it.grid_x, tile);
}

// source block (co_t x ci_t channels, all taps) of one workgroup: <= 128 (ci, tap) columns, <= 48 KB, and at least ~4
// workgroups per CU where the weight is large enough (a tile's emit loops are a few thousand fp64 sums per thread)
void pack_tile_shape(int Cout, int Cin, int taps, int& co_sh, int& ci_sh, size_t& lds) {
    ci_sh = 4; co_sh = 6;
    while (ci_sh > 0 && (taps << ci_sh) > 128) --ci_sh;
    while (co_sh > 0 && ((size_t)((taps << ci_sh) + 1) << co_sh) * sizeof(float) > 48u * 1024u) --co_sh;
    while (ci_sh > 1 && (int64_t)((Cout + (1 << co_sh) - 1) >> co_sh) * ((Cin + (1 << ci_sh) - 1) >> ci_sh) < 1024) --ci_sh;
    lds = ((size_t)((taps << ci_sh) + 1) << co_sh) * sizeof(float);
}

void pack_sections(int Cout, int Cin, int kD, int kH, int kW, int precision, int64_t* n) {
    const int64_t nw = (int64_t)Cout * Cin * kD * kH * kW;
    n[0] = nw; n[1] = n[2] = n[3] = n[4] = 0;
    if (precision >= 2 && kW == 3) {
        n[1] = nw / 3 * 4;
        if (precision == 5 && kD == 1 && kH == 1) n[4] = nw / 3 * 6;
        if (precision >= 3 && kH == 3) {
            n[2] = nw / 9 * 16;
            if (precision >= 4 && kD == 3) n[3] = nw / 27 * 64;
        }
    }
}
}  // namespace

extern "C" size_t sdc_pack_conv_weight_floats(int Cout, int Cin, int kD, int kH, int kW, int precision) {
    int64_t n[5];
    pack_sections(Cout, Cin, kD, kH, kW, precision, n);
    return (size_t)(n[0] + n[1] + n[2] + n[3] + n[4]);
}

extern "C" int sdc_pack_conv_weight(const float* w, float* out, int Cout, int Cin, int kD, int kH, int kW, int precision, int flip,
                                    void* stream) {
    SDC_REQUIRE(w && out, SDC_ENULL, "sdc_pack_conv_weight: null pointer");
    SDC_REQUIRE(Cout > 0 && Cin > 0 && kD > 0 && kH > 0 && kW > 0 && (precision == 0 || (precision >= 2 && precision <= 5)), SDC_EINVAL,
                "sdc_pack_conv_weight: bad arguments");
    PackArgs a;
    a.w = w; a.out = out; a.Cout = Cout; a.Cin = Cin; a.kD = kD; a.kH = kH; a.kW = kW; a.flip = flip;
    int64_t n[5];
    pack_sections(Cout, Cin, kD, kH, kW, precision, n);
    a.n0 = n[0]; a.n1 = n[1]; a.n2 = n[2]; a.n3 = n[3]; a.n4 = n[4];
    int co_sh, ci_sh;
    size_t lds;
    pack_tile_shape(Cout, Cin, kD * kH * kW, co_sh, ci_sh, lds);
    const int taps = kD * kH * kW;
    SDC_REQUIRE(lds <= 64u * 1024u && ((int64_t)taps * taps << co_sh) < (1 << 24), SDC_EINVAL, "sdc_pack_conv_weight: too many taps");
    const dim3 grid((unsigned)((Cout + (1 << co_sh) - 1) >> co_sh), (unsigned)((Cin + (1 << ci_sh) - 1) >> ci_sh));
    SDC_REQUIRE(grid.y < 65536u, SDC_EINVAL, "sdc_pack_conv_weight: too many input channels");
    const unsigned tap_magic = (unsigned)(((1u << 24) + taps - 1) / taps);
    hipLaunchKernelGGL(pack_weight_kernel, grid, dim3(NT), lds, sdc::as_stream(stream), a, co_sh, ci_sh, tap_magic);
    return sdc::check_launch("sdc_pack_conv_weight");
}

extern "C" int sdc_pack_batch_plan(SdcPackItem* items, int n, int* total_blocks, int* lds_bytes) {
    SDC_REQUIRE(items && total_blocks && lds_bytes, SDC_ENULL, "sdc_pack_batch_plan: null pointer");
    SDC_REQUIRE(n > 0, SDC_EINVAL, "sdc_pack_batch_plan: no items");
    int64_t blocks = 0;
    size_t lds_max = 0;
    for (int i = 0; i < n; ++i) {
        SdcPackItem& it = items[i];
        SDC_REQUIRE(it.w && it.out, SDC_ENULL, "sdc_pack_batch_plan: item %d: null pointer", i);
        SDC_REQUIRE(it.Cout > 0 && it.Cin > 0 && it.kD > 0 && it.kH > 0 && it.kW > 0 &&
                    (it.precision == 0 || (it.precision >= 2 && it.precision <= 5)), SDC_EINVAL, "sdc_pack_batch_plan: item %d: bad arguments", i);
        const int taps = it.kD * it.kH * it.kW;
        size_t lds;
        pack_tile_shape(it.Cout, it.Cin, taps, it.co_sh, it.ci_sh, lds);
        SDC_REQUIRE(lds <= 64u * 1024u && ((int64_t)taps * taps << it.co_sh) < (1 << 24), SDC_EINVAL, "sdc_pack_batch_plan: item %d: too many taps", i);
        pack_sections(it.Cout, it.Cin, it.kD, it.kH, it.kW, it.precision, it.n);
        it.tap_magic = (unsigned)(((1u << 24) + taps - 1) / taps);
        it.grid_x = (it.Cout + (1 << it.co_sh) - 1) >> it.co_sh;
        it.grid_y = (it.Cin + (1 << it.ci_sh) - 1) >> it.ci_sh;
        it.block0 = (int)blocks;
        blocks += (int64_t)it.grid_x * it.grid_y;
        SDC_REQUIRE(blocks < (1ll << 30), SDC_EINVAL, "sdc_pack_batch_plan: too many workgroups");
        if (lds > lds_max) lds_max = lds;
    }
    *total_blocks = (int)blocks;
    *lds_bytes = (int)lds_max;
    return SDC_OK;
}

extern "C" int sdc_pack_batch_run(const SdcPackItem* items_dev, int n, int total_blocks, int lds_bytes, void* stream) {
    SDC_REQUIRE(items_dev, SDC_ENULL, "sdc_pack_batch_run: null pointer");
    SDC_REQUIRE(n > 0 && total_blocks > 0 && lds_bytes > 0 && lds_bytes <= 64 * 1024, SDC_EINVAL, "sdc_pack_batch_run: bad arguments");
    hipLaunchKernelGGL(pack_weight_batch_kernel, dim3((unsigned)total_blocks), dim3(NT), (size_t)lds_bytes, sdc::as_stream(stream), items_dev, n);
    return sdc::check_launch("sdc_pack_batch_run");
}
