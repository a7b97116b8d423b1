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
#include "sdc_conv.h"

using namespace sdcconv;

namespace {

// ---- epilogue: D rows (co) live in registers, columns (positions) on lanes -> coalesced along W.
// Bias / residual loads are issued as a batch (clamped addresses, no per-element branches) so the
// workgroup pays one memory round trip per 16 outputs instead of sixteen.
template <int TM, int TN>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[TM][TN], int mw, int nw, int lane, float* ybase = nullptr) {
    const SdcConvDesc& d = a.d;
    float* const ay = ybase ? ybase : a.y;       // (sdc_conv_splitk: this split's partial copy)
    const int l31 = lane & 31, lh = lane >> 5;
    // Dense outputs (the usual case): a position decodes to (sample, offset inside the sample) with one float quotient and
    // a fix-up, the channel part of every address is scalar.  The general path below spends ~25 VALU instructions per
    // runtime integer division, six of them per 32-position tile -- a fifth of a short-K (1x1, Cin = 128) workgroup's time.
    if (a.ydense && mw + TM * 32 <= d.Cout) {
        typedef __attribute__((address_space(1))) char* gwchar_p;
        typedef __attribute__((address_space(1))) float* gwfloat_p;
        typedef const __attribute__((address_space(1))) char* gchar_p;
        const int S = d.oD * d.oH * d.oW;
        const float r_S = 1.0f / (float)S;
        const int64_t ycs4 = d.ys[1] * 4, rcs4 = d.rs[1] * 4;
        uint32_t yoff[TN], roff[TN];
        bool pok[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int pp = nw + j * 32 + l31;
            pok[j] = pp < a.Ntot;
            const int p = pok[j] ? pp : 0;
            int b = (int)((float)p * r_S);                  // within a few units of p / S, fixed up below
            int sp = p - b * S;
            while (sp < 0) { --b; sp += S; }
            while (sp >= S) { ++b; sp -= S; }
            yoff[j] = (uint32_t)(b * d.ys[0] + sp + (4 * lh) * d.ys[1]) * 4u;
            roff[j] = a.res ? (uint32_t)(b * d.rs[0] + sp + (4 * lh) * d.rs[1]) * 4u : 0u;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int cob = mw + i * 32;                    // wave-uniform; this lane's rows: cob + 4 lh + (rr & 3) + 8 (rr >> 2)
            gwchar_p yb = (gwchar_p)(__attribute__((address_space(1))) void*)uniform_ptr(ay + (int64_t)cob * d.ys[1]);
            gchar_p rb = (gchar_p)uniform_ptr(a.res ? a.res + (int64_t)cob * d.rs[1] : ay);
            if (a.res) {
                // bias and residual of the 16 rows as one batch of loads (one memory round trip per tile row, not sixteen)
                float bv[16], rv[16][TN];
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    bv[rr] = a.bias ? a.bias[cob + 4 * lh + (rr & 3) + 8 * (rr >> 2)] : 0.0f;
#pragma unroll
                    for (int j = 0; j < TN; ++j) rv[rr][j] = *(gfloat_p)(rb + roff[j]);
                    rb += (rr & 3) < 3 ? rcs4 : 5 * rcs4;
                }
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        if (pok[j]) *(gwfloat_p)(yb + yoff[j]) = (acc[i][j][rr] + bv[rr]) + rv[rr][j];
                    yb += (rr & 3) < 3 ? ycs4 : 5 * ycs4;
                }
            } else {
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    const float bv = a.bias ? a.bias[cob + 4 * lh + (rr & 3) + 8 * (rr >> 2)] : 0.0f;
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        if (pok[j]) *(gwfloat_p)(yb + yoff[j]) = acc[i][j][rr] + bv;
                    yb += (rr & 3) < 3 ? ycs4 : 5 * ycs4;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int cob = mw + i * 32 + 4 * lh;
        float bv[16];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int co = cob + (rr & 3) + 8 * (rr >> 2);
            bv[rr] = a.bias ? a.bias[co < d.Cout ? co : d.Cout - 1] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int pp = nw + j * 32 + l31;
            const bool pok = pp < a.Ntot;
            int r = pok ? pp : 0;
            const int qw = r % d.oW; r /= d.oW;
            const int qh = r % d.oH; r /= d.oH;
            const int qd = r % d.oD; const int qb = r / d.oD;
            const int64_t yoff = qb * d.ys[0] + qd * d.ys[2] + qh * d.ys[3] + qw * d.ys[4];
            float rv[16];
            if (a.res) {
                const int64_t roff = qb * d.rs[0] + qd * d.rs[2] + qh * d.rs[3] + qw * d.rs[4];
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    const int co = cob + (rr & 3) + 8 * (rr >> 2);
                    rv[rr] = a.res[roff + (co < d.Cout ? co : d.Cout - 1) * d.rs[1]];
                }
            } else {
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) rv[rr] = 0.0f;
            }
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int co = cob + (rr & 3) + 8 * (rr >> 2);
                if (pok && co < d.Cout) ay[yoff + co * d.ys[1]] = acc[i][j][rr] + bv[rr] + rv[rr];
            }
        }
    }
}

// (register cap by tile: the 64 x 256 tile spilled 21-24 registers under the 4-wave cap and is 4 % faster at 3 waves without them; the
// 128 x 128 tile spills 15-25 too but is 4 % slower at 3 waves -- measured on the sub-pixel / strided convs of the smoke net)
template <int BM, int BN, int WM, int WN, bool FAST>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu((BM == 64 && BN == 256) ? 3 : 4))) SDC_NO_DS_MERGE void conv_kernel(const ConvArgs a) {
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
    const int n0 = xcd_tile(blockIdx.x, gridDim.x) * BN;
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

    auto spatial_k = [&](int kd, int kh, int kw, bool& ok, int& id, int& ih, int& iw) {
        const int vd = vd0 + kd, vh = vh0 + kh, vw = vw0 + kw;
        id = vd >> a.lgD; ih = vh >> a.lgH; iw = vw >> a.lgW;
        ok = pvalid && vd >= 0 && vh >= 0 && vw >= 0 && id < d.iD && ih < d.iH && iw < d.iW &&
             ((vd & mD) | (vh & mH) | (vw & mW)) == 0;
    };
    auto spatial = [&](int tap, bool& ok, int& id, int& ih, int& iw) {
        const int kw = tap % d.kW;
        const int t2 = tap / d.kW;
        spatial_k(t2 / d.kH, t2 % d.kH, kw, ok, id, ih, iw);
    };

    // Loads only ISSUE here: nothing below consumes a loaded value, so no s_waitcnt lands in front of the
    // MFMAs of the current chunk.  Out-of-range elements read a clamped (valid) address and are zeroed in
    // store_chunk, after the MFMAs, where the wait belongs.
    bool bok = false;        // FAST: validity of this thread's position for the chunk in flight
    int kload = 0;           // first k of the chunk in flight
    const int aco_c = acov ? (m0 + aco) : 0;

    // FAST path walks (tap, channel-chunk) incrementally: no integer division and no 64-bit multiply per
    // chunk -- that VALU work otherwise competes with the fp32 MFMAs for the SIMD's issue slots.
    // sdc_conv_splitk (FAST path only; small grids): blockIdx.z workgroups share an output tile, each over a contiguous range of the
    // (tap, channel-chunk) walk; y is then this split's dense partial copy
    const int ksp = FAST ? a.ksplit : 1;
    const int nchunks_all = (a.Ktot + BK - 1) / BK;
    const int kc0 = ksp > 1 ? (int)blockIdx.z * (nchunks_all / ksp) : 0;
    int f_kd = 0, f_kh = 0, f_kw = 0, f_ci = 0;
    if (ksp > 1) {
        const int k0 = kc0 * BK, tap0 = k0 / a.Cin, t2 = tap0 / d.kW;
        f_ci = SDC_UNIFORM(k0 - tap0 * a.Cin);
        f_kw = SDC_UNIFORM(tap0 - t2 * d.kW);
        f_kh = SDC_UNIFORM(t2 % d.kH);
        f_kd = SDC_UNIFORM(t2 / d.kH);
    }
    bool f_ok = false;
    // per-thread 32-bit element offsets of (b, id, ih, iw) inside x0 / x1 for the current tap; the channel
    // part of every address is wave-uniform and stays in scalar registers (global_load saddr + voffset form)
    uint32_t f_v0 = 0, f_v1 = 0;
    const int brow_u = __builtin_amdgcn_readfirstlane(brow0);
    const int arow_u = __builtin_amdgcn_readfirstlane(arow0);
    // (BM = 32: a wave spans two weight rows, the second one is folded into the per-lane offset)
    const uint32_t a_v = (uint32_t)((arow0 - arow_u) * d.Cout + aco_c);
    int a_k = arow_u + kc0 * BK;        // weight row of this wave's first A load in the chunk in flight
    auto fast_tap = [&]() {
        int id, ih, iw;
        spatial_k(f_kd, f_kh, f_kw, f_ok, id, ih, iw);
        f_v0 = f_ok ? (uint32_t)(ob * d.x0s[0] + id * d.x0s[2] + ih * d.x0s[3] + iw * d.x0s[4]) : 0u;
        if (d.Cin1 > 0)
            f_v1 = f_ok ? (uint32_t)(ob * d.x1s[0] + id * d.x1s[2] + ih * d.x1s[3] + iw * d.x1s[4]) : 0u;
    };
    if constexpr (FAST) fast_tap();

    auto load_chunk = [&](int kc) {
        const int kbase = kc * BK;
        kload = kbase;
        if constexpr (FAST) {
            // Ktot is a multiple of BK here: no k clamp
            const gfloat_p ab = uniform_ptr(a.wp + (int64_t)a_k * d.Cout);
            const int astep = SDC_UNIFORM(ASTEP * d.Cout);
#pragma unroll
            for (int i = 0; i < AROWS; ++i) areg[i] = ld_sv(ab + i * astep, a_v * 4u);
            a_k = SDC_UNIFORM(a_k + BK);
            // whole chunk shares one tap and one input tensor
            const float* bsel; int sc; uint32_t voff;
            if (f_ci < d.Cin0) { sc = (int)d.x0s[1]; bsel = a.x0 + (int64_t)(f_ci + brow_u) * sc; voff = f_v0; }
            else { sc = (int)d.x1s[1]; bsel = a.x1 + (int64_t)(f_ci - d.Cin0 + brow_u) * sc; voff = f_v1; }
            const gfloat_p bb = uniform_ptr(bsel);
            bok = f_ok;
            const int step = SDC_UNIFORM(BSTEP * sc);
#pragma unroll
            for (int i = 0; i < BROWS; ++i) breg[i] = ld_sv(bb + i * step, voff * 4u);
            f_ci = SDC_UNIFORM(f_ci + BK);
            if (f_ci >= a.Cin) {
                f_ci = 0;
                if (++f_kw == d.kW) { f_kw = 0; if (++f_kh == d.kH) { f_kh = 0; ++f_kd; } }
                f_kw = SDC_UNIFORM(f_kw); f_kh = SDC_UNIFORM(f_kh); f_kd = SDC_UNIFORM(f_kd);
                if (f_kd < d.kD) fast_tap();
            }
        } else {
#pragma unroll
            for (int i = 0; i < AROWS; ++i) {
                int k = kbase + arow0 + i * ASTEP;
                k = k < a.Ktot ? k : a.Ktot - 1;
                areg[i] = a.wp[(int64_t)k * d.Cout + aco_c];
            }
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
        for (int i = 0; i < AROWS; ++i) {
            const bool kin = FAST || (kload + arow0 + i * ASTEP) < a.Ktot;
            As[buf][arow0 + i * ASTEP][aco] = (acov && kin) ? areg[i] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < BROWS; ++i) {
            if constexpr (FAST)
                Bs[buf][brow0 + i * BSTEP][bj] = bok ? breg[i] : 0.0f;
            else
                Bs[buf][brow0 + i * BSTEP][bj] = breg[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nchunks = ksp > 1 ? nchunks_all / ksp : nchunks_all;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    const int l31 = lane & 31, lh = lane >> 5;
    const int am = wm * (TM * 32) + l31;
    const int bn = wn * (TN * 32) + l31;

    for (int kc = 0; kc < nchunks; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nchunks) load_chunk(kc + 1);
        // all fragments of the chunk are read into registers first (8 k-steps x (TM+TN) ds_read_b32), so the
        // MFMAs below issue back to back instead of stalling on an LDS round trip every k-step
        float af[BK / 2][TM], bf[BK / 2][TN];
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[ks][i] = As[buf][2 * ks + lh][am + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[ks][j] = Bs[buf][2 * ks + lh][bn + j * 32];
        }
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks][i], bf[ks][j], acc[i][j], 0, 0, 0);
        }
        // the chunk's global loads, address arithmetic and LDS fragment reads go into the shadows of its MFMAs (a wave that
        // issues them in one block first leaves the matrix pipe idle meanwhile)
#pragma unroll
        for (int m = 0; m < (BK / 2) * TM * TN; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kc + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }

    conv_epilogue<TM, TN>(a, acc, m0 + wm * (TM * 32), n0 + wn * (TN * 32), lane,
                          ksp > 1 ? a.y + (int64_t)blockIdx.z * a.ypart_elems : nullptr);
}

// ------------------------------------------------------------------------------------------------
// Pointwise (1x1x1, stride 1) form over a dense spatial layout (positions of a sample contiguous, S % 4 == 0): both the
// weight tile and the activation tile of a 16-channel chunk come in with 16-byte loads (4 consecutive output channels /
// 4 consecutive positions per lane) in scalar-base form and go to LDS with ds_write_b128 -- 4 loads per thread per chunk
// instead of 16.  These layers have K = Cin <= 768: few chunks per tile, so the load issue and its latency, not the
// MFMA stream, decide.  Same k-ordered fp32 FMA chains as conv_kernel.
template <int BM, int BN, int WM, int WN>
// (at least 3 waves per SIMD = 3 workgroups per CU: at 2 the short-K layers sit out each other's prologue and epilogue; the
// register cap this implies costs no spill)
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(3))) void conv_pw_kernel(const ConvArgs a) {
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int NB4 = BK * BN / 4 / NT;      // float4 activation loads per thread per chunk
    constexpr int NA4 = BK * BM / 4 / NT;      // float4 weight loads per thread per chunk
    static_assert(NB4 >= 1 && NA4 >= 1, "tile too small");
    typedef float nfloat4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) nfloat4* gfloat4_p;
    typedef const __attribute__((address_space(1))) char* gchar_p;

    __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];

    const SdcConvDesc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // one-dimensional grid, m fastest among the logical ids an XCD runs back to back: the m-tiles of a position tile (3 for the
    // 128 -> 384 q/k/v projections) read the same activation tile -- dispatched a whole grid row apart (blockIdx.y = m) every one
    // of them fetched it from HBM again (PMC: 1.5 x the algorithmic bytes)
    const int MT = (d.Cout + BM - 1) / BM;
    const int lb = xcd_tile(blockIdx.x, gridDim.x);
    const int n0 = (lb / MT) * BN;
    const int m0 = (lb % MT) * BM;
    const int64_t S = (int64_t)d.oD * d.oH * d.oW;

    // per-thread constant parts of the addresses (bytes): (row of the chunk) * channel stride + position offset
    uint32_t bv0[NB4], bv1[NB4], av[NA4];
    bool bok[NB4], aok[NA4];
#pragma unroll
    for (int i = 0; i < NB4; ++i) {
        const int f = tid + i * NT;
        const int row = f / (BN / 4), c4 = (f % (BN / 4)) * 4;
        const int p = n0 + c4;
        bok[i] = p < a.Ntot;
        const int pp = bok[i] ? p : 0;
        const int b = (int)(pp / S);
        const int64_t sp = pp - (int64_t)b * S;
        bv0[i] = (uint32_t)((b * d.x0s[0] + row * d.x0s[1] + sp) * 4);
        bv1[i] = d.Cin1 > 0 ? (uint32_t)((b * d.x1s[0] + row * d.x1s[1] + sp) * 4) : 0u;
    }
#pragma unroll
    for (int i = 0; i < NA4; ++i) {
        const int f = tid + i * NT;
        const int row = f / (BM / 4), c4 = (f % (BM / 4)) * 4;
        aok[i] = (m0 + c4) < d.Cout;
        av[i] = (uint32_t)(((int64_t)row * d.Cout + (aok[i] ? m0 + c4 : 0)) * 4);
    }
    nfloat4 breg[NB4], areg[NA4];
    // sdc_conv_splitk (small grids: the 1-D nets at a per-rank batch of 16-32): blockIdx.y workgroups share an output tile, each over
    // Cin / ksplit input channels; y is then this split's dense partial copy
    const int kc_base = a.ksplit > 1 ? (int)blockIdx.y * (a.Cin / BK / a.ksplit) : 0;
    auto load_chunk = [&](int kc) {
        const int ci = (kc_base + kc) * BK;
        const gfloat_p wb = uniform_ptr(a.wp + (int64_t)ci * d.Cout);
#pragma unroll
        for (int i = 0; i < NA4; ++i) areg[i] = *(gfloat4_p)((gchar_p)wb + av[i]);
        const bool first = ci < d.Cin0;
        const gfloat_p xb = uniform_ptr(first ? a.x0 + (int64_t)ci * d.x0s[1] : a.x1 + (int64_t)(ci - d.Cin0) * d.x1s[1]);
#pragma unroll
        for (int i = 0; i < NB4; ++i) breg[i] = *(gfloat4_p)((gchar_p)xb + (first ? bv0[i] : bv1[i]));
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA4; ++i) {
            const int f = tid + i * NT;
            nfloat4 v = areg[i];
            if (!aok[i]) v = nfloat4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<nfloat4*>(&As[buf][f / (BM / 4)][(f % (BM / 4)) * 4]) = v;
        }
#pragma unroll
        for (int i = 0; i < NB4; ++i) {
            const int f = tid + i * NT;
            nfloat4 v = breg[i];
            if (!bok[i]) v = nfloat4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<nfloat4*>(&Bs[buf][f / (BN / 4)][(f % (BN / 4)) * 4]) = v;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nchunks = a.Cin / BK / a.ksplit;
    load_chunk(0);
    store_chunk(0);
    if (nchunks > 1) load_chunk(1);
    __syncthreads();
    const int l31 = lane & 31, lh = lane >> 5;
    const int am = wm * (TM * 32) + l31, bn = wn * (TN * 32) + l31;
    // (the LDS buffer of a chunk is a compile-time constant, body instantiated per buffer: LDS addresses = lane offset + immediate)
    auto chunk = [&](auto bufc, const int kc) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
        float af[BK / 2][TM], bf[BK / 2][TN];
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[ks][i] = As[buf][2 * ks + lh][am + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[ks][j] = Bs[buf][2 * ks + lh][bn + j * 32];
        }
        // chunk kc+1 is in registers (fetched during chunk kc-1): park it in the other LDS buffer first, then fetch kc+2
        if (kc + 1 < nchunks) store_chunk(buf ^ 1);
        if (kc + 2 < nchunks) load_chunk(kc + 2);
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks][i], bf[ks][j], acc[i][j], 0, 0, 0);
        __syncthreads();
    };
    {
        constexpr std::integral_constant<int, 0> B0{};
        constexpr std::integral_constant<int, 1> B1{};
        int kc = 0;
        for (; kc + 2 <= nchunks; kc += 2) { chunk(B0, kc); chunk(B1, kc + 1); }
        if (kc < nchunks) chunk(B0, kc);
    }
    conv_epilogue<TM, TN>(a, acc, m0 + wm * (TM * 32), n0 + wn * (TN * 32), lane,
                          a.ksplit > 1 ? a.y + (int64_t)blockIdx.y * a.ypart_elems : nullptr);
}

#include "sdc_conv_pw2.inc"

template <int BM, int BN, int WM, int WN>
void launch_pw(const ConvArgs& a, hipStream_t s) {
    dim3 grid(((a.Ntot + BN - 1) / BN) * ((a.d.Cout + BM - 1) / BM));
    hipLaunchKernelGGL((conv_pw_kernel<BM, BN, WM, WN>), grid, dim3(NT), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// Row-halo variant (stride-1 along W, no virtual upsampling, Cin % 16 == 0): the B tile of one
// (kd, kh, channel-chunk) stage is the input ROW SEGMENT with its kW-1 halo columns, staged once and read
// at kW shifted LDS offsets -- the "LDS-staged activation tile".  One stage feeds kW x 8 k-steps, so the
// gather (the measured limiter of the plain kernel: VMEM issue, not latency) drops ~kW-fold, barriers
// too; the weight tile of the stage comes in with 16-byte loads.
// LDS: As[2][KW][BK][BM], Bs[2][BK][NSEG][seg + KW - 1] with seg = min(oW, BN), NSEG = BN / seg.
// GEN: the 16 k rows of a stage are arbitrary (kd, kh, ci) triples (flattened kr = (kd*kH + kh)*Cin + ci), so
// Cin need not be a multiple of 16: the 7x7x7 / 7x7 / k7 stem convs (Cin = 7, 3, 12) run here with KW = 7.
template <int BM, int BN, int WM, int WN, int KW, bool GEN, int BKT = 16>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(BKT == 8 ? 4 : 1))) void conv_rh_kernel(const ConvArgs a) {
    constexpr int BK = BKT;                                     // k rows per stage (the 7-tap stem: 8, so that three workgroups fit a CU's LDS)
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int KSMAX = BN + (BN / 16) * (KW - 1);            // LDS floats per k row, worst case (16-wide rows)
    constexpr int NCOL = (KSMAX + 63) / 64;                     // 64-lane sweeps over one k row
    constexpr int KROWS = BK / (NT / 64);                       // k rows per wave per stage
    constexpr int NA4T = KW * BK * BM / 4;                      // float4 weight loads per stage
    constexpr int NA4 = (NA4T + NT - 1) / NT;                   // ... per thread (the last one partial when NT does not divide them)
    static_assert(NA4 >= 1 && KROWS >= 1, "weight tile too small");
    static_assert(GEN || NA4T % NT == 0, "only the generalized-row form guards a partial weight load");

    __shared__ __attribute__((aligned(16))) float As[2][KW][BK][BM];
    __shared__ float Bs[2][BK * KSMAX];

    const SdcConvDesc& d = a.d;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = SDC_UNIFORM(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int n0 = xcd_tile(blockIdx.x, gridDim.x) * BN;
    const int m0 = blockIdx.y * BM;
    const int l31 = lane & 31, lh = lane >> 5;

    const int seg = d.oW < BN ? d.oW : BN;
    const int nseg = BN / seg;
    const int rowlen = seg + KW - 1;
    const int ks_stride = nseg * rowlen;          // LDS floats per k row
    const bool two = d.Cin1 > 0;

    // ---- gather state: this lane's columns of a k row (the same for every k row and every stage).  A wave
    // loads whole k rows, so the channel part of the address is wave-uniform; only (segment, column) is per lane.
    int v0[NCOL], v1[NCOL];
    uint64_t smask[NCOL];
#pragma unroll
    for (int t = 0; t < NCOL; ++t) {
        const int cidx = lane + 64 * t;
        v0[t] = 0; v1[t] = 0; smask[t] = 0;
        if (cidx < ks_stride) {
            const int sg = cidx / rowlen;
            const int cc = cidx - sg * rowlen;
            const int pseg = n0 + sg * seg;
            if (pseg < a.Ntot) {
                int q = pseg;
                const int ow0 = q % d.oW; q /= d.oW;
                const int oh = q % d.oH; q /= d.oH;
                const int od = q % d.oD; const int ob = q / d.oD;
                const int col = ow0 + cc - d.pW;
                const int id0 = od * d.sD - d.pD, ih0 = oh * d.sH - d.pH;
                uint64_t m = 0;
                if (col >= 0 && col < d.iW) {
                    for (int kd = 0; kd < d.kD; ++kd)
                        for (int kh = 0; kh < d.kH; ++kh)
                            if (id0 + kd >= 0 && id0 + kd < d.iD && ih0 + kh >= 0 && ih0 + kh < d.iH)
                                m |= 1ull << (kd * d.kH + kh);
                }
                smask[t] = m;
                v0[t] = (int)(ob * d.x0s[0] + id0 * d.x0s[2] + ih0 * d.x0s[3] + col * d.x0s[4]);
                if (two) v1[t] = (int)(ob * d.x1s[0] + id0 * d.x1s[2] + ih0 * d.x1s[3] + col * d.x1s[4]);
            }
        }
    }
    // ---- per-lane B fragment offsets inside one k row
    int boff[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int pl = wn * (TN * 32) + j * 32 + l31;
        const int sg = pl / seg;
        boff[j] = sg * rowlen + (pl - sg * seg);
    }
    // ---- weight tile: float4 f = tid + NT*i  ->  row = f / (BM/4) in [0, KW*BK), 4 consecutive co
    int a_col[NA4];
    bool a_ok[NA4];
#pragma unroll
    for (int i = 0; i < NA4; ++i) {
        const int c4 = ((tid + i * NT) % (BM / 4)) * 4;
        a_ok[i] = (m0 + c4) < d.Cout;            // Cout % 4 == 0 (host check): a float4 is all-in or all-out
        a_col[i] = a_ok[i] ? (m0 + c4) : 0;
    }

    float breg[KROWS][NCOL];
    float4 areg[NA4];
    uint64_t mbits = 0;                          // bit r*NCOL+t: (k row r, column sweep t) of the stage in flight is valid
    int s_kd = 0, s_kh = 0, s_ci = 0;            // stage walk: kd, kh outer; channel chunk inner

    int s_st = 0;                                // GEN: stage number (k rows 16*s_st ..)
    const int KR = d.kD * d.kH * a.Cin;          // GEN: number of (kd, kh, ci) rows
    // (runtime integer divisions cost ~25 VALU instructions each, and VALU work is not hidden behind fp32 MFMAs)
    const float r_cin = 1.0f / (float)a.Cin, r_kh = 1.0f / (float)d.kH;
    bool a_rowok[NA4];

    auto load_stage = [&]() {
        if constexpr (!GEN) {
            const int tapbit = s_kd * d.kH + s_kh;
            const int tap0 = tapbit * KW;
#pragma unroll
            for (int i = 0; i < NA4; ++i) {
                const int row = (tid + i * NT) / (BM / 4);
                const int kw = row / BK, kr = row % BK;
                const int64_t wrow = (int64_t)(tap0 + kw) * a.Cin + s_ci + kr;
                areg[i] = *reinterpret_cast<const float4*>(a.wp + wrow * d.Cout + a_col[i]);
                a_rowok[i] = true;
            }
            const bool first = s_ci < d.Cin0;
            const int64_t sc = first ? d.x0s[1] : d.x1s[1];
            const int64_t toff = first ? s_kd * d.x0s[2] + s_kh * d.x0s[3] : s_kd * d.x1s[2] + s_kh * d.x1s[3];
            const float* base = (first ? a.x0 + (int64_t)s_ci * sc : a.x1 + (int64_t)(s_ci - d.Cin0) * sc) +
                                (int64_t)(wave * KROWS) * sc;                      // wave-uniform
            mbits = 0;
            int64_t off[NCOL];
#pragma unroll
            for (int t = 0; t < NCOL; ++t) {
                const bool ok = (smask[t] >> tapbit) & 1u;
                off[t] = ok ? (int64_t)(first ? v0[t] : v1[t]) + toff : 0;          // !ok: a safe in-bounds address
#pragma unroll
                for (int r = 0; r < KROWS; ++r) mbits |= (ok ? 1ull : 0ull) << (r * NCOL + t);
            }
#pragma unroll
            for (int r = 0; r < KROWS; ++r)
#pragma unroll
                for (int t = 0; t < NCOL; ++t)
                    if (t * 64 < ks_stride) breg[r][t] = (base + r * sc)[off[t]];
            s_ci += BK;
            if (s_ci >= a.Cin) { s_ci = 0; if (++s_kh == d.kH) { s_kh = 0; ++s_kd; } }
        } else {
            // weights: LDS row (kw, krl) <- Wp[((kdkh * KW + kw) * Cin + ci)], kr = 16*s_st + krl = kdkh*Cin + ci
#pragma unroll
            for (int i = 0; i < NA4; ++i) {
                const bool rin = tid + i * NT < NA4T;
                const int row = rin ? (tid + i * NT) / (BM / 4) : 0;
                const int kw = row / BK, kr = s_st * BK + row % BK;
                const bool ok = rin && kr < KR;
                const int krc = ok ? kr : 0;
                const int kdkh = (int)(((float)krc + 0.5f) * r_cin), ci = krc - kdkh * a.Cin;      // exact: krc < 2^20
                const int64_t wrow = (int64_t)(kdkh * KW + kw) * a.Cin + ci;
                areg[i] = *reinterpret_cast<const float4*>(a.wp + wrow * d.Cout + a_col[i]);
                a_rowok[i] = ok;
            }
            mbits = 0;
#pragma unroll
            for (int r = 0; r < KROWS; ++r) {
                const int kr = SDC_UNIFORM(s_st * BK + wave * KROWS + r);
                const bool rok = kr < KR;
                const int krc = rok ? kr : 0;
                const int kdkh = (int)(((float)krc + 0.5f) * r_cin), ci = krc - kdkh * a.Cin;
                const int kd = (int)(((float)kdkh + 0.5f) * r_kh), kh = kdkh - kd * d.kH;
                const bool first = ci < d.Cin0;
                const float* base = first ? a.x0 + ci * d.x0s[1] + kd * d.x0s[2] + kh * d.x0s[3]
                                          : a.x1 + (ci - d.Cin0) * d.x1s[1] + kd * d.x1s[2] + kh * d.x1s[3];
#pragma unroll
                for (int t = 0; t < NCOL; ++t) {
                    if (t * 64 < ks_stride) {
                        const bool ok = rok && ((smask[t] >> kdkh) & 1ull);
                        const int64_t off = ok ? (int64_t)(first ? v0[t] : v1[t]) : 0;
                        breg[r][t] = ok ? base[off] : (first ? a.x0 : a.x1)[0];
                        mbits |= (ok ? 1ull : 0ull) << (r * NCOL + t);
                    }
                }
            }
            ++s_st;
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA4; ++i) {
            const int f = tid + i * NT;
            const int row = f / (BM / 4), c4 = (f % (BM / 4)) * 4;
            float4 v = areg[i];
            if (!a_ok[i] || !a_rowok[i]) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (NA4T % NT == 0 || f < NA4T) *reinterpret_cast<float4*>(&As[buf][row / BK][row % BK][c4]) = v;
        }
#pragma unroll
        for (int r = 0; r < KROWS; ++r)
#pragma unroll
            for (int t = 0; t < NCOL; ++t) {
                const int cidx = lane + 64 * t;
                if (cidx < ks_stride) Bs[buf][(wave * KROWS + r) * ks_stride + cidx] = ((mbits >> (r * NCOL + t)) & 1ull) ? breg[r][t] : 0.0f;
            }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nstages = GEN ? (KR + BK - 1) / BK : d.kD * d.kH * (a.Cin / BK);
    load_stage();
    store_stage(0);
    __syncthreads();
    const int am = wm * (TM * 32) + l31;

    // (the LDS buffer of a stage as a compile-time constant, body instantiated per buffer: LDS addresses are lane offset + immediate)
    auto stage = [&](auto bufc, const bool more) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
        if (more) load_stage();
        // The fragments of tap kw+1 are read (all of them, ahead of the fence) while the MFMAs of tap kw run: left to itself
        // the scheduler sinks every LDS read next to its use and the wave waits out the LDS latency once per 4 MFMAs.
        float af[2][BK / 2][TM], bf[2][BK / 2][TN];
        auto read_tap = [&](int kw, int set) {
#pragma unroll
            for (int ks = 0; ks < BK / 2; ++ks) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[set][ks][i] = As[buf][kw][2 * ks + lh][am + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[set][ks][j] = Bs[buf][(2 * ks + lh) * ks_stride + boff[j] + kw];
            }
        };
        read_tap(0, 0);
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) {
            const int set = kw & 1;
            if (kw + 1 < KW) read_tap(kw + 1, set ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < BK / 2; ++ks)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][ks][i], bf[set][ks][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) store_stage(buf ^ 1);
        __syncthreads();
    };
    {
        constexpr std::integral_constant<int, 0> B0{};
        constexpr std::integral_constant<int, 1> B1{};
        int st = 0;
        for (; st + 2 <= nstages; st += 2) { stage(B0, true); stage(B1, st + 2 < nstages); }
        if (st < nstages) stage(B0, false);
    }
    conv_epilogue<TM, TN>(a, acc, m0 + wm * (TM * 32), n0 + wn * (TN * 32), lane);
}

// ------------------------------------------------------------------------------------------------
// precision = 2: fp32 Winograd F(2,3) along W on top of the row-halo staging (3-wide taps, stride 1).  For an
// output pair (2j, 2j+1) of a row and the four inputs d0..d3 under it, with the taps g0..g2:
//     m0 = (d0-d2) g0,  m1 = (d1+d2)(g0+g1+g2)/2,  m2 = (d2-d1)(g0-g1+g2)/2,  m3 = (d1-d3) g2
//     y[2j] = m0 + m1 + m2,   y[2j+1] = m1 - m2 - m3
// i.e. four GEMMs over K = kD*kH*Cin with N = positions/2 instead of three with N = positions: 2/3 of the fp32
// MFMA work of the direct form.  The caller stores the transformed taps Wg[(kdkh*4 + xi)*Cin + ci][Cout] behind
// Wp; the input transform is two adds on the B fragment while it is read from the staged row (ds_read_b64 of
// (d0,d1) and (d2,d3)), the output transform runs on the accumulators in the epilogue.  Still fp32 end to end
// (differences to the direct kernel are rounding-order only, ~1e-7 relative), but not bit-identical to it.
// GN (sdc_conv_gn): the GroupNorm statistics of the output are summed here, on the values as they are stored, instead
// of in a second pass over y: per lane for every 8-row block (i, rr >> 2) of its rows, reduced over the wave, then over
// the waves of the workgroup through `scratch` (the dead staging LDS) in a fixed order -> one fp64 (sum, sum of
// squares) pair per (sample, group, part), summed by sdc_gn_finalize.  Deterministic: no atomics.
// NX = 6: the F(4,3) form (1-D convs, see conv_wg_kernel): four outputs per tile from six products,
//     y0 = m0+m1+m2+m3+m4,  y1 = (m1-m2) + 2(m3-m4),  y2 = (m1+m2) + 4(m3+m4),  y3 = (m1-m2) + 8(m3-m4) + m5.
template <int TM, int TP, int BM, int BN, int WM, int WN, int NX = 4>
__device__ __forceinline__ void wg_epilogue(const ConvArgs& a, f32x16 (&acc)[NX][TM][TP], int mw, int pw, int lane,
                                            float* scratch, int wave, int m0, int n0, bool active = true, float* ybase = nullptr) {
    constexpr int NO = NX == 6 ? 4 : 2;
    const SdcConvDesc& d = a.d;
    float* const yb = ybase ? ybase : a.y;       // (sdc_conv_splitk: this split's partial copy)
    const int l31 = lane & 31, lh = lane >> 5;
    const bool v2 = a.vec2;
    const bool gn = a.gn_part != nullptr;
    // fp64 from the first add on: the sums are then independent of how the tile grid cuts the sample (1e-16), so a
    // trajectory's statistics do not depend on the batch it is launched with
    double gs[TM][4], gq[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int b4 = 0; b4 < 4; ++b4) { gs[i][b4] = 0.0; gq[i][b4] = 0.0; }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int cob = mw + i * 32 + 4 * lh;
        float bv[16];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int co = cob + (rr & 3) + 8 * (rr >> 2);
            bv[rr] = a.bias ? a.bias[co < d.Cout ? co : d.Cout - 1] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int pp = NO * (pw + j * 32 + l31);         // first position of the tile; the tile lies in one row (oW % NO == 0)
            const bool pok = pp < a.Ntot && active;          // (the second k-half of a split workgroup only joins the barriers)
            int r = pok ? pp : 0;
            const int qw = r % d.oW; r /= d.oW;
            const int qh = r % d.oH; r /= d.oH;
            const int qd = r % d.oD; const int qb = r / d.oD;
            const int64_t yoff = qb * d.ys[0] + qd * d.ys[2] + qh * d.ys[3] + qw * d.ys[4];
            if constexpr (NX == 6) {
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    const int co = cob + (rr & 3) + 8 * (rr >> 2);
                    float rv[4] = {0.f, 0.f, 0.f, 0.f};
                    if (a.res) {
                        const int64_t roff = qb * d.rs[0] + qd * d.rs[2] + qh * d.rs[3] + qw * d.rs[4];
                        const float* rp = a.res + roff + (co < d.Cout ? co : d.Cout - 1) * d.rs[1];
                        if (v2) {
                            const float2 t0 = *reinterpret_cast<const float2*>(rp), t1 = *reinterpret_cast<const float2*>(rp + 2);
                            rv[0] = t0.x; rv[1] = t0.y; rv[2] = t1.x; rv[3] = t1.y;
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; ++q) rv[q] = rp[q * d.rs[4]];
                        }
                    }
                    const float m0 = acc[0][i][j][rr], m1 = acc[1][i][j][rr], m2 = acc[2][i][j][rr], m3 = acc[3][i][j][rr],
                                m4 = acc[4][i][j][rr], m5 = acc[5][i][j][rr];
                    const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                    float yv[4];
                    yv[0] = ((m0 + s12) + s34) + bv[rr] + rv[0];
                    yv[1] = (d12 + 2.0f * d34) + bv[rr] + rv[1];
                    yv[2] = (s12 + 4.0f * s34) + bv[rr] + rv[2];
                    yv[3] = ((d12 + 8.0f * d34) + m5) + bv[rr] + rv[3];
                    if (pok && co < d.Cout) {
                        float* yp = yb + yoff + co * d.ys[1];
                        if (v2) {
                            *reinterpret_cast<float2*>(yp) = make_float2(yv[0], yv[1]);
                            *reinterpret_cast<float2*>(yp + 2) = make_float2(yv[2], yv[3]);
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; ++q) yp[q * d.ys[4]] = yv[q];
                        }
                        if (gn) {
                            gs[i][rr >> 2] += ((double)yv[0] + (double)yv[1]) + ((double)yv[2] + (double)yv[3]);
                            gq[i][rr >> 2] += ((double)yv[0] * yv[0] + (double)yv[1] * yv[1]) + ((double)yv[2] * yv[2] + (double)yv[3] * yv[3]);
                        }
                    }
                }
            } else {
            float r0[16], r1[16];
            if (a.res) {
                const int64_t roff = qb * d.rs[0] + qd * d.rs[2] + qh * d.rs[3] + qw * d.rs[4];
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    const int co = cob + (rr & 3) + 8 * (rr >> 2);
                    const float* rp = a.res + roff + (co < d.Cout ? co : d.Cout - 1) * d.rs[1];
                    if (v2) { const float2 t = *reinterpret_cast<const float2*>(rp); r0[rr] = t.x; r1[rr] = t.y; }
                    else { r0[rr] = rp[0]; r1[rr] = rp[d.rs[4]]; }
                }
            } else {
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) { r0[rr] = 0.0f; r1[rr] = 0.0f; }
            }
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int co = cob + (rr & 3) + 8 * (rr >> 2);
                const float m0 = acc[0][i][j][rr], m1 = acc[1][i][j][rr], m2 = acc[2][i][j][rr], m3 = acc[3][i][j][rr];
                const float y0 = ((m0 + m1) + m2) + bv[rr] + r0[rr];
                const float y1 = ((m1 - m2) - m3) + bv[rr] + r1[rr];
                if (pok && co < d.Cout) {
                    float* yp = yb + yoff + co * d.ys[1];
                    if (v2) *reinterpret_cast<float2*>(yp) = make_float2(y0, y1);
                    else { yp[0] = y0; yp[d.ys[4]] = y1; }
                    if (gn) { gs[i][rr >> 2] += (double)y0 + (double)y1; gq[i][rr >> 2] += (double)y0 * y0 + (double)y1 * y1; }
                }
            }
            }
        }
    }
    if (gn) {
        double* scr = reinterpret_cast<double*>(scratch);
        // scr[wave][TM*4][2]; 8-row block t of the workgroup's rows = (wm = t / (TM*4), k = t % (TM*4)), summed over wn
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int b4 = 0; b4 < 4; ++b4) {
                const double s1 = sdc::wave_sum(gs[i][b4]), q1 = sdc::wave_sum(gq[i][b4]);
                // (the idle k-half of a split workgroup shares its partner's slot: it must not write)
                if (lane == 0 && active) { scr[(wave * TM * 4 + i * 4 + b4) * 2] = s1; scr[(wave * TM * 4 + i * 4 + b4) * 2 + 1] = q1; }
            }
        __syncthreads();
        const int ngl = a.gn_cpg >= BM ? 1 : BM / a.gn_cpg;       // groups inside this workgroup's rows
        const int t = threadIdx.x;
        if (t < ngl) {
            const int r0 = a.gn_cpg >= BM ? 0 : t * a.gn_cpg, r1 = a.gn_cpg >= BM ? BM : r0 + a.gn_cpg;   // local rows
            double s = 0.0, q = 0.0;
            for (int blk = r0 / 8; blk < r1 / 8; ++blk) {
                const int wmi = blk / (TM * 4), k = blk - wmi * (TM * 4);
                for (int wni = 0; wni < WN; ++wni) {
                    s += scr[((wmi * WN + wni) * TM * 4 + k) * 2];
                    q += scr[((wmi * WN + wni) * TM * 4 + k) * 2 + 1];
                }
            }
            if (m0 + r0 < d.Cout) {
                const int g = (m0 + r0) / a.gn_cpg;
                const int b = n0 / a.gn_S, ntl = (n0 - b * a.gn_S) / BN;
                const int idx = a.gn_cpg >= BM ? ntl * (a.gn_cpg / BM) + (m0 - g * a.gn_cpg) / BM : ntl;
                double* pp = a.gn_part + (((int64_t)b * a.gn_G + g) * a.gn_nparts + idx) * 2;
                pp[0] = s; pp[1] = q;
            }
        }
    }
}

// UPS: the input is read through a virtual nearest-neighbour x2 upsampling along H and/or W (Upsample + conv,
// 1D/model/unet.py:24-37): the staged row is the upsampled one (column >> 1), and the source row of tap kh,
// (oh - pH + kh) >> 1, differs per output row, so its offset is kept per lane for the (<= 3) kh taps; kD = 1, one input.
// NX = 6: F(4,3) along W for the 1-D convs (kD = kH = 1: the tokamak Unet1D, 85 % of its FLOPs): an output quad and the six
// inputs d0..d5 under it,
//     V = B^T d:  4 d0 - 5 d2 + d4 | (d4 - 4 d2) +- (d3 - 4 d1) | (d4 - d2) +- 2 (d3 - d1) | 4 d1 - 5 d3 + d5,
// six GEMMs with N = positions / 4 instead of three with N = positions: HALF the fp32 MFMA work of the direct form (F(2,3): 2/3).
// The six taps G g (G rows (1/4,0,0), (-1/6,-1/6,-1/6), (-1/6,1/6,-1/6), (1/24,1/12,1/6), (1/24,-1/12,1/6), (0,0,1)), formed in
// fp64 and rounded once, follow the F(2,3) taps in the packed weight.  Twelve VALU operations per k-step and tile feed six
// MFMAs.  The larger transform constants cost accuracy: ~5e-6 of the output scale against fp64 (F(2,3): ~1e-6).
// KS = 2: the C3 layers are ~1024 wave tiles each -- one per SIMD -- and a lone wave cannot hide its transform / staging VALU
// behind its own MFMAs (measured: 65 TFLOP/s issued); the k-steps of every stage are therefore split over TWO waves per SIMD
// that own the same output tile, and the second half's accumulators are added through the (dead) staging LDS before the epilogue.
// P1: kD * kH == 1 (the 1-D nets' Conv1d): one tap row, so a lane's gather offsets and padding mask are the same for every stage --
// formed once before the loop instead of ~40 VALU instructions per stage (64-bit mask shifts, selects, address sums)
template <int BM, int BN, int WM, int WN, int SK, int NTH, bool UPS, int NX = 4, int KS = 1, bool P1 = false>
__global__ __launch_bounds__(NTH) void conv_wg_kernel(const ConvArgs a) {
    static_assert(NX == 4 || (NX == 6 && !UPS), "F(4,3) serves the plain 1-D convs");
    static_assert(KS == 1 || (KS == 2 && SK % 8 == 0), "k-split");
    constexpr int NO = NX == 6 ? 4 : 2;                         // outputs per tile
    constexpr int NF2 = NX / 2;                                 // float2 reads per tile: d0..d3 / d0..d5
    constexpr int TM = BM / WM / 32;
    constexpr int TP = BN / NO / WN / 32;                       // 32-tile column tiles per wave
    constexpr int KSMAX = BN + (BN / 16) * 2;                   // LDS floats per k row, worst case (16-wide rows)
    constexpr int NCOL = (KSMAX + 63) / 64;
    constexpr int KROWS = SK / (NTH / 64);
    constexpr int NA4 = NX * SK * BM / 4 / NTH;                 // float4 weight loads per thread per stage
    static_assert(TM >= 1 && TP >= 1 && KROWS >= 1 && NA4 >= 1 && WM * WN * KS * 64 == NTH, "bad tile");
    extern __shared__ __attribute__((aligned(16))) float ldsw[];
    constexpr int PITCH = NCOL * 64 + 8;                        // LDS floats per staged k row: every (lane, sweep) slot exists, so the
                                                                // staging stores need no per-lane predicate (+8: bank spread of the two half-waves)
    constexpr int ASZ = NX * SK * BM, BSZ = SK * PITCH;         // floats per buffer
    float* const As = ldsw;                                     // [2][NX][SK][BM]
    float* const Bs = ldsw + 2 * ASZ;                           // [2][SK * ks_stride]

    const SdcConvDesc& d = a.d;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = SDC_UNIFORM(tid >> 6);
    const int kh = KS == 1 ? 0 : wave / (WM * WN);              // k-half this wave multiplies
    const int wv = KS == 1 ? wave : wave % (WM * WN);
    const int wm = wv / WN, wn = wv % WN;
    const int n0 = xcd_tile(blockIdx.x, gridDim.x) * BN;
    const int m0 = blockIdx.y * BM;
    const int l31 = lane & 31, lh = lane >> 5;

    const int seg = d.oW < BN ? d.oW : BN;
    const int nseg = BN / seg;
    const int rowlen = seg + 2;
    const int ks_stride = nseg * rowlen;          // even: seg is even
    const bool two = d.Cin1 > 0;

    int v0[NCOL], v1[NCOL];
    int hoff[UPS ? NCOL : 1][3];
    uint64_t smask[NCOL];
    // Gather state.  n0 is decomposed once (wave-uniform); a segment adds whole output rows to it (nseg > 1 only when a
    // segment IS a row), and the small per-lane quotients (segment of a column, (b, od, oh) of a row number) are exact
    // float-reciprocal divisions -- runtime integer divisions here cost ~10K cycles per tile before the first MFMA.
    int q0 = n0;
    const int ow_b = q0 % d.oW; q0 /= d.oW;
    const int row_b = q0;                                       // flattened (b, od, oh) of the tile's first position
    const bool fdiv = (int64_t)d.B * d.oD * d.oH < (1 << 20);   // quotients below 2^20: the float form is exact
    const float r_rowlen = 1.0f / (float)rowlen, r_oH = 1.0f / (float)d.oH, r_oD = 1.0f / (float)d.oD;
#pragma unroll
    for (int t = 0; t < NCOL; ++t) {
        const int cidx = lane + 64 * t;
        v0[t] = 0; v1[t] = 0; smask[t] = 0;
        if constexpr (UPS) { hoff[t][0] = 0; hoff[t][1] = 0; hoff[t][2] = 0; }
        if (cidx < ks_stride) {
            const int sg = (int)(((float)cidx + 0.5f) * r_rowlen);
            const int cc = cidx - sg * rowlen;
            const int pseg = n0 + sg * seg;
            if (pseg < a.Ntot) {
                const int ow0 = ow_b;                           // n0 + sg*seg starts a row whenever sg > 0
                const int row = row_b + (nseg > 1 ? sg : 0);
                int oh, od, ob;
                if (fdiv) {
                    const int t2 = (int)(((float)row + 0.5f) * r_oH);
                    oh = row - t2 * d.oH;
                    ob = (int)(((float)t2 + 0.5f) * r_oD);
                    od = t2 - ob * d.oD;
                } else {
                    int q = row;
                    oh = q % d.oH; q /= d.oH;
                    od = q % d.oD; ob = q / d.oD;
                }
                const int col = ow0 + cc - d.pW;
                const int id0 = od * d.sD - d.pD, ih0 = oh * d.sH - d.pH;
                uint64_t m = 0;
                if constexpr (UPS) {
                    if (col >= 0 && col < (d.iW << a.lgW)) {
                        for (int kh = 0; kh < d.kH; ++kh)
                            if (ih0 + kh >= 0 && ih0 + kh < (d.iH << a.lgH)) {
                                m |= 1ull << kh;
                                hoff[t][kh] = (int)(((ih0 + kh) >> a.lgH) * d.x0s[3]);
                            }
                    }
                    smask[t] = m;
                    v0[t] = (int)(ob * d.x0s[0] + id0 * d.x0s[2] + (col >> a.lgW) * d.x0s[4]);
                } else {
                    if (col >= 0 && col < d.iW) {
                        // taps (kd, kh) inside the input: a range of kh per kd, as bit fields
                        const int h_lo = ih0 < 0 ? -ih0 : 0, h_hi = (d.iH - ih0) < d.kH ? (d.iH - ih0) : d.kH;
                        const int d_lo = id0 < 0 ? -id0 : 0, d_hi = (d.iD - id0) < d.kD ? (d.iD - id0) : d.kD;
                        if (h_hi > h_lo) {
                            const uint64_t mh = ((1ull << h_hi) - 1) & ~((1ull << h_lo) - 1);
                            for (int kd = d_lo; kd < d_hi; ++kd) m |= mh << (kd * d.kH);
                        }
                    }
                    smask[t] = m;
                    v0[t] = (int)(ob * d.x0s[0] + id0 * d.x0s[2] + ih0 * d.x0s[3] + col * d.x0s[4]);
                    if (two) v1[t] = (int)(ob * d.x1s[0] + id0 * d.x1s[2] + ih0 * d.x1s[3] + col * d.x1s[4]);
                }
            }
        }
    }
    // per-lane offset of (d0, d1) of this lane's output pair inside one staged k row (even -> 8-byte aligned)
    int boff[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const int pos = NO * (wn * (TP * 32) + j * 32 + l31);
        const int sg = pos / seg;
        boff[j] = sg * rowlen + (pos - sg * seg);
    }
    // weight fetch: lane part of the address fixed for the whole kernel (row (xi, k row) and 4 output channels), the
    // (tap, channel chunk) part wave-uniform -> `global_load_dwordx4 v, voff, s[base]`, no address arithmetic per load
    uint32_t a_voff[NA4];
    bool a_ok[NA4];
#pragma unroll
    for (int i = 0; i < NA4; ++i) {
        const int f = tid + i * NTH;
        const int c4 = (f % (BM / 4)) * 4, row = f / (BM / 4);
        const int xi = row / SK, kr = row % SK;
        a_ok[i] = (m0 + c4) < d.Cout;
        a_voff[i] = (uint32_t)(((int64_t)(xi * a.Cin + kr) * d.Cout + (a_ok[i] ? (m0 + c4) : 0)) * 4);
    }
    // transformed taps, NX per (kd, kh): the F(2,3) section follows Wp, the F(4,3) section (1-D convs only) follows that
    const float* wg = a.wp + (int64_t)a.Ktot * d.Cout + (NX == 6 ? (int64_t)(a.Ktot / 3 * 4) * d.Cout : 0);

    float breg[KROWS][NCOL];
    float4 areg[NA4];
    uint32_t mbits = 0;
    // sdc_conv_splitk on the single-tap-row form (P1; the 1-D nets at small batch leave their layers on 64 workgroups): blockIdx.z
    // workgroups share an output tile, each over Cin / ksplit input channels; y is then this split's dense partial copy
    const int ksp = P1 ? a.ksplit : 1;
    const int kbeg = ksp > 1 ? (int)blockIdx.z * (a.Cin / ksp) : 0, kend = ksp > 1 ? kbeg + a.Cin / ksp : a.Cin;
    int s_kd = 0, s_kh = 0, s_ci = kbeg;

    // The fetch of a stage is cut into NSL pieces that are issued BETWEEN the MFMA groups of the main loop (a
    // workgroup's loads all at once keep the CU's one vector-memory pipe -- 64 B/clk -- busy for hundreds of cycles
    // with the matrix cores idle behind it).  Stage st+2 is fetched to registers during the second half of
    // stage st and written to the other LDS buffer during the first half of stage st+1.
    constexpr int NSL = SK / 4;                                  // pieces (= half of the k-steps of a stage)
    int l_tap = 0, l_ci = 0;
    const float* l_base = a.x0;
    int64_t l_sc = 0;
    uint32_t l_off[NCOL];                         // byte offsets (every operand spans < 2^30 elements: host check `small`)
    uint32_t p1_off0[P1 ? NCOL : 1], p1_off1[P1 ? NCOL : 1], p1_bits = 0;     // P1: the stage-invariant offsets of the two inputs, the mask
    if constexpr (P1) {
#pragma unroll
        for (int t = 0; t < NCOL; ++t) {
            const bool ok = smask[t] & 1u;
            // (UPS with one tap row: the source row of tap kh = 0 is the same for every stage too)
            int src = v0[t];
            if constexpr (UPS) src += hoff[t][0];
            p1_off0[t] = ok ? (uint32_t)src * 4u : 0u;
            p1_off1[t] = ok ? (uint32_t)(two ? v1[t] : v0[t]) * 4u : 0u;
            p1_bits |= (ok ? 1u : 0u) << t;
        }
    }
    auto load_begin = [&]() {
        if constexpr (P1) {
            l_tap = 0;
            l_ci = s_ci;
            const bool first = s_ci < d.Cin0;
            l_sc = first ? d.x0s[1] : d.x1s[1];
            l_base = (first ? a.x0 + (int64_t)s_ci * l_sc : a.x1 + (int64_t)(s_ci - d.Cin0) * l_sc) + (int64_t)(wave * KROWS) * l_sc;
#pragma unroll
            for (int t = 0; t < NCOL; ++t) l_off[t] = first ? p1_off0[t] : p1_off1[t];
            mbits = p1_bits;
            s_ci += SK;
            if (s_ci >= kend) s_ci = kbeg;
            return;
        }
        l_tap = s_kd * d.kH + s_kh;
        l_ci = s_ci;
        const bool first = s_ci < d.Cin0;
        l_sc = first ? d.x0s[1] : d.x1s[1];
        const int64_t toff = UPS ? 0 : (first ? s_kd * d.x0s[2] + s_kh * d.x0s[3] : s_kd * d.x1s[2] + s_kh * d.x1s[3]);
        l_base = (first ? a.x0 + (int64_t)s_ci * l_sc : a.x1 + (int64_t)(s_ci - d.Cin0) * l_sc) + (int64_t)(wave * KROWS) * l_sc;
        mbits = 0;
#pragma unroll
        for (int t = 0; t < NCOL; ++t) {
            const bool ok = (smask[t] >> l_tap) & 1u;
            if constexpr (UPS)
                l_off[t] = ok ? (uint32_t)(v0[t] + (s_kh == 0 ? hoff[t][0] : (s_kh == 1 ? hoff[t][1] : hoff[t][2]))) * 4u : 0u;
            else
                l_off[t] = ok ? (uint32_t)((first ? v0[t] : v1[t]) + (int)toff) * 4u : 0u;
            mbits |= (ok ? 1u : 0u) << t;
        }
        s_ci += SK;
        // (past the last stage the walk wraps to the first one: the two extra fetches of the pipeline tail stay in
        // bounds and are never consumed -- keeping them unconditional keeps the loop free of load-skipping branches)
        if (s_ci >= a.Cin) { s_ci = 0; if (++s_kh == d.kH) { s_kh = 0; if (++s_kd == d.kD) s_kd = 0; } }
    };
    auto load_piece_to = [&](int p, float (&breg)[KROWS][NCOL], float4 (&areg)[NA4]) {
        typedef const __attribute__((address_space(1))) char* gchar_p;
        typedef float nfloat4 __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(1))) nfloat4* gfloat4_p;
        const gfloat_p wbase = uniform_ptr(wg + ((int64_t)(l_tap * NX) * a.Cin + l_ci) * d.Cout);
#pragma unroll
        for (int i = 0; i < NA4; ++i) {
            if (i % NSL != p) continue;
            asm volatile("" : "+v"(a_voff[i]));      // (keeps the load in scalar base + 32-bit lane offset form: otherwise a 64-bit add per load)
            const nfloat4 wv = *(gfloat4_p)((gchar_p)wbase + a_voff[i]);
            areg[i] = make_float4(wv.x, wv.y, wv.z, wv.w);
        }
#pragma unroll
        for (int r = 0; r < KROWS; ++r) {
            const gfloat_p rb = uniform_ptr(l_base + r * l_sc);
#pragma unroll
            for (int t = 0; t < NCOL; ++t)
                if ((NA4 + r * NCOL + t) % NSL == p) breg[r][t] = ld_sv(rb, l_off[t]);   // ks_stride > 64*(NCOL-1)
        }
    };
    auto load_piece = [&](int p) { load_piece_to(p, breg, areg); };
    auto store_piece_from = [&](int buf, int p, const float (&breg)[KROWS][NCOL], const float4 (&areg)[NA4], uint32_t mbits) {
#pragma unroll
        for (int i = 0; i < NA4; ++i) {
            if (i % NSL != p) continue;
            const int f = tid + i * NTH;
            const int row = f / (BM / 4), c4 = (f % (BM / 4)) * 4;
            // (rows beyond Cout were fetched from a valid row (a_voff) and are never stored or summed by the epilogue: no zeroing --
            // it was 16 selects per stage)
            *reinterpret_cast<float4*>(As + buf * ASZ + row * BM + c4) = areg[i];
        }
#pragma unroll
        for (int r = 0; r < KROWS; ++r)
#pragma unroll
            for (int t = 0; t < NCOL; ++t) {
                if ((NA4 + r * NCOL + t) % NSL == p)
                    Bs[buf * BSZ + (wave * KROWS + r) * PITCH + lane + 64 * t] = ((mbits >> t) & 1u) ? breg[r][t] : 0.0f;
            }
    };
    auto store_piece = [&](int buf, int p) { store_piece_from(buf, p, breg, areg, mbits); };

    f32x16 acc[NX][TM][TP];
#pragma unroll
    for (int x = 0; x < NX; ++x)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][i][j][r] = 0.0f;

    const int nstages = d.kD * d.kH * ((kend - kbeg) / SK);
    {   // prologue: the fetches of the first two stages travel together (stage 0 in a register set that dies here, before
        // the accumulators come alive) -- one memory round trip before the first MFMA instead of two
        float breg0[KROWS][NCOL];
        float4 areg0[NA4];
        load_begin();
        const uint32_t mbits0 = mbits;
#pragma unroll
        for (int p = 0; p < NSL; ++p) load_piece_to(p, breg0, areg0);
        load_begin();
#pragma unroll
        for (int p = 0; p < NSL; ++p) load_piece(p);
#pragma unroll
        for (int p = 0; p < NSL; ++p) store_piece_from(0, p, breg0, areg0, mbits0);
    }
    __syncthreads();
    const int am = wm * (TM * 32) + l31;
    // the later-dispatched half of an 8-wave workgroup loses every arbitration against its SIMD partner: static priority
    if (NTH == 512 && wave >= 4) __builtin_amdgcn_s_setprio(1);

    // P1 (an even number of stages: launch_wg): the LDS buffer of a stage is a compile-time constant, the body instantiated per
    // buffer -- LDS addresses become lane offset + immediate instead of ~15 VALU additions per 32 MFMAs
    auto stage = [&](auto bufc) __attribute__((always_inline)) {
        const int buf = bufc;
        constexpr int KST = SK / 2 / KS;                        // k-steps of a stage this wave multiplies
        const float* Ab = As + buf * ASZ + (2 * KST * kh) * BM;
        const float* Bb = Bs + buf * BSZ + (2 * KST * kh) * PITCH;
        const uint32_t mb_prev = mbits;
        float fa[2][NX][TM];
        float2 fb[2][TP][NF2];
        auto read_frag = [&](int ks, int set) {
#pragma unroll
            for (int x = 0; x < NX; ++x)
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[set][x][i] = Ab[(x * SK + 2 * ks + lh) * BM + am + i * 32];
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const float2* bp = reinterpret_cast<const float2*>(Bb + (2 * ks + lh) * PITCH + boff[j]);
#pragma unroll
                for (int q = 0; q < NF2; ++q) fb[set][j][q] = bp[q];
            }
        };
        // Input transform of a k-step's B fragments is done one step ahead, and the LDS reads, the staging piece and
        // that transform are spread over the shadows of the step's MFMAs (sched_group_barrier): a wave that issues its
        // MFMAs back to back and only then its other instructions leaves the matrix pipe idle while it catches up --
        // the two waves of a SIMD drift into running one after the other, so nobody else fills those gaps.
        float bt[2][TP][NX];
        auto transform = [&](int set) {
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                if constexpr (NX == 6) {
                    const float d0 = fb[set][j][0].x, d1 = fb[set][j][0].y, d2 = fb[set][j][1].x, d3 = fb[set][j][1].y,
                                d4 = fb[set][j][2].x, d5 = fb[set][j][2].y;
                    const float e42 = fmaf(-4.0f, d2, d4), e31 = fmaf(-4.0f, d1, d3);
                    const float c42 = d4 - d2, c31 = d3 - d1;
                    bt[set][j][0] = fmaf(4.0f, d0, fmaf(-5.0f, d2, d4));
                    bt[set][j][1] = e42 + e31;
                    bt[set][j][2] = e42 - e31;
                    bt[set][j][3] = fmaf(2.0f, c31, c42);
                    bt[set][j][4] = fmaf(-2.0f, c31, c42);
                    bt[set][j][5] = fmaf(4.0f, d1, fmaf(-5.0f, d3, d5));
                } else {
                    const float2 p0 = fb[set][j][0], p1 = fb[set][j][1];
                    bt[set][j][0] = p0.x - p1.x;
                    bt[set][j][1] = p0.y + p1.x;
                    bt[set][j][2] = p1.x - p0.y;
                    bt[set][j][3] = p0.y - p1.y;
                }
            }
        };
        read_frag(0, 0);
        transform(0);
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) {
            const int set = ks & 1;
            if (ks + 1 < KST) read_frag(ks + 1, set ^ 1);
            if constexpr (KS == 2) {
                // half as many k-steps per wave: every step parks one piece of stage st+1 and fetches the same piece of stage st+2
                static_assert(KS == 1 || KST == NSL, "one staging piece per k-step");
                store_piece_from(buf ^ 1, ks, breg, areg, mb_prev);
                if (ks == 0) load_begin();
                load_piece(ks);
            } else if (ks < NSL) {
                store_piece(buf ^ 1, ks);
            } else {
                if (ks == NSL) load_begin();
                load_piece(ks - NSL);
            }
#pragma unroll
            for (int x = 0; x < NX; ++x)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TP; ++j)
                        acc[x][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][x][i], bt[set][j][x], acc[x][i][j], 0, 0, 0);
            if (ks + 1 < KST) transform(set ^ 1);
            // interleave: the next fragments' LDS reads behind the first MFMA, then per MFMA a few VALU and one staging
            // access (LDS write / global load)
#pragma unroll
            for (int m = 0; m < NX * TM * TP; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (m < 2) __builtin_amdgcn_sched_group_barrier(0x100, (NX / 2) * TM + NF2 * TP, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x220, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    if constexpr (P1) {
        constexpr std::integral_constant<int, 0> B0{};
        constexpr std::integral_constant<int, 1> B1{};
        for (int st = 0; st + 2 < nstages; st += 2) { stage(B0); stage(B1); }
        stage(B0);
        stage(B1);
    } else {
        for (int st = 0; st < nstages; ++st) stage(st & 1);
    }
    if constexpr (KS == 2) {
        // the two k-halves of a tile meet in the staging LDS (dead after the last stage's barrier): [tile wave][register][lane]
        float* red = ldsw + (size_t)wv * (NX * TM * TP * 16) * 64 + lane;
        if (kh == 1) {
#pragma unroll
            for (int x = 0; x < NX; ++x)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TP; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) red[(((x * TM + i) * TP + j) * 16 + r) * 64] = acc[x][i][j][r];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int x = 0; x < NX; ++x)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TP; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[x][i][j][r] += red[(((x * TM + i) * TP + j) * 16 + r) * 64];
        }
        __syncthreads();
    }
    wg_epilogue<TM, TP, BM, BN, WM, WN, NX>(a, acc, m0 + wm * (TM * 32), n0 / NO + wn * (TP * 32), lane, ldsw, wv, m0, n0, kh == 0,
                                            ksp > 1 ? a.y + (int64_t)blockIdx.z * a.ypart_elems : nullptr);
}


// ------------------------------------------------------------------------------------------------
// F(4,3) along W for the 1-D convs (kD = kH = 1, stride 1, pad 1: the tokamak Unet1D's Conv1d k3, 85 % of its FLOPs) with
// the input transform done ONCE per workgroup.  conv_wg_kernel<.., NX = 6> forms V = B^T d per wave at fragment-read time:
// the four waves that share a position tile each repeat the twelve operations per k-step, in front of their own six MFMAs,
// where they are not hidden (measured: 74 TFLOP/s issued, no faster than F(2,3) at 97).  Here the pipeline is one stage
// deeper on the B side:   stage s+3 -> registers | raw rows of s+2 -> LDS | V(s+1) = B^T d (one (k row, quad) per thread,
// 12 operations per STAGE) -> LDS | MFMAs of stage s read V(s) and the six taps as they lie.
// 128 x 128 outputs per workgroup, 512 threads: waves (wm, kh) -- four 32-row tiles x the two halves of a stage's k-steps
// (two waves per SIMD on the same accumulators' tile; the halves meet in LDS before the epilogue).  One barrier per stage.
//   V = (4 d0 - 5 d2 + d4 | (d4 - 4 d2) +- (d3 - 4 d1) | (d4 - d2) +- 2 (d3 - d1) | 4 d1 - 5 d3 + d5),  six GEMMs over K = Cin.
template <int BM, int BN, int WM, int SK>
__global__ __launch_bounds__(512) void conv_f43_kernel(const ConvArgs a) {
    constexpr int NX = 6, NTH = 512, KSP = 2;
    constexpr int TM = BM / WM / 32, NQ = BN / 4;
    constexpr int KSMAX = BN + (BN / 16) * 2, NCOL = (KSMAX + 63) / 64, PITCH = NCOL * 64 + 8;
    constexpr int KROWS = SK / (NTH / 64), NA4 = NX * SK * BM / 4 / NTH, NSL = 4, KST = SK / 2 / KSP;
    constexpr int VP = NX * NQ + 32;                            // V row of one k: [xi][quad], +32: the two half-waves on disjoint banks
    constexpr int ASZ = NX * SK * BM, BSZ = SK * PITCH, VSZ = SK * VP;
    static_assert(TM == 1 && NQ == 32 && WM * KSP * 64 == NTH && KST == NSL && KROWS >= 1 && NA4 >= 1 && SK * NQ == NTH, "tile");
    extern __shared__ __attribute__((aligned(16))) float ldsw[];
    float* const As = ldsw;                                     // [2][NX][SK][BM]
    float* const Bs = ldsw + 2 * ASZ;                           // [2][SK][PITCH]   raw rows with their halo columns
    float* const Vs = Bs + 2 * BSZ;                             // [2][SK][VP]      transformed

    const SdcConvDesc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = SDC_UNIFORM(tid >> 6);
    const int kh = wave / WM, wm = wave % WM;
    const int n0 = xcd_tile(blockIdx.x, gridDim.x) * BN;
    const int m0 = blockIdx.y * BM;
    const int l31 = lane & 31, lh = lane >> 5;
    const int seg = d.oW < BN ? d.oW : BN, nseg = BN / seg, rowlen = seg + 2, ks_stride = nseg * rowlen;
    const bool two = d.Cin1 > 0;

    // gather state of this lane's columns of a staged k row (1-D: a segment is a row of a sample, or a piece of one)
    int v0[NCOL], v1[NCOL];
    uint32_t okm = 0;
    {
        const int ow_b = n0 % d.oW, row_b = n0 / d.oW;
#pragma unroll
        for (int t = 0; t < NCOL; ++t) {
            const int cidx = lane + 64 * t;
            v0[t] = 0; v1[t] = 0;
            if (cidx < ks_stride) {
                const int sg = cidx / rowlen, cc = cidx - sg * rowlen;
                if (n0 + sg * seg < a.Ntot) {
                    const int ob = row_b + (nseg > 1 ? sg : 0);
                    const int col = ow_b + cc - d.pW;
                    if (col >= 0 && col < d.iW) okm |= 1u << t;
                    v0[t] = (int)(ob * d.x0s[0] + col * d.x0s[4]);
                    if (two) v1[t] = (int)(ob * d.x1s[0] + col * d.x1s[4]);
                }
            }
        }
    }
    uint32_t a_voff[NA4];
    bool a_ok[NA4];
#pragma unroll
    for (int i = 0; i < NA4; ++i) {
        const int f = tid + i * NTH;
        const int c4 = (f % (BM / 4)) * 4, row = f / (BM / 4);
        const int xi = row / SK, kr = row % SK;
        a_ok[i] = (m0 + c4) < d.Cout;
        a_voff[i] = (uint32_t)(((int64_t)(xi * a.Cin + kr) * d.Cout + (a_ok[i] ? (m0 + c4) : 0)) * 4);
    }
    const float* wg = a.wp + (int64_t)a.Ktot * d.Cout + (int64_t)(a.Ktot / 3 * 4) * d.Cout;     // Wp | F(2,3) taps | F(4,3) taps
    // transform item of this thread: (k row, quad) -> raw offset of d0 (even: 8-byte aligned), V offset of component 0
    const int tk = tid >> 5, tq = tid & 31;
    int t_src, t_dst;
    {
        const int pos = 4 * tq, sg = pos / seg;
        t_src = tk * PITCH + sg * rowlen + (pos - sg * seg);
        t_dst = tk * VP + tq;
    }

    typedef const __attribute__((address_space(1))) char* gchar_p;
    typedef float nfloat4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) nfloat4* gfloat4_p;
    float breg[KROWS][NCOL];
    float4 areg[NA4];
    int ca = 0, cb = 0;                                         // channel walks of the weight and of the input fetches
    auto fetch_a = [&](int p, float4 (&ar)[NA4], int ci) {
        const gfloat_p wbase = uniform_ptr(wg + (int64_t)ci * d.Cout);
#pragma unroll
        for (int i = 0; i < NA4; ++i) {
            if (i % NSL != p) continue;
            const nfloat4 wv = *(gfloat4_p)((gchar_p)wbase + a_voff[i]);
            ar[i] = make_float4(wv.x, wv.y, wv.z, wv.w);
        }
    };
    auto fetch_b = [&](int p, float (&br)[KROWS][NCOL], int ci) {
        const bool first = ci < d.Cin0;
        const int64_t sc = first ? d.x0s[1] : d.x1s[1];
        const float* base = (first ? a.x0 + (int64_t)ci * sc : a.x1 + (int64_t)(ci - d.Cin0) * sc) + (int64_t)(wave * KROWS) * sc;
#pragma unroll
        for (int r = 0; r < KROWS; ++r) {
            const gfloat_p rb = uniform_ptr(base + r * sc);
#pragma unroll
            for (int t = 0; t < NCOL; ++t)
                if ((r * NCOL + t) % NSL == p) br[r][t] = ld_sv(rb, ((okm >> t) & 1u) ? (uint32_t)(first ? v0[t] : v1[t]) * 4u : 0u);
        }
    };
    auto park_a = [&](int buf, int p, const float4 (&ar)[NA4]) {
#pragma unroll
        for (int i = 0; i < NA4; ++i) {
            if (i % NSL != p) continue;
            const int f = tid + i * NTH;
            const int row = f / (BM / 4), c4 = (f % (BM / 4)) * 4;
            float4 v = ar[i];
            if (!a_ok[i]) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(As + buf * ASZ + row * BM + c4) = v;
        }
    };
    auto park_b = [&](int buf, int p, const float (&br)[KROWS][NCOL]) {
#pragma unroll
        for (int r = 0; r < KROWS; ++r)
#pragma unroll
            for (int t = 0; t < NCOL; ++t)
                if ((r * NCOL + t) % NSL == p)
                    Bs[buf * BSZ + (wave * KROWS + r) * PITCH + lane + 64 * t] = ((okm >> t) & 1u) ? br[r][t] : 0.0f;
    };
    auto next_ci = [&](int& c) { c += SK; if (c >= a.Cin) c = 0; };      // (past the last stage the walk wraps: in bounds, never consumed)
    auto transform = [&](int buf) {
        const float2* sp = reinterpret_cast<const float2*>(Bs + buf * BSZ + t_src);
        const float2 p0 = sp[0], p1 = sp[1], p2 = sp[2];
        const float d0 = p0.x, d1 = p0.y, d2 = p1.x, d3 = p1.y, d4 = p2.x, d5 = p2.y;
        const float e42 = fmaf(-4.0f, d2, d4), e31 = fmaf(-4.0f, d1, d3);
        const float c42 = d4 - d2, c31 = d3 - d1;
        float* dp = Vs + buf * VSZ + t_dst;
        dp[0 * NQ] = fmaf(4.0f, d0, fmaf(-5.0f, d2, d4));
        dp[1 * NQ] = e42 + e31;
        dp[2 * NQ] = e42 - e31;
        dp[3 * NQ] = fmaf(2.0f, c31, c42);
        dp[4 * NQ] = fmaf(-2.0f, c31, c42);
        dp[5 * NQ] = fmaf(4.0f, d1, fmaf(-5.0f, d3, d5));
    };

    f32x16 acc[NX];
#pragma unroll
    for (int x = 0; x < NX; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.0f;

    const int nstages = a.Cin / SK;
    {   // prologue: A(0) -> LDS, A(1) -> registers;  B(0) -> raw -> V(0), B(1) -> raw, B(2) -> registers
        float breg0[KROWS][NCOL], breg1[KROWS][NCOL];
        float4 areg0[NA4];
#pragma unroll
        for (int p = 0; p < NSL; ++p) { fetch_a(p, areg0, ca); fetch_b(p, breg0, cb); }
        next_ci(ca); next_ci(cb);
#pragma unroll
        for (int p = 0; p < NSL; ++p) { fetch_a(p, areg, ca); fetch_b(p, breg1, cb); }
        next_ci(ca); next_ci(cb);
#pragma unroll
        for (int p = 0; p < NSL; ++p) fetch_b(p, breg, cb);
        next_ci(cb);
#pragma unroll
        for (int p = 0; p < NSL; ++p) { park_a(0, p, areg0); park_b(0, p, breg0); park_b(1, p, breg1); }
        __syncthreads();
        transform(0);
    }
    __syncthreads();
    const int am = wm * 32 + l31;
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);

    for (int st = 0; st < nstages; ++st) {
        const int buf = st & 1;
        const float* Ab = As + buf * ASZ + (2 * KST * kh + lh) * BM + am;
        const float* Vb = Vs + buf * VSZ + (2 * KST * kh + lh) * VP + l31;
        float fa[2][NX], fv[2][NX];
        auto read_frag = [&](int ks, int set) {
#pragma unroll
            for (int x = 0; x < NX; ++x) {
                fa[set][x] = Ab[(x * SK + 2 * ks) * BM];
                fv[set][x] = Vb[(2 * ks) * VP + x * NQ];
            }
        };
        read_frag(0, 0);
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) {
            const int set = ks & 1;
            if (ks + 1 < KST) read_frag(ks + 1, set ^ 1);
            // staging piece ks: weights of stage st+1 and raw rows of stage st+2 to LDS, their registers re-used for the fetches
            // of st+2 / st+3; k-step 0 also turns this thread's item of raw(st+1) into V(st+1)
            park_a(buf ^ 1, ks, areg);
            park_b(buf, ks, breg);
            fetch_a(ks, areg, ca);
            fetch_b(ks, breg, cb);
            if (ks == 0) transform(buf ^ 1);
#pragma unroll
            for (int x = 0; x < NX; ++x) acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][x], fv[set][x], acc[x], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < NX; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (m < 2) __builtin_amdgcn_sched_group_barrier(0x100, NX, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x220, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        next_ci(ca); next_ci(cb);
        __syncthreads();
    }
    // the two k-halves of a tile meet in the staging LDS (dead after the last stage's barrier): [tile wave][register][lane]
    {
        float* red = ldsw + (size_t)wm * (NX * 16) * 64 + lane;
        if (kh == 1) {
#pragma unroll
            for (int x = 0; x < NX; ++x)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(x * 16 + r) * 64] = acc[x][r];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int x = 0; x < NX; ++x)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][r] += red[(x * 16 + r) * 64];
        }
        __syncthreads();
    }
    f32x16 acc3[NX][1][1];
#pragma unroll
    for (int x = 0; x < NX; ++x) acc3[x][0][0] = acc[x];
    wg_epilogue<1, 1, BM, BN, WM, 1, NX>(a, acc3, m0 + wm * 32, n0 / 4, lane, ldsw, wm, m0, n0, kh == 0);
}

template <int BM, int BN, int WM, int SK>
int launch_f43(const ConvArgs& a, hipStream_t s) {
    dim3 grid((a.Ntot + BN - 1) / BN, (a.d.Cout + BM - 1) / BM);
    constexpr int KSMAX = BN + (BN / 16) * 2, NCOL = (KSMAX + 63) / 64, PITCH = NCOL * 64 + 8, VP = 6 * (BN / 4) + 32;
    constexpr size_t lds = (2u * 6 * SK * BM + 2u * SK * PITCH + 2u * SK * VP) * sizeof(float);
    static_assert(lds <= 160u * 1024u && (size_t)WM * 6 * 16 * 64 * sizeof(float) <= lds, "LDS");
    static std::atomic<uint64_t> attr{0};
    SDC_LDS_OPTIN(attr, (conv_f43_kernel<BM, BN, WM, SK>), 160 * 1024, "sdc_conv[winograd F(4,3)]");
    hipLaunchKernelGGL((conv_f43_kernel<BM, BN, WM, SK>), grid, dim3(512), lds, s, a);
    return SDC_OK;
}

template <int BM, int BN, int WM, int WN, int SK, int NTH = 256, bool UPS = false, int NX = 4, int KS = 1>
int launch_wg(const ConvArgs& a, hipStream_t s) {
    // (single tap row, an even number of stages; the upsampling gather only where a split asks for it: 1-D nets at small batch)
    const bool p1 = (!UPS || a.ksplit > 1) && a.d.kD * a.d.kH == 1 && ((a.Cin / SK / a.ksplit) & 1) == 0;
    SDC_REQUIRE(a.ksplit == 1 || p1, SDC_EINVAL, "sdc_conv_splitk: the 1-D split needs the single-tap-row kernel");
    dim3 grid((a.Ntot + BN - 1) / BN, (a.d.Cout + BM - 1) / BM, a.ksplit);
    constexpr int KSMAX = BN + (BN / 16) * 2;
    constexpr int NCOLH = (KSMAX + 63) / 64;
    constexpr size_t STAGE = (2u * NX * SK * BM + 2u * SK * (NCOLH * 64 + 8)) * sizeof(float);
    // k-split: the second half's accumulators pass through the same LDS after the last stage
    constexpr size_t PART = KS == 1 ? 0 : (size_t)WM * WN * NX * (BM / WM / 32) * (BN / (NX == 6 ? 4 : 2) / WN / 32) * 16 * 64 * sizeof(float);
    constexpr size_t lds = STAGE > PART ? STAGE : PART;
    static_assert(lds <= 160u * 1024u, "stage buffers / k-split partials exceed the LDS");
    static std::atomic<uint64_t> attr{0}, attr1{0};
    if constexpr (!UPS || (BM == 64 && BN == 128 && NTH == 256)) {
        if (p1) {
            SDC_LDS_OPTIN(attr1, (conv_wg_kernel<BM, BN, WM, WN, SK, NTH, UPS, NX, KS, true>), 160 * 1024, "sdc_conv[winograd]");
            hipLaunchKernelGGL((conv_wg_kernel<BM, BN, WM, WN, SK, NTH, UPS, NX, KS, true>), grid, dim3(NTH), lds, s, a);
            return SDC_OK;
        }
    }
    (void)p1;
    SDC_LDS_OPTIN(attr, (conv_wg_kernel<BM, BN, WM, WN, SK, NTH, UPS, NX, KS>), 160 * 1024, "sdc_conv[winograd]");
    hipLaunchKernelGGL((conv_wg_kernel<BM, BN, WM, WN, SK, NTH, UPS, NX, KS>), grid, dim3(NTH), lds, s, a);
    return SDC_OK;
}

// what the dispatch chose for the last descriptor (sdc_conv_describe): kernel template instance and the share of the
// direct-form multiply-adds it issues on the matrix cores (Winograd forms issue fewer)
thread_local const char* tl_pick = "";
thread_local double tl_factor = 1.0;
thread_local bool tl_describe = false;
#define SDC_PICK(nm, f)                      \
    do {                                     \
        tl_pick = (nm);                      \
        tl_factor = (f);                     \
        if (tl_describe) return SDC_OK;      \
    } while (0)

int ilog2_pow2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

int ilog2_exact(int v) {
    if (v == 1) return 0;
    if (v == 2) return 1;
    if (v == 4) return 2;
    return -1;
}

template <int BM, int BN, int WM, int WN>
int launch(const ConvArgs& a, bool fast, hipStream_t s, const char* n_rh, const char* n_fast, const char* n_gen) {
    dim3 grid((a.Ntot + BN - 1) / BN, (a.d.Cout + BM - 1) / BM);
    const SdcConvDesc& d = a.d;
    // row-halo kernel: stride 1 along W, no virtual upsampling, kW == 3 (with kW == 1 there is no halo to share and
    // the plain kernel measured faster), whole rows or whole row segments per tile
    const bool rh = fast && a.rowhalo && d.sW == 1 && d.uD == 1 && d.uH == 1 && d.uW == 1 && d.up_mode == 0 &&
                    d.kW == 3 && d.kD * d.kH <= 32 && d.Cout % 4 == 0 &&
                    ((d.oW % BN == 0) || (BN % d.oW == 0 && d.oW >= 16)) &&
                    (reinterpret_cast<uintptr_t>(a.wp) % 16 == 0);
    if constexpr (BM >= 64) {
        if (rh) { SDC_PICK(n_rh, 1.0); hipLaunchKernelGGL((conv_rh_kernel<BM, BN, WM, WN, 3, false>), grid, dim3(NT), 0, s, a); return SDC_OK; }
    }
    if (fast) {
        SDC_PICK(n_fast, 1.0);
        hipLaunchKernelGGL((conv_kernel<BM, BN, WM, WN, true>), grid, dim3(NT), 0, s, a);
    } else {
        SDC_PICK(n_gen, 1.0);
        hipLaunchKernelGGL((conv_kernel<BM, BN, WM, WN, false>), grid, dim3(NT), 0, s, a);
    }
    return SDC_OK;
}
#define SDC_LAUNCH(BM, BN, WM, WN)                                                                      \
    launch<BM, BN, WM, WN>(a, fast, s, "conv_rh_kernel<" #BM "," #BN "," #WM "," #WN ",3,false>",        \
                           "conv_kernel<" #BM "," #BN "," #WM "," #WN ",true>", "conv_kernel<" #BM "," #BN "," #WM "," #WN ",false>")

// Winograd (precision 2) coverage and tile choice, shared by the dispatch and by sdc_conv_gnparts.
// 8-wave workgroups (two waves per SIMD) share one staged weight tile; the bigger the tile the fewer L2->LDS bytes
// per MFMA: 128 x 256 / 128 x 128 outputs for wide layers, 64 x 512 / 64 x 256 for Cout <= 64, 4-wave 64 x 128 for
// small grids.  pick: 3 = 64x128, 6 = 128x128, 7 = 64x256, 9 = 128x256, 10 = 64x512; 0 = not covered.
struct WgPick { int pick, bm, bn; bool ups; };
WgPick wg_pick(const SdcConvDesc& d, int64_t ntot, bool small, bool rowhalo) {
    WgPick w{0, 0, 0, false};
    w.ups = (d.uH > 1 || d.uW > 1);      // nearest x2 upsampling folded into the gather: one input, kD = 1, kH <= 3
    if (!(d.precision >= 2 && rowhalo && d.kW == 3 && d.sW == 1 && d.uD == 1 && d.up_mode == 0 &&
          (!w.ups || (d.uH <= 2 && d.uW <= 2 && d.kD == 1 && d.kH <= 3 && d.sH == 1 && d.Cin1 == 0)) &&
          d.kD * d.kH <= 32 && d.Cin0 % 16 == 0 && d.Cin1 % 16 == 0 && small && d.Cout % 4 == 0 && d.Cout > 32 &&
          d.oW % 2 == 0 && d.oW >= 16 && ((int64_t)d.kD * d.kH * d.kW * (d.Cin0 + d.Cin1) * d.Cout) % 4 == 0))
        return w;
    auto fits = [&](int bn) { return (d.oW % bn == 0) || (bn % d.oW == 0); };
    if (!fits(128)) return w;
    auto nblk = [&](int bm, int bn) { return ((ntot + bn - 1) / bn) * ((d.Cout + bm - 1) / bm); };
    static const int wg_tile = exp_env("SDC_WG_TILE");   // tuning knob
    int pick = wg_tile;
    if (w.ups) {
        pick = (d.Cout > 64 && nblk(128, 128) >= 256) ? 6 : ((d.Cout <= 64 && fits(256) && nblk(64, 256) >= 256) ? 7 : 3);
    } else {
        if (!pick) {
            if (d.Cout > 64) pick = (fits(256) && nblk(128, 256) >= 256) ? 9 : (nblk(128, 128) >= 256 ? 6 : 3);
            else pick = (fits(512) && nblk(64, 512) >= 512) ? 10 : (fits(256) && nblk(64, 256) >= 256 ? 7 : 3);
        }
        if ((pick == 7 || pick == 9) && !fits(256)) pick = 3;
        if (pick == 10 && !fits(512)) pick = 3;
        if (pick != 6 && pick != 7 && pick != 9 && pick != 10) pick = 3;
    }
    // precision 5 (opt-in): 1-D convs (kD = kH = 1, one row per sample and channel) whose rows are whole quads take F(4,3), half of
    // the direct form's MFMAs instead of two thirds; 128 x 128 outputs per workgroup (the C3 layers are ~256 such tiles each).
    // NOT the default: measured at C3 the six-product kernel issues 74 TFLOP/s where the F(2,3) one
    // issues 97 -- twelve transform operations per six MFMAs are not hidden behind them -- 6.09 ms against 5.99 ms for the 33
    // layers, at three times the rounding error
    static const int no_f43 = exp_env("SDC_NO_F43");
    if (!no_f43 && d.precision == 5 && !w.ups && d.kD == 1 && d.kH == 1 && d.oD == 1 && d.oH == 1 && d.oW % 4 == 0 && d.Cout >= 128) pick = 13;
    w.pick = pick;
    w.bm = (pick == 6 || pick == 9 || pick == 13) ? 128 : 64;
    w.bn = (pick == 3 || pick == 6 || pick == 13) ? 128 : (pick == 10 ? 512 : 256);
    return w;
}

// GroupNorm partial sums in the Winograd epilogue: pairs per (sample, group), 0 if the tile grid does not line up
int gn_parts_for(const SdcConvDesc& d, const WgPick& w, int G) {
    if (!w.pick || G <= 0 || d.Cout % G) return 0;
    const int cpg = d.Cout / G;
    const int64_t S = (int64_t)d.oD * d.oH * d.oW;
    if (cpg % 8 || S % w.bn || !(cpg % w.bm == 0 || w.bm % cpg == 0)) return 0;
    return (int)(S / w.bn) * (cpg >= w.bm ? cpg / w.bm : 1);
}

// Cin split of sdc_conv_splitk on the 1-D F(2,3) form (conv_wg_kernel<64,128,...,ks2> on its single-tap-row path): as wg2_ksplit --
// only where the plain launch leaves more than half of the 256 CUs without a workgroup; every split keeps >= 4 stages and an even
// number of whole 16-channel stages.  Depends on the batch (the tile count does).
int wg1_ksplit(const SdcConvDesc& d, const WgPick& w, int64_t ntot) {
    if (w.pick != 3 || d.kD * d.kH != 1 || d.precision == 5 || (w.ups && (d.uH != 1 || d.iH != 1))) return 1;
    const int64_t nb = ((ntot + 127) / 128) * ((d.Cout + 63) / 64);
    if (nb >= 256) return 1;                      // (launch_wg's 128 x 128 / 64 x 256 tiles take the bigger grids)
    const int cin = d.Cin0 + d.Cin1;
    if ((cin / 16) & 1) return 1;
    int S = 1;
    while (S < 8 && nb * S * 2 <= 256 && cin % (32 * S * 2) == 0 && cin / (16 * S * 2) >= 4) S *= 2;
    return S;
}

// Cin / tap split of sdc_conv_splitk on the direct-form kernels' smallest tile (conv_pw_kernel<64,64,2,2>, conv_kernel<64,64,2,2,FAST>):
// 1x1 convs and the sub-pixel / strided / unshuffle convs of the 1-D and 2-D nets at small batch, where a layer is <= 128 workgroups
// walking K = 2048-8192; every split keeps >= 8 chunks of 16.  (3-tap convs have their own forms above.)  Depends on the batch.
int direct_ksplit(const SdcConvDesc& d, int64_t ntot, bool fast) {
    if (!fast || d.Cout <= 32 || d.kW == 3) return 1;
    const int64_t nb = ((ntot + 63) / 64) * ((d.Cout + 63) / 64);
    const int chunks = d.kD * d.kH * d.kW * ((d.Cin0 + d.Cin1) / 16);
    int S = 1;
    while (S < 8 && nb * S * 2 <= 256 && chunks % (S * 2) == 0 && chunks / (S * 2) >= 8) S *= 2;
    return S;
}

bool conv_small(const SdcConvDesc& d) {
    auto span = [](const int64_t* st, int b, int dd, int h, int w) {
        return (int64_t)(b - 1) * st[0] + (int64_t)(dd - 1) * st[2] + (int64_t)(h - 1) * st[3] + (int64_t)(w - 1) * st[4];
    };
    return span(d.x0s, d.B, d.iD, d.iH, d.iW) < (1ll << 30) && (d.Cin1 == 0 || span(d.x1s, d.B, d.iD, d.iH, d.iW) < (1ll << 30));
}

// y (strided) = sum over the splits, in split order, of the dense partial copies of sdc_conv_splitk
__global__ __launch_bounds__(256) void splitk_sum_kernel(const float* __restrict__ part, float* __restrict__ y, int S, int64_t elems, int C,
                                                         int oD, int oH, int oW, int64_t s0, int64_t s1, int64_t s2, int64_t s3, int64_t s4,
                                                         const float* __restrict__ bias = nullptr) {
    typedef float nf4 __attribute__((ext_vector_type(4)));
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q * 4 >= elems) return;
    nf4 acc = *reinterpret_cast<const nf4*>(part + q * 4);
    for (int k = 1; k < S; ++k) acc += *reinterpret_cast<const nf4*>(part + (int64_t)k * elems + q * 4);
    int64_t r = q * 4;
    const int w = (int)(r % oW); r /= oW;
    const int h = (int)(r % oH); r /= oH;
    const int dd = (int)(r % oD); r /= oD;
    const int c = (int)(r % C);
    const int64_t b = r / C;
    if (bias) acc += bias[c];                    // (the 1-D form: its splits carry no bias)
    float* o = y + b * s0 + c * s1 + dd * s2 + h * s3 + w * s4;
    o[0] = acc.x; o[s4] = acc.y; o[2 * s4] = acc.z; o[3 * s4] = acc.w;
}

int conv_impl(const SdcConvDesc* dp, const float* x0, const float* x1, const float* wp, const float* bias,
              const float* residual, float* y, double* gn_part, int gn_G, void* stream, float* split_work = nullptr, size_t split_bytes = 0);

}  // namespace

extern "C" int sdc_conv(const SdcConvDesc* dp, const float* x0, const float* x1, const float* wp, const float* bias,
                        const float* residual, float* y, void* stream) {
    return conv_impl(dp, x0, x1, wp, bias, residual, y, nullptr, 0, stream);
}

// Same conv as sdc_conv (no fused residual), for grids that leave most CUs idle -- the fine-tuning step at batch 64 runs the deep
// 3x3 convs of the Burgers net on 64 workgroups: the input channels are split over up to 8 workgroups per output tile, the
// partial outputs go to `work` and are summed in split order (deterministic).  Other shapes: exactly sdc_conv.
extern "C" size_t sdc_conv_splitk_bytes(const SdcConvDesc* dp) {
    if (!dp) return 0;                               // (every form below checks the precision it needs itself; the direct forms need none)
    static const int no_rh = exp_env("SDC_NO_ROWHALO");
    const SdcConvDesc& d = *dp;
    if (wg3s_ok(d, conv_small(d), !no_rh) || wg3_ok(d, conv_small(d), !no_rh)) return 0;
    int S;
    if (wg2_ok(d, conv_small(d), !no_rh)) S = wg2_ksplit(d);
    else {
        const int64_t ntot = (int64_t)d.B * d.oD * d.oH * d.oW;
        const WgPick wgp = wg_pick(d, ntot, conv_small(d), !no_rh);
        if (wgp.pick) S = (d.oW % 4 == 0 && d.ys[4] == 1) ? wg1_ksplit(d, wgp, ntot) : 1;
        else S = (d.oW % 4 == 0 && d.precision != 5)
                     ? direct_ksplit(d, ntot, (d.Cin0 % 16 == 0) && (d.Cin1 % 16 == 0) && conv_small(d)) : 1;
    }
    return S > 1 ? (size_t)S * d.B * d.Cout * d.oD * d.oH * d.oW * sizeof(float) : 0;
}

extern "C" int sdc_conv_splitk(const SdcConvDesc* dp, const float* x0, const float* x1, const float* wp, const float* bias, float* y,
                               float* work, size_t work_bytes, void* stream) {
    return conv_impl(dp, x0, x1, wp, bias, nullptr, y, nullptr, 0, stream, work, work_bytes);
}

extern "C" int sdc_conv_gnparts(const SdcConvDesc* dp, int G) {
    if (!dp) return 0;
    static const int no_rh = exp_env("SDC_NO_ROWHALO");
    const int64_t ntot = (int64_t)dp->B * dp->oD * dp->oH * dp->oW;
    static const int no_wg2 = exp_env("SDC_NO_WG2");
    static const int no_wg3 = exp_env("SDC_NO_WG3");
    static const int old_wg3 = exp_env("SDC_WG3_OLD");
    if (!no_wg2 && !no_wg3 && !old_wg3 && wg3s_ok(*dp, conv_small(*dp), !no_rh)) return gn_parts_for(*dp, WgPick{22, W2_BM, W3S_TILES * 8, false}, G);
    if (!no_wg2 && !no_wg3 && wg3_ok(*dp, conv_small(*dp), !no_rh)) return gn_parts_for(*dp, WgPick{21, W2_BM, W2_TILES * 8, false}, G);
    static const int no_wg2s = exp_env("SDC_NO_WG2S");
    if (!no_wg2 && !no_wg2s && wg2s_ok(*dp, conv_small(*dp), !no_rh)) return gn_parts_for(*dp, WgPick{23, W2_BM, W3S_TILES * 4, false}, G);
    if (!no_wg2 && wg2_ok(*dp, conv_small(*dp), !no_rh)) return gn_parts_for(*dp, WgPick{20, W2_BM, W2_TILES * 4, false}, G);
    return gn_parts_for(*dp, wg_pick(*dp, ntot, conv_small(*dp), !no_rh), G);
}

extern "C" int sdc_conv_gn(const SdcConvDesc* dp, const float* x0, const float* x1, const float* wp, const float* bias,
                           const float* residual, float* y, double* parts, int G, void* stream) {
    SDC_REQUIRE(parts && G > 0, SDC_EINVAL, "sdc_conv_gn: parts buffer and a positive group count are required");
    return conv_impl(dp, x0, x1, wp, bias, residual, y, parts, G, stream);
}

extern "C" int sdc_conv_describe(const SdcConvDesc* dp, char* name, size_t cap, double* mfma_share) {
    SDC_REQUIRE(dp, SDC_ENULL, "sdc_conv_describe: null descriptor");
    float* const dummy = reinterpret_cast<float*>(uintptr_t(256));   // aligned, never dereferenced: nothing is launched
    tl_describe = true;
    tl_pick = "";
    tl_factor = 1.0;
    const int rc = conv_impl(dp, dummy, dp->Cin1 > 0 ? dummy : nullptr, dummy, nullptr, nullptr, dummy, nullptr, 0, nullptr);
    tl_describe = false;
    if (rc != SDC_OK) return rc;
    if (name && cap) { std::strncpy(name, tl_pick, cap - 1); name[cap - 1] = 0; }
    if (mfma_share) *mfma_share = tl_factor;
    return SDC_OK;
}

namespace {

int conv_impl(const SdcConvDesc* dp, const float* x0, const float* x1, const float* wp, const float* bias,
              const float* residual, float* y, double* gn_part, int gn_G, void* stream, float* split_work, size_t split_bytes) {
    SDC_REQUIRE(dp && x0 && wp && y, SDC_ENULL, "sdc_conv: null pointer");
    const SdcConvDesc& d = *dp;
    SDC_REQUIRE(d.B > 0 && d.Cin0 > 0 && d.Cin1 >= 0 && d.Cout > 0, SDC_EINVAL, "sdc_conv: bad channel/batch counts");
    SDC_REQUIRE(d.Cin1 == 0 || x1, SDC_ENULL, "sdc_conv: Cin1 > 0 but x1 is null");
    SDC_REQUIRE(d.kD > 0 && d.kH > 0 && d.kW > 0 && d.sD > 0 && d.sH > 0 && d.sW > 0, SDC_EINVAL,
                "sdc_conv: bad kernel/stride");
    SDC_REQUIRE(d.precision == 0 || (d.precision >= 2 && d.precision <= 5), SDC_EINVAL, "sdc_conv: precision must be 0 (fp32 MFMA, direct form), 2 (fp32 Winograd along W), 3 (fp32 Winograd over H and W), 4 (fp32 Winograd over D, H and W) or 5 (as 4, F(4,3) for the 1-D convs)");
    SDC_REQUIRE(!gn_part || d.precision >= 2, SDC_EINVAL, "sdc_conv_gn: fused GroupNorm statistics need precision 2, 3 or 4 (sdc_conv_gnparts returned 0)");
    // the caller sized `parts` with sdc_conv_gnparts(d, G), which sees the descriptor only: a kernel picked here on other
    // grounds (pointer alignment, a residual) with a different part count would write a table the finalize pass misreads
    const int gn_expect = gn_part ? sdc_conv_gnparts(dp, gn_G) : 0;
#define SDC_GN_PARTS_AGREE(n) SDC_REQUIRE((n) == gn_expect, SDC_EINVAL, "sdc_conv_gn: this call runs a kernel with %d partial sums per group where sdc_conv_gnparts promised %d (misaligned pointers or a residual the descriptor does not show?)", (n), gn_expect)
    ConvArgs a;
    a.d = d;
    a.lgD = ilog2_exact(d.uD); a.lgH = ilog2_exact(d.uH); a.lgW = ilog2_exact(d.uW);
    SDC_REQUIRE(a.lgD >= 0 && a.lgH >= 0 && a.lgW >= 0, SDC_EINVAL, "sdc_conv: upsample factors must be 1, 2 or 4");
    // output size must agree with what the gather will produce
    auto osz = [](int i, int u, int mode, int k, int s, int p) {
        const int v = mode ? (i - 1) * u + 1 : i * u;
        return (v + 2 * p - k) / s + 1;
    };
    // (up to k-1 extra positions per axis are allowed: they read the implicit zeros past the far edge -- one-sided padding,
    // used by the sub-pixel form of the stride-2 transposed conv; a shorter axis computes a prefix)
    auto fits = [&](int o, int i, int u, int k, int st, int p) { return o >= 1 && o <= osz(i, u, d.up_mode, k, st, p) + (k - 1); };
    SDC_REQUIRE(fits(d.oD, d.iD, d.uD, d.kD, d.sD, d.pD) && fits(d.oH, d.iH, d.uH, d.kH, d.sH, d.pH) && fits(d.oW, d.iW, d.uW, d.kW, d.sW, d.pW),
                SDC_EINVAL, "sdc_conv: output size (%d,%d,%d) inconsistent with input/kernel/stride/pad", d.oD, d.oH, d.oW);
    const int64_t ntot = (int64_t)d.B * d.oD * d.oH * d.oW;
    SDC_REQUIRE(ntot < (1ll << 31), SDC_EINVAL, "sdc_conv: too many output positions");
    a.x0 = x0; a.x1 = x1; a.wp = wp; a.bias = bias; a.res = residual; a.y = y;
    a.Ntot = (int)ntot;
    a.Cin = d.Cin0 + d.Cin1;
    a.Ktot = d.kD * d.kH * d.kW * a.Cin;
    // FAST: whole K chunks share a tap, and every per-thread offset fits the 32-bit voffset of the saddr load form
    const bool small = conv_small(d);
    static const int no_rh = exp_env("SDC_NO_ROWHALO");
    a.rowhalo = !no_rh;
    a.vec2 = 0;
    {
        auto dense = [&](const int64_t* st) { return st[4] == 1 && st[3] == d.oW && st[2] == (int64_t)d.oH * d.oW; };
        const int64_t S = (int64_t)d.oD * d.oH * d.oW;
        a.ydense = dense(d.ys) && S >= 128 && S < (1 << 24) && span5(d.ys, d.B, d.Cout, d.oD, d.oH, d.oW) < (1ll << 30) &&
                   (!residual || (dense(d.rs) && span5(d.rs, d.B, d.Cout, d.oD, d.oH, d.oW) < (1ll << 30)));
        static const int no_dense = exp_env("SDC_NO_DENSE_EPI");
        if (no_dense) a.ydense = 0;
    }
    a.wg2 = nullptr;
    a.gn_part = nullptr; a.gn_G = a.gn_cpg = a.gn_nparts = a.gn_S = 0;
    a.ksplit = 1; a.ypart_elems = 0;
    const bool fast = (d.Cin0 % BK == 0) && (d.Cin1 % BK == 0) && small && d.Cout < (1 << 30);
    hipStream_t s = sdc::as_stream(stream);
    auto even = [](const int64_t* st) { return st[4] == 1 && st[0] % 2 == 0 && st[1] % 2 == 0 && st[2] % 2 == 0 && st[3] % 2 == 0; };
    static const int no_wg2 = exp_env("SDC_NO_WG2");
    static const int no_wg3 = exp_env("SDC_NO_WG3");
    // fp32 Winograd F(2x2x2,3x3x3), two workgroups per CU (round 5): 3x3x3 stride-1 convs over whole rows, plane pairs
    static const int old_wg3 = exp_env("SDC_WG3_OLD");
    if (!no_wg2 && !no_wg3 && !old_wg3 && wg3s_ok(d, small, a.rowhalo != 0) && reinterpret_cast<uintptr_t>(wp) % 16 == 0 &&
        reinterpret_cast<uintptr_t>(x0) % 16 == 0 && (d.Cin1 == 0 || reinterpret_cast<uintptr_t>(x1) % 16 == 0) &&
        reinterpret_cast<uintptr_t>(y) % 8 == 0 && !residual) {
        a.vec2 = 1;
        a.wg2 = wp + (int64_t)a.Ktot * d.Cout + (int64_t)(a.Ktot / 3 * 4) * d.Cout + (int64_t)(a.Ktot / 9 * 16) * d.Cout;
        if (gn_part) {
            a.gn_nparts = gn_parts_for(d, WgPick{22, W2_BM, W3S_TILES * 8, false}, gn_G);
            SDC_REQUIRE(a.gn_nparts > 0, SDC_EINVAL, "sdc_conv_gn: shape not covered by the fused statistics (sdc_conv_gnparts returned 0)");
            SDC_GN_PARTS_AGREE(a.gn_nparts);
            a.gn_part = gn_part; a.gn_G = gn_G; a.gn_cpg = d.Cout / gn_G; a.gn_S = d.oD * d.oH * d.oW;
        }
        SDC_PICK(d.oW == 16 ? "conv_wg3s_kernel<16>" : (d.oW == 32 ? "conv_wg3s_kernel<32>" : "conv_wg3s_kernel<64>"), 8.0 / 27.0);
        { const int rc_ = launch_wg3s(a, s); if (rc_) return rc_; }
        return sdc::check_launch("sdc_conv[winograd 2x2x2, two workgroups per CU]");
    }
    // fp32 Winograd F(2x2x2,3x3x3), one workgroup per CU: the shapes the form above does not take
    if (!no_wg2 && !no_wg3 && wg3_ok(d, small, a.rowhalo != 0) && reinterpret_cast<uintptr_t>(wp) % 16 == 0 &&
        reinterpret_cast<uintptr_t>(x0) % 16 == 0 && (d.Cin1 == 0 || reinterpret_cast<uintptr_t>(x1) % 16 == 0) &&
        reinterpret_cast<uintptr_t>(y) % 8 == 0 && !residual) {
        a.vec2 = 1;
        a.wg2 = wp + (int64_t)a.Ktot * d.Cout + (int64_t)(a.Ktot / 3 * 4) * d.Cout + (int64_t)(a.Ktot / 9 * 16) * d.Cout;
        if (gn_part) {
            a.gn_nparts = gn_parts_for(d, WgPick{21, W2_BM, W2_TILES * 8, false}, gn_G);
            SDC_REQUIRE(a.gn_nparts > 0, SDC_EINVAL, "sdc_conv_gn: shape not covered by the fused statistics (sdc_conv_gnparts returned 0)");
            SDC_GN_PARTS_AGREE(a.gn_nparts);
            a.gn_part = gn_part; a.gn_G = gn_G; a.gn_cpg = d.Cout / gn_G; a.gn_S = d.oD * d.oH * d.oW;
        }
        SDC_PICK(d.oW == 16 ? "conv_wg3_kernel<16>" : (d.oW == 32 ? "conv_wg3_kernel<32>" : "conv_wg3_kernel<64>"), 8.0 / 27.0);
        { const int rc_ = launch_wg3(a, s); if (rc_) return rc_; }
        return sdc::check_launch("sdc_conv[winograd 2x2x2]");
    }
    // fp32 Winograd F(2x2,3x3), two workgroups per CU (round 6): 3x3 convs (kD = 1) over whole rows of 128 / 64 / 32 -- unless
    // sdc_conv_splitk wants to split this conv (small grids keep the one-workgroup form and its Cin split)
    static const int no_wg2s = exp_env("SDC_NO_WG2S");
    if (!no_wg2 && !no_wg2s && wg2s_ok(d, small, a.rowhalo != 0) && reinterpret_cast<uintptr_t>(wp) % 16 == 0 &&
        reinterpret_cast<uintptr_t>(x0) % 16 == 0 && (d.Cin1 == 0 || reinterpret_cast<uintptr_t>(x1) % 16 == 0) &&
        reinterpret_cast<uintptr_t>(y) % 8 == 0 && !residual && !(split_work && wg2_ksplit(d) > 1)) {
        a.vec2 = 1;
        a.wg2 = wp + (int64_t)a.Ktot * d.Cout + (int64_t)(a.Ktot / 3 * 4) * d.Cout;
        if (gn_part) {
            a.gn_nparts = gn_parts_for(d, WgPick{23, W2_BM, W3S_TILES * 4, false}, gn_G);
            SDC_REQUIRE(a.gn_nparts > 0, SDC_EINVAL, "sdc_conv_gn: shape not covered by the fused statistics (sdc_conv_gnparts returned 0)");
            SDC_GN_PARTS_AGREE(a.gn_nparts);
            a.gn_part = gn_part; a.gn_G = gn_G; a.gn_cpg = d.Cout / gn_G; a.gn_S = d.oD * d.oH * d.oW;
        }
        SDC_PICK(d.oW == 128 ? "conv_wg2s_kernel<128>" : (d.oW == 64 ? "conv_wg2s_kernel<64>" : "conv_wg2s_kernel<32>"), 4.0 / 9.0);
        { const int rc_ = launch_wg2s(a, s); if (rc_) return rc_; }
        return sdc::check_launch("sdc_conv[winograd 2x2, two workgroups per CU]");
    }
    // fp32 Winograd F(2x2,3x3) over (H, W): 3x3 / 3x3x3 stride-1 convs over whole rows
    if (!no_wg2 && wg2_ok(d, small, a.rowhalo != 0) && reinterpret_cast<uintptr_t>(wp) % 16 == 0 &&
        reinterpret_cast<uintptr_t>(x0) % 16 == 0 && (d.Cin1 == 0 || reinterpret_cast<uintptr_t>(x1) % 16 == 0)) {
        a.vec2 = even(d.ys) && reinterpret_cast<uintptr_t>(y) % 8 == 0 &&
                 (!residual || (even(d.rs) && reinterpret_cast<uintptr_t>(residual) % 8 == 0));
        a.wg2 = wp + (int64_t)a.Ktot * d.Cout + (int64_t)(a.Ktot / 3 * 4) * d.Cout;
        a.lgW = ilog2_pow2(d.oW);
        if (gn_part) {
            a.gn_nparts = gn_parts_for(d, WgPick{20, W2_BM, W2_TILES * 4, false}, gn_G);
            SDC_REQUIRE(a.gn_nparts > 0, SDC_EINVAL, "sdc_conv_gn: shape not covered by the fused statistics (sdc_conv_gnparts returned 0)");
            SDC_GN_PARTS_AGREE(a.gn_nparts);
            a.gn_part = gn_part; a.gn_G = gn_G; a.gn_cpg = d.Cout / gn_G; a.gn_S = d.oD * d.oH * d.oW;
        }
        SDC_PICK(d.oW == 16 ? "conv_wg2_kernel<16>" : (d.oW == 32 ? "conv_wg2_kernel<32>" : (d.oW == 64 ? "conv_wg2_kernel<64>" : "conv_wg2_kernel<128>")),
                 4.0 / 9.0);
        // sdc_conv_splitk: Cin split over ksplit workgroups per tile into the caller's partial buffer, summed in split order
        const int S = (split_work && !gn_part && !residual && !tl_describe) ? wg2_ksplit(d) : 1;
        if (S > 1) {
            const int64_t elems = (int64_t)d.B * d.Cout * d.oD * d.oH * d.oW;
            SDC_REQUIRE(split_bytes >= (size_t)S * elems * sizeof(float), SDC_EINVAL, "sdc_conv_splitk: workspace too small");
            SDC_REQUIRE(reinterpret_cast<uintptr_t>(split_work) % 16 == 0, SDC_EINVAL, "sdc_conv_splitk: workspace must be 16-byte aligned");
            ConvArgs p = a;
            p.ksplit = S; p.ypart_elems = elems; p.y = split_work; p.vec2 = 1;
            p.d.ys[4] = 1; p.d.ys[3] = d.oW; p.d.ys[2] = (int64_t)d.oH * d.oW; p.d.ys[1] = p.d.ys[2] * d.oD; p.d.ys[0] = p.d.ys[1] * d.Cout;
            SDC_REQUIRE(span5(p.d.ys, d.B, 8, d.oD, d.oH, d.oW) < (1ll << 30), SDC_EINVAL, "sdc_conv_splitk: partial copy too large");
            { const int rc_ = launch_wg2(p, s); if (rc_) return rc_; }
            const int64_t nq = elems / 4;                     // oW is a multiple of 16 here
            hipLaunchKernelGGL(splitk_sum_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, (const float*)split_work, y, S, elems,
                               d.Cout, d.oD, d.oH, d.oW, d.ys[0], d.ys[1], d.ys[2], d.ys[3], d.ys[4]);
            return sdc::check_launch("sdc_conv_splitk[winograd 2x2]");
        }
        { const int rc_ = launch_wg2(a, s); if (rc_) return rc_; }
        return sdc::check_launch("sdc_conv[winograd 2x2]");
    }
    // fp32 Winograd F(2,3) along W: 3-wide stride-1 taps, whole 16-channel chunks, even rows
    const WgPick wgp = wg_pick(d, ntot, small, a.rowhalo != 0);
    if (wgp.pick && reinterpret_cast<uintptr_t>(wp) % 16 == 0) {
        a.vec2 = even(d.ys) && reinterpret_cast<uintptr_t>(y) % 8 == 0 &&
                 (!residual || (even(d.rs) && reinterpret_cast<uintptr_t>(residual) % 8 == 0));
        if (gn_part) {
            a.gn_nparts = gn_parts_for(d, wgp, gn_G);
            SDC_REQUIRE(a.gn_nparts > 0, SDC_EINVAL, "sdc_conv_gn: shape not covered by the fused statistics (sdc_conv_gnparts returned 0)");
            SDC_GN_PARTS_AGREE(a.gn_nparts);
            a.gn_part = gn_part; a.gn_G = gn_G; a.gn_cpg = d.Cout / gn_G; a.gn_S = d.oD * d.oH * d.oW;
        }
        // sdc_conv_splitk on the 1-D form: Cin split over S workgroups per tile into the caller's partial buffer, summed in split
        // order (+ bias) by splitk_sum_kernel
        const int S1d = (split_work && !gn_part && !residual && !tl_describe && d.oW % 4 == 0 && d.ys[4] == 1) ? wg1_ksplit(d, wgp, ntot) : 1;
        if (S1d > 1) {
            const int64_t elems = (int64_t)d.B * d.Cout * d.oD * d.oH * d.oW;
            SDC_REQUIRE(split_bytes >= (size_t)S1d * elems * sizeof(float), SDC_EINVAL, "sdc_conv_splitk: workspace too small");
            SDC_REQUIRE(reinterpret_cast<uintptr_t>(split_work) % 16 == 0, SDC_EINVAL, "sdc_conv_splitk: workspace must be 16-byte aligned");
            ConvArgs p = a;
            p.ksplit = S1d; p.ypart_elems = elems; p.y = split_work; p.vec2 = 1; p.bias = nullptr;
            p.d.ys[4] = 1; p.d.ys[3] = d.oW; p.d.ys[2] = (int64_t)d.oH * d.oW; p.d.ys[1] = p.d.ys[2] * d.oD; p.d.ys[0] = p.d.ys[1] * d.Cout;
            if (wgp.ups) {
                SDC_PICK("conv_wg_kernel<64,128,2,2,16,256,ups>", 2.0 / 3.0);
                { const int rc_ = launch_wg<64, 128, 2, 2, 16, 256, true>(p, s); if (rc_) return rc_; }
            } else {
                SDC_PICK("conv_wg_kernel<64,128,2,2,16,512,ks2>", 2.0 / 3.0);
                { const int rc_ = launch_wg<64, 128, 2, 2, 16, 512, false, 4, 2>(p, s); if (rc_) return rc_; }
            }
            const int64_t nq = elems / 4;
            hipLaunchKernelGGL(splitk_sum_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, (const float*)split_work, y, S1d, elems,
                               d.Cout, d.oD, d.oH, d.oW, d.ys[0], d.ys[1], d.ys[2], d.ys[3], d.ys[4], bias);
            return sdc::check_launch("sdc_conv_splitk[winograd 1-D]");
        }
        if (wgp.ups) {
            if (wgp.pick == 6) { SDC_PICK("conv_wg_kernel<128,128,4,2,16,512,ups>", 2.0 / 3.0); { const int rc_ = launch_wg<128, 128, 4, 2, 16, 512, true>(a, s); if (rc_) return rc_; } }
            else if (wgp.pick == 7) { SDC_PICK("conv_wg_kernel<64,256,2,4,16,512,ups>", 2.0 / 3.0); { const int rc_ = launch_wg<64, 256, 2, 4, 16, 512, true>(a, s); if (rc_) return rc_; } }
            else { SDC_PICK("conv_wg_kernel<64,128,2,2,16,256,ups>", 2.0 / 3.0); { const int rc_ = launch_wg<64, 128, 2, 2, 16, 256, true>(a, s); if (rc_) return rc_; } }
            return sdc::check_launch("sdc_conv[winograd,upsample]");
        }
        if (wgp.pick == 13) { SDC_PICK("conv_f43_kernel<128,128,4,16,F43>", 0.5); { const int rc_ = launch_f43<128, 128, 4, 16>(a, s); if (rc_) return rc_; } }
        else if (wgp.pick == 6) { SDC_PICK("conv_wg_kernel<128,128,4,2,16,512>", 2.0 / 3.0); { const int rc_ = launch_wg<128, 128, 4, 2, 16, 512>(a, s); if (rc_) return rc_; } }
        else if (wgp.pick == 7) { SDC_PICK("conv_wg_kernel<64,256,2,4,16,512>", 2.0 / 3.0); { const int rc_ = launch_wg<64, 256, 2, 4, 16, 512>(a, s); if (rc_) return rc_; } }
        else if (wgp.pick == 9) { SDC_PICK("conv_wg_kernel<128,256,4,2,16,512>", 2.0 / 3.0); { const int rc_ = launch_wg<128, 256, 4, 2, 16, 512>(a, s); if (rc_) return rc_; } }       // (2 x 4 waves measured the same)
        else if (wgp.pick == 10) { SDC_PICK("conv_wg_kernel<64,512,1,8,16,512>", 2.0 / 3.0); { const int rc_ = launch_wg<64, 512, 1, 8, 16, 512>(a, s); if (rc_) return rc_; } }     // (each wave: both 32-row tiles x 32 pairs; measured 3 % ahead of 1 x 2)
        else {
            // small grids (fewer than 256 of the 128 x 128 tiles): 64 x 128 tiles, each stage's k-steps split over two waves per SIMD
            // (a lone wave per SIMD left the transform / staging VALU exposed: 68 TFLOP/s issued)
            SDC_PICK("conv_wg_kernel<64,128,2,2,16,512,ks2>", 2.0 / 3.0);
            { const int rc_ = launch_wg<64, 128, 2, 2, 16, 512, false, 4, 2>(a, s); if (rc_) return rc_; }
        }
        return sdc::check_launch("sdc_conv[winograd]");
    }
    SDC_REQUIRE(!gn_part, SDC_EINVAL, "sdc_conv_gn: shape not covered by the fused statistics (sdc_conv_gnparts returned 0)");
    // pointwise convs over a dense layout: 16-byte loads of weights and activations
    static const int no_pw = exp_env("SDC_NO_PW");
    {
        const int64_t S = (int64_t)d.oD * d.oH * d.oW;
        auto dense = [&](const int64_t* st) { return st[4] == 1 && st[3] == d.iW && st[2] == (int64_t)d.iH * d.iW && st[0] % 4 == 0 && st[1] % 4 == 0; };
        const bool pw_shape = !no_pw && fast && d.kD * d.kH * d.kW == 1 && d.sD == 1 && d.sH == 1 && d.sW == 1 && d.uD == 1 && d.uH == 1 &&
            d.uW == 1 && d.up_mode == 0 && d.pD == 0 && d.pH == 0 && d.pW == 0 && d.oD == d.iD && d.oH == d.iH && d.oW == d.iW &&
            S % 4 == 0 && d.Cout % 4 == 0 && d.Cout > 32 && dense(d.x0s) && (d.Cin1 == 0 || dense(d.x1s)) &&
            reinterpret_cast<uintptr_t>(x0) % 16 == 0 && (d.Cin1 == 0 || reinterpret_cast<uintptr_t>(x1) % 16 == 0) &&
            reinterpret_cast<uintptr_t>(wp) % 16 == 0;
        // sdc_conv_splitk on the direct-form kernels' smallest tile: K split over Sd workgroups per tile into the caller's partial
        // buffer, summed in split order (+ bias) by splitk_sum_kernel
        const int Sd = (split_work && !residual && !tl_describe && d.oW % 4 == 0 && d.precision != 5) ? direct_ksplit(d, ntot, fast) : 1;
        if (Sd > 1) {
            const int64_t elems = (int64_t)d.B * d.Cout * d.oD * d.oH * d.oW;
            SDC_REQUIRE(split_bytes >= (size_t)Sd * elems * sizeof(float), SDC_EINVAL, "sdc_conv_splitk: workspace too small");
            SDC_REQUIRE(reinterpret_cast<uintptr_t>(split_work) % 16 == 0, SDC_EINVAL, "sdc_conv_splitk: workspace must be 16-byte aligned");
            ConvArgs p = a;
            p.ksplit = Sd; p.ypart_elems = elems; p.y = split_work; p.bias = nullptr; p.res = nullptr;
            p.d.ys[4] = 1; p.d.ys[3] = d.oW; p.d.ys[2] = (int64_t)d.oH * d.oW; p.d.ys[1] = p.d.ys[2] * d.oD; p.d.ys[0] = p.d.ys[1] * d.Cout;
            SDC_REQUIRE(span5(p.d.ys, d.B, d.Cout, d.oD, d.oH, d.oW) < (1ll << 30), SDC_EINVAL, "sdc_conv_splitk: partial copy too large");
            p.ydense = S >= 128 && S < (1 << 24);
            if (pw_shape) {
                SDC_PICK("conv_pw_kernel<64,64,2,2>", 1.0);
                dim3 grid((unsigned)(((a.Ntot + 63) / 64) * ((d.Cout + 63) / 64)), (unsigned)Sd);
                hipLaunchKernelGGL((conv_pw_kernel<64, 64, 2, 2>), grid, dim3(NT), 0, s, p);
            } else {
                SDC_PICK("conv_kernel<64,64,2,2,true>", 1.0);
                dim3 grid((unsigned)((a.Ntot + 63) / 64), (unsigned)((d.Cout + 63) / 64), (unsigned)Sd);
                hipLaunchKernelGGL((conv_kernel<64, 64, 2, 2, true>), grid, dim3(NT), 0, s, p);
            }
            const int64_t nq = elems / 4;
            hipLaunchKernelGGL(splitk_sum_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, (const float*)split_work, y, Sd, elems,
                               d.Cout, d.oD, d.oH, d.oW, d.ys[0], d.ys[1], d.ys[2], d.ys[3], d.ys[4], bias);
            return sdc::check_launch("sdc_conv_splitk[direct]");
        }
        if (pw_shape) {
            const int64_t b64x128 = (int64_t)((a.Ntot + 127) / 128) * ((d.Cout + 63) / 64);
            // two-workgroups-per-CU form with interleaved tiles (round 5): whole 128-channel blocks, 16-byte aligned dense rows of y
            // (and of the residual), enough 128 x 256 tiles for two rounds of the chip
            static const int no_pw2 = exp_env("SDC_NO_PW2");
            const bool al16 = reinterpret_cast<uintptr_t>(y) % 16 == 0 && d.ys[0] % 4 == 0 && d.ys[1] % 4 == 0 &&
                              (!residual || (reinterpret_cast<uintptr_t>(residual) % 16 == 0 && d.rs[0] % 4 == 0 && d.rs[1] % 4 == 0));
            if (!no_pw2 && d.Cout % 128 == 0 && a.ydense && al16 && (int64_t)((a.Ntot + 255) / 256) * (d.Cout / 128) >= 1024) {
                SDC_PICK("conv_pw2_kernel<2,2>", 1.0);
                { const int rc_ = launch_pw2<2, 2>(a, s); if (rc_) return rc_; }
                return sdc::check_launch("sdc_conv[pointwise, two workgroups per CU]");
            }
            if (!no_pw2 && d.Cout == 64 && a.ydense && al16 && (a.Ntot + 511) / 512 >= 1024) {
                SDC_PICK("conv_pw2_kernel<1,4>", 1.0);
                { const int rc_ = launch_pw2<1, 4>(a, s); if (rc_) return rc_; }
                return sdc::check_launch("sdc_conv[pointwise, two workgroups per CU]");
            }
            if (d.Cout > 64 && a.Ntot >= 128 * 256) { SDC_PICK("conv_pw_kernel<128,128,2,2>", 1.0); launch_pw<128, 128, 2, 2>(a, s); }
            else if (d.Cout <= 64 && a.Ntot >= 256 * 1024) { SDC_PICK("conv_pw_kernel<64,256,1,4>", 1.0); launch_pw<64, 256, 1, 4>(a, s); }
            else if (b64x128 >= 1024) { SDC_PICK("conv_pw_kernel<64,128,2,2>", 1.0); launch_pw<64, 128, 2, 2>(a, s); }
            else { SDC_PICK("conv_pw_kernel<64,64,2,2>", 1.0); launch_pw<64, 64, 2, 2>(a, s); }
            return sdc::check_launch("sdc_conv[pointwise]");
        }
    }
    // stem convs (kW = 7, tiny Cin): row-halo kernel with generalized k rows
    if (a.rowhalo && d.kW == 7 && d.sW == 1 && d.uD == 1 && d.uH == 1 && d.uW == 1 && d.up_mode == 0 && d.kD * d.kH <= 64 &&
        d.Cout % 4 == 0 && d.Cout > 32 && small && ((d.oW % 128 == 0) || (128 % d.oW == 0 && d.oW >= 16)) &&
        reinterpret_cast<uintptr_t>(wp) % 16 == 0) {
        static const int stem_tile = exp_env("SDC_STEM_TILE");
        if (stem_tile == 256 && (d.oW % 256 == 0 || 256 % d.oW == 0)) {
            dim3 grid((a.Ntot + 255) / 256, (d.Cout + 63) / 64);
            SDC_PICK("conv_rh_kernel<64,256,1,4,7,true>", 1.0);
            hipLaunchKernelGGL((conv_rh_kernel<64, 256, 1, 4, 7, true>), grid, dim3(NT), 0, s, a);
            return sdc::check_launch("sdc_conv[stem]");
        }
        if (stem_tile == 512 && (d.oW % 512 == 0 || 512 % d.oW == 0)) {
            dim3 grid((a.Ntot + 511) / 512, (d.Cout + 63) / 64);
            SDC_PICK("conv_rh_kernel<64,512,1,4,7,true>", 1.0);
            hipLaunchKernelGGL((conv_rh_kernel<64, 512, 1, 4, 7, true>), grid, dim3(NT), 0, s, a);
            return sdc::check_launch("sdc_conv[stem]");
        }
        dim3 grid((a.Ntot + 127) / 128, (d.Cout + 63) / 64);
        SDC_PICK("conv_rh_kernel<64,128,2,2,7,true>", 1.0);
        if (stem_tile == 16) hipLaunchKernelGGL((conv_rh_kernel<64, 128, 2, 2, 7, true, 16>), grid, dim3(NT), 0, s, a);
        else hipLaunchKernelGGL((conv_rh_kernel<64, 128, 2, 2, 7, true, 8>), grid, dim3(NT), 0, s, a);
        return sdc::check_launch("sdc_conv[stem]");
    }
    const int64_t blocks64x128 = (int64_t)((a.Ntot + 127) / 128) * ((d.Cout + 63) / 64);
    static const int force_tile = exp_env("SDC_TILE");   // tuning knob: 1..5 picks a tile
    if (force_tile && d.Cout > 32) {
        switch (force_tile) {
            case 1: SDC_LAUNCH(128, 128, 2, 2); break;
            case 2: SDC_LAUNCH(64, 256, 1, 4); break;
            case 3: SDC_LAUNCH(64, 128, 2, 2); break;
            case 4: SDC_LAUNCH(64, 64, 2, 2); break;
            default: SDC_LAUNCH(32, 128, 1, 4); break;
        }
        if (tl_describe) return SDC_OK;
        return sdc::check_launch("sdc_conv");
    }
    if (d.Cout > 64 && a.Ntot >= 128 * 256)
        SDC_LAUNCH(128, 128, 2, 2);
    else if (d.Cout > 32 && d.Cout <= 64 && a.Ntot >= 256 * 1024)
        SDC_LAUNCH(64, 256, 1, 4);     // wide tile: each wave owns 64x64 (2x2 MFMA tiles) like the 128x128 case
    else if (d.Cout > 32 && blocks64x128 >= 1024)
        SDC_LAUNCH(64, 128, 2, 2);
    else if (d.Cout > 32)
        SDC_LAUNCH(64, 64, 2, 2);      // small-N layers: twice the workgroups, >= 2 per CU
    else
        SDC_LAUNCH(32, 128, 1, 4);
    if (tl_describe) return SDC_OK;
    return sdc::check_launch("sdc_conv");
}

}  // namespace
