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
#include <cstdlib>
#include <type_traits>

namespace {

// Dispatch overrides and tuning knobs (SDC_NO_WG2, SDC_TILE, ...) exist only in experiment builds (-DSDC_KERNEL_EXPERIMENTS,
// tools/): the shipping library reads no environment variable; the conv algorithm is chosen by SdcConvDesc.precision alone.
#ifdef SDC_KERNEL_EXPERIMENTS
inline int exp_env(const char* name) { const char* v = getenv(name); return v ? atoi(v) : 0; }
#else
constexpr int exp_env(const char*) { return 0; }
#endif


constexpr int BK = 16;          // K chunk per LDS stage (BK = 32 measured no faster: the kernel is MFMA-issue bound)
constexpr int NT = 256;

typedef float f32x16 __attribute__((ext_vector_type(16)));

// rebuild a pointer from two scalar registers: tells the compiler the base is wave-uniform so that the load
// can use the (SGPR base + 32-bit VGPR offset) addressing form
typedef const __attribute__((address_space(1))) float* gfloat_p;     // global (not flat) address space
__device__ __forceinline__ gfloat_p uniform_ptr(const float* p) {
    const uint64_t u = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return (gfloat_p)(((uint64_t)hi << 32) | lo);
}
#define SDC_UNIFORM(v) __builtin_amdgcn_readfirstlane(v)
// uniform base + 32-bit per-lane BYTE offset (kept as a byte offset so the zero-extension is exact and the
// backend can select `global_load_dword v, v_off, s[base:base+1]`)
__device__ __forceinline__ float ld_sv(gfloat_p base, uint32_t byte_off) {
    typedef const __attribute__((address_space(1))) char* gchar_p;
    return *(gfloat_p)((gchar_p)base + byte_off);
}

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
    int rowhalo;   // allow the row-halo kernel (env SDC_NO_ROWHALO=1 disables it for A/B timing)
    int vec2;      // Winograd epilogue: y (and residual) rows allow 8-byte accesses at even positions
    int ydense;    // y (and the residual) dense per sample and below 2^30 elements: conv_epilogue addresses them as scalar channel base + 32-bit lane offset
    const float* wg2;   // F(2x2,3x3) taps [kd][Cin][Cout][16] (precision 3) / F(2x2x2,3x3x3) taps [jd][Cin][Cout][16] (precision 4)
    // GroupNorm partial sums of the output (sdc_conv_gn): fp64 (sum, sum of squares) per (sample, group, part)
    double* gn_part;
    int gn_G, gn_cpg, gn_nparts, gn_S;
};

// Position-tile numbering: workgroups are dealt round-robin over the 8 XCDs, each with a private L2.  Neighbouring
// position tiles share their halo rows (kh / kd taps), so consecutive LOGICAL tiles are given to one XCD
// (bijective when the tile count is a multiple of 8; speed / HBM traffic only, never correctness).
__device__ __forceinline__ int xcd_tile(int bid, int ntiles) {
    return (ntiles & 7) == 0 ? (bid & 7) * (ntiles >> 3) + (bid >> 3) : bid;
}

// ---- epilogue: D rows (co) live in registers, columns (positions) on lanes -> coalesced along W.
// Bias / residual loads are issued as a batch (clamped addresses, no per-element branches) so the
// workgroup pays one memory round trip per 16 outputs instead of sixteen.
template <int TM, int TN>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[TM][TN], int mw, int nw, int lane) {
    const SdcConvDesc& d = a.d;
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
            gwchar_p yb = (gwchar_p)(__attribute__((address_space(1))) void*)uniform_ptr(a.y + (int64_t)cob * d.ys[1]);
            gchar_p rb = (gchar_p)uniform_ptr(a.res ? a.res + (int64_t)cob * d.rs[1] : a.y);
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
                if (pok && co < d.Cout) a.y[yoff + co * d.ys[1]] = acc[i][j][rr] + bv[rr] + rv[rr];
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, bool FAST>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4))) void conv_kernel(const ConvArgs a) {
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
    int f_kd = 0, f_kh = 0, f_kw = 0, f_ci = 0;
    bool f_ok = false;
    // per-thread 32-bit element offsets of (b, id, ih, iw) inside x0 / x1 for the current tap; the channel
    // part of every address is wave-uniform and stays in scalar registers (global_load saddr + voffset form)
    uint32_t f_v0 = 0, f_v1 = 0;
    const int brow_u = __builtin_amdgcn_readfirstlane(brow0);
    const int arow_u = __builtin_amdgcn_readfirstlane(arow0);
    // (BM = 32: a wave spans two weight rows, the second one is folded into the per-lane offset)
    const uint32_t a_v = (uint32_t)((arow0 - arow_u) * d.Cout + aco_c);
    int a_k = arow_u;                   // weight row of this wave's first A load in the chunk in flight
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

    conv_epilogue<TM, TN>(a, acc, m0 + wm * (TM * 32), n0 + wn * (TN * 32), lane);
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
    const int n0 = xcd_tile(blockIdx.x, gridDim.x) * BN;
    const int m0 = blockIdx.y * BM;
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
    auto load_chunk = [&](int kc) {
        const int ci = kc * BK;
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

    const int nchunks = a.Cin / BK;
    load_chunk(0);
    store_chunk(0);
    if (nchunks > 1) load_chunk(1);
    __syncthreads();
    const int l31 = lane & 31, lh = lane >> 5;
    const int am = wm * (TM * 32) + l31, bn = wn * (TN * 32) + l31;
    for (int kc = 0; kc < nchunks; ++kc) {
        const int buf = kc & 1;
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
    }
    conv_epilogue<TM, TN>(a, acc, m0 + wm * (TM * 32), n0 + wn * (TN * 32), lane);
}

template <int BM, int BN, int WM, int WN>
void launch_pw(const ConvArgs& a, hipStream_t s) {
    dim3 grid((a.Ntot + BN - 1) / BN, (a.d.Cout + BM - 1) / BM);
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

    for (int st = 0; st < nstages; ++st) {
        const int buf = st & 1;
        if (st + 1 < nstages) load_stage();
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
        if (st + 1 < nstages) store_stage(buf ^ 1);
        __syncthreads();
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
template <int TM, int TP, int BM, int BN, int WM, int WN>
__device__ __forceinline__ void wg_epilogue(const ConvArgs& a, f32x16 (&acc)[4][TM][TP], int mw, int pw, int lane,
                                            float* scratch, int wave, int m0, int n0) {
    const SdcConvDesc& d = a.d;
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
            const int pp = 2 * (pw + j * 32 + l31);          // even position; pp + 1 is in the same row (oW even)
            const bool pok = pp < a.Ntot;
            int r = pok ? pp : 0;
            const int qw = r % d.oW; r /= d.oW;
            const int qh = r % d.oH; r /= d.oH;
            const int qd = r % d.oD; const int qb = r / d.oD;
            const int64_t yoff = qb * d.ys[0] + qd * d.ys[2] + qh * d.ys[3] + qw * d.ys[4];
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
                    float* yp = a.y + yoff + co * d.ys[1];
                    if (v2) *reinterpret_cast<float2*>(yp) = make_float2(y0, y1);
                    else { yp[0] = y0; yp[d.ys[4]] = y1; }
                    if (gn) { gs[i][rr >> 2] += (double)y0 + (double)y1; gq[i][rr >> 2] += (double)y0 * y0 + (double)y1 * y1; }
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
                if (lane == 0) { scr[(wave * TM * 4 + i * 4 + b4) * 2] = s1; scr[(wave * TM * 4 + i * 4 + b4) * 2 + 1] = q1; }
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
template <int BM, int BN, int WM, int WN, int SK, int NTH, bool UPS>
__global__ __launch_bounds__(NTH) void conv_wg_kernel(const ConvArgs a) {
    constexpr int TM = BM / WM / 32;
    constexpr int TP = BN / 2 / WN / 32;                        // 32-pair column tiles per wave
    constexpr int KSMAX = BN + (BN / 16) * 2;                   // LDS floats per k row, worst case (16-wide rows)
    constexpr int NCOL = (KSMAX + 63) / 64;
    constexpr int KROWS = SK / (NTH / 64);
    constexpr int NA4 = 4 * SK * BM / 4 / NTH;                  // float4 weight loads per thread per stage
    static_assert(TM >= 1 && TP >= 1 && KROWS >= 1 && NA4 >= 1 && WM * WN * 64 == NTH, "bad tile");
    extern __shared__ __attribute__((aligned(16))) float ldsw[];
    constexpr int PITCH = NCOL * 64 + 8;                        // LDS floats per staged k row: every (lane, sweep) slot exists, so the
                                                                // staging stores need no per-lane predicate (+8: bank spread of the two half-waves)
    constexpr int ASZ = 4 * SK * BM, BSZ = SK * PITCH;          // floats per buffer
    float* const As = ldsw;                                     // [2][4][SK][BM]
    float* const Bs = ldsw + 2 * ASZ;                           // [2][SK * ks_stride]

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
        const int pos = 2 * (wn * (TP * 32) + j * 32 + l31);
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
    const float* wg = a.wp + (int64_t)a.Ktot * d.Cout;          // transformed taps, 4 per (kd, kh)

    float breg[KROWS][NCOL];
    float4 areg[NA4];
    uint32_t mbits = 0;
    int s_kd = 0, s_kh = 0, s_ci = 0;

    // The fetch of a stage is cut into NSL pieces that are issued BETWEEN the MFMA groups of the main loop (a
    // workgroup's loads all at once keep the CU's one vector-memory pipe -- 64 B/clk -- busy for hundreds of cycles
    // with the matrix cores idle behind it).  Stage st+2 is fetched to registers during the second half of
    // stage st and written to the other LDS buffer during the first half of stage st+1.
    constexpr int NSL = SK / 4;                                  // pieces (= half of the k-steps of a stage)
    int l_tap = 0, l_ci = 0;
    const float* l_base = a.x0;
    int64_t l_sc = 0;
    uint32_t l_off[NCOL];                         // byte offsets (every operand spans < 2^30 elements: host check `small`)
    auto load_begin = [&]() {
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
        const gfloat_p wbase = uniform_ptr(wg + ((int64_t)(l_tap * 4) * a.Cin + l_ci) * d.Cout);
#pragma unroll
        for (int i = 0; i < NA4; ++i) {
            if (i % NSL != p) continue;
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
            float4 v = areg[i];
            if (!a_ok[i]) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(As + buf * ASZ + row * BM + c4) = v;
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

    f32x16 acc[4][TM][TP];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][i][j][r] = 0.0f;

    const int nstages = d.kD * d.kH * (a.Cin / SK);
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

    for (int st = 0; st < nstages; ++st) {
        const int buf = st & 1;
        const float* Ab = As + buf * ASZ;
        const float* Bb = Bs + buf * BSZ;
        float fa[2][4][TM];
        float2 fb[2][TP][2];
        auto read_frag = [&](int ks, int set) {
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[set][x][i] = Ab[(x * SK + 2 * ks + lh) * BM + am + i * 32];
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const float2* bp = reinterpret_cast<const float2*>(Bb + (2 * ks + lh) * PITCH + boff[j]);
                fb[set][j][0] = bp[0];
                fb[set][j][1] = bp[1];
            }
        };
        // Input transform of a k-step's B fragments is done one step ahead, and the LDS reads, the staging piece and
        // that transform are spread over the shadows of the step's MFMAs (sched_group_barrier): a wave that issues its
        // MFMAs back to back and only then its other instructions leaves the matrix pipe idle while it catches up --
        // the two waves of a SIMD drift into running one after the other, so nobody else fills those gaps.
        float bt[2][TP][4];
        auto transform = [&](int set) {
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const float2 p0 = fb[set][j][0], p1 = fb[set][j][1];
                bt[set][j][0] = p0.x - p1.x;
                bt[set][j][1] = p0.y + p1.x;
                bt[set][j][2] = p1.x - p0.y;
                bt[set][j][3] = p0.y - p1.y;
            }
        };
        read_frag(0, 0);
        transform(0);
#pragma unroll
        for (int ks = 0; ks < SK / 2; ++ks) {
            const int set = ks & 1;
            if (ks + 1 < SK / 2) read_frag(ks + 1, set ^ 1);
            if (ks < NSL) {
                store_piece(buf ^ 1, ks);
            } else {
                if (ks == NSL) load_begin();
                load_piece(ks - NSL);
            }
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TP; ++j)
                        acc[x][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][x][i], bt[set][j][x], acc[x][i][j], 0, 0, 0);
            if (ks + 1 < SK / 2) transform(set ^ 1);
            // interleave: the next fragments' LDS reads behind the first MFMA, then per MFMA a few VALU and one staging
            // access (LDS write / global load)
#pragma unroll
            for (int m = 0; m < 4 * TM * TP; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (m < 2) __builtin_amdgcn_sched_group_barrier(0x100, 2 * TM + 2 * TP, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x220, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    wg_epilogue<TM, TP, BM, BN, WM, WN>(a, acc, m0 + wm * (TM * 32), n0 / 2 + wn * (TP * 32), lane, ldsw, wave, m0, n0);
}


// ------------------------------------------------------------------------------------------------
// precision = 3: fp32 Winograd F(2x2, 3x3) over (H, W) for the 3x3 (Conv2d) / 3x3x3 (Conv3d, direct along D) stride-1
// convs: per 2x2 output tile and its 4x4 input patch d,  Y = A^T [ (G g G^T) . (B^T d B) ] A  with the same G, B^T, A^T as
// the 1-D form above applied along H and along W -- 16 products per 4 outputs instead of 36: 4/9 of the direct fp32 MFMA
// work (the 1-D form: 2/3).  The caller stores U[kd][ci][co][j*4+xi] = sum_{kh,kw} G[j][kh] G[xi][kw] w[co][ci][kd][kh][kw]
// (fp64, rounded once; the 16 components of one (ci, co) contiguous) behind the 1-D taps.  16 GEMMs over K = kD*Cin, N = tiles:
//   * workgroup = 4 waves, ONE per SIMD (16 components x 32x32 accumulators = 256 registers per wave), 64 output
//     channels x 64 tiles (= RP whole row pairs of W/2 tiles, 256 output positions); K stages of 8 channels.
//   * What decides the speed (tools/mfma_fill.hip, tools/wg_probe.py with parts of the loop switched off): the fp32 MFMA runs
//     at the fp32 VALU rate and a VALU instruction issued between two MFMAs of the wave is NOT hidden behind them (~7-11
//     cycles each, a global load ~14); ds_read_b128 / ds_write_b32 are nearly free up to two per MFMA, SALU is free.  So
//     the loop is built to need few VALU and VMEM instructions per MFMA:
//   * B: the whole input transform V = B^T d B happens ONCE per workgroup, when a stage is parked in LDS -- not per wave at
//     fragment-read time.  A lane owns CPL adjacent columns of one row pair (CPL = W/16, so a 16-lane DPP row is one image
//     row; W = 16: two rows interleaved): it loads the 4 input rows with one vector load each, applies the H transform in
//     registers, then the W transform with the neighbour columns taken through DPP row shifts (bound_ctrl supplies the zero
//     padding at the row ends), and stores V[k][j][tile][xi] with ds_write_b128.  A wave then reads its B fragments with four
//     ds_read_b128 per k-step and feeds them to the MFMAs as they are.
//   * A: U tile [8][64][16] per stage with 16-byte loads (scalar base + fixed lane offset), parked with one ds_write_b128
//     and read back as four ds_read_b128 per k-step (16-byte chunks XOR-swizzled by (row >> 2) & 3: conflict-free).
//   * two LDS stage buffers, ONE barrier per stage (end of k-step 2: every park of stage st+1 is done and every read of the
//     buffer that stage st+2's parks will overwrite has returned); stage st+2 is fetched while stage st computes; the last
//     k-step of a stage already reads the first fragments of the next one.
//   * all LDS offsets and the row taps are immediates (W is a template parameter, input rows contiguous); the padding masks
//     are skipped by whole waves whose rows are all inside the image.
// Results differ from the direct form by rounding order (measured 2-6e-7 of the output scale vs fp64; direct form 1-2e-6).
constexpr int W2_SK = 8;          // channels per stage
constexpr int W2_BM = 64;         // output channels per workgroup
constexpr int W2_TILES = 64;      // 2x2 tiles per workgroup
constexpr int W2_NBUF = 2;        // LDS stage buffers
constexpr int W2_ASZ = 16 * W2_SK * W2_BM, W2_BSZ = 16 * W2_SK * W2_TILES;    // floats per stage: U tile, V tile

// single fp32 VALU ops the SLP vectoriser cannot pack (v_pk_add_f32 beside MFMAs is slower than two v_add_f32)
__device__ __forceinline__ float vsub1(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vadd1(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// Packed / DPP forms of the park-time transform (a VALU instruction between two fp32 MFMAs costs the same whether it
// produces one result or two, and a DPP operand is free):
typedef float w2f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ w2f2 pk_add2(w2f2 a, w2f2 b) { w2f2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ w2f2 pk_sub2(w2f2 a, w2f2 b) { w2f2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ w2f2 pk_mul2(w2f2 a, w2f2 b) { w2f2 r; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// a * b - c ;  c - a * b
__device__ __forceinline__ w2f2 pk_fms2(w2f2 a, w2f2 b, w2f2 c) { w2f2 r; asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ w2f2 pk_fnma2(w2f2 a, w2f2 b, w2f2 c) { w2f2 r; asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// (c1, c2) -> (c1 + c2, c2 - c1)
__device__ __forceinline__ w2f2 pk_sumdiff(w2f2 c) {
    w2f2 r;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(c));
    return r;
}
// (c1, c2) -> (c1 + c2, c1 - c2);  (s.x + a.x, s.y - a.y);  a * a + c
__device__ __forceinline__ w2f2 pk_sumdiff_fwd(w2f2 c) {
    w2f2 r;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(c));
    return r;
}
__device__ __forceinline__ w2f2 pk_addsub(w2f2 s, w2f2 a) { w2f2 r; asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(s), "v"(a)); return r; }
__device__ __forceinline__ w2f2 pk_sqacc(w2f2 a, w2f2 c) { w2f2 r; asm("v_pk_fma_f32 %0, %1, %1, %2" : "=v"(r) : "v"(a), "v"(c)); return r; }
// Sums over the 64 lanes of eight fp64 values per lane with 10 exchanges instead of 48: halve the value set at the xor-32,
// -16 and -8 levels (each lane keeps the half its lane bit selects), then an all-reduce of the one value left over xor 4, 2, 1.
// Afterwards every lane holds the wave total of value (lane >> 3).  Fixed order: the result does not depend on anything but
// the 512 inputs.
__device__ __forceinline__ double wave_sum8(const double (&v)[8], int lane) {
    double w[4], u[2];
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (b5 ? v[4 + i] : v[i]) + __shfl_xor(b5 ? v[i] : v[4 + i], 32, 64);
#pragma unroll
    for (int i = 0; i < 2; ++i) u[i] = (b4 ? w[2 + i] : w[i]) + __shfl_xor(b4 ? w[i] : w[2 + i], 16, 64);
    double t = (b3 ? u[1] : u[0]) + __shfl_xor(b3 ? u[0] : u[1], 8, 64);
    t += __shfl_xor(t, 4, 64);
    t += __shfl_xor(t, 2, 64);
    t += __shfl_xor(t, 1, 64);
    return t;
}
// value of lane - S of the same 16-lane row (0 past the row end) minus b;  a minus the value of lane + S
template <int S>
__device__ __forceinline__ float sub_prev(float x, float b) {
    float r;
    if (S == 1) asm("v_sub_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(r) : "v"(x), "v"(b));
    else asm("v_sub_f32_dpp %0, %1, %2 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(r) : "v"(x), "v"(b));
    return r;
}
template <int S>
__device__ __forceinline__ float sub_next(float a, float x) {
    float r;
    if (S == 1) asm("v_subrev_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(r) : "v"(x), "v"(a));
    else asm("v_subrev_f32_dpp %0, %1, %2 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(r) : "v"(x), "v"(a));
    return r;
}

template <int OW, int DBG>
__global__ __launch_bounds__(256) void conv_wg2_kernel(const ConvArgs a) {
    constexpr int SK = W2_SK, BM = W2_BM, NTH = 256;
    constexpr int TW = OW / 2, RP = W2_TILES / TW;
    constexpr int LGW = OW == 128 ? 7 : (OW == 64 ? 6 : (OW == 32 ? 5 : 4));
    constexpr int CPL = OW == 16 ? 2 : OW / 16;          // adjacent columns per lane
    constexpr int SH = OW == 16 ? 2 : 1;                 // DPP lane distance of the neighbouring column group
    constexpr int LPK = 128 / CPL;                       // lanes per staged channel (k row)
    constexpr int NIT = CPL == 2 ? 2 : 1;                // park items per thread and stage
    constexpr int KPW = CPL == 8 ? 4 : 2;                // k rows per (parking) wave
    constexpr int TPL = CPL / 2;                         // tiles per lane and row pair
    typedef float nfloat4 __attribute__((ext_vector_type(4)));
    typedef float nfloat2 __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(1))) char* gchar_p;
    typedef const __attribute__((address_space(1))) nfloat4* gfloat4_p;
    typedef const __attribute__((address_space(1))) nfloat2* gfloat2_p;
    extern __shared__ __attribute__((aligned(16))) float ldsw[];
    float* const As = ldsw;                          // [2][SK][BM][16]  (chunk q of row m at slot q ^ ((m >> 2) & 3))
    float* const Vs = ldsw + W2_NBUF * W2_ASZ;       // [2][SK][4 j][64 tiles][4 xi]

    const SdcConvDesc& d = a.d;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = SDC_UNIFORM(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int MT = (d.Cout + BM - 1) / BM;
    // consecutive logical blocks (one XCD, dispatched back to back) share an input tile: m fastest
    const int lb = xcd_tile(blockIdx.x, gridDim.x);
    const int m0 = (lb % MT) * BM;
    const int tile0 = (lb / MT) * W2_TILES;
    const int H2 = d.oH >> 1;
    const int RPtot = d.B * d.oD * H2;
    const int rp0 = tile0 >> (LGW - 1);
    const bool two = d.Cin1 > 0;
    const float r_H2 = 1.0f / (float)H2, r_oD = 1.0f / (float)d.oD;
    auto split_rp = [&](int rp, int& ob, int& od, int& hp) {      // rp < 2^20 (host check): float quotients are exact
        const int q = (int)(((float)rp + 0.5f) * r_H2);
        hp = rp - q * H2;
        ob = (int)(((float)q + 0.5f) * r_oD);
        od = q - ob * d.oD;
    };

    // ---- park geometry of this thread: k row inside the stage, row pair r, first column c0 (CPL columns c0 .. c0+CPL-1)
    const bool parker = CPL != 8 || wave < 2;                      // W = 128: 128 items per stage, waves 0 and 1 park
    const int ksub = CPL == 4 ? (lane >> 5) : (CPL == 8 ? (lane >> 4) : 0);        // k row inside the wave's group (CPL = 2: the item)
    const int lik = lane & (LPK - 1);
    int pr, pc0;
    if (OW == 16) { pr = 2 * (lik >> 4) + (lik & 1); pc0 = 2 * ((lik & 15) >> 1); }
    else { pr = lik >> 4; pc0 = (lik & 15) * CPL; }
    // byte offset of (b, od, 2hp, c0) inside one channel of x0 / x1 plus this lane's k-row offset; the depth-tap shift
    // (kd - pD) planes is added once per stage.  Zero padding without per-element masks: the row above / below the row pair
    // is fetched from a clamped (valid) row and multiplied by a 0 / 1 factor inside the H transform (one fused op, no extra
    // instruction); a depth tap outside the volume reads the lane's own plane and is multiplied by 0 the same way.
    uint32_t vp0 = 0, vp1 = 0, dmsk = 0, rsel0 = 0, rsel3 = 0;
    float m0f = 0.0f, m3f = 0.0f;
    {
        const int rp = rp0 + pr;
        if (parker && rp < RPtot) {
            int ob, od, hp;
            split_rp(rp, ob, od, hp);
            if (hp > 0) { m0f = 1.0f; rsel0 = OW * 4; }                       // row 2hp - 1 exists
            if (2 * hp + 2 < d.iH) { m3f = 1.0f; rsel3 = 2 * OW * 4; }        // row 2hp + 2 exists
            for (int kd = 0; kd < d.kD; ++kd) dmsk |= (od - d.pD + kd >= 0 && od - d.pD + kd < d.iD) ? (1u << kd) : 0u;
            vp0 = (uint32_t)(ksub * d.x0s[1] + ob * d.x0s[0] + od * d.x0s[2] + (2 * hp) * OW + pc0) * 4u;
            if (two) vp1 = (uint32_t)(ksub * d.x1s[1] + ob * d.x1s[0] + od * d.x1s[2] + (2 * hp) * OW + pc0) * 4u;
        }
    }
    // park position: V[k][j][tile][4]; k = wave * KPW + ksub (+ item for CPL = 2), tile = pr * TW + c0 / 2 (+ t)
    const int vpark = ((wave * KPW + ksub) * 4 * W2_TILES + pr * TW + (pc0 >> 1)) * 4;          // floats; + j * 256, + item * 1024
    // B fragments: tile n = wn * 32 + l31 of k row 2ks + lh: four 16-byte reads (j = 0..3)
    const int boff = (lh * 4 * W2_TILES + wn * 32 + l31) * 4;
    // A fragments: row (k = 2ks + lh, m): four 16-byte chunks, chunk q at slot q ^ ((m >> 2) & 3)
    const int arow = wm * 32 + l31;
    int aoff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) aoff[q] = (lh * BM + arow) * 16 + 4 * (q ^ ((arow >> 2) & 3));
    // weight fetch / park: float4 f = tid + 256 i: k = i, m = tid >> 2, chunk = tid & 3
    const int pm = tid >> 2, pq = tid & 3;
    const int apark = pm * 16 + 4 * (pq ^ ((pm >> 2) & 3));                        // + i * BM * 16 floats
    const int pmc = (m0 + pm) < d.Cout ? (m0 + pm) : d.Cout - 1;                   // rows beyond Cout: any valid row (never stored)
    uint32_t a_voff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a_voff[i] = (uint32_t)(((int64_t)i * d.Cout + pmc) * 16 + 4 * pq) * 4u;

    w2f2 braw[NIT][4][TPL];       // [item][source row j][column pair]
    nfloat4 areg[8];
    int s_kd = 0, s_ci = 0;
    const int64_t xs1_0 = d.x0s[1], xs1_1 = d.x1s[1];
    const int xs2_0 = (int)d.x0s[2], xs2_1 = (int)d.x1s[2];
    const int cin0 = d.Cin0, cin = a.Cin, kDn = d.kD, coutn = d.Cout, pDn = d.pD;
    const float* const wg2p = a.wg2;
    const float* const x0p = a.x0;
    const float* const x1p = two ? a.x1 : a.x0;
    // stage fetch state: scalar bases (weights; the first channel row of this wave), the lane offsets of the selected input
    // (rows 2hp / 2hp+1 at voff / voff + W*4, the clamped rows above / below at voff0 / voff3), the stage's 0 / 1 factors
    gfloat_p f_w = uniform_ptr(wg2p), f_x = uniform_ptr(x0p);
    int64_t f_sc = 0;
    uint32_t voff = 0, voff0 = 0, voff3 = 0;
    w2f2 mk0 = {0.f, 0.f}, mk3 = {0.f, 0.f}, mk12 = {0.f, 0.f};
    auto fetch_begin = [&]() {
        const bool first = s_ci < cin0;
        f_sc = first ? xs1_0 : xs1_1;
        const float* bsel = first ? x0p : x1p;
        const int cbase = (first ? s_ci : s_ci - cin0) + wave * KPW;
        const uint32_t dsb = (uint32_t)((s_kd - pDn) * (first ? xs2_0 : xs2_1) * 4);   // bytes, two's complement
        f_x = uniform_ptr(bsel + (int64_t)cbase * f_sc);
        f_w = uniform_ptr(wg2p + ((int64_t)(s_kd * cin + s_ci) * coutn) * 16);
        const bool dv = (dmsk >> s_kd) & 1u;
        voff = (first ? vp0 : vp1) + (dv ? dsb : 0u);
        voff0 = voff - rsel0;
        voff3 = voff + rsel3;
        const float md = dv ? 1.0f : 0.0f, a0 = dv ? m0f : 0.0f, a3 = dv ? m3f : 0.0f;
        mk12 = w2f2{md, md};
        mk0 = w2f2{a0, a0};
        mk3 = w2f2{a3, a3};
        s_ci += SK;
        // (past the last stage the walk wraps to the first one: the extra fetches of the pipeline tail stay in bounds and are
        // never consumed)
        if (s_ci >= cin) { s_ci = 0; if (++s_kd == kDn) s_kd = 0; }
    };
    // (the offset passes through an empty asm so that its zero-extension is not hoisted out of the loop as a 64-bit
    // register pair: the load then takes the scalar base + 32-bit lane offset form)
    auto fetch_a = [&](int i, nfloat4 (&ar)[8]) { uint32_t o = a_voff[i]; asm volatile("" : "+v"(o)); ar[i] = *(gfloat4_p)((gchar_p)f_w + o); };
    // the 4 input rows under the row pair, CPL adjacent columns each: one vector load per row, row tap (j - 1) as an immediate.
    // A lane whose row is outside the image reads the first elements of the channel instead and is zeroed when parked.
    auto fetch_b_row = [&](int it, int j, w2f2 (&br)[NIT][4][TPL]) {
        const gchar_p rb = (gchar_p)f_x + (CPL == 2 ? (int64_t)it * f_sc * 4 : 0);
        const gchar_p p = j == 0 ? rb + voff0 : (j == 3 ? rb + voff3 : (j == 1 ? rb + voff : rb + voff + OW * 4));
        if (CPL == 2) { const nfloat2 v = *(gfloat2_p)p; br[it][j][0] = w2f2{v.x, v.y}; }
        else {
#pragma unroll
            for (int h = 0; h < CPL / 4; ++h) {
                const nfloat4 v = *(gfloat4_p)(p + 16 * h);
                br[it][j][2 * h] = w2f2{v.x, v.y};
                br[it][j][2 * h + 1] = w2f2{v.z, v.w};
            }
        }
    };
    auto fetch_b = [&](int it, w2f2 (&br)[NIT][4][TPL]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fetch_b_row(it, j, br);
    };
    auto park_a = [&](int buf, int i, const nfloat4 (&ar)[8]) {
        *reinterpret_cast<nfloat4*>(As + buf * W2_ASZ + i * (BM * 16) + apark) = ar[i];
    };
    // input transform of one item: H transform per column pair (packed), then per transformed row j the W transform of the
    // lane's TPL tiles -- V0 = c0 - c2, (V1, V2) = (c1 + c2, c2 - c1) packed, V3 = c1 - c3, the neighbour columns c0 / c3 of the
    // edge tiles through DPP -- and one 16-byte store per tile, slot order (V1, V2, V0, V3)
    w2f2 hrow[4][TPL];
    auto park_b_h = [&](int it, const w2f2 (&br)[NIT][4][TPL], w2f2 k0, w2f2 k12, w2f2 k3) {
#pragma unroll
        for (int t = 0; t < TPL; ++t) {
            const w2f2 t1 = pk_mul2(br[it][1][t], k12), t2 = pk_mul2(br[it][2][t], k12);
            hrow[0][t] = pk_fms2(br[it][0][t], k0, t2);          // d0 - d2
            hrow[1][t] = pk_add2(t1, t2);                        // d1 + d2
            hrow[2][t] = pk_sub2(t2, t1);                        // d2 - d1
            hrow[3][t] = pk_fnma2(br[it][3][t], k3, t1);         // d1 - d3
        }
    };
    auto park_b_w = [&](int buf, int it, int j) {
        float* dst = Vs + buf * W2_BSZ + vpark + (CPL == 2 ? it * (4 * W2_TILES * 4) : 0) + j * (W2_TILES * 4);
#pragma unroll
        for (int t = 0; t < TPL; ++t) {
            const w2f2 cc = hrow[j][t];
            const w2f2 sd = pk_sumdiff(cc);
            const float v0 = t == 0 ? sub_prev<SH>(hrow[j][TPL - 1].y, cc.y) : vsub1(hrow[j][t - 1].y, cc.y);
            const float v3 = t == TPL - 1 ? sub_next<SH>(cc.x, hrow[j][0].x) : vsub1(cc.x, hrow[j][t + 1].x);
            nfloat4 v;
            v.x = sd.x; v.y = sd.y; v.z = v0; v.w = v3;
            *reinterpret_cast<nfloat4*>(dst + t * 4) = v;
        }
    };
    auto park_b = [&](int buf, int it, const w2f2 (&br)[NIT][4][TPL], w2f2 k0, w2f2 k12, w2f2 k3) {
        park_b_h(it, br, k0, k12, k3);
#pragma unroll
        for (int j = 0; j < 4; ++j) park_b_w(buf, it, j);
    };

    f32x16 acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
    // The bias is the start value of component (j, xi) = (1, 1): A^T M A hands that component to each of the 2x2 outputs with
    // coefficient +1.  Register r of a lane is channel m0 + 32 wm + 8 (r >> 2) + 4 lh + (r & 3).
    if (a.bias) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = m0 + wm * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
            acc[5][r] = a.bias[co < d.Cout ? co : d.Cout - 1];
        }
    }

    const int nstages = d.kD * (a.Cin / SK);
    nfloat4 fa[2][4];             // [set][chunk j]: U components (j, xi = 0..3) of this lane's (k, m)
    nfloat4 fv[2][4];             // [set][j]:       V components (j, xi = 0..3) of this lane's (k, tile)
    auto read_a = [&](const float* Ak, int set, int q) { fa[set][q] = *reinterpret_cast<const nfloat4*>(Ak + aoff[q]); };
    auto read_v = [&](const float* Vk, int set, int j) { fv[set][j] = *reinterpret_cast<const nfloat4*>(Vk + boff + j * (W2_TILES * 4)); };
    w2f2 pk0, pk12, pk3, nk0 = {0.f, 0.f}, nk12 = {0.f, 0.f}, nk3 = {0.f, 0.f};     // 0 / 1 factors of the items still in registers
    {   // prologue: the fetches of the first two stages travel together; stage 0 is parked in buffer 0
        w2f2 braw0[NIT][4][TPL];
        nfloat4 areg0[8];
        fetch_begin();
        const w2f2 q0 = mk0, q12 = mk12, q3 = mk3;
#pragma unroll
        for (int i = 0; i < 8; ++i) fetch_a(i, areg0);
        if (parker) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) fetch_b(it, braw0);
        }
        fetch_begin();
#pragma unroll
        for (int i = 0; i < 8; ++i) fetch_a(i, areg);
        if (parker) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) fetch_b(it, braw);
        }
        pk0 = mk0; pk12 = mk12; pk3 = mk3;
#pragma unroll
        for (int i = 0; i < 8; ++i) park_a(0, i, areg0);
        if (parker) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) park_b(0, it, braw0, q0, q12, q3);
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) read_a(As, 0, q);
#pragma unroll
    for (int j = 0; j < 4; ++j) read_v(Vs, 0, j);

    // Main loop: stage st computes from buffer st & 1 and parks stage st+1 in the other one during its k-steps 0-1, re-using
    // each register piece for the fetch of stage st+2 as soon as it is parked (a fetch then has four k-steps to arrive); the
    // one barrier sits at the end of k-step 2; k-step 3 reads the first fragments of stage st+1, so the MFMA stream runs
    // through the stage boundary.  The non-MFMA work of a k-step sits in 16 slots, one behind each MFMA, in source order;
    // no branch inside the loop (a per-slot fast / masked choice cost conservative vmcnt waits at every join).
    int rbuf = 0;
    for (int st = 0; st < nstages; ++st) {
        const int wbuf = rbuf ^ 1;
        const float* Ab = As + rbuf * W2_ASZ;
        const float* Vb = Vs + rbuf * W2_BSZ;
        const float* An = As + wbuf * W2_ASZ;
        const float* Vn = Vs + wbuf * W2_BSZ;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int set = ks & 1, nset = set ^ 1;
            const float* Ak = ks < 3 ? Ab + (2 * (ks + 1)) * (BM * 16) : An;               // fragments of the next k-step
            const float* Vk = ks < 3 ? Vb + (2 * (ks + 1)) * (4 * W2_TILES * 4) : Vn;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                // component (j = c >> 2, xi = c & 3): V slot order in LDS is (V1, V2, V0, V3)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][c >> 2][c & 3],
                                                              fv[set][c >> 2][(c & 3) == 0 ? 2 : ((c & 3) == 3 ? 3 : (c & 3) - 1)], acc[c], 0, 0, 0);
                // -- fragments of the next k-step: eight 16-byte LDS reads
                if (DBG & 2) {}
                else if (c >= 8 && c < 12) read_v(Vk, nset, c - 8);
                else if (c >= 12) read_a(Ak, nset, c - 12);
                // -- staging: k-step p = 0, 1 parks piece p of stage st+1 and re-fetches it for stage st+2
                if (ks < 2 && !(DBG & 1)) {
                    const int p = ks;
                    const bool bwork = parker && p < NIT;
                    if (c < 4) { if (!(DBG & 32)) park_a(wbuf, 2 * c + p, areg); }
                    else if (c == 4) { if (bwork && !(DBG & 8)) park_b_h(p, braw, pk0, pk12, pk3); }
                    else if (c < 9) { if (bwork && !(DBG & 8)) park_b_w(wbuf, p, c - 5); }
                    else if (c == 9) { if (p == 0) { fetch_begin(); nk0 = mk0; nk12 = mk12; nk3 = mk3; } }
                    else if (c < 14) {
                        if (bwork && !(DBG & 16)) fetch_b_row(p, c - 10, braw);
                        if (c >= 12 && !(DBG & 64)) fetch_a(2 * (c - 12) + p, areg);
                    }
                    else { if (!(DBG & 64)) fetch_a(2 * (c - 12) + p, areg); if (c == 15 && p == 1) { pk0 = nk0; pk12 = nk12; pk3 = nk3; } }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ks == 2 && !(DBG & 4)) __syncthreads();
        }
        rbuf = wbuf;
    }

    if (DBG & 128) {      // experiment: no epilogue (keeps the accumulators alive through one store)
        float sdbg = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) sdbg += acc[c][0];
        if (sdbg == 12345.678f) a.y[0] = sdbg;
        return;
    }
    // ---- epilogue: output transform Y = A^T M A on the accumulators, residual, GroupNorm partial sums.
    // Every instruction here is exposed (nothing else runs on the CU), so the common case -- all 64 channels and all 64
    // tiles of the workgroup exist, 8-byte stores allowed, no residual -- is straight-line code: per channel row 16
    // accumulator reads, 12 packed adds (the H stage on the component pairs (xi 0, 3) and (1, 2), which hands the W stage
    // its operands already paired; its results are the two adjacent outputs of a row), two 8-byte stores from a scalar
    // channel base stepped by the channel stride, 4 packed ops for the GroupNorm sums.
    const bool v2 = a.vec2;
    const bool gn = a.gn_part != nullptr;
    double gv[8];                                           // 8-row block g4: gv[2 g4] = sum, gv[2 g4 + 1] = sum of squares
#pragma unroll
    for (int i = 0; i < 8; ++i) gv[i] = 0.0;
    {
        const int cob = m0 + wm * 32;                       // wave-uniform; this lane's rows: cob + 4 lh + (rr & 3) + 8 (rr >> 2)
        const int n = tile0 + wn * 32 + l31;
        const int rp = n >> (LGW - 1), tw = n & (TW - 1);
        const bool pok = rp < RPtot;
        int ob = 0, od = 0, hp = 0;
        if (pok) split_rp(rp, ob, od, hp);
        // lane offsets in bytes (host check: the y / residual spans stay below 2^32 bytes), the 4 lh rows folded in
        const uint32_t yoff = (uint32_t)(ob * d.ys[0] + od * d.ys[2] + (2 * hp) * d.ys[3] + (2 * tw) * d.ys[4] + (4 * lh) * d.ys[1]) * 4u;
        const uint32_t roff = a.res ? (uint32_t)(ob * d.rs[0] + od * d.rs[2] + (2 * hp) * d.rs[3] + (2 * tw) * d.rs[4] + (4 * lh) * d.rs[1]) * 4u : 0u;
        const int64_t ycs = d.ys[1], rcs = d.rs[1];
        const uint32_t yrow = (uint32_t)d.ys[3] * 4u, rrow = (uint32_t)d.rs[3] * 4u, ycol = (uint32_t)d.ys[4] * 4u, rcol = (uint32_t)d.rs[4] * 4u;
        const bool full = m0 + BM <= d.Cout;                // all 64 rows of the workgroup exist
        typedef __attribute__((address_space(1))) char* gwchar_p;
        typedef __attribute__((address_space(1))) float* gwfloat_p;
        typedef __attribute__((address_space(1))) nfloat2* gwfloat2_p;
        if (full && v2 && !a.res && rp0 + RP <= RPtot) {
            gwchar_p yb = (gwchar_p)(__attribute__((address_space(1))) void*)uniform_ptr(a.y + (int64_t)cob * ycs);
            const int64_t rstep = ycs * 4, gstep = ycs * 20;                // bytes: the next row, the first row of the next block
            const uint32_t yoff1 = yoff + yrow;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                w2f2 bs2 = {0.f, 0.f}, bq2 = {0.f, 0.f};
#pragma unroll
                for (int r3 = 0; r3 < 4; ++r3) {
                    const int rr = g4 * 4 + r3;
                    w2f2 pa[4], pb[4];                      // per j: (M[j][0], M[j][3]), (M[j][1], M[j][2])
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        pa[j] = w2f2{acc[4 * j][rr], acc[4 * j + 3][rr]};
                        pb[j] = w2f2{acc[4 * j + 1][rr], acc[4 * j + 2][rr]};
                    }
                    const w2f2 t0a = pk_add2(pk_add2(pa[0], pa[1]), pa[2]), t1a = pk_sub2(pk_sub2(pa[1], pa[2]), pa[3]);
                    const w2f2 t0b = pk_add2(pk_add2(pb[0], pb[1]), pb[2]), t1b = pk_sub2(pk_sub2(pb[1], pb[2]), pb[3]);
                    const w2f2 z0 = pk_addsub(pk_sumdiff_fwd(t0b), t0a);    // (y00, y01)
                    const w2f2 z1 = pk_addsub(pk_sumdiff_fwd(t1b), t1a);    // (y10, y11)
                    *(gwfloat2_p)(yb + yoff) = nfloat2{z0.x, z0.y};
                    *(gwfloat2_p)(yb + yoff1) = nfloat2{z1.x, z1.y};
                    bs2 = pk_add2(bs2, pk_add2(z0, z1));
                    bq2 = pk_sqacc(z1, pk_sqacc(z0, bq2));
                    yb += r3 < 3 ? rstep : gstep;
                }
                gv[2 * g4] = (double)(bs2.x + bs2.y);
                gv[2 * g4 + 1] = (double)(bq2.x + bq2.y);
            }
        } else {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {                // 8-row blocks: rows cob + 8 g4 + 4 lh + (0..3)
                float bsx = 0.0f, bsy = 0.0f, bqx = 0.0f, bqy = 0.0f;      // the two halves of the packed path's sums
#pragma unroll
                for (int r3 = 0; r3 < 4; ++r3) {
                    const int rr = g4 * 4 + r3;
                    const int cou = cob + 8 * g4 + r3;      // wave-uniform part of the row
                    const bool rok = full || (cou + 4 * lh) < d.Cout;
                    const int coc = full ? cou : (cou < d.Cout - 4 ? cou : d.Cout - 8);       // clamped: in-bounds addresses for the tail
                    float t0[4], t1[4];
#pragma unroll
                    for (int xi = 0; xi < 4; ++xi) {
                        const float M0 = acc[0 + xi][rr], M1 = acc[4 + xi][rr], M2 = acc[8 + xi][rr], M3 = acc[12 + xi][rr];
                        t0[xi] = (M0 + M1) + M2;
                        t1[xi] = (M1 - M2) - M3;
                    }
                    // (the same association as the packed path above: a sample must come out bit-identical whichever path
                    // the workgroup that holds it takes -- that depends on the batch it is launched with)
                    float y00 = (t0[1] + t0[2]) + t0[0];
                    float y01 = (t0[1] - t0[2]) - t0[3];
                    float y10 = (t1[1] + t1[2]) + t1[0];
                    float y11 = (t1[1] - t1[2]) - t1[3];
                    if (a.res) {
                        const gchar_p rb = (gchar_p)uniform_ptr(a.res + (int64_t)coc * rcs);
                        if (v2) {
                            const nfloat2 u0 = *(gfloat2_p)(rb + roff), u1 = *(gfloat2_p)(rb + roff + rrow);
                            y00 += u0.x; y01 += u0.y; y10 += u1.x; y11 += u1.y;
                        } else {
                            y00 += *(gfloat_p)(rb + roff); y01 += *(gfloat_p)(rb + roff + rcol);
                            y10 += *(gfloat_p)(rb + roff + rrow); y11 += *(gfloat_p)(rb + roff + rrow + rcol);
                        }
                    }
                    if (pok && rok) {
                        const gwchar_p yb = (gwchar_p)(__attribute__((address_space(1))) void*)uniform_ptr(a.y + (int64_t)coc * ycs);
                        if (v2) {
                            *(gwfloat2_p)(yb + yoff) = nfloat2{y00, y01};
                            *(gwfloat2_p)(yb + yoff + yrow) = nfloat2{y10, y11};
                        } else {
                            *(gwfloat_p)(yb + yoff) = y00; *(gwfloat_p)(yb + yoff + ycol) = y01;
                            *(gwfloat_p)(yb + yoff + yrow) = y10; *(gwfloat_p)(yb + yoff + yrow + ycol) = y11;
                        }
                        // GroupNorm sums: the 2x2 tile and the 4 rows of the block in fp32 (1e-7 relative on a 16-element partial
                        // sum), fp64 from there on -- independent of the batch a trajectory is launched with (the tile grid cuts
                        // every sample alike)
                        bsx += y00 + y10; bsy += y01 + y11;
                        bqx = fmaf(y10, y10, fmaf(y00, y00, bqx)); bqy = fmaf(y11, y11, fmaf(y01, y01, bqy));
                    }
                }
                gv[2 * g4] = (double)(bsx + bsy);
                gv[2 * g4 + 1] = (double)(bqx + bqy);
            }
        }
    }
    if (gn) {
        // scr[wave][4][2] behind the stage buffers (a wave may get here while another still reads its last fragments);
        // 8-row block t of the workgroup's 64 rows = (wm = t / 4, k = t % 4)
        double* scr = reinterpret_cast<double*>(ldsw + W2_NBUF * (W2_ASZ + W2_BSZ));
        const double tot = wave_sum8(gv, lane);
        if ((lane & 7) == 0) scr[wave * 8 + (lane >> 3)] = tot;
        __syncthreads();
        const int ngl = a.gn_cpg >= BM ? 1 : BM / a.gn_cpg;       // groups inside this workgroup's rows
        if (tid < ngl) {
            const int r0 = a.gn_cpg >= BM ? 0 : tid * a.gn_cpg, r1 = a.gn_cpg >= BM ? BM : r0 + a.gn_cpg;   // local rows
            double sum = 0.0, sq = 0.0;
            for (int blk = r0 / 8; blk < r1 / 8; ++blk) {
                const int wmi = blk >> 2, k = blk & 3;
                for (int wni = 0; wni < 2; ++wni) {
                    sum += scr[((wmi * 2 + wni) * 4 + k) * 2];
                    sq += scr[((wmi * 2 + wni) * 4 + k) * 2 + 1];
                }
            }
            if (m0 + r0 < d.Cout) {
                const int g = (m0 + r0) / a.gn_cpg;
                const int p0 = tile0 * 4;                          // first output position of the workgroup (whole row pairs)
                const int b = p0 / a.gn_S, ntl = (p0 - b * a.gn_S) / (W2_TILES * 4);
                const int idx = a.gn_cpg >= BM ? ntl * (a.gn_cpg / BM) + (m0 - g * a.gn_cpg) / BM : ntl;
                double* pp = a.gn_part + (((int64_t)b * a.gn_G + g) * a.gn_nparts + idx) * 2;
                pp[0] = sum; pp[1] = sq;
            }
        }
    }
}

inline int64_t span5(const int64_t* st, int b, int c, int dd, int h, int w) {
    return (int64_t)(b - 1) * st[0] + (int64_t)(c - 1) * st[1] + (int64_t)(dd - 1) * st[2] + (int64_t)(h - 1) * st[3] + (int64_t)(w - 1) * st[4];
}

// coverage of the F(2x2,3x3) kernel (precision 3)
bool wg2_ok(const SdcConvDesc& d, bool small, bool rowhalo) {
    const int64_t rptot = (int64_t)d.B * d.oD * (d.oH / 2);
    return d.precision >= 3 && rowhalo && small && d.kH == 3 && d.kW == 3 && (d.kD == 1 || d.kD == 3) &&
           d.sD == 1 && d.sH == 1 && d.sW == 1 && d.uD == 1 && d.uH == 1 && d.uW == 1 && d.up_mode == 0 &&
           d.pH == 1 && d.pW == 1 && d.pD == d.kD / 2 && d.oH == d.iH && d.oW == d.iW && d.oD == d.iD &&
           (d.oW == 16 || d.oW == 32 || d.oW == 64 || d.oW == 128) && d.oH % 2 == 0 &&
           d.Cin0 % W2_SK == 0 && d.Cin1 % W2_SK == 0 && d.Cout % 4 == 0 && d.Cout > 32 && rptot < (1 << 20) &&
           ((int64_t)d.kD * 9 * (d.Cin0 + d.Cin1) * d.Cout) % 4 == 0 &&
           // input rows contiguous (the row taps are instruction immediates)
           d.x0s[4] == 1 && d.x0s[3] == d.iW && (d.Cin1 == 0 || (d.x1s[4] == 1 && d.x1s[3] == d.iW)) &&
           // the park lanes add their k row (up to 3 channel strides at W = 128, 1 otherwise) to the 32-bit byte offset of the
           // (batch, depth, row, column) part, and the depth-tap shift is formed in 32 bits
           span5(d.x0s, d.B, 1, d.iD, d.iH, d.iW) + 3 * d.x0s[1] < (1ll << 30) && d.x0s[2] < (1ll << 29) &&
           (d.Cin1 == 0 || (span5(d.x1s, d.B, 1, d.iD, d.iH, d.iW) + 3 * d.x1s[1] < (1ll << 30) && d.x1s[2] < (1ll << 29))) &&
           // the epilogue addresses y / the residual with 32-bit byte offsets from per-channel scalar bases
           span5(d.ys, d.B, 8, d.oD, d.oH, d.oW) < (1ll << 30) && span5(d.rs, d.B, 8, d.oD, d.oH, d.oW) < (1ll << 30) &&
           // ... and read with 8 / 16-byte vector loads
           d.x0s[0] % 4 == 0 && d.x0s[1] % 4 == 0 && d.x0s[2] % 4 == 0 &&
           (d.Cin1 == 0 || (d.x1s[0] % 4 == 0 && d.x1s[1] % 4 == 0 && d.x1s[2] % 4 == 0));
}

int launch_wg2(const ConvArgs& a, hipStream_t s) {
    const SdcConvDesc& d = a.d;
    const int64_t tiles = (int64_t)d.B * d.oD * (d.oH / 2) * (d.oW / 2);
    const int MT = (d.Cout + W2_BM - 1) / W2_BM;
    dim3 grid((unsigned)(((tiles + W2_TILES - 1) / W2_TILES) * MT));
    const size_t lds = (size_t)W2_NBUF * (W2_ASZ + W2_BSZ) * sizeof(float) + 4 * 8 * sizeof(double);     // stage buffers + GroupNorm scratch
#define W2_LAUNCH(OWV, D)                                                                                                        \
    do {                                                                                                                         \
        static std::atomic<uint64_t> attr{0};                                                                                    \
        SDC_LDS_OPTIN(attr, (conv_wg2_kernel<OWV, D>), 160 * 1024, "sdc_conv[winograd 2x2]");                                    \
        hipLaunchKernelGGL((conv_wg2_kernel<OWV, D>), grid, dim3(256), lds, s, a);                                               \
    } while (0)
#ifdef SDC_KERNEL_EXPERIMENTS
    // parts of the loop switched off (WRONG RESULTS): never compiled into the shipping library
    static const int dbg = exp_env("SDC_WG2_DBG");
    if (d.oW == 64 && dbg) {
        switch (dbg) {
            case 1: W2_LAUNCH(64, 1); break; case 2: W2_LAUNCH(64, 2); break; case 3: W2_LAUNCH(64, 3); break;
            case 128: W2_LAUNCH(64, 128); break; default: W2_LAUNCH(64, 3 + 128); break;
        }
        return SDC_OK;
    }
#endif
    if (d.oW == 16) W2_LAUNCH(16, 0);
    else if (d.oW == 32) W2_LAUNCH(32, 0);
    else if (d.oW == 64) W2_LAUNCH(64, 0);
    else W2_LAUNCH(128, 0);
    return SDC_OK;
}

// ------------------------------------------------------------------------------------------------
// precision = 4: fp32 Winograd F(2x2x2, 3x3x3) for the 3x3x3 stride-1 convs -- the F(2x2,3x3) kernel above with the same
// transform applied along the depth as well: 64 products per 8 outputs instead of 216, i.e. 2/3 of the MFMA work of the
// (H, W)-only form and 8/27 of the direct form.  The caller stores U3[jd][ci][co][j*4+xi] = sum G[jd][kd] G[j][kh] G[xi][kw] w
// behind the F(2x2,3x3) taps.
//   * A workgroup owns 64 output channels x 64 (h, w) tiles x ONE PAIR of output planes (2 dp, 2 dp + 1) and walks the four
//     depth components jd one after the other with the 16 (j, xi) accumulators of the kernel above.  The K loop of pass jd
//     runs over the input channels only; its B operand is the (H, W) transform of the depth combination
//        jd 0: s[-1] - s[1],   jd 1: s[0] + s[1],   jd 2: s[1] - s[0],   jd 3: s[0] - s[2]      (s[i] = input plane 2 dp + i)
//     which costs two row loads and one packed multiply-add more per row than the plain slice (the signs, the zero planes
//     past the volume and the zero rows above / below the image are factors of the same fused ops).
//   * At the end of pass jd the (H, W) output transform m_jd of the accumulators is folded into the two output planes,
//        y[2 dp] = m0 + m1 + m2,      y[2 dp + 1] = m1 - m2 - m3,
//     by read-modify-write of y by the lane that owns the element (a lane's own loads and stores of one address stay
//     ordered).  The passes run in the order m1, m2, m0, m3: m1 is parked in plane 1, the second fold reads it once and writes
//     m1 + m2 to plane 0 and m1 - m2 to plane 1, the third finishes plane 0 with m0, the fourth plane 1 with m3 -- three plane
//     read-backs and five plane stores (the natural order needs four and six).  The partial planes come back from L2 / MALL;
//     the GroupNorm sums are taken from the finished values.  The fetch pipeline of the next pass (two stages in flight) runs through
//     the fold, so only its own instructions are exposed.
// Coverage: what the F(2x2,3x3) kernel takes, and kD = 3, even depth, W in {16, 32, 64}, Cout % 64 == 0, the row pairs of
// a workgroup inside one plane, 8-byte aligned output rows, no fused residual (rs all zero).
__device__ __forceinline__ uint64_t lo64(float k) { return (uint64_t)__builtin_bit_cast(uint32_t, k); }
// a * k, a * k + c with a wave-uniform factor k (both halves)
__device__ __forceinline__ w2f2 pks_mul(w2f2 a, float k) { w2f2 r; asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "s"(lo64(k))); return r; }
__device__ __forceinline__ w2f2 pks_fma(w2f2 a, float k, w2f2 c) { w2f2 r; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "s"(lo64(k)), "v"(c)); return r; }
__device__ __forceinline__ w2f2 pk_fma2(w2f2 a, w2f2 b, w2f2 c) { w2f2 r; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

template <int OW, int DBG>
__global__ __launch_bounds__(256) void conv_wg3_kernel(const ConvArgs a) {
    constexpr int SK = W2_SK, BM = W2_BM;
    constexpr int TW = OW / 2, RP = W2_TILES / TW;
    constexpr int LGW = OW == 64 ? 6 : (OW == 32 ? 5 : 4);
    constexpr int CPL = OW == 16 ? 2 : OW / 16;          // adjacent columns per lane
    constexpr int SH = OW == 16 ? 2 : 1;                 // DPP lane distance of the neighbouring column group
    constexpr int LPK = 128 / CPL;                       // lanes per staged channel (k row)
    constexpr int NIT = CPL == 2 ? 2 : 1;                // park items per thread and stage
    constexpr int KPW = 2;                               // k rows per (parking) wave
    constexpr int TPL = CPL / 2;                         // tiles per lane and row pair
    typedef float nfloat4 __attribute__((ext_vector_type(4)));
    typedef float nfloat2 __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(1))) char* gchar_p;
    typedef const __attribute__((address_space(1))) nfloat4* gfloat4_p;
    typedef const __attribute__((address_space(1))) nfloat2* gfloat2_p;
    typedef __attribute__((address_space(1))) char* gwchar_p;
    typedef __attribute__((address_space(1))) nfloat2* gwfloat2_p;
    extern __shared__ __attribute__((aligned(16))) float ldsw[];
    float* const As = ldsw;                          // [2][SK][BM][16]  (chunk q of row m at slot q ^ ((m >> 2) & 3))
    float* const Vs = ldsw + W2_NBUF * W2_ASZ;       // [2][SK][4 j][64 tiles][4 xi]

    const SdcConvDesc& d = a.d;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = SDC_UNIFORM(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int MT = d.Cout / BM;
    const int lb = xcd_tile(blockIdx.x, gridDim.x);
    const int m0 = (lb % MT) * BM;
    const int tile0 = (lb / MT) * W2_TILES;
    const int H2 = d.oH >> 1, D2 = d.oD >> 1;
    // the workgroup's RP row pairs lie in one plane pair (host check: H2 % RP == 0): (sample ob, planes od, od + 1, first row pair hp0)
    int ob, od, hp0;
    {
        const int rp0 = tile0 >> (LGW - 1);          // < 2^20 (host check): the float quotients are exact
        const int q = (int)(((float)rp0 + 0.5f) * (1.0f / (float)H2));
        const int b = (int)(((float)q + 0.5f) * (1.0f / (float)D2));
        hp0 = SDC_UNIFORM(rp0 - q * H2);
        ob = SDC_UNIFORM(b);
        od = SDC_UNIFORM(2 * (q - b * D2));
    }
    const bool two = d.Cin1 > 0;

    // ---- park geometry of this thread: k row inside the stage, row pair, first column (CPL columns)
    const int ksub = CPL == 4 ? (lane >> 5) : 0;                    // k row inside the wave's pair (CPL = 2: the item)
    const int lik = lane & (LPK - 1);
    int pr, pc0;
    if (OW == 16) { pr = 2 * (lik >> 4) + (lik & 1); pc0 = 2 * ((lik & 15) >> 1); }
    else { pr = lik >> 4; pc0 = (lik & 15) * CPL; }
    // lane part of the input addresses (bytes): k row, row 2 hp, column; the rows above / below the pair are fetched from a
    // clamped (valid) row and multiplied by 0 where they fall outside the image.  Sample, plane and channel are scalar.
    const int hp = hp0 + pr;
    const bool up_ok = hp > 0, dn_ok = 2 * hp + 2 < d.iH;
    const w2f2 m0p = {up_ok ? 1.0f : 0.0f, up_ok ? 1.0f : 0.0f}, m3p = {dn_ok ? 1.0f : 0.0f, dn_ok ? 1.0f : 0.0f};
    const uint32_t rsel0 = up_ok ? OW * 4 : 0, rsel3 = dn_ok ? 2 * OW * 4 : 0;
    const uint32_t vp0 = (uint32_t)(ksub * d.x0s[1] + (2 * hp) * OW + pc0) * 4u;
    const uint32_t vp1 = two ? (uint32_t)(ksub * d.x1s[1] + (2 * hp) * OW + pc0) * 4u : 0u;
    const int vpark = ((wave * KPW + ksub) * 4 * W2_TILES + pr * TW + (pc0 >> 1)) * 4;          // floats; + j * 256, + item * 1024
    const int boff = (lh * 4 * W2_TILES + wn * 32 + l31) * 4;
    const int arow = wm * 32 + l31;
    int aoff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) aoff[q] = (lh * BM + arow) * 16 + 4 * (q ^ ((arow >> 2) & 3));
    const int pm = tid >> 2, pq = tid & 3;
    const int apark = pm * 16 + 4 * (pq ^ ((pm >> 2) & 3));                        // + i * BM * 16 floats
    uint32_t a_voff[4];                                  // k rows 0..3 of a stage; rows 4..7 from a second scalar base
#pragma unroll
    for (int i = 0; i < 4; ++i) a_voff[i] = (uint32_t)(((int64_t)i * d.Cout + m0 + pm) * 16 + 4 * pq) * 4u;

    w2f2 braw[NIT][2][4][TPL];    // [item][depth slice a / b][source row j][column pair]
    nfloat4 areg[8];
    int s_jd = 0, s_ci = 0;
    const int64_t xs1_0 = d.x0s[1], xs1_1 = d.x1s[1];
    const int cin0 = d.Cin0, cin = a.Cin, coutn = d.Cout;
    const float* const wg3p = a.wg2;
    // sample and plane folded into the bases
    const float* const x0p = a.x0 + (int64_t)ob * d.x0s[0] + (int64_t)od * d.x0s[2];
    const float* const x1p = two ? a.x1 + (int64_t)ob * d.x1s[0] + (int64_t)od * d.x1s[2] : x0p;
    const int64_t xs2_0 = d.x0s[2], xs2_1 = two ? d.x1s[2] : d.x0s[2];
    const uint32_t lo_u = od > 0, hi_u = od + 2 < d.iD;          // planes od - 1 / od + 2 exist
    gfloat_p f_w = uniform_ptr(wg3p), f_w4 = f_w, f_xa = uniform_ptr(x0p), f_xb = f_xa;
    int64_t f_sc = 0;
    uint32_t voff = 0, voff0 = 0, voff3 = 0;
    float mka = 0.f, mkb = 0.f;
    auto fetch_begin = [&]() __attribute__((always_inline)) {
        const bool first = s_ci < cin0;
        f_sc = first ? xs1_0 : xs1_1;
        const int64_t xs2 = first ? xs2_0 : xs2_1;
        const int cbase = (first ? s_ci : s_ci - cin0) + wave * KPW;
        const float* bsel = (first ? x0p : x1p) + (int64_t)cbase * f_sc;
        // planes (relative to od) and signs of depth component s_jd; a plane outside the volume: plane od with factor 0
        // (integer arithmetic on the float bits: nested selects became branches, and a branch in this loop costs
        // conservative memory waits at its join)
        // pass s_jd (0..3) works on depth component 1, 2, 0, 3 (see the fold)
        const uint32_t j0 = s_jd == 2, j2 = s_jd == 1, j3 = s_jd == 3;
        const int jdc = s_jd + 1 - 3 * (int)j0 - (int)j3;
        const int da = -(int)(j0 & lo_u);                                    // -1 | 0 | 0 | 0
        const int db = 1 + (int)j3 * (hi_u ? 1 : -1);                        //  1 | 1 | 1 | 2 (0 past the volume)
        mka = __builtin_bit_cast(float, (0x3F800000u & ((j0 & (lo_u ^ 1u)) - 1u)) | (j2 << 31));       // lo_ok | 1 | -1 | 1
        mkb = __builtin_bit_cast(float, (0x3F800000u & ((j3 & (hi_u ^ 1u)) - 1u)) | ((j0 | (j3 & hi_u)) << 31));   // -1 | 1 | 1 | -hi_ok
        f_xa = uniform_ptr(bsel + da * xs2);
        f_xb = uniform_ptr(bsel + db * xs2);
        f_w = uniform_ptr(wg3p + ((int64_t)(jdc * cin + s_ci) * coutn) * 16);
        f_w4 = uniform_ptr(wg3p + ((int64_t)(jdc * cin + s_ci + 4) * coutn) * 16);
        voff = first ? vp0 : vp1;
        voff0 = voff - rsel0;
        voff3 = voff + rsel3;
        s_ci += SK;
        // (past the last stage the walk wraps to the first one: the extra fetches of the pipeline tail stay in bounds and are
        // never consumed)
        if (s_ci >= cin) { s_ci = 0; if (++s_jd == 4) s_jd = 0; }
    };
    // (the offset passes through an empty asm so that its zero-extension is not hoisted out of the loop as a 64-bit
    // register pair: the load then takes the scalar base + 32-bit lane offset form)
    auto fetch_a = [&](int i, nfloat4 (&ar)[8]) { uint32_t o = a_voff[i & 3]; asm volatile("" : "+v"(o)); ar[i] = *(gfloat4_p)((gchar_p)(i < 4 ? f_w : f_w4) + o); };
    auto fetch_b_row = [&](int it, int sl, int j, w2f2 (&br)[NIT][2][4][TPL]) __attribute__((always_inline)) {
        const gchar_p rb = (gchar_p)(sl ? f_xb : f_xa) + (CPL == 2 ? (int64_t)it * f_sc * 4 : 0);
        const gchar_p p = j == 0 ? rb + voff0 : (j == 3 ? rb + voff3 : (j == 1 ? rb + voff : rb + voff + OW * 4));
        if (CPL == 2) { const nfloat2 v = *(gfloat2_p)p; br[it][sl][j][0] = w2f2{v.x, v.y}; }
        else {
            const nfloat4 v = *(gfloat4_p)p;
            br[it][sl][j][0] = w2f2{v.x, v.y};
            br[it][sl][j][1] = w2f2{v.z, v.w};
        }
    };
    auto fetch_b = [&](int it, w2f2 (&br)[NIT][2][4][TPL]) {
#pragma unroll
        for (int r = 0; r < 8; ++r) fetch_b_row(it, r >> 2, r & 3, br);
    };
    auto park_a = [&](int buf, int i, const nfloat4 (&ar)[8]) {
        *reinterpret_cast<nfloat4*>(As + buf * W2_ASZ + i * (BM * 16) + apark) = ar[i];
    };
    // depth combination e_j = ca a_j + cb b_j of the two planes fused with the H transform (packed, per column pair), then
    // the W transform as in the kernel above
    w2f2 hrow[4][TPL];
    auto park_b_h = [&](int it, const w2f2 (&br)[NIT][2][4][TPL], float ca, float cb) __attribute__((always_inline)) {
        const w2f2 ka0 = pks_mul(m0p, ca), kb0 = pks_mul(m0p, cb), ka3 = pks_mul(m3p, ca), kb3 = pks_mul(m3p, cb);
#pragma unroll
        for (int t = 0; t < TPL; ++t) {
            const w2f2 e1 = pks_fma(br[it][1][1][t], cb, pks_mul(br[it][0][1][t], ca));
            const w2f2 e2 = pks_fma(br[it][1][2][t], cb, pks_mul(br[it][0][2][t], ca));
            hrow[0][t] = pk_fma2(br[it][1][0][t], kb0, pk_fms2(br[it][0][0][t], ka0, e2));       // e0 - e2
            hrow[1][t] = pk_add2(e1, e2);
            hrow[2][t] = pk_sub2(e2, e1);
            hrow[3][t] = pk_fnma2(br[it][1][3][t], kb3, pk_fnma2(br[it][0][3][t], ka3, e1));     // e1 - e3
        }
    };
    auto park_b_w = [&](int buf, int it, int j) __attribute__((always_inline)) {
        float* dst = Vs + buf * W2_BSZ + vpark + (CPL == 2 ? it * (4 * W2_TILES * 4) : 0) + j * (W2_TILES * 4);
#pragma unroll
        for (int t = 0; t < TPL; ++t) {
            const w2f2 cc = hrow[j][t];
            const w2f2 sd = pk_sumdiff(cc);
            const float v0 = t == 0 ? sub_prev<SH>(hrow[j][TPL - 1].y, cc.y) : vsub1(hrow[j][t - 1].y, cc.y);
            const float v3 = t == TPL - 1 ? sub_next<SH>(cc.x, hrow[j][0].x) : vsub1(cc.x, hrow[j][t + 1].x);
            nfloat4 v;
            v.x = sd.x; v.y = sd.y; v.z = v0; v.w = v3;
            *reinterpret_cast<nfloat4*>(dst + t * 4) = v;
        }
    };

    f32x16 acc[16];                                  // (started from zero by the first MFMAs of each pass)
    if (DBG & 4) {
#pragma unroll
        for (int c = 0; c < 16; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
    }
    // the bias rides on component (j, xi) = (1, 1) of depth component 1 (coefficient +1 in all 8 outputs): the start value of
    // that accumulator in the first pass.  Register r of a lane is channel m0 + 32 wm + 8 (r >> 2) + 4 lh + (r & 3).
    f32x16 zero16, biasv;
#pragma unroll
    for (int r = 0; r < 16; ++r) { zero16[r] = 0.0f; biasv[r] = a.bias ? a.bias[m0 + wm * 32 + 8 * (r >> 2) + 4 * lh + (r & 3)] : 0.0f; }

    const int S1 = a.Cin / SK;                       // stages per depth component
    nfloat4 fa[2][4];
    nfloat4 fv[2][4];
    auto read_a = [&](const float* Ak, int set, int q) { fa[set][q] = *reinterpret_cast<const nfloat4*>(Ak + aoff[q]); };
    auto read_v = [&](const float* Vk, int set, int j) { fv[set][j] = *reinterpret_cast<const nfloat4*>(Vk + boff + j * (W2_TILES * 4)); };
    float pka, pkb, nka = 0.f, nkb = 0.f;            // factors of the items still in registers
    {   // prologue: the fetches of the first two stages travel together; stage 0 is parked in buffer 0
        w2f2 braw0[NIT][2][4][TPL];
        nfloat4 areg0[8];
        fetch_begin();
        const float qa = mka, qb = mkb;
#pragma unroll
        for (int i = 0; i < 8; ++i) fetch_a(i, areg0);
#pragma unroll
        for (int it = 0; it < NIT; ++it) fetch_b(it, braw0);
        fetch_begin();
#pragma unroll
        for (int i = 0; i < 8; ++i) fetch_a(i, areg);
#pragma unroll
        for (int it = 0; it < NIT; ++it) fetch_b(it, braw);
        pka = mka; pkb = mkb;
#pragma unroll
        for (int i = 0; i < 8; ++i) park_a(0, i, areg0);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            park_b_h(it, braw0, qa, qb);
#pragma unroll
            for (int j = 0; j < 4; ++j) park_b_w(0, it, j);
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) read_a(As, 0, q);
#pragma unroll
    for (int j = 0; j < 4; ++j) read_v(Vs, 0, j);

    // ---- output geometry: this lane's tile n (row pair hp0 + n / TW of the plane pair, tile column n % TW), channel rows
    // cob + 8 (rr >> 2) + 4 lh + (rr & 3); lane offsets in bytes with the 4 lh rows folded in, channel bases scalar
    const int cob = m0 + wm * 32;
    const int nloc = wn * 32 + l31;
    const int ohp = hp0 + (nloc >> (LGW - 1)), otw = nloc & (TW - 1);
    const uint32_t yoff = (uint32_t)((2 * ohp) * d.ys[3] + (2 * otw) * d.ys[4] + (4 * lh) * d.ys[1]) * 4u;
    const uint32_t yoff1 = yoff + (uint32_t)d.ys[3] * 4u;
    const bool gn = a.gn_part != nullptr;
    // GroupNorm sums of this lane: 8-row block g4 -> (sum, sum of squares) over the 4 rows x 2x2 tile in fp32, per finished
    // plane; the plane-0 values wait in LDS (behind the stage buffers and the reduction scratch) for those of plane 1
    float gp[8];
    float* const gstash = ldsw + W2_NBUF * (W2_ASZ + W2_BSZ) + 64 + tid * 8;

    // fold of pass p (compile-time after unrolling) into the plane pair.  Pass order: depth components 1, 2, 0, 3 --
    //   p 0 (m1): plane 1 <- m1                       (scratch: nothing read)
    //   p 1 (m2): plane 0 <- m1 + m2,  plane 1 <- m1 - m2      (one plane read, two written)
    //   p 2 (m0): plane 0 <- (m1 + m2) + m0  finished
    //   p 3 (m3): plane 1 <- (m1 - m2) - m3  finished
    // three plane read-backs and five plane stores per workgroup (the order 0, 1, 2, 3 needs four and six).
    auto fold = [&](const int p) __attribute__((always_inline)) {
        const bool t0 = p == 1 || p == 2, t1 = p != 2;           // planes written
        const bool ld0 = p == 2 && !(DBG & 2), ld1 = (p == 1 || p == 3) && !(DBG & 2);      // partial plane read back
        const bool fin0 = p == 2, fin1 = p == 3;                 // plane finished by this pass
        const bool always = d.Cout > 0;
        const int LEAD = (DBG & 8) ? 4 : 2;                      // blocks of partial sums requested ahead of their use
        const int64_t ycs4 = d.ys[1] * 4;                        // bytes per channel
        // plane od (+1) of sample ob, channel cob: scalar cursors, one for the reads and one for the writes of each plane
        float* const y0p = a.y + (int64_t)ob * d.ys[0] + (int64_t)od * d.ys[2] + (int64_t)cob * d.ys[1];
        gwchar_p s0 = (gwchar_p)(__attribute__((address_space(1))) void*)uniform_ptr(y0p);
        gwchar_p s1 = (gwchar_p)(__attribute__((address_space(1))) void*)uniform_ptr(y0p + d.ys[2]);
        gchar_p l0 = (gchar_p)s0, l1 = (gchar_p)s1;
        w2f2 P0[4][4][2], P1[4][4][2];                           // [block][row of the block][row of the tile]
        auto load_block = [&](int g4) __attribute__((always_inline)) {
#pragma unroll
            for (int r3 = 0; r3 < 4; ++r3) {
                if (ld0) { P0[g4][r3][0] = *(gfloat2_p)(l0 + yoff); P0[g4][r3][1] = *(gfloat2_p)(l0 + yoff1); l0 += r3 < 3 ? ycs4 : 5 * ycs4; }
                if (ld1) { P1[g4][r3][0] = *(gfloat2_p)(l1 + yoff); P1[g4][r3][1] = *(gfloat2_p)(l1 + yoff1); l1 += r3 < 3 ? ycs4 : 5 * ycs4; }
            }
        };
        if (ld0 || ld1) {
#pragma unroll
            for (int g = 0; g < LEAD; ++g) load_block(g);
        }
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            if (g4 + LEAD < 4 && (ld0 || ld1)) load_block(g4 + LEAD);
            // Each 8-row block is its own basic block (the condition always holds): in one block the compiler hoists all 256
            // accumulator reads to the top of the fold, beside the fetch pipeline's registers, and spills.
            if (!always) continue;
            w2f2 bs2 = {0.f, 0.f}, bq2 = {0.f, 0.f};
#pragma unroll
            for (int r3 = 0; r3 < 4; ++r3) {
                const int rr = g4 * 4 + r3;
                w2f2 pa[4], pb[4];                               // per j: (M[j][0], M[j][3]), (M[j][1], M[j][2])
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pa[j] = w2f2{acc[4 * j][rr], acc[4 * j + 3][rr]};
                    pb[j] = w2f2{acc[4 * j + 1][rr], acc[4 * j + 2][rr]};
                }
                const w2f2 t0a = pk_add2(pk_add2(pa[0], pa[1]), pa[2]), t1a = pk_sub2(pk_sub2(pa[1], pa[2]), pa[3]);
                const w2f2 t0b = pk_add2(pk_add2(pb[0], pb[1]), pb[2]), t1b = pk_sub2(pk_sub2(pb[1], pb[2]), pb[3]);
                const w2f2 z0 = pk_addsub(pk_sumdiff_fwd(t0b), t0a);        // rows 2 hp, 2 hp + 1 of the pass's m
                const w2f2 z1 = pk_addsub(pk_sumdiff_fwd(t1b), t1a);
                if (t0) {
                    // p 1: m1 (read from plane 1) + m2;  p 2: (m1 + m2) + m0
                    const bool have = p == 1 ? ld1 : ld0;
                    const w2f2 b0 = p == 1 ? P1[g4][r3][0] : P0[g4][r3][0], b1 = p == 1 ? P1[g4][r3][1] : P0[g4][r3][1];
                    const w2f2 u0 = have ? pk_add2(b0, z0) : z0, u1 = have ? pk_add2(b1, z1) : z1;
                    *(gwfloat2_p)(s0 + yoff) = nfloat2{u0.x, u0.y};
                    *(gwfloat2_p)(s0 + yoff1) = nfloat2{u1.x, u1.y};
                    s0 += r3 < 3 ? ycs4 : 5 * ycs4;
                    if (fin0) { bs2 = pk_add2(bs2, pk_add2(u0, u1)); bq2 = pk_sqacc(u1, pk_sqacc(u0, bq2)); }
                }
                if (t1) {
                    // p 0: m1;  p 1: m1 - m2;  p 3: (m1 - m2) - m3
                    const w2f2 u0 = ld1 ? pk_sub2(P1[g4][r3][0], z0) : z0, u1 = ld1 ? pk_sub2(P1[g4][r3][1], z1) : z1;
                    *(gwfloat2_p)(s1 + yoff) = nfloat2{u0.x, u0.y};
                    *(gwfloat2_p)(s1 + yoff1) = nfloat2{u1.x, u1.y};
                    s1 += r3 < 3 ? ycs4 : 5 * ycs4;
                    if (fin1) { bs2 = pk_add2(bs2, pk_add2(u0, u1)); bq2 = pk_sqacc(u1, pk_sqacc(u0, bq2)); }
                }
            }
            // GroupNorm sums: the 2x2 tile and the 4 rows of the block in fp32, fp64 from there on
            if (fin0 || fin1) { gp[2 * g4] = bs2.x + bs2.y; gp[2 * g4 + 1] = bq2.x + bq2.y; }
        }
        if (fin0 && gn) {
            *reinterpret_cast<nfloat4*>(gstash) = nfloat4{gp[0], gp[1], gp[2], gp[3]};
            *reinterpret_cast<nfloat4*>(gstash + 4) = nfloat4{gp[4], gp[5], gp[6], gp[7]};
        }
    };
    // Main loop, per depth component: stage st computes from buffer st & 1 and parks stage st+1 in the other one during its
    // k-steps 0-1, re-using each register piece for the fetch of stage st+2 as soon as it is parked; one barrier at the end of
    // k-step 2; k-step 3 reads the first fragments of stage st+1 (slots as in the kernel above, 8 row loads instead of 4).
    int rbuf = 0;
    // The first stage of a pass is a copy of the stage body whose first k-step starts the accumulators from zero (from the
    // bias for component (1, 1) of the first pass = depth component 1) in the MFMA itself: 256 register writes per pass less in the fold.
    auto stage = [&](auto FIRST, const int jd) __attribute__((always_inline)) {
        constexpr bool first = decltype(FIRST)::value && !(DBG & 4);     // (the no-fold experiment lets the passes accumulate on)
        {
            const int wbuf = rbuf ^ 1;
            const float* Ab = As + rbuf * W2_ASZ;
            const float* Vb = Vs + rbuf * W2_BSZ;
            const float* An = As + wbuf * W2_ASZ;
            const float* Vn = Vs + wbuf * W2_BSZ;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int set = ks & 1, nset = set ^ 1;
                const float* Ak = ks < 3 ? Ab + (2 * (ks + 1)) * (BM * 16) : An;
                const float* Vk = ks < 3 ? Vb + (2 * (ks + 1)) * (4 * W2_TILES * 4) : Vn;
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const f32x16 cstart = (first && ks == 0) ? ((c == 5 && jd == 0) ? biasv : zero16) : acc[c];
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][c >> 2][c & 3],
                                                                  fv[set][c >> 2][(c & 3) == 0 ? 2 : ((c & 3) == 3 ? 3 : (c & 3) - 1)], cstart, 0, 0, 0);
                    if (c >= 8 && c < 12) read_v(Vk, nset, c - 8);
                    else if (c >= 12) read_a(Ak, nset, c - 12);
                    if (ks < 2) {
                        const int p = ks;
                        const bool bwork = p < NIT;
                        if (c < 4) park_a(wbuf, 2 * c + p, areg);
                        else if (c == 4) { if (bwork) park_b_h(p, braw, pka, pkb); }
                        else if (c < 8) { if (bwork) park_b_w(wbuf, p, c - 5); if (c == 7 && p == 0) { fetch_begin(); nka = mka; nkb = mkb; } }
                        else {
                            if (c == 8 && bwork) park_b_w(wbuf, p, 3);
                            if (bwork) fetch_b_row(p, (c - 8) >> 2, (c - 8) & 3, braw);
                            if (c >= 12) fetch_a(2 * (c - 12) + p, areg);
                            if (c == 15 && p == 1) { pka = nka; pkb = nkb; }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (ks == 2) __syncthreads();
            }
            rbuf = wbuf;
        }
    };
    auto run_pass = [&](auto JD) __attribute__((always_inline)) {
        constexpr int jd = decltype(JD)::value;
        stage(std::true_type{}, jd);
        for (int st = 1; st < S1; ++st) stage(std::false_type{}, jd);
        if (!(DBG & 4)) {
            fold(jd);
            // (the first fragments of the next stage, read again: carried across the fold they cost 32 registers there)
#pragma unroll
            for (int q = 0; q < 4; ++q) read_a(As + rbuf * W2_ASZ, 0, q);
#pragma unroll
            for (int j = 0; j < 4; ++j) read_v(Vs + rbuf * W2_BSZ, 0, j);
        }
    };
    run_pass(std::integral_constant<int, 0>{});
    run_pass(std::integral_constant<int, 1>{});
    run_pass(std::integral_constant<int, 2>{});
    run_pass(std::integral_constant<int, 3>{});

    if (DBG & 4) {      // experiment: no folds (keeps the accumulators alive through one store)
        float sdbg = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) sdbg += acc[c][0];
        if (sdbg == 12345.678f) a.y[0] = sdbg;
        return;
    }
    if (gn) {
        double* scr = reinterpret_cast<double*>(ldsw + W2_NBUF * (W2_ASZ + W2_BSZ));
        double gv[8];                                // fp64 from here on
        {
            const nfloat4 s0 = *reinterpret_cast<const nfloat4*>(gstash), s1 = *reinterpret_cast<const nfloat4*>(gstash + 4);
            const float g0[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) gv[i] = (double)g0[i] + (double)gp[i];
        }
        const double tot = wave_sum8(gv, lane);
        if ((lane & 7) == 0) scr[wave * 8 + (lane >> 3)] = tot;
        __syncthreads();
        const int ngl = a.gn_cpg >= BM ? 1 : BM / a.gn_cpg;       // groups inside this workgroup's rows
        if (tid < ngl) {
            const int r0 = a.gn_cpg >= BM ? 0 : tid * a.gn_cpg, r1 = a.gn_cpg >= BM ? BM : r0 + a.gn_cpg;   // local rows
            double sum = 0.0, sq = 0.0;
            for (int blk = r0 / 8; blk < r1 / 8; ++blk) {
                const int wmi = blk >> 2, k = blk & 3;
                for (int wni = 0; wni < 2; ++wni) {
                    sum += scr[((wmi * 2 + wni) * 4 + k) * 2];
                    sq += scr[((wmi * 2 + wni) * 4 + k) * 2 + 1];
                }
            }
            const int g = (m0 + r0) / a.gn_cpg;
            const int ntl = tile0 / W2_TILES - ob * (D2 * H2 * TW / W2_TILES);      // part of the sample (512 positions each)
            const int idx = a.gn_cpg >= BM ? ntl * (a.gn_cpg / BM) + (m0 - g * a.gn_cpg) / BM : ntl;
            double* pp = a.gn_part + (((int64_t)ob * a.gn_G + g) * a.gn_nparts + idx) * 2;
            pp[0] = sum; pp[1] = sq;
        }
    }
}

// coverage of the F(2x2x2,3x3x3) kernel (precision 4): the F(2x2,3x3) shapes with kD = 3, an even depth, whole 64-channel
// blocks, the row pairs of a workgroup inside one plane, 8-byte aligned rows of y (and of the residual)
bool wg3_ok(const SdcConvDesc& d, bool small, bool rowhalo) {
    auto even = [](const int64_t* st) { return st[4] == 1 && st[0] % 2 == 0 && st[1] % 2 == 0 && st[2] % 2 == 0 && st[3] % 2 == 0; };
    if (!(d.precision == 4 && d.kD == 3 && d.oD % 2 == 0 && (d.oW == 16 || d.oW == 32 || d.oW == 64) && d.Cout % W2_BM == 0)) return false;
    SdcConvDesc e = d;
    e.precision = 3;
    if (!wg2_ok(e, small, rowhalo)) return false;
    const int rp = W2_TILES / (d.oW / 2);
    const bool nores = d.rs[0] == 0 && d.rs[1] == 0 && d.rs[2] == 0 && d.rs[3] == 0 && d.rs[4] == 0;     // (a fused residual: the F(2x2,3x3) kernel)
    return (d.oH / 2) % rp == 0 && even(d.ys) && nores;
}

int launch_wg3(const ConvArgs& a, hipStream_t s) {
    const SdcConvDesc& d = a.d;
    const int64_t tiles = (int64_t)d.B * (d.oD / 2) * (d.oH / 2) * (d.oW / 2);
    dim3 grid((unsigned)((tiles / W2_TILES) * (d.Cout / W2_BM)));
    const size_t lds = (size_t)W2_NBUF * (W2_ASZ + W2_BSZ) * sizeof(float) + 4 * 8 * sizeof(double) + 256 * 8 * sizeof(float);
#define W3_LAUNCH(OWV, D)                                                                                                        \
    do {                                                                                                                         \
        static std::atomic<uint64_t> attr{0};                                                                                    \
        SDC_LDS_OPTIN(attr, (conv_wg3_kernel<OWV, D>), 160 * 1024, "sdc_conv[winograd 2x2x2]");                                  \
        hipLaunchKernelGGL((conv_wg3_kernel<OWV, D>), grid, dim3(256), lds, s, a);                                               \
    } while (0)
#ifdef SDC_KERNEL_EXPERIMENTS
    // kernel experiments (2, 4: WRONG RESULTS): 2 no read-back, 4 no folds, 8 all read-backs of a fold up front
    static const int dbg = exp_env("SDC_WG3_DBG");
    if (d.oW == 64 && dbg) {
        switch (dbg) {
            case 2: W3_LAUNCH(64, 2); break; case 8: W3_LAUNCH(64, 8); break;
            default: W3_LAUNCH(64, 4); break;
        }
        return SDC_OK;
    }
#endif
    if (d.oW == 16) W3_LAUNCH(16, 0);
    else if (d.oW == 32) W3_LAUNCH(32, 0);
    else W3_LAUNCH(64, 0);
    return SDC_OK;
}

template <int BM, int BN, int WM, int WN, int SK, int NTH = 256, bool UPS = false>
int launch_wg(const ConvArgs& a, hipStream_t s) {
    dim3 grid((a.Ntot + BN - 1) / BN, (a.d.Cout + BM - 1) / BM);
    constexpr int KSMAX = BN + (BN / 16) * 2;
    constexpr int NCOLH = (KSMAX + 63) / 64;
    const size_t lds = (2u * 4u * SK * BM + 2u * SK * (NCOLH * 64 + 8)) * sizeof(float);
    static std::atomic<uint64_t> attr{0};
    SDC_LDS_OPTIN(attr, (conv_wg_kernel<BM, BN, WM, WN, SK, NTH, UPS>), 160 * 1024, "sdc_conv[winograd]");
    hipLaunchKernelGGL((conv_wg_kernel<BM, BN, WM, WN, SK, NTH, UPS>), grid, dim3(NTH), lds, s, a);
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
    w.pick = pick;
    w.bm = (pick == 6 || pick == 9) ? 128 : 64;
    w.bn = pick == 3 ? 128 : (pick == 6 ? 128 : (pick == 10 ? 512 : 256));
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

bool conv_small(const SdcConvDesc& d) {
    auto span = [](const int64_t* st, int b, int dd, int h, int w) {
        return (int64_t)(b - 1) * st[0] + (int64_t)(dd - 1) * st[2] + (int64_t)(h - 1) * st[3] + (int64_t)(w - 1) * st[4];
    };
    return span(d.x0s, d.B, d.iD, d.iH, d.iW) < (1ll << 30) && (d.Cin1 == 0 || span(d.x1s, d.B, d.iD, d.iH, d.iW) < (1ll << 30));
}

int conv_impl(const SdcConvDesc* dp, const float* x0, const float* x1, const float* wp, const float* bias,
              const float* residual, float* y, double* gn_part, int gn_G, void* stream);

}  // namespace

extern "C" int sdc_conv(const SdcConvDesc* dp, const float* x0, const float* x1, const float* wp, const float* bias,
                        const float* residual, float* y, void* stream) {
    return conv_impl(dp, x0, x1, wp, bias, residual, y, nullptr, 0, stream);
}

extern "C" int sdc_conv_gnparts(const SdcConvDesc* dp, int G) {
    if (!dp) return 0;
    static const int no_rh = exp_env("SDC_NO_ROWHALO");
    const int64_t ntot = (int64_t)dp->B * dp->oD * dp->oH * dp->oW;
    static const int no_wg2 = exp_env("SDC_NO_WG2");
    static const int no_wg3 = exp_env("SDC_NO_WG3");
    if (!no_wg2 && !no_wg3 && wg3_ok(*dp, conv_small(*dp), !no_rh)) return gn_parts_for(*dp, WgPick{21, W2_BM, W2_TILES * 8, false}, G);
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
              const float* residual, float* y, double* gn_part, int gn_G, void* stream) {
    SDC_REQUIRE(dp && x0 && wp && y, SDC_ENULL, "sdc_conv: null pointer");
    const SdcConvDesc& d = *dp;
    SDC_REQUIRE(d.B > 0 && d.Cin0 > 0 && d.Cin1 >= 0 && d.Cout > 0, SDC_EINVAL, "sdc_conv: bad channel/batch counts");
    SDC_REQUIRE(d.Cin1 == 0 || x1, SDC_ENULL, "sdc_conv: Cin1 > 0 but x1 is null");
    SDC_REQUIRE(d.kD > 0 && d.kH > 0 && d.kW > 0 && d.sD > 0 && d.sH > 0 && d.sW > 0, SDC_EINVAL,
                "sdc_conv: bad kernel/stride");
    SDC_REQUIRE(d.precision == 0 || (d.precision >= 2 && d.precision <= 4), SDC_EINVAL, "sdc_conv: precision must be 0 (fp32 MFMA, direct form), 2 (fp32 Winograd along W), 3 (fp32 Winograd over H and W) or 4 (fp32 Winograd over D, H and W)");
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
    const bool fast = (d.Cin0 % BK == 0) && (d.Cin1 % BK == 0) && small && d.Cout < (1 << 30);
    hipStream_t s = sdc::as_stream(stream);
    auto even = [](const int64_t* st) { return st[4] == 1 && st[0] % 2 == 0 && st[1] % 2 == 0 && st[2] % 2 == 0 && st[3] % 2 == 0; };
    static const int no_wg2 = exp_env("SDC_NO_WG2");
    static const int no_wg3 = exp_env("SDC_NO_WG3");
    // fp32 Winograd F(2x2x2,3x3x3): 3x3x3 stride-1 convs over whole rows, plane pairs
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
        if (wgp.ups) {
            if (wgp.pick == 6) { SDC_PICK("conv_wg_kernel<128,128,4,2,16,512,ups>", 2.0 / 3.0); { const int rc_ = launch_wg<128, 128, 4, 2, 16, 512, true>(a, s); if (rc_) return rc_; } }
            else if (wgp.pick == 7) { SDC_PICK("conv_wg_kernel<64,256,2,4,16,512,ups>", 2.0 / 3.0); { const int rc_ = launch_wg<64, 256, 2, 4, 16, 512, true>(a, s); if (rc_) return rc_; } }
            else { SDC_PICK("conv_wg_kernel<64,128,2,2,16,256,ups>", 2.0 / 3.0); { const int rc_ = launch_wg<64, 128, 2, 2, 16, 256, true>(a, s); if (rc_) return rc_; } }
            return sdc::check_launch("sdc_conv[winograd,upsample]");
        }
        if (wgp.pick == 6) { SDC_PICK("conv_wg_kernel<128,128,4,2,16,512>", 2.0 / 3.0); { const int rc_ = launch_wg<128, 128, 4, 2, 16, 512>(a, s); if (rc_) return rc_; } }
        else if (wgp.pick == 7) { SDC_PICK("conv_wg_kernel<64,256,2,4,16,512>", 2.0 / 3.0); { const int rc_ = launch_wg<64, 256, 2, 4, 16, 512>(a, s); if (rc_) return rc_; } }
        else if (wgp.pick == 9) { SDC_PICK("conv_wg_kernel<128,256,4,2,16,512>", 2.0 / 3.0); { const int rc_ = launch_wg<128, 256, 4, 2, 16, 512>(a, s); if (rc_) return rc_; } }       // (2 x 4 waves measured the same)
        else if (wgp.pick == 10) { SDC_PICK("conv_wg_kernel<64,512,1,8,16,512>", 2.0 / 3.0); { const int rc_ = launch_wg<64, 512, 1, 8, 16, 512>(a, s); if (rc_) return rc_; } }     // (each wave: both 32-row tiles x 32 pairs; measured 3 % ahead of 1 x 2)
        else { SDC_PICK("conv_wg_kernel<64,128,2,2,16,256>", 2.0 / 3.0); { const int rc_ = launch_wg<64, 128, 2, 2, 16>(a, s); if (rc_) return rc_; } }
        return sdc::check_launch("sdc_conv[winograd]");
    }
    SDC_REQUIRE(!gn_part, SDC_EINVAL, "sdc_conv_gn: shape not covered by the fused statistics (sdc_conv_gnparts returned 0)");
    // pointwise convs over a dense layout: 16-byte loads of weights and activations
    static const int no_pw = exp_env("SDC_NO_PW");
    {
        const int64_t S = (int64_t)d.oD * d.oH * d.oW;
        auto dense = [&](const int64_t* st) { return st[4] == 1 && st[3] == d.iW && st[2] == (int64_t)d.iH * d.iW && st[0] % 4 == 0 && st[1] % 4 == 0; };
        if (!no_pw && fast && d.kD * d.kH * d.kW == 1 && d.sD == 1 && d.sH == 1 && d.sW == 1 && d.uD == 1 && d.uH == 1 &&
            d.uW == 1 && d.up_mode == 0 && d.pD == 0 && d.pH == 0 && d.pW == 0 && d.oD == d.iD && d.oH == d.iH && d.oW == d.iW &&
            S % 4 == 0 && d.Cout % 4 == 0 && d.Cout > 32 && dense(d.x0s) && (d.Cin1 == 0 || dense(d.x1s)) &&
            reinterpret_cast<uintptr_t>(x0) % 16 == 0 && (d.Cin1 == 0 || reinterpret_cast<uintptr_t>(x1) % 16 == 0) &&
            reinterpret_cast<uintptr_t>(wp) % 16 == 0) {
            const int64_t b64x128 = (int64_t)((a.Ntot + 127) / 128) * ((d.Cout + 63) / 64);
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
