/*
 * libsdc_hip.so -- C ABI of the MI355X (gfx950) SafeDiffCon sampler kernels.
 *
 * Drop-in boundary: the reference has no native code on the
 * sampling path -- every stage is a torch.nn op inside
 *   1D/model/diffusion.py      GaussianDiffusion.p_sample_loop / p_sample
 *   tokamak/model/diffusion.py (same)
 *   2d/ddpm/diffusion_2d.py    GaussianDiffusion.p_sample_loop / p_sample
 * and the three U-Nets (1D/model/unet.py, tokamak/model/unet.py,
 * 2d/video_diffusion_pytorch/video_diffusion_pytorch_conv3d.py).  Each entry
 * point below replaces one *stage* of that path and cites the reference lines
 * whose arithmetic it implements.  The binding a reference maintainer adds is a
 * ctypes stub; safediffcon_amd/_lib.py is that stub.
 *
 * Conventions
 *  - plain pointers + sizes, no torch types; all tensors fp32 in device memory
 *    owned by the caller; the library never allocates, frees or retains them
 *    (except inside an explicit graph object, whose lifetime the caller ties
 *    to the buffers it captured).
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*);
 *    no hidden synchronisation, no host reads of device scalars.  The current
 *    diffusion timestep is a device-resident int (`t_dev`) so that one captured
 *    hipGraph replays for all 1000 steps.
 *  - return 0 on success, negative SDC_E* otherwise; text via sdc_last_error().
 *  - activation layout: (B, C, D, H, W) addressed through explicit element
 *    strides, so Conv1d (D=H=1), Conv2d (D=1), Conv3d and the smoke tensor's
 *    frame-major (B,F,C,H,W) storage all go through the same kernels.
 */
#ifndef SDC_H
#define SDC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDC_OK 0
#define SDC_EINVAL (-1)      /* bad shape / unsupported configuration */
#define SDC_EALIGN (-2)      /* misaligned pointer */
#define SDC_EHIP (-3)        /* HIP runtime error (launch, capture, ...) */
#define SDC_ENULL (-4)       /* required pointer is null */

int sdc_version(void);
/* copies the calling thread's last error message; returns its length */
int sdc_last_error(char* buf, size_t cap);

/* ------------------------------------------------------------------ conv */
/* Implicit-GEMM N-d convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32), gather
 * straight from the strided input(s):
 *   y[b,co,o] = bias[co] + sum_{tap,ci} Wp[tap*Cin+ci][co] * xcat[b,ci, o*stride - pad + tap] (+ residual[b,co,o])
 * xcat = channel-concat of x0 (Cin0 ch) and x1 (Cin1 ch) -- replaces torch.cat
 * before the up-path ResnetBlocks (1D/model/unet.py:414-418, conv3d.py:553,560).
 * `up` = virtual nearest-neighbour upsample of the input (nn.Upsample(2) + conv,
 * 1D/model/unet.py:24-37); up_mode 1 = zero insertion (ConvTranspose3d as a conv
 * over the zero-stuffed input with flipped taps, conv3d.py:159-160).
 * Replaces: every nn.Conv1d/2d/3d, nn.Linear and nn.ConvTranspose3d on the path
 * (1D/model/unet.py:132-134,161-163,189-197,232-236,326,345,370,378;
 *  conv3d.py:159-163,192,218,239-240,291-292,395,471).
 * Output size per axis: (i*u + 2p - k)/s + 1 (i -> (i-1)*u+1 for zero insertion); up to k-1 more positions are
 * accepted and read implicit zeros past the far edge (one-sided padding), fewer compute a prefix.
 * Wp is the caller-repacked weight [K = taps*Cin][Cout] (row-major, Cout fastest).
 */
typedef struct SdcConvDesc {
    int32_t B, Cin0, Cin1, Cout;
    int32_t iD, iH, iW;          /* stored input spatial size */
    int32_t oD, oH, oW;          /* output spatial size */
    int32_t kD, kH, kW;
    int32_t sD, sH, sW;          /* stride */
    int32_t pD, pH, pW;          /* padding, in the (virtually upsampled) input space */
    int32_t uD, uH, uW;          /* virtual input upsample factor, 1 or 2 */
    int32_t up_mode;             /* 0 nearest, 1 zero-insert */
    int32_t precision;           /* conv algorithm and layout of the wp buffer (0, 2, 3 and 4 are fp32 end to end and differ by
                                    rounding order only; the drop-in nets use 4):
                                    0 = fp32 MFMA, direct implicit GEMM everywhere (k-ordered fp32 FMA chains); wp = Wp
                                    2 = fp32 MFMA, Winograd F(2,3) along W on the 3-wide stride-1 convs (2/3 of the matrix
                                        work); wp = Wp followed, when kW == 3, by the transformed taps
                                        Wg[(kd*kH + kh)*4 + xi][Cin][Cout]  (G g, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],
                                        formed in fp64, rounded once); convs the kernel does not cover run the direct kernels
                                    3 = as 2, plus Winograd F(2x2,3x3) over (H, W) for the 3x3 / 3x3x3 stride-1 pad-1 convs over
                                        whole contiguous rows of 16 / 32 / 64 / 128 columns (4/9 of the matrix work): when
                                        kH == kW == 3 the buffer is Wp | Wg | Wg2 with Wg2[kd][ci][co][j*4 + xi] =
                                        sum_{kh,kw} G[j][kh] G[xi][kw] w[co][ci][kd][kh][kw]; other tap shapes: layout of 2.
                                        Shapes the F(2x2,3x3) kernel does not take fall back to 2's kernels on the same buffer.
                                    4 = as 3, plus Winograd F(2x2x2,3x3x3) over (D, H, W) for the 3x3x3 stride-1 pad-1 convs with
                                        an even depth, rows of 16 / 32 / 64 columns and Cout % 64 == 0 (8/27 of the matrix work):
                                        when kD == kH == kW == 3 the buffer is Wp | Wg | Wg2 | Wg3 with Wg3[jd][ci][co][j*4 + xi]
                                        = sum_{kd,kh,kw} G[jd][kd] G[j][kh] G[xi][kw] w[co][ci][kd][kh][kw]; other tap shapes:
                                        layout of 3.  y is used as scratch for partial plane sums while the kernel runs (it
                                        must not alias an input).  Shapes it does not take fall back to 3's kernels.
                                    5 = as 4, plus Winograd F(4,3) along W for the 1-D convs (kD == kH == 1, rows of whole quads,
                                        Cout >= 128; half of the matrix work): such a conv's buffer is Wp | Wg | Wg43 with
                                        Wg43[xi][Cin][Cout], xi < 6, = G43 g (rows (1/4,0,0), (-1/6,-1/6,-1/6), (-1/6,1/6,-1/6),
                                        (1/24,1/12,1/6), (1/24,-1/12,1/6), (0,0,1)).  Opt-in, not the nets' default: on MI355X it is
                                        no faster than 2's kernel (the transform is not hidden behind the MFMAs it saves) and
                                        rounds three times coarser (~3e-6 of the output scale).
                                    (1 was a split-bf16 mode in rounds 1-2; removed: every mode is fp32 arithmetic) */
    int64_t x0s[5], x1s[5], ys[5], rs[5];   /* element strides (b,c,d,h,w) */
} SdcConvDesc;

int sdc_conv(const SdcConvDesc* d, const float* x0, const float* x1, const float* wp, const float* bias,
             const float* residual, float* y, void* stream);

/* sdc_conv for grids that leave most of the chip idle (the fine-tuning step, SURVEY 8f: batch 64 puts the deep 3x3 convs of the
 * Burgers net on 32-128 workgroups for 256 CUs; the samplers' small-batch plan, SURVEY 8e: an 8-way shard of the 1-D configs leaves
 * 16-32 trajectories per GPU): the input channels of an F(2x2,3x3) conv, of a single-tap-row F(2,3) conv (Conv1d k3, with or
 * without the nearest x2 upsampling folded into its gather), or the (tap, channel) walk of a direct-form conv on its smallest tile
 * (1x1, strided, sub-pixel), are split over up to 8 workgroups per output tile, the partial outputs
 * go to `work` (sdc_conv_splitk_bytes(d) bytes, 16-byte aligned; 0 = this conv is not split: the call is then exactly sdc_conv) and
 * are summed in split order -- deterministic, but the split depends on the batch, so a sample's rounding depends on the batch it
 * rides in: the samplers use sdc_conv only unless the caller opts in (net.split_small_grids).  No fused residual. */
size_t sdc_conv_splitk_bytes(const SdcConvDesc* d);
int sdc_conv_splitk(const SdcConvDesc* d, const float* x0, const float* x1, const float* wp, const float* bias, float* y,
                    float* work, size_t work_bytes, void* stream);

/* Host-side query (measurement tooling, launches nothing): which kernel template instance sdc_conv would run for this
 * descriptor, and the share of the direct-form multiply-adds 2*B*P*Cout*Cin*taps that it issues on the matrix cores
 * (1 for the direct kernels, 2/3 for Winograd F(2,3) along W, 4/9 for F(2x2,3x3), 8/27 for F(2x2x2,3x3x3)). */
int sdc_conv_describe(const SdcConvDesc* d, char* name, size_t cap, double* mfma_share);

/* Conv + GroupNorm statistics of its output in one pass (the Block.proj -> Block.norm pair, 1D/model/unet.py:132-141,
 * conv3d.py:192-198): the conv epilogue leaves fp64 (sum, sum of squares) pairs per (sample, group, part) in `parts`
 * (B * G * nparts pairs) and sdc_gn_finalize turns them into the {mean, rstd} table sdc_gn_apply reads -- y is not read
 * again for the statistics.  sdc_conv_gnparts is a host-side query: nparts for this descriptor, or 0 when the fused form
 * does not cover it (then run sdc_conv + sdc_gn_stats).  Covered: the precision-2 / 3 / 4 Winograd convs whose tile grid
 * lines up with the samples and groups. */
int sdc_conv_gnparts(const SdcConvDesc* d, int G);
int sdc_conv_gn(const SdcConvDesc* d, const float* x0, const float* x1, const float* wp, const float* bias,
                const float* residual, float* y, double* parts, int G, void* stream);

/* ------------------------------------------------------------- group norm */
/* nn.GroupNorm(G, C, eps=1e-5) statistics over contiguous (B, C, S) data:
 * stats[(b*G+g)*2 + {0,1}] = {mean, rstd}, accumulated in fp64.
 * 1D/model/unet.py:135,140 ; conv3d.py:193,198. */
int sdc_gn_stats(const float* x, float* stats, int B, int C, int G, int64_t S, float eps, void* stream);
/* bytes the caller must allocate for `stats` (mean/rstd pairs + fp64 partial-sum scratch) */
size_t sdc_gn_stats_bytes(int B, int G);
/* stats from the partial sums of sdc_conv_gn: n_per_group = (C / G) * S elements per (sample, group) */
int sdc_gn_finalize(const double* parts, float* stats, int B, int G, int nparts, int64_t n_per_group, float eps, void* stream);

/* y = SiLU( ((x-mean)*rstd*gamma[c]+beta[c]) * (scale+1) + shift ) (+ residual)
 * Block.forward + the ResnetBlock residual add: 1D/model/unet.py:138-147,180 ; conv3d.py:196-204,230.
 * scale/shift come from a time-conditioning table ss (nullable):
 *   row = (t_dev ? *t_dev : 0) * ss_t_stride + b * ss_b_stride
 *   scale = ss[row + ss_off + c], shift = ss[row + ss_off + C + c]
 * (ResnetBlock.mlp output chunked in two, 1D/model/unet.py:169-175). */
int sdc_gn_apply(const float* x, const float* stats, const float* gamma, const float* beta, const float* ss,
                 const int32_t* t_dev, int64_t ss_t_stride, int64_t ss_b_stride, int64_t ss_off,
                 const float* residual, float* y, int B, int C, int G, int64_t S, void* stream);

/* GroupNorm statistics + apply in ONE launch for small groups (sdc_gn_fused_ok: (C/G)*S <= 32768, whatever the batch): the deep levels
 * of Unet2D / Unet1D, 1D/model/unet.py:128-147.  Arguments as sdc_gn_apply (fp64 statistics like sdc_gn_stats). */
int sdc_gn_fused_ok(int B, int C, int G, int64_t S);
int sdc_gn_fused(const float* x, const float* gamma, const float* beta, const float* ss, const int32_t* t_dev,
                 int64_t ss_t_stride, int64_t ss_b_stride, int64_t ss_off, const float* residual, float* y, int B, int C, int G,
                 int64_t S, float eps, void* stream);

/* GroupNorm apply + SiLU (+ residual) of the LAST ResnetBlock inside the 1x1(x1) output conv that is its only reader
 * (2d/video_diffusion_pytorch/video_diffusion_pytorch_conv3d.py:468-471 `final_conv`, 1D/model/unet.py:376-378,
 * tokamak/model/unet.py:355-357): y[b, o, pos] = bias[o] + sum_c w[o][c] * (SiLU((h[b, c, pos] - mean) * rstd * gamma[c] + beta[c]) +
 * residual[b, c, pos]) in one streaming pass -- the normalised tensor is never written.  h / residual contiguous (B, C, S), 16-byte
 * aligned; stats from sdc_gn_stats / sdc_gn_finalize; w = the nn.Conv weight (Cout, C) as it lies, Cout <= 16; y through strides:
 * element (b, o, s) at b * ys0 + o * ys1 + (s / plane) * ys2 + s % plane, plane = H * W (the smoke net writes eps frame-major);
 * plane and the strides multiples of 4 floats.  HBM-bound: 4 * (2 C + Cout) bytes per position. */
int sdc_gn_pointwise_out(const float* h, const float* stats, const float* gamma, const float* beta, const float* residual,
                         const float* w, const float* bias, float* y, int B, int C, int G, int Cout, int64_t S, int64_t plane,
                         int64_t ys0, int64_t ys1, int64_t ys2, void* stream);

/* ------------------------------------------------------- channel norms */
/* mode 0: channel LayerNorm, gain only, (x-mean)*rsqrt(var+eps)*g   1D/model/unet.py:53-63, conv3d.py:165-174
 * mode 1: RMSNorm  x / max(||x||_2,1e-12) * g * sqrt(C)            tokamak/model/unet.py:45-51
 * over contiguous (B, C, S); optional residual add (Residual(...) wrapper, unet.py:16-22). */
int sdc_chan_norm(const float* x, const float* g, const float* residual, float* y, int B, int C, int64_t S,
                  int mode, float eps, void* stream);

/* ---------------------------------------------------- linear attention */
/* LinearAttention / SpatialLinearAttention core, heads x 32:
 *   ctx[d,e] = sum_n softmax_n(k)[d,n] v[e,n] ; out[e,n] = sum_d ctx[d,e] softmax_d(q)[d,n]*scale
 * 1D/model/unet.py:203-221 ; conv3d.py:246-256.
 * qkv is (outer, 3*heads*32, inner, n) through strides: element(outer o, channel c, inner i, token n) at
 *   o*so + c*sc + i*si + n  (tokens contiguous).  out uses the same scheme with heads*32 channels.
 * ctx is workspace of outer*inner*heads*32*32 floats. */
int sdc_linattn(const float* qkv, float* ctx, float* out, int outer, int inner, int heads, int64_t n,
                int64_t q_so, int64_t q_sc, int64_t q_si, int64_t o_so, int64_t o_sc, int64_t o_si, void* stream);

/* Fused LinearAttention block, dim C in {64, 128}, heads 4 x 32, tokens contiguous, n % 64 == 0:
 *   y = x + post( Wo . LA(pre(x)) + bo )
 * i.e. Residual(PreNorm(dim, LinearAttention(dim))) in one call: 1D/model/unet.py:64-71,182-222,341-342;
 * tokamak/model/unet.py:186-222; conv3d.py:176-184,232-258 (SpatialLinearAttention, post_mode -1).
 * pre_mode / post_mode: 0 channel LayerNorm (gain only), 1 RMSNorm, post_mode -1 = none.
 * wqkv = packed 1x1 weight [C][384] (q | k | v), wo = packed [128][C], bo = [C] or null.
 * x and y share the layout element(o, c, i, tok) at o*so + c*sc + i*si + tok.  `work` holds
 * sdc_linattn_block_bytes(...) bytes of scratch.  Three launches: x is read three times, y written once. */
size_t sdc_linattn_block_bytes(int outer, int inner, int C, int64_t n);
int sdc_linattn_block(const float* x, const float* g_pre, const float* wqkv, const float* wo, const float* bo,
                      const float* g_post, float* work, float* y, int outer, int inner, int C, int64_t n,
                      int64_t so, int64_t sc, int64_t si, int pre_mode, int post_mode, float eps, void* stream);
/* The same block reading the RAW output of the producing conv: the ResnetBlock's second GroupNorm + SiLU (+ the residual
 * branch) -- conv3d.py:189-230 block2 / res_conv, consumed only by the attention block that follows (conv3d.py:537-545:
 * block2 -> spatial_attn -> temporal_attn -> skip) -- is applied while pass 1 loads its tiles, instead of in a separate
 * sdc_gn_apply pass:  h = SiLU(x_raw * rstd gamma + (beta - mean rstd gamma)) + residual; pass 1 leaves h in y, where pass 2
 * reads it tile by tile before overwriting it (y must not alias x_raw or the residual).
 * gn_stats = (mean, rstd) per (outer index, group) as sdc_gn_finalize / sdc_gn_stats write them; gn_residual has x_raw's
 * strides or is null.  Same arithmetic as sdc_gn_apply followed by sdc_linattn_block. */
int sdc_linattn_block_gn(const float* x_raw, const float* gn_stats, const float* gn_gamma, const float* gn_beta, int gn_G,
                         const float* gn_residual, const float* g_pre, const float* wqkv, const float* wo, const float* bo,
                         const float* g_post, float* work, float* y, int outer, int inner, int C, int64_t n,
                         int64_t so, int64_t sc, int64_t si, int pre_mode, int post_mode, float eps, void* stream);

/* Fused temporal-attention block of the smoke U-Net, dim 64, 32 frames, heads 4 x 32:
 *   y = x + Wo . softmax( rot(s Wq xn) rot(Wk xn)^T + relpos ) (Wv xn),  xn = channel LayerNorm(x) * gamma
 * Residual(PreNorm(dim, EinopsToAndFrom(Attention))): conv3d.py:165-184,262-275,277-353,383,402-405.
 * wqkv = packed [64][384] (bias-free Linear), wo = packed [128][64]; rot = [32][16][2] (cos, sin) or null,
 * bias = [heads][query][key] or null.  Element (o, c, pixel i, frame f) of x and y at o*so + c*sc + f*st + i;
 * inner = pixels per outer index, a multiple of 8.
 * wqkv and wo must be 16-byte aligned, rot 8-byte aligned (vector loads); SDC_EINVAL otherwise. */
int sdc_tattn_block(const float* x, const float* g_pre, const float* wqkv, const float* wo, const float* rot,
                    const float* bias, float* y, int outer, int inner, int C, int ntok, int64_t so, int64_t sc,
                    int64_t st, float eps, void* stream);

/* ------------------------------------------------------- softmax attention */
/* Attention core (heads x 32): out = softmax(q*scale . k^T + bias) v, optional rotary on q,k.
 * 1D/model/unet.py:247-251 ; conv3d.py:313-353 (focus_present_mask all-False).
 * Sequences are indexed (outer, inner); element (o, c, i, tok) at o*so + c*sc + i*si + tok*st.
 * rot: [ntok][16][2] cos/sin table or null; bias: [heads][ntok][ntok] or null. ntok <= 256. */
int sdc_attn(const float* qkv, float* out, const float* rot, const float* bias, int outer, int inner, int heads,
             int ntok, int64_t q_so, int64_t q_sc, int64_t q_si, int64_t q_st,
             int64_t o_so, int64_t o_sc, int64_t o_si, int64_t o_st, void* stream);

/* ------------------------------------------------------------ elementwise */
/* kind 0: SiLU, 1: exact (erf) GELU -- time_mlp / ResnetBlock.mlp (unet.py:152-155,310-315) */
int sdc_act(const float* x, float* y, int64_t n, int kind, void* stream);

/* ----------------------------------------------------- fused DDPM update */
/* One reverse step for every element, fused with guidance and the conditioning writes:
 *   x0  = a x - b eps ; eps' = eps + k g(x0) ; x0' = clamp(a x - b eps', -1, 1)
 *   x-  = c1 x0' + c2 x + sigma z ; then re-impose the conditions (unless last step for burgers/tokamak)
 * p_sample / p_mean_variance / model_predictions / q_posterior:
 *   1D/model/diffusion.py:217-306, tokamak/model/diffusion.py:201-266, 2d/ddpm/diffusion_2d.py:242-285;
 * conditioning writes 1D/model/diffusion.py:336-366,380-394, tokamak/...:295-308,330-336, 2d/...:297-301,310-312.
 * coef: [T][8] floats {a, b, c1, c2, sigma(0 at t=0), k(J_scheduler), 0, 0}; t_dev: device timestep.
 * noise: explicit z tensors [n_draw][numel] (parity mode) indexed by *draw_dev, or null -> Philox4x32-10
 * normal draws keyed by (seed, *draw_dev, element).  gscal: per-sample guidance scalars from sdc_guide_reduce.
 */
#define SDC_MODEL_BURGERS 0
#define SDC_MODEL_TOKAMAK 1
#define SDC_MODEL_SMOKE 2

typedef struct SdcStepDesc {
    int32_t model;               /* SDC_MODEL_* */
    int32_t B;
    int32_t d0, d1, d2, d3;      /* per-sample dims: burgers (C,H,W,1) ; tokamak (C,L,1,1) ; smoke (F,C,H,W) */
    int32_t guide;               /* 0 none, 1 built-in closed form, 2 external g tensor */
    int32_t clip;                /* clip_denoised */
    int32_t impose;              /* 1: apply conditioning writes to the output; 2 (smoke): control channels only */
    int32_t cond_idx;            /* burgers: condition_idx (10); tokamak: nt (122) */
    int32_t pad_zero;            /* burgers/tokamak: !train_on_padded_locations */
    int32_t use_max;             /* burgers: 0 mean-mode (use_max_safety=True), 1 amax mode */
    int32_t has_wgt;             /* burgers: w_groundtruth given ; smoke: control given */
    int32_t skip_draws;          /* noise draws consumed per step beyond the one used (calibration branch: 1) */
    int32_t ddim;                /* 1: DDIM update (ddim_sample, 1D/model/diffusion.py:451-555, 2d/...:324-404): x0 always
                                    clipped, guidance on the clipped x0, eps re-derived, coef rows = {a, b, sqrt(a_next), c,
                                    sigma, k, last} indexed by step number */
    uint64_t seed;
} SdcStepDesc;

/* gpar: device float[8] of guidance constants (kept on the device so a captured graph survives a new
 * conformal quantile Q):  burgers {w_score, u_bound^2, Q, 10}; tokamak {w_obj, w_safe, guidance_scaler,
 * safety_threshold, Q}; smoke {w_safe, safe_bound, Q, standard_fixed_ratio}.
 * sdc_guide_reduce: per-sample hinge-active flag + arg-extremum of the safety functional evaluated on
 * x0 = a x - b eps  -> gscal[4*B] = {active, arg, extremum, 1/ties}   (1D/utils/guidance.py:58-77, tokamak/utils/guidance.py:32-56,
 * tokamak/utils/metrics.py:144-151, 2d/inference_2d.py:173-186). */
int sdc_guide_reduce(const SdcStepDesc* d, const float* x, const float* eps, const float* coef, const int32_t* t_dev,
                     const float* gpar, float* gscal, void* stream);
/* guide: 0 none | 1 built-in closed-form gradient (needs gpar, gscal[, target]) | 2 external gradient tensor gext
 * | 3 write x0 = a x - b eps to x0out only (so a caller-supplied nablaJ callable can be evaluated on it). */
int sdc_step_update(const SdcStepDesc* d, const float* x, const float* eps, const float* gext, const float* coef,
                    const int32_t* t_dev, const int32_t* draw_dev, const float* noise, int64_t noise_stride,
                    const float* gpar, const float* gscal, const float* target, const float* c0, const float* c1,
                    const float* c2, float* xout, float* x0out, void* stream);
/* conditioning writes only (initial x_T) */
int sdc_impose(const SdcStepDesc* d, float* x, const float* c0, const float* c1, const float* c2, void* stream);
/* x = N(0,1) from the same Philox stream (draw index *draw_dev), then (*draw_dev)++ handled by sdc_advance */
int sdc_randn(float* x, int64_t n, uint64_t seed, const int32_t* draw_dev, void* stream);
/* *t_dev += dt ; *draw_dev += ddraw   (one thread; keeps the step counter on the device for graph replay) */
int sdc_advance(int32_t* t_dev, int dt, int32_t* draw_dev, int ddraw, void* stream);

/* DDIM: ++*idx_dev ; *t_dev = ttab[*idx_dev] (the strided timestep list lives on the device) ; *draw_dev += ddraw */
int sdc_advance_table(int32_t* idx_dev, int32_t* t_dev, const int32_t* ttab, int32_t* draw_dev, int ddraw, void* stream);

/* ------------------------------------------------------------- conformal */
/* per-sample conformal score |f(pred) - f(truth)| and weight exp(-J(truth)):
 * 1D/inference/conformal.py:68-85 + guidance.py:9-46 ; tokamak/inference/conformal.py:79-109 ;
 * 2d/inference_2d.py:83-92,139-144.  Uses SdcStepDesc for model + constants. */
int sdc_conformal_score(const SdcStepDesc* d, const float* pred, const float* truth, const float* target,
                        const float* gpar, float* score, float* weight, void* stream);

/* --------------------------------------------------- evaluation rollout */
/* Explicit finite-difference Burgers' solver used to score a sampled control:
 * burgers_numeric_solve_free, 1D/data/generate_burgers.py:207-299, called by control_trajectories,
 * 1D/utils/metrics.py:42-65.  u0 (N,s), f (N,Nt,s) -> traj (N,Nt+1,s); `steps` Euler steps of size dt, the force row
 * advances and a snapshot is recorded every `record_every` steps; coef_transport = 1/(2 dx), d0..d2 = visc*[1,-2,1]/dx^2
 * (all rounded to fp32 by the caller exactly as the reference's FloatTensor(...) does). */
int sdc_burgers_rollout(const float* u0, const float* f, float* traj, int N, int s, int Nt, int steps, int record_every,
                        float dt, float coef_transport, float d0, float d1, float d2, void* stream);

/* ------------------------------------------------- backward (fine-tuning path) */
/* The VJP kernels of the fine-tuning loss: p_losses + loss.backward() through the U-Nets,
 * 1D/model/diffusion.py:638-733, 2d/ddpm/diffusion_2d.py:434-452 (callers 1D/inference/inference_ft.py:183-187,
 * tokamak/inference/pipeline.py:238-263, 2d/inference_2d.py:267-279).  Conv DATA gradients have no entry point of their
 * own: they are convolutions with flipped / transposed taps and run on sdc_conv with re-packed weights. */

/* Weight (and bias) gradient of nn.Conv1d/2d/3d, nn.Linear and nn.ConvTranspose3d:
 *   dw[m][n][kd][kh][kw] = sum_{b, od, oh, ow} G[b][m][od][oh][ow] * X[b][n][(od sD - pD + kd) / uD][...][(ow sW - pW + kw) / uW]
 * (terms outside X are zero; uD/uH/uW in {1,2} = nearest upsampling of X folded into the read, nn.Upsample + conv).
 * For a conv  y = conv(x, w): G = dL/dy (M = Cout), X = x (N = Cin) -> dw in nn.Conv layout (Cout, Cin, k...).
 * For a transposed conv y = convT(x, w): G = x (M = Cin), X = dL/dy (N = Cout), s/p of the layer -> nn.ConvTranspose layout.
 * dbias[m] = sum G[b][m][...] or null.  (kW, sW) in {(1,1),(3,1),(7,1),(4,2),(2,2)}; rows are walked in chunks of 16 positions.
 * fp32 MFMA; the positions are split over workgroups and summed in a fixed order (deterministic). */
typedef struct SdcWgradDesc {
    int32_t B, M, N;
    int32_t oD, oH, oW;          /* positions of G */
    int32_t iD, iH, iW;          /* stored size of X */
    int32_t kD, kH, kW;
    int32_t sD, sH, sW;
    int32_t pD, pH, pW;
    int32_t uD, uH, uW;
    int32_t _pad;
    int64_t gs[5], xs[5];        /* element strides (b, c, d, h, w) of G and X */
} SdcWgradDesc;
size_t sdc_conv_wgrad_bytes(const SdcWgradDesc* d);
int sdc_conv_wgrad(const SdcWgradDesc* d, const float* g, const float* x, float* dw, float* dbias, void* work,
                   size_t work_bytes, void* stream);

/* Backward of sdc_gn_apply (GroupNorm -> (scale+1, shift) -> SiLU; Block, conv3d.py:189-204, 1D/model/unet.py:128-147):
 * h = the conv output the forward normalised (contiguous (B,C,S)), stats from the forward, ss = per-sample rows
 * [scale (C) | shift (C)] at ss + b*ss_b_stride or null.  rows: sdc_gn_silu_bwd_floats(B, C, G, S) floats, 8-byte aligned.  Writes
 * gh = dL/dh and rows[b][c] = (A1, A2) with
 * A1 = sum_S gy silu'(v), A2 = sum_S gy silu'(v) xhat, and -- when dgamma / dbeta (C floats each) are given --
 * dgamma[c] = sum_b (1+scale) A2, dbeta[c] = sum_b (1+scale) A1 and, when dss (contiguous (B, 2C)) is given too,
 * dss[b] = [gamma A2 + beta A1 | A1], the gradient of the ss rows.  (The residual the forward added passes gy through.) */
size_t sdc_gn_silu_bwd_floats(int B, int C, int G, int64_t S);
int sdc_gn_silu_bwd(const float* h, const float* gy, const float* stats, const float* gamma, const float* beta,
                    const float* ss, int64_t ss_b_stride, float* rows, float* gh, float* dgamma, float* dbeta, float* dss,
                    int B, int C, int G, int64_t S, void* stream);

/* Backward of sdc_chan_norm (mode 0 channel LayerNorm, 1 RMSNorm): gx, and partial sums of the gain gradient.
 * gpart = sdc_chan_norm_bwd_bytes(B, C, S) bytes of scratch: [C][sdc_chan_norm_bwd_parts(B, S)] partials (the caller sums the
 * last axis) followed by the kernel's per-position (mean, scale) table. */
size_t sdc_chan_norm_bwd_parts(int B, int64_t S);
size_t sdc_chan_norm_bwd_bytes(int B, int C, int64_t S);
int sdc_chan_norm_bwd(const float* x, const float* gy, const float* g, float* gx, float* gpart, int B, int C, int64_t S,
                      int mode, float eps, void* stream);

/* Kernel layout of an nn.Conv weight w (Cout, Cin, kD, kH, kW) for SdcConvDesc.precision (0, 2, 3, 4, 5) in one launch: Wp followed
 * by the Winograd taps the precision / tap shape call for (layouts: SdcConvDesc.precision above), transformed taps summed in fp64
 * and rounded once.  flip != 0 packs the DATA-GRADIENT weight of the same conv instead (channels transposed, taps flipped; then the
 * arguments Cout / Cin are those of the packed weight, i.e. swapped).  out holds sdc_pack_conv_weight_floats(...) floats. */
size_t sdc_pack_conv_weight_floats(int Cout, int Cin, int kD, int kH, int kW, int precision);
int sdc_pack_conv_weight(const float* w, float* out, int Cout, int Cin, int kD, int kH, int kW, int precision, int flip, void* stream);

/* The same packing for MANY weights in one launch (a fine-tuning step re-packs every conv of the net, twice: forward and
 * data-gradient form -- reference callers 1D/inference/inference_ft.py:183-226, 2d/inference_2d.py:267-279 step the optimiser
 * between forwards).  The caller fills w, out, Cout, Cin, kD, kH, kW, precision, flip of every item (meaning as in
 * sdc_pack_conv_weight; out holds sdc_pack_conv_weight_floats(...) floats); sdc_pack_batch_plan (host only, no GPU call)
 * fills the remaining fields and returns the launch size; the caller copies the table to device memory once and calls
 * sdc_pack_batch_run(table_dev, ...) whenever the weights changed.  Results are those of sdc_pack_conv_weight bit for bit. */
typedef struct SdcPackItem {
    const float* w;
    float* out;
    int Cout, Cin, kD, kH, kW, precision, flip;
    int co_sh, ci_sh, grid_x, grid_y, block0;     /* filled by sdc_pack_batch_plan */
    unsigned tap_magic;
    int64_t n[5];
} SdcPackItem;
int sdc_pack_batch_plan(SdcPackItem* items, int n, int* total_blocks, int* lds_bytes);
int sdc_pack_batch_run(const SdcPackItem* items_dev, int n, int total_blocks, int lds_bytes, void* stream);

/* Backward of the attention cores (same tensor conventions as sdc_attn / sdc_linattn: q, k, v = channel ranges of qkv, the
 * gradients dqkv in the same layout; dout = dL/dout in the layout of `out`).
 * sdc_attn_bwd: dbias (heads, ntok, ntok) = sum over sequences of dS, or null (built for ntok <= 32: the temporal attention's
 * relative-position bias, conv3d.py:74-112); work = sdc_attn_bwd_bytes(...) bytes when dbias is requested. */
size_t sdc_attn_bwd_bytes(int outer, int inner, int heads, int ntok);
int sdc_attn_bwd(const float* qkv, const float* dout, const float* rot, const float* bias, float* dqkv, float* dbias, void* work,
                 int outer, int inner, int heads, int ntok, int64_t q_so, int64_t q_sc, int64_t q_si, int64_t q_st,
                 int64_t o_so, int64_t o_sc, int64_t o_si, int64_t o_st, void* stream);
size_t sdc_linattn_bwd_bytes(int outer, int inner, int heads, int64_t n);      /* scratch of sdc_linattn_bwd */
int sdc_linattn_bwd(const float* qkv, const float* dout, float* dqkv, void* work, int outer, int inner, int heads, int64_t n,
                    int64_t q_so, int64_t q_sc, int64_t q_si, int64_t o_so, int64_t o_sc, int64_t o_si, void* stream);

/* gx = gy * f'(x): kind 0 SiLU, 1 GELU (exact erf) -- time_mlp, 1D/model/unet.py:300-305 */
int sdc_act_bwd(const float* x, const float* gy, float* gx, int64_t n, int kind, void* stream);

/* VJP of nearest-neighbour upsampling by (fh, fw) in {1,2}^2 over the last two axes: gx (rows,H,W) from g (rows,H*fh,W*fw) */
int sdc_sumpool2(const float* g, float* gx, int64_t rows, int H, int W, int fh, int fw, void* stream);

/* ------------------------------------------------- nn.Linear over a batch of rows (fine-tuning path) */
/* The time MLP and the ResnetBlocks' scale/shift MLPs (1D/model/unet.py:300-305, :158-162; tokamak/model/unet.py likewise;
 * 2d/video_diffusion_pytorch/video_diffusion_pytorch_conv3d.py:212-216, :399-404) run forward and backward in every fine-tuning
 * step (the samplers read them from a per-timestep table).  w is the nn.Linear weight [M][K] as torch stores it; rows of x / y /
 * gy / gx may be strided (strides in floats).  K, M and the strides must be multiples of 4 and the pointers 16-byte aligned
 * (SDC_EINVAL otherwise: the caller then uses sdc_conv / sdc_conv_wgrad, which take any shape).  Fixed summation order, no atomics.
 *   sdc_linear        y[r][m]  = bias[m] + sum_k x[r][k] w[m][k]          (bias may be null)
 *   sdc_linear_dgrad  gx[r][k] = sum_m gy[r][m] w[m][k]
 *   sdc_linear_wgrad  gw[m][k] = sum_r gy[r][m] x[r][k],  gbias[m] = sum_r gy[r][m]   (gbias may be null; gw contiguous [M][K]) */
int sdc_linear(const float* x, const float* w, const float* bias, float* y, int rows, int K, int M, int64_t x_stride,
               int64_t y_stride, void* stream);
int sdc_linear_dgrad(const float* gy, const float* w, float* gx, int rows, int K, int M, int64_t gy_stride, int64_t gx_stride,
                     void* stream);
int sdc_linear_wgrad(const float* gy, const float* x, float* gw, float* gbias, int rows, int K, int M, int64_t gy_stride,
                     int64_t x_stride, void* stream);

/* ------------------------------------------------- weight content stamp */
/* out_dev[0] = order-independent 64-bit checksum over n device spans of 32-bit words (span index, word index and bits all
 * enter it).  The host wrapper stamps a plan's packed weights with it: `p.data.lerp_()` / `p.data = ...` (the reference's EMA
 * updates, 2d/video_diffusion_pytorch/video_diffusion_pytorch_conv3d.py:121-124; ema_pytorch in 1D/model/trainer.py,
 * tokamak/model/trainer.py) change the parameters without moving any autograd version counter.  spans_dev, out_dev: device memory. */
typedef struct {
    const void* ptr;          /* 4-byte aligned device address */
    int64_t nwords;           /* 32-bit words */
} SdcSpan;
int sdc_checksum_spans(const SdcSpan* spans_dev, int n, uint64_t* out_dev, void* stream);

/* ----------------------------------------------------------------- graphs */
int sdc_graph_begin(void* stream);
int sdc_graph_end(void* stream, void** graph_exec);
int sdc_graph_launch(void* graph_exec, void* stream);
int sdc_graph_destroy(void* graph_exec);

/* ------------------------------------------------------ timing (bench only) */
int sdc_event_create(void** ev);
int sdc_event_record(void* ev, void* stream);
int sdc_event_elapsed_ms(void* ev0, void* ev1, float* ms);   /* synchronises on ev1 */
int sdc_event_destroy(void* ev);

/* ---- Tokamak score check: the KSTAR surrogate rollout (tokamak/kstar_solver.py:163-428 KSTARSolver.control / predict_0d /
 * simulate; the Keras networks of tokamak/common/model_structure.py:69-152; called per sample by control_trajectories,
 * tokamak/utils/metrics.py:60-85).  Replaces the 122 x B serial single-sample Keras predict() calls with one launch. */
#define SDC_KSTAR_SEQ 10      /* window rows (seq_len, kstar_solver.py:34) */
#define SDC_KSTAR_NIN 18      /* window columns: 4 fed-back outputs + 13 inputs + year */
#define SDC_KSTAR_UNITS 100   /* LSTM units of kstar_v220505 (model_structure.py:108) */
typedef struct {
    int nlayers;              /* BatchNormalization -> Dense blocks (Dropout is the identity at inference) */
    int width[7];             /* width[0] inputs ... width[nlayers] outputs, each <= 256 */
    int act[6];               /* per block: 0 linear, 1 sigmoid */
    const float* params;      /* per network, per block: inv[in], off[in] (BatchNorm folded: x*inv + off), kernel[in][out], bias[out] */
    int64_t stride;           /* floats between consecutive networks of an ensemble */
} SdcKstarMlp;
typedef struct {
    int n_lstm, n_bpw;        /* networks averaged (the reference's n_model_box, kstar_solver.py:156-162) */
    const float* lstm;        /* per network: bn0 inv, off [18]; K0 [18][400]; R0 [100][400]; b0 [400]; bn1 inv, off [100];
                                 K1 [100][400]; R1 [100][400]; b1 [400]   (Keras gate order i, f, c, o) */
    int64_t lstm_stride;      /* floats between networks, >= sdc_kstar_lstm_floats() */
    SdcKstarMlp head;         /* 100 -> ... -> 4 after the second LSTM (BatchNorm, Dense(50, sigmoid), BatchNorm, Dense(4)) */
    SdcKstarMlp steady;       /* kstar_nn, 17 -> 4: the steady-state first row */
    SdcKstarMlp bpw;          /* bpw_nn, 8 -> 2: (beta_p, W_mhd) */
    double lstm_ystd[4], lstm_ymean[4], nn_ystd[4], nn_ymean[4], bpw_ystd[2], bpw_ymean[2];
    double scale;             /* 10 ** np.log10(1000) as the host evaluates it (f2i / i2f, kstar_solver.py:35,107-113) */
    double inputs0[15];       /* i2f(f2i(input_init)), kstar_solver.py:84,150-152 */
    double low_action[9], high_action[9];
    double year_in;
} SdcKstarModel;
size_t sdc_kstar_lstm_floats(void);
/* actions: element (b, t, i) at actions[b*act_b_stride + t*act_t_stride + i*act_c_stride], t < nsteps, i < 9 (so a (B, C, T)
 * sample tensor is read in place).  out (B, nsteps + 1, 8) fp64 rows [bn, bp, h89, h98, q95, q0, li, wmhd]; work: 4 doubles. */
int sdc_kstar_rollout(const SdcKstarModel* m, const float* actions, int64_t act_b_stride, int64_t act_t_stride,
                      int64_t act_c_stride, double* out, double* work, int B, int nsteps, void* stream);

/* ---- Smoke score check: the fluid rollout behind InferencePipeline.multi_evaluate (2d/inference_2d.py:389-447 ->
 * 2d/dataset/apps/evaluate_solver.py:209-350 `solver`: per step get_envolve :82-111 = control ring + last interior velocity ->
 * divergence_free (phi/flow.py:317-326: float64 CG of phi/solver/base.py:63-103 on the matrix of phi/solver/sparse.py:27-77)
 * -> three semi-Lagrangian advections (phi/math/nd.py:407-428) -> bucket book-keeping :262-336).  Replaces one Python process
 * per sample with one launch: one 512-thread workgroup per sample, the 127 x 127 pressure problem on chip for the whole rollout.
 *   c1, c2          controls (B, nt, nx, nx) fp32: element (b, f, y, x) at [b*ctrl_b_stride + f*ctrl_f_stride + y*nx + x]
 *                   (so the channel slices pred[:, :, 3] / pred[:, :, 4] of a (B, nt, 7, nx, nx) sample are read in place)
 *   init_density    (B, nx, nx) fp32, batch stride dens_b_stride; init_velocity (128, 128, 2) fp32 staggered, batch stride
 *                   vel_b_stride (0 = one field for all samples, init_velocity_ evaluate_solver.py:77-79)
 *   fluid_mask      (127, 127) bytes, 1 = fluid / 0 = obstacle (build_obstacles_pi_128, evaluate_solver.py:29-60)
 *   bucket_labels, safe_labels   (128, 128) bytes: 0 = none, k = absorbing area k - 1 of get_bucket_mask / get_bucket_mask_safe
 *                   (evaluate_solver.py:114-178; areas must not overlap); n_buckets in [2, 8], n_safe in [1, 8]
 *   out             (B, nt, 7, nx, nx) fp64 = multi_evaluate's solver_out: density, velocity x, y, control x, y, smoke_out
 *                   record, safe record; out_zero (B, nt, nx, nx) fp64 = `zero_densitys`, may be null
 *   work            sdc_smoke_rollout_workspace_bytes(B) bytes, 16-byte aligned
 *   nx divides 128, nt divides per_timelength (256); ring_lo/ring_hi = 16/112: the interior the controls do not reach;
 *   accuracy / max_iterations = 1e-8 / 500 (evaluate_solver.py:108, phi/solver/sparse.py:88). */
size_t sdc_smoke_rollout_workspace_bytes(int B);
int sdc_smoke_rollout(const float* c1, const float* c2, int64_t ctrl_b_stride, int64_t ctrl_f_stride,
                      const float* init_density, int64_t dens_b_stride, const float* init_velocity, int64_t vel_b_stride,
                      const unsigned char* fluid_mask, const unsigned char* bucket_labels, const unsigned char* safe_labels,
                      int n_buckets, int n_safe, double* out, double* out_zero, void* work, size_t work_bytes, int B, int nt,
                      int nx, int per_timelength, int ring_lo, int ring_hi, double accuracy, int max_iterations, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SDC_H */
