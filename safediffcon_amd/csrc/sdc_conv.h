// Shared declarations of the conv translation units (sdc_conv.hip: direct and F(2,3) kernels + dispatch; sdc_conv_wino.hip:
// the Winograd F(2x2,3x3) and F(2x2x2,3x3x3) kernels).  gfx950 only.
#pragma once
#include "sdc_common.h"
#include <cstdlib>
#include <type_traits>

namespace sdcconv {


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
    // conv_wg2_kernel, sdc_conv_splitk only: ksplit workgroups share an output tile, each over Cin / ksplit input channels; y is then
    // the partial buffer [ksplit][B][Cout][oD][oH][oW] (d.ys its dense strides), ypart_elems the size of one partial copy
    int ksplit;
    int64_t ypart_elems;
};

// Position-tile numbering: workgroups are dealt round-robin over the 8 XCDs, each with a private L2.  Neighbouring
// position tiles share their halo rows (kh / kd taps), so consecutive LOGICAL tiles are given to one XCD
// (bijective when the tile count is a multiple of 8; speed / HBM traffic only, never correctness).
__device__ __forceinline__ int xcd_tile(int bid, int ntiles) {
    return (ntiles & 7) == 0 ? (bid & 7) * (ntiles >> 3) + (bid >> 3) : bid;
}

constexpr int W2_SK = 8;          // channels per stage
constexpr int W2_BM = 64;         // output channels per workgroup
constexpr int W2_TILES = 64;      // 2x2 tiles per workgroup
constexpr int W2_NBUF = 2;        // LDS stage buffers
constexpr int W2_ASZ = 16 * W2_SK * W2_BM, W2_BSZ = 16 * W2_SK * W2_TILES;    // floats per stage: U tile, V tile
constexpr int W3S_TILES = 32;     // 2x2 tiles per workgroup of conv_wg3s_kernel (two workgroups per CU)

inline int64_t span5(const int64_t* st, int b, int c, int dd, int h, int w) {
    return (int64_t)(b - 1) * st[0] + (int64_t)(c - 1) * st[1] + (int64_t)(dd - 1) * st[2] + (int64_t)(h - 1) * st[3] + (int64_t)(w - 1) * st[4];
}

// defined in sdc_conv_wino.hip
bool wg2_ok(const SdcConvDesc& d, bool small, bool rowhalo);
bool wg3_ok(const SdcConvDesc& d, bool small, bool rowhalo);
int launch_wg2(const ConvArgs& a, hipStream_t s);
int wg2_ksplit(const SdcConvDesc& d);         // Cin split sdc_conv_splitk would use for this conv (1: none)
int launch_wg3(const ConvArgs& a, hipStream_t s);
bool wg3s_ok(const SdcConvDesc& d, bool small, bool rowhalo);
bool wg2s_ok(const SdcConvDesc& d, bool small, bool rowhalo);
int launch_wg2s(const ConvArgs& a, hipStream_t s);
int launch_wg3s(const ConvArgs& a, hipStream_t s);

}  // namespace sdcconv
