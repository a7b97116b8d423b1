// Shared helpers for the libsdc_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <atomic>
#include "../../include/sdc.h"

namespace sdc {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return SDC_EHIP;
    }
    return SDC_OK;
}

#define SDC_REQUIRE(cond, code, ...)   \
    do {                               \
        if (!(cond)) {                 \
            sdc::set_error(__VA_ARGS__); \
            return (code);             \
        }                              \
    } while (0)

// Per-device one-time opt-in to more than 64 KB of dynamic LDS (hipFuncSetAttribute is per device).  The device bit is
// published only AFTER the attribute call has returned, so a second host thread on the same device either sees the bit (the
// attribute is applied) or repeats the idempotent call itself; a failed opt-in is reported here instead of surfacing later
// as an opaque launch error.  Keyed on the calling thread's current device, which is the device of the launch stream for
// every in-tree caller (one process per GPU, torch's current device).
inline int lds_optin(std::atomic<uint64_t>& mask, const void* kernel, int bytes, const char* what) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (mask.load(std::memory_order_acquire) & bit) return SDC_OK;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed: %s", what, bytes, hipGetErrorString(e));
        return SDC_EHIP;
    }
    mask.fetch_or(bit, std::memory_order_release);
    return SDC_OK;
}
#define SDC_LDS_OPTIN(mask, kernel, bytes, what)                                                             \
    do {                                                                                                     \
        const int rc_ = sdc::lds_optin((mask), reinterpret_cast<const void*>(kernel), (bytes), (what));      \
        if (rc_ != SDC_OK) return rc_;                                                                       \
    } while (0)

// Keeps the backend from pairing LDS accesses into ds_read2 / ds_write2 forms (function attribute, device pass only).  Their two
// offsets are 8-bit, so every pair further than 1 KB from its base register costs a VALU addition to form a new base -- 66 per
// 128 MFMAs in la_blk_out -- in kernels where a VALU instruction costs matrix-pipe time and an LDS instruction does not.
#if defined(__HIP_DEVICE_COMPILE__)
#define SDC_NO_DS_MERGE __attribute__((target("no-load-store-opt")))
#else
#define SDC_NO_DS_MERGE
#endif

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// v * sigmoid(v) with the reciprocal instruction (1 ulp) instead of the IEEE division sequence: 5 instructions instead of 14 -- in
// la_blk_ctx's GroupNorm-on-load tile (17 SiLUs per thread and tile) that was a quarter of the VALU work of a kernel in which every VALU
// instruction costs matrix-pipe time.  One definition for every kernel: the fused and the unfused paths stay bit-identical.
// (fp contract off: the product is rounded here, never fused into a caller's following add -- the GroupNorm-on-load tile adds the
// residual right behind it, the stand-alone apply kernel in another statement; hipcc's __fmul_rn is a plain multiplication)
__device__ __forceinline__ float silu_f(float v) {
#pragma clang fp contract(off)
    const float r = __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    return v * r;
}

// exp of a softmax exponent (x <= 0 after the row maximum is subtracted): v_mul + v_exp.  libm's expf is 13 instructions per
// element here (argument split, ldexp, two range guards) -- 208 of the ~440 VALU instructions of a temporal-attention head in
// ta_block_kernel, in a kernel where every VALU instruction costs matrix-pipe time.  The rounding of x * log2(e) leaves a
// relative error of |x| * 6e-8 in the term: terms that matter in a softmax have small |x|, and the row sum weighs the rest down
// (measured against fp64 in tests/test_gpu_kernels.py: no change in the fifth digit of the gates).
__device__ __forceinline__ float softmax_exp(float x) { return __expf(x); }

// wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace sdc
