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

// Per-device one-time set-up (hipFuncSetAttribute for > 64 KB of dynamic LDS is per device): true the first time the
// calling thread's current device is seen.  Racing threads may both see "first"; the guarded call is idempotent.
inline bool first_use_on_device(std::atomic<uint64_t>& mask) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (mask.load(std::memory_order_relaxed) & bit) return false;
    mask.fetch_or(bit, std::memory_order_relaxed);
    return true;
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }

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
