// Error reporting, version, hipGraph capture and event timing entry points.
#include "sdc_common.h"

namespace sdc {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace sdc


// Content stamp of a set of device spans (the parameters of a net): an order-independent 64-bit sum of a mixed
// (span, position, bits) word per 32-bit element.  The Python wrapper compares it with the stamp taken when a plan packed its
// weights: in-place writes through `.data` (EMA updates: 2d/video_diffusion_pytorch_conv3d.py:121-124, ema_pytorch) move no
// autograd version counter and may land on a recycled address, so only the content itself tells.
namespace sdc {
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void checksum_kernel(const SdcSpan* spans, unsigned long long* out) {
    const SdcSpan sp = spans[blockIdx.y];
    const uint32_t* w = reinterpret_cast<const uint32_t*>(sp.ptr);
    const uint64_t salt = mix64(0x9E3779B97F4A7C15ull * (uint64_t)(blockIdx.y + 1));
    uint64_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < sp.nwords; i += (int64_t)gridDim.x * 256)
        acc += mix64(salt ^ (((uint64_t)(i + 1) << 32) | (uint64_t)w[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)acc, o, 64), hi = __shfl_xor((uint32_t)(acc >> 32), o, 64);
        acc += ((uint64_t)hi << 32) | lo;
    }
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, (unsigned long long)acc);
}
}  // namespace sdc

extern "C" {

int sdc_version(void) { return 1; }

int sdc_last_error(char* buf, size_t cap) {
    size_t n = strlen(sdc::g_err);
    if (buf && cap) {
        size_t m = n < cap - 1 ? n : cap - 1;
        memcpy(buf, sdc::g_err, m);
        buf[m] = 0;
    }
    return (int)n;
}

#define HIP_TRY(expr)                                                    \
    do {                                                                 \
        hipError_t e_ = (expr);                                          \
        if (e_ != hipSuccess) {                                          \
            sdc::set_error("%s: %s", #expr, hipGetErrorString(e_));      \
            return SDC_EHIP;                                             \
        }                                                                \
    } while (0)

int sdc_checksum_spans(const SdcSpan* spans_dev, int n, uint64_t* out_dev, void* stream) {
    SDC_REQUIRE(spans_dev && out_dev, SDC_ENULL, "sdc_checksum_spans: null pointer");
    SDC_REQUIRE(n > 0 && n <= 65535, SDC_EINVAL, "sdc_checksum_spans: n = %d outside 1..65535", n);
    SDC_REQUIRE(((uintptr_t)out_dev & 7) == 0, SDC_EALIGN, "sdc_checksum_spans: out_dev is not 8-byte aligned");
    HIP_TRY(hipMemsetAsync(out_dev, 0, sizeof(uint64_t), sdc::as_stream(stream)));
    hipLaunchKernelGGL(sdc::checksum_kernel, dim3(32, (unsigned)n), dim3(256), 0, sdc::as_stream(stream), spans_dev,
                       reinterpret_cast<unsigned long long*>(out_dev));
    return sdc::check_launch("sdc_checksum_spans");
}

int sdc_graph_begin(void* stream) {
    HIP_TRY(hipStreamBeginCapture(sdc::as_stream(stream), hipStreamCaptureModeThreadLocal));
    return SDC_OK;
}

int sdc_graph_end(void* stream, void** graph_exec) {
    SDC_REQUIRE(graph_exec, SDC_ENULL, "sdc_graph_end: graph_exec is null");
    hipGraph_t g = nullptr;
    HIP_TRY(hipStreamEndCapture(sdc::as_stream(stream), &g));
    hipGraphExec_t ge = nullptr;
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) {
        sdc::set_error("hipGraphInstantiate: %s", hipGetErrorString(e));
        return SDC_EHIP;
    }
    *graph_exec = ge;
    return SDC_OK;
}

int sdc_graph_launch(void* graph_exec, void* stream) {
    SDC_REQUIRE(graph_exec, SDC_ENULL, "sdc_graph_launch: graph_exec is null");
    HIP_TRY(hipGraphLaunch((hipGraphExec_t)graph_exec, sdc::as_stream(stream)));
    return SDC_OK;
}

int sdc_graph_destroy(void* graph_exec) {
    if (graph_exec) HIP_TRY(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return SDC_OK;
}

int sdc_event_create(void** ev) {
    SDC_REQUIRE(ev, SDC_ENULL, "sdc_event_create: null");
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    *ev = e;
    return SDC_OK;
}
int sdc_event_record(void* ev, void* stream) {
    HIP_TRY(hipEventRecord((hipEvent_t)ev, sdc::as_stream(stream)));
    return SDC_OK;
}
int sdc_event_elapsed_ms(void* ev0, void* ev1, float* ms) {
    HIP_TRY(hipEventSynchronize((hipEvent_t)ev1));
    HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)ev0, (hipEvent_t)ev1));
    return SDC_OK;
}
int sdc_event_destroy(void* ev) {
    if (ev) HIP_TRY(hipEventDestroy((hipEvent_t)ev));
    return SDC_OK;
}

}  // extern "C"
