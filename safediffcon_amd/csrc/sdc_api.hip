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
