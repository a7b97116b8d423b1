// What a GroupNorm-apply-shaped stream (read x [+ read residual], SiLU(x * a + b), write y) reaches on 1 GiB fp32 tensors,
// by access form: plain / nontemporal loads and stores, 1 / 2 / 4 float4 per thread and iteration, grid size.
// build: hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o tools/stream_probe ; run: tools/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float silu(float v) { return v / (1.0f + __expf(-v)); }
template <int NT_LOAD, int NT_STORE, int UN, bool RES>
__global__ __launch_bounds__(256) void k(const f4* __restrict__ x, const f4* __restrict__ r, f4* __restrict__ y, int64_t nv, float a, float b) {
    const int64_t stride = (int64_t)gridDim.x * 256 * UN;
    for (int64_t i0 = (int64_t)blockIdx.x * 256 * UN + threadIdx.x; i0 < nv; i0 += stride) {
        f4 v[UN], q[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t i = i0 + u * 256;
            v[u] = NT_LOAD ? __builtin_nontemporal_load(x + i) : x[i];
            if (RES) q[u] = NT_LOAD ? __builtin_nontemporal_load(r + i) : r[i];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            f4 o;
            o.x = silu(v[u].x * a + b); o.y = silu(v[u].y * a + b); o.z = silu(v[u].z * a + b); o.w = silu(v[u].w * a + b);
            if (RES) o += q[u];
            const int64_t i = i0 + u * 256;
            if (NT_STORE) __builtin_nontemporal_store(o, y + i); else y[i] = o;
        }
    }
}
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)
template <int L, int S, int UN, bool RES>
void run(const char* name, const f4* x, const f4* r, f4* y, int64_t nv, int blocks) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<L, S, UN, RES>), dim3(blocks), dim3(256), 0, 0, x, r, y, nv, 1.01f, 0.1f);
    CK(hipEventRecord(e0, 0));
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((k<L, S, UN, RES>), dim3(blocks), dim3(256), 0, 0, x, r, y, nv, 1.01f, 0.1f);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double by = (double)nv * 16 * (RES ? 3 : 2);
    printf("%-40s blocks %6d  %7.3f ms  %6.2f TB/s\n", name, blocks, ms, by / ms / 1e9);
}
int main() {
    const int64_t nv = (int64_t)64 * 1024 * 1024;      // 64 Mi float4 = 1 GiB
    f4 *x, *r, *y;
    CK(hipMalloc(&x, nv * 16)); CK(hipMalloc(&r, nv * 16)); CK(hipMalloc(&y, nv * 16));
    CK(hipMemset(x, 0, nv * 16)); CK(hipMemset(r, 0, nv * 16));
    for (int blocks : {4096, 16384, 65536, 262144}) {
        run<0, 0, 1, false>("plain un1", x, r, y, nv, blocks);
        run<0, 1, 1, false>("nt-store un1", x, r, y, nv, blocks);
        run<1, 1, 1, false>("nt-load nt-store un1", x, r, y, nv, blocks);
        run<0, 0, 2, false>("plain un2", x, r, y, nv, blocks);
        run<1, 1, 2, false>("nt both un2", x, r, y, nv, blocks);
        run<0, 0, 4, false>("plain un4", x, r, y, nv, blocks);
        run<1, 1, 4, false>("nt both un4", x, r, y, nv, blocks);
        run<0, 0, 1, true>("res plain un1", x, r, y, nv, blocks);
        run<1, 1, 1, true>("res nt both un1", x, r, y, nv, blocks);
        run<0, 0, 2, true>("res plain un2", x, r, y, nv, blocks);
        run<1, 1, 2, true>("res nt both un2", x, r, y, nv, blocks);
    }
    return 0;
}
