// How fast does a bare v_mfma_f32_32x32x2_f32 loop run when it runs for SECONDS (the sampler's C4 step keeps the matrix
// cores busy for minutes), and what do the clock and the socket power do meanwhile?  Prints one line per ~0.25 s window:
// TFLOP/s of the window.  tools/mfma_sustain.py runs this as a child and samples the amdgpu hwmon files beside it.
// usage: mfma_sustain <seconds> <0 constant | 1 random operands>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_loop(float* out, const float* in, int iters) {
    extern __shared__ float occupancy_lds[];      // dynamic LDS only limits how many workgroups share a CU (argument 4)
    if (iters < 0) occupancy_lds[threadIdx.x] = 0.f;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(threadIdx.x * 16 + i) & 4095]; b[i] = in[(threadIdx.x * 16 + 8 + i) & 4095]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[(k + i) & 7], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// the same FLOPs through v_mfma_f32_16x16x4_f32 (8 passes, 2048 FLOP per instruction): is the sustained rate a property of the shape?
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mfma_loop16(float* out, const float* in, int iters) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i)
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(threadIdx.x * 16 + i) & 4095]; b[i] = in[(threadIdx.x * 16 + 8 + i) & 4095]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(k + i) & 7], b[(k + 2 * i) & 7], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i)
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    const int mode = argc > 2 ? atoi(argv[2]) : 1;
    const int shape = argc > 3 ? atoi(argv[3]) : 0;          // 0: 32x32x2, 1: 16x16x4 (same FLOPs per launch)
    const int wg_per_cu = argc > 4 ? atoi(argv[4]) : 4;      // workgroups (= waves per SIMD) sharing a CU: 4 (default), 3, 2 or 1, enforced through LDS
    const size_t lds = wg_per_cu >= 4 ? 0 : (wg_per_cu == 3 ? 50 * 1024 : (wg_per_cu == 2 ? 72 * 1024 : 96 * 1024));
    const int blocks = 1024, iters = 4000;             // ~17 ms per launch at 125 TFLOP/s
    float *out, *in;
    (void)hipMalloc(&out, blocks * 256 * 4);
    (void)hipMalloc(&in, 4096 * 4);
    static float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = mode ? (float)rand() / RAND_MAX * 2.f - 1.f : 0.5f;
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const double flops_per_launch = (double)blocks * 4.0 * iters * 8 * 4 * 4096.0;
    const auto t_begin = std::chrono::steady_clock::now();
    if (lds) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_loop), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    printf("mode %s operands, %s, %d workgroup(s) of 4 waves per CU\n", mode ? "random" : "constant", shape ? "v_mfma_f32_16x16x4_f32" : "v_mfma_f32_32x32x2_f32", wg_per_cu);
    for (;;) {
        (void)hipEventRecord(e0, 0);
        for (int r = 0; r < 15; ++r) {
            if (shape == 0) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), lds, 0, out, in, iters);
            else hipLaunchKernelGGL(mfma_loop16, dim3(blocks), dim3(256), 0, 0, out, in, iters);       // 64 x 2048 = 32 x 4096 FLOP per iteration
        }
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        printf("t=%.2f s  window %.1f ms  %.1f TFLOP/s\n", t, ms, 15 * flops_per_launch / (ms * 1e-3) / 1e12);
        fflush(stdout);
        if (t > seconds) break;
    }
    return 0;
}
