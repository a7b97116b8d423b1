// Calibration: what a bare v_mfma_f32_32x32x2_f32 loop sustains on this device, with trivial operands and with
// random operands (DVFS: the clock the chip holds under load depends on the data, MI355X_MICROARCH.md).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, const float* in, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(threadIdx.x * 16 + i) & 4095]; b[i] = in[(threadIdx.x * 16 + 8 + i) & 4095]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[(k + i) & 7], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// same FLOPs through v_mfma_f32_16x16x4_f32 (8 passes, 4 accumulator registers): does the smaller shape hold a higher clock?
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma16_loop(float* out, const float* in, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(threadIdx.x * 16 + i) & 4095]; b[i] = in[(threadIdx.x * 16 + 8 + i) & 4095]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b[(k + i) & 7], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    const int blocks = 1024, iters = 400;
    float *out, *in;
    (void)hipMalloc(&out, blocks * 256 * 4);
    (void)hipMalloc(&in, 4096 * 4);
    float h[4096];
    for (int mode = 0; mode < 2; ++mode) {
        for (int i = 0; i < 4096; ++i) h[i] = mode ? (float)rand() / RAND_MAX * 2.f - 1.f : 0.5f;
        (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
        (void)hipEventRecord(e0, 0);
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double flops = 10.0 * blocks * 4.0 * iters * 8 * 4 * 4096.0;
        printf("32x32x2 %s operands: %.3f ms/launch  %.1f TFLOP/s\n", mode ? "random  " : "constant", ms / 10, flops / (ms * 1e-3) / 1e12);
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(mfma16_loop<16>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
        (void)hipEventRecord(e0, 0);
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(mfma16_loop<16>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        flops = 10.0 * blocks * 4.0 * iters * 8 * 16 * 2048.0;
        printf("16x16x4 %s operands: %.3f ms/launch  %.1f TFLOP/s\n", mode ? "random  " : "constant", ms / 10, flops / (ms * 1e-3) / 1e12);
    }
    return 0;
}
