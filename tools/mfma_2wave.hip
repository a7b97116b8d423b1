// Micro-benchmark: are VALU instructions of ONE wave hidden behind the fp32 MFMAs of ANOTHER wave on the same SIMD?
// tools/mfma_fill.hip showed that a filler between two v_mfma_f32_32x32x2_f32 of the same wave costs its full issue time.
// Here a workgroup has WPS waves per SIMD (256 * WPS threads), each with NACC independent 32x32 accumulators, and N filler
// instructions behind each MFMA; a second experiment lets every wave alternate MFMA phases with VALU-only phases (the shape of
// a conv pass + its fold), the two waves of a SIMD out of phase.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_2wave.hip -o tools/mfma_2wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int WPS, int NACC, int KIND, int N>
__global__ __launch_bounds__(256 * WPS) void fill_loop(float* out, const float* in, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int tid = threadIdx.x;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[(tid * 8 + i) & 4095]; b[i] = in[(tid * 8 + 4 + i) & 4095]; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = in[(tid + i * 64) & 4095];
    f32x2 p2[4];
    for (int i = 0; i < 4; ++i) p2[i] = f32x2{in[(tid + i) & 4095], in[(tid + 7 * i) & 4095]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < NACC; ++c) {
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c & 3], b[(c >> 2) & 3], acc[c], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < N; ++n) {
                if (KIND == 1) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(v[(n + 4) & 7]) : "v"(v[n & 3]), "v"(v[(n + 1) & 3]));
                else if (KIND == 2) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p2[(n + 2) & 3]) : "v"(p2[n & 1]), "v"(p2[(n + 1) & 1]));
                else if (KIND == 3) asm volatile("v_sub_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(v[(n + 4) & 7]) : "v"(v[n & 3]), "v"(v[(n + 1) & 3]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += p2[i][0] + p2[i][1];
    out[blockIdx.x * 256 * WPS + tid] = s;
}

// phases: PM MFMAs, then PV packed VALU ops (reading the accumulators like a fold), repeated; waves 4..7 start with the VALU phase
template <int WPS, int NACC, int PV>
__global__ __launch_bounds__(256 * WPS) void phase_loop(float* out, const float* in, int iters, int mfma_rounds) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int tid = threadIdx.x;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[(tid * 8 + i) & 4095]; b[i] = in[(tid * 8 + 4 + i) & 4095]; }
    f32x2 p2[4];
    for (int i = 0; i < 4; ++i) p2[i] = f32x2{in[(tid + i) & 4095], in[(tid + 7 * i) & 4095]};
    const bool late = WPS == 2 && tid >= 256;
    auto mf = [&]() {
        for (int k = 0; k < mfma_rounds; ++k) {
#pragma unroll
            for (int c = 0; c < NACC; ++c) {
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c & 3], b[(c >> 2) & 3], acc[c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto va = [&]() {
#pragma unroll 16
        for (int n = 0; n < PV; ++n)
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p2[(n + 2) & 3]) : "v"(p2[n & 1]), "v"(p2[(n + 1) & 1]));
    };
    if (late) va();
    for (int it = 0; it < iters; ++it) { mf(); va(); }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 4; ++i) s += p2[i][0] + p2[i][1];
    out[blockIdx.x * 256 * WPS + tid] = s;
}

template <typename F>
float timeit(F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch();
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r) launch();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
}

template <int WPS, int NACC, int KIND, int N>
float run(float* out, const float* in, int blocks, int iters) {
    return timeit([&]() { hipLaunchKernelGGL((fill_loop<WPS, NACC, KIND, N>), dim3(blocks), dim3(256 * WPS), 0, 0, out, in, iters); });
}

int main() {
    const int blocks = 1024;
    float *out, *in;
    (void)hipMalloc(&out, blocks * 512 * 4);
    (void)hipMalloc(&in, 16384 * 4);
    float h[16384];
    for (int i = 0; i < 16384; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    // same MFMA count per SIMD in both forms: 1 wave x 16 acc x iters  ==  2 waves x 8 acc x iters
    const int iters = 300;
    const double nm = 4.0 * iters * 16;          // MFMAs per SIMD per launch (4 workgroups per CU in sequence)
    auto tf = [&](float ms) { return 2.0 * 32 * 32 * 2 * nm * 1024 / (ms * 1e-3) / 1e12; };
#define ROW(K, name)                                                                                                             \
    {                                                                                                                            \
        const float a0 = run<1, 16, K, 0>(out, in, blocks, iters), a1 = run<1, 16, K, 1>(out, in, blocks, iters),                   \
                    a2 = run<1, 16, K, 2>(out, in, blocks, iters), a4 = run<1, 16, K, 4>(out, in, blocks, iters);                   \
        const float b0 = run<2, 8, K, 0>(out, in, blocks, iters), b1 = run<2, 8, K, 1>(out, in, blocks, iters),                     \
                    b2 = run<2, 8, K, 2>(out, in, blocks, iters), b4 = run<2, 8, K, 4>(out, in, blocks, iters);                     \
        printf("%-12s 1 wave/SIMD x16 acc: N=0 %.3f ms (%.1f TF/s) N=1 %.3f N=2 %.3f N=4 %.3f (%.1f TF/s)\n", name, a0, tf(a0), a1, a2, a4, tf(a4)); \
        printf("%-12s 2 waves/SIMD x8 acc: N=0 %.3f ms (%.1f TF/s) N=1 %.3f N=2 %.3f N=4 %.3f (%.1f TF/s)\n", name, b0, tf(b0), b1, b2, b4, tf(b4)); \
    }
    ROW(1, "v_sub_f32")
    ROW(2, "v_pk_add_f32")
    ROW(3, "v_sub_dpp")
    // phases: 512 MFMAs per SIMD and phase (1 wave: 32 rounds of 16; 2 waves: 32 rounds of 8 each ... twice as many phases)
    {
        const int it1 = 8;
        const float p1 = timeit([&]() { hipLaunchKernelGGL((phase_loop<1, 16, 0>), dim3(blocks), dim3(256), 0, 0, out, in, it1, 32); });
        const float p2 = timeit([&]() { hipLaunchKernelGGL((phase_loop<1, 16, 560>), dim3(blocks), dim3(256), 0, 0, out, in, it1, 32); });
        const float q1 = timeit([&]() { hipLaunchKernelGGL((phase_loop<2, 8, 0>), dim3(blocks), dim3(512), 0, 0, out, in, it1, 32); });
        const float q2 = timeit([&]() { hipLaunchKernelGGL((phase_loop<2, 8, 280>), dim3(blocks), dim3(512), 0, 0, out, in, it1, 32); });
        printf("phases (512 MFMAs per SIMD and phase, then a VALU-only fold): 1 wave/SIMD: no fold %.3f ms, 560-op fold %.3f ms (+%.1f %%)\n", p1, p2, (p2 / p1 - 1) * 100);
        printf("                                                              2 waves/SIMD (out of phase): no fold %.3f ms, 280-op folds %.3f ms (+%.1f %%)\n", q1, q2, (q2 / q1 - 1) * 100);
    }
    return 0;
}
