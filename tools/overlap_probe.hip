// Can an HBM-bound pass (GroupNorm apply: 11 ms of a 253 ms C4 step, plus channel norms and the temporal attention core: 14 ms
// together) hide behind an MFMA-bound kernel of ANOTHER half of the batch?  A 512-register MFMA wave leaves no room on its SIMD,
// so the two kernels cannot share a CU; they can share the chip: the stream kernel on a quarter of the CUs (if a quarter of the
// CUs can still saturate HBM), the MFMA kernel on the rest.  This probe measures exactly that with CU-masked streams
// (hipExtStreamCreateWithCUMask): the MFMA loop of tools/mfma_sustain.hip beside a 1 GiB read + write stream, alone, on their
// CU shares, and together.  An experiment for DESIGN section 7 ("what comes next"); not part of the product.
// build: hipcc --offload-arch=gfx950 -O3 tools/overlap_probe.hip -o tools/overlap_probe ; run: tools/overlap_probe [stream CUs per 8, default 2]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

// MFMA-bound: one wave per SIMD (256 threads), LDS sized so that no second such workgroup fits a CU
__global__ __launch_bounds__(256) void mfma_loop(float* out, const float* in, int iters) {
    extern __shared__ float lds[];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(threadIdx.x * 16 + i) & 4095]; b[i] = in[(threadIdx.x * 16 + 8 + i) & 4095]; }
    if (threadIdx.x == 0) lds[0] = a[0];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[(k + i) & 7], acc[i], 0, 0, 0);
    }
    float s = lds[0];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// HBM-bound: y = silu(x a + b), UN 16-byte vectors in flight per thread, grid-stride (few CUs must keep many bytes in flight)
__device__ __forceinline__ float silu(float v) { return v / (1.0f + __expf(-v)); }
template <int UN>
__global__ __launch_bounds__(256) void stream_k(const f4* __restrict__ x, f4* __restrict__ y, int64_t nv, float a, float b) {
    const int64_t stride = (int64_t)gridDim.x * 256 * UN;
    for (int64_t i0 = (int64_t)blockIdx.x * 256 * UN + threadIdx.x; i0 < nv; i0 += stride) {
        f4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) v[u] = __builtin_nontemporal_load(x + i0 + u * 256);
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            f4 o;
            o.x = silu(v[u].x * a + b); o.y = silu(v[u].y * a + b); o.z = silu(v[u].z * a + b); o.w = silu(v[u].w * a + b);
            __builtin_nontemporal_store(o, y + i0 + u * 256);
        }
    }
}

int main(int argc, char** argv) {
    const int share = argc > 1 ? atoi(argv[1]) : 2;          // stream kernel's CUs out of every 8
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    std::vector<uint32_t> mA((ncu + 31) / 32, 0), mB((ncu + 31) / 32, 0), mAll((ncu + 31) / 32, 0);
    int nA = 0, nB = 0;
    for (int i = 0; i < ncu; ++i) {
        mAll[i / 32] |= 1u << (i % 32);
        if ((i % 8) < share) { mB[i / 32] |= 1u << (i % 32); ++nB; } else { mA[i / 32] |= 1u << (i % 32); ++nA; }
    }
    hipStream_t sA, sB, sAll, sAll2;
    CK(hipExtStreamCreateWithCUMask(&sA, (uint32_t)mA.size(), mA.data()));
    CK(hipExtStreamCreateWithCUMask(&sB, (uint32_t)mB.size(), mB.data()));
    CK(hipStreamCreate(&sAll));
    CK(hipStreamCreate(&sAll2));
    const int64_t nv = (int64_t)64 * 1024 * 1024;           // 1 GiB
    f4 *x, *y;
    float *out, *in;
    CK(hipMalloc(&x, nv * 16)); CK(hipMalloc(&y, nv * 16)); CK(hipMalloc(&out, 4096 * 256 * 4)); CK(hipMalloc(&in, 4096 * 4));
    CK(hipMemset(x, 0, nv * 16));
    {   // random operands: an MFMA loop on zeros draws far less power and is not throttled (155.9 TFLOP/s for seconds)
        static float h[4096];
        srand(5);
        for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / (float)RAND_MAX * 2.f - 1.f;
        CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    }
    const int mblocks = 2048, miters = 1000;                // MFMA kernel: 2048 workgroups, ~4 ms on the whole chip
    const size_t mlds = 96 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_loop), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
    const double mflops = (double)mblocks * 4.0 * miters * 8 * 4 * 4096.0;
    auto run_m = [&](hipStream_t s) { hipLaunchKernelGGL(mfma_loop, dim3(mblocks), dim3(256), mlds, s, out, in, miters); };
    auto run_s = [&](hipStream_t s, int blocks) { hipLaunchKernelGGL(stream_k<8>, dim3(blocks), dim3(256), 0, s, x, y, nv, 1.01f, 0.1f); };
    auto wall = [&](auto fn, int reps) {
        fn();
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, sAll));
        CK(hipStreamSynchronize(sAll));
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r) fn();
        CK(hipDeviceSynchronize());
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
    };
    printf("%d CUs; MFMA share %d CUs, stream share %d CUs\n", ncu, nA, nB);
    const double tm_all = wall([&] { run_m(sAll); }, 200);
    printf("MFMA kernel alone, all CUs            %7.3f ms  %6.1f TFLOP/s\n", tm_all, mflops / tm_all / 1e9);
    const double tm_a = wall([&] { run_m(sA); }, 200);
    printf("MFMA kernel alone, its share          %7.3f ms  %6.1f TFLOP/s\n", tm_a, mflops / tm_a / 1e9);
    for (int blocks : {ncu * 8, nB * 8, nB * 16}) {
        const double ts_all = wall([&] { run_s(sAll, blocks); }, 10);
        const double ts_b = wall([&] { run_s(sB, blocks); }, 10);
        printf("stream kernel alone, %5d blocks: all CUs %6.3f ms %5.2f TB/s | its share %6.3f ms %5.2f TB/s\n", blocks, ts_all, nv * 32.0 / ts_all / 1e9,
               ts_b, nv * 32.0 / ts_b / 1e9);
    }
    // one MFMA kernel and K stream kernels: back to back on one stream, on two unmasked streams, on the two CU shares
    for (int K : {4, 8}) {
        const int sb = nB * 16;
        const double serial = wall([&] { run_m(sAll); for (int k = 0; k < K; ++k) run_s(sAll, ncu * 8); }, 150);      // (~1.5 s each: the sustained regime)
        const double plain2 = wall([&] { run_m(sAll); for (int k = 0; k < K; ++k) run_s(sAll2, ncu * 8); }, 150);
        const double masked = wall([&] { run_m(sA); for (int k = 0; k < K; ++k) run_s(sB, sb); }, 150);
        printf("1 MFMA kernel + %d stream kernels: one stream %7.3f ms | two plain streams %7.3f ms | CU-masked shares %7.3f ms\n", K, serial, plain2, masked);
    }
    return 0;
}
