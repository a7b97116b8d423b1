// What would an fp32 GEMM gain on this chip if its products ran on the bf16 matrix pipe as three-way operand splits?
//
//   a = a1 + a2 + a3,  b = b1 + b2 + b3   (each piece a bf16: 8 significant bits, so three pieces carry the 24 of an fp32)
//   a b ~= a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1)        six bf16 MFMAs, fp32 accumulate; dropped terms <= 2^-24 |a b|
//
// v_mfma_f32_32x32x16_bf16 runs 16x the multiply-adds per cycle of v_mfma_f32_32x32x2_f32, so six of them per fp32 product
// leave 2.67x -- if the LDS traffic (1.5x the bytes), the split arithmetic at park time and the chip's sustained-rate
// ceiling (the fp32 pipe holds 123.5 of 157.3 TFLOP/s) let it through.  This probe measures exactly that on a
// plain C = A B (A: weights, split once on the host side of the timed region; B: activations, split inside the kernel as a
// conv kernel would at park time), beside the same tiling on the fp32 pipe, and reports both errors against fp64.
// NOT part of the product: an experiment for DESIGN section 7 ("what comes next").
// build: hipcc --offload-arch=gfx950 -O3 tools/split_gemm_probe.hip -o tools/split_gemm_probe ; run: tools/split_gemm_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int TM = 128, TN = 128, TK = 16, NT = 256;

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    float r = x - (float)h;
    m = (__bf16)r;
    r -= (float)m;
    l = (__bf16)r;
}

// A planes [3][K/16][M][16] bf16 (pre-split AND tile-packed like a conv weight: a stage's 128 x 16 tile is one contiguous 4 KB run;
// row-major [M][K] planes cost a quarter-used 128-byte line per lane pair and ran at 108 TFLOP/s), B [K][N] fp32, C [M][N] fp32.  NTERMS: 6 (fp32-grade) or 3 (a1 b1 + a1 b2 + a2 b1: 16-bit grade)
// PRE: B arrives pre-split and packed the same way (planes [3][K/16][N][16]): a bound without split arithmetic / transposing park
template <int NTERMS, bool PRE = false>
__global__ __launch_bounds__(NT) void gemm_split(const __bf16* __restrict__ A3, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K,
                                                 const __bf16* __restrict__ BT3 = nullptr) {
    __shared__ __attribute__((aligned(16))) __bf16 As[2][3][TM][TK];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2][3][TN][TK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, r31 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    const int am = tid >> 1, ah = tid & 1;            // A park: row, 8-k half
    const int bn = tid & 127, bh = tid >> 7;          // B park: column, 8-k half
    const size_t plane = (size_t)M * K;
    bf16x8 areg[3], bpre[3];
    float breg[8];
    const size_t bplane = (size_t)N * K;
    auto load = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 3; ++p) areg[p] = *reinterpret_cast<const bf16x8*>(A3 + p * plane + ((size_t)(k0 >> 4) * M + m0) * 16 + tid * 8);
        if (PRE) {
#pragma unroll
            for (int p = 0; p < 3; ++p) bpre[p] = *reinterpret_cast<const bf16x8*>(BT3 + p * bplane + ((size_t)(k0 >> 4) * N + n0) * 16 + tid * 8);
            return;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) breg[j] = B[(size_t)(k0 + 8 * bh + j) * N + n0 + bn];
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(&As[buf][p][am][8 * ah]) = areg[p];
        if (PRE) {
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(&Bs[buf][p][am][8 * ah]) = bpre[p];
            return;
        }
        bf16x8 b1, b2, b3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __bf16 x, y, z;
            split3(breg[j], x, y, z);
            b1[j] = x; b2[j] = y; b3[j] = z;
        }
        *reinterpret_cast<bf16x8*>(&Bs[buf][0][bn][8 * bh]) = b1;
        *reinterpret_cast<bf16x8*>(&Bs[buf][1][bn][8 * bh]) = b2;
        *reinterpret_cast<bf16x8*>(&Bs[buf][2][bn][8 * bh]) = b3;
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int S = K / TK;
    load(0);
    park(0);
    __syncthreads();
    for (int s = 0; s < S; ++s) {
        const int buf = s & 1;
        if (s + 1 < S) load((s + 1) * TK);
        bf16x8 fa[2][3], fb[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                fa[i][p] = *reinterpret_cast<const bf16x8*>(&As[buf][p][wm * 64 + i * 32 + r31][8 * h]);
                fb[i][p] = *reinterpret_cast<const bf16x8*>(&Bs[buf][p][wn * 64 + i * 32 + r31][8 * h]);
            }
        // smallest terms first
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (NTERMS == 6) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], acc[i][j], 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
            }
        if (s + 1 < S) park(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, n = n0 + wn * 64 + j * 32 + r31;
                C[(size_t)m * N + n] = acc[i][j][r];
            }
}

// the same tiling on the fp32 pipe: A [M][K] fp32, LDS images [k][m] / [k][n]
__global__ __launch_bounds__(NT) void gemm_f32(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) float As[2][TK][TM + 4];
    __shared__ __attribute__((aligned(16))) float Bs[2][TK][TN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, r31 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    f32x4 areg[2], breg[2];
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + i * NT;                       // A: row f >> 2, k quad f & 3;  B: k row f >> 5, n quad f & 31
            areg[i] = *reinterpret_cast<const f32x4*>(A + (size_t)(m0 + (f >> 2)) * K + k0 + 4 * (f & 3));
            breg[i] = *reinterpret_cast<const f32x4*>(B + (size_t)(k0 + (f >> 5)) * N + n0 + 4 * (f & 31));
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + i * NT;
#pragma unroll
            for (int j = 0; j < 4; ++j) As[buf][4 * (f & 3) + j][f >> 2] = areg[i][j];
            *reinterpret_cast<f32x4*>(&Bs[buf][f >> 5][4 * (f & 31)]) = breg[i];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int S = K / TK;
    load(0);
    park(0);
    __syncthreads();
    for (int s = 0; s < S; ++s) {
        const int buf = s & 1;
        if (s + 1 < S) load((s + 1) * TK);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            float fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = As[buf][2 * ks + h][wm * 64 + i * 32 + r31];
                fb[i] = Bs[buf][2 * ks + h][wn * 64 + i * 32 + r31];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < S) park(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, n = n0 + wn * 64 + j * 32 + r31;
                C[(size_t)m * N + n] = acc[i][j][r];
            }
}

// X [R][K] fp32 row-major -> planes [3][K/16][R][16]
__global__ void split_planes(const float* X, __bf16* X3, int R, int K) {
    const size_t n = (size_t)R * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / K), k = (int)(i - (size_t)row * K);
        const size_t o = ((size_t)(k >> 4) * R + row) * 16 + (k & 15);
        __bf16 x, y, z;
        split3(X[i], x, y, z);
        X3[o] = x; X3[n + o] = y; X3[2 * n + o] = z;
    }
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 8192, N = argc > 2 ? atoi(argv[2]) : 8192, K = argc > 3 ? atoi(argv[3]) : 2048;
    const int reps = argc > 4 ? atoi(argv[4]) : 20;
    std::vector<float> hA((size_t)M * K), hB((size_t)K * N);
    srand(7);
    for (auto& v : hA) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.1f;
    for (auto& v : hB) v = (float)rand() / RAND_MAX * 2.f - 1.f + 0.25f;
    float *A, *B, *C;
    __bf16* A3;
    CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&B, hB.size() * 4)); CK(hipMalloc(&C, (size_t)M * N * 4)); CK(hipMalloc(&A3, hA.size() * 6));
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(split_planes, dim3(4096), dim3(256), 0, 0, A, A3, M, K);
    CK(hipDeviceSynchronize());
    const dim3 grid(N / TN, M / TM);
    std::vector<float> hC((size_t)M * N);
    // fp64 reference on a sample of rows
    const int nrows = 8;
    std::vector<double> ref((size_t)nrows * N);
    double scale = 0;
    for (int i = 0; i < nrows; ++i) {
        const int m = (int)((size_t)i * 1031 % M);
        for (int n = 0; n < N; ++n) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)hA[(size_t)m * K + k] * (double)hB[(size_t)k * N + n];
            ref[(size_t)i * N + n] = s;
            scale = fmax(scale, fabs(s));
        }
    }
    auto check = [&](const char* name) {
        CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
        double e = 0, e2 = 0;
        for (int i = 0; i < nrows; ++i) {
            const int m = (int)((size_t)i * 1031 % M);
            for (int n = 0; n < N; ++n) {
                const double d = hC[(size_t)m * N + n] - ref[(size_t)i * N + n];
                e = fmax(e, fabs(d));
                e2 += d * d;
            }
        }
        printf("%-44s max|err| %.3e  rms %.3e  (relative to the output scale %.3f: %.2e / %.2e)\n", name, e, sqrt(e2 / (nrows * (double)N)), scale, e / scale,
               sqrt(e2 / (nrows * (double)N)) / scale);
    };
    auto timeit = [&](const char* name, auto launch) {
        launch();
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%-44s %8.3f ms  %7.1f TFLOP/s (fp32-equivalent, 2 M N K)\n", name, ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
        check(name);
    };
    printf("C = A B, M %d N %d K %d, tile 128 x 128 x 16, 4 waves\n", M, N, K);
    timeit("fp32 pipe  (v_mfma_f32_32x32x2_f32)", [&] { hipLaunchKernelGGL(gemm_f32, grid, dim3(NT), 0, 0, A, B, C, M, N, K); });
    // the same kernel with its occupancy cut to two workgroups per CU by an unused dynamic-LDS request (two waves per SIMD: DESIGN 3.5)
    timeit("fp32 pipe, two workgroups per CU", [&] { hipLaunchKernelGGL(gemm_f32, grid, dim3(NT), 30 * 1024, 0, A, B, C, M, N, K); });
    timeit("bf16 pipe, 6 products per fp32 product", [&] { hipLaunchKernelGGL(gemm_split<6>, grid, dim3(NT), 0, 0, A3, B, C, M, N, K); });
    {   // B^T planes for the pre-split bound
        std::vector<float> hBT((size_t)N * K);
        for (int k = 0; k < K; ++k)
            for (int n = 0; n < N; ++n) hBT[(size_t)n * K + k] = hB[(size_t)k * N + n];
        float* BT; __bf16* BT3;
        CK(hipMalloc(&BT, hBT.size() * 4)); CK(hipMalloc(&BT3, hBT.size() * 6));
        CK(hipMemcpy(BT, hBT.data(), hBT.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(split_planes, dim3(4096), dim3(256), 0, 0, BT, BT3, N, K);
        CK(hipDeviceSynchronize());
        timeit("bf16 pipe, 6 products, B pre-split (bound)", [&] { hipLaunchKernelGGL((gemm_split<6, true>), grid, dim3(NT), 0, 0, A3, B, C, M, N, K, BT3); });
    }
    timeit("bf16 pipe, 3 products (16-bit grade)", [&] { hipLaunchKernelGGL(gemm_split<3>, grid, dim3(NT), 0, 0, A3, B, C, M, N, K); });
    return 0;
}
