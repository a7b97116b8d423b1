// Micro-benchmark: what does an instruction issued between two v_mfma_f32_32x32x2_f32 of ONE wave per SIMD cost?
// 16 independent 32x32 accumulators (256 registers, like conv_wg2_kernel), 16 MFMAs per block of the loop, N filler
// instructions of one kind behind each MFMA.  Reports ns and cycles-equivalent per filler relative to the bare loop.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_fill.hip -o tools/mfma_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// KIND: 0 none, 1 v_sub_f32 (independent regs), 2 ds_read_b128, 3 ds_read2_b64, 4 ds_write_b32, 5 ds_write_b128,
//       6 global_load_dword (L2 hit), 7 s_add_u32, 8 v_sub_f32 chain on freshly read LDS data (with waits), 9 ds_read_b32, 10 s_nop
template <int KIND, int N>
__global__ __launch_bounds__(256) void fill_loop(float* out, const float* in, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int tid = threadIdx.x;
    for (int i = tid; i < 16384; i += 256) lds[i] = in[i & 4095];
    __syncthreads();
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[(tid * 8 + i) & 4095]; b[i] = in[(tid * 8 + 4 + i) & 4095]; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = in[(tid + i * 64) & 4095];
    f32x4 r4[4];
    for (int i = 0; i < 4; ++i) r4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint32_t laddr = (uint32_t)(tid * 16);            // bytes, conflict-free b128
    const float* gp = in + tid;
    const uint32_t goff = (uint32_t)(tid * 16);
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 p2[4];
    for (int i = 0; i < 4; ++i) p2[i] = f32x2{in[(tid + i) & 4095], in[(tid + 7 * i) & 4095]};
    uint32_t sacc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c & 3], b[(c >> 2) & 3], acc[c], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < N; ++n) {
                if (KIND == 1) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(v[(n + 4) & 7]) : "v"(v[n & 3]), "v"(v[(n + 1) & 3]));
                else if (KIND == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(r4[n & 3]) : "v"(laddr));
                else if (KIND == 3) asm volatile("ds_read2_b64 %0, %1 offset1:1" : "=v"(r4[n & 3]) : "v"(laddr));
                else if (KIND == 4) asm volatile("ds_write_b32 %0, %1" : : "v"(laddr), "v"(v[n & 3]));
                else if (KIND == 5) asm volatile("ds_write_b128 %0, %1" : : "v"(laddr), "v"(r4[n & 3]));
                else if (KIND == 6) asm volatile("global_load_dword %0, %1, off" : "=v"(v[(n + 4) & 7]) : "v"(gp));
                else if (KIND == 7) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
                else if (KIND == 9) asm volatile("ds_read_b32 %0, %1" : "=v"(v[(n + 4) & 7]) : "v"(laddr));
                else if (KIND == 10) asm volatile("s_nop 0");
                else if (KIND == 11) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p2[(n + 2) & 3]) : "v"(p2[n & 1]), "v"(p2[(n + 1) & 1]));
                else if (KIND == 12) asm volatile("v_sub_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(v[(n + 4) & 7]) : "v"(v[n & 3]), "v"(v[(n + 1) & 3]));
                else if (KIND == 13) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r4[n & 3]) : "v"(goff), "s"(in));
                else if (KIND == 14) asm volatile("global_load_dword %0, %1, %2" : "=v"(v[(n + 4) & 7]) : "v"(goff), "s"(in));
                else if (KIND == 15) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in + tid * 4), (__attribute__((address_space(3))) void*)(lds + 8192), 16, 0, 0);
                else if (KIND == 16) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[(n + 4) & 7]) : "v"(v[n & 3]), "v"(v[(n + 1) & 3]), "v"(v[(n + 2) & 3]));
                else if (KIND == 17) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(v[(n + 4) & 7]) : "v"(v[n & 3]), "v"(v[(n + 1) & 3]));
            }
            if (KIND == 2 || KIND == 3 || KIND == 4 || KIND == 5 || KIND == 9) { if ((c & 3) == 3) asm volatile("s_waitcnt lgkmcnt(0)"); }
            if (KIND == 6 || KIND == 13 || KIND == 14 || KIND == 15) { if ((c & 7) == 7) asm volatile("s_waitcnt vmcnt(0)"); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = (float)sacc;
    for (int i = 0; i < 16; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += r4[i][0] + r4[i][3] + p2[i][0] + p2[i][1];
    out[blockIdx.x * 256 + tid] = s + lds[tid];
}

template <int KIND, int N>
float run(float* out, const float* in, int blocks, int iters) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fill_loop<KIND, N>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t ldsb = 100 * 1024;      // one workgroup per CU, like the conv kernel
    hipLaunchKernelGGL((fill_loop<KIND, N>), dim3(blocks), dim3(256), ldsb, 0, out, in, iters);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((fill_loop<KIND, N>), dim3(blocks), dim3(256), ldsb, 0, out, in, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
}

int main() {
    const int blocks = 1024, iters = 300;       // 4 workgroups per CU in sequence, 4800 MFMAs per wave
    float *out, *in;
    (void)hipMalloc(&out, blocks * 256 * 4);
    (void)hipMalloc(&in, 16384 * 4);
    float h[16384];
    for (int i = 0; i < 16384; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    const double nm = 4.0 * iters * 16;          // MFMAs per wave per launch (4 workgroups per CU)
    const float base = run<0, 0>(out, in, blocks, iters);
    printf("bare loop: %.3f ms  -> %.1f ns per MFMA (64 cycles at %.2f GHz), %.1f TFLOP/s\n", base, base * 1e6 / nm,
           64.0 / (base * 1e6 / nm), 2.0 * 32 * 32 * 2 * nm * 1024 / (base * 1e-3) / 1e12);
#define ROW(K, name)                                                                                                   \
    {                                                                                                                  \
        const float t1 = run<K, 1>(out, in, blocks, iters), t2 = run<K, 2>(out, in, blocks, iters), t4 = run<K, 4>(out, in, blocks, iters); \
        printf("%-28s N=1 %.3f ms (+%5.1f ns/filler)  N=2 %.3f (+%5.1f)  N=4 %.3f (+%5.1f)\n", name, t1, (t1 - base) * 1e6 / nm,   \
               t2, (t2 - base) * 1e6 / nm / 2, t4, (t4 - base) * 1e6 / nm / 4);                                          \
    }
    ROW(1, "v_sub_f32")
    ROW(2, "ds_read_b128")
    ROW(3, "ds_read2_b64")
    ROW(9, "ds_read_b32")
    ROW(4, "ds_write_b32")
    ROW(5, "ds_write_b128")
    ROW(6, "global_load_dword")
    ROW(11, "v_pk_add_f32")
    ROW(12, "v_sub_f32_dpp row_shr:1")
    ROW(16, "v_fma_f32")
    ROW(17, "v_cndmask_b32")
    ROW(13, "global_load_dwordx4 saddr")
    ROW(14, "global_load_dword saddr")
    ROW(15, "global_load_lds_dwordx4")
    ROW(7, "s_add_u32")
    ROW(10, "s_nop 0")
    return 0;
}
