// Which bf16 MFMA shape issues faster per FLOP with weight-stationary operands (B fragments in registers, distinct per MFMA;
// A fragments re-used across column tiles; 8 independent accumulators): v_mfma_f32_16x16x32_bf16 vs v_mfma_f32_32x32x16_bf16.
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_mfma_shapes.bin scripts/ubench_mfma_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void shape16(float* out, const float* in, int iters) {
    bf16x8 w[32], a[2];                                  // 4 column tiles x 8 k-steps of weights, 2 row tiles of A
#pragma unroll
    for (int i = 0; i < 32; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) w[i][e] = (__bf16)in[(threadIdx.x + 17 * i + e) & 1023];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)in[(threadIdx.x * 3 + i + e) & 1023];
    f32x4 acc[2][4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], w[j * 8 + s], acc[i][j], 0, 0, 0);
        asm volatile("" : "+v"(a[0]), "+v"(a[1]));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// same FLOPs per iteration: 32 rows x 64 columns x K = 256: 2 column tiles of 32 x 16 k-steps of 16
__global__ __launch_bounds__(256) void shape32(float* out, const float* in, int iters) {
    bf16x8 w[32], a[1];
#pragma unroll
    for (int i = 0; i < 32; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) w[i][e] = (__bf16)in[(threadIdx.x + 17 * i + e) & 1023];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[0][e] = (__bf16)in[(threadIdx.x * 3 + e) & 1023];
    f32x16 acc[2] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], w[j * 16 + s], acc[j], 0, 0, 0);
        asm volatile("" : "+v"(a[0]));
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) s += acc[j][0] + acc[j][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float *out, *in;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&in, 4096);
    hipMemset(in, 0x3c, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int shape = 0; shape < 2; ++shape)
        for (int rep = 0; rep < 2; ++rep) {
            const int iters = 20000;
            hipEventRecord(e0, 0);
            if (shape == 0) hipLaunchKernelGGL(shape16, dim3(256), dim3(256), 0, 0, out, in, iters);
            else hipLaunchKernelGGL(shape32, dim3(256), dim3(256), 0, 0, out, in, iters);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = 2.0 * 32 * 64 * 256 * (double)iters * 1024;      // per wave-iteration 32x64x256, 1024 waves
            printf("%s: %.3f ms, %.1f TFLOP/s bf16, %.2f ns per 32x64x256 wave-tile\n", shape == 0 ? "16x16x32 (64 MFMAs/tile)" : "32x32x16 (32 MFMAs/tile)",
                   ms, flop / ms / 1e9, ms * 1e6 / iters);
        }
    return 0;
}
