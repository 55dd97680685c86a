// Calibration of the MFMA-utilisation arithmetic: bare MFMA loops of KNOWN instruction count, run under
//   rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -- scripts/ubench_mfma_calib.bin
// so that scripts/pmc_summary.py can check (a) how many SQ_VALU_MFMA_BUSY_CYCLES one v_mfma_f32_16x16x32_bf16 /
// v_mfma_f32_32x32x16_bf16 adds (is the counter summed over all SIMDs of the chip, or sampled?) and (b) what
// GRBM_GUI_ACTIVE / 8 reads on a dispatch of a few microseconds against one of ~20 ms (the guide: "reads high on dispatches
// shorter than about 0.3 ms").  Build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_mfma_calib.bin scripts/ubench_mfma_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int UNROLL = 16;

__global__ __launch_bounds__(256) void calib_mfma_16x16x32(float* out, int iters) {
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x ^ i)); }
    f32x4 acc[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[u & 3], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void calib_mfma_32x32x16(float* out, int iters) {
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x ^ i)); }
    f32x16 acc[2] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 1], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[0][j] + acc[1][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const int grid = 256, block = 256;          // one workgroup per CU, one wave per SIMD
    float* out = nullptr;
    CK(hipMalloc(&out, (size_t)grid * block * sizeof(float)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Case { const char* name; int shape; int iters; int reps; };
    // short: ~8 us (the denoiser GEMM's scale); long: ~20 ms (where GRBM_GUI_ACTIVE / 8 / time is the clock)
    const Case cases[] = {{"calib_mfma_16x16x32 short", 0, 64, 50}, {"calib_mfma_16x16x32 long", 0, 160000, 3},
                          {"calib_mfma_32x32x16 short", 1, 32, 50}, {"calib_mfma_32x32x16 long", 1, 80000, 3}};
    printf("{\"waves_per_launch\": %d, \"cases\": [", grid * block / 64);
    bool first = true;
    for (const Case& c : cases) {
        for (int w = 0; w < 2; ++w) {             // warm-up
            if (c.shape == 0) hipLaunchKernelGGL(calib_mfma_16x16x32, dim3(grid), dim3(block), 0, 0, out, c.iters);
            else hipLaunchKernelGGL(calib_mfma_32x32x16, dim3(grid), dim3(block), 0, 0, out, c.iters);
        }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < c.reps; ++r) {
            if (c.shape == 0) hipLaunchKernelGGL(calib_mfma_16x16x32, dim3(grid), dim3(block), 0, 0, out, c.iters);
            else hipLaunchKernelGGL(calib_mfma_32x32x16, dim3(grid), dim3(block), 0, 0, out, c.iters);
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double per_wave = (double)c.iters * UNROLL;
        const double us = ms * 1e3 / c.reps;
        // cycles per MFMA per SIMD if the loop issued back to back at 2.4 GHz
        printf("%s{\"name\": \"%s\", \"iters\": %d, \"mfma_per_wave\": %.0f, \"mfma_per_launch\": %.0f, \"launches\": %d, \"us_per_launch\": %.3f, "
               "\"ns_per_mfma_per_simd\": %.3f}", first ? "" : ", ", c.name, c.iters, per_wave, per_wave * grid * block / 64, c.reps + 2,
               us, us * 1e3 / per_wave);
        first = false;
    }
    printf("]}\n");
    CK(hipFree(out));
    return 0;
}
