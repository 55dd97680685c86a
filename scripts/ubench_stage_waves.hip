// How much of a pipeline stage's time per block is exposed latency that a second wave per SIMD would cover?
// One LIN / FFN-like stage (systolic.hip, MlpRole at 16 rows: x[16,256] -> act(x W1_slice^T)[16,128] -> . W2_slice^T -> [16,256])
// with its weight slice in registers, looping over blocks with the next block's rows prefetched, as
//   NW = 4 waves (one per SIMD, 256 weight VGPRs each: what the pipeline kernel does today) and
//   NW = 8 waves (two per SIMD, the slice split over them: 128 weight VGPRs each).
// No flags / polls; plain loads and stores, or the pipeline's sc1 loads and write-through stores: only the in-stage time per block.
// One workgroup per CU (100 KiB of LDS).  Measured: 4 waves 2.3 - 2.8 us per block, 8 waves 2.0 - 2.15 us.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I ladiff_amd/csrc -I include -o scripts/ubench_stage_waves.bin scripts/ubench_stage_waves.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "tile_mma.h"

using namespace ladiff;

template <int NW, int SC1>
__global__ __launch_bounds__(64 * NW, 1) void stage_kernel(const float* __restrict__ x, const float* __restrict__ w1s,
                                                           const float* __restrict__ w2s, float* __restrict__ out, int nblk) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    constexpr int NTH = 64 * NW, NT1 = 8 / NW, NT2 = 16 / NW, U = 512 / NTH;      // column tiles per wave; 8-column units per thread
    char* const atile = lds;                                                       // [16] x K = 256
    char* const htile = lds + 16 * 1024;                                           // [16] x K = 128
    float* const ct = reinterpret_cast<float*>(lds + 16 * 1024 + 16 * 512);        // [16][CLD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, frow = lane & 15, fk = lane >> 4;
    WFrag<0, NT1, 8> w1;
    WFrag<0, NT2, 4> w2;
    load_w(w1, w1s, 256, 0, [&](int j) { return 16 * NT1 * wave + 16 * j; });
    load_w(w2, w2s, 1024, 0, [&](int j) { return 16 * NT2 * wave + 16 * j; });
    const float* xb = x + (size_t)blockIdx.x * nblk * 16 * 256;
    float* ob = out + (size_t)blockIdx.x * nblk * 16 * 256;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, 0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(ob, 0, 0xffffffffu, 0x00020000);
    f32x4 cur[U][2], nxt[U][2];
    auto issue = [&](int b, f32x4 (&v)[U][2]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int id = tid + NTH * u, row = id >> 5, c8 = id & 31;
            if constexpr (SC1) {                                                   // the pipeline's hand-off path: 16-byte loads served by the memory side
                v[u][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (unsigned)(((size_t)b * 16 + row) * 1024 + c8 * 32), 0, 16));
                v[u][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (unsigned)(((size_t)b * 16 + row) * 1024 + c8 * 32 + 16), 0, 16));
            } else {
                v[u][0] = ld4(xb + ((size_t)b * 16 + row) * 256 + c8 * 8);
                v[u][1] = ld4(xb + ((size_t)b * 16 + row) * 256 + c8 * 8 + 4);
            }
        }
    };
    issue(0, cur);
    for (int b = 0; b < nblk; ++b) {
#pragma unroll
        for (int u = 0; u < U; ++u) {                                              // commit: fp32 rows -> S-format operand tile
            const int id = tid + NTH * u, row = id >> 5, c8 = id & 31;
            bf16x8 hi, lo;
            split8(cur[u][0], cur[u][1], hi, lo);
            *reinterpret_cast<bf16x8*>(a_slot<4>(atile, row, c8 >> 3, c8 & 7)) = hi;
            *reinterpret_cast<bf16x8*>(a_slot<4>(atile, row, c8 >> 3, 8 + (c8 & 7))) = lo;
        }
        __syncthreads();
        f32x4 acc1[1][NT1];
        zero_acc(acc1);
        mma<0, 4, NT1, 8, 1>(atile, w1, acc1);
#pragma unroll
        for (int j = 0; j < NT1; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) tile_put1<0, 2>(htile, 4 * fk + r, 16 * NT1 * wave + 16 * j + frow, fmaxf(acc1[0][j][r], 0.f));
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");            // LDS-only barrier: the prefetch below stays in flight
        if (b + 1 < nblk) issue(b + 1, nxt);
        f32x4 acc2[1][NT2];
        zero_acc(acc2);
        mma<0, 2, NT2, 4, 1>(htile, w2, acc2);
        stage_c(ct, acc2, [&](int j) { return 16 * NT2 * wave + 16 * j; });
        constexpr int CW = 16 * NT2, LPR = CW / 4, RPI = 64 / LPR;                 // the wave stores the columns it staged itself
#pragma unroll
        for (int q = 0; q < 16 / RPI; ++q) {
            const int row = RPI * q + lane / LPR, cc = CW * wave + 4 * (lane % LPR);
            if constexpr (SC1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ld4(ct + row * CLD + cc)), ro, (unsigned)((((size_t)b * 16 + row) * 256 + cc) * 4), 0, 16);   // write-through
            else st4(ob + ((size_t)b * 16 + row) * 256 + cc, ld4(ct + row * CLD + cc));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { cur[u][0] = nxt[u][0]; cur[u][1] = nxt[u][1]; }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");            // the next commit overwrites atile
    }
}

template <int NW, int SC1>
static float run(const float* x, const float* w1, const float* w2, float* out, int nwg, int nblk) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(stage_kernel<NW, SC1>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((stage_kernel<NW, SC1>), dim3(nwg), dim3(64 * NW), 100 * 1024, 0, x, w1, w2, out, nblk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((stage_kernel<NW, SC1>), dim3(nwg), dim3(64 * NW), 100 * 1024, 0, x, w1, w2, out, nblk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    const int nwg = 255, nblk = 400;
    const size_t nx = (size_t)nwg * nblk * 16 * 256;
    float *x, *out, *w1, *w2;
    hipMalloc(&x, nx * 4); hipMalloc(&out, nx * 4); hipMalloc(&w1, 128 * 256 * 4); hipMalloc(&w2, 256 * 1024 * 4);
    std::vector<float> h(nx);
    for (size_t i = 0; i < nx; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice);
    hipMemcpy(w1, h.data(), 128 * 256 * 4, hipMemcpyHostToDevice);                 // any bit pattern is a valid S-format row
    hipMemcpy(w2, h.data(), 256 * 1024 * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        const float t4 = run<4, 0>(x, w1, w2, out, nwg, nblk), t8 = run<8, 0>(x, w1, w2, out, nwg, nblk);
        const float s4 = run<4, 1>(x, w1, w2, out, nwg, nblk), s8 = run<8, 1>(x, w1, w2, out, nwg, nblk);
        printf("LIN-like stage, %d workgroups x %d blocks of 16 rows, us per block: plain loads / stores 4 waves %.3f, 8 waves %.3f (%.2fx); "
               "sc1 loads + write-through stores 4 waves %.3f, 8 waves %.3f\n", nwg, nblk, t4 * 1e3f / nblk, t8 * 1e3f / nblk, t4 / t8,
               s4 * 1e3f / nblk, s8 * 1e3f / nblk);
    }
    return 0;
}
