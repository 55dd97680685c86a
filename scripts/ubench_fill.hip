// Micro-benchmark: how fast can 256 workgroups fill 144 KiB of LDS each right after a kernel boundary, as a function of
// how the workgroups share their source lines?  (Question behind it: does an XCD-aware tile map make the per-XCD L2
// serve the re-reads of the denoiser GEMMs?)  Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench_fill.hip -o gpurun_out/ubench_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int A_BYTES = 80 * 1024, W_BYTES = 64 * 1024;   // the <80,64> tile of gemm_kr: 80 + 64 rows of 1 KiB

__device__ __forceinline__ void glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// mode: 0 private regions, 1 one region for everybody, 2 region = blockIdx % 8, 3 region = blockIdx / 32,
//       4 GEMM natural (bm = bid / 16, bn = bid % 16), 5 GEMM XCD-aware (xcd = bid % 8 owns bm in {2x, 2x+1}, all bn)
//       6 GEMM, A only;  7 GEMM, W only
__global__ __launch_bounds__(256) void fill_kernel(const char* __restrict__ src, int mode, float* __restrict__ sink,
                                                   int* __restrict__ xcc) {
    extern __shared__ char lds[];
    const int bid = blockIdx.x, tid = threadIdx.x;
    size_t a_off, w_off;
    const size_t A_REGION = 16ull * A_BYTES;      // 16 A tiles, then 16 W tiles, then private space
    int bm = bid / 16, bn = bid % 16;
    if (mode == 5) { const int x = bid % 8, j = bid / 8; bm = 2 * x + (j >> 4); bn = j & 15; }
    switch (mode) {
        case 0: a_off = (size_t)bid * (A_BYTES + W_BYTES); w_off = a_off + A_BYTES; break;
        case 1: a_off = 0; w_off = A_REGION; break;
        case 2: a_off = (size_t)(bid % 8) * A_BYTES; w_off = A_REGION + (size_t)(bid % 8) * W_BYTES; break;
        case 3: a_off = (size_t)(bid / 32) * A_BYTES; w_off = A_REGION + (size_t)(bid / 32) * W_BYTES; break;
        default: a_off = (size_t)bm * A_BYTES; w_off = A_REGION + (size_t)bn * W_BYTES; break;
    }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    if (mode != 7)
        for (int i = wave; i < A_BYTES / 1024; i += 4) glds16(src + a_off + i * 1024 + lane * 16, lds + i * 1024);
    if (mode != 6)
        for (int i = wave; i < W_BYTES / 1024; i += 4) glds16(src + w_off + i * 1024 + lane * 16, lds + A_BYTES + i * 1024);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float v = reinterpret_cast<float*>(lds)[tid * 37 % (A_BYTES / 4)] + reinterpret_cast<float*>(lds)[A_BYTES / 4 + tid];
    if (v == 12345.678f) sink[bid] = v;
    if (tid == 0 && xcc != nullptr) {
        int id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[bid] = id;
    }
}

// second experiment: the consumer's W cycles through 72 MiB of weights (like the 99 matrices of a denoiser step) and its
// A tiles were written by the previous kernel (the producer), as in the real chain
__global__ __launch_bounds__(256) void produce_kernel(char* __restrict__ a, float v) {
    float4* p = reinterpret_cast<float4*>(a + (size_t)blockIdx.x * (16 * A_BYTES / 256));
    for (int i = threadIdx.x; i < 16 * A_BYTES / 256 / 16; i += 256) p[i] = float4{v, v, v, v};
}
__global__ __launch_bounds__(256) void consume_kernel(const char* __restrict__ a, const char* __restrict__ w, float* __restrict__ sink) {
    extern __shared__ char lds[];
    const int bid = blockIdx.x, tid = threadIdx.x;
    const int x = bid % 8, j = bid / 8, bm = 2 * x + (j >> 4), bn = j & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    for (int i = wave; i < A_BYTES / 1024; i += 4) glds16(a + (size_t)bm * A_BYTES + i * 1024 + lane * 16, lds + i * 1024);
    for (int i = wave; i < W_BYTES / 1024; i += 4) glds16(w + (size_t)bn * W_BYTES + i * 1024 + lane * 16, lds + A_BYTES + i * 1024);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float v = reinterpret_cast<float*>(lds)[tid * 37 % (A_BYTES / 4)] + reinterpret_cast<float*>(lds)[A_BYTES / 4 + tid];
    if (v == 12345.678f) sink[bid] = v;
}

static void chain(hipStream_t s, hipEvent_t e0, hipEvent_t e1, char* a, char* a2, char* wts, float* sink, bool prod, bool cons,
                  bool cycle, bool a_static, const char* name) {
    const int n = 216;
    for (int rep = 0; rep < 2; ++rep) {
        if (rep) CK(hipEventRecord(e0, s));
        for (int i = 0; i < n; ++i) {
            if (prod) hipLaunchKernelGGL(produce_kernel, dim3(256), dim3(256), 0, s, a, (float)i);
            if (cons) hipLaunchKernelGGL(consume_kernel, dim3(256), dim3(256), A_BYTES + W_BYTES, s, a_static ? a2 : a,
                                         wts + (cycle ? (size_t)(i % 72) << 20 : 0), sink);
        }
    }
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-58s %7.2f us/iteration\n", name, ms * 1e3 / n);
}

int main() {
    const size_t bytes = 256ull * (A_BYTES + W_BYTES) + (64 << 20);
    char* src; float* sink; int* xcc;
    CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 1, bytes));
    CK(hipMalloc(&sink, 4096)); CK(hipMalloc(&xcc, 256 * 4));
    CK(hipFuncSetAttribute((const void*)fill_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, A_BYTES + W_BYTES));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), A_BYTES + W_BYTES, s, src, 0, sink, xcc);
    CK(hipStreamSynchronize(s));
    std::vector<int> h(256); CK(hipMemcpy(h.data(), xcc, 1024, hipMemcpyDeviceToHost));
    printf("XCC_ID of blocks 0..31:");
    for (int i = 0; i < 32; ++i) printf(" %d", h[i] & 0xf);
    printf("\n");
    const char* names[] = {"private 144K/block", "one region for all", "region = bid % 8", "region = bid / 32", "GEMM natural map",
                           "GEMM XCD-aware map", "GEMM A only (80K)", "GEMM W only (64K)"};
    for (int mode = 0; mode < 8; ++mode) {
        const int n = 200;
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), A_BYTES + W_BYTES, s, src, mode, sink, nullptr);
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), A_BYTES + W_BYTES, s, src, mode, sink, nullptr);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("mode %d %-22s %7.2f us/launch\n", mode, names[mode], ms * 1e3 / n);
    }
    char *a, *a2, *wts;
    CK(hipMalloc(&a, 16 * A_BYTES)); CK(hipMalloc(&a2, 16 * A_BYTES)); CK(hipMalloc(&wts, 73ull << 20));
    CK(hipMemset(a, 0, 16 * A_BYTES)); CK(hipMemset(a2, 0, 16 * A_BYTES)); CK(hipMemset(wts, 0, 73ull << 20));
    CK(hipFuncSetAttribute((const void*)consume_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, A_BYTES + W_BYTES));
    chain(s, e0, e1, a, a2, wts, sink, false, true, false, false, "consumer, static W, static A");
    chain(s, e0, e1, a, a2, wts, sink, false, true, true, false, "consumer, W cycling through 72 MiB, static A");
    chain(s, e0, e1, a, a2, wts, sink, true, false, false, false, "producer only (1.25 MiB written)");
    chain(s, e0, e1, a, a2, wts, sink, true, true, false, true, "producer + consumer, static W, A not the produced one");
    chain(s, e0, e1, a, a2, wts, sink, true, true, false, false, "producer + consumer, static W, A just produced");
    chain(s, e0, e1, a, a2, wts, sink, true, true, true, false, "producer + consumer, cycling W, A just produced");
    return 0;
}
