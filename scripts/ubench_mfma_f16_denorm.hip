// Does v_mfma_f32_16x16x32_f16 keep SUBNORMAL fp16 inputs (a hi + lo fp16 split puts the lo parts of small values there), and how do
// three split products compare: bf16 pairs (the library's S-format, 16 mantissa bits) against fp16 pairs (22 bits) on random data?
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_mfma_f16_denorm.bin scripts/ubench_mfma_f16_denorm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// one wave: D[16x16] = A[16xK] B[16xK]^T, K = 32 per MFMA; lane (r = lane & 15, q = lane >> 4) holds A[r][8q .. 8q+7] and B[r][8q .. 8q+7]
__global__ void denorm_probe(float* out, float a_val, float b_val) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)a_val; b[e] = (_Float16)b_val; }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}
__global__ void pkrtz_probe(float* out, float a, float b) {          // does v_cvt_pkrtz_f16_f32 saturate (round toward zero never reaches infinity)?
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 h = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(a, b));
    if (threadIdx.x == 0) { out[0] = (float)h[0]; out[1] = (float)h[1]; }
}
template <bool F16>
__global__ void split_product(const float* A, const float* B, float* D, int K) {   // A [16][K], B [16][K] row-major, D [16][16]
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 32) {
        float av[8], bv[8];
        for (int e = 0; e < 8; ++e) { av[e] = A[r * K + k0 + 8 * q + e]; bv[e] = B[r * K + k0 + 8 * q + e]; }
        if constexpr (F16) {
            f16x8 ah, al, bh, bl;
            for (int e = 0; e < 8; ++e) {
                ah[e] = (_Float16)av[e]; al[e] = (_Float16)(av[e] - (float)ah[e]);
                bh[e] = (_Float16)bv[e]; bl[e] = (_Float16)(bv[e] - (float)bh[e]);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
        } else {
            bf16x8 ah, al, bh, bl;
            for (int e = 0; e < 8; ++e) {
                ah[e] = (__bf16)av[e]; al[e] = (__bf16)(av[e] - (float)ah[e]);
                bh[e] = (__bf16)bv[e]; bl[e] = (__bf16)(bv[e] - (float)bh[e]);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
        }
    }
    // D layout of 16x16x32: lane holds rows 4q .. 4q+3 of column r  (D[i][j] = sum_k A[i][k] B[j][k])
    for (int e = 0; e < 4; ++e) D[(4 * q + e) * 16 + r] = acc[e];
}
int main() {
    float* out; hipMalloc(&out, 4);
    const float vals[4] = {1.0f, 6.1035156e-5f /* 2^-14: smallest normal */, 9.5367432e-7f /* 2^-20: subnormal */, 5.9604645e-8f /* 2^-24: smallest subnormal */};
    for (float v : vals) {
        denorm_probe<<<1, 64>>>(out, v, 1.0f);
        float h; hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost);
        printf("A = %.8e (all 32 k), B = 1: D = %.8e  expected %.8e  -> %s\n", v, h, 32.0 * v, h == 32.f * v ? "kept" : (h == 0.f ? "FLUSHED" : "other"));
    }
    {
        float* o2; hipMalloc(&o2, 8);
        const float pairs[3][2] = {{1e6f, -1e6f}, {65504.f, 65520.f}, {1.0009765f, -3e38f}};
        for (auto& pr : pairs) {
            pkrtz_probe<<<1, 64>>>(o2, pr[0], pr[1]);
            float h[2]; hipMemcpy(h, o2, 8, hipMemcpyDeviceToHost);
            printf("v_cvt_pkrtz_f16_f32(%.8g, %.8g) = (%.8g, %.8g)\n", pr[0], pr[1], h[0], h[1]);
        }
    }
    const int K = 768;
    std::mt19937 g(1); std::normal_distribution<float> n(0.f, 1.f);
    for (float wscale : {1.0f, 0.02f, 0.02f * 4096.f}) {
        std::vector<float> A(16 * K), B(16 * K);
        for (auto& x : A) x = n(g);
        for (auto& x : B) x = wscale * n(g);
        float *dA, *dB, *dD; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1024);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        double ref[256];
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * B[j * K + k]; ref[i * 16 + j] = s; }
        double rms = 0; for (double r : ref) rms += r * r; rms = std::sqrt(rms / 256);
        for (int f16 = 0; f16 < 2; ++f16) {
            if (f16) split_product<true><<<1, 64>>>(dA, dB, dD, K); else split_product<false><<<1, 64>>>(dA, dB, dD, K);
            float D[256]; hipMemcpy(D, dD, 1024, hipMemcpyDeviceToHost);
            double e2 = 0, emax = 0; for (int i = 0; i < 256; ++i) { double e = D[i] - ref[i]; e2 += e * e; emax = std::fmax(emax, std::fabs(e)); }
            printf("K = %d, A ~ N(0,1), B ~ %.3g N(0,1): %s split x3: rms error / rms value = %.3e, max error / rms value = %.3e\n", K, wscale, f16 ? "fp16" : "bf16",
                   std::sqrt(e2 / 256) / rms, emax / rms);
        }
        // fp32 reference accumulation error for scale
        double e2 = 0; for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < K; ++k) s = fmaf(A[i * K + k], B[j * K + k], s); double e = s - ref[i * 16 + j]; e2 += e * e; }
        printf("      fp32 fma chain: rms error / rms value = %.3e\n", std::sqrt(e2 / 256) / rms);
    }
    return 0;
}
