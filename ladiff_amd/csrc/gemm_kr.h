// K-resident LDS-DMA GEMM (small-M / denoiser side); see gemm_kr.hip.
#pragma once
#include "common.h"

namespace ladiff {

struct KrArgs {
    const float* A = nullptr;  int lda = 0;       // [M, K1]
    const float* A2 = nullptr; int lda2 = 0;      // optional second K segment (skip concat), K1 multiple of 256
    int K1 = 0;
    const float* W = nullptr;  int ldw = 0;       // [N, K]
    const float* bias = nullptr;
    float* Y = nullptr;        int ldy = 0;       // K > 256: K/256 partial planes [K/256][M][ldy] (split-K); may be null when Ys is set
    float* Ys = nullptr;                          // optional S-format copy of the result (row stride ldy), K == 256 only
    int split = 0;                                // 1: A, A2 and W are S-format (f16x3 products), see common.h
    int M = 0, N = 0, K = 0;                      // K multiple of 256
    int act = ACT_NONE;
    const float* res = nullptr; int ldres = 0;    // + residual after the activation
    unsigned long long* stamps = nullptr;         // diagnostic builds only (-DLADIFF_STAMPS): 8 x u64 per workgroup
};

int launch_gemm_kr(const KrArgs& a, hipStream_t s);

// gemm_rowln.hip: Y / Ys = LayerNorm( A . W^T + bias + res ) over 256-wide rows, S-format A [M, K] and W [256, K]
struct RowLnArgs {
    const float* A = nullptr;  int lda = 0;
    const float* A2 = nullptr; int lda2 = 0; int K1 = 0;      // optional second K segment (columns >= K1), skip-connection concat
    const float* W = nullptr;  int ldw = 0;
    const float* bias = nullptr;
    const float* res = nullptr; int ldres = 0;
    const float* ln_g = nullptr; const float* ln_b = nullptr; // both null: no LayerNorm
    float* Y = nullptr; float* Ys = nullptr; int ldy = 0;     // fp32 result and / or its S-format twin
    int M = 0, K = 0;                                          // K multiple of 64
};
int launch_gemm_rowln(const RowLnArgs& a, hipStream_t s);

// gemm_rowln.hip: Y / Ys = res + ( SiLU( LN(sum_s P[s] + bias2) * (1 + scale) + shift ) ) . W^T + bias,  K = N = 256;
// scale | shift = tab + step * tab_step_stride (256 + 256 floats), step = *d_step
struct CombineGemmArgs {
    const float* P = nullptr; size_t plane = 0; int S = 0;   // split-K partial planes [S][M][256]
    const float* bias2 = nullptr;
    const float* ln_g = nullptr; const float* ln_b = nullptr;
    const float* tab = nullptr; int tab_step_stride = 0; const int32_t* d_step = nullptr;
    const float* W = nullptr; int ldw = 0;                   // S-format [256, 256]
    const float* bias = nullptr;
    const float* res = nullptr; int ldres = 0;
    float* Y = nullptr; float* Ys = nullptr; int ldy = 0;
    int M = 0;
};
int launch_combine_gemm(const CombineGemmArgs& a, hipStream_t s);

}  // namespace ladiff
