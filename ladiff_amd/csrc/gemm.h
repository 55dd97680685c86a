// fp32-input MFMA GEMM with fused prologue / epilogue:  Y = epi( [A | A2] . W^T )
#pragma once
#include "common.h"

namespace ladiff {

struct GemmArgs {
    // operands
    const float* A = nullptr;  int lda = 0;      // [M, K1]
    const float* A2 = nullptr; int lda2 = 0;     // optional second K segment [M, K-K1] (skip-connection concat)
    int K1 = 0;                                  // columns taken from A (== K when A2 == nullptr)
    const float* W = nullptr;  int ldw = 0;      // [N, K] row-major (nn.Linear.weight layout)
    const float* bias = nullptr;                 // [N] or null
    float* Y = nullptr;        int ldy = 0;      // [M, N]; may be null when Ys is set (split path)
    float* Ys = nullptr;                         // optional S-format copy of the result (row stride ldy), split path only
    int split = 0;                               // 1: A, A2, W are S-format rows (f16x3 products, common.h); large-M kernel only
    int M = 0, N = 0, K = 0;
    // epilogue, applied in this order
    int act = ACT_NONE;                          // activation on (acc + bias)
    const float* res = nullptr; int ldres = 0;   // + residual[M, N]
    const float* ln_g = nullptr; const float* ln_b = nullptr;    // LayerNorm over the row (needs N == 256)
    const float* ln2_g = nullptr; const float* ln2_b = nullptr;  // second LayerNorm (decoder.norm after norm3)
    const float* mod = nullptr;                  // AdaLN: v*(1+scale)+shift, scale|shift = mod + *d_step*mod_stride
    const int32_t* d_step = nullptr; int mod_stride = 0;
    int post_act = ACT_NONE;                     // activation after LN / modulation
    const int32_t* row_len = nullptr; int rows_per_item = 0;     // zero rows with (row % rows_per_item) >= row_len[row / rows_per_item]
    const int32_t* row_map = nullptr;            // row r is written to Y row row_map[r] (ragged rows scattered into a padded tensor)
    // split mode, large-M kernel only: the K range in `ksplit` equal parts over blockIdx.y, part z's RAW sums (no bias / activation /
    // residual) to Y + z * plane; the caller sums the planes.  For launches whose tiles alone leave the chip one workgroup per CU or
    // less: a workgroup keeps ONE K stage in flight (gemm_big.hip), so a lone workgroup on a CU pays every stage's load latency.
    int ksplit = 1; size_t plane = 0;
};

int launch_gemm(const GemmArgs& a, hipStream_t stream);
// n <= GEMM_BATCH_MAX independent GEMMs of ONE shape (M, N, K, the same epilogue kind; pointers differ) as one launch:
// the per-layer projections of a prologue are each too small to fill the chip and would otherwise run one after another
constexpr int GEMM_BATCH_MAX = 9;
struct GemmBatch { GemmArgs a[GEMM_BATCH_MAX]; };
int launch_gemm_batch(const GemmArgs* list, int n, hipStream_t stream);
int launch_gemm_big_batch(const GemmBatch& b, int n, hipStream_t stream);
bool gemm_big_supported(const GemmArgs& a);          // gemm_big.hip: large-M LDS-DMA kernels
int launch_gemm_big(const GemmArgs& a, hipStream_t stream);

}  // namespace ladiff
