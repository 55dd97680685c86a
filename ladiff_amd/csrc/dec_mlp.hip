// The decoder layer's feed-forward block as ONE kernel (bf16x3 mode):  y = LN3( x + W2 gelu(W1 x + b1) + b2 )  [+ a second LayerNorm]
// TransformerDecoderLayer.forward_post, cross_attention.py:410-412 (+ decoder.norm, :150-151, on the last layer).
//
// Before: linear1 GEMM (writes the [M,1024] hidden rows, 103 MB at M = 25088) + linear2 GEMM (reads them back) + a LayerNorm row
// kernel: 60 + 60 + 14 us per layer, each GEMM tile re-ingesting 256 KB of operands through LDS for 25 MFLOP (bound by the
// L2 -> LDS stream at ~27 GB/s per CU, profiles/r2/10).  Here the ACTIVATIONS stay in registers and only the WEIGHTS stream:
//   * a workgroup owns 128 rows, a wave 16 of them.  The wave's x rows are MFMA operand fragments in registers (S-format hi / lo,
//     loaded straight from global memory); the hidden rows and the output rows never leave the wave's registers.
//   * every product is computed TRANSPOSED, D^T = W . x^T (weight fragment = MFMA "A" operand, x fragment = "B" operand): a lane
//     then holds 4 CONSECUTIVE columns of ONE row per accumulator, so that (a) GELU'd hidden values turn into the next product's
//     operand fragment by an in-lane conversion, with no LDS round trip, (b) LayerNorm is an in-lane sum + two cross-lane steps,
//     (c) the results go to memory as 16-byte stores straight from the accumulators.
//     For (a) the weight rows of a 128-row panel are dealt to the MFMA tiles in a permuted order (PI below) by the DMA source
//     addresses - free - so that the tile pair (2c, 2c+1) leaves lane group g with columns 32c + 8g .. + 7: exactly the eight k
//     values that lane group needs as an operand of the next product's k-step c.
//   * the weights (2 MB in S-format) stream through a ring of eight 16-KiB LDS stages filled by LDS-DMA
//     (`global_load_lds_dwordx4`), six stages ahead of the MFMAs; a stage = 128 weight rows x 32 k (64 B hi + 64 B lo per row),
//     the layout and slot swizzle of gemm_big_split_kernel (conflict-free ds_read_b128 for the 16x16x32 operand layout).  Per
//     128 hidden columns: 8 stages of W1 (k = the 256 model columns) then 8 stages of W2 (its 256 rows in two halves x the 128
//     hidden columns in four k-steps).  Per 128 rows that is 2 MB of ingest for 403 MFLOP (bf16): 200 FLOP per byte, against
//     96 for a 128x128x256 GEMM tile - the MFMA pipe, not the LDS fill, is the bound.
//   * eight waves (two per SIMD): one wave's LDS-DMA issue, LDS reads, GELU and barrier waits hide under its partner's MFMAs; waves
//     4-7 issue their DMA pieces after the stage's MFMAs, waves 0-3 before (partners must not do the same thing at the same time).
#include "model.h"
#include "tile_mma.h"

namespace ladiff {

namespace {

constexpr int MLP_BM = 128;                    // rows per workgroup
constexpr int MLP_NS = 8;                      // ring stages
constexpr int MLP_STAGE = 16384;               // bytes: 128 weight rows x 128 B
constexpr int MLP_AHEAD = 6;                   // stages in flight behind the one being multiplied
constexpr int MLP_LDS = MLP_NS * MLP_STAGE + FF * 4;      // ring + linear1's bias

typedef unsigned u32x4_m __attribute__((ext_vector_type(4)));

struct MlpArgs {
    const float* xs;       // [M,256] S-format: the operand
    const float* x;        // [M,256] fp32: the residual
    const float* w1;       // [1024,256] S-format
    const float* b1;       // [1024]
    const float* w2;       // [256,1024] S-format
    const float* b2;       // [256]
    const float* g3; const float* be3;         // LayerNorm
    const float* g4; const float* be4;         // optional second LayerNorm (decoder.norm)
    float* y; float* ys;                       // fp32 / S-format results (either may be NULL)
    int M;
};

// weight row (within a 128-row panel) that sits in LDS row p of a stage: tile j = p >> 4, lane row n = p & 15
__device__ __forceinline__ int pi_row(int p) { const int j = p >> 4, n = p & 15; return 32 * (j >> 1) + 8 * (n >> 2) + 4 * (j & 1) + (n & 3); }

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int OFF>
__device__ __forceinline__ void fetch16(u32x4_m& v, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_lgkm(u32x4_m& a, u32x4_m& b, u32x4_m& c, u32x4_m& d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
__device__ __forceinline__ bf16x8 as_bf(const u32x4_m v) { return __builtin_bit_cast(bf16x8, v); }

}  // namespace

__global__ __launch_bounds__(512, 2) void dec_mlp_kernel(const MlpArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    float* const b1s = reinterpret_cast<float*>(lds + MLP_NS * MLP_STAGE);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fk = lane >> 4;
    const int row0 = blockIdx.x * MLP_BM + 16 * wave;
    const int M = p.M;
    int myrow = row0 + frow;
    const bool live = myrow < M;
    myrow = live ? myrow : M - 1;

    // ---- this wave's x rows as operand fragments: k-step s (32 columns) -> hi / lo 16 bytes of lane (row frow, k 8 fk .. + 7)
    bf16x8 xh[8], xl[8];
    {
        const char* xr = reinterpret_cast<const char*>(p.xs) + (size_t)myrow * 1024 + fk * 16;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            xh[s] = *reinterpret_cast<const bf16x8*>(xr + (s >> 1) * 256 + (s & 1) * 64);
            xl[s] = *reinterpret_cast<const bf16x8*>(xr + (s >> 1) * 256 + (s & 1) * 64 + 128);
        }
    }
    for (int i = tid; i < FF / 4; i += 512) st4(b1s + 4 * i, ld4(p.b1 + 4 * i));

    // ---- LDS-DMA of one stage: wave w brings LDS rows 64 g + 8 w + (lane >> 3), g = 0, 1 (a piece = 8 rows x 128 B)
    const int prow = 8 * wave + (lane >> 3);                             // LDS row within a 64-row half
    const int cs = (lane & 7) ^ ((prow >> 1) & 7);                       // source slot that lands in LDS slot (lane & 7)
    const int koff = (cs < 4 ? cs * 4 : 32 + (cs - 4) * 4);              // floats inside the 64-float S-block: hi | lo halves
    const int src0 = pi_row(prow), src1 = pi_row(64 + prow);            // weight rows (within the panel) of this lane's two pieces
    char* const dma_dst = lds + prow * 0 + (8 * wave) * 128;             // + 64 g rows; the hardware adds lane * 16
    // stage (hs, u): u < 8: W1 rows 128 hs .., k-step u;  u >= 8: v = u - 8, k-step c = v >> 1 of the hidden slice, W2 rows 128 (v & 1) ..
    auto issue = [&](int hs, auto uc, auto slotc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value, slot = decltype(slotc)::value;
        const float* base; int ldw, n0, kb, half;
        if constexpr (u < 8) { base = p.w1; ldw = D; n0 = 128 * hs; kb = u >> 1; half = u & 1; }
        else { constexpr int v = u - 8, c = v >> 1; base = p.w2; ldw = FF; n0 = 128 * (v & 1); kb = 2 * hs + (c >> 1); half = c & 1; }
        const float* s0 = base + (size_t)(n0 + src0) * ldw + kb * 64 + half * 16 + koff;
        const float* s1 = base + (size_t)(n0 + src1) * ldw + kb * 64 + half * 16 + koff;
        __builtin_amdgcn_global_load_lds(s0, (__attribute__((address_space(3))) void*)(dma_dst + slot * MLP_STAGE), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(s1, (__attribute__((address_space(3))) void*)(dma_dst + slot * MLP_STAGE + 64 * 128), 16, 0, 0);
    };

    // fragment read addresses: tile j of a stage = LDS rows 16 j + frow; hi slot fk, lo slot 4 + fk, XORed with (frow >> 1) & 7
    const int sw = (frow >> 1) & 7;
    const unsigned rd_hi = lds_addr(lds) + frow * 128 + ((fk ^ sw) << 4);
    const unsigned rd_lo = lds_addr(lds) + frow * 128 + (((4 + fk) ^ sw) << 4);
    const unsigned rd_hi2 = rd_hi + 4 * MLP_STAGE, rd_lo2 = rd_lo + 4 * MLP_STAGE;     // slots 4..7: the offset field holds 16 bits

    f32x4 hacc[8], oacc[16];
    bf16x8 hh[4], hl[4];
#pragma unroll
    for (int j = 0; j < 16; ++j) oacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one stage's products: acc[j0 + j] += W(tile j) . b^T over its 32 k, j < 8; three bf16 MFMAs per product (x_lo W_hi, x_hi W_lo,
    // x_hi W_hi: the order of gemm_big_split_kernel).  The fragments of the next tile pair are requested before a pair's MFMAs.
    auto stage_mma = [&](auto slotc, f32x4* acc, const bf16x8 bh, const bf16x8 bl) __attribute__((always_inline)) {
        constexpr int slot = decltype(slotc)::value;
        const unsigned ah = slot < 4 ? rd_hi : rd_hi2, al = slot < 4 ? rd_lo : rd_lo2;
        constexpr int so = (slot & 3) * MLP_STAGE;
        u32x4_m wh[2][2], wl[2][2];
        auto fetch = [&](auto prc, auto bc) __attribute__((always_inline)) {
            constexpr int pr = decltype(prc)::value, bf = decltype(bc)::value;
            fetch16<so + (2 * pr) * 2048>(wh[bf][0], ah); fetch16<so + (2 * pr) * 2048>(wl[bf][0], al);
            fetch16<so + (2 * pr + 1) * 2048>(wh[bf][1], ah); fetch16<so + (2 * pr + 1) * 2048>(wl[bf][1], al);
        };
        fetch(IntC<0>{}, IntC<0>{});
        static_for<4>([&](auto prc) {
            constexpr int pr = decltype(prc)::value, cur = pr & 1;
            if constexpr (pr + 1 < 4) fetch(IntC<pr + 1>{}, IntC<cur ^ 1>{});
            wait_lgkm<(pr + 1 < 4 ? 4 : 0)>(wh[cur][0], wl[cur][0], wh[cur][1], wl[cur][1]);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 a0 = acc[2 * pr], a1 = acc[2 * pr + 1];
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(wh[cur][0]), bl, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(wh[cur][1]), bl, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(wl[cur][0]), bh, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(wl[cur][1]), bh, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(wh[cur][0]), bh, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(wh[cur][1]), bh, a1, 0, 0, 0);
            acc[2 * pr] = a0; acc[2 * pr + 1] = a1;
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // prologue: stages 0 .. AHEAD-1 of hidden slice 0
    static_for<MLP_AHEAD>([&](auto uc) { constexpr int u = decltype(uc)::value; issue(0, IntC<u>{}, IntC<u % MLP_NS>{}); });
    __syncthreads();                                                     // linear1's bias is in LDS (plain stores: lgkmcnt, compiler-tracked)

    const bool early = wave < 4;                                         // waves 0-3 issue their DMA before the MFMAs, 4-7 after
#pragma unroll 1
    for (int hs = 0; hs < 8; ++hs) {
        static_for<16>([&](auto uc) {
            constexpr int u = decltype(uc)::value, slot = u % MLP_NS;
            constexpr int ut = (u + MLP_AHEAD) % 16, slot_t = (u + MLP_AHEAD) % MLP_NS;
            const int hs_t = (hs + (u + MLP_AHEAD >= 16 ? 1 : 0)) & 7;  // past the last slice: a harmless re-fetch keeps the counts uniform
            // the stage AHEAD further goes into the slot consumed two stages ago (every wave left it before the previous barrier)
            if (early) issue(hs_t, IntC<ut>{}, IntC<slot_t>{});
            // this wave's pieces of stage (hs, u) have landed: all but the youngest AHEAD (early) / AHEAD - 1 (late) stages' pieces
            if (early) wait_vm<2 * MLP_AHEAD>(); else wait_vm<2 * (MLP_AHEAD - 1)>();
            __builtin_amdgcn_s_barrier();                                // ... and every other wave's
            if constexpr (u < 8) {
                if constexpr (u == 0) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) hacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                stage_mma(IntC<slot>{}, hacc, xh[u], xl[u]);
            } else {
                constexpr int v = u - 8, c = v >> 1, nh = v & 1;
                if constexpr (nh == 0) {
                    // hidden columns 32 c + 8 fk .. + 7 of the slice (tiles 2c, 2c+1): + bias, GELU, split -> operand fragment of k-step c
                    const f32x4 ba = ld4(b1s + 128 * hs + 32 * c + 8 * fk), bb = ld4(b1s + 128 * hs + 32 * c + 8 * fk + 4);
                    f32x4 va, vb;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { va[e] = gelu_erf(hacc[2 * c][e] + ba[e]); vb[e] = gelu_erf(hacc[2 * c + 1][e] + bb[e]); }
                    split8(va, vb, hh[c], hl[c]);
                }
                stage_mma(IntC<slot>{}, oacc + 8 * nh, hh[c], hl[c]);
            }
            if (!early) issue(hs_t, IntC<ut>{}, IntC<slot_t>{});
        });
    }
    wait_vm<0>();                                                        // no LDS-DMA may land after the workgroup has gone

    // ---- epilogue, per lane: row frow of the wave, columns col(jj) .. + 3 of accumulator jj:  + bias + residual, LayerNorm(s), store
    const size_t rbase = (size_t)myrow * D;
    f32x4 v[16];
    float s = 0.f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const int col = 32 * (jj >> 1) + 8 * fk + 4 * (jj & 1);
        const f32x4 bv = ld4(p.b2 + col), rv = ld4(p.x + rbase + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[jj][e] = oacc[jj][e] + bv[e] + rv[e]; }
        s += (v[jj][0] + v[jj][1]) + (v[jj][2] + v[jj][3]);
    }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const float* gp = pass == 0 ? p.g3 : p.g4;
        const float* bp = pass == 0 ? p.be3 : p.be4;
        if (gp == nullptr) break;
        if (pass == 1) {
            s = 0.f;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) s += (v[jj][0] + v[jj][1]) + (v[jj][2] + v[jj][3]);
        }
        s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);          // the row's four lane groups
        const float mean = s * (1.f / 256.f);
        float q = 0.f;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[jj][e] - mean; q = fmaf(d, d, q); }
        q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
        const float rstd = rsqrtf(q * (1.f / 256.f) + LN_EPS);
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int col = 32 * (jj >> 1) + 8 * fk + 4 * (jj & 1);
            const f32x4 ga = ld4(gp + col), be = ld4(bp + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[jj][e] = (v[jj][e] - mean) * rstd * ga[e] + be[e];
        }
    }
    if (live) {
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int col = 32 * (jj >> 1) + 8 * fk + 4 * (jj & 1);
            if (p.y != nullptr) st4(p.y + rbase + col, v[jj]);
            if (p.ys != nullptr) store_split4(p.ys + rbase, col, v[jj]);
        }
    }
}

// y / ys [M,256] = LN3(x + lin2(gelu(lin1(x)))) (then LN4 when g4 != NULL); xs = S-format twin of x, w1 / w2 S-format
int launch_dec_mlp(const float* xs, const float* x, const float* w1, const float* b1, const float* w2, const float* b2, const float* g3,
                   const float* be3, const float* g4, const float* be4, float* y, float* ys, int M, hipStream_t s) {
    if (M <= 0) return 0;
    static bool attr_set[64] = {};
    int dev = 0;
    LADIFF_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return LADIFF_ERR_ARG;
    if (!attr_set[dev]) {
        LADIFF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dec_mlp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS));
        attr_set[dev] = true;
    }
    MlpArgs a{xs, x, w1, b1, w2, b2, g3, be3, g4, be4, y, ys, M};
    hipLaunchKernelGGL(dec_mlp_kernel, dim3((M + MLP_BM - 1) / MLP_BM), dim3(512), MLP_LDS, s, a);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

}  // namespace ladiff
