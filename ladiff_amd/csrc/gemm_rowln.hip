// Y = [LayerNorm]( [A | A2] . W^T + bias + res ) for the 256-wide rows of the denoiser loop, f16x3 products (S-format operands):
// the self-attention out-projection + residual + norm1 of `TransformerEncoderLayer.forward_post`
// (mdiff_transformer.py:57-63) in ONE launch instead of GEMM + row kernel.
//
// A LayerNorm needs whole rows, so a workgroup owns 16 rows x all 256 columns and streams K in 64-wide stages through a
// two-stage LDS ring (16 A rows + 256 W rows = 68 KiB per stage).  M = 1280 gives 80 workgroups; each pulls the whole
// weight matrix (256 KiB at K = 256) through its texture path at 64 B/clk, which is what bounds the kernel (~4.3 k
// cycles) - still shorter than a second launch.  Producer / consumer split as in gemm_kr.hip: waves 4-7 issue the LDS-DMA
// (counted vmcnt, stage hand-over through the workgroup barrier), waves 0-3 own 64 columns each and run the MFMAs; the
// epilogue stages the tile through LDS once and normalises one row per wave-instruction (DPP reductions).
#include "gemm_kr.h"

namespace ladiff {

namespace {
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void dma16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
}  // namespace

__global__ __launch_bounds__(512) void gemm_rowln_kernel(const RowLnArgs p) {
    constexpr int BM = 16, BN = 256;
    constexpr int ROWS = BM + BN;                  // 272 LDS rows of 256 B per stage
    constexpr int STAGE = ROWS * 64;
    constexpr int PCS = ROWS / 4;                  // 68 one-KiB pieces per stage
    constexpr int PPW = PCS / 4;                   // 17 per producer wave
    constexpr int RN = 4;                          // 16-column tiles per consumer wave
    constexpr int CLD = BN + 4;
    static_assert(2 * PPW <= 63, "vmcnt is 6 bits");

    __shared__ __attribute__((aligned(1024))) float lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int argM = pin_s(p.M), nk = pin_s(p.K) >> 6;
    const int row0 = blockIdx.x * BM;

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        const int pw = wave - 4;
        const int rl = 4 * pw + (lane >> 4);       // LDS row & 15 of every piece this wave issues
        const int kl = ((lane & 15) ^ rl) << 2;    // swizzled source slot (floats)
        const float* const argA = pin_s(p.A); const float* const argW = pin_s(p.W);
        const int lda = pin_s(p.lda), ldw = pin_s(p.ldw);
        int gr = row0 + rl; gr = gr < argM ? gr : argM - 1;
        const float* const arow = argA + (size_t)gr * lda + kl;           // this lane's A row (piece 0 of every stage)
        const int K1 = pin_s(p.K1);                                        // columns >= K1 come from A2 (skip-connection concat)
        const float* const arow2 = p.A2 != nullptr ? p.A2 + (size_t)gr * p.lda2 + kl - K1 : arow;
        const float* const wrow = argW + (size_t)rl * ldw + kl;           // W row rl; piece i adds 16 (i - 1) rows
        float* const lbase = lds + 4 * pw * 64;
        auto issue = [&](int kt) __attribute__((always_inline)) {
            float* const dst = lbase + (kt & 1) * STAGE;
            const int k0 = kt << 6;
            dma16((k0 < K1 ? arow : arow2) + k0, dst);                     // A rows 4 pw .. 4 pw + 3
#pragma unroll
            for (int i = 1; i < PPW; ++i) dma16(wrow + (size_t)(16 * (i - 1)) * ldw + k0, dst + 16 * i * 64);
        };
        // ONE stage in flight and every wait `vmcnt(0)`: LDS-DMA requests of a wave do not complete in issue order when their latencies
        // differ (an A piece that misses the L2 against W pieces that hit), so a counted wait with a younger stage in flight can be
        // satisfied by the wrong pieces (gemm_big.hip, header)
        issue(0);
        for (int kt = 0; kt < nk; ++kt) {
            wait_vm<0>();
            __builtin_amdgcn_s_barrier();          // A(kt): stage landed (and B(kt - 1) has passed: the other buffer is free)
            if (kt + 1 < nk) issue(kt + 1);
            __builtin_amdgcn_s_barrier();          // B(kt): stage consumed
        }
        __builtin_amdgcn_s_barrier();              // C tile staged (the producers take no part in the epilogue)
        return;
    }

    // ---------------------------------------------------------------------- consumers: wave w owns columns 64 w .. 64 w + 63
    float* const argY = pin_s(p.Y); float* const argYs = pin_s(p.Ys);
    const float* const argRes = pin_s(p.res);
    const int ldres = pin_s(p.ldres), ldy = pin_s(p.ldy);
    const int c = lane * 4;                        // epilogue: one row per wave-instruction, 4 columns per lane
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 bi = p.bias != nullptr ? ld4(p.bias + c) : zero;
    const bool ln = p.ln_g != nullptr;
    const f32x4 gg = ln ? ld4(p.ln_g + c) : zero, bb = ln ? ld4(p.ln_b + c) : zero;
    f32x4 rv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int gr = row0 + 4 * wave + e;
        rv[e] = (argRes != nullptr && gr < argM) ? ld4g(argRes + (size_t)gr * ldres + c) : zero;
    }

    f32x4 acc[RN];
#pragma unroll
    for (int j = 0; j < RN; ++j) acc[j] = zero;
    const int frow = lane & 15, fk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const float* sa = lds + (kt & 1) * STAGE;
        const float* sb = sa + (BM + 64 * wave) * 64;
        __builtin_amdgcn_s_barrier();              // A(kt)
        s16x8 ah[2], al[2], bh[2][RN], bl[2][RN];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int ch = 4 * g + fk, cl = 8 + 4 * g + fk;
            ah[g] = __builtin_bit_cast(s16x8, ld4(sa + frow * 64 + ((ch ^ frow) << 2)));
            al[g] = __builtin_bit_cast(s16x8, ld4(sa + frow * 64 + ((cl ^ frow) << 2)));
#pragma unroll
            for (int j = 0; j < RN; ++j) {
                const int r = j * 16 + frow;
                bh[g][j] = __builtin_bit_cast(s16x8, ld4(sb + r * 64 + ((ch ^ frow) << 2)));
                bl[g][j] = __builtin_bit_cast(s16x8, ld4(sb + r * 64 + ((cl ^ frow) << 2)));
            }
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int j = 0; j < RN; ++j) acc[j] = MFMA16_S16(al[g], bh[g][j], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < RN; ++j) acc[j] = MFMA16_S16(ah[g], bl[g][j], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < RN; ++j) acc[j] = MFMA16_S16(ah[g], bh[g][j], acc[j], 0, 0, 0);
            if (g == 0) {                          // both fragment sets are in registers: release the stage
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();      // B(kt)
            }
        }
    }

    // ---------------------------------------------------------------------- epilogue: C tile through LDS, then row-wise
    // stage (kt = nk - 1) & 1 may still be read by a slower consumer's second fragment set?  No: B(nk-1) was passed by every
    // wave after its LDS reads completed.  The other stage is idle as well (nothing was issued after stage nk-1).
    float* ct = lds;
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) ct[(4 * fk + r) * CLD + 64 * wave + 16 * j + frow] = acc[j][r];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x4 bi_ = bi, gg_ = gg, bb_ = bb;            // landed (vmcnt(0) above): no compiler-made waits in the store loop
    reg_touch(bi_); reg_touch(gg_); reg_touch(bb_);
#pragma unroll
    for (int e = 0; e < 4; ++e) reg_touch(rv[e]);
    f32x4 v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = ld4(ct + (4 * wave + e) * CLD + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[e][i] += bi_[i] + rv[e][i];
        if (!ln) continue;
        const float mean = wave_sum((v[e][0] + v[e][1]) + (v[e][2] + v[e][3])) * (1.f / 256.f);
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float d = v[e][i] - mean; sq += d * d; }
        const float rstd = rsqrtf(wave_sum(sq) * (1.f / 256.f) + LN_EPS);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[e][i] = (v[e][i] - mean) * rstd * gg_[i] + bb_[i];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int gr = row0 + 4 * wave + e;
        if (gr < argM) {
            if (argY != nullptr) st4g(argY + (size_t)gr * ldy + c, v[e]);
            if (argYs != nullptr) store_split4g(argYs + (size_t)gr * ldy, c, v[e]);
        }
    }
}

int launch_gemm_rowln(const RowLnArgs& a, hipStream_t s) {
    LADIFF_CHECK_ARG(a.A && a.W && (a.ln_g == nullptr) == (a.ln_b == nullptr) && (a.Y || a.Ys) && a.M >= 0 && a.K > 0);
    if (a.A2 != nullptr && (a.K1 % 64 != 0 || a.K1 <= 0 || a.K1 >= a.K || (a.lda2 % 4))) return LADIFF_ERR_SHAPE;
    if (a.K % 64 != 0 || (a.lda % 4) || (a.ldw % 4) || (a.ldy % 64) || (a.res && (a.ldres % 4))) return LADIFF_ERR_SHAPE;
    if (a.M == 0) return 0;
    RowLnArgs b = a;
    if (b.A2 == nullptr) b.K1 = b.K;
    hipLaunchKernelGGL(gemm_rowln_kernel, dim3((b.M + 15) / 16), dim3(512), 0, s, b);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// x' = res + out( SiLU( LN(sum_s P[s] + b2) * (1 + scale_step) + shift_step ) ): the split-K combine of ffn.linear2, the
// StylizationBlock (LayerNorm, AdaLN modulation, SiLU) and its output projection + residual (mdiff_transformer.py:161-163,
// :262) in ONE launch instead of row kernel + GEMM.  Same 16-row workgroups; the A operand never exists in global memory:
// the consumer waves build their 4 rows each (S-format, swizzled like a DMA'd tile) while the producers stream the first
// weight stages, then the K loop runs as above with A resident and only W in the two-stage ring.
__global__ __launch_bounds__(512) void combine_gemm_kernel(const CombineGemmArgs p) {
    constexpr int BM = 16, BN = 256, NK = 4;       // K = 256
    constexpr int STAGE = BN * 64;                 // W rows only: 64 KiB per stage
    constexpr int ASZ = NK * BM * 64;              // resident A: [NK][16 rows][256 B]
    constexpr int PPW = BN / 16;                   // 16 one-KiB pieces per producer wave per stage
    constexpr int RN = 4;
    constexpr int CLD = BN + 4;
    static_assert(2 * PPW <= 63, "vmcnt is 6 bits");

    __shared__ __attribute__((aligned(1024))) float lds[2 * STAGE + ASZ];
    float* const As = lds + 2 * STAGE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int argM = pin_s(p.M);
    const int row0 = blockIdx.x * BM;

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers: W stages
        const int pw = wave - 4;
        const int rl = 4 * pw + (lane >> 4);
        const int kl = ((lane & 15) ^ rl) << 2;
        const float* const wrow = p.W + (size_t)rl * p.ldw + kl;
        const int ldw = pin_s(p.ldw);
        float* const lbase = lds + 4 * pw * 64;
        auto issue = [&](int kt) __attribute__((always_inline)) {
            float* const dst = lbase + (kt & 1) * STAGE;
#pragma unroll
            for (int i = 0; i < PPW; ++i) dma16(wrow + (size_t)(16 * i) * ldw + (kt << 6), dst + 16 * i * 64);
        };
        issue(0);                                  // one stage in flight, `vmcnt(0)` waits only (see gemm_rowln_kernel above)
        for (int kt = 0; kt < NK; ++kt) {
            wait_vm<0>();
            __builtin_amdgcn_s_barrier();          // A(kt): W stage landed (kt = 0: and the A rows are built)
            if (kt + 1 < NK) issue(kt + 1);
            __builtin_amdgcn_s_barrier();          // B(kt)
        }
        __builtin_amdgcn_s_barrier();              // C tile staged
        return;
    }

    // ---------------------------------------------------------------------- consumers
    float* const argY = pin_s(p.Y); float* const argYs = pin_s(p.Ys);
    const int ldy = pin_s(p.ldy);
    const int c = lane * 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    // prologue: rows 4 w .. 4 w + 3 of the tile, one row per wave-instruction (reduce_rows_kernel, RED_LN_MOD)
    {
        const int step = p.d_step != nullptr ? *p.d_step : 0;
        const float* t = p.tab + (size_t)step * p.tab_step_stride;
        const f32x4 sc = ld4(t + c), sh = ld4(t + 256 + c);
        const f32x4 b2 = p.bias2 != nullptr ? ld4(p.bias2 + c) : zero;
        const f32x4 gg = ld4(p.ln_g + c), bb = ld4(p.ln_b + c);
        f32x4 pl[4][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int gr = row0 + 4 * wave + e; gr = gr < argM ? gr : argM - 1;
#pragma unroll
            for (int s = 0; s < 4; ++s) pl[e][s] = s < p.S ? ld4(p.P + (size_t)s * p.plane + (size_t)gr * BN + c) : zero;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ((pl[e][0][i] + pl[e][1][i]) + pl[e][2][i]) + pl[e][3][i] + b2[i];
            const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.f / 256.f);
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float d = v[i] - mean; sq += d * d; }
            const float rstd = rsqrtf(wave_sum(sq) * (1.f / 256.f) + LN_EPS);
            s16x4 hi, lo;
            {
                f32x4 u;
#pragma unroll
                for (int i = 0; i < 4; ++i) u[i] = silu(((v[i] - mean) * rstd * gg[i] + bb[i]) * (1.f + sc[i]) + sh[i]);
                split4(u, hi, lo);
            }
            // column c of row r lives in K stage c / 64, 16-byte slot (cc / 8) [hi] and 8 + cc / 8 [lo], slot ^ (r & 15)
            const int r = 4 * wave + e, kt = c >> 6, cc = c & 63;
            char* rowp = reinterpret_cast<char*>(As + (kt * BM + r) * 64);
            *reinterpret_cast<s16x4*>(rowp + ((((cc >> 3)) ^ r) << 4) + ((cc & 7) << 1)) = hi;
            *reinterpret_cast<s16x4*>(rowp + (((8 + (cc >> 3)) ^ r) << 4) + ((cc & 7) << 1)) = lo;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // epilogue operands
    const f32x4 bi = p.bias != nullptr ? ld4(p.bias + c) : zero;
    f32x4 rv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int gr = row0 + 4 * wave + e;
        rv[e] = (p.res != nullptr && gr < argM) ? ld4(p.res + (size_t)gr * p.ldres + c) : zero;
    }

    f32x4 acc[RN];
#pragma unroll
    for (int j = 0; j < RN; ++j) acc[j] = zero;
    const int frow = lane & 15, fk = lane >> 4;
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
        const float* sa = As + kt * BM * 64;
        const float* sb = lds + (kt & 1) * STAGE + (64 * wave) * 64;
        __builtin_amdgcn_s_barrier();              // A(kt)
        s16x8 ah[2], al[2], bh[2][RN], bl[2][RN];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int ch = 4 * g + fk, cl = 8 + 4 * g + fk;
            ah[g] = __builtin_bit_cast(s16x8, ld4(sa + frow * 64 + ((ch ^ frow) << 2)));
            al[g] = __builtin_bit_cast(s16x8, ld4(sa + frow * 64 + ((cl ^ frow) << 2)));
#pragma unroll
            for (int j = 0; j < RN; ++j) {
                const int r = j * 16 + frow;
                bh[g][j] = __builtin_bit_cast(s16x8, ld4(sb + r * 64 + ((ch ^ frow) << 2)));
                bl[g][j] = __builtin_bit_cast(s16x8, ld4(sb + r * 64 + ((cl ^ frow) << 2)));
            }
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int j = 0; j < RN; ++j) acc[j] = MFMA16_S16(al[g], bh[g][j], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < RN; ++j) acc[j] = MFMA16_S16(ah[g], bl[g][j], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < RN; ++j) acc[j] = MFMA16_S16(ah[g], bh[g][j], acc[j], 0, 0, 0);
            if (g == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();      // B(kt)
            }
        }
    }

    float* ct = lds;
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) ct[(4 * fk + r) * CLD + 64 * wave + 16 * j + frow] = acc[j][r];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x4 bi_ = bi;
    reg_touch(bi_);
#pragma unroll
    for (int e = 0; e < 4; ++e) reg_touch(rv[e]);
    f32x4 v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = ld4(ct + (4 * wave + e) * CLD + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[e][i] += bi_[i] + rv[e][i];
        const int gr = row0 + 4 * wave + e;
        if (gr < argM) {
            if (argY != nullptr) st4g(argY + (size_t)gr * ldy + c, v[e]);
            if (argYs != nullptr) store_split4g(argYs + (size_t)gr * ldy, c, v[e]);
        }
    }
}

int launch_combine_gemm(const CombineGemmArgs& a, hipStream_t s) {
    LADIFF_CHECK_ARG(a.P && a.W && a.ln_g && a.ln_b && a.tab && (a.Y || a.Ys) && a.M >= 0 && a.S >= 1 && a.S <= 4);
    if ((a.ldw % 4) || (a.ldy % 64) || (a.res && (a.ldres % 4))) return LADIFF_ERR_SHAPE;
    if (a.M == 0) return 0;
    hipLaunchKernelGGL(combine_gemm_kernel, dim3((a.M + 15) / 16), dim3(512), 0, s, a);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

}  // namespace ladiff
