// fp32-input MFMA GEMM for the large-M (decoder) side: Y[M,N] = LN?( act(A[M,K] . W[N,K]^T + b) + res ), M ~ 25 K rows.
//
//   * 128x128 tiles (4 waves, 64x64 each) or, when LayerNorm follows, 64x256 tiles that own whole rows (4 waves, 32x128);
//   * K in stages of 32 floats, two stages in LDS, filled by LDS-DMA (`global_load_lds_dwordx4`, one 1-KiB piece = 8 rows
//     x 128 B per wave-instruction, no staging VGPRs); stage kt+1 is requested behind the barrier in front of stage kt's MFMAs
//     and is the ONLY request in flight when a wave waits for it - every wait is `vmcnt(0)`; two workgroups per CU hide each
//     other's barriers, epilogues and what is left of the load latency.
//     Why not two stages in flight behind a counted `vmcnt(PPW)` (rounds 1 - 4): LDS-DMA requests of one wave do NOT complete in
//     issue order when their latencies differ - an activation piece that misses the L2 is overtaken by the next stage's weight
//     pieces that hit, `vmcnt(PPW)` is then satisfied by the wrong eight and the MFMAs read rows that have not landed.  Never
//     seen with the GPU to itself; with a second stream's kernels loading the memory system 1 - 2 % of the decodes came out
//     with a wrong tile (scripts/two_streams_*.py, profiles/r5/11_*: 60 of 3000 against 0 of 3000 in this form, same speed);
//   * LDS rows are 128 B; the 16-byte chunk c of row r sits in slot c ^ ((r >> 1) & 7) (applied on the DMA source
//     address and on the fragment reads) -> conflict-free ds_read_b128 for the 32x32x2 operand layout;
//   * blockIdx is remapped so that the N/BN workgroups that share an A row tile run on the same XCD (they hit in its L2);
//   * epilogue: the accumulators go through LDS once so that bias / activation / residual / LayerNorm run row-wise -
//     whole 512-byte or 1-KiB rows per wave-instruction for the residual loads and the stores, DPP wave reductions for
//     the LayerNorm statistics - instead of 4-byte accesses in the MFMA accumulator layout.  The residual rows are
//     fetched before the last K stage so their latency hides under its MFMAs.
#include "gemm.h"

namespace ladiff {

namespace {
constexpr int BKB = 32;
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void dma16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
}  // namespace

template <int BM, int BN, int WM, int WN, bool LN>
__device__ __forceinline__ void gemm_big_body(const GemmArgs& p) {
    constexpr int NW = 4;
    static_assert(WM * WN == NW, "four waves");
    constexpr int TMW = BM / WM, TNW = BN / WN, RM = TMW / 32, RN = TNW / 32;
    constexpr int ROWS = BM + BN;
    constexpr int STAGE = ROWS * BKB;              // floats
    constexpr int GRP = 8 * NW;                    // rows covered by one piece per wave
    constexpr int GA = BM / GRP, GT = ROWS / GRP;
    static_assert(BM % GRP == 0 && BN % GRP == 0, "tiles are multiples of 32 rows");
    constexpr int CLD = BN + 4;                    // row stride of the staged C tile
    constexpr int LDSF = (2 * STAGE > BM * CLD) ? 2 * STAGE : BM * CLD;
    constexpr int LPR = BN / 4;                    // lanes per output row (16-byte units)
    constexpr int RPI = 64 / LPR;                  // rows per wave-instruction
    constexpr int EI = BM / NW / RPI;              // epilogue instructions per wave
    static_assert(!LN || BN == 256, "LayerNorm needs whole rows");

    __shared__ __attribute__((aligned(1024))) float lds[LDSF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile map: workgroups b and b+8 share an XCD; give each XCD whole row tiles
    const int nbn = p.N / BN;
    const int nbm = (p.M + BM - 1) / BM;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int bm = (local / nbn) * 8 + xcd, bn = local % nbn;
    if (bm >= nbm) return;
    const int row0 = bm * BM, col0 = bn * BN;
    const int nk = p.K / BKB;

    // ---- LDS-DMA of one K stage: wave w brings rows 32 g + 8 w .. + 7 of every 32-row group g of [A | W]
    const int rl = 8 * wave + (lane >> 3);                         // row within a 32-row group
    const int kl = (((lane & 7) ^ ((rl >> 1) & 7)) << 2);          // swizzled source chunk (floats)
    float* const lbase = lds + 8 * wave * BKB;
    auto issue = [&](int kt, int buf) __attribute__((always_inline)) {
        const int k0 = kt * BKB;
        const float* abase; int ald;
        if (k0 < p.K1) { abase = p.A + k0; ald = p.lda; } else { abase = p.A2 + (k0 - p.K1); ald = p.lda2; }
#pragma unroll
        for (int g = 0; g < GT; ++g) {
            const float* src;
            if (g < GA) {
                int gr = row0 + GRP * g + rl; gr = gr < p.M ? gr : p.M - 1;
                src = abase + (size_t)gr * ald + kl;
            } else {
                src = p.W + (size_t)(col0 + GRP * (g - GA) + rl) * p.ldw + k0 + kl;
            }
            dma16(src, lbase + buf * STAGE + GRP * g * BKB);
        }
    };

    f32x16 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fk = lane >> 5;
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const float* sa = lds + buf * STAGE + (wm * TMW) * BKB;
        const float* sb = lds + buf * STAGE + (BM + wn * TNW) * BKB;
#pragma unroll
        for (int g = 0; g < BKB / 8; ++g) {
            const int c = 2 * g + fk;
            f32x4 fa[RM], fb[RN];
#pragma unroll
            for (int i = 0; i < RM; ++i) { const int r = i * 32 + frow; fa[i] = ld4(sa + r * BKB + ((c ^ ((r >> 1) & 7)) << 2)); }
#pragma unroll
            for (int j = 0; j < RN; ++j) { const int r = j * 32 + frow; fb[j] = ld4(sb + r * BKB + ((c ^ ((r >> 1) & 7)) << 2)); }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < RM; ++i)
#pragma unroll
                    for (int j = 0; j < RN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        }
    };

    // epilogue geometry: wave w owns rows w*BM/4 .. of the tile; instruction e covers RPI rows, lane -> (row, 4 columns)
    const int er = wave * (BM / NW) + lane / LPR;      // + e * RPI
    const int ec = 4 * (lane % LPR);
    f32x4 rv[EI];
#pragma unroll
    for (int e = 0; e < EI; ++e) rv[e] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        wait_vm<0>();                                  // this wave's pieces of stage kt: nothing younger is in flight (header)
        __builtin_amdgcn_s_barrier();                  // every wave's have landed; every wave has left the other buffer
        if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
        if (kt == nk - 1 && p.res != nullptr) {        // residual rows: latency hides under the last stage's MFMAs
#pragma unroll
            for (int e = 0; e < EI; ++e) {
                const int gr = row0 + er + e * RPI;
                if (gr < p.M) rv[e] = ld4(p.res + (size_t)gr * p.ldres + col0 + ec);
            }
        }
        compute(buf);
    }
    __builtin_amdgcn_s_barrier();                      // every wave has left the last stage: the C tile takes its place

    // ---- accumulators -> LDS (C tile, row stride BN + 4)
    float* ct = lds;
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ct[(wm * TMW + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * CLD + wn * TNW + j * 32 + (lane & 31)] = acc[i][j][r];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int e = 0; e < EI; ++e) reg_touch(rv[e]);  // landed: keep compiler-made vmcnt(0) out of the store loop

    // ---- row-wise: bias, activation, residual, LayerNorm(s), store
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr) bv = ld4(p.bias + col0 + ec);
    f32x4 g1 = bv, b1 = bv, g2 = bv, b2 = bv;
    if constexpr (LN) {
        g1 = ld4(p.ln_g + ec); b1 = ld4(p.ln_b + ec);
        if (p.ln2_g != nullptr) { g2 = ld4(p.ln2_g + ec); b2 = ld4(p.ln2_b + ec); }
    }
    act_dispatch(p.act, [&](auto ACT) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < EI; ++e) {
        const int lr = er + e * RPI;
        const int gr = row0 + lr;
        f32x4 v = ld4(ct + lr * CLD + ec);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = act_c<decltype(ACT)::value>(v[q] + bv[q]) + rv[e][q];
        if constexpr (LN) {
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                if (pass == 1 && p.ln2_g == nullptr) break;
                const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.f / 256.f);
                float sq = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) { const float d = v[q] - mean; sq += d * d; }
                const float rstd = rsqrtf(wave_sum(sq) * (1.f / 256.f) + LN_EPS);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = (v[q] - mean) * rstd * (pass ? g2[q] : g1[q]) + (pass ? b2[q] : b1[q]);
            }
        }
        if (gr < p.M) st4(p.Y + (size_t)gr * p.ldy + col0 + ec, v);
    }
    });
}

// ---------------------------------------------------------------------------------------------- f16x3 variant
// Operands in S-format (common.h), every product is hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 (fp32 accumulate).
// One 128x128 tile per workgroup, 4 waves of 64x64.  A K stage is HALF an S-format block (32 k: 64 B hi + 64 B lo per
// row), so the two stages take 64 KiB and the staged C tile 66 KiB: TWO workgroups fit on a CU, and one's epilogue (stores
// run at HBM write speed) and barrier stalls hide under the other's MFMAs.  Measured at M = 25088 against the two designs
// it replaced - 64-k stages with one workgroup per CU, and a persistent producer/consumer workgroup with register-direct
// stores: 51 us vs 60 / 70 us (N=768, K=256), 91 vs 129 / 105 us (N=1024, K=256, GELU), 434 vs 504 / 483 us (N=768,
// K=3072).  LDS rows are 128 B; 16-byte slot c of row r sits at c ^ ((r >> 1) & 7) (DMA source side and fragment reads),
// conflict-free for the 16x16x32 operand layout.  Output: fp32 and/or S-format, bias / activation / residual fused;
// LayerNorm runs as a row kernel afterwards (a 256-wide tile would move 4x the operand bytes per FLOP).
template <int BM, int BN>
__device__ __forceinline__ void gemm_big_split_body(const GemmArgs& p) {
    constexpr int NW = 4, WM = 2, WN = 2;
    constexpr int TMW = BM / WM, TNW = BN / WN, RM = TMW / 16, RN = TNW / 16;
    constexpr int ROWS = BM + BN;
    constexpr int STAGE = ROWS * 32;               // floats (128 B per row)
    constexpr int GRP = 8 * NW;                    // rows covered by one piece per wave (a piece = 8 rows x 128 B)
    constexpr int GA = BM / GRP, GT = ROWS / GRP;
    constexpr int CLD = BN + 4;
    constexpr int LPR = BN / 4, RPI = 64 / LPR, EI = BM / NW / RPI;
    constexpr int LDSF = (2 * STAGE > BM * CLD) ? 2 * STAGE : BM * CLD;

    __shared__ __attribute__((aligned(1024))) float lds[LDSF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int argM = pin_s(p.M), argK1 = pin_s(p.K1), ldy = pin_s(p.ldy), act = pin_s(p.act);
    float* const argY = pin_s(p.Y); float* const argYs = pin_s(p.Ys);
    const int nbn = p.N / BN;
    const int nbm = (argM + BM - 1) / BM;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int bm = (local / nbn) * 8 + xcd, bn = local % nbn;
    if (bm >= nbm) return;
    const int row0 = bm * BM, col0 = bn * BN;
    const int ksplit = pin_s(p.ksplit), kz = ksplit > 1 ? (int)blockIdx.y : 0;       // (the batch kernel passes ksplit = 1: its blockIdx.y is the argument set)
    const int nk_all = p.K / 32, nk = nk_all / ksplit, kt0 = kz * nk;

    const int rl = 8 * wave + (lane >> 3);                        // row within a 32-row group
    const int cs = (lane & 7) ^ ((rl >> 1) & 7);                  // source slot that lands in LDS slot (lane & 7)
    const int kl = (cs < 4 ? cs * 4 : 32 + (cs - 4) * 4);         // float offset inside the 64-float block: hi | lo halves
    float* const lbase = lds + 8 * wave * 32;
    auto issue = [&](int ktl, int buf) __attribute__((always_inline)) {
        const int kt = kt0 + ktl;
        const int k0 = (kt >> 1) * 64, hf = (kt & 1) * 16;       // block start (fp32 columns), half offset (floats)
        const float* abase; int ald;
        if (k0 < argK1) { abase = p.A + k0; ald = p.lda; } else { abase = p.A2 + (k0 - argK1); ald = p.lda2; }
#pragma unroll
        for (int g = 0; g < GT; ++g) {
            const float* src;
            if (g < GA) {
                int gr = row0 + GRP * g + rl; gr = gr < argM ? gr : argM - 1;
                src = abase + (size_t)gr * ald + hf + kl;
            } else {
                src = p.W + (size_t)(col0 + GRP * (g - GA) + rl) * p.ldw + k0 + hf + kl;
            }
            dma16(src, lbase + buf * STAGE + GRP * g * 32);
        }
    };

    f32x4 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fk = lane >> 4;
    const int sw = (frow >> 1) & 7;                // tiles start at multiples of 16 rows: (r >> 1) & 7 == (frow >> 1) & 7
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const float* sa = lds + buf * STAGE + (wm * TMW) * 32;
        const float* sb = lds + buf * STAGE + (BM + wn * TNW) * 32;
        s16x8 ah[RM], al[RM], bh[RN], bl[RN];
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            const int r = i * 16 + frow;
            ah[i] = __builtin_bit_cast(s16x8, ld4(sa + r * 32 + ((fk ^ sw) << 2)));
            al[i] = __builtin_bit_cast(s16x8, ld4(sa + r * 32 + (((4 + fk) ^ sw) << 2)));
        }
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const int r = j * 16 + frow;
            bh[j] = __builtin_bit_cast(s16x8, ld4(sb + r * 32 + ((fk ^ sw) << 2)));
            bl[j] = __builtin_bit_cast(s16x8, ld4(sb + r * 32 + (((4 + fk) ^ sw) << 2)));
        }
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j) acc[i][j] = MFMA16_S16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j) acc[i][j] = MFMA16_S16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j) acc[i][j] = MFMA16_S16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    };

    const int er = wave * (BM / NW) + lane / LPR;
    const int ec = 4 * (lane % LPR);
    f32x4 rv[EI];
#pragma unroll
    for (int e = 0; e < EI; ++e) rv[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};

    // one stage in flight, every wait `vmcnt(0)` (see the header: LDS-DMA requests do not complete in issue order)
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();              // stage kt has landed for every wave; every wave has left the other buffer
        if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
        if (kt == nk - 1 && ksplit == 1) {         // epilogue operands: their latency hides under the last stage's MFMAs
            if (p.bias != nullptr) bv = ld4(p.bias + col0 + ec);
            if (p.res != nullptr) {
#pragma unroll
                for (int e = 0; e < EI; ++e) {
                    const int gr = row0 + er + e * RPI;
                    if (gr < argM) rv[e] = ld4(p.res + (size_t)gr * p.ldres + col0 + ec);
                }
            }
        }
        compute(buf);
    }
    __builtin_amdgcn_s_barrier();                  // every wave has left the last stage: the C tile takes its place

    float* ct = lds;
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                ct[(wm * TMW + i * 16 + 4 * (lane >> 4) + r) * CLD + wn * TNW + j * 16 + (lane & 15)] = acc[i][j][r];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    reg_touch(bv);
#pragma unroll
    for (int e = 0; e < EI; ++e) reg_touch(rv[e]);

    if (ksplit > 1) {                              // a K part: raw sums to its plane
        float* const Yz = argY + (size_t)kz * p.plane;
#pragma unroll
        for (int e = 0; e < EI; ++e) {
            const int lr = er + e * RPI;
            const int gr = row0 + lr;
            if (gr < argM) st4g(Yz + (size_t)gr * ldy + col0 + ec, ld4(ct + lr * CLD + ec));
        }
        return;
    }
    act_dispatch(act, [&](auto ACT) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < EI; ++e) {
            const int lr = er + e * RPI;
            const int gr = row0 + lr;
            f32x4 v = ld4(ct + lr * CLD + ec);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = act_c<decltype(ACT)::value>(v[q] + bv[q]) + rv[e][q];
            if (gr < argM) {
                if (argY != nullptr) st4g(argY + (size_t)gr * ldy + col0 + ec, v);
                if (argYs != nullptr) store_split4g(argYs + (size_t)gr * ldy, col0 + ec, v);
            }
        }
    });
}

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void gemm_big_split_kernel(const GemmArgs p) { gemm_big_split_body<BM, BN>(p); }
// up to GEMM_BATCH_MAX independent same-shape f16x3 GEMMs as one launch (grid.y = argument set): the nine layers' c-table GEMMs
template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void gemm_big_split_batch_kernel(const GemmBatch b) { gemm_big_split_body<BM, BN>(b.a[blockIdx.y]); }   // (launch_gemm_big_batch refuses ksplit > 1)

template <int BM, int BN, int WM, int WN, bool LN>
__global__ __launch_bounds__(256) void gemm_big_kernel(const GemmArgs p) { gemm_big_body<BM, BN, WM, WN, LN>(p); }
template <int BM, int BN, int WM, int WN, bool LN>
__global__ __launch_bounds__(256) void gemm_big_batch_kernel(const GemmBatch b) { gemm_big_body<BM, BN, WM, WN, LN>(b.a[blockIdx.y]); }

template <int BM, int BN, int WM, int WN, bool LN>
static int launch_big_batch(const GemmBatch& b, int n, hipStream_t s) {
    const GemmArgs& a = b.a[0];
    const int nbm = (a.M + BM - 1) / BM, nbn = a.N / BN;
    const int grid = ((nbm + 7) / 8) * 8 * nbn;
    hipLaunchKernelGGL((gemm_big_batch_kernel<BM, BN, WM, WN, LN>), dim3(grid, n), dim3(256), 0, s, b);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

template <int BM, int BN, int WM, int WN, bool LN>
static int launch_big(const GemmArgs& a, hipStream_t s) {
    const int nbm = (a.M + BM - 1) / BM, nbn = a.N / BN;
    const int grid = ((nbm + 7) / 8) * 8 * nbn;
    hipLaunchKernelGGL((gemm_big_kernel<BM, BN, WM, WN, LN>), dim3(grid), dim3(256), 0, s, a);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// true when the shape is served by this kernel family (otherwise the caller uses the staged kernel of gemm.hip)
bool gemm_big_supported(const GemmArgs& a) {
    if (a.split) return a.M > 0 && a.K % 64 == 0 && a.K1 % 64 == 0 && a.N % 128 == 0 && a.lda % 4 == 0 && a.ldw % 4 == 0 && (!a.A2 || a.lda2 % 4 == 0) &&
                        a.ldy % 64 == 0 && !a.ln_g && !a.mod && !a.row_len && !a.row_map && a.post_act == ACT_NONE && (a.Y || a.Ys);
    if (a.M < 4096 || a.K % BKB || a.K1 % BKB || (a.lda % 4) || (a.ldw % 4) || (a.A2 && (a.lda2 % 4))) return false;
    if ((a.ldy % 4) || (a.res && (a.ldres % 4)) || a.mod || a.row_len || a.row_map || a.post_act != ACT_NONE) return false;
    if (a.ln_g != nullptr) return a.N == 256;
    return a.N % 128 == 0;
}

int launch_gemm_big(const GemmArgs& a, hipStream_t s) {
    if (a.split) {
        const int nbm = (a.M + 127) / 128, nbn = a.N / 128;
        // Few row tiles (the ragged CLIP tower: ~2,300 rows, N = 768): 128-row tiles leave the launch one partial round of workgroups whose
        // length is ONE tile's k loop (K = 3072: 82 us for 108 workgroups); 64-row tiles double the workgroups and halve each one's work
        const int ks = a.ksplit > 1 ? a.ksplit : 1;
        if (ks > 1 && ((a.K / 32) % ks != 0 || a.Y == nullptr || a.Ys != nullptr || a.plane == 0)) return LADIFF_ERR_ARG;
        // (measured, round 6: K parts on 256x128 tiles - half the LDS-DMA bytes per flop, one workgroup per CU - run CLIP's fc2 in 81 us
        // against 54 us on the 64-row tiles below and 70 us without K parts, and its fc1 - N = 3072, no K parts - in 103 us against 66 us on 128x128
        // tiles, two workgroups per CU: one workgroup per CU has nobody to hide its stage's load latency behind.  Not instantiated.  Nor is load latency what
        // bounds the 64-row tiles: with THREE LDS stages and two in flight (wave pairs own alternate stages, one per wave, every wait
        // vmcnt(0)) they run 36.3 us on average against 37.3 - the ~25 GB/s a CU takes in through LDS-DMA is the bound either way)
        if (nbm * nbn <= 256 && a.M > 64) {
            const int nbm64 = (a.M + 63) / 64;
            hipLaunchKernelGGL((gemm_big_split_kernel<64, 128>), dim3(((nbm64 + 7) / 8) * 8 * nbn, ks), dim3(256), 0, s, a);
        } else {
            hipLaunchKernelGGL((gemm_big_split_kernel<128, 128>), dim3(((nbm + 7) / 8) * 8 * nbn, ks), dim3(256), 0, s, a);
        }
        LADIFF_LAUNCH_CHECK();
        return 0;
    }
    if (a.ksplit > 1) return LADIFF_ERR_ARG;      // K parts: the split-mode kernel only
    if (a.ln_g != nullptr) return launch_big<64, 256, 2, 2, true>(a, s);
    return launch_big<128, 128, 2, 2, false>(a, s);
}

// (the caller checked gemm_big_supported on the common shape)
int launch_gemm_big_batch(const GemmBatch& b, int n, hipStream_t s) {
    for (int i = 0; i < n; ++i) if (b.a[i].ksplit > 1) return LADIFF_ERR_ARG;
    if (b.a[0].split) {
        const GemmArgs& a = b.a[0];
        const int nbm = (a.M + 127) / 128, nbn = a.N / 128;
        hipLaunchKernelGGL((gemm_big_split_batch_kernel<128, 128>), dim3(((nbm + 7) / 8) * 8 * nbn, n), dim3(256), 0, s, b);
        LADIFF_LAUNCH_CHECK();
        return 0;
    }
    if (b.a[0].ln_g != nullptr) return launch_big_batch<64, 256, 2, 2, true>(b, n, s);
    return launch_big_batch<128, 128, 2, 2, false>(b, n, s);
}

}  // namespace ladiff
