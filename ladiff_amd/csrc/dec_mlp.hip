// The decoder layer's feed-forward block as ONE kernel (f16x3 mode):  y = LN3( x + W2 gelu(W1 x + b1) + b2 )  [+ a second LayerNorm]
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
//     (`global_load_lds_dwordx4`) in GROUPS of four stages, each wave with ONE group's pieces in flight and every wait for them
//     `vmcnt(0)` with nothing younger outstanding (round 6, below); a stage = 128 weight rows x 32 k (64 B hi + 64 B lo per row),
//     the layout and slot swizzle of gemm_big_split_kernel (conflict-free ds_read_b128 for the 16x16x32 operand layout).  Per
//     128 hidden columns: 8 stages of W1 (k = the 256 model columns) then 8 stages of W2 (its 256 rows in two halves x the 128
//     hidden columns in four k-steps).  Per 128 rows that is 2 MB of ingest for 403 MFLOP (bf16): 200 FLOP per byte, against
//     96 for a 128x128x256 GEMM tile - the MFMA pipe, not the LDS fill, is the bound.
//   * workgroup forms (template): eight waves x 16 rows (default from 160 row tiles up since round 6), four waves x 32 rows (a weight
//     fragment read from LDS feeds two row tiles; the default of rounds 3 - 5), four waves x 16 rows (64-row workgroups, twice as many,
//     when there are few rows).
//   * the weight fragments travel through a ring of four tile buffers in registers that runs ACROSS the stage boundaries (the
//     barrier of stage u + 1 stands in front of tile 5 of stage u), each fragment requested 17 MFMAs ahead of its use, the two
//     ds_read_b128 of a tile behind the previous tile's first MFMA.
//   * epilogue: bias / residual / LayerNorm(s) in the accumulator layout with the per-column vectors in LDS; the tile then goes
//     through the (now free) ring so that every store writes one whole 1-KiB row.
// What the time is made of (profiles/r3/13, 25088 rows, 99 us): with the GELU replaced by the identity 84 us, without the epilogue
// 84 us (of which ~10 us are the 77 MB all workgroups move at the same moment), without the LDS-DMA 94 us.  NOTE on the builds that
// take the MFMA operands from registers nothing writes (no fragment reads: 78 us; reads issued and waited for but not used: 80 us):
// they do not show that the reads cost 20 us - constant operands let the chip clock the matrix pipe ~25 % higher than random
// weights do (MI355X_MICROARCH.md: 1.5 - 1.7 GHz in MFMA-dense loops on random data, 2.39 GHz on zeros).  Deeper fragment prefetch
// (10 -> 17 MFMAs) moved the kernel by 2 us: the waits are not what binds.
#include <mutex>
#include "model.h"
#include "tile_mma.h"

namespace ladiff {

namespace {

constexpr int MLP_NS = 8;                      // ring stages: two groups of MLP_GRP
constexpr int MLP_GRP = 4;                     // stages a wave requests as ONE batch and waits for with vmcnt(0)
constexpr int MLP_STAGE = 16384;               // bytes: 128 weight rows x 128 B
constexpr int MLP_LDS = MLP_NS * MLP_STAGE + FF * 4 + 5 * D * 4;      // ring + linear1's bias + b2, gamma / beta of the LayerNorm(s)

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
template <int OFF>
__device__ __forceinline__ void fetch16_keep(u32x4_m& v, unsigned addr) {   // the destination stays allocated between calls (DIAG 6: nothing else keeps it)
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(v) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_lgkm(u32x4_m& a, u32x4_m& b, u32x4_m& c, u32x4_m& d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
__device__ __forceinline__ s16x8 as_bf(const u32x4_m v) { return __builtin_bit_cast(s16x8, v); }

}  // namespace

// NW = waves per workgroup, RT = 16-row tiles per wave: a workgroup owns 16 RT NW rows.
//   <4, 2>: 128 rows, ONE wave per SIMD with 32 rows: every weight fragment read from LDS feeds two row tiles (6 MFMAs per 2 KB
//           instead of 3: with 16 rows per wave the kernel ran at the LDS-read latency, 1100 - 1800 cycles per stage for 384 - 768 of
//           MFMA, profiles/r3), the wave's four LDS-DMA pieces of a stage are issued one after each group of 12 MFMAs;
//   <8, 1>: 128 rows, two waves per SIMD with 16 rows each (the first half of the waves issues its DMA early in a stage, the second
//           half late: partners must not do the same thing at the same time);
//   <4, 1>: 64 rows (twice as many workgroups when there are few rows).
// DIAG (timing experiments only, results are garbage): 1 = no LDS-DMA inside the stage loop, 2 = no MFMAs, 3 = no LDS fragment reads,
// 4 = no GELU (identity), 5 = no epilogue (residual loads, LayerNorm, stores), 6 = fragment reads issued but never waited for or used,
// 7 = fragment reads waited for as usual, the MFMAs take other registers
template <int NW, int RT, int DIAG = 0>
__global__ __launch_bounds__(64 * NW, NW / 4) void dec_mlp_kernel(const MlpArgs p) {
    constexpr int NT = 64 * NW, RW = 16 * RT, BM = RW * NW, PPW = 16 / NW;   // threads, rows per wave / workgroup, DMA pieces per wave and stage
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    float* const b1s = reinterpret_cast<float*>(lds + MLP_NS * MLP_STAGE);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fk = lane >> 4;
    const int M = p.M;
    int myrow[RT]; bool live[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        myrow[rt] = blockIdx.x * BM + RW * wave + 16 * rt + frow;
        live[rt] = myrow[rt] < M;
        myrow[rt] = live[rt] ? myrow[rt] : M - 1;
    }

    // ---- this wave's x rows as operand fragments: k-step s (32 columns) -> hi / lo 16 bytes of lane (row frow, k 8 fk .. + 7)
    s16x8 xh[RT][8], xl[RT][8];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const char* xr = reinterpret_cast<const char*>(p.xs) + (size_t)myrow[rt] * 1024 + fk * 16;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            xh[rt][s] = *reinterpret_cast<const s16x8*>(xr + (s >> 1) * 256 + (s & 1) * 64);
            xl[rt][s] = *reinterpret_cast<const s16x8*>(xr + (s >> 1) * 256 + (s & 1) * 64 + 128);
        }
    }
    for (int i = tid; i < FF / 4; i += NT) st4(b1s + 4 * i, ld4(p.b1 + 4 * i));
    // the epilogue's per-column vectors: from LDS a lane's 16 B cost a ds_read, from memory each was a scattered 16-row load (5 x 32
    // of them per lane: ~6 us of the kernel)
    float* const prm = b1s + FF;                                          // [5][256]: b2, g3, be3, g4, be4
    for (int i = tid; i < 5 * D / 4; i += NT) {
        const int which = i / (D / 4), c4 = i % (D / 4);
        const float* src = which == 0 ? p.b2 : which == 1 ? p.g3 : which == 2 ? p.be3 : which == 3 ? p.g4 : p.be4;
        if (src != nullptr) st4(prm + 4 * i, ld4(src + 4 * c4));
    }

    // ---- LDS-DMA of one stage: wave w brings the PPW pieces (a piece = 8 LDS rows x 128 B) q = w + NW i: LDS rows 8 q + (lane >> 3)
    const int cs = (lane & 7) ^ ((lane >> 4) & 3) ^ (4 * (wave & 1));    // source slot that lands in LDS slot (lane & 7): slot ^ ((row >> 1) & 7), row = 8 q + (lane >> 3), q = wave (mod 2) for even NW
    const int koff = (cs < 4 ? cs * 4 : 32 + (cs - 4) * 4);              // floats inside the 64-float S-block: hi | lo halves
    int src[PPW];                                                        // weight rows (within the panel) behind this lane's pieces
#pragma unroll
    for (int i = 0; i < PPW; ++i) src[i] = pi_row(8 * (wave + NW * i) + (lane >> 3));
    char* const dma_dst = lds + (8 * wave) * 128;                        // + 8 NW i rows; the hardware adds lane * 16
    // piece i of stage (hs, u): u < 8: W1 rows 128 hs .., k-step u;  u >= 8: v = u - 8, k-step c = v >> 1 of the hidden slice, W2 rows 128 (v & 1) ..
    auto issue = [&](int hs, auto uc, auto slotc, auto ic) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value, slot = decltype(slotc)::value, i = decltype(ic)::value;
        const float* base; int ldw, n0, kb, half;
        const int hsv = hs < 0 ? 0 : hs;
        if constexpr (u < 8) { base = p.w1; ldw = D; n0 = 128 * hsv; kb = u >> 1; half = u & 1; }
        else { constexpr int v = u - 8, c = v >> 1; base = p.w2; ldw = FF; n0 = 128 * (v & 1); kb = 2 * hsv + (c >> 1); half = c & 1; }
        const float* s0 = base + (size_t)(n0 + src[i]) * ldw + kb * 64 + half * 16 + koff;
        if constexpr (DIAG == 1) { if (hs >= 0) return; }
        __builtin_amdgcn_global_load_lds(s0, (__attribute__((address_space(3))) void*)(dma_dst + slot * MLP_STAGE + 8 * NW * i * 128), 16, 0, 0);
    };

    // fragment read addresses: tile j of a stage = LDS rows 16 j + frow; hi slot fk, lo slot 4 + fk, XORed with (frow >> 1) & 7
    const int sw = (frow >> 1) & 7;
    const unsigned rd_hi = lds_addr(lds) + frow * 128 + ((fk ^ sw) << 4);
    const unsigned rd_lo = lds_addr(lds) + frow * 128 + (((4 + fk) ^ sw) << 4);
    const unsigned rd_hi2 = rd_hi + 4 * MLP_STAGE, rd_lo2 = rd_lo + 4 * MLP_STAGE;     // slots 4..7: the offset field holds 16 bits

    f32x4 hacc[RT][8], oacc[RT][16];
    s16x8 hh[RT][4], hl[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int j = 0; j < 16; ++j) oacc[rt][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one stage's products: acc[rt][J0 + j] += W(tile j) . b[rt]^T over its 32 k, j < 8; three bf16 MFMAs per product (x_lo W_hi,
    // x_hi W_lo, x_hi W_hi: the order of gemm_big_split_kernel), a tile at a time over its RT accumulators.
    // Weight fragments: a RING of four tile buffers (hi + lo, 32 registers) that runs through the stages - tile j + 4 is requested
    // behind the first MFMA of tile j + 1 (its buffer is the one tile j has just left) and used three tiles = 18 MFMAs later; with
    // two-tile double buffering (10 MFMAs ahead) a fragment wait still exposed LDS latency: the reads cost 24 of the kernel's 101 us.
    // The ring crosses stage boundaries: `next()` - wait for this wave's DMA pieces of the next stage, the workgroup barrier - sits
    // in front of tile 5, the first tile that requests a fragment of the next stage.  between(g) runs behind tile 2 g + 1.
    u32x4_m wt[4][2];                                                    // [tile & 3][hi, lo]
    u32x4_m dummy[4][2];                                                 // DIAG 6: the reads land here, nothing waits for them
    if constexpr (DIAG == 3 || DIAG == 6 || DIAG == 7) {                  // opaque (not undefined) operands
#pragma unroll
        for (int i = 0; i < 4; ++i) { asm volatile("" : "=v"(wt[i][0])); asm volatile("" : "=v"(wt[i][1])); asm volatile("" : "=v"(dummy[i][0])); asm volatile("" : "=v"(dummy[i][1])); }
    }
    auto fetch_tile = [&](auto slotc, auto jc) __attribute__((always_inline)) {
        constexpr int slot = decltype(slotc)::value, j = decltype(jc)::value;
        if constexpr (DIAG == 3) return;
        const unsigned ah = slot < 4 ? rd_hi : rd_hi2, al = slot < 4 ? rd_lo : rd_lo2;
        constexpr int so = (slot & 3) * MLP_STAGE;
        if constexpr (DIAG == 6) { fetch16_keep<so + j * 2048>(dummy[j & 3][0], ah); fetch16_keep<so + j * 2048>(dummy[j & 3][1], al); return; }
        fetch16<so + j * 2048>(wt[j & 3][0], ah); fetch16<so + j * 2048>(wt[j & 3][1], al);
    };
    auto stage_mma = [&](auto slotc, auto& acc, auto j0c, const s16x8 (&bh)[RT], const s16x8 (&bl)[RT], auto&& between, auto&& next) __attribute__((always_inline)) {
        constexpr int slot = decltype(slotc)::value, J0 = decltype(j0c)::value;
        static_for<8>([&](auto jc) {
            constexpr int j = decltype(jc)::value, cur = j & 3;
            next(jc);
            // outstanding behind tile j's two reads: tiles j + 1, j + 2 (tile j + 3 follows this tile's first MFMA)
            if constexpr (DIAG != 3 && DIAG != 6) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(wt[cur][0]), "+v"(wt[cur][1]));
            __builtin_amdgcn_sched_barrier(0);
            static_for<3 * RT>([&](auto mc) {
                constexpr int m = decltype(mc)::value, prod = m / RT, rt = m % RT;
                if constexpr (DIAG == 7) {                               // waits as in the real kernel, operands that no read wrote
                    acc[rt][J0 + j] = MFMA16_S16(as_bf(dummy[cur][prod == 1]), prod == 0 ? bl[rt] : bh[rt], acc[rt][J0 + j], 0, 0, 0);
                } else if constexpr (DIAG != 2) {
                    if constexpr (prod == 0) acc[rt][J0 + j] = MFMA16_S16(as_bf(wt[cur][0]), bl[rt], acc[rt][J0 + j], 0, 0, 0);
                    else if constexpr (prod == 1) acc[rt][J0 + j] = MFMA16_S16(as_bf(wt[cur][1]), bh[rt], acc[rt][J0 + j], 0, 0, 0);
                    else acc[rt][J0 + j] = MFMA16_S16(as_bf(wt[cur][0]), bh[rt], acc[rt][J0 + j], 0, 0, 0);
                }
                if constexpr (m == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (j + 3 < 8) fetch_tile(slotc, IntC<j + 3>{});
                    else fetch_tile(IntC<(slot + 1) % MLP_NS>{}, IntC<j + 3 - 8>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            if constexpr (DIAG == 2) acc[0][J0 + j][0] += __builtin_bit_cast(float, wt[cur][0][0]) + __builtin_bit_cast(float, wt[cur][1][1]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (j & 1) between(IntC<j / 2>{});
        });
    };

    // ---- the weight stream (round 6: exact waits).  `s_waitcnt vmcnt(N)`, N > 0, does NOT retire the older of several LDS-DMA batches
    // a wave has in flight (the requests complete out of issue order when their latencies differ: DESIGN 4b, profiles/r5/11_*), so a
    // wave has ONE batch outstanding and waits for it with vmcnt(0).  A batch = the wave's PPW pieces of each of the four stages of a
    // group; group G + 1 is requested while group G is multiplied, into the ring half group G - 1 has left:
    //   B1 (in front of tile 5 of a group's last stage, the first tile that reads a fragment of the next group): every wave waits
    //      vmcnt(0) - its whole batch of the next group, nothing younger in flight - then the workgroup barrier: the next group is in LDS.
    //      Every wave has also finished READING stages 0 .. 2 of the finishing group (the lgkmcnt waits of their tiles 7 lie behind it),
    //      so from here on the batch after next goes into those three slots;
    //   B2 (a plain barrier in front of tile 5 of a group's first stage): every wave has finished reading the previous group's LAST
    //      stage; its slot takes the fourth stage of the batch.
    // Issue schedule (H = PPW / 2 pieces behind each of the MFMA groups named; an LDS-DMA piece costs its wave 60 - 180 cycles of issue,
    // which should fall where the matrix pipe has queued work): stage 3 of a group, behind tiles 5, 7: stage 0 of the batch; stage 0,
    // behind tiles 1, 3, 5, 7: stages 1, 2; stage 1, behind tiles 1, 3: stage 3.  The batch is complete 2.5 stages (~1.9 us) before B1.
    // Two barriers per four stages instead of four.  Measured (profiles/r6/01_*, one box, decode of 128 x 196 frames): the rest of the
    // batch squeezed into the group's first stage (B2 in front of its tile 1: longer landing time) is SLOWER (2.08 against 2.02 ms
    // with <4, 2>) - what costs is issue time taken from the matrix pipe in one place, not landing time.
    static_for<MLP_GRP>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        static_for<PPW>([&](auto ic) { issue(DIAG == 1 ? -1 : 0, IntC<u>{}, IntC<u>{}, ic); });
    });
    __syncthreads();                                                     // linear1's bias is in LDS (plain stores: lgkmcnt, compiler-tracked)
    auto landed = [&]() __attribute__((always_inline)) {
        wait_vm<0>();                                                    // this wave's batch (and, the first time, its x rows); nothing younger in flight
        __builtin_amdgcn_s_barrier();
    };
    landed();
    static_for<PPW>([&](auto ic) { issue(DIAG == 1 ? -1 : 0, IntC<MLP_GRP>{}, IntC<MLP_GRP>{}, ic); });   // stage 0 of group 1 (in the loop: behind tiles 5, 7 of the previous group's last stage)
    static_for<3>([&](auto jc) { fetch_tile(IntC<0>{}, jc); });
    constexpr int H = PPW / 2;
#pragma unroll 1
    for (int hs = 0; hs < 8; ++hs) {
        static_for<16>([&](auto uc) {
            constexpr int u = decltype(uc)::value, slot = u % MLP_NS, r = u % MLP_GRP;
            // pieces [H * half, H * half + H) of the stage `ahead` further; past the last slice a harmless re-fetch (slice 0) keeps the loop uniform
            auto batch_pieces = [&](auto aheadc, auto halfc) __attribute__((always_inline)) {
                constexpr int ahead = decltype(aheadc)::value, half = decltype(halfc)::value;
                constexpr int ut = (u + ahead) % 16, slot_t = (u + ahead) % MLP_NS;
                const int hs_t = (hs + (u + ahead >= 16 ? 1 : 0)) & 7;
                static_for<H>([&](auto ic) { issue(hs_t, IntC<ut>{}, IntC<slot_t>{}, IntC<H * half + decltype(ic)::value>{}); });
            };
            auto between = [&](auto gc) __attribute__((always_inline)) {
                constexpr int g = decltype(gc)::value;
                if constexpr (r == 3 && g >= 2) batch_pieces(IntC<5>{}, IntC<g - 2>{});          // stage 0 of the batch: behind B1
                else if constexpr (r == 0 && g < 2) batch_pieces(IntC<5>{}, IntC<g>{});          // stage 1
                else if constexpr (r == 0 && g >= 2) batch_pieces(IntC<6>{}, IntC<g - 2>{});     // stage 2
                else if constexpr (r == 1 && g < 2) batch_pieces(IntC<6>{}, IntC<g>{});          // stage 3: its slot is free behind B2
            };
            auto next = [&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                if constexpr (r == 3 && j == 5) landed();                                                  // B1
                else if constexpr (r == 0 && j == 5) __builtin_amdgcn_s_barrier();                         // B2
            };
            if constexpr (u < 8) {
                if constexpr (u == 0) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int j = 0; j < 8; ++j) hacc[rt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                s16x8 bh[RT], bl[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) { bh[rt] = xh[rt][u]; bl[rt] = xl[rt][u]; }
                stage_mma(IntC<slot>{}, hacc, IntC<0>{}, bh, bl, between, next);
            } else {
                constexpr int v = u - 8, c = v >> 1, nh = v & 1;
                if constexpr (nh == 0) {
                    // hidden columns 32 c + 8 fk .. + 7 of the slice (tiles 2c, 2c+1): + bias, GELU, split -> operand fragment of k-step c
                    const f32x4 ba = ld4(b1s + 128 * hs + 32 * c + 8 * fk), bb = ld4(b1s + 128 * hs + 32 * c + 8 * fk + 4);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        f32x4 va, vb;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if constexpr (DIAG == 4) { va[e] = hacc[rt][2 * c][e] + ba[e]; vb[e] = hacc[rt][2 * c + 1][e] + bb[e]; }
                            else { va[e] = gelu_erf(hacc[rt][2 * c][e] + ba[e]); vb[e] = gelu_erf(hacc[rt][2 * c + 1][e] + bb[e]); }
                        }
                        split8(va, vb, hh[rt][c], hl[rt][c]);
                    }
                }
                s16x8 bh[RT], bl[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) { bh[rt] = hh[rt][c]; bl[rt] = hl[rt][c]; }
                stage_mma(IntC<slot>{}, oacc, IntC<8 * nh>{}, bh, bl, between, next);
            }
        });
    }
    wait_vm<0>();                                                        // no LDS-DMA may land after the workgroup has gone
    if constexpr (DIAG == 6) {
        wait_lgkm<0>(dummy[0][0], dummy[0][1], dummy[1][0], dummy[1][1]);
        wait_lgkm<0>(dummy[2][0], dummy[2][1], dummy[3][0], dummy[3][1]);
    } else if constexpr (DIAG != 3) {                                    // the fetches behind the last stage
        wait_lgkm<0>(wt[0][0], wt[0][1], wt[1][0], wt[1][1]);
        wait_lgkm<0>(wt[2][0], wt[2][1], wt[3][0], wt[3][1]);
    }

    if constexpr (DIAG == 5) {                                            // no epilogue traffic: one word per lane
        float s = 0.f;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) s += oacc[rt][jj][0] + oacc[rt][jj][1] + oacc[rt][jj][2] + oacc[rt][jj][3];
        if (live[0] && p.y != nullptr) p.y[(size_t)myrow[0] * D + fk] = s;
        return;
    }
    // ---- epilogue, per lane and row tile: row frow, columns col(jj) .. + 3 of accumulator jj:  + bias + residual, LayerNorm(s) in
    // the accumulator layout (an in-lane sum + two cross-lane steps per statistic); then the tile goes through LDS - the ring is free
    // now, 16 KiB per wave - so that every store instruction writes ONE whole row (1 KiB contiguous) instead of 16-byte pieces of
    // 16 rows (96 such stores per lane cost ~150 cycles of issue each).  LDS image of a tile: row r at 1 KiB r, 16-byte chunk ch at
    // ch ^ r (conflict-free for the accumulator-layout writes - 16 rows, one chunk - and the row reads alike).
    __syncthreads();                                                      // every wave is done with the ring
    char* const stg = lds + wave * 16384;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const size_t rbase = (size_t)myrow[rt] * D;
        f32x4 (&v)[16] = oacc[rt];
        float s = 0.f;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int col = 32 * (jj >> 1) + 8 * fk + 4 * (jj & 1);
            const f32x4 bv = ld4(prm + col), rv = ld4(p.x + rbase + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[jj][e] = v[jj][e] + bv[e] + rv[e]; }
            s += (v[jj][0] + v[jj][1]) + (v[jj][2] + v[jj][3]);
        }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1 && p.g4 == nullptr) break;
            const float* gp = prm + (pass == 0 ? 1 : 3) * D;
            const float* bp = prm + (pass == 0 ? 2 : 4) * D;
            if (pass == 1) {
                s = 0.f;
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) s += (v[jj][0] + v[jj][1]) + (v[jj][2] + v[jj][3]);
            }
            s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);      // the row's four lane groups
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
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int ch = 8 * (jj >> 1) + 2 * fk + (jj & 1);
            *reinterpret_cast<f32x4*>(stg + frow * 1024 + ((ch ^ frow) << 4)) = v[jj];
        }
        const int row0 = blockIdx.x * BM + RW * wave + 16 * rt;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(stg + r * 1024 + ((lane ^ r) << 4));
            if (row0 + r < M) {
                if (p.y != nullptr) st4(p.y + (size_t)(row0 + r) * D + 4 * lane, o);
                if (p.ys != nullptr) store_split4(p.ys + (size_t)(row0 + r) * D, 4 * lane, o);
            }
        }
    }
}

std::atomic<int> g_mlp_variant{0};   // measurement switch (ladiff_debug_set_mlp_variant)

// hipFuncSetAttribute is per DEVICE and must not land inside a stream capture: once per device, under a mutex; the graphed decode
// calls it before it begins its capture
int dec_mlp_prepare() {
    static std::mutex mu;
    static bool attr_set[64] = {};
    int dev = 0;
    LADIFF_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return LADIFF_ERR_ARG;
    std::lock_guard<std::mutex> lock(mu);
    if (!attr_set[dev]) {
        const void* k[3] = {reinterpret_cast<const void*>(dec_mlp_kernel<4, 2>), reinterpret_cast<const void*>(dec_mlp_kernel<8, 1>),
                            reinterpret_cast<const void*>(dec_mlp_kernel<4, 1>)};
        for (int i = 0; i < 3; ++i) LADIFF_HIP(hipFuncSetAttribute(k[i], hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS));
#ifdef LADIFF_STAMPS
        const void* d[7] = {reinterpret_cast<const void*>(dec_mlp_kernel<4, 2, 1>), reinterpret_cast<const void*>(dec_mlp_kernel<4, 2, 2>),
                            reinterpret_cast<const void*>(dec_mlp_kernel<4, 2, 3>), reinterpret_cast<const void*>(dec_mlp_kernel<4, 2, 4>),
                            reinterpret_cast<const void*>(dec_mlp_kernel<4, 2, 5>), reinterpret_cast<const void*>(dec_mlp_kernel<4, 2, 6>),
                            reinterpret_cast<const void*>(dec_mlp_kernel<4, 2, 7>)};
        for (int i = 0; i < 7; ++i) LADIFF_HIP(hipFuncSetAttribute(d[i], hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS));
#endif
        attr_set[dev] = true;
    }
    return 0;
}
// Rows from which the fused kernel beats linear1 + linear2 + LayerNorm as three launches: a workgroup streams all 2 MB of weights
// whatever its share of the rows, ~60 us even alone on the chip, while the three launches scale down with the rows (6272 rows: 56 us
// against 64, 12544 rows: 78 against 71, profiles/r3)
int dec_mlp_min_rows() { return 10000; }

// y / ys [M,256] = LN3(x + lin2(gelu(lin1(x)))) (then LN4 when g4 != NULL); xs = S-format twin of x, w1 / w2 S-format
int launch_dec_mlp(const float* xs, const float* x, const float* w1, const float* b1, const float* w2, const float* b2, const float* g3,
                   const float* be3, const float* g4, const float* be4, float* y, float* ys, int M, hipStream_t s) {
    if (M <= 0) return 0;
    LADIFF_TRY(dec_mlp_prepare());
    MlpArgs a{xs, x, w1, b1, w2, b2, g3, be3, g4, be4, y, ys, M};
    // every workgroup streams all 2 MB of weights: 128-row workgroups when they fill the chip, 64-row ones (twice as many) otherwise
    // (measured, profiles/r3: 25088 rows 111 us <4,2> / 116 <8,1> / 137 <4,1>; 12544 rows 106 / 108 / 71; the caller keeps the
    // three-launch form below dec_mlp_min_rows())
    int form = g_mlp_variant;                      // 0: by size, 1: <8, 1>, 2: <4, 1>, 3: <4, 2>
    if (form >= 21) form = 0;                      // 21 .. 23: timing builds of the attention kernel (dec_qkv_attn.hip)
    // round 6: with one batch per wave behind vmcnt(0) the two-waves-per-SIMD form is the fastest at 25088 rows (decode 1.92 ms against
    // 2.02 with <4, 2>; round 5's counted-wait ring: 2.00 / 1.97) - a wave's LDS-DMA issue falls under its SIMD partner's MFMAs
    if (form == 0) form = (M + 127) / 128 < 160 ? 2 : 1;
#ifdef LADIFF_STAMPS
    // diagnostic twin only (the product library has no such instantiation and rejects the values): timing builds with garbage
    // results - <4, 2> without DMA / MFMAs / fragment reads / GELU / epilogue
    if (form >= 11 && form <= 17) {
        const void* k[7] = {reinterpret_cast<const void*>(dec_mlp_kernel<4, 2, 1>), reinterpret_cast<const void*>(dec_mlp_kernel<4, 2, 2>),
                            reinterpret_cast<const void*>(dec_mlp_kernel<4, 2, 3>), reinterpret_cast<const void*>(dec_mlp_kernel<4, 2, 4>),
                            reinterpret_cast<const void*>(dec_mlp_kernel<4, 2, 5>), reinterpret_cast<const void*>(dec_mlp_kernel<4, 2, 6>),
                            reinterpret_cast<const void*>(dec_mlp_kernel<4, 2, 7>)};
        void* args[] = {&a};
        LADIFF_HIP(hipLaunchKernel(k[form - 11], dim3((M + 127) / 128), dim3(256), args, MLP_LDS, s));
    } else
#endif
    if (form == 3) hipLaunchKernelGGL((dec_mlp_kernel<4, 2>), dim3((M + 127) / 128), dim3(256), MLP_LDS, s, a);
    else if (form == 1) hipLaunchKernelGGL((dec_mlp_kernel<8, 1>), dim3((M + 127) / 128), dim3(512), MLP_LDS, s, a);
    else hipLaunchKernelGGL((dec_mlp_kernel<4, 1>), dim3((M + 63) / 64), dim3(256), MLP_LDS, s, a);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

}  // namespace ladiff
