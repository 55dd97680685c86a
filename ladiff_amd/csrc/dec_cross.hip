// Decoder cross-attention block as a per-sample low-rank map (TransformerDecoderLayer.forward_post, cross_attention.py:373-376
// + norm2, :409): tgt = LN2(tgt + out_proj(softmax(q k^T / 8) v)) with q = in_proj_q(tgt) and only T <= 8 memory tokens.
//
// With so few keys the two 256x256 projections around the attention are better folded INTO the keys and values:
//   score[h,j] = (x Wq_h^T + bq_h) . k[b,j,h] / 8 = x . G[b,h,j] + c[b,h,j],      G[b,h,j] = Wq_h^T k[b,j,h] / 8   (256-vector)
//   out        = sum_h sum_j p[h,j] (Wo[:, h] v[b,j,h]) + bo = sum_{h,j} p[h,j] U[b,h,j] + bo,   U[b,h,j] = Wo[:, h] v[b,j,h]
// - exact algebra (a re-association of the reference's sums), fp32 throughout.  Per frame row that is 4 T dot products and
// 4 T axpys of length 256 (20 kFLOP at T = 5) instead of two 256x256 projections (262 kFLOP), and the query / attention
// output tensors ([B F, 256] each, written and re-read) never exist.  Before: q GEMM + cross-attention kernel + out GEMM +
// LayerNorm kernel, 102 us per layer at B = 128, F = 196 (profiles/r2/03); now prep (per sample, 10 us) + apply.
#include <mutex>
#include "kernels.h"
#include "tile_mma.h"

namespace ladiff { extern unsigned long long* g_sys_stamps; }   // systolic.hip (diagnostic twin build)

namespace ladiff {

namespace {
constexpr int TM = LADIFF_MAX_LATENTS;
}

// G | U | c of a (layer, head) are three small matrix products over the memory rows (bt = j B + b: kv is [T][B][512]) with K = 64:
//   G^T [bt, n] = K_h [bt, d] Wq_h [d, n] / 8,     U^T [bt, n] = V_h [bt, d] Wo_h [n, d],     c [bt] = K_h [bt, d] bq_h [d] / 8
// on the fp32 MFMA (`v_mfma_f32_16x16x4_f32`: exact fp32 products, fp32 accumulate).  One workgroup per (group of 16-row tiles of bt,
// head, layer); wave w owns output columns 64 w .. 64 w + 63 and keeps its slice of Wq_h and Wo_h (64 + 64 values per lane) as B
// fragments in REGISTERS for all its tiles; a tile's K_h / V_h rows are A fragments loaded straight from memory.  The contraction
// index d is dealt to the MFMA's (k-step s, k-phase kq) as d = 16 kq + s - any bijection does, A and B share it - so a lane's 16
// values per row are 64 contiguous bytes.  Before (rounds 3 - 4): one thread per output column with the K | V values as scalar
// loads into SGPRs, bound by their latency: 68 us per decode at 128 x 5 memory rows; this form 33 - 40 us (rocprofv3), 8 x 2 rows: the
// decode of config c1 0.556 -> 0.50 ms.
__global__ __launch_bounds__(256) void dec_cross_prep_kernel(const DecCrossPrepBatch pb, int B, int T, int tiles_per_wg) {
    const int h = blockIdx.y, layer = blockIdx.z, tid = threadIdx.x, lane = tid & 63, c16 = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* __restrict__ kv = pb.kv[layer]; const float* __restrict__ wq = pb.wq[layer]; const float* __restrict__ bq = pb.bq[layer];
    const float* __restrict__ wo = pb.wo[layer]; float* __restrict__ gu = pb.gu[layer];
    float* __restrict__ cc = gu + (size_t)B * H * T * 2 * D;
    const int M = T * B, ntiles = (M + 15) / 16;
    float gq[4][16], uo[4][16], bqv[16];                                           // [column tile][k-step]: Wq[h 64 + 16 kq + s][n], Wo[n][h 64 + 16 kq + s]
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int n = 64 * wave + 16 * t + c16;
#pragma unroll
        for (int s = 0; s < 16; ++s) gq[t][s] = wq[(size_t)(h * DH + 16 * kq + s) * D + n];
#pragma unroll
        for (int s4 = 0; s4 < 16; s4 += 4) {
            const f32x4 w4 = ld4(wo + (size_t)n * D + h * DH + 16 * kq + s4);
#pragma unroll
            for (int e = 0; e < 4; ++e) uo[t][s4 + e] = w4[e];
        }
    }
#pragma unroll
    for (int s4 = 0; s4 < 16; s4 += 4) {
        const f32x4 b4 = ld4(bq + h * DH + 16 * kq + s4);
#pragma unroll
        for (int e = 0; e < 4; ++e) bqv[s4 + e] = b4[e];
    }
    const int tile1 = (blockIdx.x + 1) * tiles_per_wg < ntiles ? (blockIdx.x + 1) * tiles_per_wg : ntiles;
    for (int tile = blockIdx.x * tiles_per_wg; tile < tile1; ++tile) {
        const int arow = 16 * tile + c16 < M ? 16 * tile + c16 : M - 1;            // this lane's A row (clamped: rows >= M are not stored)
        const float* kr = kv + (size_t)arow * 2 * D + h * DH + 16 * kq;
        float ak[16], av[16];
#pragma unroll
        for (int s4 = 0; s4 < 16; s4 += 4) {
            const f32x4 k4 = ld4(kr + s4), v4 = ld4(kr + D + s4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { ak[s4 + e] = k4[e]; av[s4 + e] = v4[e]; }
        }
        f32x4 ag[4], au[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { ag[t] = f32x4{0.f, 0.f, 0.f, 0.f}; au[t] = ag[t]; }
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                ag[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ak[s], gq[t][s], ag[t], 0, 0, 0);
                au[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], uo[t][s], au[t], 0, 0, 0);
            }
        // accumulator of lane (c16, kq): column 64 wave + 16 t + c16 of rows 16 tile + 4 kq + i
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int bt = 16 * tile + 4 * kq + i;
            if (bt < M) {
                const int j = bt / B, b = bt - j * B;
                float* o = gu + ((((size_t)b * H + h) * T + j) * 2) * D + 64 * wave + c16;
#pragma unroll
                for (int t = 0; t < 4; ++t) { o[16 * t] = ag[t][i] * 0.125f; o[D + 16 * t] = au[t][i]; }      // 1 / sqrt(64), exact
            }
        }
        // c[h][j] = bq_h . k[b,j,h] / 8 (wave 0: every wave holds the same K fragments)
        if (wave == 0) {
            float c = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) c = fmaf(bqv[s], ak[s], c);
            c += __shfl_xor(c, 16, 64); c += __shfl_xor(c, 32, 64);                 // the row's four k-phases
            const int bt = 16 * tile + c16;
            if (kq == 0 && bt < M) { const int j = bt / B, b = bt - j * B; cc[((size_t)b * H + h) * T + j] = c * 0.125f; }
        }
    }
}

// the folded attention of ONE frame row held on 16 lanes (lane l16: columns 4 l16 + 64 k): acc += sum_{h,j} softmax_j(x . G[h,j] + c[h,j]) U[h,j]
// sm: the sample's G | U [H][T][2][256] in LDS, cs: its score offsets [H][T]; tokens >= nv are masked (cross_attention.py:408-409)
template <int T>
__device__ __forceinline__ void cross_row(const f32x4 (&xv)[4], f32x4 (&acc)[4], const float* sm, const float* cs, int nv, int c0) {
#pragma unroll 1                                  // one head at a time: unrolled over the heads the 8 T LDS reads per head pile up (spills)
    for (int h = 0; h < H; ++h) {
        float sc[T];
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const float* gp = sm + ((size_t)(h * T + j) * 2) * D + c0;
            float d = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 g4 = ld4(gp + 64 * k);
                d = fmaf(xv[k][0], g4[0], fmaf(xv[k][1], g4[1], fmaf(xv[k][2], g4[2], fmaf(xv[k][3], g4[3], d))));
            }
            sc[j] = d;
        }
#pragma unroll
        for (int j = 0; j < T; ++j) sc[j] = row16_sum(sc[j]) + cs[h * T + j];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < T; ++j) { sc[j] = j < nv ? sc[j] : -INFINITY; m = fmaxf(m, sc[j]); }
        float l = 0.f;
#pragma unroll
        for (int j = 0; j < T; ++j) { sc[j] = __builtin_amdgcn_exp2f((sc[j] - m) * 1.4426950408889634f); l += sc[j]; }
        const float inv = 1.f / l;
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const float pj = sc[j] * inv;
            const float* up = sm + ((size_t)(h * T + j) * 2 + 1) * D + c0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 u4 = ld4(up + 64 * k);
                acc[k][0] = fmaf(pj, u4[0], acc[k][0]); acc[k][1] = fmaf(pj, u4[1], acc[k][1]);
                acc[k][2] = fmaf(pj, u4[2], acc[k][2]); acc[k][3] = fmaf(pj, u4[3], acc[k][3]);
            }
        }
    }
}

// grid (B, chunks): the sample's G | U in LDS, 16 frame rows per workgroup pass; TT = number of memory tokens (compile time, so
// the TT score reductions of a head are independent chains the scheduler can interleave)
template <int TT>
__global__ __launch_bounds__(256) void dec_cross_apply_kernel(const float* __restrict__ x, const float* __restrict__ gu,
                                                              const float* __restrict__ cc, const int32_t* __restrict__ counts,
                                                              const float* __restrict__ bo, const float* __restrict__ g2,
                                                              const float* __restrict__ b2, int F, int rows_per_wg,
                                                              float* __restrict__ y, float* __restrict__ ys,
                                                              const int32_t* __restrict__ row_off, const float* __restrict__ g1,
                                                              const float* __restrict__ b1) {
    extern __shared__ __attribute__((aligned(16))) float sm[];                    // [H][TT][2][256], then c [H][TT]
    constexpr int T = TT, HT = H * TT;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* src = gu + (size_t)b * HT * 2 * D;
    for (int u = tid; u < HT * 2 * D / 4; u += 256) st4(sm + 4 * u, ld4(src + 4 * u));
    float* cs = sm + (size_t)HT * 2 * D;
    if (tid < HT) cs[tid] = cc[(size_t)b * HT + tid];
    int nv = counts != nullptr ? counts[b] : T;
    nv = nv > T ? T : nv;
    __syncthreads();
    // a frame row on 16 lanes (lane l: columns 4 l + 64 k, k < 4), four rows per wave: the 4 T score reductions of a row are
    // four DPP steps each (one wave per row needed six steps + a readlane per reduction, and was bound by them)
    const int l16 = lane & 15, grp = lane >> 4, c0 = 4 * l16;
    f32x4 bo4[4], gg[4], bb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { bo4[k] = ld4(bo + c0 + 64 * k); gg[k] = ld4(g2 + c0 + 64 * k); bb[k] = ld4(b2 + c0 + 64 * k); }
    size_t row0 = (size_t)b * F;
    if (row_off != nullptr) { row0 = row_off[b]; F = row_off[b + 1] - row_off[b]; }     // ragged rows: only the sample's own frames
    const int f0 = blockIdx.y * rows_per_wg;
    const int f1 = f0 + rows_per_wg < F ? f0 + rows_per_wg : F;
    for (int fb = f0; fb < f1; fb += 16) {
        const int f = fb + 4 * wave + grp;
        const bool live = f < f1;
        const size_t row = row0 + (live ? f : f1 - 1);
        f32x4 xv[4], acc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) xv[k] = ld4(x + row * D + c0 + 64 * k);
        if (g1 != nullptr) {                                                                // the row arrives before norm1: apply it here (two-pass)
            float s1 = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) s1 += (xv[k][0] + xv[k][1]) + (xv[k][2] + xv[k][3]);
            const float mean1 = row16_sum(s1) * (1.f / 256.f);
            float q1 = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float dlt = xv[k][i] - mean1; q1 += dlt * dlt; }
            const float rstd1 = rsqrtf(row16_sum(q1) * (1.f / 256.f) + LN_EPS);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 ga = ld4(g1 + c0 + 64 * k), be = ld4(b1 + c0 + 64 * k);
#pragma unroll
                for (int i = 0; i < 4; ++i) xv[k][i] = (xv[k][i] - mean1) * rstd1 * ga[i] + be[i];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[k][i] = bo4[k][i] + xv[k][i];                   // out_proj bias + residual
        cross_row<T>(xv, acc, sm, cs, nv, c0);
        // norm2 over the row (two-pass, as rowops.hip)
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) sum += (acc[k][0] + acc[k][1]) + (acc[k][2] + acc[k][3]);
        const float mean = row16_sum(sum) * (1.f / 256.f);
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float dlt = acc[k][i] - mean; q += dlt * dlt; }
        const float rstd = rsqrtf(row16_sum(q) * (1.f / 256.f) + LN_EPS);
        if (live) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (acc[k][i] - mean) * rstd * gg[k][i] + bb[k][i];
                st4(y + row * D + c0 + 64 * k, o);
                if (ys != nullptr) store_split4(ys + row * D, c0 + 64 * k, o);
            }
        }
    }
}


// ---------------------------------------------------------------- self-attention out_proj + norm1 + the block above, one kernel (f16x3 mode)
//   y = LN2( x1n + cross(x1n) ),  x1n = LN1( x0 + att Wo^T + bo )        cross_attention.py:369-376, :407-409
// Before: a 128x128-tile GEMM that writes x0 + out_proj(att) (23 us at 25088 rows, 20 % MFMA use, ~80 MB moved) and the row kernel
// above that reads it back (29 us, no MFMA, bound by the LDS: every frame row reads all 8 T KiB of its sample's G | U): 52 us per layer.
// Here a workgroup of eight waves keeps EVERY operand that does not change in its REGISTERS for all its rows, as the stages of the
// pipeline kernel do (tile_mma.h): Wo (256 KB in S-format; wave w: output columns 32 w .. 32 w + 31) and the sample's folded keys /
// values as f16x3 fragments (G: k-step w of the 32 padded (head, token) columns; U: this wave's 32 columns).  It walks its sample's
// frame rows 32 at a time, three small matrix products per pass:
//   1. att rows (S-format, copied 16 bytes at a time into the swizzled operand tile - the global row layout IS the tile's block
//      layout) x Wo -> staging tile -> + bias + residual, norm1 on 16 lanes per row -> x1n kept (fp32, LDS) and written as operand tile
//   2. scores [32 x 32] = x1n x G^T, split over the waves by k-step (12 MFMAs each), the eight partial tiles summed in wave order;
//      + c, mask, softmax per head -> P as operand tile (K = 32)
//   3. P x U [32 x 256] (12 MFMAs per wave) -> staging tile -> + bias + residual, norm2 -> y, ys
// So the cross-attention's 20 dot products and 20 axpys of length 256 per row run on the MFMA (as hi*hi + hi*lo + lo*hi, like every
// other product of f16x3 mode) instead of out of the LDS, and x0 + out_proj(att) never exists in memory.
constexpr int OC_ROWS = 32;
constexpr int OC_TILE = OC_ROWS * 1024;                                  // operand tile: att rows, then x1n rows
constexpr int OC_CT = OC_ROWS * CLD * 4;                                 // fp32 staging tile (products 1 and 3)
constexpr int OC_XN = OC_ROWS * CLD * 4;                                 // x1n rows, fp32 (the cross-attention's residual)
constexpr int OC_SLD = 36;                                               // row stride of a score tile (floats)
constexpr int OC_PS = 8 * OC_ROWS * OC_SLD * 4;                          // the eight k-step partial score tiles
constexpr int OC_PT = OC_ROWS * 256;                                     // P operand tile (one 64-column block, 32 used)

struct OutCrossArgs {
    const float* att;          // [M,256] S-format: the self-attention output (heads concatenated)
    const float* x0;           // [M,256] fp32: the layer input (residual)
    const float* wo;           // self_attn.out_proj.weight [256,256] S-format
    const float* bo; const float* g1; const float* b1;                    // out_proj bias, norm1
    const float* gu; const float* cc;                                    // the layer's G | U and score offsets (dec_cross_prep_kernel)
    const float* bo_c; const float* g2; const float* b2;                  // cross-attention out_proj bias, norm2
    const int32_t* counts; const int32_t* row_off;
    float* y; float* ys;
    int F, rows_per_wg;
    unsigned long long* stamps;    // diagnostic twin build only
};

#ifdef LADIFF_STAMPS
#define OC_STAMP(i) do { if (p.stamps != nullptr && blockIdx.x == 3 && blockIdx.y == 0 && threadIdx.x == 0 && npass < 6) p.stamps[npass * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define OC_STAMP(i) do { } while (0)
#endif

// workgroup barrier without __syncthreads' release fence: with an LDS-DMA in flight that fence waits for it (vmcnt(0)); the waits for
// the DMA'd rows are placed by hand where the rows are needed
__device__ __forceinline__ void oc_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int TT>
__global__ __launch_bounds__(512, 1) void dec_out_cross_kernel(const OutCrossArgs p) {
    // Separate LDS objects, not one carved buffer: the compiler orders every LDS access that MAY alias an LDS-DMA in flight behind that
    // DMA (s_waitcnt vmcnt(0)); with distinct objects the softmax that follows the request for the next att tile, and the product that
    // follows the request for the next residual rows, do not wait for them.
    __shared__ __attribute__((aligned(1024))) char tile[OC_TILE];          // DMA target: att rows; then the x1n operand tile
    __shared__ __attribute__((aligned(1024))) char ps_b[OC_PS];            // DMA target: residual rows x0; then the partial score tiles
    __shared__ __attribute__((aligned(1024))) char ct_b[OC_CT];
    __shared__ __attribute__((aligned(1024))) char xn_b[OC_XN];
    __shared__ __attribute__((aligned(16))) float sc[OC_ROWS * OC_SLD];
    __shared__ __attribute__((aligned(256))) char ptile[OC_PT];
    __shared__ __attribute__((aligned(16))) float vec[6 * D];
    __shared__ float cs[32];
    constexpr int T = TT, HT = H * TT;
    static_assert(HT <= 32, "the (head, token) columns fit one k-step");
    float* ct = reinterpret_cast<float*>(ct_b);
    float* xn = reinterpret_cast<float*>(xn_b);
    float* ps = reinterpret_cast<float*>(ps_b);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, frow = lane & 15, fk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);            // wave-uniform: row and address arithmetic in scalar registers
    int npass = 0;
    OC_STAMP(15);
    // ---- what stays in registers for all rows
    WFrag<0, 2, 1> gf, uf;                                                 // G: columns (h, t) 16 j + frow, k-step `wave`; U: output columns 32 wave + 16 j + frow, k = (h, t)
    {
        const float* gub = p.gu + (size_t)b * HT * 2 * D;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int hj = 16 * j + frow;
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = a;
            if (hj < HT) { const float* g = gub + (size_t)hj * 2 * D + 32 * wave + 8 * fk; a = ld4(g); c = ld4(g + 4); }
            split8(a, c, gf.hi[j][0], gf.lo[j][0]);
            const int col = 32 * wave + 16 * j + frow;
            f32x4 u0 = {0.f, 0.f, 0.f, 0.f}, u1 = u0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (8 * fk + e < HT) u0[e] = gub[((size_t)(8 * fk + e) * 2 + 1) * D + col];
                if (8 * fk + 4 + e < HT) u1[e] = gub[((size_t)(8 * fk + 4 + e) * 2 + 1) * D + col];
            }
            split8(u0, u1, uf.hi[j][0], uf.lo[j][0]);
        }
        if (tid < 32) cs[tid] = tid < HT ? p.cc[(size_t)b * HT + tid] : 0.f;
        const float* vs[6] = {p.bo, p.g1, p.b1, p.bo_c, p.g2, p.b2};
        if (tid < 6 * (D / 4)) st4(vec + 4 * tid, ld4(vs[tid / (D / 4)] + 4 * (tid % (D / 4))));
        // the P tile's k columns 32 .. 63 are never read; its columns HT .. 31 are written as zeros every pass
    }
    WFrag<0, 2, 8> wf;                                                     // Wo rows 32 wave .. + 31, all of K
    {
        // Through the LDS, half of Wo (128 rows = 128 KB) at a time: read from memory as whole 1-KiB rows (a fragment load straight
        // from global memory touches 16 rows x 64 bytes per instruction, and 256 workgroups do it at once), laid out as operand tiles
        // (a_slot: conflict-free 16-byte fragment reads) in the four big buffers - 32 rows each, one per wave of the half that owns them
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            if (half == 1) __syncthreads();                               // the first half has been picked up
            // LDS-DMA, no registers: one instruction = one 1-KiB row (the hardware adds lane x 16 to the LDS address); the lane that
            // lands in physical slot (lane & 15) of block (lane >> 4) fetches the piece whose home that is under the tile's swizzle
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int q = (qq + b) & 3;                                 // buffer order rotated per workgroup: 256 of them read the same 256 KB at once
                char* dst = q == 0 ? tile : q == 1 ? ps_b : q == 2 ? ct_b : xn_b;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int lr32 = wave + 8 * kk, r = 32 * q + lr32;
                    const char* src = reinterpret_cast<const char*>(p.wo) + ((size_t)(128 * half + r) << 10) + ((lane & 48) << 4) + ((((lane & 15) ^ r) & 15) << 4);
                    __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src), (__attribute__((address_space(3))) void*)(dst + lr32 * 1024), 16, 0, 0);
                }
            }
            __builtin_amdgcn_s_waitcnt(0xF70);
            __syncthreads();
            if ((wave >> 2) == half) {
                const int q = wave & 3;
                const char* srcb = q == 0 ? tile : q == 1 ? ps_b : q == 2 ? ct_b : xn_b;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int st = 0; st < 8; ++st) {
                        const int r = 16 * j + frow;
                        wf.hi[j][st] = *reinterpret_cast<const s16x8*>(a_slot<4>(const_cast<char*>(srcb), r, st >> 1, 4 * (st & 1) + fk));
                        wf.lo[j][st] = *reinterpret_cast<const s16x8*>(a_slot<4>(const_cast<char*>(srcb), r, st >> 1, 8 + 4 * (st & 1) + fk));
                    }
            }
        }
        __syncthreads();                                                   // the staging buffers become the tiles
    }
    int nv = p.counts != nullptr ? p.counts[b] : T;
    nv = nv > T ? T : nv;
    int F = p.F;
    size_t row0 = (size_t)b * F;
    if (p.row_off != nullptr) { row0 = p.row_off[b]; F = p.row_off[b + 1] - p.row_off[b]; }     // ragged rows: only the sample's own frames
    const int f0 = blockIdx.y * p.rows_per_wg;
    const int f1 = f0 + p.rows_per_wg < F ? f0 + p.rows_per_wg : F;
    if (f0 >= f1) return;                                                  // uniform over the workgroup: no barrier is skipped by a part of it
    const int l16 = lane & 15, grp = lane >> 4, c0 = 4 * l16, lr = 4 * wave + grp;   // this 16-lane group's row of a pass
    typedef unsigned u32x4_o __attribute__((ext_vector_type(4)));
    // A pass's inputs arrive by LDS-DMA (no registers, nothing for the compiler to spill or to wait on early): the S-format att rows
    // straight into the swizzled operand tile (wave w: rows w + 8 i; see the prologue for the lane -> piece map), the fp32 residual
    // rows x0 into the area of the partial score tiles while that is idle.
    const float* x0s = ps;                                                 // [32][256] fp32 while it holds the residual rows
    auto dma_att = [&](int fb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = wave + 8 * i, f = fb + r < f1 ? fb + r : f1 - 1;
            unsigned vo = ((lane & 48) << 4) + ((((lane & 15) ^ r) & 15) << 4);
            asm volatile("" : "+v"(vo));           // opaque: keeps `base + lane part` from being hoisted out of the pass loop as 64-bit register pairs (spills)
            const char* src = reinterpret_cast<const char*>(p.att) + ((row0 + f) << 10) + vo;
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src), (__attribute__((address_space(3))) void*)(tile + r * 1024), 16, 0, 0);
        }
    };
    auto dma_x0 = [&](int fb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = wave + 8 * i, f = fb + r < f1 ? fb + r : f1 - 1;
            unsigned vo = lane << 4;
            asm volatile("" : "+v"(vo));
            const char* src = reinterpret_cast<const char*>(p.x0) + ((row0 + f) << 10) + vo;
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src), (__attribute__((address_space(3))) void*)(ps_b + r * 1024), 16, 0, 0);
        }
    };
    dma_att(f0);
    dma_x0(f0);
    __builtin_amdgcn_s_waitcnt(0xF70);
    __syncthreads();                                                       // the first pass's inputs are there, the vectors too
    for (int fb = f0; fb < f1; fb += OC_ROWS, ++npass) {
        OC_STAMP(0);
        OC_STAMP(1);
        f32x4 acc2[2][2];
        zero_acc(acc2);
        mma<0, 4, 2, 8, 2>(tile, wf, acc2);
        OC_STAMP(2);
        const int f = fb + lr;
        const bool live = f < f1;
        const size_t row = row0 + (live ? f : f1 - 1);
        f32x4 xv[4];
        stage_c(ct, acc2, [&](int j) { return 32 * wave + 16 * j; });
        __builtin_amdgcn_s_waitcnt(0xF70);                   // this wave's x0 pieces (requested a pass ago) have landed
        __syncthreads();                                                   // (2) staging tile and residual rows whole; every wave is done with the att tile
        OC_STAMP(3);
        {
            const float* cr = ct + lr * CLD + c0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 o = ld4(cr + 64 * k), bo = ld4(vec + c0 + 64 * k), r0 = ld4(x0s + lr * D + c0 + 64 * k);
#pragma unroll
                for (int i = 0; i < 4; ++i) xv[k][i] = (o[i] + bo[i]) + r0[i];              // out_proj + bias, + residual (the GEMM's epilogue order)
            }
            float s1 = 0.f;                                                                 // norm1 (two-pass)
#pragma unroll
            for (int k = 0; k < 4; ++k) s1 += (xv[k][0] + xv[k][1]) + (xv[k][2] + xv[k][3]);
            const float mean1 = row16_sum(s1) * (1.f / 256.f);
            float q1 = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float dlt = xv[k][i] - mean1; q1 += dlt * dlt; }
            const float rstd1 = rsqrtf(row16_sum(q1) * (1.f / 256.f) + LN_EPS);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 ga = ld4(vec + D + c0 + 64 * k), be = ld4(vec + 2 * D + c0 + 64 * k);
#pragma unroll
                for (int i = 0; i < 4; ++i) xv[k][i] = (xv[k][i] - mean1) * rstd1 * ga[i] + be[i];
                st4(xn + lr * CLD + c0 + 64 * k, xv[k]);                                    // x1n: the residual of the block, read back at the end
                tile_put4<0, 4>(tile, lr, c0 + 64 * k, xv[k]);                              // and the operand of the score product
            }
        }
        OC_STAMP(4);
        __syncthreads();                                                   // (3) x1n tile whole
        OC_STAMP(5);
        {   // scores: this wave's k-step of x1n . G^T, both row tiles x both column tiles
            u32x4_o ah[2], al[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const u32x4_o*>(a_slot<4>(tile, 16 * i + frow, wave >> 1, 4 * (wave & 1) + fk));
                al[i] = *reinterpret_cast<const u32x4_o*>(a_slot<4>(tile, 16 * i + frow, wave >> 1, 8 + 4 * (wave & 1) + fk));
            }
            f32x4 as[2][2];
            zero_acc(as);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) as[i][j] = MFMA16_S16(__builtin_bit_cast(s16x8, al[i]), gf.hi[j][0], as[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) as[i][j] = MFMA16_S16(__builtin_bit_cast(s16x8, ah[i]), gf.lo[j][0], as[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) as[i][j] = MFMA16_S16(__builtin_bit_cast(s16x8, ah[i]), gf.hi[j][0], as[i][j], 0, 0, 0);
            float* pw = ps + wave * (OC_ROWS * OC_SLD);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pw[(16 * i + 4 * fk + r) * OC_SLD + 16 * j + frow] = as[i][j][r];
        }
        __syncthreads();                                                   // (4) the eight partial score tiles are there
        OC_STAMP(6);
        // every wave has read its x1n fragments: the operand tile takes the next pass's att rows now
        if (fb + OC_ROWS < f1) dma_att(fb + OC_ROWS);
        {   // row lr on 16 lanes: lane l16 owns (head, token) columns l16 and 16 + l16
            float tot[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int hj = 16 * u + l16;
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) a += ps[w * (OC_ROWS * OC_SLD) + lr * OC_SLD + hj];      // wave order: fixed
                tot[u] = a + cs[hj];
                sc[lr * OC_SLD + hj] = tot[u];
            }
            // the other lanes of the row read these totals: same wave, and a wave's LDS operations complete in order.  No fence: a fence
            // also waits for the vector-memory counter, i.e. for the DMA of the next pass that has just been issued
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            float hs[2][T];                                                                 // the scores of this lane's two heads: 2 T reads in flight together
            int hh[2], jj[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int hj = 16 * u + l16;
                hh[u] = hj < HT ? hj / T : 0; jj[u] = hj - hh[u] * T;
#pragma unroll
                for (int t = 0; t < T; ++t) hs[u][t] = sc[lr * OC_SLD + hh[u] * T + t];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int hj = 16 * u + l16;
                float m = -INFINITY, l = 0.f;
#pragma unroll
                for (int t = 0; t < T; ++t) m = t < nv ? fmaxf(m, hs[u][t]) : m;
#pragma unroll
                for (int t = 0; t < T; ++t) l += t < nv ? __builtin_amdgcn_exp2f((hs[u][t] - m) * 1.4426950408889634f) : 0.f;
                float pr = __builtin_amdgcn_exp2f((tot[u] - m) * 1.4426950408889634f) / l;
                if (!(hj < HT && jj[u] < nv)) pr = 0.f;                                      // tokens >= count masked (:408-409); padded columns
                tile_put1<0, 1>(ptile, lr, hj, pr);
            }
        }
        oc_barrier();                                                      // (5) P tile whole
        OC_STAMP(7);
        f32x4 ao[2][2];
        zero_acc(ao);
        mma<0, 1, 2, 1, 2>(ptile, uf, ao);
        stage_c(ct, ao, [&](int j) { return 32 * wave + 16 * j; });
        // this wave's four att pieces of the next pass have landed (nothing else is in flight: the stores come later).  A builtin, so that
        // the compiler's own bookkeeping of LDS-DMA in flight sees the wait and puts none of its own in front of the LDS reads below
        __builtin_amdgcn_s_waitcnt(0xF70);
        oc_barrier();                                                      // (6) attention output staged; the next pass's att tile whole
        OC_STAMP(8);
        {
            const float* cr = ct + lr * CLD + c0;
            f32x4 acc[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 o = ld4(cr + 64 * k), bc = ld4(vec + 3 * D + c0 + 64 * k), x1 = ld4(xn + lr * CLD + c0 + 64 * k);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[k][i] = (bc[i] + x1[i]) + o[i];             // cross out_proj bias + residual + attention
            }
            // The next pass's residual rows go into the area of the partial score tiles (summed long ago).  Requested HERE, behind this
            // phase's LDS reads: the compiler makes every LDS read that follows an LDS-DMA it cannot tell apart wait for it.  They have
            // norm2, the stores and the next pass's first product to land.
            if (fb + OC_ROWS < f1) dma_x0(fb + OC_ROWS);
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) sum += (acc[k][0] + acc[k][1]) + (acc[k][2] + acc[k][3]);
            const float mean = row16_sum(sum) * (1.f / 256.f);
            float q = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float dlt = acc[k][i] - mean; q += dlt * dlt; }
            const float rstd = rsqrtf(row16_sum(q) * (1.f / 256.f) + LN_EPS);
            if (live) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 ga = ld4(vec + 4 * D + c0 + 64 * k), be = ld4(vec + 5 * D + c0 + 64 * k);
                    f32x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (acc[k][i] - mean) * rstd * ga[i] + be[i];
                    st4(p.y + row * D + c0 + 64 * k, o);
                    if (p.ys != nullptr) store_split4(p.ys + row * D, c0 + 64 * k, o);
                }
            }
        }
        OC_STAMP(9);
    }
}

size_t dec_cross_ws_floats(int B, int T) { return (size_t)B * H * T * (2 * D + 1); }

// per-device kernel attribute (T = 8 needs 64.1 KiB of dynamic LDS): once per device, under a mutex, outside any stream capture
// (the graphed decode calls it before it begins to capture)
int dec_cross_prepare() {
    static std::mutex mu;
    static bool attr_set[64] = {};
    int dev = 0;
    LADIFF_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return LADIFF_ERR_ARG;
    std::lock_guard<std::mutex> lock(mu);
    if (!attr_set[dev]) {
        LADIFF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dec_cross_apply_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(((size_t)H * TM * 2 * D + H * TM) * sizeof(float))));
        const void* oc[8] = {reinterpret_cast<const void*>(dec_out_cross_kernel<1>), reinterpret_cast<const void*>(dec_out_cross_kernel<2>),
                             reinterpret_cast<const void*>(dec_out_cross_kernel<3>), reinterpret_cast<const void*>(dec_out_cross_kernel<4>),
                             reinterpret_cast<const void*>(dec_out_cross_kernel<5>), reinterpret_cast<const void*>(dec_out_cross_kernel<6>),
                             reinterpret_cast<const void*>(dec_out_cross_kernel<7>), reinterpret_cast<const void*>(dec_out_cross_kernel<8>)};
        for (int t = 1; t <= 8; ++t)
            (void)oc[t - 1];           // static LDS only (155 KB, in the code object's descriptor): nothing to set
        attr_set[dev] = true;
    }
    return 0;
}

// G | U | c of n layers at once (they depend on z and the weights only): pb.gu[l] holds dec_cross_ws_floats(B, T) floats
int launch_decoder_cross_prep(const DecCrossPrepBatch& pb, int n, int B, int T, hipStream_t s) {
    if (B == 0 || n == 0) return 0;
    if (T < 1 || T > TM || n > DEC_PREP_MAX) return LADIFF_ERR_SHAPE;
    // 16-row tiles of the T B memory rows per workgroup: enough workgroups to cover the chip, then as many tiles as possible behind one
    // read of the head's 128 KB of weights
    const int ntiles = (T * B + 15) / 16;
    int tpw = 1;
    while (tpw < 8 && (ntiles + 2 * tpw - 1) / (2 * tpw) * H * n >= 160) tpw *= 2;      // (1, 2, 4, 8 tiles: 33 - 40 us at 128 x 5 rows, no trend)
    hipLaunchKernelGGL(dec_cross_prep_kernel, dim3((ntiles + tpw - 1) / tpw, H, n), dim3(256), 0, s, pb, B, T, tpw);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// y = LN2(x + cross_attention(x, kv)) from the layer's prepared G | U | c (gu_ws, launch_decoder_cross_prep)
int launch_decoder_cross_apply(const float* x, const float* bo, const float* g2, const float* b2, const int32_t* counts, int B, int F,
                               int T, const float* gu_ws, float* y, float* ys, hipStream_t s, const int32_t* row_off, const float* g1,
                               const float* b1) {
    if (B == 0 || F == 0) return 0;
    if (T < 1 || T > TM) return LADIFF_ERR_SHAPE;
    const float* gu = gu_ws;
    const float* cc = gu_ws + (size_t)B * H * T * 2 * D;
    // enough workgroups to cover the chip twice; each re-loads the sample's 8 T KiB of G | U
    int chunks = (512 + B - 1) / B;
    if (chunks < 1) chunks = 1;
    if (chunks > (F + 3) / 4) chunks = (F + 3) / 4;
    const int rows_per_wg = ((F + chunks - 1) / chunks + 15) / 16 * 16;
    chunks = (F + rows_per_wg - 1) / rows_per_wg;
    const size_t lds = ((size_t)H * T * 2 * D + H * T) * sizeof(float);
    LADIFF_TRY(dec_cross_prepare());
#define LADIFF_DC_CASE(TT)                                                                                                   \
    case TT:                                                                                                                 \
        hipLaunchKernelGGL(dec_cross_apply_kernel<TT>, dim3(B, chunks), dim3(256), lds, s, x, gu, cc, counts, bo, g2, b2, F, \
                           rows_per_wg, y, ys, row_off, g1, b1);                                                                              \
        break;
    switch (T) {
        LADIFF_DC_CASE(1) LADIFF_DC_CASE(2) LADIFF_DC_CASE(3) LADIFF_DC_CASE(4)
        LADIFF_DC_CASE(5) LADIFF_DC_CASE(6) LADIFF_DC_CASE(7) LADIFF_DC_CASE(8)
        default: return LADIFF_ERR_SHAPE;
    }
#undef LADIFF_DC_CASE
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// y / ys = LN2(x1n + cross(x1n)), x1n = LN1(x0 + att Wo^T + bo): the decoder layer from the attention output to the feed-forward input
int launch_decoder_out_cross(const float* att_s, const float* x0, const float* wo_s, const float* bo, const float* g1, const float* b1,
                             const float* bo_c, const float* g2, const float* b2, const int32_t* counts, int B, int F, int T,
                             const float* gu_ws, float* y, float* ys, hipStream_t s, const int32_t* row_off) {
    if (B == 0 || F == 0) return 0;
    if (T < 1 || T > TM) return LADIFF_ERR_SHAPE;
    // one workgroup per CU (it holds Wo in its registers): split a sample's frames only as far as that takes to cover the chip
    int chunks = (256 + B - 1) / B;
    if (chunks > (F + OC_ROWS - 1) / OC_ROWS) chunks = (F + OC_ROWS - 1) / OC_ROWS;
    if (chunks < 1) chunks = 1;
    const int rows_per_wg = ((F + chunks - 1) / chunks + OC_ROWS - 1) / OC_ROWS * OC_ROWS;
    chunks = (F + rows_per_wg - 1) / rows_per_wg;
    LADIFF_TRY(dec_cross_prepare());
    OutCrossArgs a{att_s, x0, wo_s, bo, g1, b1, gu_ws, gu_ws + (size_t)B * H * T * 2 * D, bo_c, g2, b2, counts, row_off, y, ys, F, rows_per_wg, nullptr};
#ifdef LADIFF_STAMPS
    a.stamps = g_sys_stamps;
#endif
#define LADIFF_OC_CASE(TT) \
    case TT: hipLaunchKernelGGL(dec_out_cross_kernel<TT>, dim3(B, chunks), dim3(512), 0, s, a); break;
    switch (T) {
        LADIFF_OC_CASE(1) LADIFF_OC_CASE(2) LADIFF_OC_CASE(3) LADIFF_OC_CASE(4)
        LADIFF_OC_CASE(5) LADIFF_OC_CASE(6) LADIFF_OC_CASE(7) LADIFF_OC_CASE(8)
        default: return LADIFF_ERR_SHAPE;
    }
#undef LADIFF_OC_CASE
    LADIFF_LAUNCH_CHECK();
    return 0;
}

}  // namespace ladiff
