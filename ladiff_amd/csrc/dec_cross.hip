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

namespace ladiff {

namespace {
constexpr int TM = LADIFF_MAX_LATENTS;
}

// one workgroup per (sample, head): G | U rows [T][2][256] of that head and the score offsets c [T] from the sample's K|V rows.
// Wq rows h*64 .. are read coalesced over the output column; the Wo tile [256][64] goes through LDS (row stride 65: the
// per-thread row walk is conflict-free).
__global__ __launch_bounds__(256) void dec_cross_prep_kernel(const DecCrossPrepBatch pb, int B, int T) {
    __shared__ float wos[D * 65];
    __shared__ float ks[TM * DH], vs[TM * DH];
    const int b = blockIdx.x, h = blockIdx.y, n = threadIdx.x, layer = blockIdx.z;
    const float* __restrict__ kv = pb.kv[layer]; const float* __restrict__ wq = pb.wq[layer]; const float* __restrict__ bq = pb.bq[layer];
    const float* __restrict__ wo = pb.wo[layer]; float* __restrict__ gu = pb.gu[layer];
    float* __restrict__ cc = gu + (size_t)B * H * T * 2 * D;
    float qv[DH];                                                                  // Wq[h*64 + d][n]: all 64 loads in flight at once
#pragma unroll
    for (int d = 0; d < DH; ++d) qv[d] = wq[(size_t)(h * DH + d) * D + n];
    for (int u = n; u < D * DH; u += 256) wos[(u >> 6) * 65 + (u & 63)] = wo[(size_t)(u >> 6) * D + h * DH + (u & 63)];
    for (int u = n; u < T * DH; u += 256) {
        const int j = u >> 6, d = u & 63;
        ks[u] = kv[((size_t)j * B + b) * 2 * D + h * DH + d];
        vs[u] = kv[((size_t)j * B + b) * 2 * D + D + h * DH + d];
    }
    __syncthreads();
    float g[TM], u_[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) { g[j] = 0.f; u_[j] = 0.f; }
#pragma unroll
    for (int d = 0; d < DH; ++d) {
        const float q = qv[d];
        const float w = wos[n * 65 + d];                                           // Wo[n][h*64 + d]
#pragma unroll
        for (int j = 0; j < TM; ++j)
            if (j < T) { g[j] = fmaf(q, ks[j * DH + d], g[j]); u_[j] = fmaf(w, vs[j * DH + d], u_[j]); }
    }
#pragma unroll
    for (int j = 0; j < TM; ++j)
        if (j < T) {
            float* o = gu + ((((size_t)b * H + h) * T + j) * 2) * D;
            o[n] = g[j] * 0.125f;                                                  // 1 / sqrt(64), exact
            o[D + n] = u_[j];
        }
    if (n < T) {                                                                   // c[h][j] = bq_h . k[b,j,h] / 8
        float c = 0.f;
        for (int d = 0; d < DH; ++d) c = fmaf(bq[h * DH + d], ks[n * DH + d], c);
        cc[((size_t)b * H + h) * T + n] = c * 0.125f;
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
            for (int j = 0; j < T; ++j) { sc[j] = j < nv ? sc[j] : -INFINITY; m = fmaxf(m, sc[j]); }   // tokens >= count masked (:408-409)
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
        attr_set[dev] = true;
    }
    return 0;
}

// G | U | c of n layers at once (they depend on z and the weights only): pb.gu[l] holds dec_cross_ws_floats(B, T) floats
int launch_decoder_cross_prep(const DecCrossPrepBatch& pb, int n, int B, int T, hipStream_t s) {
    if (B == 0 || n == 0) return 0;
    if (T < 1 || T > TM || n > DEC_PREP_MAX) return LADIFF_ERR_SHAPE;
    hipLaunchKernelGGL(dec_cross_prep_kernel, dim3(B, H, n), dim3(256), 0, s, pb, B, T);
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

}  // namespace ladiff
