// General-N text conditioning of the denoiser layer (more than one text token per prompt: the `clip_hidden` / `bert`
// branches of MldTextEncoder, mld_clip.py:80-86), fp32 arithmetic:
//   * sa_block attention over [T latent rows | N text tokens | time token] keys          mdiff_transformer.py:296-313
//   * LinearTemporalCrossAttention (:219-247) in two parts:
//       per call   att[b,h,d,l] = sum_n softmax_n(key)[b,n,h,d] * value[b,n,h,l]          (text only: step-invariant)
//       per step   y[b,t,h,l]   = sum_d softmax_d(query)[b,t,h,d] * valid[b,t] * att[b,h,d,l],
//                  u            = SiLU( LN(y) * (1 + scale) + shift )                      (StylizationBlock input, :152-162)
//     the query / out projections around them are the existing GEMMs.
// With ONE text token the whole block collapses into a table (denoiser.hip); these kernels are the literal
// path for N > 1, latency-bound row work on [2B T, 256] like the rest of the fp32 loop.
#include "kernels.h"

namespace ladiff {

// one workgroup per (sample, head): att[d][l] = sum_n softmax_n(k[n][d]) v[n][l]
__global__ __launch_bounds__(256) void lca_kv_kernel(const float* __restrict__ key, const float* __restrict__ value, int N,
                                                     float* __restrict__ att) {
    extern __shared__ __attribute__((aligned(16))) float sm[];          // ks [N][64], vs [N][64]
    float* ks = sm;
    float* vs = sm + (size_t)N * 64;
    const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
    for (int u = tid; u < N * 64; u += 256) {
        const int n = u >> 6, d = u & 63;
        ks[u] = key[((size_t)b * N + n) * D + h * 64 + d];
        vs[u] = value[((size_t)b * N + n) * D + h * 64 + d];
    }
    __syncthreads();
    if (tid < 64) {                                                     // softmax over the N tokens, per feature d   (:235)
        float m = -INFINITY;
        for (int n = 0; n < N; ++n) m = fmaxf(m, ks[n * 64 + tid]);
        float l = 0.f;
        for (int n = 0; n < N; ++n) { const float e = expf(ks[n * 64 + tid] - m); ks[n * 64 + tid] = e; l += e; }
        const float inv = 1.f / l;
        for (int n = 0; n < N; ++n) ks[n * 64 + tid] *= inv;
    }
    __syncthreads();
    // 64 x 64 outputs over 256 threads: thread -> (d = tid / 4 .. , 16 consecutive l)
    const int d = tid >> 2, l0 = (tid & 3) * 16;
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int n = 0; n < N; ++n) {
        const float kd = ks[n * 64 + d];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fmaf(kd, vs[n * 64 + l0 + i], acc[i]);
    }
    float* o = att + (((size_t)b * H + h) * 64 + d) * 64 + l0;
#pragma unroll
    for (int i = 0; i < 16; i += 4) st4(o + i, f32x4{acc[i], acc[i + 1], acc[i + 2], acc[i + 3]});
}

int launch_lca_kv(const float* key, const float* value, int B, int N, float* att, hipStream_t s) {
    if (B == 0) return 0;
    if (N < 1 || N > LADIFF_CLIP_MAX_POSITIONS) return LADIFF_ERR_SHAPE;
    hipLaunchKernelGGL(lca_kv_kernel, dim3(B, H), dim3(256), (size_t)N * 128 * sizeof(float), s, key, value, N, att);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// one workgroup per sample: its T rows.  q [B T, 256] pre-softmax; mod = scale | shift (512 floats) at
// mod + step * step_stride + sample * sample_stride; u [B T, 256].
template <int TMAX>
__global__ __launch_bounds__(256) void lca_apply_kernel(const float* __restrict__ q, const float* __restrict__ att,
                                                        const int32_t* __restrict__ counts, int Bs, int b_off, int T,
                                                        const float* __restrict__ mod, int step_stride, int sample_stride,
                                                        const int32_t* __restrict__ d_step, const float* __restrict__ g,
                                                        const float* __restrict__ be, float* __restrict__ u) {
    __shared__ __attribute__((aligned(16))) float qs[TMAX * D];
    __shared__ float red[TMAX][4][2];
    const int b = blockIdx.x, bg = b_off + b, tid = threadIdx.x, h = tid >> 6, l = tid & 63, lane = tid & 63;
    int nv = counts != nullptr ? counts[bg % Bs] : T;
    nv = nv > T ? T : nv;
    // softmax over the 64 features of (row, head): wave = head, lane = feature                       (:234)
    for (int t = 0; t < T; ++t) {
        const float v = q[((size_t)b * T + t) * D + tid];
        float m = v;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        const float e = expf(v - m);
        const float ssum = wave_sum(e);
        qs[t * D + tid] = t < nv ? e / ssum : 0.f;                      // padded latent rows: query zeroed   (:242-243)
    }
    __syncthreads();
    // y[t][h][l] = sum_d qs[t][h][d] att[b][h][d][l]: thread = (h, l), coalesced over l
    float y[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) y[t] = 0.f;
    const float* ap = att + (((size_t)bg * H + h) * 64) * 64 + l;
    for (int d = 0; d < 64; ++d) {
        const float a = ap[(size_t)d * 64];
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T) y[t] = fmaf(qs[t * D + h * 64 + d], a, y[t]);
    }
    // LayerNorm over the 256 outputs of a row (two-pass), AdaLN, SiLU                                  (:158-162)
    const int step = d_step != nullptr ? *d_step : 0;
    const float* mp = mod + (size_t)step * step_stride + (size_t)bg * sample_stride;
    const float scale = mp[tid], shift = mp[D + tid], gg = g[tid], bb = be[tid];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        if (t < T) {
            const float s1 = wave_sum(y[t]);
            if (lane == 0) red[t][h][0] = s1;
        }
    }
    __syncthreads();
    float mean[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        mean[t] = 0.f;
        if (t < T) {
            mean[t] = (red[t][0][0] + red[t][1][0] + red[t][2][0] + red[t][3][0]) * (1.f / 256.f);
            const float dlt = y[t] - mean[t];
            const float s2 = wave_sum(dlt * dlt);
            if (lane == 0) red[t][h][1] = s2;
        }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        if (t < T) {
            const float var = (red[t][0][1] + red[t][1][1] + red[t][2][1] + red[t][3][1]) * (1.f / 256.f);
            const float hn = (y[t] - mean[t]) * rsqrtf(var + LN_EPS) * gg + bb;
            u[((size_t)b * T + t) * D + tid] = silu(hn * (1.f + scale) + shift);
        }
    }
}

int launch_lca_apply(const float* q, const float* att, const int32_t* counts, int Bs, int b_off, int b_n, int T, const float* mod,
                     int step_stride, int sample_stride, const int32_t* d_step, const float* g, const float* be, float* u,
                     hipStream_t s) {
    if (b_n == 0) return 0;
    if (T < 1 || T > LADIFF_MAX_LATENTS) return LADIFF_ERR_SHAPE;
    hipLaunchKernelGGL(lca_apply_kernel<LADIFF_MAX_LATENTS>, dim3(b_n), dim3(256), 0, s, q, att, counts, Bs, b_off, T, mod, step_stride,
                       sample_stride, d_step, g, be, u);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// sa_block attention with N text tokens: queries = the T latent rows of a sample, keys = [T latents | N text | time].
// One workgroup per sample, wave = head, lane = feature; text K|V rows [B2][N][512] (k | v) from the text cache.
template <int TMAX>
__global__ __launch_bounds__(256) void den_self_attn_general_kernel(const float* __restrict__ qkv, const float* __restrict__ text_kv,
                                                                    int N, const float* __restrict__ tables, int kv_off,
                                                                    int step_stride, const int32_t* __restrict__ d_step,
                                                                    const int32_t* __restrict__ counts, int Bs, int b_off, int T,
                                                                    float* __restrict__ out) {
    const int b2 = blockIdx.x, bg = b_off + b2, col = threadIdx.x;
    int nv = counts != nullptr ? counts[bg % Bs] : T;
    nv = nv > T ? T : nv;
    const float* tk = tables + (size_t)(d_step != nullptr ? *d_step : 0) * step_stride + kv_off;
    const float ktime = tk[col], vtime = tk[256 + col];
    float qv[TMAX], kl[TMAX], vl[TMAX];
#pragma unroll
    for (int i = 0; i < TMAX; ++i) {
        qv[i] = kl[i] = vl[i] = 0.f;
        if (i < T) {
            const float* r = qkv + ((size_t)b2 * T + i) * 768 + col;
            qv[i] = r[0] * 0.125f; kl[i] = r[256]; vl[i] = r[512];
        }
    }
    // online softmax per query row over the three key groups (scores of one (row, key) = a wave reduction over 64 features)
    float m[TMAX], l[TMAX], o[TMAX];
#pragma unroll
    for (int i = 0; i < TMAX; ++i) { m[i] = -INFINITY; l[i] = 0.f; o[i] = 0.f; }
    auto feed = [&](int i, float sc, float v) {
        const float mn = fmaxf(m[i], sc);
        const float a = expf(m[i] - mn), e = expf(sc - mn);
        l[i] = l[i] * a + e; o[i] = o[i] * a + e * v; m[i] = mn;
    };
#pragma unroll
    for (int j = 0; j < TMAX; ++j) {
        if (j < nv) {
#pragma unroll
            for (int i = 0; i < TMAX; ++i)
                if (i < T) feed(i, wave_sum(qv[i] * kl[j]), vl[j]);
        }
    }
    for (int n = 0; n < N; ++n) {
        const float kt = text_kv[((size_t)bg * N + n) * 512 + col], vt = text_kv[((size_t)bg * N + n) * 512 + 256 + col];
#pragma unroll
        for (int i = 0; i < TMAX; ++i)
            if (i < T) feed(i, wave_sum(qv[i] * kt), vt);
    }
#pragma unroll
    for (int i = 0; i < TMAX; ++i) {
        if (i < T) {
            feed(i, wave_sum(qv[i] * ktime), vtime);
            out[((size_t)b2 * T + i) * D + col] = o[i] / l[i];
        }
    }
}

int launch_denoiser_self_attention_general(const float* qkv, const float* text_kv, int N, const float* tables, int kv_off,
                                           int step_stride, const int32_t* d_step, const int32_t* counts, int Bs, int b_off,
                                           int b_n, int T, float* out, hipStream_t s) {
    if (b_n == 0) return 0;
    if (T < 1 || T > LADIFF_MAX_LATENTS || N < 1) return LADIFF_ERR_SHAPE;
    hipLaunchKernelGGL(den_self_attn_general_kernel<LADIFF_MAX_LATENTS>, dim3(b_n), dim3(256), 0, s, qkv, text_kv, N, tables, kv_off,
                       step_stride, d_step, counts, Bs, b_off, T, out);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

}  // namespace ladiff
