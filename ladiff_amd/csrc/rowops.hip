// Row-wise / elementwise kernels (HBM- or latency-bound; one wave per 256-float row, 16-byte accesses).
#include "kernels.h"

namespace ladiff {

constexpr int ROWS_PER_BLOCK = 4;   // 256 threads = 4 waves = 4 rows

// mean / rstd of a 256-wide row held as one f32x4 per lane
__device__ __forceinline__ void row_stats(const f32x4 v, float& mean, float& rstd) {
    float s = wave_sum(v[0] + v[1] + v[2] + v[3]);
    mean = s * (1.f / 256.f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float d = v[i] - mean; q += d * d; }
    q = wave_sum(q);
    rstd = rsqrtf(q * (1.f / 256.f) + LN_EPS);
}

// y = LayerNorm(x)   (nn.LayerNorm(256), eps 1e-5)
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ b, float* __restrict__ y, int M) {
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int c = (threadIdx.x & 63) * 4;
    if (row >= M) return;
    f32x4 v = ld4(x + (size_t)row * D + c);
    float mean, rstd;
    row_stats(v, mean, rstd);
    const f32x4 gg = ld4(g + c), bb = ld4(b + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (v[i] - mean) * rstd * gg[i] + bb[i];
    st4(y + (size_t)row * D + c, v);
}

__global__ __launch_bounds__(256) void layernorm_batch_kernel(const RowBatch rb, int M) {
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int c = (threadIdx.x & 63) * 4;
    if (row >= M) return;
    const int k = blockIdx.y;
    f32x4 v = ld4(rb.a[k] + (size_t)row * D + c);
    float mean, rstd;
    row_stats(v, mean, rstd);
    const f32x4 gg = ld4(rb.g[k] + c), bb = ld4(rb.b[k] + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (v[i] - mean) * rstd * gg[i] + bb[i];
    st4(rb.y[k] + (size_t)row * D + c, v);
}
int launch_layernorm_batch(const RowBatch& rb, int n, int M, hipStream_t s) {
    if (M == 0 || n == 0) return 0;
    LADIFF_CHECK_ARG(n >= 1 && n <= ROW_BATCH_MAX);
    hipLaunchKernelGGL(layernorm_batch_kernel, dim3((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK, n), dim3(256), 0, s, rb, M);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

int launch_layernorm(const float* x, const float* g, const float* b, float* y, int M, hipStream_t s) {
    if (M == 0) return 0;
    hipLaunchKernelGGL(layernorm_kernel, dim3((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0, s, x, g, b, y, M);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// Combine split-K partial planes and apply what follows the GEMM in the reference, one wave per 256-wide row:
//   x = sum_s P[s][row] + bias (+ res[row])
//   RED_PLAIN   y = x                                              (linear_blocks, cross_attention.py:81)
//   RED_LN      y = LN(x)                                          (norm1, mdiff_transformer.py:63)
//   RED_LN_ADD  y = LN(x) + c[step][sample | pad]                  (norm2, then the hoisted ca_block: mdiff_transformer.py:66, :246)
//   RED_LN_MOD  y = SiLU( LN(x) * (1 + scale_step) + shift_step )  (StylizationBlock of the FFN: mdiff_transformer.py:161-162)
struct RedArgs {
    const float* P; const float* bias; const float* res; const float* g; const float* b; const float* tab;
    const int32_t* d_step; const int32_t* counts; float* out; float* outs;
    const int32_t* d_base;            // RED_LN_ADD: the table's first row belongs to step *d_base (windowed c table), NULL = 0
    const float* g2; const float* b2; // RED_LN: a second LayerNorm on the result (the decoder's last layer: norm3, then decoder.norm)
    size_t plane;
    int S, mode, tab_step_stride, Bs, T, pad_row, b_off, M;
};

__global__ __launch_bounds__(256) void reduce_rows_kernel(const RedArgs p) {
    // all arguments are read in one go and pinned (common.h: pin_s): the kernel is a chain of latencies and a kernarg
    // re-read next to each use adds an s_load + wait per pointer
    const float* const P = pin_s(p.P); const float* const bias = pin_s(p.bias); const float* const res = pin_s(p.res);
    const float* const g = pin_s(p.g); const float* const b = pin_s(p.b); const float* const tab = pin_s(p.tab);
    const int32_t* const d_step = pin_s(p.d_step); const int32_t* const counts = pin_s(p.counts);
    const int32_t* const d_base = pin_s(p.d_base);
    float* const out = pin_s(p.out); float* const outs = pin_s(p.outs);
    const size_t plane = pin_s(p.plane);
    const int S = pin_s(p.S), mode = pin_s(p.mode), tab_step_stride = pin_s(p.tab_step_stride), Bs = pin_s(p.Bs),
              T = pin_s(p.T), pad_row = pin_s(p.pad_row), b_off = pin_s(p.b_off), M = pin_s(p.M);
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int c = (threadIdx.x & 63) * 4;
    if (row >= M) return;
    // every load that does not depend on another one is issued first (planes, bias, residual, gamma/beta, the step
    // counter, the sample's latent count), then the table rows that depend on step / count, and only then the
    // arithmetic: two round trips instead of six
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const float* prow = P + (size_t)row * D + c;
    f32x4 pl[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) pl[s] = s < S ? ld4g(prow + s * plane) : zero;
    const f32x4 bi = bias != nullptr ? ld4g(bias + c) : zero;
    const f32x4 rs = res != nullptr ? ld4g(res + (size_t)row * D + c) : zero;
    f32x4 gg = zero, bb = zero, gg2 = zero, bb2 = zero;
    if (mode != RED_PLAIN) { gg = ld4g(g + c); bb = ld4g(b + c); }
    const bool twice = mode == RED_LN && p.g2 != nullptr;
    if (twice) { gg2 = ld4g(p.g2 + c); bb2 = ld4g(p.b2 + c); }
    int step = d_step != nullptr ? *(const __attribute__((address_space(1))) int32_t*)d_step : 0;
    if (d_base != nullptr) step -= *(const __attribute__((address_space(1))) int32_t*)d_base;
    const int b2 = b_off + row / T, tt = row % T;
    int cnt = 0x7fffffff;
    if (mode == RED_LN_ADD && counts != nullptr) cnt = *(const __attribute__((address_space(1))) int32_t*)(counts + b2 % Bs);
    const float* t = tab + (size_t)step * tab_step_stride;
    f32x4 t0 = zero, t1 = zero;
    if (mode == RED_LN_ADD) { t0 = ld4g(t + (size_t)b2 * D + c); t1 = ld4g(t + (size_t)pad_row * D + c); }   // both candidates
    else if (mode == RED_LN_MOD) { t0 = ld4g(t + c); t1 = ld4g(t + 256 + c); }

    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = ((pl[0][i] + pl[1][i]) + pl[2][i]) + pl[3][i] + bi[i] + rs[i];
    if (mode != RED_PLAIN) {
        float mean, rstd;
        row_stats(v, mean, rstd);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (v[i] - mean) * rstd * gg[i] + bb[i];
        if (twice) {
            row_stats(v, mean, rstd);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (v[i] - mean) * rstd * gg2[i] + bb2[i];
        }
        if (mode == RED_LN_ADD) {
            const bool valid = tt < cnt;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += valid ? t0[i] : t1[i];
        } else if (mode == RED_LN_MOD) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = silu(v[i] * (1.f + t0[i]) + t1[i]);
        }
    }
    if (out != nullptr) st4g(out + (size_t)row * D + c, v);
    if (outs != nullptr) store_split4g(outs + (size_t)row * D, c, v);
}

int launch_reduce_rows(const float* P, int S, int M, const float* bias, const float* res, int mode, const float* g,
                       const float* b, const float* tab, int tab_step_stride, const int32_t* d_step,
                       const int32_t* counts, int Bs, int T, int pad_row, int b_off, float* out, float* outs, hipStream_t s,
                       const int32_t* d_base, const float* g2, const float* b2) {
    if (S < 1 || S > 4) return LADIFF_ERR_SHAPE;       // split-K planes: K / 256 <= 4
    RedArgs a;
    a.g2 = g2; a.b2 = b2;
    a.P = P; a.bias = bias; a.res = res; a.g = g; a.b = b; a.tab = tab; a.d_step = d_step; a.counts = counts; a.out = out;
    a.outs = outs; a.plane = (size_t)M * D; a.S = S; a.mode = mode; a.tab_step_stride = tab_step_stride; a.Bs = Bs; a.T = T;
    a.pad_row = pad_row; a.b_off = b_off; a.M = M; a.d_base = d_base;
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0, s, a);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// Input rows of the hoisted cross-attention table: u[step, b] = SiLU(nval[b] * (1 + scale_step) + shift_step) for b < B2
// and u[step, B2] = SiLU(beta * (1 + scale_step) + shift_step) (padded latent rows).  `mod` points at this layer's
// scale|shift of step 0; consecutive steps are `step_stride` floats apart.
__global__ __launch_bounds__(256) void ca_table_input_kernel(const float* __restrict__ nval, const float* __restrict__ beta,
                                                             const float* __restrict__ mod, int step_stride, int B2, int M,
                                                             float* __restrict__ u) {
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int c = (threadIdx.x & 63) * 4;
    if (row >= M) return;
    const int step = row / (B2 + 1), b = row % (B2 + 1);
    const float* m = mod + (size_t)step * step_stride;
    const f32x4 sc = ld4(m + c), sh = ld4(m + 256 + c);
    f32x4 v = b < B2 ? ld4(nval + (size_t)b * D + c) : ld4(beta + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = silu(v[i] * (1.f + sc[i]) + sh[i]);
    st4(u + (size_t)row * D + c, v);
}
__global__ __launch_bounds__(256) void ca_table_input_batch_kernel(const RowBatch rb, int step_stride, int B2, int M, int split_out) {
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int c = (threadIdx.x & 63) * 4;
    if (row >= M) return;
    const int k = blockIdx.y;
    const int step = row / (B2 + 1), b = row % (B2 + 1);
    const float* m = rb.b[k] + (size_t)step * step_stride;
    const f32x4 sc = ld4(m + c), sh = ld4(m + 256 + c);
    f32x4 v = b < B2 ? ld4(rb.a[k] + (size_t)b * D + c) : ld4(rb.g[k] + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = silu(v[i] * (1.f + sc[i]) + sh[i]);
    if (split_out) store_split4(rb.y[k] + (size_t)row * D, c, v);       // the only reader is a f16x3 GEMM: S-format operand rows
    else st4(rb.y[k] + (size_t)row * D + c, v);
}
int launch_ca_table_input_batch(const RowBatch& rb, int n_layers, int step_stride, int n, int B2, hipStream_t s, int split_out) {
    const int M = n * (B2 + 1);
    if (M == 0 || n_layers == 0) return 0;
    LADIFF_CHECK_ARG(n_layers >= 1 && n_layers <= ROW_BATCH_MAX);
    hipLaunchKernelGGL(ca_table_input_batch_kernel, dim3((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK, n_layers), dim3(256), 0, s, rb,
                       step_stride, B2, M, split_out);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

int launch_ca_table_input(const float* nval, const float* beta, const float* mod, int step_stride, int n, int B2, float* u,
                          hipStream_t s) {
    const int M = n * (B2 + 1);
    hipLaunchKernelGGL(ca_table_input_kernel, dim3((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0, s, nval, beta, mod,
                       step_stride, B2, M, u);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// x[b2,t,:] = sample[b2 % Bs, t, :] + pe[t, :]     (ladiff.py:472-474 duplication + position_encoding.py:158)
__global__ __launch_bounds__(256) void add_pe_kernel(const float* __restrict__ sample, const float* __restrict__ pe,
                                                     int Bs, int T, int M, int b_off, float* __restrict__ x, float* __restrict__ xs) {
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int c = (threadIdx.x & 63) * 4;
    if (row >= M) return;
    const int b2 = b_off + row / T, t = row % T;
    f32x4 v = ld4(sample + ((size_t)(b2 % Bs) * T + t) * D + c);
    const f32x4 p = ld4(pe + (size_t)t * D + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += p[i];
    st4(x + (size_t)row * D + c, v);
    if (xs != nullptr) store_split4(xs + (size_t)row * D, c, v);
}

int launch_add_pe(const float* sample, const float* pe, int Bs, int b_off, int b_n, int T, float* x, float* xs, hipStream_t s) {
    const int M = b_n * T;   // rows of samples [b_off, b_off + b_n) of the (duplicated) batch; sample b2 reads latent row b2 % Bs
    hipLaunchKernelGGL(add_pe_kernel, dim3((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0, s, sample, pe, Bs, T, M, b_off, x, xs);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// q0[b,f,:] = pe[f,:]   (queries = zeros + learned PE, ladiff_vae.py:299,:334)
__global__ __launch_bounds__(256) void broadcast_pe_kernel(const float* __restrict__ pe, int F, int M, float* __restrict__ x,
                                                           float* __restrict__ xs) {
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int c = (threadIdx.x & 63) * 4;
    if (row >= M) return;
    const f32x4 v = ld4(pe + (size_t)(row % F) * D + c);
    st4(x + (size_t)row * D + c, v);
    if (xs != nullptr) store_split4(xs + (size_t)row * D, c, v);
}

__global__ __launch_bounds__(256) void broadcast_pe_ragged_kernel(const float* __restrict__ pe, const int32_t* __restrict__ row_off,
                                                                  int F, int F_out, float* __restrict__ x, float* __restrict__ xs,
                                                                  int32_t* __restrict__ row_out) {
    const int b = blockIdx.y, f = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int c = (threadIdx.x & 63) * 4;
    const int r0 = row_off[b];
    if (f >= F || f >= row_off[b + 1] - r0) return;
    const size_t row = (size_t)r0 + f;
    const f32x4 v = ld4(pe + (size_t)f * D + c);
    st4(x + row * D + c, v);
    if (xs != nullptr) store_split4(xs + row * D, c, v);
    if (c == 0) row_out[row] = b * F_out + f;
}
int launch_broadcast_pe_ragged(const float* pe, const int32_t* row_off, int B, int F, int F_out, float* x, float* xs, int32_t* row_out,
                               hipStream_t s) {
    if (B == 0) return 0;
    hipLaunchKernelGGL(broadcast_pe_ragged_kernel, dim3((F + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK, B), dim3(256), 0, s, pe, row_off, F,
                       F_out, x, xs, row_out);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

int launch_broadcast_pe(const float* pe, int B, int F, float* x, float* xs, hipStream_t s) {
    const int M = B * F;
    hipLaunchKernelGGL(broadcast_pe_kernel, dim3((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0, s, pe, F, M, x, xs);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// fp32 [R][K] -> S-format [R][K] (weights are split once per weight table; K multiple of 64)
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int K, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const size_t row = (i * 4) / K;
    const int k = (int)((i * 4) % K);
    store_split4(y + row * K, k, ld4(x + i * 4));
}
int launch_split_rows(const float* x, float* y, int R, int K, hipStream_t s) {
    const size_t n4 = (size_t)R * K / 4;
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, x, y, K, n4);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- LA-VAE encoder plumbing (ladiff_vae.py:162-286)
// out[r, 0:Cp] = [in[r, 0:C], 0 ...]: pads rows to a multiple of 32 columns so that K = nfeats GEMMs can use 16-byte chunks
__global__ __launch_bounds__(256) void pad_cols_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int Cp, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const size_t r = i / Cp; const int c = (int)(i % Cp);
    y[i] = c < C ? x[r * C + c] : 0.f;
}
int launch_pad_cols(const float* x, float* y, int R, int C, int Cp, hipStream_t s) {
    const size_t n = (size_t)R * Cp;
    if (n == 0) return 0;
    hipLaunchKernelGGL(pad_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, C, Cp, n);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// xseq[b, s] = (s < 2T ? motion_token[s] : emb[b, s - 2T]) + pe[s]            ladiff_vae.py:189, :212, :219
// keybits[b] = validity map of the S = 2T + F keys: mu tokens < count, logvar tokens < count, frames < len   :193-209
__global__ __launch_bounds__(256) void encoder_assemble_kernel(const float* __restrict__ token, const float* __restrict__ emb,
                                                               const float* __restrict__ pe, const int32_t* __restrict__ lengths,
                                                               const int32_t* __restrict__ counts, int F, int T2, int S, int M,
                                                               float* __restrict__ x, float* __restrict__ xs,
                                                               uint32_t* __restrict__ keybits) {
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int c = (threadIdx.x & 63) * 4;
    if (row >= M) return;
    const int b = row / S, sq = row % S;
    f32x4 v = sq < T2 ? ld4(token + (size_t)sq * D + c) : ld4(emb + ((size_t)b * F + (sq - T2)) * D + c);
    const f32x4 p = ld4(pe + (size_t)sq * D + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += p[i];
    st4(x + (size_t)row * D + c, v);
    if (xs != nullptr) store_split4(xs + (size_t)row * D, c, v);
    if (sq < 8 && (threadIdx.x & 63) == 0) {     // 8 words of the key map, written by the first 8 rows of the sample
        const int T = T2 / 2, cnt = counts[b], len = lengths[b];
        uint32_t w = 0;
        for (int k = 32 * sq; k < 32 * sq + 32 && k < S; ++k) {
            const bool ok = k < T ? k < cnt : (k < T2 ? (k - T) < cnt : (k - T2) < len);
            if (ok) w |= 1u << (k & 31);
        }
        keybits[(size_t)b * 8 + sq] = w;
    }
}
int launch_encoder_assemble(const float* token, const float* emb, const float* pe, const int32_t* lengths,
                            const int32_t* counts, int B, int F, int T, float* x, float* xs, uint32_t* keybits, hipStream_t s) {
    const int S = 2 * T + F, M = B * S;
    if (S < 8) return LADIFF_ERR_SHAPE;
    hipLaunchKernelGGL(encoder_assemble_kernel, dim3((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0, s, token, emb, pe,
                       lengths, counts, F, 2 * T, S, M, x, xs, keybits);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// mu = dist[0:T], logvar = dist[T:2T]; std = exp(logvar)^0.5; latent = mu + std * eps, rows >= count zeroed
// (ladiff_vae.py:258-268); outputs are sequence-first [T, B, 256] like the reference's.
__global__ __launch_bounds__(256) void encoder_finalize_kernel(const float* __restrict__ out, const float* __restrict__ eps,
                                                               const int32_t* __restrict__ counts, int B, int T, int S,
                                                               float* __restrict__ mu, float* __restrict__ sd,
                                                               float* __restrict__ latent) {
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);      // row = b * T + t
    const int c = (threadIdx.x & 63) * 4;
    if (row >= B * T) return;
    const int b = row / T, t = row % T;
    const f32x4 m = ld4(out + ((size_t)b * S + t) * D + c), lv = ld4(out + ((size_t)b * S + T + t) * D + c);
    const size_t o = ((size_t)t * B + b) * D + c;
    const f32x4 e = ld4(eps + o);
    f32x4 sdv, z;
    const bool valid = t < counts[b];
#pragma unroll
    for (int i = 0; i < 4; ++i) { sdv[i] = sqrtf(expf(lv[i])); z[i] = valid ? m[i] + sdv[i] * e[i] : 0.f; }
    st4(mu + o, m); st4(sd + o, sdv); st4(latent + o, z);
}
int launch_encoder_finalize(const float* out, const float* eps, const int32_t* counts, int B, int T, int S, float* mu, float* sd,
                            float* latent, hipStream_t s) {
    hipLaunchKernelGGL(encoder_finalize_kernel, dim3((B * T + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0, s, out, eps,
                       counts, B, T, S, mu, sd, latent);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// y = relu(x), flat
__global__ __launch_bounds__(256) void relu_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 v = ld4(x + i * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
    st4(y + i * 4, v);
}
int launch_relu(const float* x, float* y, size_t n, hipStream_t s) {
    const size_t n4 = n / 4;
    hipLaunchKernelGGL(relu_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, x, y, n4);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// x[0 .. n) = 0 (x 16-byte aligned).  For everything that runs INSIDE a captured graph: a hipMemsetAsync there becomes a memset
// node, and the memset node of an older graph exec is what faulted in round 3 (api.hip, g_graph_epoch) - kernel nodes do not.
__global__ __launch_bounds__(256) void zero_fill_kernel(float* __restrict__ x, size_t n4, int tail) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) st4(x + i * 4, f32x4{0.f, 0.f, 0.f, 0.f});
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) x[n4 * 4 + threadIdx.x] = 0.f;
}
int launch_zero_fill(float* x, size_t n, hipStream_t s) {
    if (n == 0) return 0;
    const size_t n4 = n / 4;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)((n4 + 255) / 256 > 0 ? (n4 + 255) / 256 : 1)), dim3(256), 0, s, x, n4, (int)(n % 4));
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// dst [Np][256] = src [C][256] followed by zero rows, dstb [Np] = srcb [C] followed by zeros: the decoder's final_layer brought to whole
// 128-column GEMM tiles INSIDE the library (an S-format row is 1 KiB like an fp32 one, and zero rows are zero in both formats), so that a
// caller's table of C rows is never read past its end
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ src, const float* __restrict__ srcb, float* __restrict__ dst,
                                                       float* __restrict__ dstb, int C) {
    const int r = blockIdx.x, t = threadIdx.x;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    if (t < 64) st4(dst + (size_t)r * D + 4 * t, r < C ? ld4(src + (size_t)r * D + 4 * t) : zero);
    if (t == 64) dstb[r] = r < C ? srcb[r] : 0.f;
}
int launch_pad_rows(const float* src, const float* srcb, float* dst, float* dstb, int C, int Np, hipStream_t s) {
    hipLaunchKernelGGL(pad_rows_kernel, dim3(Np), dim3(256), 0, s, src, srcb, dst, dstb, C);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// y = silu(x), flat
__global__ __launch_bounds__(256) void silu_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 v = ld4(x + i * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = silu(v[k]);
    st4(y + i * 4, v);
}
int launch_silu(const float* x, float* y, size_t n, hipStream_t s) {
    const size_t n4 = n / 4;
    hipLaunchKernelGGL(silu_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, x, y, n4);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// Timesteps(768, flip_sin_to_cos=True, freq_shift=0): out[n, 0:384] = cos(t f_i), out[n, 384:768] = sin(t f_i),
// f_i = exp(-ln(1e4) i / 384)   (tools/embeddings.py:263-280)
__global__ __launch_bounds__(384) void sinusoid_kernel(const int64_t* __restrict__ t, float* __restrict__ out) {
    // fp32 op order of the reference (exponent, frequency and angle are rounded to fp32 exactly where torch rounds
    // them); exp / cos / sin are evaluated in fp64 and rounded once, i.e. correctly rounded fp32 results.
    const int n = blockIdx.x, i = threadIdx.x;
    const float ex = (-9.210340371976184f * (float)i) / 384.f;
    const float f = (float)exp((double)ex);
    const float a = (float)t[n] * f;
    out[(size_t)n * 768 + i] = (float)cos((double)a);
    out[(size_t)n * 768 + 384 + i] = (float)sin((double)a);
}
int launch_sinusoid(const int64_t* t, int n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(sinusoid_kernel, dim3(n), dim3(384), 0, s, t, out);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// Classifier-free guidance + one scheduler step (ladiff.py:487-492), elementwise on [B,T,256].
__global__ __launch_bounds__(256) void cfg_step_kernel(const float* __restrict__ eps, float* __restrict__ lat,
                                                       const float* __restrict__ coef, const int32_t* __restrict__ d_step,
                                                       const float* __restrict__ noise, float g, int cfg, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int step = *d_step;
    const float* c = coef + (size_t)step * LADIFF_COEF_STRIDE;
    const float sa = c[0], sb = c[1], kx0 = c[2], kx = c[3], ke = c[4], kn = c[5];
    f32x4 e = ld4(eps + i * 4);
    if (cfg) {
        const f32x4 ec = ld4(eps + (n4 + i) * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) e[k] = e[k] + g * (ec[k] - e[k]);
    }
    f32x4 x = ld4(lat + i * 4);
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (noise != nullptr && kn != 0.f) z = ld4(noise + ((size_t)step * n4 + i) * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float x0 = (x[k] - sb * e[k]) / sa;
        x[k] = kx0 * x0 + kx * x[k] + ke * e[k] + kn * z[k];
    }
    st4(lat + i * 4, x);
}
int launch_cfg_step(const float* eps, float* lat, const float* coef, const int32_t* d_step, const float* noise,
                    float g, int cfg, int B, int T, hipStream_t s) {
    const size_t n4 = (size_t)B * T * D / 4;
    hipLaunchKernelGGL(cfg_step_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, eps, lat, coef, d_step, noise, g, cfg, n4);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- fused end of a denoiser step
// One wave per latent row (b, t): final LayerNorm of both guidance branches (encoder.norm, cross_attention.py:84-85),
// eps = eps_u + g (eps_c - eps_u) (ladiff.py:487-490), the scheduler step (coefficient row of this step), and the NEXT
// step's network input x = cat([latents]*2) + query_pos.pe (ladiff.py:472-474, ladiff_denoiser.py:251) written in place
// over the last layer's output (fp32 + optional S-format twin).  The last workgroup to finish bumps the step counter
// (every workgroup has read it before taking its ticket), so the whole tail of a step is one launch instead of five.
__global__ __launch_bounds__(256) void step_tail_kernel(float* __restrict__ x, float* __restrict__ xs, const float* __restrict__ ng,
                                                        const float* __restrict__ nb, float* __restrict__ lat,
                                                        const float* __restrict__ coef, int32_t* __restrict__ d_step,
                                                        const float* __restrict__ noise, const float* __restrict__ pe, float g,
                                                        int cfg, int B, int T, const NoiseGen gen) {
    const int step = d_step[0];
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);      // row = b * T + t of the B prompts
    const int c = (threadIdx.x & 63) * 4;
    const int M = B * T;
    if (row < M) {
        const float* cf = coef + (size_t)step * LADIFF_COEF_STRIDE;
        const float sa = cf[0], sb = cf[1], kx0 = cf[2], kx = cf[3], ke = cf[4], kn = cf[5];
        const f32x4 gg = ld4(ng + c), bb = ld4(nb + c);
        f32x4 eu = ld4(x + (size_t)row * D + c), ec = eu;
        if (cfg) ec = ld4(x + (size_t)(M + row) * D + c);
        float mean, rstd;
        row_stats(eu, mean, rstd);
#pragma unroll
        for (int i = 0; i < 4; ++i) eu[i] = (eu[i] - mean) * rstd * gg[i] + bb[i];
        if (cfg) {                       // without guidance the network ran on the B latents only (ladiff.py:472-490)
            row_stats(ec, mean, rstd);
#pragma unroll
            for (int i = 0; i < 4; ++i) ec[i] = (ec[i] - mean) * rstd * gg[i] + bb[i];
        }
        f32x4 l = ld4(lat + (size_t)row * D + c);
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if (gen.on && kn != 0.f) {       // drawn here (noise_gen.h): the same function of (seed, step, global prompt, latent, column) as everywhere
            float zz[4];
            noise_normal4(gen, step, gen.prompt0 + (unsigned)(row / T), row % T, c / 4, zz);
            z = f32x4{zz[0], zz[1], zz[2], zz[3]};
        } else if (noise != nullptr && kn != 0.f) z = ld4(noise + ((size_t)step * M + row) * D + c);
        const f32x4 p = ld4(pe + (size_t)(row % T) * D + c);
        f32x4 xn;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float e = cfg ? eu[i] + g * (ec[i] - eu[i]) : eu[i];
            const float x0 = (l[i] - sb * e) / sa;
            l[i] = kx0 * x0 + kx * l[i] + ke * e + kn * z[i];
            xn[i] = l[i] + p[i];
        }
        st4(lat + (size_t)row * D + c, l);
        st4(x + (size_t)row * D + c, xn);
        if (cfg) st4(x + (size_t)(M + row) * D + c, xn);
        if (xs != nullptr) {
            store_split4(xs + (size_t)row * D, c, xn);
            if (cfg) store_split4(xs + (size_t)(M + row) * D, c, xn);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int ticket = atomicAdd(&d_step[1], 1);
        if (ticket == (int)gridDim.x - 1) { d_step[1] = 0; d_step[0] = step + 1; }
    }
}
int launch_step_tail(float* x, float* xs, const float* ng, const float* nb, float* lat, const float* coef, int32_t* d_step,
                     const float* noise, const float* pe, float g, int cfg, int B, int T, hipStream_t s, const NoiseGen& gen) {
    const int M = B * T;
    hipLaunchKernelGGL(step_tail_kernel, dim3((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0, s, x, xs, ng, nb, lat, coef,
                       d_step, noise, pe, g, cfg, B, T, gen);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// the generator's values as a tensor: out[i][b][t][:] = the noise of schedule position step0 + i, global prompt prompt0 + b, latent t
__global__ __launch_bounds__(256) void noise_fill_kernel(const NoiseGen gen, int step0, int n, int B, int T, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;          // one 16-byte chunk per thread
    const size_t per_step = (size_t)B * T * (D / 4);
    if (i >= per_step * n) return;
    const int st = (int)(i / per_step);
    const size_t r = i % per_step;
    const int chunk = (int)(r % (D / 4)), row = (int)(r / (D / 4));
    float z[4];
    noise_normal4(gen, step0 + st, gen.prompt0 + (unsigned)(row / T), row % T, chunk, z);
    st4(out + i * 4, f32x4{z[0], z[1], z[2], z[3]});
}
int launch_noise_fill(const NoiseGen& gen, int step0, int n, int B, int T, float* out, hipStream_t s) {
    const size_t n4 = (size_t)n * B * T * (D / 4);
    if (n4 == 0) return 0;
    hipLaunchKernelGGL(noise_fill_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, gen, step0, n, B, T, out);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

__global__ void advance_kernel(int32_t* d_step) { if (threadIdx.x == 0) *d_step += 1; }
int launch_advance(int32_t* d_step, hipStream_t s) {
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(64), 0, s, d_step);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// latents = noise * valid * sigma   (ladiff.py:380-390, :407)
__global__ __launch_bounds__(256) void init_latents_kernel(const float* __restrict__ noise, const int32_t* __restrict__ counts,
                                                           float sigma, int T, int M, float* __restrict__ lat) {
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int c = (threadIdx.x & 63) * 4;
    if (row >= M) return;
    const int b = row / T, t = row % T;
    f32x4 v = ld4(noise + (size_t)row * D + c);
    const bool valid = counts == nullptr || t < counts[b];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = valid ? v[i] * sigma : 0.f;
    st4(lat + (size_t)row * D + c, v);
}
int launch_init_latents(const float* noise, const int32_t* counts, float sigma, float* lat, int B, int T, hipStream_t s) {
    const int M = B * T;
    hipLaunchKernelGGL(init_latents_kernel, dim3((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0, s, noise, counts, sigma, T, M, lat);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// z[t,b,:] = latents[b,t,:] with rows t >= counts[b] zeroed   (ladiff.py:500, :562-566)
// status (optional): the pipeline loop's abort word.  When it is non-zero the loop did not finish and the latents are partial:
// the result is POISONED with NaN so that a caller who never reads the status cannot mistake it for a sample.
__global__ __launch_bounds__(256) void finalize_latents_kernel(const float* __restrict__ lat, const int32_t* __restrict__ counts,
                                                               int B, int T, int M, float* __restrict__ z,
                                                               const unsigned* __restrict__ status) {
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int c = (threadIdx.x & 63) * 4;
    if (row >= M) return;
    const int b = row / T, t = row % T;
    f32x4 v = ld4(lat + (size_t)row * D + c);
    const bool valid = counts == nullptr || t < counts[b];
    if (!valid) v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (status != nullptr && status[0] != 0u) { const float q = __builtin_nanf(""); v = f32x4{q, q, q, q}; }
    st4(z + ((size_t)t * B + b) * D + c, v);
}
int launch_finalize_latents(const float* lat, const int32_t* counts, float* z, int B, int T, hipStream_t s, const unsigned* status) {
    const int M = B * T;
    hipLaunchKernelGGL(finalize_latents_kernel, dim3((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0, s, lat, counts, B, T, M, z, status);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// feats[B, F, C] rows from the padded tile output tmp[M, ldt] of the final projection: row r goes to feats row row_map[r] (ragged rows:
// the caller zero-filled feats) or to row r itself, zeroed when (r % F) >= row_len[r / F] (padded frames, ladiff_vae.py:356-360).
// C = 263 / 251 is not a multiple of 4: 4-byte stores, a row per wave.
__global__ __launch_bounds__(256) void scatter_feats_kernel(const float* __restrict__ tmp, int ldt, int C, int M, int F,
                                                            const int32_t* __restrict__ row_len, const int32_t* __restrict__ row_map,
                                                            float* __restrict__ feats) {
    const int r = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= M) return;
    const int dst = row_map != nullptr ? row_map[r] : r;
    const bool valid = row_len == nullptr || (r % F) < row_len[r / F];
    const float* src = tmp + (size_t)r * ldt;
    float* out = feats + (size_t)dst * C;
    for (int c = lane; c < C; c += 64) out[c] = valid ? src[c] : 0.f;
}
int launch_scatter_feats(const float* tmp, int ldt, int C, int M, int F, const int32_t* row_len, const int32_t* row_map, float* feats,
                         hipStream_t s) {
    if (M == 0) return 0;
    hipLaunchKernelGGL(scatter_feats_kernel, dim3((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0, s, tmp, ldt, C, M, F, row_len, row_map, feats);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

}  // namespace ladiff
