// Self-attention input projection + attention core of the denoiser's sa_block in ONE launch (f16x3 mode):
//   qkv = x . in_proj_weight^T + in_proj_bias ; per head: softmax(q k^T / 8) v over [T latent rows | text token | time token]
// (nn.MultiheadAttention inside TransformerEncoderLayer, mdiff_transformer.py:57-61 / :296-313; masks as in
// den_self_attn_kernel, attention.hip).  Before: a GEMM launch (240 workgroups) + an attention launch (one workgroup per
// sample); the attention launch is ~4 us of pure launch / latency for 0.3 MFLOP of work per sample.
//
// A workgroup owns (R samples, one head): R * T <= 48 rows x 192 columns (q_h | k_h | v_h) of the projection, K = 256
// streamed in four 64-wide stages through a two-stage LDS ring (48 A rows + 192 W rows = 60 KiB per stage), producer /
// consumer waves as in gemm_kr.hip / gemm_rowln.hip.  The q/k/v tile then stays in LDS and all eight waves run the
// attention on it: one thread per (row, key) dot product, one thread per row for the softmax, one thread per (row, 4
// columns) for P.V, S-format stores.  The text / time tokens' K|V slices are fetched by the producers at entry.
#include "model.h"

namespace ladiff {

namespace {
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void dma16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
constexpr int QA_BN = 192;
constexpr int QA_MAXR = 15;                       // samples per workgroup: (R + 1) * 32 float4 of extra K|V <= 512 producer slots
}  // namespace

struct QkvAttnArgs {
    const float* x; const float* w; const float* bias;      // x [M, 256] S-format, in_proj_weight [768, 256] S-format, bias [768]
    const float* text_kv; const float* tables;              // text cache rows [B2][512] (k | v), time tables
    const int32_t* d_step; const int32_t* counts;
    float* out;                                             // [M, 256] S-format
    int kv_off, step_stride, Bs, b_off, b_n, T, R;
};

template <int BM>
__global__ __launch_bounds__(512) void qkv_attn_kernel(const QkvAttnArgs p) {
    constexpr int BN = QA_BN;
    constexpr int ROWS = BM + BN;                  // 240 LDS rows of 256 B per stage
    constexpr int STAGE = ROWS * 64;
    constexpr int PPW = ROWS / 16;                 // one-KiB pieces per producer wave per stage
    constexpr int GA = BM / 16;                    // of which A pieces
    constexpr int NK = 4;                          // K = 256
    constexpr int QLD = BN + 4;                    // row stride of the fp32 q|k|v tile
    constexpr int RM = BM / 16, RN = BN / 4 / 16;  // RM x 3 MFMA tiles per consumer wave
    // LDS after the GEMM (stage 0 region): q|k|v tile, extra keys/values, scores
    constexpr int OFF_X = BM * QLD;                // extras: [R][128] text k|v of this head, then [128] time k|v
    constexpr int OFF_S = OFF_X + (QA_MAXR + 1) * 128;  // scores / probabilities [BM][16]
    static_assert(OFF_S + BM * 16 <= STAGE, "attention scratch must fit in one stage");

    __shared__ __attribute__((aligned(1024))) float lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = pin_s(p.T), R = pin_s(p.R), b_n = pin_s(p.b_n);
    const int h = blockIdx.y;
    const int s0 = blockIdx.x * R;                 // first local sample of this workgroup
    const int ns = (b_n - s0) < R ? (b_n - s0) : R;
    const int row0 = s0 * T, nrows = ns * T, M = b_n * T;
    float xk[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};     // producers: extra K|V words, parked until the GEMM is done
    // attention role of this thread, known up front: (row, key) for the scores; its sample's latent count is fetched now
    const int nkeys = T + 2;
    const int arow_ = tid / nkeys, akey = tid - arow_ * nkeys;
    int nv = T;
    if (arow_ < nrows && p.counts != nullptr) nv = p.counts[(p.b_off + s0 + arow_ / T) % p.Bs];
    nv = nv > T ? T : nv;

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        const int pw = wave - 4, ptid = tid - 256;
        // text token K|V of the ns samples (head slice: 64 k + 64 v floats each) and the time token's; oldest loads in
        // the queue, so the counted waits below retire them first
        const int step = p.d_step != nullptr ? *p.d_step : 0;
        const float* tkv = p.tables + (size_t)step * p.step_stride + p.kv_off;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int f4 = ptid + u * 256;          // float4 index into [ns + 1][128]
            const int sx = f4 >> 5, c4 = (f4 & 31) * 4;
            if (sx <= ns) {
                const float* src = sx < ns ? p.text_kv + (size_t)(p.b_off + s0 + sx) * 512 + (c4 < 64 ? c4 : 192 + c4) + h * 64
                                           : tkv + (c4 < 64 ? c4 : 192 + c4) + h * 64;
                const f32x4 v = ld4(src);
                xk[u][0] = v[0]; xk[u][1] = v[1]; xk[u][2] = v[2]; xk[u][3] = v[3];
            }
        }
        const int rl = 4 * pw + (lane >> 4);
        const int kl = ((lane & 15) ^ rl) << 2;
        const float* arow[GA];
#pragma unroll
        for (int i = 0; i < GA; ++i) {
            int gr = row0 + 16 * i + rl; gr = gr < M ? gr : M - 1;
            arow[i] = p.x + (size_t)gr * D + kl;
        }
        // W tile row wr = 16 (i - GA) + rl: part = wr / 64 (q, k, v), global row = part * 256 + h * 64 + wr % 64
        const float* const wbase = p.w + (size_t)(h * 64 + rl) * D + kl;
        float* const lbase = lds + 4 * pw * 64;
        auto issue = [&](int kt) __attribute__((always_inline)) {
            float* const dst = lbase + (kt & 1) * STAGE;
            const int k0 = kt << 6;
#pragma unroll
            for (int i = 0; i < GA; ++i) dma16(arow[i] + k0, dst + 16 * i * 64);
#pragma unroll
            for (int i = GA; i < PPW; ++i) {
                const int wr = 16 * (i - GA);       // + rl
                dma16(wbase + (size_t)((wr >> 6) * 256 + (wr & 63)) * D + k0, dst + 16 * i * 64);
            }
        };
        // one stage in flight, every wait `vmcnt(0)`: LDS-DMA requests do not complete in issue order (gemm_big.hip, header)
        issue(0);
        for (int kt = 0; kt < NK; ++kt) {
            wait_vm<0>();
            __builtin_amdgcn_s_barrier();          // A(kt)
            if (kt + 1 < NK) issue(kt + 1);
            __builtin_amdgcn_s_barrier();          // B(kt)
        }
        // park the extra K|V in LDS (stage 0 is idle: every consumer passed B(3) after its last LDS read)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int f4 = ptid + u * 256;
            if ((f4 >> 5) <= ns) st4(lds + OFF_X + f4 * 4, f32x4{xk[u][0], xk[u][1], xk[u][2], xk[u][3]});
        }
    } else {
        // ------------------------------------------------------------------ consumers: wave w owns tile columns 48 w .. 48 w + 47
        f32x4 acc[RM][RN];
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int frow = lane & 15, fk = lane >> 4;
        // bias of this lane's columns: tile column tc = 48 w + 16 j + frow -> part tc / 64, within tc % 64
        float bcol[RN];
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const int tc = 48 * wave + 16 * j + frow;
            bcol[j] = p.bias[(tc >> 6) * 256 + h * 64 + (tc & 63)];
        }
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            const float* sa = lds + (kt & 1) * STAGE;
            const float* sb = sa + (BM + 48 * wave) * 64;
            __builtin_amdgcn_s_barrier();          // A(kt)
            s16x8 ah[2][RM], al[2][RM], bh[2][RN], bl[2][RN];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int ch = 4 * g + fk, cl = 8 + 4 * g + fk;
#pragma unroll
                for (int i = 0; i < RM; ++i) {
                    const int r = i * 16 + frow;
                    ah[g][i] = __builtin_bit_cast(s16x8, ld4(sa + r * 64 + ((ch ^ frow) << 2)));
                    al[g][i] = __builtin_bit_cast(s16x8, ld4(sa + r * 64 + ((cl ^ frow) << 2)));
                }
#pragma unroll
                for (int j = 0; j < RN; ++j) {
                    const int r = j * 16 + frow;   // (BM + 48 w + r) & 15 == frow: 48 and BM are multiples of 16
                    bh[g][j] = __builtin_bit_cast(s16x8, ld4(sb + r * 64 + ((ch ^ frow) << 2)));
                    bl[g][j] = __builtin_bit_cast(s16x8, ld4(sb + r * 64 + ((cl ^ frow) << 2)));
                }
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) {
#pragma unroll
                for (int i = 0; i < RM; ++i)
#pragma unroll
                    for (int j = 0; j < RN; ++j) acc[i][j] = MFMA16_S16(al[g][i], bh[g][j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < RM; ++i)
#pragma unroll
                    for (int j = 0; j < RN; ++j) acc[i][j] = MFMA16_S16(ah[g][i], bl[g][j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < RM; ++i)
#pragma unroll
                    for (int j = 0; j < RN; ++j) acc[i][j] = MFMA16_S16(ah[g][i], bh[g][j], acc[i][j], 0, 0, 0);
                if (g == 0) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();  // B(kt)
                }
            }
        }
        // q|k|v tile (+ bias; q scaled by 1/sqrt(64), exact) -> LDS, fp32 [BM][QLD]
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j) {
                const int tc = 48 * wave + 16 * j + frow;
                const float sc = tc < 64 ? 0.125f : 1.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) lds[(16 * i + 4 * fk + r) * QLD + tc] = (acc[i][j][r] + bcol[j]) * sc;
            }
    }
    __syncthreads();                               // tile and extras are in LDS

    // ---------------------------------------------------------------------- attention, all 512 threads
    const float* qt = lds;                         // q cols 0-63, k 64-127, v 128-191
    const float* xt = lds + OFF_X;                 // [ns][128] text k|v, [128] time k|v at index ns
    float* st = lds + OFF_S;                       // [BM][16] scores -> probabilities
    {   // one thread per (row, key): 64-long dot product
        const int row = arow_, j = akey;
        if (row < nrows) {
            const int sx = row / T;
            const float* kp = j < T ? qt + (sx * T + j) * QLD + 64 : (j == T ? xt + sx * 128 : xt + ns * 128);
            const float* qp = qt + row * QLD;
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < 64; c += 4) {
                const f32x4 a = ld4(qp + c), b = ld4(kp + c);
                d = fmaf(a[0], b[0], d); d = fmaf(a[1], b[1], d); d = fmaf(a[2], b[2], d); d = fmaf(a[3], b[3], d);
            }
            st[row * 16 + j] = (j < T && j >= nv) ? -INFINITY : d;
        }
    }
    __syncthreads();
    if (tid < nrows) {                             // softmax over the T + 2 keys of a row (the two extra keys are never masked)
        float s[LADIFF_MAX_LATENTS + 2];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < LADIFF_MAX_LATENTS + 2; ++j) {
            s[j] = j < nkeys ? st[tid * 16 + j] : -INFINITY;
            m = fmaxf(m, s[j]);
        }
        float l = 0.f;
#pragma unroll
        for (int j = 0; j < LADIFF_MAX_LATENTS + 2; ++j) { s[j] = j < nkeys ? expf(s[j] - m) : 0.f; l += s[j]; }
        const float inv = 1.f / l;
#pragma unroll
        for (int j = 0; j < LADIFF_MAX_LATENTS + 2; ++j)
            if (j < nkeys) st[tid * 16 + j] = s[j] * inv;
    }
    __syncthreads();
    for (int u = tid; u < nrows * 16; u += 512) {  // one thread per (row, 4 columns): o = sum_j p_j v_j
        const int row = u >> 4, c4 = (u & 15) * 4;
        const int sx = row / T;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < nkeys; ++j) {
            const float pj = st[row * 16 + j];
            const float* vp = j < T ? qt + (sx * T + j) * QLD + 128 : (j == T ? xt + sx * 128 + 64 : xt + ns * 128 + 64);
            const f32x4 v = ld4(vp + c4);
            o[0] = fmaf(pj, v[0], o[0]); o[1] = fmaf(pj, v[1], o[1]); o[2] = fmaf(pj, v[2], o[2]); o[3] = fmaf(pj, v[3], o[3]);
        }
        store_split4(p.out + (size_t)(row0 + row) * D, h * 64 + c4, o);
    }
}

int launch_qkv_attention(const float* x, const float* w, const float* bias, const float* text_kv, const float* tables,
                         int kv_off, int step_stride, const int32_t* d_step, const int32_t* counts, int Bs, int b_off,
                         int b_n, int T, float* out, hipStream_t s) {
    if (T < 1 || T > LADIFF_MAX_LATENTS) return LADIFF_ERR_SHAPE;
    if (b_n == 0) return 0;
    QkvAttnArgs a;
    a.x = x; a.w = w; a.bias = bias; a.text_kv = text_kv; a.tables = tables; a.d_step = d_step; a.counts = counts; a.out = out;
    a.kv_off = kv_off; a.step_stride = step_stride; a.Bs = Bs; a.b_off = b_off; a.b_n = b_n; a.T = T;
    // samples per workgroup: enough workgroups to cover the chip (every workgroup streams its head's 192 KiB of weights,
    // so fewer samples per workgroup cost no extra fill per CU), at most BM / T rows and QA_MAXR extra K|V slots
    int R = (b_n * H + 255) / 256;
    if (R < 1) R = 1;
    const int bm = R * T <= 32 ? 32 : 48;
    if (R * T > bm) R = bm / T;
    if (R > QA_MAXR) R = QA_MAXR;
    a.R = R;
    const dim3 grid((b_n + R - 1) / R, H);
    if (bm == 32) hipLaunchKernelGGL(qkv_attn_kernel<32>, grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL(qkv_attn_kernel<48>, grid, dim3(512), 0, s, a);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

}  // namespace ladiff
