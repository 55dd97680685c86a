// CLIP ViT-L/14 text tower -> pooled, projected prompt embedding [B, 768]  (the caller side of the path).
// Replaces MldTextEncoder.forward, mld_clip.py:51-86 ("clip" branch): text_model.get_text_features(input_ids) of the
// third-party `transformers` CLIPModel - token + position embedding, 12 pre-LN transformer layers (12 heads of 64,
// quick_gelu MLP 768 -> 3072 -> 768) under a causal mask only (the reference passes no attention_mask), final LayerNorm,
// the hidden state at the EOS position (argmax of the ids = first 49407) and text_projection (no bias).
//
// Exact shortcut: under a causal mask the EOS row depends only on rows <= EOS, so only the first `L` positions are
// evaluated (L >= 1 + max EOS index over the batch, chosen by the host from the token ids); rows behind are never built.
// Rows are batch-major (row = b * L + t) - or RAGGED (round 5): prompt b owns rows [row_off[b], row_off[b + 1]) = its OWN positions
// 0 .. eos_b (the same argument per prompt: nothing behind a prompt's EOS reaches its pooled row), 2.3x fewer rows than the padded
// batch for prompts of 1 .. 30 words; q / k / v are one batched GEMM launch.  Large GEMMs go through launch_gemm (fp32 MFMA, or f16x3 with S-format
// operands when a split weight table is given); the attention core is the decoder's MFMA kernel with 12 heads + causal.
#include "model.h"
#include "gemm_kr.h"

namespace ladiff {

constexpr int CW = LADIFF_TEXT_DIM;        // 768
constexpr int CFF = 4 * CW;                // 3072
constexpr int CH = CW / 64;                // 12 heads
constexpr int CNV = CW / 256;              // f32x4 per lane in a one-wave row

const std::vector<std::string>& clip_param_names() {
    static const std::vector<std::string> names = [] {
        std::vector<std::string> v;
        v.push_back("text_model.embeddings.token_embedding.weight");
        v.push_back("text_model.embeddings.position_embedding.weight");
        v.push_back("text_model.final_layer_norm.weight"); v.push_back("text_model.final_layer_norm.bias");
        v.push_back("text_projection.weight");
        for (int l = 0; l < CLIP_MAX_LAYERS; ++l) {
            const std::string p = "text_model.encoder.layers." + std::to_string(l) + ".";
            for (const char* m : {"layer_norm1", "self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.out_proj",
                                  "layer_norm2", "mlp.fc1", "mlp.fc2"}) {
                v.push_back(p + m + ".weight"); v.push_back(p + m + ".bias");
            }
        }
        return v;
    }();
    return names;
}

// ---- row kernels: one wave per 768-wide row, 3 x 16 bytes per lane
__device__ __forceinline__ void wide_stats(const f32x4 (&v)[CNV], float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CNV; ++j) s += v[j][0] + v[j][1] + v[j][2] + v[j][3];
    mean = wave_sum(s) * (1.f / CW);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < CNV; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float d = v[j][i] - mean; q += d * d; }
    rstd = rsqrtf(wave_sum(q) * (1.f / CW) + LN_EPS);
}

// x[b*L+t] = token_embedding[ids[b*S+t]] + position_embedding[t]          (CLIPTextEmbeddings.forward)
__global__ __launch_bounds__(256) void clip_embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ tok,
                                                         const float* __restrict__ pos, int vocab, int B, int M, int S, int L,
                                                         const int32_t* __restrict__ row_seq, const int32_t* __restrict__ row_off,
                                                         float* __restrict__ x) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    int b, t;
    if (row_seq != nullptr) { b = row_seq[row]; b = b < 0 ? 0 : (b >= B ? B - 1 : b); t = row - row_off[b]; }      // ragged rows
    else { b = row / L; t = row - b * L; }
    t = t < 0 ? 0 : (t >= L ? L - 1 : t);      // the three ragged arrays come from the caller: inconsistent ones must not read outside ids / pos (L <= S <= 77)
    long long id = ids[(size_t)b * S + t];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);          // ids are validated on the host; clamp for memory safety only
    const int c = (threadIdx.x & 63) * 4;
#pragma unroll
    for (int j = 0; j < CNV; ++j) {
        const f32x4 a = ld4(tok + (size_t)id * CW + j * 256 + c), p = ld4(pos + (size_t)t * CW + j * 256 + c);
        st4(x + (size_t)row * CW + j * 256 + c, f32x4{a[0] + p[0], a[1] + p[1], a[2] + p[2], a[3] + p[3]});
    }
}

// eos[b] = first position of the largest id among the L evaluated ones     (input_ids.argmax(dim=-1), pooled output)
__global__ __launch_bounds__(64) void clip_eos_kernel(const int64_t* __restrict__ ids, int S, int L, int32_t* __restrict__ eos) {
    const int b = blockIdx.x, lane = threadIdx.x;
    long long best = -0x7fffffffffffffffLL - 1; int at = 0x7fffffff;
    for (int t = lane; t < L; t += 64) {
        const long long v = ids[(size_t)b * S + t];
        if (v > best) { best = v; at = t; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const long long ob = __shfl_xor(best, o, 64); const int oa = __shfl_xor(at, o, 64);
        if (ob > best || (ob == best && oa < at)) { best = ob; at = oa; }
    }
    if (lane == 0) eos[b] = at;
}

// y[r] = LayerNorm(x[src(r)]) with src(r) = r, or r * L + eos[r] + adj when `eos` is given (the pooled row; ragged rows: L = 0,
// eos = row_off + 1, adj = -1: the prompt's last row);  fp32 and / or S-format
__global__ __launch_bounds__(256) void clip_ln_kernel(const float* __restrict__ x, const int32_t* __restrict__ eos, int L, int adj,
                                                      const float* __restrict__ g, const float* __restrict__ bt, int M,
                                                      float* __restrict__ y, float* __restrict__ ys) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const size_t src = eos != nullptr ? (size_t)((long long)row * L + eos[row] + adj) : (size_t)row;
    const int c = (threadIdx.x & 63) * 4;
    f32x4 v[CNV];
#pragma unroll
    for (int j = 0; j < CNV; ++j) v[j] = ld4(x + src * CW + j * 256 + c);
    float mean, rstd;
    wide_stats(v, mean, rstd);
#pragma unroll
    for (int j = 0; j < CNV; ++j) {
        const f32x4 gg = ld4(g + j * 256 + c), bb = ld4(bt + j * 256 + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[j][i] = (v[j][i] - mean) * rstd * gg[i] + bb[i];
        if (y != nullptr) st4(y + (size_t)row * CW + j * 256 + c, v[j]);
        if (ys != nullptr) store_split4(ys + (size_t)row * CW, j * 256 + c, v[j]);
    }
}

static int ln_rows(const float* x, const int32_t* eos, int L, const NormW& n, int M, float* y, float* ys, hipStream_t s, int adj = 0) {
    hipLaunchKernelGGL(clip_ln_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, eos, L, adj, n.g, n.b, M, y, ys);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// Few rows (a demo.py call: the empty prompt + one prompt = ~40 rows; round 6): the 128-row tiles of the large-M GEMM leave N / 128 =
// 6 .. 24 workgroups, each walking the whole K - 45 us per GEMM whatever its size, 1.9 of a single prompt's 10 ms (profiles/r6/11_*).
// Up to CLIP_SMALL_ROWS rows the split mode's GEMMs run on the denoiser's K-resident 64x64 tiles instead (gemm_kr.hip: one 256-wide
// K slice per workgroup, K / 256 = 3 or 12 partial planes, 36 .. 144 workgroups that each move 64 KB) and a row pass sums the
// planes and applies bias / quick_gelu / residual: the same S-format operands and split products, summed per K slice.
constexpr int CLIP_SMALL_ROWS = 256;
constexpr int CLIP_PLANE_COLS = 12 * CW;       // floats per row of the partial planes: max over the GEMMs of (K / 256) * N = 3 * 3072 = 12 * 768 = 9216

// Many rows: fc2 (K = 3072, N = 768) is 6 column tiles x a few dozen row tiles - about one workgroup per CU, each walking 96 K stages with
// ONE stage in flight (70 us at 2,260 rows, most of it load latency nobody hides).  Its K range is cut in CLIP_FC2_KPARTS parts over
// blockIdx.y (GemmArgs::ksplit: four times the workgroups, two per CU hiding each other's stages) and the row pass sums the planes.
constexpr int CLIP_FC2_KPARTS = 4;
constexpr int CLIP_FC2_KPARTS_MAX_ROWS = 8192;  // beyond, the tiles alone fill the chip
static size_t clip_plane_floats(int M) {
    if (M <= CLIP_SMALL_ROWS) return (size_t)M * CLIP_PLANE_COLS;
    return M <= CLIP_FC2_KPARTS_MAX_ROWS ? (size_t)M * CLIP_FC2_KPARTS * CW : 0;
}
size_t clip_ws_floats_rows(int B, int M) {
    return (size_t)M * (2 * CW /*x ping-pong*/ + CW /*h*/ + 3 * CW /*qkv*/ + CW /*att*/ + CFF /*mlp*/) + (size_t)B * CW + (size_t)B + 64 +
           clip_plane_floats(M);
}
size_t clip_ws_floats(int B, int L) { return clip_ws_floats_rows(B, B * L); }

// y = act(sum_k planes[k] + bias) + res over [M, N] (N % 4 == 0): fp32 and / or S-format; planes [np][M][ld]
__global__ __launch_bounds__(256) void clip_reduce_kernel(const float* __restrict__ planes, int np, size_t plane, int ld, int M, int N,
                                                          const float* __restrict__ bias, int act, const float* __restrict__ res, int ldres,
                                                          float* __restrict__ y, float* __restrict__ ys, int ldy) {
    const int n4 = N / 4;
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (size_t)M * n4) return;
    const int row = (int)(id / n4), c = 4 * (int)(id % n4);
    f32x4 v = ld4(planes + (size_t)row * ld + c);
    for (int k = 1; k < np; ++k) {
        const f32x4 t = ld4(planes + k * plane + (size_t)row * ld + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += t[i];
    }
    const f32x4 b = ld4(bias + c);
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    if (res != nullptr) r = ld4(res + (size_t)row * ldres + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = act_apply(v[i] + b[i], act) + r[i];
    if (y != nullptr) st4(y + (size_t)row * ldy + c, v);
    if (ys != nullptr) store_split4(ys + (size_t)row * ldy, c, v);
}

// seq_len / row_off / row_seq (all or none): ragged rows - prompt b has seq_len[b] = 1 + (its EOS position) rows from row_off[b]
// (row_off has B + 1 entries, row_off[B] = total_rows), row_seq[r] = the prompt of row r; L then only bounds the lengths.
int clip_text_encode(const ClipW& w, const ClipW* wsp, int n_layers, int vocab, const int64_t* ids, int B, int S, int L,
                     float* out, float* ws, size_t ws_floats, hipStream_t s, const int32_t* seq_len, const int32_t* row_off,
                     const int32_t* row_seq, int total_rows) {
    if (n_layers < 1 || n_layers > CLIP_MAX_LAYERS || S < 1 || S > CLIP_MAX_POSITIONS || L < 1 || L > S || vocab < 1)
        return LADIFF_ERR_SHAPE;
    const bool ragged = row_off != nullptr;
    if (ragged && (seq_len == nullptr || row_seq == nullptr || total_rows < B || (long long)total_rows > (long long)B * L)) return LADIFF_ERR_ARG;
    if (ws_floats < (ragged ? clip_ws_floats_rows(B, total_rows) : clip_ws_floats(B, L))) return LADIFF_ERR_WORKSPACE;
    if (B == 0) return 0;
    const int M = ragged ? total_rows : B * L;
    const bool sp = wsp != nullptr;
    float* p = ws;
    float* x = p; p += (size_t)M * CW;          // residual stream, ping-pong with x2 across the two sub-blocks
    float* x2 = p; p += (size_t)M * CW;
    float* h = p; p += (size_t)M * CW;          // LN output (S-format in split mode)
    float* qkv = p; p += (size_t)M * 3 * CW;
    float* att = p; p += (size_t)M * CW;
    float* mlp = p; p += (size_t)M * CFF;
    const bool small = sp && M <= CLIP_SMALL_ROWS;
    float* planes = p; p += clip_plane_floats(M);   // partial planes: [K / 256][M][ld] of the small-row path, [CLIP_FC2_KPARTS][M][768] of fc2 otherwise (16-byte aligned: every size so far is a multiple of 768 floats)
    float* pooled = p; p += (size_t)B * CW;
    int32_t* eos = reinterpret_cast<int32_t*>(p);

    // small-row path: S-format A [M, K] x S-format W [N, K] -> K / 256 planes (row stride ld, written at column col0) ...
    auto kr_planes = [&](const float* A, int K, const float* Wsp, int N, int ld, int col0) -> int {
        KrArgs g;
        g.A = A; g.lda = K; g.W = Wsp; g.ldw = K; g.Y = planes + col0; g.ldy = ld; g.M = M; g.N = N; g.K = K; g.split = 1;
        if (K == 256) return LADIFF_ERR_SHAPE;   // (a single slice would apply the epilogue itself: not a shape of this tower)
        return launch_gemm_kr(g, s);
    };
    // ... and their sum + bias, activation, residual
    auto reduce = [&](int K, int N, int ld, const float* bias, int act, const float* res, int ldres, float* Y, float* Ys, int ldy) -> int {
        const size_t n = (size_t)M * (N / 4);
        hipLaunchKernelGGL(clip_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, planes, K / 256, (size_t)M * ld, ld, M, N, bias, act,
                           res, ldres, Y, Ys, ldy);
        LADIFF_LAUNCH_CHECK();
        return 0;
    };
    auto gemm = [&](const float* A, int K, const LinearW& l, const LinearW& ls, float* Y, int ldy, int N, int act,
                    const float* res, bool split_out) -> int {
        if (small) {
            LADIFF_TRY(kr_planes(A, K, ls.w, N, N, 0));
            return reduce(K, N, N, l.b, act, res, ldy, split_out ? nullptr : Y, split_out ? Y : nullptr, ldy);
        }
        GemmArgs g;
        g.A = A; g.lda = K; g.W = sp ? ls.w : l.w; g.ldw = K; g.bias = l.b; g.M = M; g.N = N; g.K = K; g.act = act; g.ldy = ldy;
        g.res = res; g.ldres = ldy; g.split = sp ? 1 : 0;
        if (sp && split_out) g.Ys = Y; else g.Y = Y;
        return launch_gemm(g, s);
    };

    hipLaunchKernelGGL(clip_embed_kernel, dim3((M + 3) / 4), dim3(256), 0, s, ids, w.tok, w.pos, vocab, B, M, S, L, row_seq, row_off, x);
    LADIFF_LAUNCH_CHECK();
    if (!ragged) {
        hipLaunchKernelGGL(clip_eos_kernel, dim3(B), dim3(64), 0, s, ids, S, L, eos);
        LADIFF_LAUNCH_CHECK();
    }

    for (int l = 0; l < n_layers; ++l) {                       // CLIPEncoderLayer.forward (pre-LN residual blocks)
        const ClipLayerW& W = w.layer[l];
        const ClipLayerW& Ws = sp ? wsp->layer[l] : w.layer[l];
        LADIFF_TRY(ln_rows(x, nullptr, 0, W.ln1, M, sp ? nullptr : h, sp ? h : nullptr, s));
        // q | k | v packed by columns; the 1/sqrt(64) query scale is applied (exactly) inside the attention kernel.  One batched
        // launch of the three same-shape products (each alone leaves half of the chip idle at a few thousand rows)
        if (small) {                             // q | k | v: three launches into one set of planes [3][M][2304], one row pass (the biases differ per part)
            const LinearW* lw[3] = {&W.q, &W.k, &W.v};
            const LinearW* lws[3] = {&Ws.q, &Ws.k, &Ws.v};
            for (int i = 0; i < 3; ++i) LADIFF_TRY(kr_planes(h, CW, lws[i]->w, CW, 3 * CW, i * CW));
            for (int i = 0; i < 3; ++i) {
                const size_t n = (size_t)M * (CW / 4);
                hipLaunchKernelGGL(clip_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, planes + i * CW, CW / 256, (size_t)M * 3 * CW,
                                   3 * CW, M, CW, lw[i]->b, (int)ACT_NONE, (const float*)nullptr, 0, qkv + i * CW, (float*)nullptr, 3 * CW);
                LADIFF_LAUNCH_CHECK();
            }
        } else {
            GemmArgs g3[3];
            const LinearW* lw[3] = {&W.q, &W.k, &W.v};
            const LinearW* lws[3] = {&Ws.q, &Ws.k, &Ws.v};
            for (int i = 0; i < 3; ++i) {
                GemmArgs& g = g3[i];
                g.A = h; g.lda = CW; g.W = sp ? lws[i]->w : lw[i]->w; g.ldw = CW; g.bias = lw[i]->b; g.M = M; g.N = CW; g.K = CW; g.ldy = 3 * CW;
                g.Y = qkv + i * CW; g.split = sp ? 1 : 0;
            }
            LADIFF_TRY(launch_gemm_batch(g3, 3, s));
        }
        if (sp) LADIFF_TRY(launch_self_attention_split(qkv, seq_len, nullptr, att, B, L, CH, 1, 1, s, row_off, 0));
        else LADIFF_TRY(launch_self_attention(qkv, seq_len, nullptr, att, B, L, CH, 1, 0, s, row_off));
        LADIFF_TRY(gemm(att, CW, W.o, Ws.o, x2, CW, CW, ACT_NONE, x, false));             // x2 = x + out_proj(attn)
        LADIFF_TRY(ln_rows(x2, nullptr, 0, W.ln2, M, sp ? nullptr : h, sp ? h : nullptr, s));
        LADIFF_TRY(gemm(h, CW, W.fc1, Ws.fc1, mlp, CFF, CFF, ACT_QGELU, nullptr, true));  // quick_gelu(fc1)
        if (sp && !small && M <= CLIP_FC2_KPARTS_MAX_ROWS) {                             // x = x2 + fc2(...), fc2 in K parts
            GemmArgs g;
            g.A = mlp; g.lda = CFF; g.W = Ws.fc2.w; g.ldw = CFF; g.M = M; g.N = CW; g.K = CFF; g.ldy = CW; g.split = 1;
            g.Y = planes; g.ksplit = CLIP_FC2_KPARTS; g.plane = (size_t)M * CW;
            LADIFF_TRY(launch_gemm(g, s));
            const size_t n = (size_t)M * (CW / 4);
            hipLaunchKernelGGL(clip_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, planes, CLIP_FC2_KPARTS, (size_t)M * CW, CW, M, CW,
                               W.fc2.b, (int)ACT_NONE, (const float*)x2, CW, x, (float*)nullptr, CW);
            LADIFF_LAUNCH_CHECK();
        } else
        LADIFF_TRY(gemm(mlp, CFF, W.fc2, Ws.fc2, x, CW, CW, ACT_NONE, x2, false));        // x = x2 + fc2(...)
    }
    // pooled = final_layer_norm(x)[b, eos[b]];  text_embeds = text_projection(pooled)   (fp32: B rows only)
    if (ragged) LADIFF_TRY(ln_rows(x, row_off + 1, 0, w.final_ln, B, pooled, nullptr, s, -1));     // the prompt's last row = its EOS position
    else LADIFF_TRY(ln_rows(x, eos, L, w.final_ln, B, pooled, nullptr, s));
    GemmArgs g;
    g.A = pooled; g.lda = CW; g.W = w.proj; g.ldw = CW; g.Y = out; g.ldy = CW; g.M = B; g.N = CW; g.K = CW;
    return launch_gemm(g, s);
}

}  // namespace ladiff
