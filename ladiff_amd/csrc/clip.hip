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

size_t clip_ws_floats(int B, int L) {
    const size_t M = (size_t)B * L;
    return M * (2 * CW /*x ping-pong*/ + CW /*h*/ + 3 * CW /*qkv*/ + CW /*att*/ + CFF /*mlp*/) + (size_t)B * CW + (size_t)B + 64;
}

size_t clip_ws_floats_rows(int B, int M) {
    return (size_t)M * (2 * CW + CW + 3 * CW + CW + CFF) + (size_t)B * CW + (size_t)B + 64;
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
    float* pooled = p; p += (size_t)B * CW;
    int32_t* eos = reinterpret_cast<int32_t*>(p);

    auto gemm = [&](const float* A, int K, const LinearW& l, const LinearW& ls, float* Y, int ldy, int N, int act,
                    const float* res, bool split_out) -> int {
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
        {
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
