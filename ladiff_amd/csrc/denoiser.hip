// Host-side sequencing of the length-aware denoiser (LADiffDenoiser.forward, ladiff_denoiser.py:153-295)
// on top of the gfx950 kernels.  Everything is enqueued on the caller's stream; no allocation, no sync.
#include "model.h"

namespace ladiff {

static std::vector<std::string> block_prefixes(const char* root) {
    std::vector<std::string> v;
    for (int i = 0; i < NSKIP; ++i) v.push_back(std::string(root) + ".input_blocks." + std::to_string(i));
    v.push_back(std::string(root) + ".middle_block");
    for (int i = 0; i < NSKIP; ++i) v.push_back(std::string(root) + ".output_blocks." + std::to_string(i));
    return v;
}
static void wb(std::vector<std::string>& v, const std::string& p) { v.push_back(p + ".weight"); v.push_back(p + ".bias"); }
static void mha_names(std::vector<std::string>& v, const std::string& p) {
    v.push_back(p + ".in_proj_weight"); v.push_back(p + ".in_proj_bias"); wb(v, p + ".out_proj");
}
static void styl_names(std::vector<std::string>& v, const std::string& p) {
    wb(v, p + ".emb_layers.1"); wb(v, p + ".norm"); wb(v, p + ".out_layers.2");
}

const std::vector<std::string>& denoiser_param_names() {
    static const std::vector<std::string> names = [] {
        std::vector<std::string> v;
        wb(v, "time_embedding.linear_1"); wb(v, "time_embedding.linear_2"); wb(v, "emb_proj.1");
        v.push_back("query_pos.pe"); v.push_back("mem_pos.pe");
        wb(v, "encoder.norm");
        for (const auto& p : block_prefixes("encoder")) {
            wb(v, p + ".ca_block.norm"); wb(v, p + ".ca_block.text_norm");
            wb(v, p + ".ca_block.query"); wb(v, p + ".ca_block.key"); wb(v, p + ".ca_block.value");
            styl_names(v, p + ".ca_block.proj_out");
            wb(v, p + ".ffn.linear1"); wb(v, p + ".ffn.linear2");
            styl_names(v, p + ".ffn.proj_out");
            mha_names(v, p + ".sa_block.self_attn");
            wb(v, p + ".sa_block.linear1"); wb(v, p + ".sa_block.linear2");
            wb(v, p + ".sa_block.norm1"); wb(v, p + ".sa_block.norm2");
        }
        for (int i = 0; i < NSKIP; ++i) wb(v, "encoder.linear_blocks." + std::to_string(i));
        return v;
    }();
    return names;
}

const std::vector<std::string>& decoder_param_names() {
    static const std::vector<std::string> names = [] {
        std::vector<std::string> v;
        v.push_back("query_pos_decoder.pe");
        for (const auto& p : block_prefixes("decoder")) {
            mha_names(v, p + ".self_attn"); mha_names(v, p + ".multihead_attn");
            wb(v, p + ".linear1"); wb(v, p + ".linear2");
            wb(v, p + ".norm1"); wb(v, p + ".norm2"); wb(v, p + ".norm3");
        }
        for (int i = 0; i < NSKIP; ++i) wb(v, "decoder.linear_blocks." + std::to_string(i));
        wb(v, "decoder.norm"); wb(v, "final_layer");
        return v;
    }();
    return names;
}

const std::vector<std::string>& encoder_param_names() {
    static const std::vector<std::string> names = [] {
        std::vector<std::string> v;
        v.push_back("global_motion_token"); v.push_back("query_pos_encoder.pe");
        for (const auto& p : block_prefixes("encoder")) {
            mha_names(v, p + ".self_attn");
            wb(v, p + ".linear1"); wb(v, p + ".linear2");
            wb(v, p + ".norm1"); wb(v, p + ".norm2");
        }
        for (int i = 0; i < NSKIP; ++i) wb(v, "encoder.linear_blocks." + std::to_string(i));
        wb(v, "encoder.norm"); wb(v, "skel_embedding");
        return v;
    }();
    return names;
}

// ------------------------------------------------------------------ small GEMM helper
static GemmArgs lin(const float* A, int lda, const LinearW& l, float* Y, int ldy, int M, int N, int K, int act = ACT_NONE) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = l.w; g.ldw = K; g.bias = l.b; g.Y = Y; g.ldy = ldy; g.M = M; g.N = N; g.K = K; g.act = act;
    return g;
}

// ------------------------------------------------------------------ time tables
size_t den_tables_floats(int n_steps) { return (size_t)n_steps * DEN_STEP_STRIDE; }
static size_t time_ws_floats(int n) { return (size_t)n * D * 3; }

int denoiser_time_tables(const DenoiserW& w, const float* sinus, int n, float* tables, float* ws, size_t ws_floats,
                         hipStream_t s) {
    if (ws_floats < time_ws_floats(n)) return LADIFF_ERR_WORKSPACE;
    float* h1 = ws;                 // SiLU(linear_1(sinusoid))       tools/embeddings.py:296-301
    float* temb = h1 + (size_t)n * D;   // time_emb                     :303
    float* semb = temb + (size_t)n * D; // SiLU(time_emb), input of every StylizationBlock.emb_layers
    LADIFF_TRY(launch_gemm(lin(sinus, TEXT_DIM, w.time1, h1, D, n, D, TEXT_DIM, ACT_SILU), s));
    LADIFF_TRY(launch_gemm(lin(h1, D, w.time2, temb, D, n, D, D), s));
    LADIFF_TRY(launch_silu(temb, semb, (size_t)n * D, s));
    for (int l = 0; l < NL; ++l) {
        const DenLayerW& L = w.layer[l];
        float* base = tables + (size_t)l * DEN_LAYER_STRIDE;
        LADIFF_TRY(launch_gemm(lin(semb, D, L.ca_proj.emb, base + DEN_OFF_CA_MOD, DEN_STEP_STRIDE, n, 2 * D, D), s));
        LADIFF_TRY(launch_gemm(lin(semb, D, L.ffn_proj.emb, base + DEN_OFF_FFN_MOD, DEN_STEP_STRIDE, n, 2 * D, D), s));
        // K | V of the time token = rows [256, 768) of the packed in_proj applied to time_emb
        LinearW kvw{L.sa_attn.in_w + (size_t)D * D, L.sa_attn.in_b + D};
        LADIFF_TRY(launch_gemm(lin(temb, D, kvw, base + DEN_OFF_TIME_KV, DEN_STEP_STRIDE, n, 2 * D, D), s));
    }
    return 0;
}

// ------------------------------------------------------------------ text cache
//   [B2,256]              emb_proj(text)                      (ladiff_denoiser.py:198)
//   [9][B2,512]           K | V of the text token per layer   (sa_block in_proj rows 256..767)
//   [9][n][B2+1,256]      c table: the whole ca_block delta   (mdiff_transformer.py:219-247 with ONE text token)
// With one text token softmax(key) over the token axis is exactly 1 and sum_d softmax(query)_d = 1, so the attention
// output of a valid latent row is the value vector v_b of its sample and of a padded row is 0; the block then adds
//   c[step, layer, b]   = out( SiLU( LN(v_b) * (1 + scale) + shift ) )     for valid rows
//   c[step, layer, pad] = out( SiLU( beta    * (1 + scale) + shift ) )     for padded rows (LN(0) = beta)
// which depends on (step, layer, sample) only - never on the latents - so it is computed once per call for every
// step instead of 9 x n_steps times inside the loop.
// ntxt > 1 (general-N conditioning, linear_ca.hip): [B2 N,256] projected tokens, [9][B2 N,512] K|V of the text tokens,
// [9][B2][4][64][64] the cross-attention's key^T value matrices (text only, so step-invariant) instead of the c table
size_t den_text_cache_floats(int B2, int n, int ntxt) {
    if (ntxt > 1) return (size_t)B2 * ntxt * D + (size_t)NL * B2 * ntxt * 2 * D + (size_t)NL * B2 * H * DH * DH;
    return (size_t)B2 * D + (size_t)NL * B2 * 2 * D + (size_t)NL * B2 * D + (size_t)NL * n * (B2 + 1) * D;
}
size_t den_text_ws_floats(int B2, int n, int ntxt) {
    if (ntxt > 1) return (size_t)B2 * ntxt * (TEXT_DIM + 3 * D);
    // relu(text) | per-layer LN(text projection) | c-table inputs of all layers (the per-layer launches are batched)
    return (size_t)B2 * TEXT_DIM + (size_t)NL * B2 * D + (size_t)NL * n * (B2 + 1) * D;
}

static int denoiser_text_cache_general(const DenoiserW& w, const float* text, int B2, int N, float* cache, float* ws, size_t ws_floats,
                                       hipStream_t s) {
    const int R = B2 * N;
    if (ws_floats < den_text_ws_floats(B2, 1, N)) return LADIFF_ERR_WORKSPACE;
    float* rl = ws;
    float* tn = rl + (size_t)R * TEXT_DIM;
    float* key = tn + (size_t)R * D;
    float* val = key + (size_t)R * D;
    float* tproj = cache;
    float* tkv = cache + (size_t)R * D;
    float* catt = tkv + (size_t)NL * R * 2 * D;
    LADIFF_TRY(launch_relu(text, rl, (size_t)R * TEXT_DIM, s));
    LADIFF_TRY(launch_gemm(lin(rl, TEXT_DIM, w.emb_proj, tproj, D, R, D, TEXT_DIM), s));          // ladiff_denoiser.py:198
    for (int l = 0; l < NL; ++l) {
        const DenLayerW& L = w.layer[l];
        LinearW kvw{L.sa_attn.in_w + (size_t)D * D, L.sa_attn.in_b + D};
        LADIFF_TRY(launch_gemm(lin(tproj, D, kvw, tkv + (size_t)l * R * 2 * D, 2 * D, R, 2 * D, D), s));
        LADIFF_TRY(launch_layernorm(tproj, L.ca_text_norm.g, L.ca_text_norm.b, tn, R, s));            // mdiff_transformer.py:233
        LADIFF_TRY(launch_gemm(lin(tn, D, L.ca_key, key, D, R, D, D), s));
        LADIFF_TRY(launch_gemm(lin(tn, D, L.ca_value, val, D, R, D, D), s));                           // :237
        LADIFF_TRY(launch_lca_kv(key, val, B2, N, catt + (size_t)l * B2 * H * DH * DH, s));            // :235, :239
    }
    return 0;
}

// cache layout, one text token: [B2,256] emb_proj | [9][B2,512] text K|V | [9][B2,256] LN(value) rows | [9][n][B2+1,256] c table
const float* den_cache_tkv(const float* cache, int B2, int ntxt) { return cache + (size_t)B2 * ntxt * D; }
const float* den_cache_ctab(const float* cache, int B2, int ntxt) {
    const float* p = den_cache_tkv(cache, B2, ntxt) + (size_t)NL * B2 * ntxt * 2 * D;
    return ntxt > 1 ? p : p + (size_t)NL * B2 * D;
}

// the part of the text cache that does not depend on the step: once per call
int denoiser_text_static(const DenoiserW& w, const float* text, int B2, float* cache, float* ws, size_t ws_floats, hipStream_t s) {
    if (ws_floats < (size_t)B2 * (TEXT_DIM + NL * D)) return LADIFF_ERR_WORKSPACE;
    float* rl = ws;
    float* tn = rl + (size_t)B2 * TEXT_DIM;                  // [NL][B2][256]
    float* tproj = cache;
    float* tkv = cache + (size_t)B2 * D;
    float* nval = tkv + (size_t)NL * B2 * 2 * D;
    LADIFF_TRY(launch_relu(text, rl, (size_t)B2 * TEXT_DIM, s));
    LADIFF_TRY(launch_gemm(lin(rl, TEXT_DIM, w.emb_proj, tproj, D, B2, D, TEXT_DIM), s));
    // the nine layers' projections of the text token are independent and tiny: one launch per kind, not per layer
    GemmArgs kv[NL], val[NL];
    RowBatch ln;
    for (int l = 0; l < NL; ++l) {
        const DenLayerW& L = w.layer[l];
        LinearW kvw{L.sa_attn.in_w + (size_t)D * D, L.sa_attn.in_b + D};
        kv[l] = lin(tproj, D, kvw, tkv + (size_t)l * B2 * 2 * D, 2 * D, B2, 2 * D, D);
        ln.a[l] = tproj; ln.g[l] = L.ca_text_norm.g; ln.b[l] = L.ca_text_norm.b; ln.y[l] = tn + (size_t)l * B2 * D;
        val[l] = lin(tn + (size_t)l * B2 * D, D, L.ca_value, nval + (size_t)l * B2 * D, D, B2, D, D);
        val[l].ln_g = L.ca_proj.norm.g; val[l].ln_b = L.ca_proj.norm.b;
    }
    LADIFF_TRY(launch_gemm_batch(kv, NL, s));
    LADIFF_TRY(launch_layernorm_batch(ln, NL, B2, s));
    return launch_gemm_batch(val, NL, s);
}

// c table rows of `n` consecutive steps; `tables_lo` = the time tables at the first of them.  Sampling loops with many steps
// build it one window at a time (ladiff_diffusion_reverse), so the table is O(window x B), not O(steps x B).
// wsp != NULL (f16x3 mode): the nine out-projections run as ONE batched f16x3 launch on S-format inputs (the table is 15 GFLOP per
// 50-step window at B = 128: 157 us on the fp32-input MFMA, profiles/r3/01)
int denoiser_ctab(const DenoiserW& w, const float* tables_lo, int n, float* cache, int B2, float* u, size_t u_floats, hipStream_t s,
                  const DenoiserW* wsp) {
    const int R = B2 + 1;
    if (u_floats < (size_t)n * R * D) return LADIFF_ERR_WORKSPACE;
    const float* nval = cache + (size_t)B2 * D + (size_t)NL * B2 * 2 * D;
    float* ctab = const_cast<float*>(den_cache_ctab(cache, B2, 1));
    if (u_floats >= (size_t)NL * n * R * D) {                // all layers at once: one input launch, one batched GEMM
        RowBatch rb;
        GemmArgs g[NL];
        for (int l = 0; l < NL; ++l) {
            const DenLayerW& L = w.layer[l];
            float* ul = u + (size_t)l * n * R * D;
            rb.a[l] = nval + (size_t)l * B2 * D; rb.g[l] = L.ca_proj.norm.b;
            rb.b[l] = tables_lo + (size_t)l * DEN_LAYER_STRIDE + DEN_OFF_CA_MOD; rb.y[l] = ul;
            g[l] = lin(ul, D, L.ca_proj.out, ctab + (size_t)l * n * R * D, D, n * R, D, D);
            if (wsp != nullptr) { g[l].W = wsp->layer[l].ca_proj.out.w; g[l].split = 1; }
        }
        LADIFF_TRY(launch_ca_table_input_batch(rb, NL, DEN_STEP_STRIDE, n, B2, s, wsp != nullptr ? 1 : 0));
        return launch_gemm_batch(g, NL, s);
    }
    for (int l = 0; l < NL; ++l) {                           // scratch for one layer only
        const DenLayerW& L = w.layer[l];
        LADIFF_TRY(launch_ca_table_input(nval + (size_t)l * B2 * D, L.ca_proj.norm.b, tables_lo + (size_t)l * DEN_LAYER_STRIDE + DEN_OFF_CA_MOD,
                                         DEN_STEP_STRIDE, n, B2, u, s));
        LADIFF_TRY(launch_gemm(lin(u, D, L.ca_proj.out, ctab + (size_t)l * n * R * D, D, n * R, D, D), s));
    }
    return 0;
}

int denoiser_text_cache(const DenoiserW& w, const float* text, int B2, const float* tables, int n, float* cache,
                        float* ws, size_t ws_floats, hipStream_t s, int ntxt) {
    if (ntxt > 1) return denoiser_text_cache_general(w, text, B2, ntxt, cache, ws, ws_floats, s);
    if (ws_floats < den_text_ws_floats(B2, n)) return LADIFF_ERR_WORKSPACE;
    LADIFF_TRY(denoiser_text_static(w, text, B2, cache, ws, ws_floats, s));
    float* u = ws + (size_t)B2 * (TEXT_DIM + NL * D);
    return denoiser_ctab(w, tables, n, cache, B2, u, ws_floats - (size_t)B2 * (TEXT_DIM + NL * D), s, nullptr);
}

// ------------------------------------------------------------------ one ca_block, literal (unit entry for the N > 1 path)
// out = x + StylizationBlock( softmax_d(query(LN x)) . sum_n softmax_n(key(LN_t xf)) value(LN_t xf)^T , emb )   :219-247
size_t linear_cross_attention_ws_floats(int B, int T, int N) {
    return (size_t)B * N * 3 * D + (size_t)B * H * DH * DH + (size_t)B * T * 2 * D + (size_t)B * 3 * D;
}
int linear_cross_attention(const DenoiserW& w, int layer, const float* x, const float* xf, const float* emb, const int32_t* counts,
                           int B, int T, int N, float* out, float* ws, size_t ws_floats, hipStream_t s) {
    if (layer < 0 || layer >= NL || T < 1 || T > LADIFF_MAX_LATENTS || N < 1) return LADIFF_ERR_SHAPE;
    if (ws_floats < linear_cross_attention_ws_floats(B, T, N)) return LADIFF_ERR_WORKSPACE;
    const DenLayerW& L = w.layer[layer];
    const int R = B * N, M = B * T;
    float* tn = ws; float* key = tn + (size_t)R * D; float* val = key + (size_t)R * D;
    float* catt = val + (size_t)R * D;
    float* xn = catt + (size_t)B * H * DH * DH; float* q = xn + (size_t)M * D;
    float* semb = q + (size_t)M * D; float* mod = semb + (size_t)B * D;
    LADIFF_TRY(launch_layernorm(xf, L.ca_text_norm.g, L.ca_text_norm.b, tn, R, s));
    LADIFF_TRY(launch_gemm(lin(tn, D, L.ca_key, key, D, R, D, D), s));
    LADIFF_TRY(launch_gemm(lin(tn, D, L.ca_value, val, D, R, D, D), s));
    LADIFF_TRY(launch_lca_kv(key, val, B, N, catt, s));
    LADIFF_TRY(launch_layernorm(x, L.ca_norm.g, L.ca_norm.b, xn, M, s));
    LADIFF_TRY(launch_gemm(lin(xn, D, L.ca_query, q, D, M, D, D), s));
    LADIFF_TRY(launch_silu(emb, semb, (size_t)B * D, s));                                             // emb_layers = SiLU, Linear  :141-144
    LADIFF_TRY(launch_gemm(lin(semb, D, L.ca_proj.emb, mod, 2 * D, B, 2 * D, D), s));
    LADIFF_TRY(launch_lca_apply(q, catt, counts, B, 0, B, T, mod, 0, 2 * D, nullptr, L.ca_proj.norm.g, L.ca_proj.norm.b, xn, s));
    GemmArgs g = lin(xn, D, L.ca_proj.out, out, D, M, D, D);
    g.res = x; g.ldres = D;
    return launch_gemm(g, s);
}

// ------------------------------------------------------------------ forward
// Per layer, fp32 mode (11 launches, 13 on the four output blocks); M = 2B*T rows:
//   [skip]  x   = linear_blocks([x | xs.pop()])      split-K 2 + combine                       cross_attention.py:79-82
//   qkv         = in_proj(x)                          N=768                                     mdiff_transformer.py:60-61
//   att         = softmax over [latents | text | time] keys . V                                 :296-313
//   R1          = x + out_proj(att)                   N=256 (the pre-norm1 sum)                 :62
//   X1          = LN1(R1)                             row kernel                                :63
//   hid         = relu(linear1(X1))                                                             :64
//   part        = linear2(hid)                        K=1024 as split-K 4                       :64
//   X3          = LN2(X1 + sum part + b) + c          combine kernel; c = hoisted ca_block      :65-66, :219-247
//   hid         = gelu(ffn.linear1(X3))                                                         :260
//   part        = ffn.linear2(hid)                    split-K 4
//   u           = SiLU(LN(sum part + b) * (1 + scale_t) + shift_t)   combine kernel            :152-162
//   x'          = X3 + out_layers(u)                  N=256                                     :162, :261
// f16x3 mode (8 launches, 9 on the output blocks): the same arithmetic with the steps grouped into fused kernels -
//   [skip] gemm_rowln (concat GEMM + bias) | qkv_attn (qkv + att) | gemm_rowln (R1, X1) | linear1 | linear2 split-K |
//   reduce_rows (X3) | ffn.linear1 | ffn.linear2 split-K | combine_gemm (u, x').
size_t den_forward_ws_floats(int B2, int T) { return (size_t)B2 * T * (16 * D + 3 * D + D + FF + 4 * D); }

void den_loop_io(float* ws, int rows, float** x, float** xs) {
    *x = ws;                                   // P[0]
    *xs = ws + (size_t)8 * rows * D;           // Ps[0]: after P[0..3] and SK[0..3]
}

static KrArgs kr(const float* A, int lda, const float* W, const float* b, float* Y, int ldy, int M, int N, int K, int act = ACT_NONE) {
    KrArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = K; g.bias = b; g.Y = Y; g.ldy = ldy; g.M = M; g.N = N; g.K = K; g.act = act;
    return g;
}

// Processes samples [b_lo, b_lo + b_n) of the duplicated batch of B2 = Bs * dup samples (tables / caches are sized for
// B2); independent sample ranges can run concurrently on different streams with disjoint workspaces.
// den_loop_io() exposes the buffer that holds both the network input and the last layer's output (loop_mode).
//
// ws != nullptr selects the f16x3 matrix path: `ws` holds the S-format copies of the weight matrices (same table order
// as `w`), GEMM operands travel in S-format (every tensor that is both a GEMM operand and a residual is written twice:
// fp32 for the residual / LayerNorm consumers, S-format for the MFMA), accumulation and everything else stay fp32.
int denoiser_forward(const DenoiserW& w, const DenoiserW* wsp, const float* tables, const int32_t* d_step, const float* cache,
                     int n_steps, const float* sample, int Bs, int dup, int T, const int32_t* counts, float* eps, float* ws,
                     size_t ws_floats, hipStream_t s, int b_lo, int b_n, int loop_mode, int ntxt, const int32_t* d_base) {
    const int B2 = Bs * dup;
    if (b_n < 0) { b_lo = 0; b_n = B2; }
    const int M = b_n * T;
    if (T < 1 || T > LADIFF_MAX_LATENTS || b_lo < 0 || b_lo + b_n > B2) return LADIFF_ERR_SHAPE;
    if (M == 0) return 0;
    if (ws_floats < den_forward_ws_floats(b_n, T)) return LADIFF_ERR_WORKSPACE;
    eps += (size_t)b_lo * T * D;
    const bool sp = wsp != nullptr;
    const size_t MD = (size_t)M * D;
    float* P[4]; float* SK[NSKIP]; float* Ps[4]; float* SKs[NSKIP];
    float* p = ws;
    for (int i = 0; i < 4; ++i) { P[i] = p; p += MD; }
    for (int i = 0; i < NSKIP; ++i) { SK[i] = p; p += MD; }
    for (int i = 0; i < 4; ++i) { Ps[i] = sp ? p : nullptr; p += MD; }          // S-format twins
    for (int i = 0; i < NSKIP; ++i) { SKs[i] = sp ? p : nullptr; p += MD; }
    float* qkv = p; p += 3 * MD;
    float* att = p; p += MD;
    float* hid = p; p += (size_t)M * FF;
    float* part = p;                                  // split-K partial planes [4][M][256]
    const float* tkv = den_cache_tkv(cache, B2, ntxt);
    const float* ctab = den_cache_ctab(cache, B2, ntxt);               // ntxt > 1: the [9][B2][4][64][64] key^T value matrices
    const int R = B2 + 1;
    // operand view of a tensor: its S-format twin in the f16x3 path, the fp32 tensor otherwise
    auto gemm = [&](KrArgs g) { g.split = sp ? 1 : 0; return launch_gemm_kr(g, s); };

    // x = cat([sample]*dup) + query_pos.pe[:T]        ladiff.py:472-474, ladiff_denoiser.py:251
    // loop_mode: the caller's step-tail kernel has already written x into P[0] / Ps[0] and will apply encoder.norm itself
    if (!loop_mode) LADIFF_TRY(launch_add_pe(sample, w.query_pe, Bs, b_lo, b_n, T, P[0], Ps[0], s));
    const float* cur = P[0]; const float* curs = Ps[0];
    for (int l = 0; l < NL; ++l) {
        const DenLayerW& L = w.layer[l];
        const DenLayerW& Ls = sp ? wsp->layer[l] : w.layer[l];      // matrices as the MFMA reads them
        const float* tl = tables + (size_t)l * DEN_LAYER_STRIDE;
        const bool is_in = l < NSKIP, is_out = l > NSKIP;
        if (is_out) {
            const LinearW& sk = w.skip[l - NSKIP - 1];
            const LinearW& sks = sp ? wsp->skip[l - NSKIP - 1] : sk;
            if (sp) {   // linear_blocks[i](cat(x, skip)): whole 256-wide rows per workgroup, K = 512 streamed (gemm_rowln.hip)
                RowLnArgs g;
                g.A = curs; g.lda = D; g.A2 = SKs[NL - 1 - l]; g.lda2 = D; g.K1 = D; g.W = sks.w; g.ldw = 2 * D; g.bias = sk.b;
                g.Y = P[3]; g.Ys = Ps[3]; g.ldy = D; g.M = M; g.K = 2 * D;
                LADIFF_TRY(launch_gemm_rowln(g, s));
            } else {
                KrArgs g = kr(cur, D, sks.w, nullptr, part, D, M, D, 2 * D);
                g.A2 = SK[NL - 1 - l]; g.lda2 = D; g.K1 = D;
                LADIFF_TRY(gemm(g));
                LADIFF_TRY(launch_reduce_rows(part, 2, M, sk.b, nullptr, RED_PLAIN, nullptr, nullptr, nullptr, 0, nullptr,
                                              nullptr, 1, 1, 0, 0, P[3], Ps[3], s));
            }
            cur = P[3]; curs = Ps[3];
        }
        const float* att_op = att;                    // the attention output as out_proj's operand
        if (sp && ntxt == 1) {   // in_proj + attention core in one launch (qkv_attn.hip)
            LADIFF_TRY(launch_qkv_attention(curs, Ls.sa_attn.in_w, L.sa_attn.in_b, tkv + (size_t)l * B2 * 2 * D, tl,
                                            DEN_OFF_TIME_KV, DEN_STEP_STRIDE, d_step, counts, Bs, b_lo, b_n, T, att, s));
        } else {
            LADIFF_TRY(gemm(kr(sp ? curs : cur, D, Ls.sa_attn.in_w, L.sa_attn.in_b, qkv, 3 * D, M, 3 * D, D)));
            if (ntxt > 1) {
                // N text tokens as N extra keys (fp32 softmax in both modes); f16x3 mode: the S-format twin of the result goes to
                // the (still unused) hidden buffer
                LADIFF_TRY(launch_denoiser_self_attention_general(qkv, tkv + (size_t)l * B2 * ntxt * 2 * D, ntxt, tl, DEN_OFF_TIME_KV,
                                                                  DEN_STEP_STRIDE, d_step, counts, Bs, b_lo, b_n, T, att, s));
                if (sp) { LADIFF_TRY(launch_split_rows(att, hid, M, D, s)); att_op = hid; }
            } else
                LADIFF_TRY(launch_denoiser_self_attention(qkv, tkv + (size_t)l * B2 * 2 * D, tl, DEN_OFF_TIME_KV, DEN_STEP_STRIDE,
                                                          d_step, counts, Bs, b_lo, b_n, T, att, 0, s));
        }
        if (sp) {   // X1 = LN1(x + out_proj(att)) -> P[2] / Ps[2], one launch (gemm_rowln.hip)
            RowLnArgs g;
            g.A = att_op; g.lda = D; g.W = Ls.sa_attn.out_w; g.ldw = D; g.bias = L.sa_attn.out_b; g.res = cur; g.ldres = D;
            g.ln_g = L.sa_norm1.g; g.ln_b = L.sa_norm1.b; g.Y = P[2]; g.Ys = Ps[2]; g.ldy = D; g.M = M; g.K = D;
            LADIFF_TRY(launch_gemm_rowln(g, s));
        } else {
            // R1 = x + out_proj(att) -> P[1]
            KrArgs g = kr(att, D, Ls.sa_attn.out_w, L.sa_attn.out_b, P[1], D, M, D, D);
            g.res = cur; g.ldres = D;
            LADIFF_TRY(gemm(g));
            // X1 = LN1(R1) -> P[2]
            LADIFF_TRY(launch_reduce_rows(P[1], 1, M, nullptr, nullptr, RED_LN, L.sa_norm1.g, L.sa_norm1.b, nullptr, 0, nullptr,
                                          nullptr, 1, 1, 0, 0, P[2], Ps[2], s));
        }
        // hid = relu(linear1(X1))
        {
            KrArgs g = kr(sp ? Ps[2] : P[2], D, Ls.sa_lin1.w, L.sa_lin1.b, sp ? nullptr : hid, FF, M, FF, D, ACT_RELU);
            if (sp) g.Ys = hid;
            LADIFF_TRY(gemm(g));
        }
        // X3 = LN2(X1 + linear2(hid)) + c[step, layer, sample] -> P[1]
        LADIFF_TRY(gemm(kr(hid, FF, Ls.sa_lin2.w, nullptr, part, D, M, D, FF)));
        const float* x3 = P[1];
        if (ntxt > 1) {
            // literal LinearTemporalCrossAttention (mdiff_transformer.py:219-247): X2 = LN2(..) -> P[1]; q = query(LN(X2));
            // u = SiLU(AdaLN(LN(softmax_d(q) . att_b)));  X3 = X2 + out(u) -> first M x 256 of the qkv buffer
            // f16x3 mode: the two projections run on S-format operands (3 bf16 MFMAs per product), softmax / LayerNorm / AdaLN fp32
            LADIFF_TRY(launch_reduce_rows(part, 4, M, L.sa_lin2.b, P[2], RED_LN, L.sa_norm2.g, L.sa_norm2.b, nullptr, 0, nullptr,
                                          nullptr, 1, 1, 0, 0, P[1], nullptr, s));
            if (sp) LADIFF_TRY(launch_reduce_rows(P[1], 1, M, nullptr, nullptr, RED_LN, L.ca_norm.g, L.ca_norm.b, nullptr, 0, nullptr,
                                                  nullptr, 1, 1, 0, 0, P[2], Ps[2], s));
            else LADIFF_TRY(launch_layernorm(P[1], L.ca_norm.g, L.ca_norm.b, P[2], M, s));
            LADIFF_TRY(gemm(kr(sp ? Ps[2] : P[2], D, Ls.ca_query.w, L.ca_query.b, att, D, M, D, D)));
            LADIFF_TRY(launch_lca_apply(att, ctab + (size_t)l * B2 * H * DH * DH, counts, Bs, b_lo, b_n, T, tl + DEN_OFF_CA_MOD,
                                        DEN_STEP_STRIDE, 0, d_step, L.ca_proj.norm.g, L.ca_proj.norm.b, P[2], s));
            if (sp) LADIFF_TRY(launch_split_rows(P[2], Ps[2], M, D, s));
            KrArgs g = kr(sp ? Ps[2] : P[2], D, Ls.ca_proj.out.w, L.ca_proj.out.b, qkv, D, M, D, D);
            g.res = P[1]; g.ldres = D; g.Ys = sp ? Ps[1] : nullptr;      // X3 (fp32: residual of the ffn block) + its S-format twin (ffn.linear1's operand)
            LADIFF_TRY(gemm(g));
            x3 = qkv;
        } else {
            LADIFF_TRY(launch_reduce_rows(part, 4, M, L.sa_lin2.b, P[2], RED_LN_ADD, L.sa_norm2.g, L.sa_norm2.b,
                                          ctab + (size_t)l * n_steps * R * D, R * D, d_step, counts, Bs, T, B2, b_lo, P[1], Ps[1], s, d_base));
        }
        // u = SiLU(AdaLN(ffn.linear2(gelu(ffn.linear1(X3))))) -> P[2]
        {
            KrArgs g = kr(sp ? Ps[1] : x3, D, Ls.ffn1.w, L.ffn1.b, sp ? nullptr : hid, FF, M, FF, D, ACT_GELU);
            if (sp) g.Ys = hid;
            LADIFF_TRY(gemm(g));
        }
        LADIFF_TRY(gemm(kr(hid, FF, Ls.ffn2.w, nullptr, part, D, M, D, FF)));
        float* dst = is_in ? SK[l] : P[0];
        float* dsts = is_in ? SKs[l] : Ps[0];
        if (sp) {   // combine + StylizationBlock + out projection + residual in one launch (gemm_rowln.hip)
            CombineGemmArgs g;
            g.P = part; g.plane = MD; g.S = 4; g.bias2 = L.ffn2.b; g.ln_g = L.ffn_proj.norm.g; g.ln_b = L.ffn_proj.norm.b;
            g.tab = tl + DEN_OFF_FFN_MOD; g.tab_step_stride = DEN_STEP_STRIDE; g.d_step = d_step;
            g.W = Ls.ffn_proj.out.w; g.ldw = D; g.bias = L.ffn_proj.out.b; g.res = x3; g.ldres = D;
            g.Y = dst; g.Ys = dsts; g.ldy = D; g.M = M;
            LADIFF_TRY(launch_combine_gemm(g, s));
        } else {
            LADIFF_TRY(launch_reduce_rows(part, 4, M, L.ffn2.b, nullptr, RED_LN_MOD, L.ffn_proj.norm.g, L.ffn_proj.norm.b,
                                          tl + DEN_OFF_FFN_MOD, DEN_STEP_STRIDE, d_step, nullptr, 1, 1, 0, 0, P[2], nullptr, s));
            // x' = X3 + out_layers(u)
            KrArgs g = kr(P[2], D, Ls.ffn_proj.out.w, L.ffn_proj.out.b, dst, D, M, D, D);
            g.res = x3; g.ldres = D; g.Ys = dsts;
            LADIFF_TRY(gemm(g));
        }
        cur = dst; curs = dsts;
    }
    // encoder.norm, then [B2,T,256] out   cross_attention.py:84-85, ladiff_denoiser.py:272,292
    if (loop_mode) return 0;          // left in P[0] for launch_step_tail
    return launch_layernorm(cur, w.norm.g, w.norm.b, eps, M, s);
}

}  // namespace ladiff
