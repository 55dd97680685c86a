// Host-side sequencing of the LA-VAE decoder (LADiffVae.decode, ladiff_vae.py:288-362) on the gfx950 kernels.
// Activations are kept batch-major [B*F, 256] (the reference is [F, B, 256]); all ops are per-row or per-sample,
// so this only changes strides and lets the final write land directly in [B, F, C].
#include "model.h"

namespace ladiff {

static GemmArgs lin(const float* A, int lda, const float* W, const float* bias, float* Y, int ldy, int M, int N, int K,
                    int act = ACT_NONE) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = K; g.bias = bias; g.Y = Y; g.ldy = ldy; g.M = M; g.N = N; g.K = K; g.act = act;
    return g;
}

size_t dec_ws_floats(int B, int F, int T) {
    const size_t M = (size_t)B * F;
    return M * (8 * D + 3 * D + D + FF) + (size_t)T * B * 2 * D;
}

int vae_decode(const DecoderW& w, const float* z, const int32_t* lengths, const int32_t* counts, int B, int F, int T,
               int C, float* feats, float* ws, size_t ws_floats, hipStream_t s) {
    if (F < 1 || F > LADIFF_MAX_FRAMES || T < 1 || T > LADIFF_MAX_LATENTS || C < 1) return LADIFF_ERR_SHAPE;
    if (ws_floats < dec_ws_floats(B, F, T)) return LADIFF_ERR_WORKSPACE;
    const int M = B * F;
    if (M == 0) return 0;
    const size_t MD = (size_t)M * D;
    float* P[4]; float* SK[NSKIP];
    float* p = ws;
    for (int i = 0; i < 4; ++i) { P[i] = p; p += MD; }
    for (int i = 0; i < NSKIP; ++i) { SK[i] = p; p += MD; }
    float* qkv = p; p += 3 * MD;
    float* att = p; p += MD;
    float* hid = p; p += (size_t)M * FF;
    float* kv = p;
    float* qb = qkv;   // cross-attention queries reuse the (dead) packed qkv buffer

    // queries = zeros + query_pos_decoder.pe[:F]     ladiff_vae.py:299, :334
    LADIFF_TRY(launch_broadcast_pe(w.query_pe, B, F, P[0], s));
    const float* cur = P[0];
    for (int l = 0; l < NL; ++l) {
        const DecLayerW& L = w.layer[l];
        const bool is_in = l < NSKIP, is_out = l > NSKIP, last = l == NL - 1;
        if (is_out) {   // x = linear(cat([x, xs.pop()]))   cross_attention.py:140-142
            GemmArgs g = lin(cur, D, w.skip[l - NSKIP - 1].w, w.skip[l - NSKIP - 1].b, P[3], D, M, D, 2 * D);
            g.A2 = SK[NL - 1 - l]; g.lda2 = D; g.K1 = D;
            LADIFF_TRY(launch_gemm(g, s));
            cur = P[3];
        }
        // ---- self-attention over frames, keys >= len masked   cross_attention.py:367-371
        LADIFF_TRY(launch_gemm(lin(cur, D, L.self_attn.in_w, L.self_attn.in_b, qkv, 3 * D, M, 3 * D, D), s));
        LADIFF_TRY(launch_decoder_self_attention(qkv, lengths, att, B, F, s));
        {
            GemmArgs g = lin(att, D, L.self_attn.out_w, L.self_attn.out_b, P[1], D, M, D, D);
            g.res = cur; g.ldres = D; g.ln_g = L.norm1.g; g.ln_b = L.norm1.b;
            LADIFF_TRY(launch_gemm(g, s));
        }
        // ---- cross-attention to the latent tokens, tokens >= ceil(len/48) masked   :373-376, :408-409
        LADIFF_TRY(launch_gemm(lin(P[1], D, L.cross_attn.in_w, L.cross_attn.in_b, qb, D, M, D, D), s));
        LADIFF_TRY(launch_gemm(lin(z, D, L.cross_attn.in_w + (size_t)D * D, L.cross_attn.in_b + D, kv, 2 * D, T * B, 2 * D, D), s));
        LADIFF_TRY(launch_decoder_cross_attention(qb, kv, counts, att, B, F, T, s));
        {
            GemmArgs g = lin(att, D, L.cross_attn.out_w, L.cross_attn.out_b, P[2], D, M, D, D);
            g.res = P[1]; g.ldres = D; g.ln_g = L.norm2.g; g.ln_b = L.norm2.b;
            LADIFF_TRY(launch_gemm(g, s));
        }
        // ---- feed-forward, GELU(erf)   :410-412
        LADIFF_TRY(launch_gemm(lin(P[2], D, L.lin1.w, L.lin1.b, hid, FF, M, FF, D, ACT_GELU), s));
        float* dst = is_in ? SK[l] : P[0];
        {
            GemmArgs g = lin(hid, FF, L.lin2.w, L.lin2.b, dst, D, M, D, FF);
            g.res = P[2]; g.ldres = D; g.ln_g = L.norm3.g; g.ln_b = L.norm3.b;
            if (last) { g.ln2_g = w.norm.g; g.ln2_b = w.norm.b; }   // decoder.norm, cross_attention.py:150-151
            LADIFF_TRY(launch_gemm(g, s));
        }
        cur = dst;
    }
    // final_layer + zero padded frames, written as [B, F, C]   ladiff_vae.py:356-360
    GemmArgs g = lin(cur, D, w.final_layer.w, w.final_layer.b, feats, C, M, C, D);
    g.row_len = lengths; g.rows_per_item = F;
    return launch_gemm(g, s);
}

}  // namespace ladiff
