// Host-side sequencing of the LA-VAE encoder (LADiffVae.encode, ladiff_vae.py:162-286; live branch: pe "mld", MAX_IT > 0,
// LAD, no MLP_DIST / JOINT_DISTRO_FIX / DVAE) - SURVEY.md §8f-3, the row next to the sampling path.  It reuses the
// decoder's kernels: large-M GEMMs with fused residual + LayerNorm, the MFMA self-attention core (now with an arbitrary
// key map: the masked latent tokens sit in the middle of the sequence), row kernels.  Sequence = [T mu tokens | T logvar
// tokens | F frames] per sample, batch-major rows (row = b * S + s), S = 2T + F <= 224.
#include "model.h"

namespace ladiff {

static GemmArgs lin(const float* A, int lda, const float* W, const float* bias, float* Y, int ldy, int M, int N, int K,
                    int act = ACT_NONE) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = K; g.bias = bias; g.Y = Y; g.ldy = ldy; g.M = M; g.N = N; g.K = K; g.act = act;
    return g;
}

static int pad32(int c) { return (c + 31) / 32 * 32; }

size_t enc_ws_floats(int B, int F, int T, int C) {
    const size_t M = (size_t)B * (2 * T + F), Cp = pad32(C);
    return M * (16 * D + 3 * D + D + FF) + (size_t)B * F * Cp + (size_t)D * Cp + (size_t)B * F * D + (size_t)B * 8 + 64;
}

int vae_encode(const EncoderW& w, const EncoderW* wsp, const float* features, const int32_t* lengths, const int32_t* counts,
               const float* eps, int B, int F, int T, int C, float* mu, float* sd, float* latent, float* ws,
               size_t ws_floats, hipStream_t s) {
    const int S = 2 * T + F;
    if (F < 1 || S > LADIFF_MAX_FRAMES || T < 1 || T > LADIFF_MAX_LATENTS || C < 1) return LADIFF_ERR_SHAPE;
    if (ws_floats < enc_ws_floats(B, F, T, C)) return LADIFF_ERR_WORKSPACE;
    const int M = B * S;
    if (B == 0) return 0;
    const bool sp = wsp != nullptr;
    const size_t MD = (size_t)M * D;
    const int Cp = pad32(C);
    float* P[4]; float* SK[NSKIP]; float* Ps[4]; float* SKs[NSKIP];
    float* p = ws;
    for (int i = 0; i < 4; ++i) { P[i] = p; p += MD; }
    for (int i = 0; i < NSKIP; ++i) { SK[i] = p; p += MD; }
    for (int i = 0; i < 4; ++i) { Ps[i] = sp ? p : nullptr; p += MD; }
    for (int i = 0; i < NSKIP; ++i) { SKs[i] = sp ? p : nullptr; p += MD; }
    float* qkv = p; p += 3 * MD;
    float* att = p; p += MD;
    float* hid = p; p += (size_t)M * FF;
    float* featp = p; p += (size_t)B * F * Cp;
    float* wskel = p; p += (size_t)D * Cp;
    float* emb = p; p += (size_t)B * F * D;
    uint32_t* keybits = reinterpret_cast<uint32_t*>(p);

    auto gemm_ln = [&](const float* A, int K, const float* W, const float* Wsp, const float* bias, const float* res,
                       const NormW& n1, const NormW* n2, float* dst, float* dsts) -> int {
        GemmArgs g = lin(A, K, sp ? Wsp : W, bias, dst, D, M, D, K);
        g.res = res; g.ldres = D;
        if (!sp) {
            g.ln_g = n1.g; g.ln_b = n1.b;
            if (n2) { g.ln2_g = n2->g; g.ln2_b = n2->b; }
            return launch_gemm(g, s);
        }
        g.split = 1;
        LADIFF_TRY(launch_gemm(g, s));
        if (n2) {
            LADIFF_TRY(launch_reduce_rows(dst, 1, M, nullptr, nullptr, RED_LN, n1.g, n1.b, nullptr, 0, nullptr, nullptr, 1, 1, 0,
                                          0, dst, nullptr, s));
            return launch_reduce_rows(dst, 1, M, nullptr, nullptr, RED_LN, n2->g, n2->b, nullptr, 0, nullptr, nullptr, 1, 1, 0,
                                      0, dst, dsts, s);
        }
        return launch_reduce_rows(dst, 1, M, nullptr, nullptr, RED_LN, n1.g, n1.b, nullptr, 0, nullptr, nullptr, 1, 1, 0, 0,
                                  dst, dsts, s);
    };

    // x = skel_embedding(features)  (K = nfeats is padded to a multiple of 32 columns)            ladiff_vae.py:182
    LADIFF_TRY(launch_pad_cols(features, featp, B * F, C, Cp, s));
    LADIFF_TRY(launch_pad_cols(w.skel.w, wskel, D, C, Cp, s));
    LADIFF_TRY(launch_gemm(lin(featp, Cp, wskel, w.skel.b, emb, D, B * F, D, Cp), s));
    // xseq = cat(global_motion_token, x) + query_pos_encoder.pe;  key map from lengths / counts       :189-219
    LADIFF_TRY(launch_encoder_assemble(w.motion_token, emb, w.query_pe, lengths, counts, B, F, T, P[0], Ps[0], keybits, s));

    const float* cur = P[0]; const float* curs = Ps[0];
    for (int l = 0; l < NL; ++l) {       // SkipTransformerEncoder, MD_trans == False branch            cross_attention.py:48-67
        const EncLayerW& L = w.layer[l];
        const EncLayerW& Ls = sp ? wsp->layer[l] : w.layer[l];
        const bool is_in = l < NSKIP, is_out = l > NSKIP, last = l == NL - 1;
        if (is_out) {
            const LinearW& sk = w.skip[l - NSKIP - 1];
            GemmArgs g = lin(sp ? curs : cur, D, sp ? wsp->skip[l - NSKIP - 1].w : sk.w, sk.b, P[3], D, M, D, 2 * D);
            g.A2 = sp ? SKs[NL - 1 - l] : SK[NL - 1 - l]; g.lda2 = D; g.K1 = D;
            g.split = sp ? 1 : 0; g.Ys = Ps[3];
            LADIFF_TRY(launch_gemm(g, s));
            cur = P[3]; curs = Ps[3];
        }
        {   // TransformerEncoderLayer.forward_post                                                     cross_attention.py:293-307
            GemmArgs g = lin(sp ? curs : cur, D, Ls.self_attn.in_w, L.self_attn.in_b, qkv, 3 * D, M, 3 * D, D);
            g.split = sp ? 1 : 0;
            LADIFF_TRY(launch_gemm(g, s));
        }
        if (sp) LADIFF_TRY(launch_self_attention_split(qkv, nullptr, keybits, att, B, S, H, 0, 1, s));
        else LADIFF_TRY(launch_decoder_self_attention(qkv, nullptr, keybits, att, B, S, 0, s));
        LADIFF_TRY(gemm_ln(att, D, L.self_attn.out_w, Ls.self_attn.out_w, L.self_attn.out_b, cur, L.norm1, nullptr, P[1], Ps[1]));
        {
            GemmArgs g = lin(sp ? Ps[1] : P[1], D, Ls.lin1.w, L.lin1.b, sp ? nullptr : hid, FF, M, FF, D, ACT_GELU);
            g.split = sp ? 1 : 0; if (sp) g.Ys = hid;
            LADIFF_TRY(launch_gemm(g, s));
        }
        float* dst = is_in ? SK[l] : P[0];
        float* dsts = is_in ? SKs[l] : Ps[0];
        LADIFF_TRY(gemm_ln(hid, FF, L.lin2.w, Ls.lin2.w, L.lin2.b, P[1], L.norm2, last ? &w.norm : nullptr, dst, dsts));
        cur = dst; curs = dsts;
    }
    return launch_encoder_finalize(cur, eps, counts, B, T, S, mu, sd, latent, s);
}

}  // namespace ladiff
