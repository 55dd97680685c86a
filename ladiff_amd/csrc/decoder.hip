// Host-side sequencing of the LA-VAE decoder (LADiffVae.decode, ladiff_vae.py:288-362) on the gfx950 kernels.
// Activations are kept batch-major [B*F, 256] (the reference is [F, B, 256]); all ops are per-row or per-sample,
// so this only changes strides and lets the final write land directly in [B, F, C].
#include "model.h"
#ifdef LADIFF_STAMPS
#include <cstdlib>
#endif

namespace ladiff {

// Diagnostic twin only: LADIFF_DEC_CUT=n in the environment (read per call) ends a decode after its first n launches (garbage output) -
// scripts/two_streams_aggressor_bisect.py bisects with it which launch of a co-running decode disturbs another one.  Nothing of it is in the product.
#ifdef LADIFF_STAMPS
#define DEC_CUT() do { if (dec_cut >= 0 && ++dec_launched >= dec_cut) return 0; } while (0)
#else
#define DEC_CUT() do { } while (0)
#endif

constexpr int DEC_SMALL_ROWS = 4096;
std::atomic<int> g_dec_out_cross{1};          // measurement switch (+ 64): self-attention out_proj GEMM + cross-attention row kernel as two launches (the path before)
std::atomic<int> g_dec_fused_attn{1};         // measurement switch (+ 16): in_proj GEMM + attention kernel as two launches (the path before)
std::atomic<int> g_dec_final_split{1};        // measurement switch (ladiff_debug_set_decoder_fusion + 8): final_layer on the fp32 kernel as in round 2
std::atomic<int> g_dec_small_rows_path{1};    // measurement switch (ladiff_debug_set_decoder_fusion bit 2 clear / set): the small-M GEMM routing
std::atomic<int> g_dec_fused_mlp{1};          // measurement switch (ladiff_debug_set_decoder_fusion): 0 = linear1 / linear2 / LayerNorm as three launches, 1 = fused from dec_mlp_min_rows() rows, 2 = fused always

static GemmArgs lin(const float* A, int lda, const float* W, const float* bias, float* Y, int ldy, int M, int N, int K,
                    int act = ACT_NONE) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = K; g.bias = bias; g.Y = Y; g.ldy = ldy; g.M = M; g.N = N; g.K = K; g.act = act;
    return g;
}

// rows = B * F for the padded layout, or the sum of the lengths for the ragged one
size_t dec_ws_floats(int B, size_t rows, int T) {
    return rows * (16 * D + 3 * D + D + FF) + (rows + 63) / 64 * 64 + (size_t)NL * ((size_t)T * B * 2 * D + dec_cross_ws_floats(B, T)) +
           (size_t)LADIFF_MAX_FRAMES * (2 * D + 3 * D);       // layer 0: the position table as a GEMM operand and its q|k|v
}

// wsp != nullptr selects the f16x3 matrix path (S-format copies of the weight matrices in `wsp`, GEMM operands in
// S-format, LayerNorm as a row kernel after each fused GEMM); see denoiser.hip and common.h.
// row_off != nullptr (B + 1 entries, row_off[B] = R): RAGGED rows - sample b owns rows [row_off[b], row_off[b] + lengths[b]) and
// only the R valid frames of a mixed-length batch are computed.  Every op of the decoder is per row or per sample (padded
// frames are masked keys, cross_attention.py:367-371, and zeroed at the end, ladiff_vae.py:356-360), so a sample's frames come
// out as in the padded layout; they are scattered to feats[b, f < length] of a [B, F, C] tensor the caller has zero-filled.
int vae_decode(const DecoderW& w, const DecoderW* wsp, const float* z, const int32_t* lengths, const int32_t* counts,
               const int32_t* row_off, int R, int B, int F, int T, int C, float* feats, float* ws, size_t ws_floats, hipStream_t s) {
    if (F < 1 || F > LADIFF_MAX_FRAMES || T < 1 || T > LADIFF_MAX_LATENTS || C < 1) return LADIFF_ERR_SHAPE;
    const bool ragged = row_off != nullptr;
    if (ragged && (R < 0 || (size_t)R > (size_t)B * F)) return LADIFF_ERR_SHAPE;
    const int M = ragged ? R : B * F;
    if (ws_floats < dec_ws_floats(B, (size_t)M, T)) return LADIFF_ERR_WORKSPACE;
    if (M == 0) return 0;
#ifdef LADIFF_STAMPS
    const char* dec_cut_env = getenv("LADIFF_DEC_CUT");
    const int dec_cut = dec_cut_env ? atoi(dec_cut_env) : -1;
    int dec_launched = 0;
    if (dec_cut == 0) return 0;
#endif
    const bool sp = wsp != nullptr;
    const size_t MD = (size_t)M * D;
    float* P[4]; float* SK[NSKIP]; float* Ps[4]; float* SKs[NSKIP];
    float* p = ws;
    for (int i = 0; i < 4; ++i) { P[i] = p; p += MD; }
    for (int i = 0; i < NSKIP; ++i) { SK[i] = p; p += MD; }
    for (int i = 0; i < 4; ++i) { Ps[i] = sp ? p : nullptr; p += MD; }
    for (int i = 0; i < NSKIP; ++i) { SKs[i] = sp ? p : nullptr; p += MD; }
    float* qkv = p; p += 3 * MD;
    float* att = p; p += MD;
    float* hid = p; p += (size_t)M * FF;
    float* kv = p; p += (size_t)NL * T * B * 2 * D;      // memory K | V of every layer
    float* guws = p; p += (size_t)NL * dec_cross_ws_floats(B, T);   // G | U | c of the folded cross-attention of every layer (dec_cross.hip)
    int32_t* row_out = reinterpret_cast<int32_t*>(p); p += ((size_t)M + 63) / 64 * 64;   // ragged: place of each row in the padded output (rounded: what follows is read 16 bytes at a time and by LDS-DMA)
    float* pex = p; p += (size_t)LADIFF_MAX_FRAMES * D;   // layer 0: pe[:F] (+ S-format twin) and its in_proj, shared by all samples
    float* pexs = p; p += (size_t)LADIFF_MAX_FRAMES * D;
    float* qkv0 = p;
    const size_t kv_l = (size_t)T * B * 2 * D, gu_l = dec_cross_ws_floats(B, T);

    // The memory side of every layer's cross-attention depends on z and the weights only: the nine K | V projections and the
    // nine folds into G | U | c run up front, one launch each (fp32 in both modes).   cross_attention.py:373-376
    {
        GemmArgs g[NL];
        DecCrossPrepBatch pb;
        for (int l = 0; l < NL; ++l) {
            const DecLayerW& L = w.layer[l];
            g[l] = lin(z, D, L.cross_attn.in_w + (size_t)D * D, L.cross_attn.in_b + D, kv + l * kv_l, 2 * D, T * B, 2 * D, D);
            pb.kv[l] = kv + l * kv_l; pb.wq[l] = L.cross_attn.in_w; pb.bq[l] = L.cross_attn.in_b; pb.wo[l] = L.cross_attn.out_w;
            pb.gu[l] = guws + l * gu_l;
        }
        LADIFF_TRY(launch_gemm_batch(g, NL, s)); DEC_CUT();
        LADIFF_TRY(launch_decoder_cross_prep(pb, NL, B, T, s)); DEC_CUT();
    }

    // Few frame rows (config c1: 8 x 60 = 480): a 128x128-tile launch is then 8 - 32 workgroups that each walk the whole K, ~16 us per
    // GEMM whatever its size (profiles/r3: 40 of them were 0.63 of a 0.9 ms decode).  Below DEC_SMALL_ROWS the f16x3 path runs its
    // GEMMs on the denoiser's small-M kernels instead (gemm_kr.hip: K-resident 32 / 64 / 80-row tiles, K = 1024 as four partial planes
    // that the LayerNorm row pass sums) - the same S-format operands and products, 3 - 4x the workgroups.
    const bool small = sp && M < DEC_SMALL_ROWS && g_dec_small_rows_path;
    // in_proj inside the attention kernel (dec_qkv_attn.hip): from DEC_SMALL_ROWS frame rows up (8 x 60 frames: 0.47 ms against 0.45
    // with the small-M in_proj + attention launches; 128 x 196: 2.28 against 2.41, profiles/r3/15)
    const bool fused_attn = sp && g_dec_fused_attn && (M >= DEC_SMALL_ROWS || g_dec_fused_attn == 2);
    auto krs = [&](const float* A, int K, const float* W, const float* bias, float* Y, float* Ys, int ldy, int N, int act,
                   const float* res, int rows) -> int {
        KrArgs g;
        g.A = A; g.lda = K; g.W = W; g.ldw = K; g.bias = bias; g.Y = Y; g.Ys = Ys; g.ldy = ldy; g.M = rows; g.N = N; g.K = K; g.act = act;
        g.res = res; g.ldres = D; g.split = 1;
        return launch_gemm_kr(g, s);
    };

    // GEMM + (residual) + LayerNorm: fused epilogue in the fp32 path; GEMM(+residual) then a LayerNorm row kernel in the
    // f16x3 path.  `A`/`As`: operand in fp32 / S-format; result (fp32 + S-format twin) goes to dst / dsts.
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
        if (n2)      // norm3, then decoder.norm: one row pass
            return launch_reduce_rows(dst, 1, M, nullptr, nullptr, RED_LN, n1.g, n1.b, nullptr, 0, nullptr, nullptr, 1, 1, 0,
                                      0, dst, dsts, s, nullptr, n2->g, n2->b);
        return launch_reduce_rows(dst, 1, M, nullptr, nullptr, RED_LN, n1.g, n1.b, nullptr, 0, nullptr, nullptr, 1, 1, 0, 0,
                                  dst, dsts, s);
    };

    // queries = zeros + query_pos_decoder.pe[:F]     ladiff_vae.py:299, :334
    if (ragged) { LADIFF_TRY(launch_broadcast_pe_ragged(w.query_pe, row_off, B, F, F, P[0], Ps[0], row_out, s)); DEC_CUT(); }
    else { LADIFF_TRY(launch_broadcast_pe(w.query_pe, B, F, P[0], Ps[0], s)); DEC_CUT(); }
    const float* cur = P[0]; const float* curs = Ps[0];
    for (int l = 0; l < NL; ++l) {
        const DecLayerW& L = w.layer[l];
        const DecLayerW& Ls = sp ? wsp->layer[l] : w.layer[l];
        const bool is_in = l < NSKIP, is_out = l > NSKIP, last = l == NL - 1;
        if (is_out) {   // x = linear(cat([x, xs.pop()]))   cross_attention.py:140-142
            const LinearW& sk = w.skip[l - NSKIP - 1];
            if (small) {        // whole 256-wide rows per workgroup, K = 512 streamed (gemm_rowln.hip, as the denoiser's skip layers)
                RowLnArgs g;
                g.A = curs; g.lda = D; g.A2 = SKs[NL - 1 - l]; g.lda2 = D; g.K1 = D; g.W = wsp->skip[l - NSKIP - 1].w; g.ldw = 2 * D; g.bias = sk.b;
                g.Y = P[3]; g.Ys = Ps[3]; g.ldy = D; g.M = M; g.K = 2 * D;
                LADIFF_TRY(launch_gemm_rowln(g, s)); DEC_CUT();
            } else {
            GemmArgs g = lin(sp ? curs : cur, D, sp ? wsp->skip[l - NSKIP - 1].w : sk.w, sk.b, P[3], D, M, D, 2 * D);
            g.A2 = sp ? SKs[NL - 1 - l] : SK[NL - 1 - l]; g.lda2 = D; g.K1 = D;
            g.split = sp ? 1 : 0; g.Ys = Ps[3];
            LADIFF_TRY(launch_gemm(g, s)); DEC_CUT();
            }
            cur = P[3]; curs = Ps[3];
        }
        // ---- self-attention over frames, keys >= len masked   cross_attention.py:367-371
        // Layer 0's input is the position table for every sample (queries = zeros + pe, ladiff_vae.py:299, :334): its in_proj runs on
        // the F table rows once and every sample's attention reads those q | k | v (its own length still masks the keys).
        const bool shared = l == 0;
        if (shared) {
            LADIFF_TRY(launch_broadcast_pe(w.query_pe, 1, F, pex, sp ? pexs : nullptr, s)); DEC_CUT();
            if (small) { LADIFF_TRY(krs(pexs, D, Ls.self_attn.in_w, L.self_attn.in_b, qkv0, nullptr, 3 * D, 3 * D, ACT_NONE, nullptr, F)); DEC_CUT(); }
            else {
            GemmArgs g = lin(sp ? pexs : pex, D, Ls.self_attn.in_w, L.self_attn.in_b, qkv0, 3 * D, F, 3 * D, D);
            g.split = sp ? 1 : 0;
            LADIFF_TRY(launch_gemm(g, s)); DEC_CUT();
            }
        } else if (fused_attn) {
            // f16x3 mode: in_proj inside the attention kernel (dec_qkv_attn.hip) - the [M, 768] q | k | v rows are never written
        } else if (small) {
            LADIFF_TRY(krs(curs, D, Ls.self_attn.in_w, L.self_attn.in_b, qkv, nullptr, 3 * D, 3 * D, ACT_NONE, nullptr, M)); DEC_CUT();
        } else {
            GemmArgs g = lin(sp ? curs : cur, D, Ls.self_attn.in_w, L.self_attn.in_b, qkv, 3 * D, M, 3 * D, D);
            g.split = sp ? 1 : 0;
            LADIFF_TRY(launch_gemm(g, s)); DEC_CUT();
        }
        const float* qkv_l = shared ? qkv0 : qkv;
        if (fused_attn && !shared) { LADIFF_TRY(launch_dec_qkv_attn(curs, Ls.self_attn.in_w, L.self_attn.in_b, lengths, row_off, att, B, F, 1, s)); DEC_CUT(); }
        else if (sp) { LADIFF_TRY(launch_self_attention_split(qkv_l, lengths, nullptr, att, B, F, H, 0, 1, s, row_off, shared)); DEC_CUT(); }
        else { LADIFF_TRY(launch_decoder_self_attention(qkv_l, lengths, nullptr, att, B, F, 0, s, row_off, shared)); DEC_CUT(); }
        // norm1: fused in the GEMM epilogue in fp32 mode; in f16x3 mode the GEMM writes x + out_proj(att) and the only reader of
        // norm1's output, the cross-attention kernel below, normalises its rows as it loads them (one row kernel pass less)
        const NormW* n1_late = nullptr;
        if (sp && !small && g_dec_out_cross) {
            // f16x3 mode, many rows: out_proj + residual + norm1 + cross-attention + residual + norm2 in ONE kernel that keeps Wo in
            // its registers (dec_cross.hip): x + out_proj(att) is never written
            LADIFF_TRY(launch_decoder_out_cross(att, cur, Ls.self_attn.out_w, L.self_attn.out_b, L.norm1.g, L.norm1.b, L.cross_attn.out_b,
                                                L.norm2.g, L.norm2.b, counts, B, F, T, guws + l * gu_l, P[2], Ps[2], s, row_off)); DEC_CUT();
        } else {
        if (sp) {
            if (small) { LADIFF_TRY(krs(att, D, Ls.self_attn.out_w, L.self_attn.out_b, P[1], nullptr, D, D, ACT_NONE, cur, M)); DEC_CUT(); }
            else {
            GemmArgs g = lin(att, D, Ls.self_attn.out_w, L.self_attn.out_b, P[1], D, M, D, D);
            g.res = cur; g.ldres = D; g.split = 1;
            LADIFF_TRY(launch_gemm(g, s)); DEC_CUT();
            }
            n1_late = &L.norm1;
        } else {
            LADIFF_TRY(gemm_ln(att, D, L.self_attn.out_w, Ls.self_attn.out_w, L.self_attn.out_b, cur, L.norm1, nullptr, P[1], nullptr)); DEC_CUT();
        }
        // ---- cross-attention to the latent tokens, tokens >= ceil(len/48) masked, + residual + norm2   :373-376, :408-409
        // (the q / out projections are folded into the <= 8 keys / values per sample: dec_cross.hip; fp32 in both modes)
        LADIFF_TRY(launch_decoder_cross_apply(P[1], L.cross_attn.out_b, L.norm2.g, L.norm2.b, counts, B, F, T, guws + l * gu_l, P[2],
                                              Ps[2], s, row_off, n1_late ? n1_late->g : nullptr, n1_late ? n1_late->b : nullptr)); DEC_CUT();
        }
        // ---- feed-forward, GELU(erf), + residual + norm3 (+ decoder.norm on the last layer, cross_attention.py:150-151)   :410-412
        float* dst = is_in ? SK[l] : P[0];
        float* dsts = is_in ? SKs[l] : Ps[0];
        if (sp && g_dec_fused_mlp && (M >= dec_mlp_min_rows() || g_dec_fused_mlp == 2)) {     // one kernel: the hidden rows never leave the registers (dec_mlp.hip)
            LADIFF_TRY(launch_dec_mlp(Ps[2], P[2], Ls.lin1.w, L.lin1.b, Ls.lin2.w, L.lin2.b, L.norm3.g, L.norm3.b,
                                      last ? w.norm.g : nullptr, last ? w.norm.b : nullptr, dst, dsts, M, s)); DEC_CUT();
        } else if (small) {
            LADIFF_TRY(krs(Ps[2], D, Ls.lin1.w, L.lin1.b, nullptr, hid, FF, FF, ACT_GELU, nullptr, M)); DEC_CUT();
            // linear2: K = 1024 as four partial planes (the q|k|v + attention buffers are free by now), summed by the LayerNorm pass
            LADIFF_TRY(krs(hid, FF, Ls.lin2.w, nullptr, qkv, nullptr, D, D, ACT_NONE, nullptr, M)); DEC_CUT();
            LADIFF_TRY(launch_reduce_rows(qkv, 4, M, L.lin2.b, P[2], RED_LN, L.norm3.g, L.norm3.b, nullptr, 0, nullptr, nullptr, 1, 1, 0, 0,
                                          dst, dsts, s, nullptr, last ? w.norm.g : nullptr, last ? w.norm.b : nullptr));
        } else {
            GemmArgs g = lin(sp ? Ps[2] : P[2], D, Ls.lin1.w, L.lin1.b, sp ? nullptr : hid, FF, M, FF, D, ACT_GELU);
            g.split = sp ? 1 : 0; if (sp) g.Ys = hid;
            LADIFF_TRY(launch_gemm(g, s)); DEC_CUT();
            LADIFF_TRY(gemm_ln(hid, FF, L.lin2.w, Ls.lin2.w, L.lin2.b, P[2], L.norm3, last ? &w.norm : nullptr, dst, dsts)); DEC_CUT();
        }
        cur = dst; curs = dsts;
    }
    // final_layer + zero padded frames, written as [B, F, C]   ladiff_vae.py:356-360
    if (sp && M >= DEC_SMALL_ROWS && g_dec_final_split) {
        // f16x3 mode: the projection runs on the large-M f16x3 kernel over whole 128-column tiles, into the (free) hidden buffer, and a
        // row kernel moves the C real columns into [B, F, C] (zeroing padded frames / scattering ragged rows): 94 us -> ~40 us at 25088
        // rows.  The tiles need Np = ceil(C / 128) 128 weight rows: the library pads the caller's C rows ITSELF (a 1-KiB-per-row copy
        // into the in_proj buffer, free by now) - a table of exactly C rows is never read past its end.
        const int Np = (C + 127) / 128 * 128;
        if ((size_t)Np > (size_t)FF || (size_t)Np * (D + 1) > 3 * MD) return LADIFF_ERR_SHAPE;
        float* wpad = qkv; float* bpad = qkv + (size_t)Np * D;
        LADIFF_TRY(launch_pad_rows(wsp->final_layer.w, w.final_layer.b, wpad, bpad, C, Np, s)); DEC_CUT();
        GemmArgs g = lin(curs, D, wpad, bpad, hid, Np, M, Np, D);
        g.split = 1;
        LADIFF_TRY(launch_gemm(g, s)); DEC_CUT();
        return launch_scatter_feats(hid, Np, C, M, F, ragged ? nullptr : lengths, ragged ? row_out : nullptr, feats, s);
    }
    GemmArgs g = lin(cur, D, w.final_layer.w, w.final_layer.b, feats, C, M, C, D);
    if (ragged) g.row_map = row_out;                      // every computed row is a valid frame
    else { g.row_len = lengths; g.rows_per_item = F; }
    return launch_gemm(g, s);
}

}  // namespace ladiff
