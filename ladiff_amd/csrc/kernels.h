// Launchers of the non-GEMM kernels (rowops.hip, attention.hip).
#pragma once
#include "common.h"
#include "noise_gen.h"

namespace ladiff {

// rowops.hip
enum RedMode : int { RED_PLAIN = 0, RED_LN = 1, RED_LN_ADD = 2, RED_LN_MOD = 3 };
int launch_reduce_rows(const float* P, int S, int M, const float* bias, const float* res, int mode, const float* g,
                       const float* b, const float* tab, int tab_step_stride, const int32_t* d_step,
                       const int32_t* counts, int Bs, int T, int pad_row, int b_off, float* out, float* outs, hipStream_t s,
                       const int32_t* d_base = nullptr, const float* g2 = nullptr, const float* b2 = nullptr);   // g2 / b2: RED_LN twice
int launch_split_rows(const float* x, float* y, int R, int K, hipStream_t s);
int launch_layernorm(const float* x, const float* g, const float* b, float* y, int M, hipStream_t s);
constexpr int DEC_PREP_MAX = 9;
struct DecCrossPrepBatch { const float* kv[DEC_PREP_MAX]; const float* wq[DEC_PREP_MAX]; const float* bq[DEC_PREP_MAX];
                           const float* wo[DEC_PREP_MAX]; float* gu[DEC_PREP_MAX]; };
// the same op on up to ROW_BATCH_MAX independent (x, g, b, y) sets of one M, one launch (grid.y = set)
constexpr int ROW_BATCH_MAX = 9;
struct RowBatch { const float* a[ROW_BATCH_MAX]; const float* g[ROW_BATCH_MAX]; const float* b[ROW_BATCH_MAX]; float* y[ROW_BATCH_MAX]; };
int launch_layernorm_batch(const RowBatch& rb, int n, int M, hipStream_t s);
// ca_table_input for n_layers layers at once: a = nval, g = beta, b = mod (step 0) , y = u of each layer
int launch_ca_table_input_batch(const RowBatch& rb, int n_layers, int step_stride, int n, int B2, hipStream_t s, int split_out = 0);
int launch_ca_table_input(const float* nval, const float* beta, const float* mod, int step_stride, int n, int B2, float* u,
                          hipStream_t s);
int launch_add_pe(const float* sample, const float* pe, int Bs, int b_off, int b_n, int T, float* x, float* xs, hipStream_t s);
int launch_broadcast_pe(const float* pe, int B, int F, float* x, float* xs, hipStream_t s);
// ragged form: sample b owns rows [row_off[b], row_off[b + 1]); also writes each row's place in a [B, F_out, *] tensor to row_out
int launch_broadcast_pe_ragged(const float* pe, const int32_t* row_off, int B, int F, int F_out, float* x, float* xs, int32_t* row_out,
                               hipStream_t s);
int launch_relu(const float* x, float* y, size_t n, hipStream_t s);
int launch_pad_rows(const float* src, const float* srcb, float* dst, float* dstb, int C, int Np, hipStream_t s);
int launch_zero_fill(float* x, size_t n, hipStream_t s);          // kernel, not a memset node (graphs: api.hip g_graph_epoch)
int launch_silu(const float* x, float* y, size_t n, hipStream_t s);
int launch_sinusoid(const int64_t* t, int n, float* out, hipStream_t s);
int launch_cfg_step(const float* eps, float* lat, const float* coef, const int32_t* d_step, const float* noise,
                    float g, int cfg, int B, int T, hipStream_t s);
int launch_advance(int32_t* d_step, hipStream_t s);
int launch_step_tail(float* x, float* xs, const float* ng, const float* nb, float* lat, const float* coef, int32_t* d_step,
                     const float* noise, const float* pe, float g, int cfg, int B, int T, hipStream_t s, const NoiseGen& gen = NoiseGen{0u, 0u, 0u, 0});
int launch_noise_fill(const NoiseGen& gen, int step0, int n, int B, int T, float* out, hipStream_t s);
int launch_init_latents(const float* noise, const int32_t* counts, float sigma, float* lat, int B, int T, hipStream_t s);
int launch_finalize_latents(const float* lat, const int32_t* counts, float* z, int B, int T, hipStream_t s, const unsigned* status = nullptr);

int launch_dec_qkv_attn(const float* xs, const float* w, const float* bias, const int32_t* lengths, const int32_t* row_off, float* out,
                        int B, int F, int split_out, hipStream_t s);
int launch_scatter_feats(const float* tmp, int ldt, int C, int M, int F, const int32_t* row_len, const int32_t* row_map, float* feats,
                         hipStream_t s);
int launch_pad_cols(const float* x, float* y, int R, int C, int Cp, hipStream_t s);
int launch_encoder_assemble(const float* token, const float* emb, const float* pe, const int32_t* lengths,
                            const int32_t* counts, int B, int F, int T, float* x, float* xs, uint32_t* keybits, hipStream_t s);
int launch_encoder_finalize(const float* out, const float* eps, const int32_t* counts, int B, int T, int S, float* mu, float* sd,
                            float* latent, hipStream_t s);

// linear_ca.hip: text conditioning with more than one text token per prompt
int launch_lca_kv(const float* key, const float* value, int B, int N, float* att, hipStream_t s);
int launch_lca_apply(const float* q, const float* att, const int32_t* counts, int Bs, int b_off, int b_n, int T, const float* mod,
                     int step_stride, int sample_stride, const int32_t* d_step, const float* g, const float* be, float* u,
                     hipStream_t s);
int launch_denoiser_self_attention_general(const float* qkv, const float* text_kv, int N, const float* tables, int kv_off,
                                           int step_stride, const int32_t* d_step, const int32_t* counts, int Bs, int b_off,
                                           int b_n, int T, float* out, hipStream_t s);

// dec_cross.hip: the decoder's cross-attention block (q projection, attention over <= 8 memory tokens, out projection,
// residual, norm2) as a per-sample low-rank map
size_t dec_cross_ws_floats(int B, int T);
int launch_decoder_cross_prep(const DecCrossPrepBatch& pb, int n, int B, int T, hipStream_t s);
// g1 / b1 (optional): x holds pre-norm1 rows and the kernel applies LayerNorm(g1, b1) to each row as it loads it
int launch_decoder_out_cross(const float* att_s, const float* x0, const float* wo_s, const float* bo, const float* g1, const float* b1,
                             const float* bo_c, const float* g2, const float* b2, const int32_t* counts, int B, int F, int T,
                             const float* gu_ws, float* y, float* ys, hipStream_t s, const int32_t* row_off);
int launch_decoder_cross_apply(const float* x, const float* bo, const float* g2, const float* b2, const int32_t* counts, int B, int F,
                               int T, const float* gu_ws, float* y, float* ys, hipStream_t s, const int32_t* row_off = nullptr,
                               const float* g1 = nullptr, const float* b1 = nullptr);

// feats2joints.hip
int launch_feats2joints(const float* feats, const float* mean, const float* stdv, int B, int F, int C, int J, float* joints,
                        hipStream_t s);

// attention.hip
int launch_denoiser_self_attention(const float* qkv, const float* text_kv, const float* tables, int kv_off,
                                   int step_stride, const int32_t* d_step, const int32_t* counts, int Bs, int b_off,
                                   int b_n, int T, float* out, int split_out, hipStream_t s);
// row_off (optional, B + 1 entries): ragged rows - sample b owns rows [row_off[b], row_off[b + 1]) and F only bounds the lengths
int launch_decoder_self_attention(const float* qkv, const int32_t* lengths, const uint32_t* keybits, float* out, int B, int F,
                                  int split_out, hipStream_t s, const int32_t* row_off = nullptr, int shared_qkv = 0);
int launch_self_attention(const float* qkv, const int32_t* lengths, const uint32_t* keybits, float* out, int B, int F, int nheads,
                          int causal, int split_out, hipStream_t s, const int32_t* row_off = nullptr);
int launch_self_attention_split(const float* qkv, const int32_t* lengths, const uint32_t* keybits, float* out, int B, int F,
                                 int nheads, int causal, int split_out, hipStream_t s, const int32_t* row_off = nullptr, int shared_qkv = 0);
int launch_decoder_cross_attention(const float* q, const float* kv, const int32_t* counts, float* out, int B, int F,
                                   int T, int split_out, hipStream_t s);

}  // namespace ladiff
