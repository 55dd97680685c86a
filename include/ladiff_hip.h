/*
 * ladiff_hip.h - C ABI of libladiff_hip.so, the MI355X (gfx950) implementation of the LADiff
 * latent-diffusion sampling hot path.
 *
 * The reference (AlessioSam/LADiff) is pure Python/PyTorch and has no FFI of its own; the
 * "interface each entry point replaces" is therefore a Python call site of the reference, cited
 * per function as `src/...py:line`.  The binding a maintainer adds is a ctypes stub - see
 * INTEGRATION.md and ladiff_amd/_lib.py.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; no torch / C++ types.
 *   - every pointer is a DEVICE pointer (fp32 unless typed otherwise) except those named h_*.
 *   - return 0 = ok, < 0 = argument / shape / workspace error (LADIFF_ERR_*), > 0 = hipError_t.
 *   - work is enqueued on the given stream and scratch comes from the caller's workspace (size from the
 *     *_workspace_bytes query).  The unit kernels, the denoiser / decoder / encoder / CLIP / evaluator entries and
 *     ladiff_diffusion_reverse with sampler == NULL neither allocate, free nor synchronise and are hipGraph-capturable;
 *     calls are re-entrant across distinct (workspace, stream) pairs.  EXCEPTIONS, each stated again at the entry:
 *       ladiff_diffusion_reverse with a sampler  instantiates hipGraphs (captured on a stream the handle creates for itself, replayed
 *                                    on `stream`: a capture on the caller's stream would be invalidated by any other thread's
 *                                    hipEventQuery of an event of that stream - torch.distributed's watchdog does that) and creates
 *                                    two events on first use; when its capture
 *                                    key changes it calls hipStreamSynchronize(stream) before destroying the old graphs and after
 *                                    uploading a new stage table; it is NOT capturable itself (it captures); a pipeline launch
 *                                    takes a process-wide mutex and chains through one event per device, so that two pipeline
 *                                    kernels (which each need every CU) never share the GPU
 *       ladiff_sampler_destroy       hipDeviceSynchronize
 *       ladiff_sampler_loop_ms       hipEventSynchronize on the loop's end event
 *       ladiff_reverse_status        blocking hipMemcpy (use ladiff_reverse_status_offset_bytes + an async copy to poll)
 *       the first pipeline launch on a device  runs a probe kernel on a private stream (hipMalloc / hipFree / stream create)
 *   - tensors are dense row-major fp32.  Arithmetic: w_split == NULL -> fp32-input MFMA, fp32 accumulate (exact fp32
 *     fma chains inside every product; the persistent pipeline kernel of ladiff_diffusion_reverse additionally clears the last
 *     mantissa bit of every activation word it hands from one stage to the next - the bit carries the hand-off's parity tag,
 *     csrc/systolic.hip tag4: a truncation of <= 1 ulp toward zero at each of the 59 hand-offs of a step, in both hand-off
 *     protocols; 1e-5 ... 3e-5 on the decoded frames against the fp64-checked oracle); w_split != NULL (and ladiff_gemm_split / ladiff_self_attention_split / split = 1) -> "f16x3": operands as
 *     fp16 hi + lo pairs (22 significant bits; each half saturates at +-65504 - round 6; rounds 1 - 5 and a -DLADIFF_SPLIT_BF16 build: bf16
 *     pairs, 16 bits, "bf16x3"; ladiff_split_format() tells), three 16-bit MFMAs per product, fp32 accumulate; softmax, LayerNorm statistics, guidance and the
 *     scheduler are fp32 in both.
 *   - weights are passed as an array of device pointers, one per state-dict tensor, in the order
 *     given by ladiff_{denoiser,decoder}_param_name(i) (names = the reference's state-dict keys,
 *     SURVEY.md Appendix A).
 */
#ifndef LADIFF_HIP_H
#define LADIFF_HIP_H

#include <stddef.h>
#include <stdint.h>

/* The library is built with -fvisibility=hidden: only the entries declared here (and, for measurement, in
 * ladiff_hip_debug.h) are exported. */
#ifndef LADIFF_API
#define LADIFF_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ladiff_stream_t; /* hipStream_t */

enum {
    LADIFF_OK = 0,
    LADIFF_ERR_ARG = -1,         /* null pointer / negative size */
    LADIFF_ERR_SHAPE = -2,       /* shape outside what the kernels are built for */
    LADIFF_ERR_WORKSPACE = -3,   /* workspace too small */
    LADIFF_ERR_UNSUPPORTED = -4  /* configuration branch of the reference that is not built */
};

/* activation codes for ladiff_gemm */
enum { LADIFF_ACT_NONE = 0, LADIFF_ACT_RELU = 1, LADIFF_ACT_GELU = 2, LADIFF_ACT_SILU = 3, LADIFF_ACT_QGELU = 4 /* x*sigmoid(1.702x) */,
       LADIFF_ACT_LRELU = 5 /* LeakyReLU(0.2) */ };

#define LADIFF_ABI_VERSION 6
#define LADIFF_LATENT_DIM 256     /* model.latent_dim[-1], config_ladiff_humanml3d.yaml:132 */
#define LADIFF_NUM_HEADS 4        /* configs/modules/denoiser.yaml:7 */
#define LADIFF_NUM_LAYERS 9       /* configs/modules/denoiser.yaml:6, motion_vae.yaml:5 */
#define LADIFF_FF_SIZE 1024       /* configs/modules/denoiser.yaml:5 */
#define LADIFF_TEXT_DIM 768       /* configs/modules/denoiser.yaml:4 */
#define LADIFF_MAX_LATENTS 8      /* MAX_IT <= 8 (shipped: 5, config_ladiff_humanml3d.yaml:58) */
#define LADIFF_MAX_FRAMES 224     /* frames per motion <= 224 (reference MAX_LEN 196, base.yaml:78) */
#define LADIFF_COEF_STRIDE 8      /* floats per scheduler-coefficient row */
#define LADIFF_CLIP_MAX_LAYERS 12  /* CLIP ViT-L/14 text tower: 12 layers, width 768 (= LADIFF_TEXT_DIM), 12 heads, MLP 3072 */
#define LADIFF_CLIP_MAX_POSITIONS 77

LADIFF_API int ladiff_version(void);
/* Element type of the two halves of an S-format pair in THIS build of the library: 1 = IEEE fp16 (the product: x ~ hi + lo carries 22
 * significant bits, saturating at +-65504 per half), 0 = bf16 (rounds 1 - 5 and `-DLADIFF_SPLIT_BF16`: 16 bits, fp32's exponent range).
 * Callers that only pass S-format buffers between entries of this library never need it; tests that decode the format do. */
LADIFF_API int ladiff_split_format(void);
LADIFF_API const char* ladiff_error_string(int code);

/* ------------------------------------------------------------------ weight tables */
LADIFF_API int ladiff_denoiser_num_params(void);
LADIFF_API const char* ladiff_denoiser_param_name(int i); /* state-dict key of LADiffDenoiser, ladiff_denoiser.py:62-123 */
LADIFF_API int ladiff_decoder_num_params(void);
LADIFF_API const char* ladiff_decoder_param_name(int i);  /* keys LADiffVae.decode reads, ladiff_vae.py:334-356 */

/* ------------------------------------------------------------------ unit kernels (parity tests)
 * Y[M,N] = LN?( act(A[M,K] . W[N,K]^T + bias) + res ), A optionally the concat [A | A2] along K.
 * Replaces nn.Linear (+ F.relu / F.gelu / residual / nn.LayerNorm) call sites such as
 * mdiff_transformer.py:62-66, cross_attention.py:79-82 and :408-412.  ln_gamma != NULL needs N == 256. */
LADIFF_API int ladiff_gemm(const float* A, int lda, const float* A2, int lda2, int K1, const float* W, int ldw,
                const float* bias, const float* res, int ldres, const float* ln_gamma,
                const float* ln_beta, float* Y, int ldy, int M, int N, int K, int act,
                ladiff_stream_t stream);

/* Small-M variant used by the denoiser loop (K-resident LDS-DMA tiles, K multiple of 256, N multiple of 4).
 *   K == 256:        Y = act( A . W^T + bias ) + res
 *   K == 512 / 1024: split-K - Y receives K/256 raw partial planes [K/256][M][ldy]; bias/act/res are NOT applied,
 *                    combine them with ladiff_combine_rows. */
LADIFF_API int ladiff_gemm_resident(const float* A, int lda, const float* A2, int lda2, int K1, const float* W, int ldw,
                         const float* bias, const float* res, int ldres, float* Y, int ldy, int M, int N, int K,
                         int act, int split, float* Ys, ladiff_stream_t stream);

/* Large-M f16x3 GEMM of the decoder / encoder / CLIP (128x128 tiles, persistent producer/consumer workgroups):
 *   Y and/or Ys = act( [A | A2] . W^T + bias ) (+ res);  A, A2, W are S-format rows (ladiff_split_rows), K and K1
 *   multiples of 64, N multiple of 128, ldy multiple of 64; Y fp32 and Ys its S-format twin, either may be NULL. */
LADIFF_API int ladiff_gemm_split(const float* A, int lda, const float* A2, int lda2, int K1, const float* W, int ldw,
                      const float* bias, const float* res, int ldres, float* Y, float* Ys, int ldy, int M, int N, int K,
                      int act, ladiff_stream_t stream);

/* f16x3 operand format ("S-format"): a row of K fp32 values (K multiple of 64) is stored in the same K*4 bytes as
 * K/64 blocks of [64 x 16-bit hi | 64 x 16-bit lo], x ~ hi + lo (fp16 halves, or bf16 in a -DLADIFF_SPLIT_BF16 build).  With split = 1 ladiff_gemm_resident reads A, A2 and W in
 * this format and evaluates every product as hi*hi + hi*lo + lo*hi on the 16-bit MFMA with fp32 accumulation
 * (~2^-16 relative error per product, fp32 exponent range); Ys (may be NULL) receives the K == 256 result in
 * S-format, Y (may then be NULL) in fp32.  ladiff_split_rows converts fp32 [R,K] -> S-format [R,K]. */
LADIFF_API int ladiff_split_rows(const float* x, float* y, int R, int K, ladiff_stream_t stream);

/* The decoder layer's feed-forward block as ONE kernel, f16x3 arithmetic (csrc/dec_mlp.hip):
 *   y / ys [M,256] = LN( x + W2 gelu(W1 x + b1) + b2 ), then a second LayerNorm when ln2_gamma != NULL
 * TransformerDecoderLayer.forward_post, cross_attention.py:410-412 (tgt = norm3(tgt + linear2(gelu(linear1(tgt))))) and, on the
 * last layer, decoder.norm (:150-151).  xs = S-format twin of x (the operand; x is the fp32 residual), w1s [1024,256] and
 * w2s [256,1024] S-format (ladiff_split_rows), b1 [1024], b2 [256]; y fp32 and / or ys S-format (either may be NULL). */
LADIFF_API int ladiff_mlp_ln_fused(const float* xs, const float* x, const float* w1s, const float* b1, const float* w2s, const float* b2,
                        const float* ln_gamma, const float* ln_beta, const float* ln2_gamma, const float* ln2_beta, float* y,
                        float* ys, int M, ladiff_stream_t stream);

/* Rows of 256: x = sum of n_planes partial planes [n_planes][M][256] + bias (+ res), then
 *   mode 0: x;   mode 1: LN(x);   mode 2: LN(x) + table[sample row | pad_row] (rows grouped T per sample, sample = row / T, padded when
 *   row % T >= counts[sample % Bs]);   mode 3: SiLU(LN(x) * (1 + table[0:256]) + table[256:512]).
 * Replaces the residual / LayerNorm / StylizationBlock element-wise tails at mdiff_transformer.py:65-66, :160-162. */
LADIFF_API int ladiff_combine_rows(const float* partials, int n_planes, int M, const float* bias, const float* res, int mode,
                        const float* ln_gamma, const float* ln_beta, const float* table, const int32_t* counts,
                        int Bs, int T, int pad_row, float* out, ladiff_stream_t stream);

/* y = LayerNorm(x) over rows of 256 (nn.LayerNorm, eps 1e-5), cross_attention.py:84-85, :150-151 */
LADIFF_API int ladiff_layernorm(const float* x, const float* gamma, const float* beta, float* y, int M,
                     ladiff_stream_t stream);

/* Self-attention core (softmax(QK^T/8 + key mask) V per sample and head) on packed qkv[B*F,768], F <= 224.
 * Key validity: keys < lengths[b], or - when keybits != NULL - bit k of the 256-bit map keybits[b][8] (uint32 words,
 * LSB first).  The nn.MultiheadAttention inside TransformerDecoderLayer.forward_post (cross_attention.py:367-369)
 * and TransformerEncoderLayer.forward_post (:298-300), without the in/out projections. */
LADIFF_API int ladiff_decoder_self_attention(const float* qkv, const int32_t* lengths, const uint32_t* keybits, float* out,
                                  int B, int F, ladiff_stream_t stream);

/* The same core in f16x3 arithmetic (q, k, v and the probabilities as hi + lo pairs, three 16-bit MFMAs per product,
 * fp32 softmax and accumulation) for any number of 64-wide heads: qkv[B*F, 3*64*nheads] packed [q | k | v],
 * out[B*F, 64*nheads] fp32.  lengths and keybits may both be NULL (all keys valid); causal != 0 adds key <= query. */
LADIFF_API int ladiff_self_attention_split(const float* qkv, const int32_t* lengths, const uint32_t* keybits, float* out, int B, int F,
                                 int nheads, int causal, ladiff_stream_t stream);

/* Decoder cross-attention core: q[B*F,256] against the T memory tokens kv[T*B,512] (row = t*B+b,
 * K | V), tokens >= counts[b] masked.  cross_attention.py:373-376. */
LADIFF_API int ladiff_decoder_cross_attention(const float* q, const float* kv, const int32_t* counts, float* out,
                                   int B, int F, int T, ladiff_stream_t stream);

/* ------------------------------------------------------------------ denoiser (LADiffDenoiser.forward)
 * Step-invariant and t-only work is hoisted:
 *   time tables  [n_steps][9 layers][1536] = AdaLN (scale|shift) of ca_block and ffn, K|V of the time token
 *                (tools/embeddings.py:245-305, mdiff_transformer.py:158-160, :308-311)
 *   text cache   emb_proj output, per-layer K|V of the text token, and the c table [9][n_steps][B2+1][256]: the
 *                whole cross-attention block (mdiff_transformer.py:219-247), which with ONE text token adds a
 *                vector that depends on (step, layer, sample) only (ladiff_denoiser.py:193-198).  It is built
 *                from the time tables, so call ladiff_denoiser_time_tables first.                             */
LADIFF_API size_t ladiff_denoiser_tables_floats(int n_steps);
/* n_text = text tokens per prompt.  1 (the CLIP pooled token, mld_clip.py:75-78) is the hoisted form above.  n_text > 1
 * (`clip_hidden` / `bert`, mld_clip.py:80-86) is the literal path in fp32 arithmetic (w_split must be NULL): the text
 * cache then holds emb_proj of every token, their per-layer K|V for the self-attention over [latents | text | time], and
 * per (layer, sample, head) the 64x64 matrix sum_n softmax_n(key) value^T of LinearTemporalCrossAttention
 * (mdiff_transformer.py:235-239), which depends on the text only. */
LADIFF_API size_t ladiff_denoiser_text_cache_floats(int B2, int n_steps, int n_text);
LADIFF_API size_t ladiff_denoiser_workspace_bytes(int B2, int T, int n_steps, int n_text);

/* sinusoid[n_steps,768] = Timesteps(768, flip_sin_to_cos, freq_shift 0)(t) for every step of the schedule
 * (tools/embeddings.py:245-285).  It is a t-only table like the scheduler coefficients; the host may fill it
 * itself (bit-identical to the reference's fp32 ops) or with ladiff_timestep_sinusoid. */
LADIFF_API int ladiff_timestep_sinusoid(const int64_t* timesteps, int n_steps, float* sinusoid, ladiff_stream_t stream);
LADIFF_API int ladiff_denoiser_time_tables(const float* const* w, const float* sinusoid, int n_steps,
                                float* tables, void* ws, size_t ws_bytes, ladiff_stream_t stream);
LADIFF_API int ladiff_denoiser_text_cache(const float* const* w, const float* text_emb /*[B2,n_text,768]*/, int B2, int n_text,
                               const float* tables, int n_steps, float* cache, void* ws, size_t ws_bytes,
                               ladiff_stream_t stream);

/* eps[Bs*dup,T,256] = denoiser(cat([sample]*dup), t = step *d_step of the time tables, text, counts).
 * ladiff_denoiser.py:153-295 (call site ladiff.py:472-485).  counts[Bs] (int32, valid latent rows per
 * prompt, ceil(len/48)) may be NULL = no masking (TEST_EFFICIENCY / max_iter_elements=None).
 * w_split = NULL: fp32-input MFMA everywhere (bit-for-bit fp32 fma chains).  w_split != NULL: the f16x3 matrix path;
 * it is a second pointer table in the same order as w whose >= 2-D entries are the S-format copies of the weight
 * matrices (ladiff_split_rows), the other entries are ignored. */
LADIFF_API int ladiff_denoiser_forward(const float* const* w, const float* const* w_split, const float* tables,
                            const int32_t* d_step, const float* text_cache, int n_text, int n_steps,
                            const float* sample /*[Bs,T,256]*/, int Bs, int dup, int T, const int32_t* counts, float* eps, void* ws,
                            size_t ws_bytes, ladiff_stream_t stream);

/* One LinearTemporalCrossAttention block, literal (mdiff_transformer.py:219-247), for any number of text tokens:
 *   out[B,T,256] = x + proj_out( softmax_d(query(LN x)) . sum_n softmax_n(key(LN_t xf)) value(LN_t xf)^T , emb )
 * x [B,T,256], xf [B,n_text,256] (text tokens already in the latent width), emb [B,256] time embedding per sample,
 * counts [B] valid latent rows (rows >= counts[b] have their query zeroed; NULL = none), `layer` = block index 0..8 in
 * the order input_blocks, middle_block, output_blocks.  Unit entry point of the n_text > 1 path. */
LADIFF_API size_t ladiff_linear_cross_attention_workspace_bytes(int B, int T, int n_text);
LADIFF_API int ladiff_linear_cross_attention(const float* const* w, int layer, const float* x, const float* xf, const float* emb,
                                  const int32_t* counts, int B, int T, int n_text, float* out, void* ws, size_t ws_bytes,
                                  ladiff_stream_t stream);

/* ------------------------------------------------------------------ guidance + scheduler step
 * latents <- step(eps_u + g (eps_c - eps_u)) with one row of coef per step:
 *   {sqrt(a_t), sqrt(1-a_t), k_x0, k_x, k_eps, k_noise, 0, 0}
 *   x0 = (x - sqrt(1-a_t) e) / sqrt(a_t);  x' = k_x0 x0 + k_x x + k_eps e + k_noise z
 * (DDIM: k_x0 = sqrt(a_prev), k_eps = sqrt(1-a_prev-sigma^2), k_noise = sigma; DDPM: posterior mean
 * coefficients and sqrt(variance)).  Replaces ladiff.py:487-492 + diffusers *Scheduler.step. */
LADIFF_API int ladiff_cfg_scheduler_step(const float* eps, float* latents /*[B,T,256] in/out*/, const float* coef,
                              const int32_t* d_step, const float* step_noise /*[n,B,T,256] or NULL*/,
                              float guidance_scale, int cfg, int B, int T, ladiff_stream_t stream);
LADIFF_API int ladiff_advance_step(int32_t* d_step, ladiff_stream_t stream);

/* latents[B,T,256] = noise * valid * sigma  (ladiff.py:380-390, :407) */
LADIFF_API int ladiff_init_latents(const float* noise, const int32_t* counts, float init_noise_sigma, float* latents,
                        int B, int T, ladiff_stream_t stream);
/* z[T,B,256] = permute(latents) with rows >= counts[b] zeroed  (ladiff.py:500, :562-566).  (Inside
 * ladiff_diffusion_reverse the same kernel also reads the pipeline's abort word and writes NaN when the loop was abandoned.) */
LADIFF_API int ladiff_finalize_latents(const float* latents, const int32_t* counts, float* z, int B, int T,
                            ladiff_stream_t stream);

/* ------------------------------------------------------------------ whole reverse loop
 * LADIFF._diffusion_reverse (ladiff.py:333-571, live branch).  `sampler` (from ladiff_sampler_create, or
 * NULL) owns hipGraphs of the per-call prologue and of up to 10 steps (denoiser + guidance + scheduler + step
 * counter) that are captured on first use and replayed; they are re-captured when any argument changes.  The
 * weight tables are identified by a hash over all their pointers plus `weights_generation`, a number the caller
 * bumps whenever it rebuilds a table (a rebuilt table can reuse old addresses).
 *   cfg = 1: classifier-free guidance, text_emb is [2B,1,768] (unconditional half first, ladiff.py:258-264) and the
 *            network runs on cat([latents]*2) (:472-474); cfg = 0: text_emb is [B,1,768], no guidance (:472-490).
 *   counts       [B] latent rows per motion used as the denoiser's key mask and to zero the initial noise
 *                (ladiff.py:379-390; NULL = unmasked, the TEST_EFFICIENCY branch, ladiff_denoiser.py:254)
 *   final_counts [B] rows >= final_counts[b] of the result are zeroed (ladiff.py:559-566, applied in every branch;
 *                NULL = no zeroing)
 *   h_counts     [B] the same numbers as `counts` in HOST memory (or NULL): lets the pipeline loop pack prompts by their
 *                latent count so that padded latent rows are not computed at all (they never influence valid rows: masked as
 *                keys, every other op is per row, and the final zeroing removes them)
 * reuse_time_tables = 1 tells the call that `ws` still holds the time tables of a previous call with the same weights
 * and schedule.
 * Synchronisation / allocation: with sampler == NULL the call only enqueues.  With a sampler it instantiates graphs and creates
 * events on first use, calls hipStreamSynchronize(stream) whenever its capture key changes (before the old graphs are destroyed,
 * after a new stage table is uploaded, before a changed block plan replaces the previous host copy), and a pipeline launch takes
 * a process-wide mutex and waits (on the stream, not the host) for the previous pipeline launch of the device. */
LADIFF_API int ladiff_sampler_create(void** sampler);
LADIFF_API int ladiff_sampler_destroy(void* sampler);
/* How a sampler runs the N steps: 1 (default) = ONE persistent pipeline kernel for the whole loop when the call qualifies
 * (guidance on, one text token, a CU per pipeline stage; both arithmetic modes) - every CU keeps one stage's weights in
 * registers and blocks of prompts flow through the stages (csrc/systolic.hip).  The block geometry is planned per call: 32-row
 * blocks (both guidance branches of three prompts, padded to T latent rows) or LENGTH-AWARE 16-row blocks (one guidance branch of
 * as many prompts as fit with only their valid latent rows; needs h_counts) - whichever the stage-time model predicts faster;
 * 2 / 3 force the 16- / 32-row plan; 0 = one launch per stage, captured in a hipGraph of up to 10 steps. */
LADIFF_API int ladiff_sampler_set_loop(void* sampler, int mode);
/* The block plan ladiff_diffusion_reverse would use for a batch (host arithmetic only, no GPU call): rows per block (16 = the
 * length-aware packing, 32 = padded blocks) and the number of blocks.  h_counts = latent counts on the host or NULL, masked = the
 * call passes device counts, loop_mode 1 / 2 / 3 as ladiff_sampler_set_loop, f16x3 = the call passes w_split, cfg = the call's
 * guidance flag (without guidance: one-branch 16-row blocks; LADIFF_ERR_UNSUPPORTED when such a call has device-only counts - it runs
 * launch-per-stage and has no block plan). */
LADIFF_API int ladiff_reverse_plan(int B, int T, const int32_t* h_counts, int masked, int loop_mode, int f16x3, int cfg, int* rows_per_block,
                        int* n_blocks);
/* Device time of the N-step loop of the sampler's last call (HIP events recorded on the call's stream right around the
 * pipeline kernel, or around the graph replays); blocks until it has finished.  Measurement aid (bench.py). */
LADIFF_API int ladiff_sampler_loop_ms(void* sampler, float* ms);
/* Measurement aids (bench.py --config c3).  A schedule longer than 64 steps runs as several windows (the hoisted cross-attention
 * table is rebuilt per window, the latents carry over): with window timing on, every window's loop launches are bracketed by
 * their own event pair, and ladiff_sampler_window_ms returns their sum and count for the last call (blocks until they have
 * completed) - ladiff_sampler_loop_ms minus that sum is what the table rebuilds between the windows cost.
 * ladiff_sampler_last_loop: whether the last call ran the persistent pipeline kernel (1) or launch-per-stage graphs (0), and
 * the pipeline's block plan (rows per block, blocks; 0 otherwise). */
LADIFF_API int ladiff_sampler_set_window_timing(void* sampler, int on);
LADIFF_API int ladiff_sampler_window_ms(void* sampler, float* loop_ms_sum, int* n_windows);
LADIFF_API int ladiff_sampler_last_loop(void* sampler, int* pipeline, int* rows_per_block, int* n_blocks);
/* Blocking read (hipMemcpy: synchronises the device) of the pipeline kernel's status words of the last call in this workspace:
 * code 0 = completed, 2 = a stage timed out waiting for its producer (info = workgroup) - the loop was abandoned, every later
 * window of the call ended at once and z was filled with NaN; code 0 with info -1 = completed, but the workgroups were not on
 * the XCDs the plan assumed and every hand-off was written through (slower, same results).  The words are cleared once per
 * ladiff_diffusion_reverse call and are sticky over its windows; loop forms other than the pipeline leave them at 0. */
LADIFF_API int ladiff_reverse_status(void* ws, int B, int T, int n_steps, int n_text, int* code, int* info);
/* Where those two uint32 words {code, info} live: byte offset from the workspace base (0 = bad arguments).  A caller that must
 * not block copies the 8 bytes to pinned host memory with an async copy on the call's stream and reads them once an event
 * recorded behind the copy has completed (ladiff_amd/pipeline.py does, and raises / re-runs the call launch-per-stage). */
LADIFF_API size_t ladiff_reverse_status_offset_bytes(int B, int T, int n_steps, int n_text);
/* Fault injection for the abort-path tests, a property of ONE sampler handle: workgroup >= 0 makes that pipeline workgroup leave right
 * after the start-up handshake of every launch of this sampler, so that its consumers time out (-1: off); timeout_ms > 0 replaces the
 * 1.5 s bound of every wait (0: default).  Other samplers of the process are not affected. */
LADIFF_API int ladiff_sampler_set_fault(void* sampler, int workgroup, int timeout_ms);
/* Per-step noise of the stochastic schedulers (DDPM, DDIM with eta > 0) drawn ON THE DEVICE, where it is consumed, instead of read
 * from a [n_steps,B,T,256] tensor (655 MB for 1000 steps x 128 prompts).  Replaces diffusers' `randn_tensor` inside
 * `scheduler.step` (reference call site: ladiff.py:492; configs/modules_novae/scheduler.yaml:16-29).  A value is a pure function of
 * (seed, schedule position, global prompt index = first_prompt + b, latent, column): Philox4x32-10 bits, Box-Muller normals
 * (csrc/noise_gen.h; oracle/ladiff_oracle.py:device_noise is the numpy restatement) - so a batch sharded over ranks or cut into
 * chunks draws what the whole batch would, and a call is reproducible from its seed.
 *   ladiff_sampler_set_noise_generator: with enable != 0, calls of ladiff_diffusion_reverse on this sampler that pass
 *     step_noise = NULL draw from the generator (a non-NULL step_noise still wins); enable = 0: such calls add no noise (as before).
 *   ladiff_noise_fill: the same values as a tensor, out[i][b][t][:] for schedule positions first_step + i (tests, oracle checks,
 *     callers that want to keep the noise). */
LADIFF_API int ladiff_sampler_set_noise_generator(void* sampler, uint64_t seed, uint32_t first_prompt, int enable);
LADIFF_API int ladiff_noise_fill(uint64_t seed, uint32_t first_prompt, int first_step, int n_steps, int B, int T, float* out /*[n_steps,B,T,256]*/,
                      ladiff_stream_t stream);
LADIFF_API size_t ladiff_reverse_workspace_bytes(int B, int T, int n_steps, int n_text);
LADIFF_API int ladiff_diffusion_reverse(void* sampler, const float* const* w, const float* const* w_split /*or NULL*/,
                             uint64_t weights_generation, const float* text_emb /*[2B or B,n_text,768]*/,
                             const float* init_noise /*[B,T,256]*/, const int32_t* counts /*[B] or NULL*/,
                             const int32_t* final_counts /*[B] or NULL*/,
                             const int32_t* h_counts /*HOST copy of counts, or NULL*/,
                             const float* sinusoid /*[n,768]*/, const float* coef /*[n,8]*/,
                             const float* step_noise /*[n,B,T,256] or NULL*/, float guidance_scale,
                             float init_noise_sigma, int cfg, int B, int T, int n_text, int n_steps, float* z /*[T,B,256]*/,
                             void* ws, size_t ws_bytes, int reuse_time_tables, ladiff_stream_t stream);

/* ------------------------------------------------------------------ LA-VAE decoder (LADiffVae.decode)
 * feats[B,F,C] from z[T,B,256]; frames >= lengths[b] come out zero.  ladiff_vae.py:288-362
 * (call site ladiff.py:283).  lengths/counts are int32 device arrays of B entries. */
/* w_split (f16x3 mode): the S-format copies of the weight matrices as for the denoiser.  final_layer.weight needs its C rows only
 * (ABI 4; ABI 3 asked for tables padded to ceil(C / 128) * 128 rows - such tables still work, the extra rows are not read): from 4,096
 * frame rows up the final projection runs on whole 128-column f16x3 tiles, and the library pads the rows it needs itself.  The bias is
 * taken from the fp32 table `w`. */
LADIFF_API size_t ladiff_decoder_workspace_bytes(int B, int F, int T, int C);
LADIFF_API int ladiff_vae_decode(const float* const* w, const float* const* w_split /*or NULL: fp32 MFMA*/, const float* z,
                      const int32_t* lengths, const int32_t* counts, int B, int F, int T, int C, float* feats,
                      void* ws, size_t ws_bytes, ladiff_stream_t stream);
/* The same decode computing ONLY the valid frames of a mixed-length batch (ragged rows): row_off[b] (device int32,
 * B + 1 entries, exclusive prefix sum of lengths, row_off[B] = total_rows) places sample b's frames in the packed row
 * space the decoder works in; the result is scattered to feats[B,F,C] (frames >= lengths[b] zero, as above).  A sample's
 * frames do not depend on the other samples or on padding - padded frames are masked keys (cross_attention.py:367-371)
 * and zeroed at the end (ladiff_vae.py:356-360) - so the output equals ladiff_vae_decode's.  The workspace of
 * ladiff_decoder_workspace_bytes(B, F, T, C) is enough. */
LADIFF_API int ladiff_vae_decode_ragged(const float* const* w, const float* const* w_split /*or NULL*/, const float* z,
                             const int32_t* lengths, const int32_t* counts, const int32_t* row_off, int total_rows, int B,
                             int F, int T, int C, float* feats, void* ws, size_t ws_bytes, ladiff_stream_t stream);

/* The same decode (padded when row_off == NULL, ragged otherwise) captured into a hipGraph on first use and replayed: for batches
 * of few frame rows, where the ~110 launches of a decode are a few microseconds each and the host's launch rate, not the GPU, sets
 * the time (config c1: 8 motions x 60 frames).  `graph` comes from ladiff_decoder_graph_create; the graph is keyed on every pointer
 * argument (z, lengths, counts, row_off, feats, ws, stream), the shapes and the weight tables (hash of their pointers +
 * `weights_generation`, as ladiff_diffusion_reverse) and is re-captured - after a hipStreamSynchronize(stream) - when any of them
 * changes, so callers keep z / feats / ws in persistent buffers.  `stream` must not be the null stream (capture is illegal there).
 * ladiff_decoder_graph_destroy synchronises the device. */
LADIFF_API int ladiff_decoder_graph_create(void** graph);
LADIFF_API int ladiff_decoder_graph_destroy(void* graph);
LADIFF_API int ladiff_vae_decode_graphed(void* graph, const float* const* w, const float* const* w_split /*or NULL*/, uint64_t weights_generation,
                              const float* z, const int32_t* lengths, const int32_t* counts, const int32_t* row_off /*or NULL*/,
                              int total_rows, int B, int F, int T, int C, float* feats, void* ws, size_t ws_bytes,
                              ladiff_stream_t stream);

/* ------------------------------------------------------------------ LA-VAE encoder
 * LADiffVae.encode, ladiff_vae.py:162-286 (call sites ladiff.py:269, :324, :1096): features[B,F,C] ->
 * mu, std, latent, each [T,B,256] (sequence-first like the reference); latent = mu + std * eps with rows >= counts[b]
 * zeroed; eps[T,B,256] stands in for the draw inside Normal.rsample().  F + 2T <= 224.  Weight table as for the
 * decoder: ladiff_encoder_param_name(i) lists the state-dict keys. */
LADIFF_API int ladiff_encoder_num_params(void);
LADIFF_API const char* ladiff_encoder_param_name(int i);
LADIFF_API size_t ladiff_encoder_workspace_bytes(int B, int F, int T, int C);
LADIFF_API int ladiff_vae_encode(const float* const* w, const float* const* w_split /*or NULL*/, const float* features,
                      const int32_t* lengths, const int32_t* counts, const float* eps, int B, int F, int T, int C,
                      float* mu, float* std, float* latent, void* ws, size_t ws_bytes, ladiff_stream_t stream);

/* ------------------------------------------------------------------ CLIP text encoder (the caller side)
 * MldTextEncoder.forward, mld_clip.py:51-86, "clip" branch (call sites ladiff.py:265, :1048, :1144):
 * text_model.get_text_features(input_ids) of transformers' CLIPModel -> out[B,768] (the reference then unsqueezes to
 * [B,1,768]).  ids[B,S] int64 token ids as the CLIP tokenizer emits them (padding="max_length", S <= 77; tokenising is
 * host string work and stays with the caller).  Only the first L <= S positions are evaluated: under the causal mask
 * the pooled EOS row (argmax of the ids) does not depend on later positions, so any L > max_b argmax(ids[b]) gives the
 * same result as L = S.  The pointer table lists LADIFF_CLIP_MAX_LAYERS layers; a model with n_layers < 12 fills the
 * first 5 + 16 * n_layers entries (rest may be NULL).  vocab = rows of the token-embedding table. */
LADIFF_API int ladiff_clip_num_params(void);
LADIFF_API const char* ladiff_clip_param_name(int i);     /* keys of transformers.CLIPModel.state_dict() (text side) */
LADIFF_API size_t ladiff_clip_workspace_bytes(int B, int L);
LADIFF_API int ladiff_clip_text_encode(const float* const* w, const float* const* w_split /*or NULL*/, int n_layers, int vocab,
                            const int64_t* ids, int B, int S, int L, float* out, void* ws, size_t ws_bytes,
                            ladiff_stream_t stream);
/* The same with RAGGED rows: prompt b is evaluated at its OWN positions 0 .. eos_b only (same argument as L above, per prompt: nothing
 * behind a prompt's EOS reaches its pooled row), 2.3x fewer rows than the padded batch for prompts of 1 .. 30 words.  seq_len[B] =
 * eos_b + 1 (the caller computes it from the ids as the reference's argmax does, mld_clip.py:75-78 -> CLIPTextTransformer), row_off[B+1]
 * = exclusive prefix sums of seq_len (row_off[B] = total_rows), row_seq[total_rows] = the prompt of each row; all int32 on the device;
 * L = max seq_len.  Same result as ladiff_clip_text_encode within the arithmetic mode's rounding (the row tiling of the GEMMs differs).
 * The three arrays must be mutually consistent (only total_rows is range-checked on the host): inconsistent ones give wrong embeddings,
 * never an access outside ids / the position table (the kernels clamp prompt and position indices). */
LADIFF_API size_t ladiff_clip_workspace_bytes_ragged(int B, int total_rows);
LADIFF_API int ladiff_clip_text_encode_ragged(const float* const* w, const float* const* w_split /*or NULL*/, int n_layers, int vocab,
                            const int64_t* ids, int B, int S, int L, const int32_t* seq_len, const int32_t* row_off,
                            const int32_t* row_seq, int total_rows, float* out, void* ws, size_t ws_bytes, ladiff_stream_t stream);

/* ------------------------------------------------------------------ T2M evaluator encoders (evaluation)
 * The three frozen networks `t2m_eval` runs to get the embeddings of the TM2T metrics (ladiff.py:1264-1271), fp32:
 *   movement: MovementConvEncoder.forward, t2m_motionenc.py:21-25: the first Cin columns of feats[B,F,ld] (the caller
 *             drops the 4 foot-contact columns, ladiff.py:1264: Cin = nfeats - 4, ld = nfeats) -> out[B, (F/2)/2, 512].
 *   motion:   MotionEncoderBiGRUCo.forward, t2m_motionenc.py:51-64: movements[B,T,512], m_lens[B] (= lengths / 4,
 *             1 <= m_lens <= T; the reference's pack_padded_sequence wants them sorted descending, this entry does not
 *             care) -> out[B,512].
 *   text:     TextEncoderBiGRUCo.forward, t2m_textenc.py:32-48: word_embs[B,L,300], pos_onehot[B,L,15], cap_lens[B]
 *             -> out[B,512].
 * Pointer tables: ladiff_t2m_*_param_name(i) are the keys of the `movement_encoder` / `motion_encoder` / `text_encoder`
 * state dicts of the evaluator checkpoint (ladiff.py:205-212). */
LADIFF_API int ladiff_t2m_movement_num_params(void);
LADIFF_API const char* ladiff_t2m_movement_param_name(int i);
LADIFF_API int ladiff_t2m_motion_num_params(void);
LADIFF_API const char* ladiff_t2m_motion_param_name(int i);
LADIFF_API int ladiff_t2m_text_num_params(void);
LADIFF_API const char* ladiff_t2m_text_param_name(int i);
LADIFF_API size_t ladiff_t2m_movement_workspace_bytes(int B, int F, int Cin);
LADIFF_API size_t ladiff_t2m_motion_workspace_bytes(int B, int T);
LADIFF_API size_t ladiff_t2m_text_workspace_bytes(int B, int L);
LADIFF_API int ladiff_t2m_movement_encode(const float* const* w, const float* feats, int ld, int B, int F, int Cin, float* out,
                               void* ws, size_t ws_bytes, ladiff_stream_t stream);
LADIFF_API int ladiff_t2m_motion_encode(const float* const* w, const float* movements, const int32_t* m_lens, int B, int T, float* out,
                             void* ws, size_t ws_bytes, ladiff_stream_t stream);
LADIFF_API int ladiff_t2m_text_encode(const float* const* w, const float* word_embs, const float* pos_onehot, const int32_t* cap_lens,
                           int B, int L, float* out, void* ws, size_t ws_bytes, ladiff_stream_t stream);

/* ------------------------------------------------------------------ feats2joints (the step after the path)
 * joints[B,F,njoints,3] = recover_from_ric(feats * std + mean): HumanML3DDataModule.feats2joints
 * (data/HumanML3D.py:44-48, data/Kit.py:48-53; motion_process.py:362-381, :415-430).  feats [B,F,C] as produced by
 * ladiff_vae_decode, mean/std [C] (device), C = 263 with 22 joints or 251 with 21 joints, F <= 256. */
LADIFF_API int ladiff_feats2joints(const float* feats, const float* mean, const float* std, int B, int F, int C, int njoints,
                        float* joints, ladiff_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LADIFF_HIP_H */
