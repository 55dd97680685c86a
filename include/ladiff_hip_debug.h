/*
 * ladiff_hip_debug.h - measurement and test switches of libladiff_hip.so.  NOT part of the product interface
 * (include/ladiff_hip.h): nothing a caller of the sampling path needs is declared here.  Every switch is a
 * process-wide atomic; every accepted value gives the same results within the tests' tolerances (the pipeline
 * switches: the same bits).  Used by tests/ and scripts/ only.
 */
#ifndef LADIFF_HIP_DEBUG_H
#define LADIFF_HIP_DEBUG_H

#include "ladiff_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Measurement switch (process-wide): waves per SIMD of the stage workgroups of the 16-row plan, 2 (default: 512-thread workgroups,
 * each stage's weight slice split over the two waves of a SIMD) or 1 (256 threads). */
LADIFF_API int ladiff_debug_set_stage_waves(int waves_per_simd);
/* Measurement switch (process-wide): how the eight-wave stages of the 16-row plan hand a block's rows to the next stage.
 * 1 (default) = the rows carry a parity tag in the last mantissa bit of every word and a consumer loads them until all its words
 * show the parity of the step (no drain, no flag, no separate poll; csrc/systolic.hip, tag4); 0 = the flag protocol (write-through
 * or XCD-local stores, drain, barrier, one epoch word per producer, polled by every consumer wave).  Both give the same bits. */
LADIFF_API int ladiff_debug_set_handoff(int tagged);
/* Measurement switch (process-wide; takes effect for stage tables built afterwards, i.e. for new samplers): how a layer's workgroups of
 * the 16-row pipeline plan that hold no MLP slice are dealt.  0 (default) = one OUT workgroup, STYL as two groups on alternating blocks x
 * two row parts (255 workgroups); 1 = OUT as two groups on alternating blocks, STYL as one group x two row parts (246).  Same results. */
LADIFF_API int ladiff_debug_set_stage_plan(int v);
/* Measurement switch (process-wide): stage types of the tagged pipeline whose waves rest `len` x ~60 ns between two polls of rows that
 * are not there yet (mask bits: 1 LIN, 2 RED2, 4 STYL, 8 FFN, 16 / 32 the loader waves of QKV / OUT, 64 SKIP).  Same results; measured: no shape moves by
 * more than 1 % (scripts/pause_ab.py) - the loop is not bound by poll traffic. */
LADIFF_API int ladiff_debug_set_poll_pause(int mask, int len);
/* Stage types (bits 1 LIN, 2 RED2, 4 STYL, 8 FFN, 64 SKIP) whose workgroups idle `len` x ~60 ns after every block before they look for the
 * next one's rows (process-wide).  mask = -1 (default): chosen per launch - LIN and FFN, len 4, in launches of <= 60 blocks (a block's
 * trip through the stages bounds the step there, and eight workgroups polling the lines a critical-path stage is still storing to do not
 * make it faster: loop -2.2 % at 32 ... 64 prompts and at mixed-length batches of 100 / 128, profiles/r4/14_*), nobody in larger ones.
 * mask = 0: nobody, whatever the size.  Same results. */
LADIFF_API int ladiff_debug_set_stage_delay(int mask, int len);
/* Pacing of the tagged pipeline's polling (process-wide): a stage that waited W for a block's rows sleeps eighths / 8 x W before it
 * starts to poll for the next block's; mask: the stage types that do (only STYL, bit 4, has the code compiled in).  Built-in (until this is called): 4, 4 in
 * launches above 60 blocks - the STYL workgroups (4 per layer, each poll = 72 KB of the FFN stages' partial rows) stop loading the lines
 * the busiest stage type is storing to for half of their wait: loop kernel -3 % at 128 and 256 prompts (scripts/pause_ab.py,
 * profiles/r4/12_*) - and nobody in smaller ones (pacing costs 0.4 % there, profiles/r6/12_*).  A call fixes the values for every
 * launch size (eighths = -1: back to the built-in choice); 0, 0 switches pacing off.  Same results. */
LADIFF_API int ladiff_debug_set_pacing(int eighths, int mask);
/* Measurement switch (process-wide, read when a sampler builds its stage table): 1 (default) = the pipeline stages are dealt to
 * the XCDs in chain order and a stage whose readers share its XCD hands its rows over through that XCD's L2 (plain stores);
 * 0 = every hand-off writes through to the memory side, stages in table order; 2 = as 1, but one workgroup of every launch
 * reports a placement that disagrees with the others (test aid: the launch must then agree to write through everywhere and still
 * give the same bits; ladiff_reverse_status reports code 0, info -1 for such a launch). */
LADIFF_API int ladiff_debug_set_xcd_local(int on);
/* Measurement switch (process-wide): 1 (default) = the decoder's feed-forward block runs as the fused kernel of csrc/dec_mlp.hip in
 * f16x3 mode from 10,000 frame rows up, 2 = at every size, 0 = as linear1 GEMM + linear2 GEMM + LayerNorm row kernel (the round-2
 * path; same arithmetic per product).  + 4: decodes of fewer than 4,096 frame rows keep the large-M GEMM kernels instead of the
 * small-M ones (the round-2 routing).  + 8: final_layer on the fp32-input kernel in f16x3 mode too (the round-2 path).  + 16: the
 * decoder's self-attention as in_proj GEMM + attention kernel (two launches, q | k | v rows through memory) instead of the kernel
 * that computes its head's q | k | v itself (csrc/dec_qkv_attn.hip; default from 4,096 frame rows up); + 32: that kernel at every size.
 * + 64: the self-attention out_proj GEMM and the cross-attention row kernel as two launches (x + out_proj(att) through memory) instead
 * of the one kernel that keeps out_proj's weight in registers (csrc/dec_cross.hip; default from 4,096 frame rows up in f16x3 mode). */
LADIFF_API int ladiff_debug_set_decoder_fusion(int on);
/* Measurement switch (process-wide) of the fused feed-forward kernel's form: 0 (default) = chosen by the row count, 1 = 128-row
 * workgroups of eight waves x 16 rows, 2 = 64-row workgroups of four waves x 16 rows, 3 = 128-row workgroups of four waves x 32 rows.
 * Every accepted value gives the same result; anything else returns LADIFF_ERR_ARG.  (The timing builds of rounds 3 - values 11 .. 17 and
 * 21 .. 26, kernels with one ingredient removed whose results are garbage - are not in this library: they are instantiated in the
 * diagnostic twin libladiff_hip_stamps.so only, `python -m ladiff_amd.build --stamps`, for scripts/mlp_speed.py and attn_speed.py.) */
LADIFF_API int ladiff_debug_set_mlp_variant(int v);

/* Block-count thresholds of the tagged pipeline's polling policy (process-wide; csrc/systolic.hip, launch_systolic_loop): stages may
 * request the next block's rows early in launches of >= look_ahead_from blocks; the LIN / FFN workgroups rest after every block in
 * launches of <= small_upto blocks.  -1 keeps the built-in value.  Same results. */
LADIFF_API int ladiff_debug_set_loop_thresholds(int look_ahead_from, int small_upto);
/* The graph re-instantiation rule, process-wide - samplers AND decode graphs (csrc/api.hip: an older graph exec is never replayed after a newer instantiation):
 * 1 (default) on, 0 off - tests/test_gpu_stress.py replays old execs on purpose. */
LADIFF_API int ladiff_debug_set_graph_epoch_rule(int on);
/* Number of hipGraph instantiations the process has made so far (prologue + step graphs of all samplers): tests assert that a repeated
 * call re-instantiates nothing. */
LADIFF_API int ladiff_debug_graph_instantiations(void);

#ifdef LADIFF_STAMPS
/* diagnostic twin (libladiff_hip_stamps*.so) only: in-kernel timeline buffers and timing probes with garbage results */
LADIFF_API void ladiff_debug_set_stamps(unsigned long long* p);
LADIFF_API void ladiff_debug_set_sys_stamps(unsigned long long* p);
LADIFF_API void ladiff_debug_set_probe(int v);
#endif

#ifdef __cplusplus
}
#endif
#endif
