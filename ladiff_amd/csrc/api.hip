// extern "C" surface of libladiff_hip.so (declared in include/ladiff_hip.h).
#include <algorithm>
#include <cstdlib>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "model.h"
#include "../../include/ladiff_hip_debug.h"

using namespace ladiff;

namespace {

inline hipStream_t S(ladiff_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

template <class W>
bool load_weights(W& dst, const float* const* ptrs) {
    constexpr int n = sizeof(W) / sizeof(const float*);
    if (ptrs == nullptr) return false;
    for (int i = 0; i < n; ++i)
        if (ptrs[i] == nullptr) return false;
    std::memcpy(&dst, ptrs, sizeof(W));
    return true;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- reverse-loop workspace carve-up (floats)
struct ReverseWs {
    float *tables, *cache, *latents, *eps, *fwd, *sys, *cws;
    int32_t* d_step;
    size_t fwd_floats, cws_floats, total_bytes, sys_off;      // sys_off: floats from the workspace base to `sys`
    int window;                            // steps whose c-table rows are resident at a time
};
// The hoisted cross-attention table is [9][steps][2B+1][256] floats: 118 MB for 50 steps at B = 128, but 2.4 GB for a
// 1000-step DDPM schedule.  Long schedules are run window by window (the largest divisor of n_steps that is <= 64 and a
// multiple of the graph unroll), the table rebuilt before each window from the per-layer LN(value) rows kept in the cache.
int reverse_window(int n) {
    if (n <= 64) return n;
    int best = 0;
    for (int w = 64; w >= 10; --w)
        if (n % w == 0 && w % 10 == 0) { best = w; break; }
    return best > 0 ? best : n;
}
ReverseWs carve_reverse(void* ws, int B, int T, int n, int ntxt = 1) {
    ReverseWs r;
    const int B2 = 2 * B;
    size_t off = 0;
    auto take = [&](size_t floats) { float* p = ws ? reinterpret_cast<float*>(ws) + off : nullptr; off += align_up(floats, 64); return p; };
    r.d_step = reinterpret_cast<int32_t*>(take(64));
    r.tables = take(den_tables_floats(n));
    r.window = reverse_window(n);
    r.cache = take(den_text_cache_floats(B2, r.window, ntxt));
    r.latents = take((size_t)B * T * D);
    r.eps = take((size_t)B2 * T * D);
    size_t pre = (size_t)n * D * 3;                                    // time-table scratch
    const size_t txt = den_text_ws_floats(B2, 1, ntxt);                // text-cache scratch (static part)
    if (txt > pre) pre = txt;
    r.fwd_floats = den_forward_ws_floats(B2, T);
    if (pre > r.fwd_floats) r.fwd_floats = pre;
    r.fwd = take(r.fwd_floats);
    r.sys_off = off;
    r.sys = take(sys_ws_floats(B, T));                                 // block buffers, flags and stage table of the pipeline loop
    r.cws_floats = (size_t)NL * r.window * (B2 + 1) * D;               // scratch of the c-table builder (all layers' input rows)
    r.cws = take(r.cws_floats);
    r.total_bytes = off * sizeof(float);
    return r;
}

struct Sampler {
    hipGraphExec_t exec = nullptr;
    hipGraphExec_t setup = nullptr;       // per-call prologue (text cache, initial latents, counter reset, first network input)
    int unroll = 1;                       // denoiser steps captured per graph launch
    int loop_mode = 1;                    // 1: pick per call, 2: 16-row length-aware blocks, 3: 32-row blocks
    std::vector<unsigned char> blocks;    // host copy of the block descriptors last uploaded (geometry of the previous call)
    int plan_mr = 0, plan_nb = 0;
    int loop = 1;                         // 1: persistent pipeline kernel when the call qualifies (systolic.hip), 0: launch per stage
    std::vector<unsigned char> stages;    // host copy of the pipeline's stage table (source of the upload)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // bracket the N-step loop (pipeline kernel or graph replays) of the last call
    bool time_windows = false;            // measurement aid: one event pair per window of the schedule (ladiff_sampler_set_window_timing)
    std::vector<hipEvent_t> wev;          // [2 i], [2 i + 1]: around the loop launches of window i of the last call
    int n_windows = 0;
    int last_pipeline = 0;                // the last call ran the persistent pipeline kernel (1) or launch-per-stage graphs (0)
    // fault injection for the abort-path tests (ladiff_sampler_set_fault): THIS sampler's pipeline launches lose one workgroup right after
    // the start-up handshake and bound their waits; -1 / 0 = none / the default bound.  A field of the handle, not of the process.
    int fault_wg = -1;
    unsigned long long timeout_ticks = 0;
    // per-step noise drawn on the device (ladiff_sampler_set_noise_generator): used by the calls that pass step_noise = NULL
    NoiseGen gen = NoiseGen{0u, 0u, 0u, 0};
    // capture key: a graph bakes pointers, shapes and scalars into its kernel nodes.  The weight tables are identified by
    // a hash over EVERY pointer of both tables plus the caller's generation id (bumped whenever a table is rebuilt), not
    // by the address of the host array (which a rebuilt table can land on again).
    const void* key_ptrs[9] = {nullptr};
    int key_ints[4] = {0};
    unsigned key_gen_noise[4] = {0u, 0u, 0u, 0u};
    float key_f[2] = {0.f, 0.f};
    uint64_t key_hash = 0, key_gen = 0;
    uint64_t epoch = 0;                   // g_graph_epoch when these graphs were instantiated (see there)
    std::vector<hipGraphExec_t> retired;  // replaced while a launch of them could still be queued: destroyed at the next drain
    // Graphs are CAPTURED on a stream of the handle's own and replayed on the caller's (round 5).  While a stream captures, a
    // hipEventQuery of any event that belongs to it is refused and invalidates the capture - and torch.distributed's watchdog thread
    // polls the end events of synchronous collectives, which run on the caller's CURRENT stream: a capture on that stream died about
    // once in fifteen bench runs under torchrun (profiles/r5/26_*).  Nobody else holds events of this stream.
    hipStream_t cap = nullptr;
    int cap_dev = -1;                     // the device `cap` was created on
};
// The capture stream belongs to the device that was current when it was created; a handle that is later used with another device
// current gets a new one (the old graphs hold that device's pointers and are rebuilt by their key anyway).
static int capture_stream(hipStream_t* cap, int* cap_dev) {
    int dev = 0;
    LADIFF_HIP(hipGetDevice(&dev));
    if (*cap != nullptr && *cap_dev != dev) { (void)hipStreamDestroy(*cap); *cap = nullptr; }
    if (*cap == nullptr) { LADIFF_HIP(hipStreamCreateWithFlags(cap, hipStreamNonBlocking)); *cap_dev = dev; }
    return 0;
}

// Graph replay and the round-3 memory fault.  Seen on ROCm 7.2 / MI355X (scripts/repro_seq.py, scripts/repro_graph.py): the graphs of one
// sampler, replayed after two OTHER samplers had instantiated theirs and a blocking hipMemcpy had run in between, faulted at a wild
// address (MEMORY_APERTURE_VIOLATION / an address in the host heap's range), every captured pointer still alive.  Round 4 bisected it
// (profiles/r4/06_*): with the rule below switched off the fault reproduces every time; it goes away when the prologue graph's ONE
// memset node (hipMemsetAsync of the 16-byte step counter) is issued outside the graph, and stays away with every KERNEL node of both
// graphs replayed from the old execs.  So: an older exec's MEMSET NODE is what the runtime replays wrongly after newer instantiations
// - kernel nodes (by-value argument blocks up to 3.8 KB, 300 nodes) are fine, also in a library-free program
// (scripts/repro_graph_args.hip: clean in every configuration, the memset-node case included - the trigger needs more than that
// program has, and was not reduced further).  Fix: NOTHING captured by this library is a memset node any more (launch_zero_fill
// kernels: the step counter in the prologue graph, the ragged decode's output clear); tests/test_gpu_pipeline.py replays old execs on
// purpose (LADIFF_GRAPH_EPOCH_OFF) and gets identical bits.  The rule stays as a second line, cheap (a few hundred microseconds when
// samplers alternate): a graph is replayed only while it is the newest instantiation of THIS library; instantiations by other
// components of the process (torch CUDA graphs, RCCL) do not count - they were never implicated (scripts/repro_graph.py 'graphs').
std::atomic<uint64_t> g_graph_epoch{0};
std::atomic<int> g_graph_epoch_rule{1};       // ladiff_debug_set_graph_epoch_rule
std::atomic<int> g_graph_instantiations{0};   // ladiff_debug_graph_instantiations

void drain_retired(Sampler* sp) {            // call with the stream drained
    for (hipGraphExec_t g : sp->retired) (void)hipGraphExecDestroy(g);
    sp->retired.clear();
}

uint64_t hash_ptrs(const float* const* p, int n, uint64_t h) {            // FNV-1a over the pointer values
    for (int i = 0; i < n; ++i) {
        uint64_t v = reinterpret_cast<uint64_t>(p[i]);
        for (int b = 0; b < 8; ++b) { h ^= (v >> (8 * b)) & 0xff; h *= 1099511628211ull; }
    }
    return h;
}

}  // namespace

#ifdef LADIFF_STAMPS
static unsigned long long* g_stamps = nullptr;
#endif

extern "C" {

int ladiff_version(void) { return LADIFF_ABI_VERSION; }
int ladiff_split_format(void) {
#ifndef LADIFF_SPLIT_BF16
    return 1;
#else
    return 0;
#endif
}

const char* ladiff_error_string(int code) {
    switch (code) {
        case LADIFF_OK: return "ok";
        case LADIFF_ERR_ARG: return "invalid argument (null pointer or negative size)";
        case LADIFF_ERR_SHAPE: return "shape not supported by the gfx950 kernels";
        case LADIFF_ERR_WORKSPACE: return "workspace too small";
        case LADIFF_ERR_UNSUPPORTED: return "configuration branch not built";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown ladiff error";
    }
}

int ladiff_denoiser_num_params(void) { return DEN_NPARAMS; }
const char* ladiff_denoiser_param_name(int i) {
    const auto& n = denoiser_param_names();
    return (i >= 0 && i < (int)n.size()) ? n[i].c_str() : nullptr;
}
int ladiff_decoder_num_params(void) { return DEC_NPARAMS; }
const char* ladiff_decoder_param_name(int i) {
    const auto& n = decoder_param_names();
    return (i >= 0 && i < (int)n.size()) ? n[i].c_str() : nullptr;
}

// ------------------------------------------------------------------ unit kernels
int ladiff_gemm(const float* A, int lda, const float* A2, int lda2, int K1, const float* W, int ldw, const float* bias,
                const float* res, int ldres, const float* ln_gamma, const float* ln_beta, float* Y, int ldy, int M,
                int N, int K, int act, ladiff_stream_t stream) {
    LADIFF_CHECK_ARG(A && W && Y && M >= 0 && N > 0 && K > 0);
    GemmArgs g;
    g.A = A; g.lda = lda; g.A2 = A2; g.lda2 = lda2; g.K1 = A2 ? K1 : K; g.W = W; g.ldw = ldw; g.bias = bias;
    g.res = res; g.ldres = ldres; g.ln_g = ln_gamma; g.ln_b = ln_beta; g.Y = Y; g.ldy = ldy; g.M = M; g.N = N; g.K = K;
    g.act = act;
    return launch_gemm(g, S(stream));
}

int ladiff_gemm_resident(const float* A, int lda, const float* A2, int lda2, int K1, const float* W, int ldw,
                         const float* bias, const float* res, int ldres, float* Y, int ldy, int M, int N, int K, int act,
                         int split, float* Ys, ladiff_stream_t stream) {
    LADIFF_CHECK_ARG(A && W && (Y || Ys) && M >= 0 && N > 0 && K > 0);
    if (M == 0) return 0;
    KrArgs g;
    g.A = A; g.lda = lda; g.A2 = A2; g.lda2 = lda2; g.K1 = A2 ? K1 : K; g.W = W; g.ldw = ldw; g.bias = bias;
    g.res = res; g.ldres = ldres; g.Y = Y; g.ldy = ldy; g.M = M; g.N = N; g.K = K; g.act = act;
    g.split = split; g.Ys = Ys;
#ifdef LADIFF_STAMPS
    g.stamps = g_stamps;
#endif
    return launch_gemm_kr(g, S(stream));
}

int ladiff_gemm_split(const float* A, int lda, const float* A2, int lda2, int K1, const float* W, int ldw, const float* bias,
                      const float* res, int ldres, float* Y, float* Ys, int ldy, int M, int N, int K, int act,
                      ladiff_stream_t stream) {
    LADIFF_CHECK_ARG(A && W && (Y || Ys) && M >= 0 && N > 0 && K > 0);
    if (M == 0) return 0;
    GemmArgs g;
    g.A = A; g.lda = lda; g.A2 = A2; g.lda2 = lda2; g.K1 = A2 ? K1 : K; g.W = W; g.ldw = ldw; g.bias = bias;
    g.res = res; g.ldres = ldres; g.Y = Y; g.Ys = Ys; g.ldy = ldy; g.M = M; g.N = N; g.K = K; g.act = act; g.split = 1;
    return launch_gemm(g, S(stream));
}

#ifdef LADIFF_STAMPS
void ladiff_debug_set_stamps(unsigned long long* p) { g_stamps = p; }   // diagnostic builds only
void ladiff_debug_set_sys_stamps(unsigned long long* p) { ladiff::g_sys_stamps = p; }
void ladiff_debug_set_probe(int v) { ladiff::g_sys_probe = v; }        // timing probes of the pipeline kernel: garbage results
#endif

int ladiff_combine_rows(const float* partials, int n_planes, int M, const float* bias, const float* res, int mode,
                        const float* ln_gamma, const float* ln_beta, const float* table, const int32_t* counts, int Bs,
                        int T, int pad_row, float* out, ladiff_stream_t stream) {
    LADIFF_CHECK_ARG(partials && out && n_planes > 0 && M >= 0 && Bs > 0 && T > 0);
    if (mode != RED_PLAIN && mode != RED_LN && mode != RED_LN_ADD && mode != RED_LN_MOD) return LADIFF_ERR_ARG;
    if (mode != RED_PLAIN && !(ln_gamma && ln_beta)) return LADIFF_ERR_ARG;
    if ((mode == RED_LN_ADD || mode == RED_LN_MOD) && !table) return LADIFF_ERR_ARG;
    if (M == 0) return 0;
    return launch_reduce_rows(partials, n_planes, M, bias, res, mode, ln_gamma, ln_beta, table, 0, nullptr, counts, Bs, T,
                              pad_row, 0, out, nullptr, S(stream));
}

int ladiff_layernorm(const float* x, const float* gamma, const float* beta, float* y, int M, ladiff_stream_t stream) {
    LADIFF_CHECK_ARG(x && gamma && beta && y && M >= 0);
    return launch_layernorm(x, gamma, beta, y, M, S(stream));
}

int ladiff_timestep_sinusoid(const int64_t* timesteps, int n, float* out, ladiff_stream_t stream) {
    LADIFF_CHECK_ARG(timesteps && out && n >= 0);
    if (n == 0) return 0;
    return launch_sinusoid(timesteps, n, out, S(stream));
}

int ladiff_decoder_self_attention(const float* qkv, const int32_t* lengths, const uint32_t* keybits, float* out, int B,
                                  int F, ladiff_stream_t stream) {
    LADIFF_CHECK_ARG(qkv && (lengths || keybits) && out && B >= 0);
    return launch_decoder_self_attention(qkv, lengths, keybits, out, B, F, 0, S(stream));
}

int ladiff_self_attention_split(const float* qkv, const int32_t* lengths, const uint32_t* keybits, float* out, int B, int F,
                                 int nheads, int causal, ladiff_stream_t stream) {
    LADIFF_CHECK_ARG(qkv && out && B >= 0);
    return launch_self_attention_split(qkv, lengths, keybits, out, B, F, nheads, causal, 0, S(stream));
}

int ladiff_decoder_cross_attention(const float* q, const float* kv, const int32_t* counts, float* out, int B, int F,
                                   int T, ladiff_stream_t stream) {
    LADIFF_CHECK_ARG(q && kv && out && B >= 0 && F >= 0);
    return launch_decoder_cross_attention(q, kv, counts, out, B, F, T, 0, S(stream));
}

// ------------------------------------------------------------------ denoiser
size_t ladiff_denoiser_tables_floats(int n_steps) { return den_tables_floats(n_steps); }
size_t ladiff_denoiser_text_cache_floats(int B2, int n_steps, int n_text) { return den_text_cache_floats(B2, n_steps, n_text); }
size_t ladiff_denoiser_workspace_bytes(int B2, int T, int n_steps, int n_text) {
    size_t f = den_forward_ws_floats(B2, T);
    const size_t a = (size_t)n_steps * D * 3, b = den_text_ws_floats(B2, n_steps, n_text);
    if (a > f) f = a;
    if (b > f) f = b;
    return f * sizeof(float);
}

int ladiff_denoiser_time_tables(const float* const* w, const float* sinusoid, int n_steps, float* tables, void* ws,
                                size_t ws_bytes, ladiff_stream_t stream) {
    DenoiserW W;
    LADIFF_CHECK_ARG(load_weights(W, w) && sinusoid && tables && ws && n_steps > 0);
    return denoiser_time_tables(W, sinusoid, n_steps, tables, (float*)ws, ws_bytes / sizeof(float), S(stream));
}

int ladiff_denoiser_text_cache(const float* const* w, const float* text_emb, int B2, int n_text, const float* tables, int n_steps,
                               float* cache, void* ws, size_t ws_bytes, ladiff_stream_t stream) {
    DenoiserW W;
    LADIFF_CHECK_ARG(load_weights(W, w) && text_emb && tables && cache && ws && B2 > 0 && n_steps > 0 && n_text >= 1);
    return denoiser_text_cache(W, text_emb, B2, tables, n_steps, cache, (float*)ws, ws_bytes / sizeof(float), S(stream), n_text);
}

int ladiff_denoiser_forward(const float* const* w, const float* const* w_split, const float* tables, const int32_t* d_step,
                            const float* text_cache, int n_text, int n_steps, const float* sample, int Bs, int dup, int T,
                            const int32_t* counts, float* eps, void* ws, size_t ws_bytes, ladiff_stream_t stream) {
    DenoiserW W, WS;
    LADIFF_CHECK_ARG(load_weights(W, w) && tables && d_step && text_cache && sample && eps && ws && Bs > 0 && dup > 0 && n_steps > 0 &&
                     n_text >= 1);
    if (w_split != nullptr) LADIFF_CHECK_ARG(load_weights(WS, w_split));
    return denoiser_forward(W, w_split ? &WS : nullptr, tables, d_step, text_cache, n_steps, sample, Bs, dup, T, counts, eps, (float*)ws,
                            ws_bytes / sizeof(float), S(stream), 0, -1, 0, n_text);
}

size_t ladiff_linear_cross_attention_workspace_bytes(int B, int T, int n_text) {
    return linear_cross_attention_ws_floats(B, T, n_text) * sizeof(float);
}
int ladiff_linear_cross_attention(const float* const* w, int layer, const float* x, const float* xf, const float* emb,
                                  const int32_t* counts, int B, int T, int n_text, float* out, void* ws, size_t ws_bytes,
                                  ladiff_stream_t stream) {
    DenoiserW W;
    LADIFF_CHECK_ARG(load_weights(W, w) && x && xf && emb && out && ws && B > 0);
    return linear_cross_attention(W, layer, x, xf, emb, counts, B, T, n_text, out, (float*)ws, ws_bytes / sizeof(float), S(stream));
}

// ------------------------------------------------------------------ guidance + scheduler
int ladiff_cfg_scheduler_step(const float* eps, float* latents, const float* coef, const int32_t* d_step,
                              const float* step_noise, float guidance_scale, int cfg, int B, int T,
                              ladiff_stream_t stream) {
    LADIFF_CHECK_ARG(eps && latents && coef && d_step && B > 0 && T > 0);
    return launch_cfg_step(eps, latents, coef, d_step, step_noise, guidance_scale, cfg, B, T, S(stream));
}
int ladiff_advance_step(int32_t* d_step, ladiff_stream_t stream) {
    LADIFF_CHECK_ARG(d_step);
    return launch_advance(d_step, S(stream));
}
int ladiff_init_latents(const float* noise, const int32_t* counts, float sigma, float* latents, int B, int T,
                        ladiff_stream_t stream) {
    LADIFF_CHECK_ARG(noise && latents && B > 0 && T > 0);
    return launch_init_latents(noise, counts, sigma, latents, B, T, S(stream));
}
int ladiff_finalize_latents(const float* latents, const int32_t* counts, float* z, int B, int T, ladiff_stream_t stream) {
    LADIFF_CHECK_ARG(latents && z && B > 0 && T > 0);
    return launch_finalize_latents(latents, counts, z, B, T, S(stream));
}

// ------------------------------------------------------------------ whole reverse loop
int ladiff_sampler_create(void** sampler) {
    LADIFF_CHECK_ARG(sampler);
    *sampler = new Sampler();
    return 0;
}
int ladiff_sampler_destroy(void* sampler) {
    Sampler* sp = reinterpret_cast<Sampler*>(sampler);
    if (sp == nullptr) return 0;
    (void)hipDeviceSynchronize();         // a replay of these graphs may still be queued
    drain_retired(sp);
    if (sp->exec) (void)hipGraphExecDestroy(sp->exec);
    if (sp->setup) (void)hipGraphExecDestroy(sp->setup);
    if (sp->ev0) (void)hipEventDestroy(sp->ev0);
    if (sp->ev1) (void)hipEventDestroy(sp->ev1);
    for (hipEvent_t e : sp->wev) (void)hipEventDestroy(e);
    if (sp->cap) (void)hipStreamDestroy(sp->cap);
    delete sp;
    return 0;
}

}  // extern "C"

// Block plan of the pipeline loop for one call (host only).  loop_mode: 1 = pick by the cost model, 2 / 3 = force 16- / 32-row blocks.
static void choose_plan(int B, int T, const int32_t* h_counts, bool masked, int loop_mode, bool f16x3, std::vector<unsigned char>& plan,
                        int* plan_mr, int* plan_nb, bool cfg = true) {
    int mr16 = 1, nb16 = 0, mr32 = 2, nb32 = 0;
    std::vector<unsigned char> p16, p32;
    if (!cfg) {             // no guidance: one-branch 16-row blocks only (the caller made sure the counts are on the host, or absent)
        sys_pack_blocks(B, T, 1, h_counts, masked, false, plan, plan_mr, plan_nb);
        return;
    }
    sys_pack_blocks(B, T, 2, h_counts, masked, true, p32, &mr32, &nb32);
    int want = loop_mode == 2 ? 1 : (loop_mode == 3 ? 2 : 0);
    if (want != 2) sys_pack_blocks(B, T, 1, h_counts, masked, true, p16, &mr16, &nb16);
    if (want == 0) {
        // measured (scripts/try_pipeline.py uniform, 1 ... 128 prompts, final build of round 2): the busiest stage's time per block
        // and one block's unloaded trip through the 59 stages, in us, for 16- / 32-row blocks
        const double c16 = f16x3 ? 2.45 : 5.05, c32 = f16x3 ? 5.3 : 12.1, lat16 = f16x3 ? 172.0 : 310.0, lat32 = f16x3 ? 282.0 : 525.0;
        const double e16 = mr16 == 1 ? std::max(lat16, nb16 * c16) : 1e30, e32 = std::max(lat32, nb32 * c32);
        want = e16 < e32 ? 1 : 2;
    }
    if (want == 1 && mr16 == 1) { plan.swap(p16); *plan_mr = 1; *plan_nb = nb16; }
    else { plan.swap(p32); *plan_mr = 2; *plan_nb = nb32; }
}

extern "C" {

int ladiff_reverse_plan(int B, int T, const int32_t* h_counts, int masked, int loop_mode, int f16x3, int cfg, int* rows_per_block, int* n_blocks) {
    LADIFF_CHECK_ARG(B >= 1 && T >= 1 && T <= LADIFF_MAX_LATENTS && loop_mode >= 1 && loop_mode <= 3 && rows_per_block && n_blocks);
    if (!cfg && masked && h_counts == nullptr) return LADIFF_ERR_UNSUPPORTED;      // such a call runs launch-per-stage (no block plan)
    std::vector<unsigned char> plan;
    int mr = 2, nb = 0;
    choose_plan(B, T, h_counts, masked != 0, loop_mode, f16x3 != 0, plan, &mr, &nb, cfg != 0);
    *rows_per_block = 16 * mr; *n_blocks = nb;
    return 0;
}

int ladiff_debug_set_stage_waves(int waves_per_simd) {
    LADIFF_CHECK_ARG(waves_per_simd == 1 || waves_per_simd == 2);
    g_waves16 = waves_per_simd;
    return 0;
}

int ladiff_debug_set_handoff(int tagged) {
    LADIFF_CHECK_ARG(tagged == 0 || tagged == 1);
    g_handoff = tagged;
    return 0;
}

int ladiff_debug_set_mlp_variant(int v) {
#ifdef LADIFF_STAMPS
    LADIFF_CHECK_ARG((v >= 0 && v <= 3) || (v >= 11 && v <= 17) || (v >= 21 && v <= 26));    // the diagnostic twin also carries the timing builds
#else
    LADIFF_CHECK_ARG(v >= 0 && v <= 3);          // workgroup forms of the fused feed-forward kernel; every one gives the same result
#endif
    g_mlp_variant = v;
    return 0;
}

int ladiff_debug_set_decoder_fusion(int on) {
    LADIFF_CHECK_ARG(on >= 0 && on <= 126 && (on & 3) != 3 && (on & 48) != 48);
    g_dec_out_cross = (on & 64) ? 0 : 1;
    g_dec_fused_mlp = on & 3;
    g_dec_small_rows_path = (on & 4) ? 0 : 1;
    g_dec_final_split = (on & 8) ? 0 : 1;
    g_dec_fused_attn = (on & 16) ? 0 : (on & 32) ? 2 : 1;
    return 0;
}

int ladiff_debug_set_stage_plan(int v) {
    LADIFF_CHECK_ARG(v >= 0 && v <= 1);
    g_stage_plan = v;
    return 0;
}

int ladiff_debug_set_poll_pause(int mask, int len) {
    LADIFF_CHECK_ARG(mask >= 0 && mask <= 255 && len >= 0 && len <= 64);
    g_poll_pause = mask | (len << 8);
    return 0;
}

int ladiff_debug_set_pacing(int eighths, int mask) {
    if (eighths == -1) { g_pace = -1; return 0; }       // back to the built-in choice by launch size
    LADIFF_CHECK_ARG(eighths >= 0 && eighths <= 8 && mask >= 0 && mask <= 255);
    g_pace = eighths | (mask << 8);
    return 0;
}

int ladiff_debug_set_stage_delay(int mask, int len) {
    LADIFF_CHECK_ARG(mask >= -1 && mask <= 255 && len >= 0 && len <= 64);
    g_stage_delay = mask < 0 ? -1 : (mask | (len << 8));
    return 0;
}

int ladiff_debug_set_loop_thresholds(int look_ahead_from, int small_upto) {
    LADIFF_CHECK_ARG(look_ahead_from >= -1 && small_upto >= -1);
    g_look_ahead_from = look_ahead_from;
    g_small_upto = small_upto;
    return 0;
}

int ladiff_debug_set_graph_epoch_rule(int on) {
    LADIFF_CHECK_ARG(on == 0 || on == 1);
    g_graph_epoch_rule = on;
    return 0;
}

int ladiff_debug_graph_instantiations(void) { return g_graph_instantiations.load(); }

int ladiff_debug_set_xcd_local(int on) {
    LADIFF_CHECK_ARG(on >= 0 && on <= 2);
    g_xcd_local = on;
    return 0;
}

int ladiff_sampler_set_loop(void* sampler, int mode) {
    Sampler* sp = reinterpret_cast<Sampler*>(sampler);
    LADIFF_CHECK_ARG(sp != nullptr && mode >= 0 && mode <= 3);
    sp->loop = mode != 0;
    sp->loop_mode = mode;
    return 0;
}

int ladiff_sampler_loop_ms(void* sampler, float* ms) {
    Sampler* sp = reinterpret_cast<Sampler*>(sampler);
    LADIFF_CHECK_ARG(sp != nullptr && ms != nullptr && sp->ev1 != nullptr);
    LADIFF_HIP(hipEventSynchronize(sp->ev1));
    LADIFF_HIP(hipEventElapsedTime(ms, sp->ev0, sp->ev1));
    return 0;
}

int ladiff_sampler_set_window_timing(void* sampler, int on) {
    Sampler* sp = reinterpret_cast<Sampler*>(sampler);
    LADIFF_CHECK_ARG(sp != nullptr);
    sp->time_windows = on != 0;
    return 0;
}

int ladiff_sampler_window_ms(void* sampler, float* loop_ms_sum, int* n_windows) {
    Sampler* sp = reinterpret_cast<Sampler*>(sampler);
    LADIFF_CHECK_ARG(sp != nullptr && loop_ms_sum != nullptr && n_windows != nullptr);
    float sum = 0.f;
    for (int i = 0; i < sp->n_windows; ++i) {
        float ms = 0.f;
        LADIFF_HIP(hipEventSynchronize(sp->wev[2 * i + 1]));
        LADIFF_HIP(hipEventElapsedTime(&ms, sp->wev[2 * i], sp->wev[2 * i + 1]));
        sum += ms;
    }
    *loop_ms_sum = sum; *n_windows = sp->n_windows;
    return 0;
}

int ladiff_sampler_last_loop(void* sampler, int* pipeline, int* rows_per_block, int* n_blocks) {
    Sampler* sp = reinterpret_cast<Sampler*>(sampler);
    LADIFF_CHECK_ARG(sp != nullptr && pipeline != nullptr);
    *pipeline = sp->last_pipeline;
    if (rows_per_block) *rows_per_block = sp->last_pipeline ? 16 * sp->plan_mr : 0;
    if (n_blocks) *n_blocks = sp->last_pipeline ? sp->plan_nb : 0;
    return 0;
}

int ladiff_sampler_set_noise_generator(void* sampler, uint64_t seed, uint32_t first_prompt, int enable) {
    Sampler* sp = reinterpret_cast<Sampler*>(sampler);
    LADIFF_CHECK_ARG(sp != nullptr);
    sp->gen = NoiseGen{(unsigned)(seed & 0xffffffffull), (unsigned)(seed >> 32), first_prompt, enable ? 1 : 0};
    return 0;
}

int ladiff_noise_fill(uint64_t seed, uint32_t first_prompt, int first_step, int n_steps, int B, int T, float* out, ladiff_stream_t stream) {
    LADIFF_CHECK_ARG(out != nullptr && first_step >= 0 && n_steps >= 0 && B >= 0);
    if (T < 1 || T > LADIFF_MAX_LATENTS) return LADIFF_ERR_SHAPE;
    return launch_noise_fill(NoiseGen{(unsigned)(seed & 0xffffffffull), (unsigned)(seed >> 32), first_prompt, 1}, first_step, n_steps, B, T, out,
                             S(stream));
}

int ladiff_sampler_set_fault(void* sampler, int workgroup, int timeout_ms) {
    Sampler* sp = reinterpret_cast<Sampler*>(sampler);
    LADIFF_CHECK_ARG(sp != nullptr && workgroup >= -1 && workgroup < 256 && timeout_ms >= 0);
    sp->fault_wg = workgroup;
    sp->timeout_ticks = (unsigned long long)timeout_ms * 100000ull;    // s_memrealtime: 100 MHz
    return 0;
}

size_t ladiff_reverse_status_offset_bytes(int B, int T, int n_steps, int n_text) {
    if (B < 1 || T < 1 || T > LADIFF_MAX_LATENTS || n_steps < 1 || n_text < 1) return 0;
    const ReverseWs r = carve_reverse(nullptr, B, T, n_steps, n_text);
    return (r.sys_off + sys_status_offset_floats(B, T)) * sizeof(float);
}

int ladiff_reverse_status(void* ws, int B, int T, int n_steps, int n_text, int* code, int* info) {
    LADIFF_CHECK_ARG(ws && code && B > 0 && n_steps > 0);
    if (T < 1 || T > LADIFF_MAX_LATENTS) return LADIFF_ERR_SHAPE;
    const ReverseWs r = carve_reverse(ws, B, T, n_steps, n_text);
    unsigned st[2] = {0u, 0u};
    LADIFF_HIP(hipMemcpy(st, r.sys + sys_status_offset_floats(B, T), sizeof(st), hipMemcpyDeviceToHost));   // synchronises
    *code = (int)st[0];
    if (info) *info = (int)st[1];
    return 0;
}

size_t ladiff_reverse_workspace_bytes(int B, int T, int n_steps, int n_text) { return carve_reverse(nullptr, B, T, n_steps, n_text).total_bytes; }

int ladiff_mlp_ln_fused(const float* xs, const float* x, const float* w1s, const float* b1, const float* w2s, const float* b2,
                        const float* ln_gamma, const float* ln_beta, const float* ln2_gamma, const float* ln2_beta, float* y, float* ys,
                        int M, ladiff_stream_t stream) {
    LADIFF_CHECK_ARG(xs && x && w1s && b1 && w2s && b2 && ln_gamma && ln_beta && (y || ys) && M >= 0);
    if ((ln2_gamma == nullptr) != (ln2_beta == nullptr)) return LADIFF_ERR_ARG;
    return launch_dec_mlp(xs, x, w1s, b1, w2s, b2, ln_gamma, ln_beta, ln2_gamma, ln2_beta, y, ys, M, S(stream));
}

int ladiff_split_rows(const float* x, float* y, int R, int K, ladiff_stream_t stream) {
    LADIFF_CHECK_ARG(x && y && R >= 0 && K > 0);
    if (K % 64) return LADIFF_ERR_SHAPE;
    if (R == 0) return 0;
    return launch_split_rows(x, y, R, K, S(stream));
}

int ladiff_diffusion_reverse(void* sampler, const float* const* w, const float* const* w_split, uint64_t weights_generation,
                             const float* text_emb, const float* init_noise, const int32_t* counts, const int32_t* final_counts,
                             const int32_t* h_counts, const float* sinusoid, const float* coef, const float* step_noise, float guidance_scale,
                             float init_noise_sigma, int cfg, int B, int T, int n_text, int n_steps, float* z, void* ws, size_t ws_bytes,
                             int reuse_time_tables, ladiff_stream_t stream) {
    DenoiserW W, WS;
    LADIFF_CHECK_ARG(load_weights(W, w) && text_emb && init_noise && sinusoid && coef && z && ws && B > 0 && n_steps > 0);
    if (w_split != nullptr) LADIFF_CHECK_ARG(load_weights(WS, w_split));
    const DenoiserW* WSp = w_split ? &WS : nullptr;
    if (T < 1 || T > LADIFF_MAX_LATENTS || n_text < 1) return LADIFF_ERR_SHAPE;
    ReverseWs r = carve_reverse(ws, B, T, n_steps, n_text);
    if (ws_bytes < r.total_bytes) return LADIFF_ERR_WORKSPACE;
    hipStream_t s = S(stream);
    const int dup = cfg ? 2 : 1;          // guidance: the network sees cat([latents]*2) with text [uncond | cond]  ladiff.py:472-474
    const int B2 = dup * B;

    Sampler* sp = reinterpret_cast<Sampler*>(sampler);
    // the sampler's generator stands in for a step-noise tensor the caller did not pass (schedules without noise never look at either)
    // A generator that is off is all zeros: its seed must not be part of any graph key (the Python loop owner draws a fresh seed per call,
    // also for deterministic schedules - a key that changed with it re-captured ~150-node graphs on every launch-per-stage call).
    const NoiseGen gen = (sp != nullptr && step_noise == nullptr && sp->gen.on) ? sp->gen : NoiseGen{0u, 0u, 0u, 0};
    float *xio = nullptr, *xios = nullptr;
    den_loop_io(r.fwd, B2 * T, &xio, &xios);
    if (WSp == nullptr) xios = nullptr;

    // hoisted, once per call: time tables for every step, text cache, initial latents, step counter, first network input.
    // The time tables depend on (weights, schedule) only: a caller that re-runs with both unchanged in the same workspace
    // may keep them (saves ~30 small GEMM launches per call).  The rest (~50 small launches) depends on this call's text
    // and noise; with a sampler it is replayed as a graph so that the host does not pace the GPU through it.
    // abort / diagnostic words of the pipeline loop: cleared once per call (they are sticky over the call's windows; every
    // other loop form leaves them at "completed")
    LADIFF_TRY(sys_reset_status(r.sys, s));
    if (!reuse_time_tables) LADIFF_TRY(denoiser_time_tables(W, sinusoid, n_steps, r.tables, r.fwd, r.fwd_floats, s));
    auto prologue = [&](hipStream_t st) -> int {
        if (n_text > 1) LADIFF_TRY(denoiser_text_cache(W, text_emb, B2, r.tables, n_steps, r.cache, r.fwd, r.fwd_floats, st, n_text));
        else LADIFF_TRY(denoiser_text_static(W, text_emb, B2, r.cache, r.fwd, r.fwd_floats, st));      // the c table: per window, below
        LADIFF_TRY(launch_init_latents(init_noise, counts, init_noise_sigma, r.latents, B, T, st));
        // [0] step index, [1] tail-kernel ticket, [2] window base.  A KERNEL, not hipMemsetAsync: this runs inside the captured prologue
        // graph, and a memset NODE is what an older exec replayed wrongly (see g_graph_epoch)
        LADIFF_TRY(launch_zero_fill(reinterpret_cast<float*>(r.d_step), 4, st));
        // One step = the nine denoiser layers + ONE tail launch (final LayerNorm of the guidance branches, guidance,
        // scheduler step, next step's network input, step counter).  The network input / last-layer output buffer of the
        // forward workspace is primed here.
        return launch_add_pe(r.latents, W.query_pe, B, 0, B2, T, xio, xios, st);
    };
    auto one_step = [&](hipStream_t st) -> int {
        LADIFF_TRY(denoiser_forward(W, WSp, r.tables, r.d_step, r.cache, r.window, r.latents, B, dup, T, counts, r.eps, r.fwd,
                                    r.fwd_floats, st, 0, B2, 1, n_text, r.d_step + 2));
        return launch_step_tail(xio, xios, W.norm.g, W.norm.b, r.latents, coef, r.d_step, step_noise, W.query_pe,
                                guidance_scale, cfg, B, T, st, gen);
    };
    // without guidance the pipeline runs one-branch 16-row blocks, which need the latent counts on the host (or no masking at all)
    const bool pipeline = sp != nullptr && sp->loop == 1 && n_text == 1 && sys_supported(B, T, cfg, WSp != nullptr) &&
                          (cfg || counts == nullptr || h_counts != nullptr);
    // Block geometry of the pipeline for THIS call's lengths.  16-row blocks carry only the valid latent rows of each prompt
    // (length-aware packing; needs the counts on the host), 32-row blocks the padded T rows.  A step costs the larger of (blocks x
    // the busiest stage's time per block) and one block's trip through the 59 stages: choose_plan() picks the cheaper plan.
    std::vector<unsigned char> plan;
    int plan_mr = 2, plan_nb = 0;
    if (pipeline) choose_plan(B, T, h_counts, counts != nullptr, sp->loop_mode, WSp != nullptr, plan, &plan_mr, &plan_nb, cfg != 0);
    // c-table rows of the window that starts at step `lo` (plain launches, outside the graphs: `lo` changes per window)
    auto open_window = [&](int lo) -> int {
        if (n_text > 1) return 0;
        LADIFF_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(r.d_step + 2), lo, 1, s));
        return denoiser_ctab(W, r.tables + (size_t)lo * DEN_STEP_STRIDE, r.window, r.cache, B2, r.cws, r.cws_floats, s, WSp);
    };
    if (sp == nullptr) {
        LADIFF_TRY(prologue(s));
        for (int i = 0; i < n_steps; ++i) {
            if (i % r.window == 0) LADIFF_TRY(open_window(i));
            LADIFF_TRY(one_step(s));
        }
    } else {
        const void* kp[9] = {ws, counts, final_counts, coef, step_noise, stream, text_emb, init_noise, z};
        const int ki[4] = {B, T, n_steps + 65536 * plan_nb, cfg + 2 * (pipeline ? plan_mr : 0) + 16 * n_text + 4096 * (WSp ? 1 : 0)};
        const float kf[2] = {guidance_scale, init_noise_sigma};
        const unsigned kn[4] = {gen.seed_lo, gen.seed_hi, gen.prompt0, (unsigned)gen.on};      // baked into the step graphs' tail nodes
        uint64_t h = hash_ptrs(w, DEN_NPARAMS, 1469598103934665603ull);
        if (w_split) h = hash_ptrs(w_split, DEN_NPARAMS, h ^ 0x9e3779b97f4a7c15ull);
        const bool same = sp->setup && std::memcmp(kp, sp->key_ptrs, sizeof(kp)) == 0 &&
                          std::memcmp(ki, sp->key_ints, sizeof(ki)) == 0 && std::memcmp(kf, sp->key_f, sizeof(kf)) == 0 &&
                          (pipeline || std::memcmp(kn, sp->key_gen_noise, sizeof(kn)) == 0) &&
                          h == sp->key_hash && weights_generation == sp->key_gen;
        // ladiff_debug_set_graph_epoch_rule(0) (test aid): trust an older exec, as tests/test_gpu_stress.py does to show that the graphs -
        // kernel nodes only since round 4 - replay correctly however old they are
        const bool newest = sp->epoch == g_graph_epoch.load() || g_graph_epoch_rule.load() == 0;
        if (!same || !newest) {
            if (!same) {
                // replays of the old graphs may still be queued (the host never paces the GPU): drain before destroying them
                if (sp->exec || sp->setup || !sp->retired.empty()) LADIFF_HIP(hipStreamSynchronize(s));
                drain_retired(sp);
                if (sp->exec) { (void)hipGraphExecDestroy(sp->exec); sp->exec = nullptr; }
                if (sp->setup) { (void)hipGraphExecDestroy(sp->setup); sp->setup = nullptr; }
            } else {
                // same key, but another sampler has instantiated since: capture again.  No drain (a chunked batch alternates two
                // samplers launch after launch): the old graphs are set aside and destroyed at the next drain
                if (sp->retired.size() >= 8) { LADIFF_HIP(hipStreamSynchronize(s)); drain_retired(sp); }
                if (sp->exec) { sp->retired.push_back(sp->exec); sp->exec = nullptr; }
                if (sp->setup) { sp->retired.push_back(sp->setup); sp->setup = nullptr; }
            }
            hipGraph_t graph = nullptr;
            LADIFF_TRY(capture_stream(&sp->cap, &sp->cap_dev));
            const hipStream_t cs = sp->cap;      // captured here, replayed on `s` (see Sampler::cap)
            {   // prologue graph
                LADIFF_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
                const int rc0 = prologue(cs);
                const hipError_t e0 = hipStreamEndCapture(cs, &graph);
                if (rc0 != 0) { if (graph) (void)hipGraphDestroy(graph); return rc0; }
                LADIFF_HIP(e0);
                const hipError_t i0 = hipGraphInstantiate(&sp->setup, graph, nullptr, nullptr, 0);
                ++g_graph_instantiations;
                (void)hipGraphDestroy(graph);
                graph = nullptr;
                LADIFF_HIP(i0);
            }
            if (pipeline && same) {
                // only the prologue graph was renewed: the stage table in the workspace is this key's
            } else if (pipeline) {
                // stage table of the persistent pipeline (pointers of this call's weights and workspace): built and uploaded
                // once per key; the host copy stays alive in the sampler until the next rebuild
                LADIFF_TRY(sys_build_stages(W, WSp ? WS : W, r.sys, plan_mr, plan_nb, sp->stages));
                sp->blocks.clear();            // the descriptor area moved with the layout: upload again
                LADIFF_HIP(hipMemcpyAsync(r.sys, sp->stages.data(), sp->stages.size(), hipMemcpyHostToDevice, s));
                LADIFF_HIP(hipStreamSynchronize(s));
            } else {
                LADIFF_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
                // several steps per graph launch (the step index lives in device memory): fewer ~9 us replay gaps
                int unroll = 1;
                for (int u = 2; u <= 10; ++u) if (r.window % u == 0) unroll = u;
                sp->unroll = unroll;
                int rc = 0;
                for (int u = 0; u < unroll && rc == 0; ++u) rc = one_step(cs);
                const hipError_t ec = hipStreamEndCapture(cs, &graph);
                if (rc != 0) { if (graph) (void)hipGraphDestroy(graph); return rc; }
                LADIFF_HIP(ec);
                const hipError_t ei = hipGraphInstantiate(&sp->exec, graph, nullptr, nullptr, 0);
                ++g_graph_instantiations;
                (void)hipGraphDestroy(graph);
                LADIFF_HIP(ei);
            }
            std::memcpy(sp->key_ptrs, kp, sizeof(kp));
            std::memcpy(sp->key_ints, ki, sizeof(ki));
            std::memcpy(sp->key_f, kf, sizeof(kf));
            std::memcpy(sp->key_gen_noise, kn, sizeof(kn));
            sp->key_hash = h;
            sp->key_gen = weights_generation;
            sp->epoch = ++g_graph_epoch;
        }
        LADIFF_HIP(hipGraphLaunch(sp->setup, s));
        if (sp->ev0 == nullptr) { LADIFF_HIP(hipEventCreate(&sp->ev0)); LADIFF_HIP(hipEventCreate(&sp->ev1)); }
        if (pipeline && (plan_mr != sp->plan_mr || plan_nb != sp->plan_nb || plan != sp->blocks)) {
            // this call's block descriptors (a few KB; the runtime stages a pageable source before it returns)
            if (!sp->blocks.empty()) LADIFF_HIP(hipStreamSynchronize(s));      // a copy from the old buffer may still be in flight
            sp->blocks = plan; sp->plan_mr = plan_mr; sp->plan_nb = plan_nb;
            LADIFF_HIP(hipMemcpyAsync(r.sys + sys_blocks_offset_floats(plan_mr, plan_nb), sp->blocks.data(), sp->blocks.size(),
                                      hipMemcpyHostToDevice, s));
        }
        sp->last_pipeline = pipeline ? 1 : 0;
        sp->n_windows = 0;
        for (int lo = 0; lo < n_steps; lo += r.window) {
            LADIFF_TRY(open_window(lo));
            if (lo == 0) LADIFF_HIP(hipEventRecord(sp->ev0, s));       // the loop itself: from the first step's first launch
            const int wi = lo / r.window;
            if (sp->time_windows) {
                while ((int)sp->wev.size() < 2 * (wi + 1)) { hipEvent_t e; LADIFF_HIP(hipEventCreate(&e)); sp->wev.push_back(e); }
                LADIFF_HIP(hipEventRecord(sp->wev[2 * wi], s));
            }
            if (pipeline) {
                LADIFF_TRY(launch_systolic_loop(W, r.sys, r.tables, den_cache_tkv(r.cache, B2, 1), den_cache_ctab(r.cache, B2, 1), r.window,
                                                coef, step_noise, r.latents, counts, guidance_scale, B, T, lo, r.window, WSp ? 0 : 1, plan_mr, plan_nb, s, cfg,
                                                sp->fault_wg, sp->timeout_ticks, gen));
            } else {
                for (int i = 0; i < r.window / sp->unroll; ++i) LADIFF_HIP(hipGraphLaunch(sp->exec, s));
            }
            if (sp->time_windows) { LADIFF_HIP(hipEventRecord(sp->wev[2 * wi + 1], s)); sp->n_windows = wi + 1; }
        }
        LADIFF_HIP(hipEventRecord(sp->ev1, s));
    }
    // final zeroing of the rows past each motion's latent count: applied even when the denoiser ran unmasked
    // (TEST_EFFICIENCY), as ladiff.py:559-566 does.  An aborted pipeline launch leaves partial latents: z is then NaN.
    return launch_finalize_latents(r.latents, final_counts, z, B, T, s,
                                   reinterpret_cast<const unsigned*>(r.sys + sys_status_offset_floats(B, T)));
}

// ------------------------------------------------------------------ LA-VAE encoder
int ladiff_encoder_num_params(void) { return ENC_NPARAMS; }
const char* ladiff_encoder_param_name(int i) {
    const auto& n = encoder_param_names();
    return (i >= 0 && i < (int)n.size()) ? n[i].c_str() : nullptr;
}
size_t ladiff_encoder_workspace_bytes(int B, int F, int T, int C) { return enc_ws_floats(B, F, T, C) * sizeof(float); }

int ladiff_vae_encode(const float* const* w, const float* const* w_split, const float* features, const int32_t* lengths,
                      const int32_t* counts, const float* eps, int B, int F, int T, int C, float* mu, float* std,
                      float* latent, void* ws, size_t ws_bytes, ladiff_stream_t stream) {
    EncoderW W, WS;
    LADIFF_CHECK_ARG(load_weights(W, w) && features && lengths && counts && eps && mu && std && latent && ws && B >= 0);
    if (w_split != nullptr) LADIFF_CHECK_ARG(load_weights(WS, w_split));
    return vae_encode(W, w_split ? &WS : nullptr, features, lengths, counts, eps, B, F, T, C, mu, std, latent, (float*)ws,
                      ws_bytes / sizeof(float), S(stream));
}

// ------------------------------------------------------------------ CLIP text encoder
int ladiff_clip_num_params(void) { return CLIP_NPARAMS; }
const char* ladiff_clip_param_name(int i) {
    const auto& n = clip_param_names();
    return (i >= 0 && i < (int)n.size()) ? n[i].c_str() : nullptr;
}
size_t ladiff_clip_workspace_bytes(int B, int L) { return clip_ws_floats(B, L) * sizeof(float); }

static bool load_clip(ClipW& dst, const float* const* ptrs, int n_layers) {     // only the first n_layers must be present
    if (ptrs == nullptr || n_layers < 1 || n_layers > CLIP_MAX_LAYERS) return false;
    const int n = CLIP_HEAD_NPARAMS + CLIP_LAYER_NPARAMS * n_layers;
    for (int i = 0; i < n; ++i)
        if (ptrs[i] == nullptr) return false;
    std::memset(&dst, 0, sizeof(dst));
    std::memcpy(&dst, ptrs, n * sizeof(const float*));
    return true;
}

int ladiff_clip_text_encode(const float* const* w, const float* const* w_split, int n_layers, int vocab, const int64_t* ids,
                            int B, int seq, int L, float* out, void* ws, size_t ws_bytes, ladiff_stream_t stream) {
    ClipW W, WS;
    LADIFF_CHECK_ARG(load_clip(W, w, n_layers) && ids && out && ws && B >= 0);
    if (w_split != nullptr) LADIFF_CHECK_ARG(load_clip(WS, w_split, n_layers));
    return clip_text_encode(W, w_split ? &WS : nullptr, n_layers, vocab, ids, B, seq, L, out, (float*)ws,
                            ws_bytes / sizeof(float), S(stream));
}

size_t ladiff_clip_workspace_bytes_ragged(int B, int total_rows) { return clip_ws_floats_rows(B, total_rows) * sizeof(float); }

int ladiff_clip_text_encode_ragged(const float* const* w, const float* const* w_split, int n_layers, int vocab, const int64_t* ids,
                                   int B, int seq, int L, const int32_t* seq_len, const int32_t* row_off, const int32_t* row_seq,
                                   int total_rows, float* out, void* ws, size_t ws_bytes, ladiff_stream_t stream) {
    ClipW W, WS;
    LADIFF_CHECK_ARG(load_clip(W, w, n_layers) && ids && out && ws && B >= 0 && seq_len && row_off && row_seq && total_rows >= 0);
    if (w_split != nullptr) LADIFF_CHECK_ARG(load_clip(WS, w_split, n_layers));
    return clip_text_encode(W, w_split ? &WS : nullptr, n_layers, vocab, ids, B, seq, L, out, (float*)ws,
                            ws_bytes / sizeof(float), S(stream), seq_len, row_off, row_seq, total_rows);
}

// ------------------------------------------------------------------ T2M evaluator encoders
static bool all_set(const float* const* w, size_t n) {
    if (w == nullptr) return false;
    for (size_t i = 0; i < n; ++i)
        if (w[i] == nullptr) return false;
    return true;
}
static const char* name_at(const std::vector<std::string>& n, int i) { return (i >= 0 && i < (int)n.size()) ? n[i].c_str() : nullptr; }
int ladiff_t2m_movement_num_params(void) { return (int)t2m_move_param_names().size(); }
const char* ladiff_t2m_movement_param_name(int i) { return name_at(t2m_move_param_names(), i); }
int ladiff_t2m_motion_num_params(void) { return (int)t2m_motion_param_names().size(); }
const char* ladiff_t2m_motion_param_name(int i) { return name_at(t2m_motion_param_names(), i); }
int ladiff_t2m_text_num_params(void) { return (int)t2m_text_param_names().size(); }
const char* ladiff_t2m_text_param_name(int i) { return name_at(t2m_text_param_names(), i); }
size_t ladiff_t2m_movement_workspace_bytes(int B, int F, int Cin) { return t2m_move_ws_floats(B, F, Cin) * sizeof(float); }
size_t ladiff_t2m_motion_workspace_bytes(int B, int T) { return t2m_motion_ws_floats(B, T) * sizeof(float); }
size_t ladiff_t2m_text_workspace_bytes(int B, int L) { return t2m_text_ws_floats(B, L) * sizeof(float); }

int ladiff_t2m_movement_encode(const float* const* w, const float* feats, int ld, int B, int F, int Cin, float* out, void* ws,
                               size_t ws_bytes, ladiff_stream_t stream) {
    LADIFF_CHECK_ARG(all_set(w, t2m_move_param_names().size()) && feats && out && ws && B >= 0);
    return t2m_movement_encode(w, feats, ld, B, F, Cin, out, (float*)ws, ws_bytes / sizeof(float), S(stream));
}
int ladiff_t2m_motion_encode(const float* const* w, const float* movements, const int32_t* m_lens, int B, int T, float* out,
                             void* ws, size_t ws_bytes, ladiff_stream_t stream) {
    LADIFF_CHECK_ARG(all_set(w, t2m_motion_param_names().size()) && movements && m_lens && out && ws && B >= 0);
    return t2m_motion_encode(w, movements, m_lens, B, T, out, (float*)ws, ws_bytes / sizeof(float), S(stream));
}
int ladiff_t2m_text_encode(const float* const* w, const float* word_embs, const float* pos_onehot, const int32_t* cap_lens,
                           int B, int L, float* out, void* ws, size_t ws_bytes, ladiff_stream_t stream) {
    LADIFF_CHECK_ARG(all_set(w, t2m_text_param_names().size()) && word_embs && pos_onehot && cap_lens && out && ws && B >= 0);
    return t2m_text_encode(w, word_embs, pos_onehot, cap_lens, B, L, out, (float*)ws, ws_bytes / sizeof(float), S(stream));
}

// ------------------------------------------------------------------ feats2joints (the step after the path)
int ladiff_feats2joints(const float* feats, const float* mean, const float* std, int B, int F, int C, int njoints,
                        float* joints, ladiff_stream_t stream) {
    LADIFF_CHECK_ARG(feats && mean && std && joints && B >= 0);
    return launch_feats2joints(feats, mean, std, B, F, C, njoints, joints, S(stream));
}

// ------------------------------------------------------------------ LA-VAE decoder
size_t ladiff_decoder_workspace_bytes(int B, int F, int T, int C) {
    (void)C;
    return dec_ws_floats(B, (size_t)B * F, T) * sizeof(float);
}

int ladiff_vae_decode(const float* const* w, const float* const* w_split, const float* z, const int32_t* lengths,
                      const int32_t* counts, int B, int F, int T, int C, float* feats, void* ws, size_t ws_bytes,
                      ladiff_stream_t stream) {
    DecoderW W, WS;
    LADIFF_CHECK_ARG(load_weights(W, w) && z && lengths && feats && ws && B >= 0);
    if (w_split != nullptr) LADIFF_CHECK_ARG(load_weights(WS, w_split));
    return vae_decode(W, w_split ? &WS : nullptr, z, lengths, counts, nullptr, 0, B, F, T, C, feats, (float*)ws, ws_bytes / sizeof(float),
                      S(stream));
}

// ---- the decode as a replayed hipGraph (launch-bound sizes: config c1's 8 x 60 frames is ~110 launches of a few microseconds)
namespace {
struct DecodeGraph {
    hipGraphExec_t exec = nullptr;
    hipStream_t cap = nullptr;            // captured on a stream of its own, replayed on the caller's (Sampler::cap)
    int cap_dev = -1;
    const void* key_ptrs[7] = {nullptr};
    uint64_t epoch = 0;                   // g_graph_epoch at instantiation: replayed only while it is the newest graph of the process
    int key_ints[6] = {0};
    uint64_t key_hash = 0, key_gen = 0;
};
}  // namespace

int ladiff_decoder_graph_create(void** graph) {
    LADIFF_CHECK_ARG(graph);
    *graph = new DecodeGraph();
    return 0;
}
int ladiff_decoder_graph_destroy(void* graph) {
    DecodeGraph* g = reinterpret_cast<DecodeGraph*>(graph);
    if (g == nullptr) return 0;
    (void)hipDeviceSynchronize();         // a replay may still be queued
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->cap) (void)hipStreamDestroy(g->cap);
    delete g;
    return 0;
}

int ladiff_vae_decode_graphed(void* graph, const float* const* w, const float* const* w_split, uint64_t weights_generation, const float* z,
                              const int32_t* lengths, const int32_t* counts, const int32_t* row_off, int total_rows, int B, int F, int T,
                              int C, float* feats, void* ws, size_t ws_bytes, ladiff_stream_t stream) {
    DecodeGraph* dg = reinterpret_cast<DecodeGraph*>(graph);
    DecoderW W, WS;
    LADIFF_CHECK_ARG(dg && load_weights(W, w) && z && lengths && feats && ws && B >= 0 && stream != nullptr);
    if (w_split != nullptr) LADIFF_CHECK_ARG(load_weights(WS, w_split));
    hipStream_t s = S(stream);
    const void* kp[7] = {z, lengths, counts, row_off, feats, ws, stream};
    const int ki[6] = {B, F, T, C, total_rows, w_split ? 1 : 0};
    uint64_t h = hash_ptrs(w, DEC_NPARAMS, 1469598103934665603ull);
    if (w_split) h = hash_ptrs(w_split, DEC_NPARAMS, h ^ 0x9e3779b97f4a7c15ull);
    // the measurement switches change the launch sequence: part of the key
    h ^= (uint64_t)(g_dec_fused_mlp + 4 * g_dec_small_rows_path + 8 * g_dec_final_split + 16 * g_mlp_variant + 4096 * g_dec_fused_attn) * 0x100000001b3ull;
    const bool same = dg->exec && std::memcmp(kp, dg->key_ptrs, sizeof(kp)) == 0 && std::memcmp(ki, dg->key_ints, sizeof(ki)) == 0 &&
                      h == dg->key_hash && weights_generation == dg->key_gen &&
                      (dg->epoch == g_graph_epoch.load() || g_graph_epoch_rule.load() == 0);     // the rule's switch is process-wide: samplers and decode graphs
    if (!same) {
        if (dg->exec) { LADIFF_HIP(hipStreamSynchronize(s)); (void)hipGraphExecDestroy(dg->exec); dg->exec = nullptr; }
        LADIFF_TRY(dec_mlp_prepare());            // kernel attributes are set outside the capture
        LADIFF_TRY(dec_qkv_attn_prepare());
        LADIFF_TRY(dec_cross_prepare());
        hipGraph_t gr = nullptr;
        LADIFF_TRY(capture_stream(&dg->cap, &dg->cap_dev));
        const hipStream_t cs = dg->cap;
        LADIFF_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        int rc = 0;
        if (row_off != nullptr) {      // inside the capture: a zero-fill KERNEL, never a memset node (g_graph_epoch)
            rc = launch_zero_fill(feats, (size_t)B * F * C, cs);
        }
        if (rc == 0) rc = vae_decode(W, w_split ? &WS : nullptr, z, lengths, counts, row_off, total_rows, B, F, T, C, feats, (float*)ws,
                                     ws_bytes / sizeof(float), cs);
        const hipError_t ec = hipStreamEndCapture(cs, &gr);
        if (rc != 0) { if (gr) (void)hipGraphDestroy(gr); return rc; }
        LADIFF_HIP(ec);
        const hipError_t ei = hipGraphInstantiate(&dg->exec, gr, nullptr, nullptr, 0);
        ++g_graph_instantiations;
        (void)hipGraphDestroy(gr);
        LADIFF_HIP(ei);
        std::memcpy(dg->key_ptrs, kp, sizeof(kp));
        std::memcpy(dg->key_ints, ki, sizeof(ki));
        dg->key_hash = h; dg->key_gen = weights_generation;
        dg->epoch = ++g_graph_epoch;
    }
    LADIFF_HIP(hipGraphLaunch(dg->exec, s));
    return 0;
}

int ladiff_vae_decode_ragged(const float* const* w, const float* const* w_split, const float* z, const int32_t* lengths,
                             const int32_t* counts, const int32_t* row_off, int total_rows, int B, int F, int T, int C, float* feats,
                             void* ws, size_t ws_bytes, ladiff_stream_t stream) {
    DecoderW W, WS;
    LADIFF_CHECK_ARG(load_weights(W, w) && z && lengths && row_off && feats && ws && B >= 0 && total_rows >= 0);
    if (w_split != nullptr) LADIFF_CHECK_ARG(load_weights(WS, w_split));
    // frames past each length: zero, whatever the buffer held (ladiff_vae.py:356-360)
    LADIFF_HIP(hipMemsetAsync(feats, 0, (size_t)B * F * C * sizeof(float), S(stream)));
    return vae_decode(W, w_split ? &WS : nullptr, z, lengths, counts, row_off, total_rows, B, F, T, C, feats, (float*)ws,
                      ws_bytes / sizeof(float), S(stream));
}

}  // extern "C"
