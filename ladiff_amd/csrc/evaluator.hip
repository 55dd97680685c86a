// T2M evaluator encoders: what `t2m_eval` runs on the decoded features to produce the embeddings the
// TM2T metrics are computed from (ladiff.py:1264-1271):
//   MovementConvEncoder   t2m_motionenc.py:6-25   Conv1d(C-4 -> 512, k4 s2 p1) + LeakyReLU(0.2), Conv1d(512 -> 512, k4 s2 p1)
//                                                 + LeakyReLU(0.2), Linear(512, 512): feats [B, F, C] -> [B, F / 4, 512]
//   MotionEncoderBiGRUCo  t2m_motionenc.py:28-64  Linear(512 -> 1024), bidirectional GRU(1024) over pack_padded_sequence
//                                                 (lengths F_i / 4), final hidden states of both directions ->
//                                                 Linear(2048, 1024) + LayerNorm + LeakyReLU(0.2) + Linear(1024, 512)
//   TextEncoderBiGRUCo    t2m_textenc.py:6-48     Linear(15 -> 300) on the POS one-hots + word embeddings, Linear(300 -> 512),
//                                                 bidirectional GRU(512), the same head -> 512
// Evaluation only, fp32, built from the general staged GEMM (gemm.hip) plus small kernels: the convolutions are GEMMs over an
// im2col view whose column order (c * 4 + tap) equals the Conv1d weight's memory order, the GRU input projections of all
// time steps are one GEMM per direction, the recurrence is one [B, H] x [3H, H] GEMM + one gate kernel per step for both
// directions; packed-sequence semantics = "a sample's state only advances while t < its length" (forward from 0, backward
// from T - 1 downwards).
#include "model.h"

#include <cstring>

namespace ladiff {

namespace {

int pad32(int c) { return (c + 31) / 32 * 32; }

// A[(b, t), c * 4 + k] = x[b, 2 t - 1 + k, c] (0 outside the sequence), columns >= 4 C zero;  x rows have `ldx` floats
__global__ __launch_bounds__(256) void im2col_k4s2_kernel(const float* __restrict__ x, int Tin, int ldx, int C, int Tout, int Kp,
                                                          size_t n, float* __restrict__ A) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int col = (int)(i % Kp);
    const size_t row = i / Kp;
    const int t = (int)(row % Tout);
    const size_t b = row / Tout;
    float v = 0.f;
    if (col < 4 * C) {
        const int c = col >> 2, f = 2 * t - 1 + (col & 3);
        if (f >= 0 && f < Tin) v = x[(b * Tin + f) * ldx + c];
    }
    A[i] = v;
}

// word_embs + pos_emb(pos_onehot) is a GEMM with K = 15 and a residual: handled by the GEMM epilogue (res = word_embs)

// GRU cell for both directions (gate order r | z | n, torch.nn.GRU):  h' = (1 - z) n + z h,  n = tanh(gi_n + r gh_n).
// gi [2][B][T][3H] holds x W_ih^T + b_ih for every step, gh [2][B][3H] = h W_hh^T + b_hh of this step; direction 0 reads
// step `t`, direction 1 step `T - 1 - t`; a sample advances only while that step index is < its length.
__global__ __launch_bounds__(256) void gru_gate_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                       const int32_t* __restrict__ lens, int B, int T, int Hs, int t,
                                                       float* __restrict__ h) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)2 * B * Hs;
    if (i >= n) return;
    const int j = (int)(i % Hs);
    const int b = (int)((i / Hs) % B);
    const int d = (int)(i / ((size_t)Hs * B));
    const int step = d == 0 ? t : T - 1 - t;
    if (step >= lens[b]) return;
    const float* gir = gi + (((size_t)d * B + b) * T + step) * 3 * Hs;
    const float* ghr = gh + ((size_t)d * B + b) * 3 * Hs;
    const float r = 1.f / (1.f + expf(-(gir[j] + ghr[j])));
    const float z = 1.f / (1.f + expf(-(gir[Hs + j] + ghr[Hs + j])));
    const float nn = tanhf(gir[2 * Hs + j] + r * ghr[2 * Hs + j]);
    h[i] = (1.f - z) * nn + z * h[i];
}

// h[d][b][:] = hidden[d][0][:]
__global__ __launch_bounds__(256) void gru_init_kernel(const float* __restrict__ hidden, int B, int Hs, float* __restrict__ h) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)2 * B * Hs) return;
    h[i] = hidden[(i / ((size_t)Hs * B)) * Hs + i % Hs];
}

// cat[b][d * H + j] = h[d][b][j]
__global__ __launch_bounds__(256) void gru_cat_kernel(const float* __restrict__ h, int B, int Hs, float* __restrict__ cat) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)2 * B * Hs) return;
    const int j = (int)(i % Hs), b = (int)((i / Hs) % B), d = (int)(i / ((size_t)Hs * B));
    cat[(size_t)b * 2 * Hs + d * Hs + j] = h[i];
}

// y = LeakyReLU_0.2(LayerNorm(x)) over rows of width W (multiple of 64), one wave per row
__global__ __launch_bounds__(256) void ln_lrelu_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                       const float* __restrict__ b, int W, int M, float* __restrict__ y) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (size_t)row * W;
    float s = 0.f;
    for (int c = lane; c < W; c += 64) s += xr[c];
    const float mean = wave_sum(s) / W;
    float q = 0.f;
    for (int c = lane; c < W; c += 64) { const float d = xr[c] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / W + LN_EPS);
    for (int c = lane; c < W; c += 64) {
        const float v = (xr[c] - mean) * rstd * g[c] + b[c];
        y[(size_t)row * W + c] = v > 0.f ? v : 0.2f * v;
    }
}

GemmArgs lin(const float* A, int lda, const float* W, int ldw, const float* bias, float* Y, int ldy, int M, int N, int K,
             int act = ACT_NONE) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.Y = Y; g.ldy = ldy; g.M = M; g.N = N; g.K = K; g.act = act;
    return g;
}

// bidirectional GRU over [B, T, Hs] inputs with per-sample lengths, then the co-embedding head; returns [B, out]
// ws: gi [2][B][T][3H], gh [2][B][3H], h [2][B][H], cat [B][2H], hid [B][H]
struct GruW { const float *w_ih, *w_hh, *b_ih, *b_hh, *w_ih_r, *w_hh_r, *b_ih_r, *b_hh_r; };
struct HeadW { LinearW l0; NormW norm; LinearW l3; };

size_t gru_head_floats(int B, int T, int Hs) {
    return (size_t)2 * B * T * 3 * Hs + (size_t)2 * B * 3 * Hs + (size_t)2 * B * Hs + (size_t)B * 2 * Hs + (size_t)2 * B * Hs + 64;
}

int gru_head(const float* emb, const int32_t* lens, const GruW& gw, const float* hidden, const HeadW& hw, int B, int T, int Hs,
             int out_dim, float* out, float* ws, hipStream_t s) {
    float* gi = ws;
    float* gh = gi + (size_t)2 * B * T * 3 * Hs;
    float* h = gh + (size_t)2 * B * 3 * Hs;
    float* cat = h + (size_t)2 * B * Hs;
    float* hid = cat + (size_t)B * 2 * Hs;
    const int H3 = 3 * Hs;
    // input projections of every step, both directions
    LADIFF_TRY(launch_gemm(lin(emb, Hs, gw.w_ih, Hs, gw.b_ih, gi, H3, B * T, H3, Hs), s));
    LADIFF_TRY(launch_gemm(lin(emb, Hs, gw.w_ih_r, Hs, gw.b_ih_r, gi + (size_t)B * T * H3, H3, B * T, H3, Hs), s));
    const size_t nh = (size_t)2 * B * Hs;
    hipLaunchKernelGGL(gru_init_kernel, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, s, hidden, B, Hs, h);
    LADIFF_LAUNCH_CHECK();
    for (int t = 0; t < T; ++t) {
        LADIFF_TRY(launch_gemm(lin(h, Hs, gw.w_hh, Hs, gw.b_hh, gh, H3, B, H3, Hs), s));
        LADIFF_TRY(launch_gemm(lin(h + (size_t)B * Hs, Hs, gw.w_hh_r, Hs, gw.b_hh_r, gh + (size_t)B * H3, H3, B, H3, Hs), s));
        hipLaunchKernelGGL(gru_gate_kernel, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, s, gi, gh, lens, B, T, Hs, t, h);
        LADIFF_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(gru_cat_kernel, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, s, h, B, Hs, cat);
    LADIFF_LAUNCH_CHECK();
    // output_net: Linear(2H, H), LayerNorm(H), LeakyReLU(0.2), Linear(H, out)
    LADIFF_TRY(launch_gemm(lin(cat, 2 * Hs, hw.l0.w, 2 * Hs, hw.l0.b, hid, Hs, B, Hs, 2 * Hs), s));
    hipLaunchKernelGGL(ln_lrelu_kernel, dim3((B + 3) / 4), dim3(256), 0, s, hid, hw.norm.g, hw.norm.b, Hs, B, hid + (size_t)B * Hs);
    LADIFF_LAUNCH_CHECK();
    return launch_gemm(lin(hid + (size_t)B * Hs, Hs, hw.l3.w, Hs, hw.l3.b, out, out_dim, B, out_dim, Hs), s);
}

}  // namespace

// ------------------------------------------------------------------ parameter tables (checkpoint keys, finest.tar sub-dicts)
struct T2mMoveW { LinearW conv0, conv3, out_net; };                                       // "movement_encoder"
struct T2mGruW { LinearW input_emb; GruW gru; HeadW head; const float* hidden; };         // "motion_encoder"
struct T2mTextW { LinearW pos_emb, input_emb; GruW gru; HeadW head; const float* hidden; };  // "text_encoder"

static void gru_names(std::vector<std::string>& v) {
    for (const char* sfx : {"", "_reverse"})
        for (const char* n : {"gru.weight_ih_l0", "gru.weight_hh_l0", "gru.bias_ih_l0", "gru.bias_hh_l0"}) v.push_back(std::string(n) + sfx);
    // struct order: w_ih, w_hh, b_ih, b_hh, then the reverse direction
}
static void head_names(std::vector<std::string>& v) {
    for (const char* n : {"output_net.0.weight", "output_net.0.bias", "output_net.1.weight", "output_net.1.bias",
                          "output_net.3.weight", "output_net.3.bias"}) v.push_back(n);
}
const std::vector<std::string>& t2m_move_param_names() {
    static const std::vector<std::string> v = {"main.0.weight", "main.0.bias", "main.3.weight", "main.3.bias",
                                               "out_net.weight", "out_net.bias"};
    return v;
}
const std::vector<std::string>& t2m_motion_param_names() {
    static const std::vector<std::string> names = [] {
        std::vector<std::string> v = {"input_emb.weight", "input_emb.bias"};
        gru_names(v); head_names(v); v.push_back("hidden");
        return v;
    }();
    return names;
}
const std::vector<std::string>& t2m_text_param_names() {
    static const std::vector<std::string> names = [] {
        std::vector<std::string> v = {"pos_emb.weight", "pos_emb.bias", "input_emb.weight", "input_emb.bias"};
        gru_names(v); head_names(v); v.push_back("hidden");
        return v;
    }();
    return names;
}

static_assert(sizeof(T2mMoveW) == 6 * sizeof(void*) && sizeof(T2mGruW) == 17 * sizeof(void*) && sizeof(T2mTextW) == 19 * sizeof(void*),
              "tables are plain pointer arrays in name order");

constexpr int MOVE_H = 512, MOTION_H = 1024, TEXT_H = 512, COEMB = 512, WORD = 300, POS = 15;

// ------------------------------------------------------------------ movement encoder
size_t t2m_move_ws_floats(int B, int F, int Cin) {
    const int T1 = F / 2, T2 = T1 / 2;
    const int K1 = pad32(4 * Cin);
    return (size_t)B * T1 * K1 + (size_t)MOVE_H * K1 + (size_t)B * T1 * MOVE_H + (size_t)B * T2 * 4 * MOVE_H + (size_t)B * T2 * MOVE_H + 64;
}

int t2m_movement_encode(const float* const* w, const float* feats, int ld, int B, int F, int Cin, float* out, float* ws,
                        size_t ws_floats, hipStream_t s) {
    T2mMoveW W; std::memcpy(&W, w, sizeof(W));
    if (Cin < 1 || ld < Cin || F < 4) return LADIFF_ERR_SHAPE;        // two k4 s2 p1 convolutions: F -> F / 2 -> F / 4 frames
    if (ws_floats < t2m_move_ws_floats(B, F, Cin)) return LADIFF_ERR_WORKSPACE;
    if (B == 0) return 0;
    const int T1 = F / 2, T2 = T1 / 2, K1 = pad32(4 * Cin), K2 = 4 * MOVE_H;
    float* a1 = ws;                                   // [B*T1, K1]
    float* w1 = a1 + (size_t)B * T1 * K1;             // conv0 weight padded to K1 columns
    float* y1 = w1 + (size_t)MOVE_H * K1;             // [B*T1, 512]
    float* a2 = y1 + (size_t)B * T1 * MOVE_H;         // [B*T2, 2048]
    float* y2 = a2 + (size_t)B * T2 * K2;             // [B*T2, 512]
    size_t n = (size_t)B * T1 * K1;
    hipLaunchKernelGGL(im2col_k4s2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, feats, F, ld, Cin, T1, K1, n, a1);
    LADIFF_LAUNCH_CHECK();
    LADIFF_TRY(launch_pad_cols(W.conv0.w, w1, MOVE_H, 4 * Cin, K1, s));
    LADIFF_TRY(launch_gemm(lin(a1, K1, w1, K1, W.conv0.b, y1, MOVE_H, B * T1, MOVE_H, K1, ACT_LRELU), s));
    n = (size_t)B * T2 * K2;
    hipLaunchKernelGGL(im2col_k4s2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, y1, T1, MOVE_H, MOVE_H, T2, K2, n, a2);
    LADIFF_LAUNCH_CHECK();
    LADIFF_TRY(launch_gemm(lin(a2, K2, W.conv3.w, K2, W.conv3.b, y2, MOVE_H, B * T2, MOVE_H, K2, ACT_LRELU), s));
    return launch_gemm(lin(y2, MOVE_H, W.out_net.w, MOVE_H, W.out_net.b, out, MOVE_H, B * T2, MOVE_H, MOVE_H), s);
}

// ------------------------------------------------------------------ motion encoder
size_t t2m_motion_ws_floats(int B, int T) { return (size_t)B * T * MOTION_H + gru_head_floats(B, T, MOTION_H); }

int t2m_motion_encode(const float* const* w, const float* mov, const int32_t* m_lens, int B, int T, float* out, float* ws,
                      size_t ws_floats, hipStream_t s) {
    T2mGruW W; std::memcpy(&W, w, sizeof(W));
    if (T < 1) return LADIFF_ERR_SHAPE;
    if (ws_floats < t2m_motion_ws_floats(B, T)) return LADIFF_ERR_WORKSPACE;
    if (B == 0) return 0;
    float* emb = ws;
    LADIFF_TRY(launch_gemm(lin(mov, MOVE_H, W.input_emb.w, MOVE_H, W.input_emb.b, emb, MOTION_H, B * T, MOTION_H, MOVE_H), s));
    return gru_head(emb, m_lens, W.gru, W.hidden, W.head, B, T, MOTION_H, COEMB, out, emb + (size_t)B * T * MOTION_H, s);
}

// ------------------------------------------------------------------ text encoder
size_t t2m_text_ws_floats(int B, int L) {
    const int Kp = pad32(POS), Kw = pad32(WORD);
    return (size_t)B * L * Kp + (size_t)WORD * Kp + (size_t)B * L * Kw + (size_t)TEXT_H * Kw + (size_t)B * L * Kw + (size_t)B * L * TEXT_H +
           gru_head_floats(B, L, TEXT_H);
}

int t2m_text_encode(const float* const* w, const float* word_embs, const float* pos_onehot, const int32_t* cap_lens, int B, int L,
                    float* out, float* ws, size_t ws_floats, hipStream_t s) {
    T2mTextW W; std::memcpy(&W, w, sizeof(W));
    if (L < 1) return LADIFF_ERR_SHAPE;
    if (ws_floats < t2m_text_ws_floats(B, L)) return LADIFF_ERR_WORKSPACE;
    if (B == 0) return 0;
    const int Kp = pad32(POS), Kw = pad32(WORD), M = B * L;
    float* posp = ws;                                 // [M, Kp]
    float* wpos = posp + (size_t)M * Kp;              // pos_emb.weight padded [300, Kp]
    float* wordp = wpos + (size_t)WORD * Kp;          // word_embs padded [M, Kw] (residual of the pos GEMM)
    float* winp = wordp + (size_t)M * Kw;             // input_emb.weight padded [512, Kw]
    float* inp = winp + (size_t)TEXT_H * Kw;          // word_embs + pos_emb(pos_onehot), [M, Kw] (columns >= 300 zero)
    float* emb = inp + (size_t)M * Kw;                // [M, 512]
    LADIFF_TRY(launch_pad_cols(pos_onehot, posp, M, POS, Kp, s));
    LADIFF_TRY(launch_pad_cols(W.pos_emb.w, wpos, WORD, POS, Kp, s));
    LADIFF_TRY(launch_pad_cols(word_embs, wordp, M, WORD, Kw, s));
    LADIFF_TRY(launch_pad_cols(W.input_emb.w, winp, TEXT_H, WORD, Kw, s));
    LADIFF_HIP(hipMemsetAsync(inp, 0, (size_t)M * Kw * sizeof(float), s));
    {   // inputs = word_embs + pos_emb(pos_onehot)                                  t2m_textenc.py:36-37
        GemmArgs g = lin(posp, Kp, wpos, Kp, W.pos_emb.b, inp, Kw, M, WORD, Kp);
        g.res = wordp; g.ldres = Kw;
        LADIFF_TRY(launch_gemm(g, s));
    }
    LADIFF_TRY(launch_gemm(lin(inp, Kw, winp, Kw, W.input_emb.b, emb, TEXT_H, M, TEXT_H, Kw), s));
    return gru_head(emb, cap_lens, W.gru, W.hidden, W.head, B, L, TEXT_H, COEMB, out, emb + (size_t)M * TEXT_H, s);
}

}  // namespace ladiff
