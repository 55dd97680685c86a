// The denoiser loop as ONE persistent, weight-stationary pipeline over the whole chip (bf16x3 mode, guidance on).
//
// Why: a guided DDIM step is 9 layers x ~8 dependent matrix stages on only M = 2*B*T = 1280 rows.  As separate launches
// (denoiser.hip: 77 per step) every stage costs a kernel boundary, a ramp-up in which each of ~256 workgroups re-fetches
// its weight tile from the Infinity Cache, a few hundred MFMA cycles and a ramp-down: 7 us per stage, 550 us per step, 4 %
// of the MFMA peak (profiles/r1).  Here the roles are turned around.  Every CU is ONE stage of the network for the whole
// loop and keeps its slice of the weights in REGISTERS (all in-loop weights are 51 MB in S-format; the chip has 128 MB of
// VGPRs): 27 CUs per layer -
//     QKV  x4  in_proj rows of one head (192x256) + the 7-key attention of that head     mdiff_transformer.py:57-61, :296-313
//     OUT  x1  attention out-projection + residual + norm1                               :62-63
//     LIN  x8  linear1 (128 hidden columns, ReLU) + that slice's part of linear2          :64
//     RED2 x3  sum of the 8 partial products + bias + residual + norm2 + hoisted ca_block :65-66, :219-247
//     FFN  x8  ffn.linear1 (128 columns, GELU) + that slice's part of ffn.linear2         :259-260
//     STYL x3  sum of the 8 partials + StylizationBlock (LN, AdaLN, SiLU, 256x256 out) + residual   :152-162, :261
// - plus SKIP x2 on the four output layers (linear_blocks on cat(x, skip), cross_attention.py:79-82) and one TAIL CU
// (encoder.norm, guidance, scheduler step, next input: ladiff.py:472-492).  The ACTIVATIONS flow: the batch is cut into
// blocks of P prompts (both guidance branches, 2*P*T <= 32 rows = two MFMA row tiles); a block's rows travel from stage to
// stage through global memory, and every block is at a different stage, so all stages work at once.  Prompts never mix
// (attention is per sample, LayerNorm per row, guidance pairs the two branches of one prompt), so a block needs nothing
// but its own previous stage - there is no grid-wide barrier anywhere, and the 50 steps are one launch.
//
// Hand-off (MI355X_MICROARCH.md "inter-workgroup visibility", cdna_hip_programming.md G16 R1): the eight L2s are not
// coherent, so a producer stores its rows write-through (`buffer_store ... sc1`), every storing wave drains
// (`s_waitcnt vmcnt(0)`), the workgroup meets at a barrier, and ONE lane publishes an epoch in a flag word (agent-scope
// relaxed store); the consumer's wave 0 polls that word (agent-scope relaxed loads, `s_sleep` between polls), the
// workgroup meets at a barrier, and EVERY load of handed-off bytes is a `buffer_load ... sc1` to registers (never LDS-DMA,
// never a plain load).  Epoch = local step + 1; flag words are zeroed by a memset node before every launch.  A buffer is
// rewritten one step later, by which time its consumer has long finished with it (the rewrite depends on it through the
// chain of flags), so there is no back-pressure channel.  Every spin is bounded (wall clock); a timeout raises an abort word
// that all pollers watch, and the host reports it.
//
// Numerics: the same arithmetic as the launch-per-stage bf16x3 path (S-format operands, hi*hi + hi*lo + lo*hi on
// v_mfma_f32_16x16x32_bf16, fp32 accumulation and fp32 everything else); only the summation order of the split products
// differs (8 hidden slices instead of 4 K-slices).
#include <cstring>
#include <vector>

#include "model.h"

namespace ladiff {

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int RT = 32;                    // rows of a block tile
constexpr int CLD = D + 4;                // row stride of the fp32 staging tile in LDS (floats)
constexpr int NSLICE = 8;                 // hidden slices of the two MLPs (128 columns each)
constexpr int HS = FF / NSLICE;           // 128
constexpr int NRED = 3;                   // row parts of the two reduce stages
constexpr int FLAG_SLOTS = 8;
constexpr int SYS_LDS_BYTES = 100 * 1024;   // > 80 KiB: one workgroup per CU, so the <= 256 workgroups sit on distinct CUs
constexpr int GROUPS_PER_LAYER = 7;
enum Group : int { G_XIN = 0, G_ATT = 1, G_X1 = 2, G_PC = 3, G_X2 = 4, G_PE = 5, G_XO = 6 };
enum Role : int { R_QKV = 0, R_OUT = 1, R_LIN = 2, R_RED2 = 3, R_FFN = 4, R_STYL = 5, R_SKIP = 6, R_TAIL = 7 };

struct Stage {                            // one per workgroup
    int role, layer, slice, act;
    int wait_group, wait_n, out_group, out_slot;
    const float *w0, *w1;                 // S-format matrices
    const float *b0, *b1;                 // biases
    const float *g, *be;                  // LayerNorm gamma / beta
    const float *in0, *in1, *in2;         // block-layout activations: [NB][RT][256] (partials: [8][NB][RT][256])
    float* out;
};

struct SysArgs {
    const Stage* stages;
    unsigned* flags;                      // [groups][NB][FLAG_SLOTS]
    unsigned* status;                     // [0] abort code (0 = ok), [1] diagnostic
    const float* tables;                  // time tables [n_total][9][1536]
    const float* tkv;                     // text K|V [9][2B][512]
    const float* ctab;                    // hoisted cross-attention [9][n_chunk][2B+1][256]
    const float* coef; const float* noise; const float* pe; const float* ng; const float* nb;
    float* lat;                           // latents [B][T][256]
    const int32_t* counts;
    float gscale;
    int B, T, P, NB, step_lo, n_steps, n_ctab;
};

// ---------------------------------------------------------------- hand-off primitives
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xffffffffu, 0x00020000);
}
__device__ __forceinline__ f32x4 ld_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) {       // 16 bytes, bypasses this CU's L1
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16));
}
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v) { // 16 bytes, write-through
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, 16);
}
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr unsigned long long TIMEOUT_TICKS = 150000000ull;     // s_memrealtime runs at 100 MHz: 1.5 s per wait

// wave 0 polls flags[0 .. n) until all are >= epoch; everybody leaves through the barrier.  Returns false on abort.
__device__ __forceinline__ bool wait_epoch(const unsigned* flags, int n, unsigned epoch, unsigned* status, int* lds_abort) {
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        int bad = 0;
        for (unsigned spins = 1;; ++spins) {
            unsigned v = epoch;
            if (lane < n) v = __hip_atomic_load((const gu32*)flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(v >= epoch)) break;
            if ((spins & 31u) == 0u) {
                const unsigned a = __hip_atomic_load((const gu32*)status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a != 0u) { bad = 1; break; }
                if (__builtin_amdgcn_s_memrealtime() - t0 > TIMEOUT_TICKS) {
                    if (lane == 0) {
                        __hip_atomic_store((gu32*)status + 1, (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store((gu32*)status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    bad = 1;
                    break;
                }
            }
            __builtin_amdgcn_s_sleep(2);
        }
        if (lane == 0) *lds_abort = bad;
    }
    __syncthreads();
    return *lds_abort == 0;
}

// all rows of this workgroup are stored: drain (every wave), meet, publish
__device__ __forceinline__ void publish(unsigned* flag, unsigned epoch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store((gu32*)flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ unsigned* flag_of(const SysArgs& p, int group, int b, int slot) {      // group = layer * 7 + Group
    return p.flags + ((size_t)(group * p.NB + b) * FLAG_SLOTS + slot);
}

// ---------------------------------------------------------------- LDS images
// S-format operand tile: row = KB blocks of 256 B, block = 8 hi slots + 8 lo slots of 16 B, slot index XORed with (row & 15)
template <int KB>
__device__ __forceinline__ char* a_slot(char* tile, int row, int kb, int slot) {
    return tile + row * (KB * 256) + kb * 256 + (((slot ^ row) & 15) << 4);
}
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hi[e] = (__bf16)a[e]; lo[e] = (__bf16)(a[e] - (float)hi[e]);
        hi[4 + e] = (__bf16)b[e]; lo[4 + e] = (__bf16)(b[e] - (float)hi[4 + e]);
    }
}
// fill 256 columns (4 k-blocks starting at kb0) of the RT-row operand tile from a block of fp32 rows in global memory
template <int KB>
__device__ __forceinline__ void fill_a256(char* tile, int kb0, __amdgpu_buffer_rsrc_t r, unsigned base) {
    f32x4 v[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int id = threadIdx.x + 256 * u, row = id >> 5, c8 = id & 31;
        v[u][0] = ld_sc1(r, base + row * 1024 + c8 * 32);
        v[u][1] = ld_sc1(r, base + row * 1024 + c8 * 32 + 16);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int id = threadIdx.x + 256 * u, row = id >> 5, c8 = id & 31;
        bf16x8 hi, lo;
        split8(v[u][0], v[u][1], hi, lo);
        *reinterpret_cast<bf16x8*>(a_slot<KB>(tile, row, kb0 + (c8 >> 3), c8 & 7)) = hi;
        *reinterpret_cast<bf16x8*>(a_slot<KB>(tile, row, kb0 + (c8 >> 3), 8 + (c8 & 7))) = lo;
    }
}

// weights of NT column tiles x KS k-steps for this wave, register resident
template <int NT, int KS>
struct WFrag { bf16x8 hi[NT][KS], lo[NT][KS]; };

// w: S-format matrix, row stride ldw floats; tile j covers matrix rows row_of(j) + (lane & 15); k-steps start at k block kb0
template <int NT, int KS, class RowOf>
__device__ __forceinline__ void load_w(WFrag<NT, KS>& f, const float* w, int ldw, int kb0, RowOf row_of) {
    const int lane = threadIdx.x & 63, frow = lane & 15, fk = lane >> 4;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const char* rp = reinterpret_cast<const char*>(w + (size_t)(row_of(j) + frow) * ldw);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const char* bp = rp + (kb0 + (s >> 1)) * 256 + ((4 * (s & 1) + fk) << 4);
            f.hi[j][s] = *reinterpret_cast<const bf16x8*>(bp);
            f.lo[j][s] = *reinterpret_cast<const bf16x8*>(bp + 128);
        }
    }
}

// acc[i][j] += A(tile rows 16 i ..) . W(tile j)^T over KS k-steps; MR = row tiles used
template <int KB, int NT, int KS, int MR>
__device__ __forceinline__ void mma(const char* tile, const WFrag<NT, KS>& f, f32x4 (&acc)[MR][NT]) {
    const int lane = threadIdx.x & 63, frow = lane & 15, fk = lane >> 4;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        bf16x8 ah[MR], al[MR];
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            const int row = 16 * i + frow;
            ah[i] = *reinterpret_cast<const bf16x8*>(a_slot<KB>(const_cast<char*>(tile), row, s >> 1, 4 * (s & 1) + fk));
            al[i] = *reinterpret_cast<const bf16x8*>(a_slot<KB>(const_cast<char*>(tile), row, s >> 1, 8 + 4 * (s & 1) + fk));
        }
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], f.hi[j][s], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], f.lo[j][s], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], f.hi[j][s], acc[i][j], 0, 0, 0);
            }
    }
}

template <int MR, int NT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[MR][NT]) {
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// accumulators -> fp32 staging tile [rows][CLD]; tile j of this wave holds columns col_of(j) + (lane & 15)
template <int MR, int NT, class ColOf>
__device__ __forceinline__ void stage_c(float* ct, const f32x4 (&acc)[MR][NT], ColOf col_of) {
    const int lane = threadIdx.x & 63, frow = lane & 15, fk = lane >> 4;
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) ct[(16 * i + 4 * fk + r) * CLD + col_of(j) + frow] = acc[i][j][r];
}

__device__ __forceinline__ void row_stats4(const f32x4 v, float& mean, float& rstd) {   // as rowops.hip: two-pass, fp32
    const float s = wave_sum(v[0] + v[1] + v[2] + v[3]);
    mean = s * (1.f / 256.f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float d = v[i] - mean; q += d * d; }
    q = wave_sum(q);
    rstd = rsqrtf(q * (1.f / 256.f) + LN_EPS);
}

// block geometry: row r of block b = (branch br, prompt b*P + pl, latent t), r = (br*P + pl)*T + t
struct RowInfo { int valid, b2, prompt, t; };
__device__ __forceinline__ RowInfo row_info(const SysArgs& p, int b, int r) {
    RowInfo o;
    const int sb = r / p.T;
    o.t = r - sb * p.T;
    const int br = sb / p.P, pl = sb - br * p.P;
    o.prompt = b * p.P + pl;
    o.valid = (br < 2) && (o.prompt < p.B);
    o.b2 = br * p.B + o.prompt;
    return o;
}

// ---------------------------------------------------------------- roles
// QKV: one head.  in_proj rows {q,k,v} x 64 of head `slice` on the block, then softmax(q k^T / 8) v over the T latent
// keys of the sample (masked by its latent count), the text token and the time token.
__device__ __forceinline__ void run_qkv(const SysArgs& p, const Stage& st, char* lds, int* lds_abort) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, frow = lane & 15;
    const int h = st.slice, T = p.T, nkeys = p.T + 2;
    constexpr int QLD = 196;
    char* const atile = lds;                                              // 32 KiB operand tile
    float* const qt = reinterpret_cast<float*>(lds + 32768);              // [RT][QLD] q | k | v (fp32)
    float* const xt = qt + RT * QLD;                                      // [RT / 1 + 1][128] text k|v per sample-branch, time k|v last
    float* const sc = xt + (RT + 1) * 128;                                // [RT][16] scores
    WFrag<3, 8> wf;
    // tile column tc = 48 wave + 16 j + frow: part tc / 64 (q, k, v), matrix row part * 256 + h * 64 + tc % 64
    load_w(wf, st.w0, D, 0, [&](int j) { const int tc = 48 * wave + 16 * j; return (tc >> 6) * 256 + h * 64 + (tc & 63); });
    float bcol[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { const int tc = 48 * wave + 16 * j + frow; bcol[j] = st.b0[(tc >> 6) * 256 + h * 64 + (tc & 63)]; }
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(st.in0), rout = rsrc_of(st.out);
    const float* tkv = p.tkv + (size_t)st.layer * 2 * p.B * 512;
    const int nsb = 2 * p.P;                                              // sample-branches per block
    for (int s = 0; s < p.n_steps; ++s) {
        const float* timekv = p.tables + (size_t)(p.step_lo + s) * DEN_STEP_STRIDE + st.layer * DEN_LAYER_STRIDE + DEN_OFF_TIME_KV;
        for (int b = 0; b < p.NB; ++b) {
            // text / time K|V slices of this head (read-only data of earlier kernels: plain loads), issued before the wait
            f32x4 xk[2];
            int xi[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int f4 = tid + 256 * u, sx = f4 >> 5, c4 = (f4 & 31) * 4;     // [nsb + 1][128]
                xi[u] = sx <= nsb ? f4 : -1;
                xk[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (sx < nsb) {
                    const int br = sx / p.P, prompt = b * p.P + (sx - br * p.P);
                    if (prompt < p.B) xk[u] = ld4(tkv + (size_t)(br * p.B + prompt) * 512 + (c4 < 64 ? c4 : 192 + c4) + h * 64);
                } else if (sx == nsb) {
                    xk[u] = ld4(timekv + (c4 < 64 ? c4 : 192 + c4) + h * 64);
                }
            }
            if (!wait_epoch(flag_of(p, st.wait_group, b, 0), st.wait_n, s + 1, p.status, lds_abort)) return;
            fill_a256<4>(atile, 0, rin, (unsigned)b * RT * 1024);
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (xi[u] >= 0) st4(xt + xi[u] * 4, xk[u]);
            __syncthreads();
            f32x4 acc[2][3];
            zero_acc(acc);
            mma<4, 3, 8, 2>(atile, wf, acc);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int tc = 48 * wave + 16 * j + frow;
                    const float scl = tc < 64 ? 0.125f : 1.f;           // q / sqrt(64), exact
#pragma unroll
                    for (int r = 0; r < 4; ++r) qt[(16 * i + 4 * (lane >> 4) + r) * QLD + tc] = (acc[i][j][r] + bcol[j]) * scl;
                }
            __syncthreads();
            const int nrows = nsb * T;
            for (int u = tid; u < nrows * nkeys; u += 256) {             // one thread per (row, key)
                const int row = u / nkeys, j = u - row * nkeys, sx = row / T;
                const int br = sx / p.P, prompt = b * p.P + (sx - br * p.P);
                int nv = T;
                if (p.counts != nullptr && prompt < p.B) { nv = p.counts[prompt]; nv = nv > T ? T : nv; }
                const float* kp = j < T ? qt + (sx * T + j) * QLD + 64 : (j == T ? xt + sx * 128 : xt + nsb * 128);
                const float* qp = qt + row * QLD;
                float d = 0.f;
#pragma unroll
                for (int c = 0; c < 64; c += 4) {
                    const f32x4 a = ld4(qp + c), k4 = ld4(kp + c);
                    d = fmaf(a[0], k4[0], d); d = fmaf(a[1], k4[1], d); d = fmaf(a[2], k4[2], d); d = fmaf(a[3], k4[3], d);
                }
                sc[row * 16 + j] = (j < T && j >= nv) ? -INFINITY : d;
            }
            __syncthreads();
            if (tid < nrows) {
                float e[LADIFF_MAX_LATENTS + 2];
                float m = -INFINITY;
#pragma unroll
                for (int j = 0; j < LADIFF_MAX_LATENTS + 2; ++j) { e[j] = j < nkeys ? sc[tid * 16 + j] : -INFINITY; m = fmaxf(m, e[j]); }
                float l = 0.f;
#pragma unroll
                for (int j = 0; j < LADIFF_MAX_LATENTS + 2; ++j) { e[j] = j < nkeys ? expf(e[j] - m) : 0.f; l += e[j]; }
                const float inv = 1.f / l;
#pragma unroll
                for (int j = 0; j < LADIFF_MAX_LATENTS + 2; ++j)
                    if (j < nkeys) sc[tid * 16 + j] = e[j] * inv;
            }
            __syncthreads();
            for (int u = tid; u < RT * 16; u += 256) {                   // one thread per (row, 4 columns)
                const int row = u >> 4, c4 = (u & 15) * 4;
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                if (row < nrows) {
                    const int sx = row / T;
                    for (int j = 0; j < nkeys; ++j) {
                        const float pj = sc[row * 16 + j];
                        const float* vp = j < T ? qt + (sx * T + j) * QLD + 128 : (j == T ? xt + sx * 128 + 64 : xt + nsb * 128 + 64);
                        const f32x4 v = ld4(vp + c4);
                        o[0] = fmaf(pj, v[0], o[0]); o[1] = fmaf(pj, v[1], o[1]); o[2] = fmaf(pj, v[2], o[2]); o[3] = fmaf(pj, v[3], o[3]);
                    }
                }
                st_sc1(rout, ((unsigned)b * RT + row) * 1024 + (h * 64 + c4) * 4, o);
            }
            publish(flag_of(p, st.out_group, b, st.out_slot), s + 1);
        }
    }
}

// OUT: X1 = LN1(x + out_proj(att))
__device__ __forceinline__ void run_out(const SysArgs& p, const Stage& st, char* lds, int* lds_abort) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char* const atile = lds;
    float* const ct = reinterpret_cast<float*>(lds + 32768);
    WFrag<4, 8> wf;
    load_w(wf, st.w0, D, 0, [&](int j) { return 64 * wave + 16 * j; });
    const int c = 4 * lane;
    const f32x4 bias = ld4(st.b0 + c), gg = ld4(st.g + c), bb = ld4(st.be + c);
    const __amdgpu_buffer_rsrc_t ratt = rsrc_of(st.in0), rx = rsrc_of(st.in1), rout = rsrc_of(st.out);
    for (int s = 0; s < p.n_steps; ++s)
        for (int b = 0; b < p.NB; ++b) {
            if (!wait_epoch(flag_of(p, st.wait_group, b, 0), st.wait_n, s + 1, p.status, lds_abort)) return;
            const unsigned base = (unsigned)b * RT * 1024;
            f32x4 res[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) res[q] = ld_sc1(rx, base + (wave + 4 * q) * 1024 + c * 4);
            fill_a256<4>(atile, 0, ratt, base);
            __syncthreads();
            f32x4 acc[2][4];
            zero_acc(acc);
            mma<4, 4, 8, 2>(atile, wf, acc);
            stage_c(ct, acc, [&](int j) { return 64 * wave + 16 * j; });
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int row = wave + 4 * q;
                f32x4 v = ld4(ct + row * CLD + c);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = v[i] + bias[i] + res[q][i];
                float mean, rstd;
                row_stats4(v, mean, rstd);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (v[i] - mean) * rstd * gg[i] + bb[i];
                st_sc1(rout, base + row * 1024 + c * 4, v);
            }
            publish(flag_of(p, st.out_group, b, st.out_slot), s + 1);
        }
}

// LIN / FFN: hidden slice = act(x W1_slice^T + b1_slice) (128 columns), partial = hidden . W2[:, slice]^T (256 columns)
template <int ACT>
__device__ __forceinline__ void run_mlp(const SysArgs& p, const Stage& st, char* lds, int* lds_abort) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, frow = lane & 15, fk = lane >> 4;
    char* const atile = lds;                                            // [RT] x K=256
    char* const htile = lds + 32768;                                    // [RT] x K=128 (hidden slice, S-format)
    float* const ct = reinterpret_cast<float*>(lds + 32768 + 16384);
    const int j0 = st.slice * HS;
    WFrag<2, 8> w1;
    WFrag<4, 4> w2;
    load_w(w1, st.w0, D, 0, [&](int j) { return j0 + 32 * wave + 16 * j; });
    load_w(w2, st.w1, FF, 2 * st.slice, [&](int j) { return 64 * wave + 16 * j; });
    float b1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) b1[j] = st.b0[j0 + 32 * wave + 16 * j + frow];
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(st.in0), rout = rsrc_of(st.out);
    const unsigned plane = (unsigned)st.slice * p.NB * RT * 1024;
    for (int s = 0; s < p.n_steps; ++s)
        for (int b = 0; b < p.NB; ++b) {
            if (!wait_epoch(flag_of(p, st.wait_group, b, 0), st.wait_n, s + 1, p.status, lds_abort)) return;
            const unsigned base = (unsigned)b * RT * 1024;
            fill_a256<4>(atile, 0, rin, base);
            __syncthreads();
            f32x4 acc1[2][2];
            zero_acc(acc1);
            mma<4, 2, 8, 2>(atile, w1, acc1);
            // hidden slice -> S-format operand tile (k = hidden column within the slice)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int k = 32 * wave + 16 * j + frow;             // 0..127
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * i + 4 * fk + r;
                        const float v = act_c<ACT>(acc1[i][j][r] + b1[j]);
                        const __bf16 hi = (__bf16)v, lo = (__bf16)(v - (float)hi);
                        __bf16* hp = reinterpret_cast<__bf16*>(a_slot<2>(htile, row, k >> 6, (k & 63) >> 3)) + (k & 7);
                        __bf16* lp = reinterpret_cast<__bf16*>(a_slot<2>(htile, row, k >> 6, 8 + ((k & 63) >> 3))) + (k & 7);
                        *hp = hi; *lp = lo;
                    }
                }
            __syncthreads();
            f32x4 acc2[2][4];
            zero_acc(acc2);
            mma<2, 4, 4, 2>(htile, w2, acc2);
            stage_c(ct, acc2, [&](int j) { return 64 * wave + 16 * j; });
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int row = wave + 4 * q;
                st_sc1(rout, plane + base + row * 1024 + lane * 16, ld4(ct + row * CLD + 4 * lane));
            }
            publish(flag_of(p, st.out_group, b, st.out_slot), s + 1);
        }
}

// rows of a block handled by reduce part `part`: [lo, hi)
__device__ __forceinline__ void part_rows(int part, int& lo, int& hi) {
    constexpr int PER = (RT + NRED - 1) / NRED;       // 11
    lo = part * PER;
    hi = lo + PER < RT ? lo + PER : RT;
}

// RED2: X2 = LN2(X1 + sum_j partial_j + b2) + c[step, layer, sample | pad]
__device__ __forceinline__ void run_red2(const SysArgs& p, const Stage& st, char* lds, int* lds_abort) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = 4 * lane;
    const f32x4 bias = ld4(st.b0 + c), gg = ld4(st.g + c), bb = ld4(st.be + c);
    const __amdgpu_buffer_rsrc_t rp = rsrc_of(st.in0), rx = rsrc_of(st.in1), rout = rsrc_of(st.out);
    int lo, hi;
    part_rows(st.slice, lo, hi);
    const unsigned pstride = (unsigned)p.NB * RT * 1024;
    const int R = 2 * p.B + 1;
    for (int s = 0; s < p.n_steps; ++s) {
        const float* ct = p.ctab + ((size_t)st.layer * p.n_ctab + s) * R * D;
        for (int b = 0; b < p.NB; ++b) {
            if (!wait_epoch(flag_of(p, st.wait_group, b, 0), st.wait_n, s + 1, p.status, lds_abort)) return;
            const unsigned base = (unsigned)b * RT * 1024;
            for (int row = lo + wave; row < hi; row += 4) {
                f32x4 pl[NSLICE];
#pragma unroll
                for (int j = 0; j < NSLICE; ++j) pl[j] = ld_sc1(rp, j * pstride + base + row * 1024 + c * 4);
                const f32x4 rs = ld_sc1(rx, base + row * 1024 + c * 4);
                const RowInfo ri = row_info(p, b, row);
                int cnt = 0x7fffffff;
                if (p.counts != nullptr && ri.valid) cnt = p.counts[ri.prompt];
                const int trow = (ri.valid && ri.t < cnt) ? ri.b2 : 2 * p.B;
                const f32x4 tv = ld4(ct + (size_t)trow * D + c);
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    v[i] = (((pl[0][i] + pl[1][i]) + (pl[2][i] + pl[3][i])) + ((pl[4][i] + pl[5][i]) + (pl[6][i] + pl[7][i]))) + bias[i] + rs[i];
                float mean, rstd;
                row_stats4(v, mean, rstd);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (v[i] - mean) * rstd * gg[i] + bb[i] + tv[i];
                st_sc1(rout, base + row * 1024 + c * 4, v);
            }
            publish(flag_of(p, st.out_group, b, st.out_slot), s + 1);
        }
    }
}

// STYL: x' = X2 + out( SiLU( LN(sum_j partial_j + b2) * (1 + scale_t) + shift_t ) )
__device__ __forceinline__ void run_styl(const SysArgs& p, const Stage& st, char* lds, int* lds_abort) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char* const atile = lds;                                            // 16 rows used
    float* const ct = reinterpret_cast<float*>(lds + 32768);
    WFrag<4, 8> wf;
    load_w(wf, st.w0, D, 0, [&](int j) { return 64 * wave + 16 * j; });
    const int c = 4 * lane;
    const f32x4 bias2 = ld4(st.b1 + c), bias = ld4(st.b0 + c), gg = ld4(st.g + c), bb = ld4(st.be + c);
    const __amdgpu_buffer_rsrc_t rp = rsrc_of(st.in0), rx = rsrc_of(st.in1), rout = rsrc_of(st.out);
    int lo, hi;
    part_rows(st.slice, lo, hi);
    const unsigned pstride = (unsigned)p.NB * RT * 1024;
    for (int s = 0; s < p.n_steps; ++s) {
        const float* mod = p.tables + (size_t)(p.step_lo + s) * DEN_STEP_STRIDE + st.layer * DEN_LAYER_STRIDE + DEN_OFF_FFN_MOD;
        const f32x4 scl = ld4(mod + c), shf = ld4(mod + D + c);
        for (int b = 0; b < p.NB; ++b) {
            if (!wait_epoch(flag_of(p, st.wait_group, b, 0), st.wait_n, s + 1, p.status, lds_abort)) return;
            const unsigned base = (unsigned)b * RT * 1024;
            f32x4 res[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int lr = wave + 4 * q, row = lo + lr;              // local row 0..11 of this part
                res[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row < hi) {
                    f32x4 pl[NSLICE];
#pragma unroll
                    for (int j = 0; j < NSLICE; ++j) pl[j] = ld_sc1(rp, j * pstride + base + row * 1024 + c * 4);
                    res[q] = ld_sc1(rx, base + row * 1024 + c * 4);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        v[i] = (((pl[0][i] + pl[1][i]) + (pl[2][i] + pl[3][i])) + ((pl[4][i] + pl[5][i]) + (pl[6][i] + pl[7][i]))) + bias2[i];
                    float mean, rstd;
                    row_stats4(v, mean, rstd);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = silu(((v[i] - mean) * rstd * gg[i] + bb[i]) * (1.f + scl[i]) + shf[i]);
                }
                // u row -> S-format operand tile (local row lr; rows 12..15 of the tile are zero)
                bf16x4 h4, l4;
#pragma unroll
                for (int e = 0; e < 4; ++e) { h4[e] = (__bf16)v[e]; l4[e] = (__bf16)(v[e] - (float)h4[e]); }
                *reinterpret_cast<bf16x4*>(a_slot<4>(atile, lr, c >> 6, (c & 63) >> 3) + (c & 7) * 2) = h4;
                *reinterpret_cast<bf16x4*>(a_slot<4>(atile, lr, c >> 6, 8 + ((c & 63) >> 3)) + (c & 7) * 2) = l4;
            }
            {   // local rows 12..15: zero
                const int lr = 12 + wave;
                const bf16x4 z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
                *reinterpret_cast<bf16x4*>(a_slot<4>(atile, lr, c >> 6, (c & 63) >> 3) + (c & 7) * 2) = z;
                *reinterpret_cast<bf16x4*>(a_slot<4>(atile, lr, c >> 6, 8 + ((c & 63) >> 3)) + (c & 7) * 2) = z;
            }
            __syncthreads();
            f32x4 acc[1][4];
            zero_acc(acc);
            mma<4, 4, 8, 1>(atile, wf, acc);
            stage_c(ct, acc, [&](int j) { return 64 * wave + 16 * j; });
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int lr = wave + 4 * q, row = lo + lr;
                if (row < hi) {
                    f32x4 v = ld4(ct + lr * CLD + c);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = v[i] + bias[i] + res[q][i];
                    st_sc1(rout, base + row * 1024 + c * 4, v);
                }
            }
            publish(flag_of(p, st.out_group, b, st.out_slot), s + 1);
        }
    }
}

// SKIP: half of the 256 output columns of linear_blocks[i](cat(x, skip))
__device__ __forceinline__ void run_skip(const SysArgs& p, const Stage& st, char* lds, int* lds_abort) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char* const atile = lds;                                            // [RT] x K=512: 64 KiB
    float* const ct = reinterpret_cast<float*>(lds + 65536);
    const int n0 = st.slice * 128;
    WFrag<2, 16> wf;
    load_w(wf, st.w0, 2 * D, 0, [&](int j) { return n0 + 32 * wave + 16 * j; });
    const __amdgpu_buffer_rsrc_t rx = rsrc_of(st.in0), rs = rsrc_of(st.in1), rout = rsrc_of(st.out);
    for (int s = 0; s < p.n_steps; ++s)
        for (int b = 0; b < p.NB; ++b) {
            if (!wait_epoch(flag_of(p, st.wait_group, b, 0), st.wait_n, s + 1, p.status, lds_abort)) return;
            const unsigned base = (unsigned)b * RT * 1024;
            fill_a256<8>(atile, 0, rx, base);
            fill_a256<8>(atile, 4, rs, base);
            __syncthreads();
            f32x4 acc[2][2];
            zero_acc(acc);
            mma<8, 2, 16, 2>(atile, wf, acc);
            stage_c(ct, acc, [&](int j) { return n0 + 32 * wave + 16 * j; });
            __syncthreads();
            for (int u = tid; u < RT * 32; u += 256) {                   // (row, 4 columns) of this half
                const int row = u >> 5, cc = n0 + (u & 31) * 4;
                f32x4 v = ld4(ct + row * CLD + cc);
                const f32x4 bv = ld4(st.b0 + cc);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] += bv[i];
                st_sc1(rout, base + row * 1024 + cc * 4, v);
            }
            publish(flag_of(p, st.out_group, b, st.out_slot), s + 1);
        }
}

// TAIL: encoder.norm on both branches, guidance, scheduler step, latents, next step's network input (x = latents + pe)
__device__ __forceinline__ void run_tail(const SysArgs& p, const Stage& st, char* lds, int* lds_abort) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = 4 * lane, T = p.T, P = p.P;
    const f32x4 gg = ld4(p.ng + c), bb = ld4(p.nb + c);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(st.in0), rout = rsrc_of(st.out);
    const int M = p.B * T;
    // local step 0: the first network input from the latents the prologue left (plain memory of earlier kernels)
    for (int b = 0; b < p.NB; ++b) {
        const unsigned base = (unsigned)b * RT * 1024;
        for (int q = wave; q < RT; q += 4) {
            f32x4 xn = {0.f, 0.f, 0.f, 0.f};
            const RowInfo ri = row_info(p, b, q);
            if (ri.valid) {
                const f32x4 l = ld4(p.lat + ((size_t)ri.prompt * T + ri.t) * D + c), pe = ld4(p.pe + (size_t)ri.t * D + c);
#pragma unroll
                for (int i = 0; i < 4; ++i) xn[i] = l[i] + pe[i];
            }
            st_sc1(rout, base + q * 1024 + c * 4, xn);
        }
        publish(flag_of(p, st.out_group, b, st.out_slot), 1);
    }
    for (int s = 0; s < p.n_steps; ++s) {
        const int step = p.step_lo + s;
        const float* cf = p.coef + (size_t)step * LADIFF_COEF_STRIDE;
        const float sa = cf[0], sb = cf[1], kx0 = cf[2], kx = cf[3], ke = cf[4], kn = cf[5];
        for (int b = 0; b < p.NB; ++b) {
            if (!wait_epoch(flag_of(p, st.wait_group, b, 0), st.wait_n, s + 1, p.status, lds_abort)) return;
            const unsigned base = (unsigned)b * RT * 1024;
            for (int q = wave; q < P * T; q += 4) {                      // (prompt in block, latent) pairs
                const int pl = q / T, t = q - pl * T, prompt = b * P + pl;
                if (prompt >= p.B) continue;
                const int ru = pl * T + t, rc = (P + pl) * T + t;
                f32x4 eu = ld_sc1(rin, base + ru * 1024 + c * 4), ec = ld_sc1(rin, base + rc * 1024 + c * 4);
                float mean, rstd;
                row_stats4(eu, mean, rstd);
#pragma unroll
                for (int i = 0; i < 4; ++i) eu[i] = (eu[i] - mean) * rstd * gg[i] + bb[i];
                row_stats4(ec, mean, rstd);
#pragma unroll
                for (int i = 0; i < 4; ++i) ec[i] = (ec[i] - mean) * rstd * gg[i] + bb[i];
                const size_t lrow = (size_t)prompt * T + t;
                f32x4 l = ld4(p.lat + lrow * D + c);
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                if (p.noise != nullptr && kn != 0.f) z = ld4(p.noise + ((size_t)step * M + lrow) * D + c);
                const f32x4 pe = ld4(p.pe + (size_t)t * D + c);
                f32x4 xn;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float e = eu[i] + p.gscale * (ec[i] - eu[i]);
                    const float x0 = (l[i] - sb * e) / sa;
                    l[i] = kx0 * x0 + kx * l[i] + ke * e + kn * z[i];
                    xn[i] = l[i] + pe[i];
                }
                st4(p.lat + lrow * D + c, l);
                if (s + 1 < p.n_steps) {
                    st_sc1(rout, base + ru * 1024 + c * 4, xn);
                    st_sc1(rout, base + rc * 1024 + c * 4, xn);
                }
            }
            if (s + 1 < p.n_steps) publish(flag_of(p, st.out_group, b, st.out_slot), s + 2);
            else __syncthreads();
        }
    }
}

}  // namespace

__global__ __launch_bounds__(256, 1) void systolic_loop_kernel(const SysArgs p) {
    // all LDS is dynamic (a static variable would shift the dynamic base off its 16-byte alignment, cdna_hip_programming.md G17)
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    int& lds_abort = *reinterpret_cast<int*>(lds + SYS_LDS_BYTES - 16);
    const Stage st = p.stages[blockIdx.x];
    switch (st.role) {
        case R_QKV: run_qkv(p, st, lds, &lds_abort); break;
        case R_OUT: run_out(p, st, lds, &lds_abort); break;
        case R_LIN: run_mlp<ACT_RELU>(p, st, lds, &lds_abort); break;
        case R_RED2: run_red2(p, st, lds, &lds_abort); break;
        case R_FFN: run_mlp<ACT_GELU>(p, st, lds, &lds_abort); break;
        case R_STYL: run_styl(p, st, lds, &lds_abort); break;
        case R_SKIP: run_skip(p, st, lds, &lds_abort); break;
        case R_TAIL: run_tail(p, st, lds, &lds_abort); break;
        default: break;
    }
}


// ================================================================== host side
namespace {
struct SysLayout {
    size_t blk;                   // floats of one [NB][RT][256] buffer
    size_t off_stages, off_flags, off_status, off_xin0, off_xs, off_xo, off_att, off_x1, off_x2, off_pc, off_pe, total;
    int nwg, NB, P;
};
SysLayout sys_layout(int B, int T) {
    SysLayout L;
    int P = RT / (2 * T);
    if (P > 7) P = 7;             // the QKV stage parks (2P + 1) x 128 floats of extra K|V through 512 thread slots
    if (P < 1) P = 1;
    L.P = P;
    L.NB = (B + P - 1) / P;
    L.nwg = NL * (4 + 1 + NSLICE + NRED + NSLICE + NRED) + 2 * NSKIP + 1;
    L.blk = (size_t)L.NB * RT * D;
    size_t off = 0;
    auto take = [&](size_t floats) { const size_t o = off; off += (floats + 63) / 64 * 64; return o; };
    L.off_stages = take((size_t)256 * sizeof(Stage) / sizeof(float));
    L.off_flags = take((size_t)NL * GROUPS_PER_LAYER * L.NB * FLAG_SLOTS);
    L.off_status = take(64);
    L.off_xin0 = take(L.blk);
    L.off_xs = take(NSKIP * L.blk);
    L.off_xo = take(NL * L.blk);
    L.off_att = take(L.blk);
    L.off_x1 = take(L.blk);
    L.off_x2 = take(L.blk);
    L.off_pc = take(NSLICE * L.blk);
    L.off_pe = take(NSLICE * L.blk);
    L.total = off;
    return L;
}
}  // namespace

size_t sys_ws_floats(int B, int T) { return sys_layout(B, T).total; }

bool sys_supported(int B, int T, int cfg, bool split) {
    if (!cfg || !split || B < 1 || T < 1 || T > LADIFF_MAX_LATENTS) return false;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    return cus >= sys_layout(B, T).nwg;          // every stage needs a CU of its own, all resident at once
}

// Builds the stage table (host) for this call's pointers.  `ws` = the systolic region of the reverse workspace.
int sys_build_stages(const DenoiserW& W, const DenoiserW& WS, float* ws, int B, int T, std::vector<unsigned char>& host) {
    const SysLayout L = sys_layout(B, T);
    std::vector<Stage> st;
    float* xin0 = ws + L.off_xin0;
    float* att = ws + L.off_att; float* x1 = ws + L.off_x1; float* x2 = ws + L.off_x2;
    float* pc = ws + L.off_pc; float* pe = ws + L.off_pe;
    auto XO = [&](int l) { return ws + L.off_xo + (size_t)l * L.blk; };
    auto XS = [&](int l) { return ws + L.off_xs + (size_t)(l - NSKIP - 1) * L.blk; };
    auto G = [&](int l, int g) { return l * GROUPS_PER_LAYER + g; };
    for (int l = 0; l < NL; ++l) {
        const DenLayerW& w = W.layer[l];
        const DenLayerW& ws_ = WS.layer[l];
        const float* xin; int xg, xn;
        if (l == 0) { xin = xin0; xg = G(0, G_XIN); xn = 1; }
        else if (l <= NSKIP) { xin = XO(l - 1); xg = G(l - 1, G_XO); xn = NRED; }
        else { xin = XS(l); xg = G(l, G_XIN); xn = 2; }
        if (l > NSKIP) {
            const int i = l - NSKIP - 1;
            for (int c = 0; c < 2; ++c) {
                Stage s{};
                s.role = R_SKIP; s.layer = l; s.slice = c; s.wait_group = G(l - 1, G_XO); s.wait_n = NRED;
                s.out_group = G(l, G_XIN); s.out_slot = c;
                s.w0 = WS.skip[i].w; s.b0 = W.skip[i].b; s.in0 = XO(l - 1); s.in1 = XO(NL - 1 - l); s.out = XS(l);
                st.push_back(s);
            }
        }
        for (int h = 0; h < H; ++h) {
            Stage s{};
            s.role = R_QKV; s.layer = l; s.slice = h; s.wait_group = xg; s.wait_n = xn; s.out_group = G(l, G_ATT); s.out_slot = h;
            s.w0 = ws_.sa_attn.in_w; s.b0 = w.sa_attn.in_b; s.in0 = xin; s.out = att;
            st.push_back(s);
        }
        {
            Stage s{};
            s.role = R_OUT; s.layer = l; s.wait_group = G(l, G_ATT); s.wait_n = H; s.out_group = G(l, G_X1); s.out_slot = 0;
            s.w0 = ws_.sa_attn.out_w; s.b0 = w.sa_attn.out_b; s.g = w.sa_norm1.g; s.be = w.sa_norm1.b; s.in0 = att; s.in1 = xin; s.out = x1;
            st.push_back(s);
        }
        for (int j = 0; j < NSLICE; ++j) {
            Stage s{};
            s.role = R_LIN; s.layer = l; s.slice = j; s.wait_group = G(l, G_X1); s.wait_n = 1; s.out_group = G(l, G_PC); s.out_slot = j;
            s.w0 = ws_.sa_lin1.w; s.w1 = ws_.sa_lin2.w; s.b0 = w.sa_lin1.b; s.in0 = x1; s.out = pc;
            st.push_back(s);
        }
        for (int q = 0; q < NRED; ++q) {
            Stage s{};
            s.role = R_RED2; s.layer = l; s.slice = q; s.wait_group = G(l, G_PC); s.wait_n = NSLICE; s.out_group = G(l, G_X2); s.out_slot = q;
            s.b0 = w.sa_lin2.b; s.g = w.sa_norm2.g; s.be = w.sa_norm2.b; s.in0 = pc; s.in1 = x1; s.out = x2;
            st.push_back(s);
        }
        for (int j = 0; j < NSLICE; ++j) {
            Stage s{};
            s.role = R_FFN; s.layer = l; s.slice = j; s.wait_group = G(l, G_X2); s.wait_n = NRED; s.out_group = G(l, G_PE); s.out_slot = j;
            s.w0 = ws_.ffn1.w; s.w1 = ws_.ffn2.w; s.b0 = w.ffn1.b; s.in0 = x2; s.out = pe;
            st.push_back(s);
        }
        for (int q = 0; q < NRED; ++q) {
            Stage s{};
            s.role = R_STYL; s.layer = l; s.slice = q; s.wait_group = G(l, G_PE); s.wait_n = NSLICE; s.out_group = G(l, G_XO); s.out_slot = q;
            s.w0 = ws_.ffn_proj.out.w; s.b0 = w.ffn_proj.out.b; s.b1 = w.ffn2.b; s.g = w.ffn_proj.norm.g; s.be = w.ffn_proj.norm.b;
            s.in0 = pe; s.in1 = x2; s.out = XO(l);
            st.push_back(s);
        }
    }
    {
        Stage s{};
        s.role = R_TAIL; s.layer = NL; s.wait_group = G(NL - 1, G_XO); s.wait_n = NRED; s.out_group = G(0, G_XIN); s.out_slot = 0;
        s.in0 = XO(NL - 1); s.out = xin0;
        st.push_back(s);
    }
    if ((int)st.size() != L.nwg || st.size() > 256) return LADIFF_ERR_SHAPE;
    host.resize(st.size() * sizeof(Stage));
    std::memcpy(host.data(), st.data(), host.size());
    return 0;
}

// One launch = local steps [step_lo, step_lo + n) of the loop on the latents in `lat`.  The stage table must already be in
// the workspace (sys_upload_stages); `ctab` holds the hoisted cross-attention rows of n_ctab >= n steps starting at step_lo.
int launch_systolic_loop(const DenoiserW& W, float* ws, const float* tables, const float* tkv, const float* ctab, int n_ctab,
                         const float* coef, const float* noise, float* lat, const int32_t* counts, float gscale, int B, int T,
                         int step_lo, int n, hipStream_t s) {
    const SysLayout L = sys_layout(B, T);
    static bool attr_set = false;
    if (!attr_set) {
        LADIFF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(systolic_loop_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       SYS_LDS_BYTES));
        attr_set = true;
    }
    SysArgs a;
    a.stages = reinterpret_cast<const Stage*>(ws + L.off_stages);
    a.flags = reinterpret_cast<unsigned*>(ws + L.off_flags);
    a.status = reinterpret_cast<unsigned*>(ws + L.off_status);
    a.tables = tables; a.tkv = tkv; a.ctab = ctab; a.coef = coef; a.noise = noise; a.pe = W.query_pe; a.ng = W.norm.g; a.nb = W.norm.b;
    a.lat = lat; a.counts = counts; a.gscale = gscale; a.B = B; a.T = T; a.P = L.P; a.NB = L.NB; a.step_lo = step_lo; a.n_steps = n;
    a.n_ctab = n_ctab;
    // flags and the abort word are contiguous: one memset node, a multiple of 16 bytes
    LADIFF_HIP(hipMemsetAsync(a.flags, 0, (L.off_xin0 - L.off_flags) * sizeof(float), s));
    hipLaunchKernelGGL(systolic_loop_kernel, dim3(L.nwg), dim3(256), SYS_LDS_BYTES, s, a);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

size_t sys_stage_bytes(int B, int T) { return (size_t)sys_layout(B, T).nwg * sizeof(Stage); }
size_t sys_status_offset_floats(int B, int T) { return sys_layout(B, T).off_status; }

}  // namespace ladiff
