// Attention cores of the hot path.
//   * decoder self-attention over F <= 224 frames (fp32 MFMA, K/V of one (sample, head) resident in LDS)
//   * decoder cross-attention to the <= 8 latent tokens (VALU, one wave per frame row)
//   * denoiser self-attention over the T + 2 <= 10 tokens [latents | text | time] (VALU, one wave per head)
#include "kernels.h"

namespace ladiff {

// ===================================================================== decoder self-attention
// nn.MultiheadAttention core of TransformerDecoderLayer.forward_post (cross_attention.py:367-369):
//   P = softmax(Q K^T / 8 + mask(key >= len_b)),  O = P V,  per sample b and head h, dh = 64.
// One workgroup = (b, h, group of 4 query tiles); wave w owns query tile qt = 4*blockIdx.x + w (32 queries).
//   S^T tile [32 keys x 32 queries] = K_tile [32 x 64] . Q_tile^T   (mfma 32x32x2 f32, A = K from LDS, B = Q in regs)
// so a query's scores sit in ONE lane pair (lane, lane^32): the row softmax needs a single cross-lane op, and
// the exponentiated tile is directly the B operand of  O^T [64 d x 32 queries] += V_tile^T . P_tile^T.
constexpr int SA_FMAX = LADIFF_MAX_FRAMES;   // 224 = 7 key tiles
constexpr int SA_NKT = SA_FMAX / 32;

// Key validity is either a prefix (keys < lengths[b]) or, when `keybits` is given, an arbitrary 256-bit map per sample
// (the LA-VAE encoder masks latent tokens in the middle of the sequence, ladiff_vae.py:193-209).
// `g` describes where the heads live: row stride of the packed qkv, column offsets of the K and V blocks, number of
// heads, row stride of the output; g.causal adds key <= query (CLIP's text transformer).
struct AttnGeom { int nheads, ld, koff, voff, out_ld, causal; };

__global__ __launch_bounds__(256) void dec_self_attn_kernel(const float* __restrict__ qkv, const int32_t* __restrict__ lengths,
                                                            const uint32_t* __restrict__ keybits, float* __restrict__ out,
                                                            int B, int F, int split_out, const AttnGeom g,
                                                            const int32_t* __restrict__ row_off, int shared_qkv) {
    __shared__ __attribute__((aligned(16))) float Ks[SA_FMAX * DH];   // chunk c of row r at slot c ^ (r & 15)
    __shared__ __attribute__((aligned(16))) float Vs[SA_FMAX * DH];   // plain [key][d]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y / g.nheads, h = blockIdx.y % g.nheads;
    int len = F;                                  // keys >= len are never valid
    uint32_t kb[SA_NKT + 1];
    if (keybits != nullptr) {
        len = 1;
#pragma unroll
        for (int i = 0; i < SA_NKT; ++i) {
            kb[i] = keybits[(size_t)b * 8 + i];
            if (kb[i]) len = 32 * i + 32 - __builtin_clz(kb[i]);
        }
        len = len > F ? F : len;
    } else {
        len = lengths != nullptr ? lengths[b] : F;
        len = len < 1 ? 1 : (len > F ? F : len);
#pragma unroll
        for (int i = 0; i < SA_NKT; ++i) kb[i] = len >= 32 * i + 32 ? 0xFFFFFFFFu : (len > 32 * i ? (1u << (len - 32 * i)) - 1u : 0u);
    }
    const int nkt = (len + 31) >> 5;
    size_t row0 = (size_t)b * F;
    if (row_off != nullptr) { row0 = row_off[b]; F = row_off[b + 1] - row_off[b]; }     // ragged rows: the sample's own F = its length
    // shared_qkv: every sample reads the SAME [F] rows of q|k|v (decoder layer 0: its input is the position table, ladiff_vae.py:299)
    const size_t base = (shared_qkv ? (size_t)0 : row0) * g.ld + h * DH;

    for (int id = tid; id < nkt * 32 * 16; id += 256) {
        const int r = id >> 4, c = id & 15;
        f32x4 kk = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
        if (r < F) {
            const float* src = qkv + base + (size_t)r * g.ld + c * 4;
            kk = ld4(src + g.koff);
            vv = ld4(src + g.voff);
        }
        st4(Ks + r * DH + ((c ^ (r & 15)) << 2), kk);
        st4(Vs + r * DH + c * 4, vv);
    }
    __syncthreads();

    const int qt = blockIdx.x * 4 + wave;
    if (qt * 32 >= F) return;
    const int q = lane & 31, h2 = lane >> 5;
    const int qrow = qt * 32 + q;

    f32x4 qf[8];
#pragma unroll
    for (int g8 = 0; g8 < 8; ++g8) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (qrow < F) v = ld4(qkv + base + (size_t)qrow * g.ld + g8 * 8 + h2 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= 0.125f;   // q / sqrt(64), exact
        qf[g8] = v;
    }

    f32x16 sT[SA_NKT];
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < SA_NKT; ++kt) {
        if (kt < nkt) {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            const int r = kt * 32 + q;
#pragma unroll
            for (int g8 = 0; g8 < 8; ++g8) {
                const f32x4 a = ld4(Ks + r * DH + ((((g8 << 1) + h2) ^ (r & 15)) << 2));
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], qf[g8][e], acc, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kin = (i & 3) + 8 * (i >> 2) + 4 * h2;          // key within the tile
                const bool ok = ((kb[kt] >> kin) & 1u) && (!g.causal || kt * 32 + kin <= qrow);
                const float s = ok ? acc[i] : -INFINITY;
                acc[i] = s;
                m = fmaxf(m, s);
            }
            sT[kt] = acc;
        }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));

    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < SA_NKT; ++kt) {
        if (kt < nkt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pv = expf(sT[kt][i] - m);
                sT[kt][i] = pv;
                l += pv;
            }
        }
    }
    l += __shfl_xor(l, 32, 64);

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
#pragma unroll
    for (int kt = 0; kt < SA_NKT; ++kt) {
        if (kt < nkt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h2;
                const float a0 = Vs[key * DH + q];
                const float a1 = Vs[key * DH + 32 + q];
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, sT[kt][i], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, sT[kt][i], o1, 0, 0, 0);
            }
        }
    }

    if (qrow < F) {
        const float inv = 1.f / l;
        float* rowp = out + (row0 + qrow) * g.out_ld;
        const int c0 = h * DH + 4 * h2;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            f32x4 v0, v1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] = o0[4 * rg + e] * inv; v1[e] = o1[4 * rg + e] * inv; }
            if (split_out) {
                store_split4(rowp, c0 + 8 * rg, v0);
                store_split4(rowp, c0 + 32 + 8 * rg, v1);
            } else {
                st4(rowp + c0 + 8 * rg, v0);
                st4(rowp + c0 + 32 + 8 * rg, v1);
            }
        }
    }
}

// ---- f16x3 version (precision mode "f16x3"): the same S^T / O^T formulation on v_mfma_f32_32x32x16_bf16 with every
// product evaluated as lo*hi + hi*lo + hi*hi of bf16 pairs (q, k, v and the probabilities are split; fp32 accumulate,
// fp32 softmax).  The fp32 MFMA (32x32x2, 64 cycles) made the fp32 kernel matrix-pipe-bound at 28.7 k cycles per query
// tile; here the two products take 5.4 k.  One workgroup per (sample, head), 8 waves = up to 7 query tiles, so K / V are
// staged once.  LDS: K as [key][64 d] bf16 hi / lo planes (16-byte chunk c of row r at c ^ ((r >> 1) & 7)), V transposed
// to [d][key] hi / lo planes (row stride 232 keys) so that the k = key operand of O^T += V^T . P^T is two 8-byte reads:
// the accumulator tile of S^T has its query on the lane and its keys in the 16 registers, so registers 8m .. 8m+7 are,
// unmoved, the B operand of MFMA m of a key tile (keys 16m + 4g + {0..3, 8..11}); V^T is read with the same key map.

__device__ __forceinline__ void split8(const float (&v)[8], s16x8& hi, s16x8& lo) {
    split8(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]}, hi, lo);
}

// FM = the most keys a sample can have (a multiple of 32), NT = threads: <224, 512> is the decoder / encoder form (a wave per query tile of 32,
// 116 KB of LDS: one workgroup per CU); <32, 64> serves sequences of up to 32 rows with ONE wave and 18 KB (round 6: the CLIP tower's ragged
// rows are <= 31 per prompt - 1,548 (prompt, head) workgroups of the big form ran six deep on 256 CUs with seven of eight waves idle:
// 36 us per layer, profiles/r6/13_*).
template <int FM, int NT>
__global__ __launch_bounds__(NT, NT == 64 ? 2 : 1) void self_attn_split_kernel(const float* __restrict__ qkv, const int32_t* __restrict__ lengths,
                                                               const uint32_t* __restrict__ keybits, float* __restrict__ out,
                                                               int B, int F, int split_out, const AttnGeom g,
                                                               const int32_t* __restrict__ row_off, int shared_qkv) {
    constexpr int NKT = FM / 32, IT = FM * 16 / NT, VLD = FM + 8;        // key tiles; 16-byte staging units per thread; keys per row of the transposed V planes (a multiple of 8)
    static_assert(FM % 32 == 0 && (FM * 16) % NT == 0 && NT % 64 == 0 && (FM / 4) % (NT / 64) == 0 && NT / 64 >= FM / 32, "a wave per query tile");
    __shared__ __attribute__((aligned(16))) s16 Kp[2 * FM * DH];      // hi plane, lo plane; first the fp32 staging of V
    __shared__ __attribute__((aligned(16))) s16 Vt[2 * DH * VLD];       // hi plane, lo plane, [d][key]
    s16* const Kh = Kp; s16* const Kl = Kp + FM * DH;
    s16* const Vth = Vt; s16* const Vtl = Vt + DH * VLD;
    float* const Vtmp = reinterpret_cast<float*>(Kp);                         // [key][64] fp32 = exactly the two K planes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / g.nheads, h = blockIdx.x % g.nheads;
    int len = F;
    uint32_t kb[NKT + 1];
    if (keybits != nullptr) {
        len = 1;
#pragma unroll
        for (int i = 0; i < NKT; ++i) {
            kb[i] = keybits[(size_t)b * 8 + i];
            if (kb[i]) len = 32 * i + 32 - __builtin_clz(kb[i]);
        }
        len = len > F ? F : len;
    } else {
        len = lengths != nullptr ? lengths[b] : F;
        len = len < 1 ? 1 : (len > F ? F : len);
#pragma unroll
        for (int i = 0; i < NKT; ++i) kb[i] = len >= 32 * i + 32 ? 0xFFFFFFFFu : (len > 32 * i ? (1u << (len - 32 * i)) - 1u : 0u);
    }
    const int nkt = (len + 31) >> 5;
    size_t row0 = (size_t)b * F;
    if (row_off != nullptr) { row0 = row_off[b]; F = row_off[b + 1] - row_off[b]; }     // ragged rows: the sample's own F = its length
    // shared_qkv: every sample reads the SAME [F] rows of q|k|v (decoder layer 0: its input is the position table, ladiff_vae.py:299)
    const size_t base = (shared_qkv ? (size_t)0 : row0) * g.ld + h * DH;

    const int qt = wave;                      // query tile of this wave; its q rows are fetched together with K / V
    const int q = lane & 31, h2 = lane >> 5;
    const int qrow = qt * 32 + q;

    s16x8 qh[4], ql[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (qrow < F) {
            const f32x4 a = ld4(qkv + base + (size_t)qrow * g.ld + 16 * ks + 8 * h2);
            const f32x4 c = ld4(qkv + base + (size_t)qrow * g.ld + 16 * ks + 8 * h2 + 4);
#pragma unroll
            // q / sqrt(64) * log2(e): the scores come out in base-2 units, so the softmax numerators are one v_exp_f32 each
            for (int e = 0; e < 4; ++e) { v[e] = a[e] * (0.125f * 1.4426950408889634f); v[4 + e] = c[e] * (0.125f * 1.4426950408889634f); }
        }
        split8(v, qh[ks], ql[ks]);
    }


    // ---- staging.  All global loads first (7 x (k, v) float4 per thread), V through an fp32 image for the transpose.
    f32x4 kk[IT], vv[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int id = tid + it * NT, r = id >> 4, c = id & 15;
        kk[it] = f32x4{0.f, 0.f, 0.f, 0.f}; vv[it] = kk[it];
        if (r < nkt * 32 && r < F) {
            const float* src = qkv + base + (size_t)r * g.ld + c * 4;
            kk[it] = ld4(src + g.koff);
            vv[it] = ld4(src + g.voff);
        }
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int id = tid + it * NT, r = id >> 4, c = id & 15;
        if (r < nkt * 32) st4(Vtmp + r * DH + c * 4, vv[it]);
    }
    __syncthreads();
    {   // transpose + split: thread (d, key group) turns 4 keys of column d into 8 bytes of each plane
        const int d = tid & 63, kg = tid >> 6;
        for (int p = 0; p < FM / 4 / (NT / 64); ++p) {
            const int k0 = (p * (NT / 64) + kg) * 4;
            if (k0 < nkt * 32) {
                s16x4 hi, lo;
                split4(Vtmp[k0 * DH + d], Vtmp[(k0 + 1) * DH + d], Vtmp[(k0 + 2) * DH + d], Vtmp[(k0 + 3) * DH + d], hi, lo);
                *reinterpret_cast<s16x4*>(Vth + d * VLD + k0) = hi;
                *reinterpret_cast<s16x4*>(Vtl + d * VLD + k0) = lo;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int id = tid + it * NT, r = id >> 4, c = id & 15;
        if (r < nkt * 32) {
            s16x4 hi, lo;
            split4(kk[it], hi, lo);
            const int off = r * DH + ((((c >> 1) ^ ((r >> 1) & 7)) << 3) | ((c & 1) << 2));   // bf16 elements
            *reinterpret_cast<s16x4*>(Kh + off) = hi;
            *reinterpret_cast<s16x4*>(Kl + off) = lo;
        }
    }
    __syncthreads();

    if (qt * 32 >= F) return;

    f32x16 sT[NKT];
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        if (kt < nkt) {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            const int r = kt * 32 + q;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off = r * DH + (((2 * ks + h2) ^ ((r >> 1) & 7)) << 3);
                const s16x8 kh = *reinterpret_cast<const s16x8*>(Kh + off);
                const s16x8 kl = *reinterpret_cast<const s16x8*>(Kl + off);
                acc = MFMA32_S16(kl, qh[ks], acc, 0, 0, 0);
                acc = MFMA32_S16(kh, ql[ks], acc, 0, 0, 0);
                acc = MFMA32_S16(kh, qh[ks], acc, 0, 0, 0);
            }
            if (kb[kt] == 0xFFFFFFFFu && !g.causal) {             // every key of the tile is valid (wave-uniform): no masking
#pragma unroll
                for (int i = 0; i < 16; ++i) m = fmaxf(m, acc[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int kin = (i & 3) + 8 * (i >> 2) + 4 * h2;
                    const bool ok = ((kb[kt] >> kin) & 1u) && (!g.causal || kt * 32 + kin <= qrow);
                    const float s = ok ? acc[i] : -INFINITY;
                    acc[i] = s;
                    m = fmaxf(m, s);
                }
            }
            sT[kt] = acc;
        }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));

    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        if (kt < nkt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pv = __builtin_amdgcn_exp2f(sT[kt][i] - m);
                sT[kt][i] = pv;
                l += pv;
            }
        }
    }
    l += __shfl_xor(l, 32, 64);

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        if (kt < nkt) {
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                float pv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) pv[j] = sT[kt][8 * mm + j];
                s16x8 ph, pl;
                split8(pv, ph, pl);
                const int k1 = 32 * kt + 16 * mm + 4 * h2;            // keys k1 .. k1+3 and k1+8 .. k1+11
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int voff = (32 * t + q) * VLD + k1;
                    const s16x4 h_a = *reinterpret_cast<const s16x4*>(Vth + voff), h_b = *reinterpret_cast<const s16x4*>(Vth + voff + 8);
                    const s16x4 l_a = *reinterpret_cast<const s16x4*>(Vtl + voff), l_b = *reinterpret_cast<const s16x4*>(Vtl + voff + 8);
                    const s16x8 vh = {h_a[0], h_a[1], h_a[2], h_a[3], h_b[0], h_b[1], h_b[2], h_b[3]};
                    const s16x8 vl = {l_a[0], l_a[1], l_a[2], l_a[3], l_b[0], l_b[1], l_b[2], l_b[3]};
                    f32x16& o = t == 0 ? o0 : o1;
                    o = MFMA32_S16(vl, ph, o, 0, 0, 0);
                    o = MFMA32_S16(vh, pl, o, 0, 0, 0);
                    o = MFMA32_S16(vh, ph, o, 0, 0, 0);
                }
            }
        }
    }

    if (qrow < F) {
        const float inv = 1.f / l;
        float* rowp = out + (row0 + qrow) * g.out_ld;
        const int c0 = h * DH + 4 * h2;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            f32x4 v0, v1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] = o0[4 * rg + e] * inv; v1[e] = o1[4 * rg + e] * inv; }
            if (split_out) {
                store_split4(rowp, c0 + 8 * rg, v0);
                store_split4(rowp, c0 + 32 + 8 * rg, v1);
            } else {
                st4(rowp + c0 + 8 * rg, v0);
                st4(rowp + c0 + 32 + 8 * rg, v1);
            }
        }
    }
}

// f16x3 entry (same arguments as launch_self_attention)
int launch_self_attention_split(const float* qkv, const int32_t* lengths, const uint32_t* keybits, float* out, int B, int F,
                                 int nheads, int causal, int split_out, hipStream_t s, const int32_t* row_off, int shared_qkv) {
    if (F > SA_FMAX || F < 1 || nheads < 1) return LADIFF_ERR_SHAPE;
    if (B == 0) return 0;
    const int W = nheads * DH;
    const AttnGeom g{nheads, 3 * W, W, 2 * W, W, causal};
    if (F <= 32) hipLaunchKernelGGL((self_attn_split_kernel<32, 64>), dim3(B * nheads), dim3(64), 0, s, qkv, lengths, keybits, out, B, F, split_out, g, row_off, shared_qkv);
    else hipLaunchKernelGGL((self_attn_split_kernel<SA_FMAX, 512>), dim3(B * nheads), dim3(512), 0, s, qkv, lengths, keybits, out, B, F, split_out, g, row_off, shared_qkv);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// generic entry: `nheads` heads of 64, packed rows [q | k | v] of width 3 * 64 * nheads, optional causal mask
int launch_self_attention(const float* qkv, const int32_t* lengths, const uint32_t* keybits, float* out, int B, int F, int nheads,
                          int causal, int split_out, hipStream_t s, const int32_t* row_off) {
    if (F > SA_FMAX || F < 1 || nheads < 1) return LADIFF_ERR_SHAPE;
    if (B == 0) return 0;
    const int W = nheads * DH;
    const AttnGeom g{nheads, 3 * W, W, 2 * W, W, causal};
    const int nqt = (F + 31) / 32;
    hipLaunchKernelGGL(dec_self_attn_kernel, dim3((nqt + 3) / 4, B * nheads), dim3(256), 0, s, qkv, lengths, keybits, out, B, F,
                       split_out, g, row_off, 0);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

int launch_decoder_self_attention(const float* qkv, const int32_t* lengths, const uint32_t* keybits, float* out, int B, int F,
                                  int split_out, hipStream_t s, const int32_t* row_off, int shared_qkv) {
    if (F > SA_FMAX || F < 1) return LADIFF_ERR_SHAPE;
    if (lengths == nullptr && keybits == nullptr) return LADIFF_ERR_ARG;
    if (B == 0) return 0;
    const AttnGeom g{H, 3 * D, D, 2 * D, D, 0};
    const int nqt = (F + 31) / 32;
    hipLaunchKernelGGL(dec_self_attn_kernel, dim3((nqt + 3) / 4, B * H), dim3(256), 0, s, qkv, lengths, keybits, out, B, F,
                       split_out, g, row_off, shared_qkv);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// ===================================================================== decoder cross-attention
// multihead_attn core (cross_attention.py:373-376): frame query against the T latent tokens of its sample,
// tokens >= counts[b] masked.  One wave per frame row; lane owns 4 of the 256 columns (16 lanes per head).
template <int T>
__global__ __launch_bounds__(256) void dec_cross_attn_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                             const int32_t* __restrict__ counts, float* __restrict__ out,
                                                             int B, int F, int M, int split_out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int c = (threadIdx.x & 63) * 4;
    const int b = row / F;
    int nv = counts ? counts[b] : T;
    nv = nv < 1 ? 1 : (nv > T ? T : nv);
    f32x4 qv = ld4(q + (size_t)row * D + c);
    float s[T];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < T; ++j) {
        const f32x4 kk = ld4(kv + ((size_t)j * B + b) * 512 + c);
        float d = (qv[0] * 0.125f) * kk[0] + (qv[1] * 0.125f) * kk[1] + (qv[2] * 0.125f) * kk[2] + (qv[3] * 0.125f) * kk[3];
        d = row16_sum(d);
        s[j] = j < nv ? d : -INFINITY;
        m = fmaxf(m, s[j]);
    }
    float l = 0.f;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < T; ++j) {
        const float pv = expf(s[j] - m);
        l += pv;
        const f32x4 vv = ld4(kv + ((size_t)j * B + b) * 512 + 256 + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += pv * vv[e];
    }
    const float inv = 1.f / l;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] *= inv;
    if (split_out) store_split4(out + (size_t)row * D, c, o);
    else st4(out + (size_t)row * D + c, o);
}

int launch_decoder_cross_attention(const float* q, const float* kv, const int32_t* counts, float* out, int B, int F,
                                   int T, int split_out, hipStream_t s) {
    const int M = B * F;
    if (M == 0) return 0;
    const dim3 grid((M + 3) / 4), block(256);
#define LADIFF_CA_CASE(TT) \
    case TT: hipLaunchKernelGGL(dec_cross_attn_kernel<TT>, grid, block, 0, s, q, kv, counts, out, B, F, M, split_out); break;
    switch (T) {
        LADIFF_CA_CASE(1) LADIFF_CA_CASE(2) LADIFF_CA_CASE(3) LADIFF_CA_CASE(4)
        LADIFF_CA_CASE(5) LADIFF_CA_CASE(6) LADIFF_CA_CASE(7) LADIFF_CA_CASE(8)
        default: return LADIFF_ERR_SHAPE;
    }
#undef LADIFF_CA_CASE
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// ===================================================================== denoiser self-attention
// sa_block attention of LinearTemporalDiffusionTransformerDecoderLayer (mdiff_transformer.py:296-313): queries
// are the T latent rows; keys/values are [T latent rows | text token | time token]; latent keys >= counts[b]
// are masked, the two extra tokens never are.  The text token's K|V are step-invariant (text cache) and the
// time token's K|V depend on (step, layer) only (time table).  One workgroup per sample, wave = head, lane = d.
template <int T>
__global__ __launch_bounds__(256) void den_self_attn_kernel(const float* __restrict__ qkv, const float* __restrict__ text_kv,
                                                            const float* __restrict__ tables, int kv_off, int step_stride,
                                                            const int32_t* __restrict__ d_step,
                                                            const int32_t* __restrict__ counts, int Bs, int b_off,
                                                            float* __restrict__ out, int split_out) {
    const int b2 = blockIdx.x;             // local sample: rows of qkv / out
    const int bg = b_off + b2;             // sample of the whole (duplicated) batch: text cache row, counts
    const int col = threadIdx.x;   // = head * 64 + d
    float qv[T], kk[T + 2], vv[T + 2];
#pragma unroll
    for (int i = 0; i < T; ++i) {
        const float* r = qkv + ((size_t)b2 * T + i) * 768 + col;
        qv[i] = r[0] * 0.125f;
        kk[i] = r[256];
        vv[i] = r[512];
    }
    kk[T] = text_kv[(size_t)bg * 512 + col];
    vv[T] = text_kv[(size_t)bg * 512 + 256 + col];
    const float* tk = tables + (size_t)(*d_step) * step_stride + kv_off;
    kk[T + 1] = tk[col];
    vv[T + 1] = tk[256 + col];
    int nv = counts ? counts[bg % Bs] : T;
    nv = nv > T ? T : nv;
    __shared__ __attribute__((aligned(16))) float so[T * D];      // staged outputs for the S-format store
#pragma unroll
    for (int i = 0; i < T; ++i) {
        float s[T + 2];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < T + 2; ++j) {
            const float d = wave_sum(qv[i] * kk[j]);
            s[j] = (j < T && j >= nv) ? -INFINITY : d;
            m = fmaxf(m, s[j]);
        }
        float l = 0.f, o = 0.f;
#pragma unroll
        for (int j = 0; j < T + 2; ++j) {
            const float pv = expf(s[j] - m);
            l += pv;
            o += pv * vv[j];
        }
        if (split_out) so[i * D + col] = o / l;
        else out[((size_t)b2 * T + i) * D + col] = o / l;
    }
    if (split_out) {      // 16-byte units: 4 consecutive columns -> 8 B of hi + 8 B of lo
        __syncthreads();
        for (int u = threadIdx.x; u < T * (D / 4); u += 256) {
            const int r = u / (D / 4), c4 = (u % (D / 4)) * 4;
            store_split4(out + ((size_t)b2 * T + r) * D, c4, ld4(so + r * D + c4));
        }
    }
}

int launch_denoiser_self_attention(const float* qkv, const float* text_kv, const float* tables, int kv_off,
                                   int step_stride, const int32_t* d_step, const int32_t* counts, int Bs, int b_off,
                                   int b_n, int T, float* out, int split_out, hipStream_t s) {
    if (b_n == 0) return 0;
    const dim3 grid(b_n), block(256);
#define LADIFF_SA_CASE(TT)                                                                                          \
    case TT:                                                                                                        \
        hipLaunchKernelGGL(den_self_attn_kernel<TT>, grid, block, 0, s, qkv, text_kv, tables, kv_off, step_stride, \
                           d_step, counts, Bs, b_off, out, split_out);                                                                \
        break;
    switch (T) {
        LADIFF_SA_CASE(1) LADIFF_SA_CASE(2) LADIFF_SA_CASE(3) LADIFF_SA_CASE(4)
        LADIFF_SA_CASE(5) LADIFF_SA_CASE(6) LADIFF_SA_CASE(7) LADIFF_SA_CASE(8)
        default: return LADIFF_ERR_SHAPE;
    }
#undef LADIFF_SA_CASE
    LADIFF_LAUNCH_CHECK();
    return 0;
}

}  // namespace ladiff
