// Decoder self-attention with its in_proj inside (bf16x3 mode): one workgroup per (sample, head) computes that head's q | k | v rows
// from the S-format x rows and attends, instead of an in_proj GEMM that writes [M, 768] fp32 (77 MB at M = 25088) and an attention
// kernel that reads it back and re-splits it (cross_attention.py:367-369; nn.MultiheadAttention's packed in_proj).
//
//   * wave w owns frame rows 32 w .. 32 w + 31 both as queries and as keys / values.  Its x rows are MFMA operand fragments in
//     registers (16 k-steps x (hi, lo) bf16x8 per lane, loaded straight from the S-format rows).
//   * the head's 192 in_proj rows (64 q, 64 k, 64 v) stream through LDS as twelve 16-KiB stages (32 weight rows x 128 k, hi and lo
//     planes), double-buffered by LDS-DMA one stage ahead; a stage's fragment = one conflict-free ds_read_b128 per lane.
//   * q and k are computed TRANSPOSED (D^T = W x^T, v_mfma_f32_32x32x16_bf16: weight fragment first), v plainly (x first): the
//     accumulator of a q^T / k^T tile then has the row's query / key on the lane and 16 of its d values in the registers, and since
//     d is the contraction index of q.k ANY order of d is fine as long as q and k share it: registers 8 j .. 8 j + 7 of tile T ARE the
//     q operand fragment of k-step 2 T + j of the score product (no cross-lane move), and the same registers of a k^T tile are one
//     16-byte write into chunk 2 (2 T + j) + h2 of the K image.  A v tile has its d on the lane and four consecutive keys per
//     register quad: an 8-byte write into the transposed V image the output product reads.
//   * then the score / softmax / output core of self_attn_bf16x3_kernel (attention.hip), unchanged: S^T = K Q^T, fp32 softmax in
//     base 2, O^T = V^T P^T, every product as lo*hi + hi*lo + hi*hi.
// Arithmetic: the products of the separate kernels (same splits, fp32 accumulate); q.k sums its 64 d in another order.
#include "model.h"
#include "tile_mma.h"

namespace ladiff {

namespace {

constexpr int QA_FMAX = LADIFF_MAX_FRAMES;         // 224 = 7 tiles of 32 rows
constexpr int QA_NKT = QA_FMAX / 32;
constexpr int QA_VLD = 232;                        // keys per row of the transposed V planes
constexpr int QA_STAGE = 16384;                    // 8 k-steps x (hi, lo) x 32 rows x 32 B
constexpr int QA_NSTAGE = 12;                      // (q0 q1 k0 k1 v0 v1) x two k halves
constexpr int QA_K_BYTES = 2 * QA_FMAX * DH * 2;
constexpr int QA_V_BYTES = 2 * DH * QA_VLD * 2;
constexpr int QA_LDS = QA_K_BYTES + QA_V_BYTES + 2 * QA_STAGE + 3 * DH * 4;

struct QkvAttnArgs {
    const float* xs;           // [M, 256] S-format rows
    const float* w;            // in_proj_weight [768, 256] S-format
    const float* bias;         // in_proj_bias [768]
    const int32_t* lengths;    // [B]
    const int32_t* row_off;    // ragged rows: sample b owns rows row_off[b] .. row_off[b + 1] (NULL: b F ..)
    float* out;                // [M, 256] fp32 or S-format
    int B, F, split_out;
};

__device__ __forceinline__ void split8v(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { hi[e] = (__bf16)v[e]; lo[e] = (__bf16)(v[e] - (float)hi[e]); }
}

}  // namespace

__global__ __launch_bounds__(512) void dec_qkv_attn_kernel(const QkvAttnArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    __bf16* const Kh = reinterpret_cast<__bf16*>(lds);
    __bf16* const Kl = Kh + QA_FMAX * DH;
    __bf16* const Vth = reinterpret_cast<__bf16*>(lds + QA_K_BYTES);
    __bf16* const Vtl = Vth + DH * QA_VLD;
    char* const ring = lds + QA_K_BYTES + QA_V_BYTES;
    float* const bsm = reinterpret_cast<float*>(ring + 2 * QA_STAGE);          // [q 64 | k 64 | v 64] of this head
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    int F = p.F;
    size_t row0 = (size_t)b * F;
    if (p.row_off != nullptr) { row0 = p.row_off[b]; F = p.row_off[b + 1] - p.row_off[b]; }
    int len = p.lengths != nullptr ? p.lengths[b] : F;
    len = len < 1 ? 1 : (len > F ? F : len);
    uint32_t kb[QA_NKT];
#pragma unroll
    for (int i = 0; i < QA_NKT; ++i) kb[i] = len >= 32 * i + 32 ? 0xFFFFFFFFu : (len > 32 * i ? (1u << (len - 32 * i)) - 1u : 0u);
    const int nkt = (len + 31) >> 5;

    const int qt = wave, q = lane & 31, h2 = lane >> 5;
    const int qrow = qt * 32 + q;
    const bool active = qt * 32 < F;               // wave-uniform: this wave has rows

    if (tid < 3 * DH) bsm[tid] = p.bias[(tid >> 6) * D + h * DH + (tid & 63)];

    // ---- x rows as operand fragments: k-step s (16 columns) -> lane (row q, half h2) holds columns 16 s + 8 h2 .. + 7, hi and lo
    bf16x8 xh[16], xl[16];
    {
        const bool live = qrow < F;
        const char* xr = reinterpret_cast<const char*>(p.xs) + (row0 + (live ? qrow : 0)) * 1024 + h2 * 16;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            bf16x8 a, c;
#pragma unroll
            for (int e = 0; e < 8; ++e) { a[e] = (__bf16)0.f; c[e] = (__bf16)0.f; }
            if (live && active) {
                a = *reinterpret_cast<const bf16x8*>(xr + (s >> 2) * 256 + (s & 3) * 32);
                c = *reinterpret_cast<const bf16x8*>(xr + (s >> 2) * 256 + (s & 3) * 32 + 128);
            }
            xh[s] = a; xl[s] = c;
        }
    }

    // ---- LDS-DMA of stage g: slab (k-step s, plane) = 1 KiB = [half 2][row 32][16 B]; wave w brings slabs 2 w, 2 w + 1.
    // DMA lane l lands at byte 16 l of the slab: it must fetch weight row (l & 31), half (l >> 5).
    auto issue = [&](int g) __attribute__((always_inline)) {
        const int T = g >> 1, part = T >> 1, hf = T & 1;
        const char* wrow = reinterpret_cast<const char*>(p.w) + (size_t)(part * D + h * DH + 32 * hf + (lane & 31)) * 1024 + (lane >> 5) * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int slab = 2 * wave + i, s = slab >> 1, pl = slab & 1, sa = 8 * (g & 1) + s;
            const char* src = wrow + (sa >> 2) * 256 + pl * 128 + (sa & 3) * 32;
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src),
                                             (__attribute__((address_space(3))) void*)(ring + (g & 1) * QA_STAGE + slab * 1024), 16, 0, 0);
        }
    };

    bf16x8 qh[4], ql[4];
    f32x16 acc;
    issue(0);
    static_for<QA_NSTAGE>([&](auto gc) {
        constexpr int g = decltype(gc)::value, T = g >> 1, khalf = g & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's slabs of stage g (and, the first time, its x rows)
        __syncthreads();                                                // every wave's; and every wave is done with the other buffer
        if constexpr (g + 1 < QA_NSTAGE) issue(g + 1);
        if (active) {
            if constexpr (khalf == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            }
            const char* st = ring + (g & 1) * QA_STAGE + lane * 16;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const bf16x8 wh = *reinterpret_cast<const bf16x8*>(st + (2 * s) * 1024);
                const bf16x8 wl = *reinterpret_cast<const bf16x8*>(st + (2 * s + 1) * 1024);
                const bf16x8 bh = xh[8 * khalf + s], bl = xl[8 * khalf + s];
                if constexpr (T < 4) {             // q^T / k^T tile: [32 d x 32 rows] += W . x^T
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, bl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, bh, acc, 0, 0, 0);
                } else {                           // v tile: [32 rows x 32 d] += x . W^T
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, wh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, wl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, wh, acc, 0, 0, 0);
                }
            }
            if constexpr (khalf == 1) {
                if constexpr (T < 2) {             // q: + bias, / sqrt(64) * log2(e) (the scores come out in base-2 units), split
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int i = 8 * j + e, d = 32 * T + (i & 3) + 8 * (i >> 2) + 4 * h2;
                            v[e] = (acc[i] + bsm[d]) * (0.125f * 1.4426950408889634f);
                        }
                        split8v(v, qh[2 * T + j], ql[2 * T + j]);
                    }
                } else if constexpr (T < 4) {      // k: + bias, split, one 16-byte write per plane and k-step of the score product
                    constexpr int Tk = T - 2;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int i = 8 * j + e, d = 32 * Tk + (i & 3) + 8 * (i >> 2) + 4 * h2;
                            v[e] = acc[i] + bsm[DH + d];
                        }
                        bf16x8 hi, lo;
                        split8v(v, hi, lo);
                        const int ks = 2 * Tk + j;
                        const int off = qrow * DH + (((2 * ks + h2) ^ ((qrow >> 1) & 7)) << 3);
                        *reinterpret_cast<bf16x8*>(Kh + off) = hi;
                        *reinterpret_cast<bf16x8*>(Kl + off) = lo;
                    }
                } else {                           // v: lane = d, four consecutive keys per register quad
                    constexpr int Tv = T - 4;
                    const int d = 32 * Tv + q;
                    const float bv = bsm[2 * DH + d];
#pragma unroll
                    for (int grp = 0; grp < 4; ++grp) {
                        bf16x4 hi, lo;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = acc[4 * grp + e] + bv;
                            hi[e] = (__bf16)v; lo[e] = (__bf16)(v - (float)hi[e]);
                        }
                        const int key0 = 32 * qt + 8 * grp + 4 * h2;
                        *reinterpret_cast<bf16x4*>(Vth + d * QA_VLD + key0) = hi;
                        *reinterpret_cast<bf16x4*>(Vtl + d * QA_VLD + key0) = lo;
                    }
                }
            }
        }
    });
    __syncthreads();                                                    // the K / V images are complete
    if (!active) return;

    // ---- scores, softmax, output: the core of self_attn_bf16x3_kernel
    f32x16 sT[QA_NKT];
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < QA_NKT; ++kt) {
        if (kt < nkt) {
            f32x16 a;
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = 0.f;
            const int r = kt * 32 + q;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off = r * DH + (((2 * ks + h2) ^ ((r >> 1) & 7)) << 3);
                const bf16x8 kh = *reinterpret_cast<const bf16x8*>(Kh + off);
                const bf16x8 kl = *reinterpret_cast<const bf16x8*>(Kl + off);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[ks], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[ks], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[ks], a, 0, 0, 0);
            }
            if (kb[kt] == 0xFFFFFFFFu) {
#pragma unroll
                for (int i = 0; i < 16; ++i) m = fmaxf(m, a[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int kin = (i & 3) + 8 * (i >> 2) + 4 * h2;
                    const float s = ((kb[kt] >> kin) & 1u) ? a[i] : -INFINITY;
                    a[i] = s;
                    m = fmaxf(m, s);
                }
            }
            sT[kt] = a;
        }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));

    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < QA_NKT; ++kt) {
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
    for (int kt = 0; kt < QA_NKT; ++kt) {
        if (kt < nkt) {
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                float pv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) pv[j] = sT[kt][8 * mm + j];
                bf16x8 ph, pl;
                split8v(pv, ph, pl);
                const int k1 = 32 * kt + 16 * mm + 4 * h2;            // keys k1 .. k1+3 and k1+8 .. k1+11
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int voff = (32 * t + q) * QA_VLD + k1;
                    const bf16x4 h_a = *reinterpret_cast<const bf16x4*>(Vth + voff), h_b = *reinterpret_cast<const bf16x4*>(Vth + voff + 8);
                    const bf16x4 l_a = *reinterpret_cast<const bf16x4*>(Vtl + voff), l_b = *reinterpret_cast<const bf16x4*>(Vtl + voff + 8);
                    const bf16x8 vh = {h_a[0], h_a[1], h_a[2], h_a[3], h_b[0], h_b[1], h_b[2], h_b[3]};
                    const bf16x8 vl = {l_a[0], l_a[1], l_a[2], l_a[3], l_b[0], l_b[1], l_b[2], l_b[3]};
                    f32x16& o = t == 0 ? o0 : o1;
                    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, o, 0, 0, 0);
                    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, o, 0, 0, 0);
                    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, o, 0, 0, 0);
                }
            }
        }
    }

    if (qrow < F) {
        const float inv = 1.f / l;
        float* rowp = p.out + (row0 + qrow) * D;
        const int c0 = h * DH + 4 * h2;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            f32x4 v0, v1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] = o0[4 * rg + e] * inv; v1[e] = o1[4 * rg + e] * inv; }
            if (p.split_out) {
                store_split4(rowp, c0 + 8 * rg, v0);
                store_split4(rowp, c0 + 32 + 8 * rg, v1);
            } else {
                st4(rowp + c0 + 8 * rg, v0);
                st4(rowp + c0 + 32 + 8 * rg, v1);
            }
        }
    }
}

// out [M,256] = self-attention(x W_in^T + b_in) per sample over its frames, keys >= lengths[b] masked; xs / w S-format
int launch_dec_qkv_attn(const float* xs, const float* w, const float* bias, const int32_t* lengths, const int32_t* row_off, float* out,
                        int B, int F, int split_out, hipStream_t s) {
    if (F > QA_FMAX || F < 1) return LADIFF_ERR_SHAPE;
    if (B == 0) return 0;
    static bool attr_set[64] = {};
    int dev = 0;
    LADIFF_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return LADIFF_ERR_ARG;
    if (!attr_set[dev]) {
        LADIFF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dec_qkv_attn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, QA_LDS));
        attr_set[dev] = true;
    }
    const QkvAttnArgs a{xs, w, bias, lengths, row_off, out, B, F, split_out};
    hipLaunchKernelGGL(dec_qkv_attn_kernel, dim3(B * H), dim3(512), QA_LDS, s, a);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

}  // namespace ladiff
