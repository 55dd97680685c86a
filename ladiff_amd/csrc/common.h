// Shared definitions for the LADiff gfx950 kernels (CDNA4, wave64, fp32-input MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ladiff_hip.h"

namespace ladiff {

constexpr int D = 256;     // latent / model width      (config_ladiff_humanml3d.yaml:132 latent_dim[-1])
constexpr int H = 4;       // heads                     (configs/modules/denoiser.yaml:7)
constexpr int DH = 64;     // head dim
constexpr float LN_EPS = 1e-5f;
constexpr int CLIP_MAX_LAYERS = LADIFF_CLIP_MAX_LAYERS;
constexpr int CLIP_MAX_POSITIONS = LADIFF_CLIP_MAX_POSITIONS;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- error plumbing: 0 ok, <0 argument error, >0 hipError_t (include/ladiff_hip.h)
#define LADIFF_CHECK_ARG(cond) \
    do { if (!(cond)) return LADIFF_ERR_ARG; } while (0)
#define LADIFF_HIP(expr) \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)
#define LADIFF_TRY(expr) \
    do { int r_ = (expr); if (r_ != 0) return r_; } while (0)
#define LADIFF_LAUNCH_CHECK() LADIFF_HIP(hipGetLastError())

// ---- activations
// erf for the exact-GELU: Abramowitz-Stegun 7.1.26, |error| <= 6e-7 absolute evaluated in fp32 (GELU: <= 5e-7), 1 rcp + 1 exp2 +
// 8 FMAs.  The device library's erff (~1 ulp, two branches) costs ~60 instructions per value, which at 20 values per
// thread was 2 us of every ffn.linear1 launch of the denoiser loop (and 40 us of the decoder's linear1 GEMM).
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
    float q = fmaf(t, 1.061405429f, -1.453152027f);
    q = fmaf(t, q, 1.421413741f);
    q = fmaf(t, q, -0.284496736f);
    q = fmaf(t, q, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    return copysignf(fmaf(-q * t, e, 1.f), x);
}
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.f + erf_as(v * 0.70710678118654752440f)); }
// two values at once with the packed fp32 instructions of gfx950 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: one issue slot for two
// IEEE operations): the same operations in the same order as gelu_erf on each component, hence the same bits; where a role's GELU is
// bound by vector issue (the pipeline's FFN stage: 4 values per lane, two waves per SIMD) the polynomial and the affine steps cost half
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 v) {
    const f32x2 x = v * 0.70710678118654752440f;
    const f32x2 ax = f32x2{fabsf(x[0]), fabsf(x[1])};
    const f32x2 d = __builtin_elementwise_fma(f32x2{0.3275911f, 0.3275911f}, ax, f32x2{1.f, 1.f});
    const f32x2 t = f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    f32x2 q = __builtin_elementwise_fma(t, f32x2{1.061405429f, 1.061405429f}, f32x2{-1.453152027f, -1.453152027f});
    q = __builtin_elementwise_fma(t, q, f32x2{1.421413741f, 1.421413741f});
    q = __builtin_elementwise_fma(t, q, f32x2{-0.284496736f, -0.284496736f});
    q = __builtin_elementwise_fma(t, q, f32x2{0.254829592f, 0.254829592f});
    const f32x2 a2 = (-1.4426950408889634f * ax) * ax;
    const f32x2 e = f32x2{__builtin_amdgcn_exp2f(a2[0]), __builtin_amdgcn_exp2f(a2[1])};
    const f32x2 r = __builtin_elementwise_fma(-q * t, e, f32x2{1.f, 1.f});
    const f32x2 er = f32x2{copysignf(r[0], x[0]), copysignf(r[1], x[1])};
    return (0.5f * v) * (1.f + er);
}

enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_SILU = 3, ACT_QGELU = 4, ACT_LRELU = 5 };

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case ACT_RELU: return fmaxf(v, 0.f);
        case ACT_GELU: return gelu_erf(v);                                            // erf GELU (F.gelu default)
        case ACT_SILU: return v / (1.f + expf(-v));
        case ACT_QGELU: return v / (1.f + expf(-1.702f * v));                         // CLIP's quick_gelu: x * sigmoid(1.702 x)
        case ACT_LRELU: return v > 0.f ? v : 0.2f * v;                                // LeakyReLU(0.2) of the T2M evaluators
        default: return v;
    }
}

// Compile-time activation + a dispatcher that hoists the (uniform) switch out of the element loops.  A per-element
// runtime switch costs a handful of taken scalar branches per value and, after every kernel boundary, instruction-cache
// misses across a multi-KiB epilogue: measured 5 k cycles for the 20 values per thread of the 80x64 denoiser tile.
template <int ACT>
__device__ __forceinline__ float act_c(float v) {
    if constexpr (ACT == ACT_RELU) return fmaxf(v, 0.f);
    else if constexpr (ACT == ACT_GELU) return gelu_erf(v);
    else if constexpr (ACT == ACT_SILU) return v / (1.f + expf(-v));
    else if constexpr (ACT == ACT_QGELU) return v / (1.f + expf(-1.702f * v));
    else if constexpr (ACT == ACT_LRELU) return v > 0.f ? v : 0.2f * v;
    else return v;
}
template <int V> struct IntC { static constexpr int value = V; };
template <class F>
__device__ __forceinline__ void act_dispatch(int act, F&& f) {
    switch (act) {
        case ACT_RELU: f(IntC<ACT_RELU>{}); break;
        case ACT_GELU: f(IntC<ACT_GELU>{}); break;
        case ACT_SILU: f(IntC<ACT_SILU>{}); break;
        case ACT_QGELU: f(IntC<ACT_QGELU>{}); break;
        case ACT_LRELU: f(IntC<ACT_LRELU>{}); break;
        default: f(IntC<ACT_NONE>{}); break;
    }
}

__device__ __forceinline__ float silu(float v) { return v / (1.f + expf(-v)); }

// ---- wave-level reductions (64 lanes)
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {   // sum over aligned groups of WIDTH lanes
#pragma unroll
    for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int WIDTH>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int o = WIDTH / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- DPP reductions (VALU speed: no ds_bpermute round trips).  gfx9 DPP controls: quad_perm 0x00-0xFF,
// row_half_mirror 0x141, row_mirror 0x140, row_bcast15 0x142, row_bcast31 0x143.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_add(float v) {     // v + dpp(v); lanes without a source add 0 (bound_ctrl)
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {   // sum over each aligned group of 16 lanes, result in all 16
    v = dpp_add<0xB1>(v);      // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);      // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);     // row_half_mirror
    v = dpp_add<0x140>(v);     // row_mirror
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {    // sum over the 64 lanes, result in all lanes
    v = row16_sum(v);
    v = dpp_add<0x142, 0xA>(v);    // row_bcast15: rows 1,3 += last lane of rows 0,2
    v = dpp_add<0x143, 0xC>(v);    // row_bcast31: rows 2,3 += lane 31
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ---- "split" format (S-format).  A row of K fp32 values occupies the same K*4 bytes as K/64 blocks of
// [64 x s16 hi | 64 x s16 lo] with x ~ hi + lo.  A product of two split operands is evaluated as hi*hi + hi*lo + lo*hi on the 16-bit
// MFMA (fp32 accumulate).  The half type s16 is a property of the BUILD (ladiff_split_format()):
//   * fp16 (the product since round 6, "f16x3"): 11 + 11 = 22 significant bits - a K = 768 product of random operands has an rms error
//     of 2.7e-7 of its rms value, LESS than a sequential fp32 fma chain's 5e-7, against 4.6e-6 for bf16 pairs
//     (scripts/ubench_mfma_f16_denorm.hip, profiles/r6/02_*); the MFMA keeps subnormal fp16 inputs (same probe), which is what the lo
//     halves of small values are.  The price is fp16's exponent range: each half SATURATES at +-65504 (v_cvt_pkrtz_f16_f32 rounds
//     toward zero and therefore never produces an infinity), so |x| <= 65504 is exact to 22 bits, up to 131008 degrades gracefully,
//     beyond is clipped.  The largest GEMM operand of the path on random-init weights (the hard case: latents reach |x| ~ 280) is
//     stated in DESIGN.md 1; every product input of these networks is a LayerNorm output, an activation of one, a softmax
//     probability or a latent.  Same rate, same instruction count as the bf16 form (one v_cvt_pkrtz per two values where the bf16
//     form needs one v_cvt_pk_bf16 per value).
//   * bf16 (-DLADIFF_SPLIT_BF16, rounds 1 - 5, "bf16x3"): 16 significant bits, fp32's exponent range; 2e-4 on the decoded frames of the
//     50-step benchmark where fp16 pairs give 3e-5.
#ifndef LADIFF_SPLIT_BF16
typedef _Float16 s16;                          // one half of an S-format pair
#define MFMA16_S16 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define MFMA32_S16 __builtin_amdgcn_mfma_f32_32x32x16_f16
__device__ __forceinline__ s16 s16_of(float v) { return (_Float16)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f); }   // saturating
typedef s16 s16x2 __attribute__((ext_vector_type(2)));
// two values -> (hi, hi), (lo, lo): v_cvt_pkrtz_f16_f32 converts a pair per instruction and, rounding toward zero, SATURATES at +-65504
// instead of producing an infinity; x - hi is exact in fp32 (hi keeps x's leading 11 bits), its conversion keeps 11 more: 22 bits.
__device__ __forceinline__ void split2(float a, float b, s16x2& hi, s16x2& lo) {
    hi = __builtin_bit_cast(s16x2, __builtin_amdgcn_cvt_pkrtz(a, b));
    lo = __builtin_bit_cast(s16x2, __builtin_amdgcn_cvt_pkrtz(a - (float)hi[0], b - (float)hi[1]));
}
#else
typedef __bf16 s16;
#define MFMA16_S16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define MFMA32_S16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
__device__ __forceinline__ s16 s16_of(float v) { return (__bf16)v; }
typedef s16 s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(float a, float b, s16x2& hi, s16x2& lo) {
    hi[0] = (__bf16)a; hi[1] = (__bf16)b;
    lo[0] = (__bf16)(a - (float)hi[0]); lo[1] = (__bf16)(b - (float)hi[1]);
}
#endif
typedef s16 s16x8 __attribute__((ext_vector_type(8)));
typedef s16 s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split4(float a, float b, float c, float d, s16x4& hi, s16x4& lo) {
    s16x2 h0, l0, h1, l1;
    split2(a, b, h0, l0); split2(c, d, h1, l1);
    hi = s16x4{h0[0], h0[1], h1[0], h1[1]}; lo = s16x4{l0[0], l0[1], l1[0], l1[1]};
}
__device__ __forceinline__ void split4(const f32x4 v, s16x4& hi, s16x4& lo) { split4(v[0], v[1], v[2], v[3], hi, lo); }
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, s16x8& hi, s16x8& lo) {
    s16x4 ha, la, hb, lb;
    split4(a, ha, la); split4(b, hb, lb);
    hi = s16x8{ha[0], ha[1], ha[2], ha[3], hb[0], hb[1], hb[2], hb[3]}; lo = s16x8{la[0], la[1], la[2], la[3], lb[0], lb[1], lb[2], lb[3]};
}
__device__ __forceinline__ void store_split4(float* row, int k, const float (&v)[4]) {   // columns k..k+3 of an S-format row
    s16x4 hi, lo;
    split4(v[0], v[1], v[2], v[3], hi, lo);
    char* base = reinterpret_cast<char*>(row) + ((k >> 6) << 8) + ((k & 63) << 1);
    *reinterpret_cast<s16x4*>(base) = hi;
    *reinterpret_cast<s16x4*>(base + 128) = lo;
}
__device__ __forceinline__ void store_split4(float* row, int k, f32x4 v) {
    const float t[4] = {v[0], v[1], v[2], v[3]};
    store_split4(row, k, t);
}
__device__ __forceinline__ void store_split1(float* row, int k, float v) {
    const s16 hi = s16_of(v), lo = s16_of(v - (float)hi);
    s16* base = reinterpret_cast<s16*>(reinterpret_cast<char*>(row) + ((k >> 6) << 8)) + (k & 63);
    base[0] = hi;
    base[64] = lo;
}

// Kernel-argument fields used late in a kernel (epilogue pointers, strides, flags): read them once at entry and pin the
// copy in SGPRs.  Otherwise the compiler re-reads the kernarg segment next to every use - after each global store it can
// no longer prove the field unchanged - and every re-read is an s_load + s_waitcnt on the critical path (measured:
// 3 per 16-byte store in the GEMM epilogues, ~2 k cycles per workgroup).
template <class T>
__device__ __forceinline__ T pin_s(T v) { asm volatile("" : "+s"(v)); return v; }

// Redefine a register value in the compiler's eyes (no instruction).  Used on values that were loaded long ago and are
// known to have landed (explicit s_waitcnt): without it the compiler guards every later use with its own conservative
// `s_waitcnt vmcnt(0)`, which in a store loop means "wait for the previous store to reach memory" (~500 cycles each).
__device__ __forceinline__ void reg_touch(f32x4& v) { asm volatile("" : "+v"(v)); }

// pin_s() launders a pointer, so the compiler no longer knows it points to global memory and would emit flat_* accesses;
// these say so explicitly.
typedef __attribute__((address_space(1))) f32x4 g_f32x4;
typedef __attribute__((address_space(1))) s16x4 g_s16x4;
__device__ __forceinline__ void st4g(float* p, f32x4 v) { *(g_f32x4*)p = v; }
__device__ __forceinline__ f32x4 ld4g(const float* p) { return *(const g_f32x4*)p; }
__device__ __forceinline__ void store_split4g(float* row, int k, f32x4 v) {      // store_split4 on a global row
    s16x4 hi, lo;
    split4(v, hi, lo);
    char* base = reinterpret_cast<char*>(row) + ((k >> 6) << 8) + ((k & 63) << 1);
    *(g_s16x4*)base = hi;
    *(g_s16x4*)(base + 128) = lo;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

}  // namespace ladiff
