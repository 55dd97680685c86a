// Operand tiles in LDS, register-resident weight fragments and the MFMA inner loops of the persistent pipeline
// (systolic.hip).  Two arithmetic policies, see AR below.
#pragma once
#include "common.h"

namespace ladiff {
namespace {

constexpr int CLD = D + 4;                // row stride of the fp32 staging tile in LDS (floats)

// ---------------------------------------------------------------- LDS images
// S-format operand tile: row = KB blocks of 256 B, block = 8 hi slots + 8 lo slots of 16 B, slot index XORed with (row & 15)
template <int KB>
__device__ __forceinline__ char* a_slot(char* tile, int row, int kb, int slot) {
    return tile + row * (KB * 256) + kb * 256 + (((slot ^ row) & 15) << 4);
}
// ---- arithmetic policy AR: 0 = f16x3 (S-format operand tiles, 3 x v_mfma_f32_16x16x32_bf16 per product),
//                           1 = fp32 (fp32 operand tiles, v_mfma_f32_16x16x4_f32: exact fp32 fma chains, the strict-parity mode)
// fp32 tile: row stride K + 4 floats, element k of a row at (k & 3) * (K / 4) + (k >> 2): the lane (row l & 15, k-phase l >> 4) of
// the 16x16x4 MFMA reads its A values of four consecutive k-steps with ONE conflict-free ds_read_b128.
template <int AR, int KB> constexpr int tile_bytes(int rows) { return AR == 0 ? rows * KB * 256 : rows * (KB * 64 + 4) * 4; }

template <int AR, int KB>
__device__ __forceinline__ void tile_put4(char* tile, int row, int k0, const f32x4 v) {            // k0 % 4 == 0
    if constexpr (AR == 0) {
        s16x4 h4, l4;
        split4(v, h4, l4);
        *reinterpret_cast<s16x4*>(a_slot<KB>(tile, row, k0 >> 6, (k0 & 63) >> 3) + (k0 & 7) * 2) = h4;
        *reinterpret_cast<s16x4*>(a_slot<KB>(tile, row, k0 >> 6, 8 + ((k0 & 63) >> 3)) + (k0 & 7) * 2) = l4;
    } else {
        float* r = reinterpret_cast<float*>(tile) + row * (KB * 64 + 4) + (k0 >> 2);
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e * (KB * 16)] = v[e];
    }
}
template <int AR, int KB>
__device__ __forceinline__ void tile_put1(char* tile, int row, int k, float v) {
    if constexpr (AR == 0) {
        const s16 hi = s16_of(v), lo = s16_of(v - (float)hi);
        *(reinterpret_cast<s16*>(a_slot<KB>(tile, row, k >> 6, (k & 63) >> 3)) + (k & 7)) = hi;
        *(reinterpret_cast<s16*>(a_slot<KB>(tile, row, k >> 6, 8 + ((k & 63) >> 3))) + (k & 7)) = lo;
    } else {
        reinterpret_cast<float*>(tile)[row * (KB * 64 + 4) + (k & 3) * (KB * 16) + (k >> 2)] = v;
    }
}

// weights of NT column tiles x KS k-steps (of 32) for this wave, register resident: 8 VGPRs per (tile, step) in both modes
template <int AR, int NT, int KS> struct WFrag;
template <int NT, int KS> struct WFrag<0, NT, KS> { s16x8 hi[NT][KS], lo[NT][KS]; };
template <int NT, int KS> struct WFrag<1, NT, KS> { float w[NT][KS * 8]; };

// w: weight matrix (S-format for AR 0, fp32 for AR 1), row stride ldw floats; tile j covers matrix rows row_of(j) + (lane & 15);
// the k range starts at k block kb0 (64 columns per block)
template <int AR, int NT, int KS, class RowOf>
__device__ __forceinline__ void load_w(WFrag<AR, NT, KS>& f, const float* w, int ldw, int kb0, RowOf row_of) {
    const int lane = threadIdx.x & 63, frow = lane & 15, fk = lane >> 4;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        if constexpr (AR == 0) {
            const char* rp = reinterpret_cast<const char*>(w + (size_t)(row_of(j) + frow) * ldw);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const char* bp = rp + (kb0 + (s >> 1)) * 256 + ((4 * (s & 1) + fk) << 4);
                f.hi[j][s] = *reinterpret_cast<const s16x8*>(bp);
                f.lo[j][s] = *reinterpret_cast<const s16x8*>(bp + 128);
            }
        } else {
            const float* rp = w + (size_t)(row_of(j) + frow) * ldw + kb0 * 64 + fk;
#pragma unroll
            for (int s4 = 0; s4 < KS * 8; ++s4) f.w[j][s4] = rp[4 * s4];
        }
    }
}

// The weight fragments are loaded once, in front of a role's block loop.  To the compiler's wait-count pass their loads are still
// "possibly in flight" at the loop header (a merge point), so it guards their first use INSIDE the loop with `s_waitcnt vmcnt(0)` - on
// every iteration, where it also waits for whatever the iteration has requested itself (the next block's rows: the whole load round
// trip, 0.3 us per block, stood in front of the first product of the busiest stage).  landed(): wait for every load once, in front of
// the loop, with the BUILTIN (an instruction the pass sees and accounts for - an asm statement it does not).
__device__ __forceinline__ void landed() { __builtin_amdgcn_s_waitcnt(0x0F70); }      // vmcnt(0), nothing else (gfx9 encoding)

// ---- LDS operand fetches the compiler does not reschedule.  Left to itself the compiler sinks every ds_read next to its use
// and waits with lgkmcnt(0): the LDS latency is then paid in front of every 2 - 4 MFMAs.  These reads are issued as asm (the
// compiler does not track their completion) and waited for with an explicit counted s_waitcnt that is tied to the registers
// it guards (the "+v" operands), so the reads of step s + 1 are in flight under the MFMAs of step s.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {                     // f(IntC<0>{}) ... f(IntC<N - 1>{})
    if constexpr (I < N) { f(IntC<I>{}); static_for<N, I + 1>(f); }
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
template <int OFF>
__device__ __forceinline__ void lds_fetch16(u32x4_t& v, unsigned addr) {          // 16 bytes at LDS address addr + OFF
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait(u32x4_t& a, u32x4_t& b) {               // all but the N youngest LDS operations are done
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int N>
__device__ __forceinline__ void lds_wait(u32x4_t& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }

// acc[i][j] += A(tile rows 16 i ..) . W(tile j)^T over KS k-steps of 32; MR = row tiles.  The operand fragments of the next
// PF steps are requested before the MFMAs of the current one are issued, so their LDS latency hides under the MFMAs.
// NTF >= NT: the fragment set may hold more column tiles than are multiplied (its first NT are used).
// PF = how many k-steps the fragment fetches run ahead (f16x3 tiles): a step of ONE column tile is 3 MFMAs = 48 cycles of the SIMD's
// matrix pipe (96 with the SIMD's second wave in the same phase) - less than an LDS read takes when all eight waves fetch at once, so
// with PF = 1 every step of such a product waits for its fragments; PF = 2 keeps two steps in flight (8 more registers per row tile).
// TR: the product TRANSPOSED, D^T = W . A^T (the weight fragment as the MFMA's first operand): the same products summed in the same
// order - the same bits - but the accumulator of lane (l & 15, l >> 4) then holds FOUR CONSECUTIVE COLUMNS 4 (l >> 4) .. + 3 of tile
// row l & 15 instead of one column of four rows: what follows (an operand tile, the fp32 staging tile) is written 8 / 16 bytes at a
// time instead of 2 / 4 (tile_put4 / stage_c_t against tile_put1 / stage_c).
template <int AR, int KB, int NT, int KS, int MR, int NTF = NT, int PF = 1, bool TR = false>
__device__ __forceinline__ void mma(const char* tile, const WFrag<AR, NTF, KS>& f, f32x4 (&acc)[MR][NT]) {
    const int lane = threadIdx.x & 63, frow = lane & 15, fk = lane >> 4;
    if constexpr (AR == 0) {
        static_assert(PF == 1 || PF == 2, "fetch distance");
        constexpr int NB = PF + 1;                                       // fragment buffers
        // lane part of the four slot addresses a step can need: hi / lo plane x even / odd step of a 64-column block
        // (a_slot: slot ^ row, and row = 16 i + frow leaves the low four bits to frow); the rest is an immediate offset
        unsigned base[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) base[v] = lds_addr(tile) + frow * (KB * 256) + ((((v & 1) * 4 + (v >> 1) * 8 + fk) ^ frow) & 15) * 16;
        u32x4_t ah[NB][MR], al[NB][MR];
        auto fetch = [&](auto sc) {                                      // step sc.value into buffer sc.value % NB
            constexpr int s = decltype(sc)::value, bf = s % NB;
            static_for<MR>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                lds_fetch16<16 * i * KB * 256 + (s >> 1) * 256>(ah[bf][i], base[s & 1]);
                lds_fetch16<16 * i * KB * 256 + (s >> 1) * 256>(al[bf][i], base[2 + (s & 1)]);
            });
        };
        static_for<(PF < KS ? PF : KS)>([&](auto sc) { fetch(sc); });
        static_for<KS>([&](auto sc) {
            constexpr int s = decltype(sc)::value, cur = s % NB;
            if constexpr (s + PF < KS) fetch(IntC<s + PF>{});
            constexpr int ahead = (KS - 1 - s) < PF ? (KS - 1 - s) : PF;   // steps whose fragments may still be in flight behind this one's
            static_for<MR>([&](auto ic) {                                // the current step's fragments: all but the 2 MR x `ahead` requested after them
                constexpr int i = decltype(ic)::value;
                lds_wait<2 * MR * ahead>(ah[cur][i], al[cur][i]);
            });
            __builtin_amdgcn_sched_barrier(0);
            // the three products of a split operand pair go to the same accumulator: one product at a time over all MR x NT
            // accumulators, so that consecutive MFMAs never depend on each other
            auto mm = [&](const s16x8 a, const s16x8 w, const f32x4 c) __attribute__((always_inline)) {
                if constexpr (TR) return MFMA16_S16(w, a, c, 0, 0, 0);
                else return MFMA16_S16(a, w, c, 0, 0, 0);
            };
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mm(__builtin_bit_cast(s16x8, al[cur][i]), f.hi[j][s], acc[i][j]);
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mm(__builtin_bit_cast(s16x8, ah[cur][i]), f.lo[j][s], acc[i][j]);
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mm(__builtin_bit_cast(s16x8, ah[cur][i]), f.hi[j][s], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        });
    } else {
        constexpr int LD = KB * 64 + 4, NG = KS * 2;                     // groups of four k-steps of 4
        const unsigned base = lds_addr(tile) + (frow * LD + fk * (KB * 16)) * 4;
        u32x4_t a[2][MR];
        auto fetch = [&](auto gc, auto bc) {
            constexpr int g = decltype(gc)::value, bf = decltype(bc)::value;
            static_for<MR>([&](auto ic) { constexpr int i = decltype(ic)::value; lds_fetch16<(16 * i * LD + 4 * g) * 4>(a[bf][i], base); });
        };
        fetch(IntC<0>{}, IntC<0>{});
        static_for<NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value, cur = g & 1;
            if constexpr (g + 1 < NG) fetch(IntC<g + 1>{}, IntC<cur ^ 1>{});
            static_for<MR>([&](auto ic) { constexpr int i = decltype(ic)::value; lds_wait<(g + 1 < NG ? MR : 0)>(a[cur][i]); });
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        if constexpr (TR) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w[j][4 * g + e], __builtin_bit_cast(f32x4, a[cur][i])[e], acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(f32x4, a[cur][i])[e], f.w[j][4 * g + e], acc[i][j], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
        });
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

// the same for the accumulators of a TRANSPOSED product (mma<..., TR = true>): lane (frow, fk) holds columns col_of(j) + 4 fk .. + 3 of
// row 16 i + frow - one 16-byte write per tile instead of four 4-byte ones
template <int MR, int NT, class ColOf>
__device__ __forceinline__ void stage_c_t(float* ct, const f32x4 (&acc)[MR][NT], ColOf col_of) {
    const int lane = threadIdx.x & 63, frow = lane & 15, fk = lane >> 4;
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4*>(ct + (16 * i + frow) * CLD + col_of(j) + 4 * fk) = acc[i][j];
}

}  // namespace
}  // namespace ladiff
