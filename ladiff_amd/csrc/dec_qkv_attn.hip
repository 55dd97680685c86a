// Decoder self-attention with its in_proj inside (f16x3 mode): one workgroup per (sample, head) computes that head's q | k | v rows
// from the S-format x rows and attends, instead of an in_proj GEMM that writes [M, 768] fp32 (77 MB at M = 25088) and an attention
// kernel that reads it back and re-splits it (cross_attention.py:367-369; nn.MultiheadAttention's packed in_proj).
//
//   * wave w owns frame rows 32 w .. 32 w + 31 both as queries and as keys / values.  Its x rows are MFMA operand fragments in
//     registers (16 k-steps x (hi, lo) s16x8 per lane, loaded straight from the S-format rows).
//   * the head's 192 in_proj rows (64 q, 64 k, 64 v) stream through LDS as twelve 16-KiB stages (32 weight rows x 128 k, hi and lo
//     planes) by LDS-DMA; a stage's fragment = one conflict-free ds_read_b128 per lane, issued by asm one k-step ahead of its MFMAs.
//     The stages land in the K / V image areas while those are not written yet (eight of the twelve are requested before the
//     loop starts, see slab_off), so no stage is requested less than two stages ahead although the images leave only 32 KiB.
//     Every wait for a stage is a `vmcnt(0)` of the waves that requested it (see `issue`).
//   * q and k are computed TRANSPOSED (D^T = W x^T, v_mfma_f32_32x32x16_bf16: weight fragment first), v plainly (x first): the
//     accumulator of a q^T / k^T tile then has the row's query / key on the lane and 16 of its d values in the registers, and since
//     d is the contraction index of q.k ANY order of d is fine as long as q and k share it: registers 8 j .. 8 j + 7 of tile T ARE the
//     q operand fragment of k-step 2 T + j of the score product (no cross-lane move), and the same registers of a k^T tile are one
//     16-byte write into chunk 2 (2 T + j) + h2 of the K image.  A v tile has its d on the lane and four consecutive keys per
//     register quad: an 8-byte write into the transposed V image the output product reads.
//   * then the score / softmax / output core of self_attn_split_kernel (attention.hip), unchanged: S^T = K Q^T, fp32 softmax in
//     base 2, O^T = V^T P^T, every product as lo*hi + hi*lo + hi*hi.
//   * the four heads of a sample get block ids of one residue class mod 8 (one XCD, one L2: they read the same x rows); the S-format
//     result leaves through the dead K image as whole 256-byte blocks.
// Arithmetic: the products of the separate kernels (same splits, fp32 accumulate); q.k sums its 64 d in another order.
// Measured (128 x 196 frames, profiles/r3/15_*): 70 us per layer against 46 (in_proj GEMM) + 41 (attention kernel); decode 2.41 ->
// 2.26 ms.  What the 70 us are (timing builds, 2 rounds of 256 workgroups): x row fragments 20 us (16-byte pieces of 32 rows per
// load instruction: the texture path moves them at a quarter of its rate; LDS has no room to stage 196 KiB of rows), projection
// 25 us (its MFMAs 12), scores / softmax / output 27 us (MFMAs ~11, the rest is the softmax and bf16-split VALU of two waves per SIMD
// that reach the same phase together).  Steps that did NOT move it: look-ahead of two stages instead of one (the compiler had put
// `vmcnt(0)` in front of every plain LDS read - LDS-DMA may alias - so the first build never looked ahead at all; with asm reads
// and counted waits the waits are simply not what binds), asm-pipelined K / V fragment reads in the core, whole-block stores.
#include <mutex>
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
constexpr int QA_LDS_BASE = QA_K_BYTES + QA_V_BYTES + 2 * QA_STAGE + 3 * DH * 4;      // images, ring, bias table
constexpr int QA_LDS = QA_LDS_BASE + 3 * 4096;                                     // + the x-row scratch of waves 4 - 6 (see the kernel)
static_assert(QA_LDS <= 160 * 1024 && QA_K_BYTES - 3 * QA_STAGE >= 8192 && QA_V_BYTES - 3 * QA_STAGE >= 8192, "scratch areas");

struct QkvAttnArgs {
    const float* xs;           // [M, 256] S-format rows
    const float* w;            // in_proj_weight [768, 256] S-format
    const float* bias;         // in_proj_bias [768]
    const int32_t* lengths;    // [B]
    const int32_t* row_off;    // ragged rows: sample b owns rows row_off[b] .. row_off[b + 1] (NULL: b F ..)
    float* out;                // [M, 256] fp32 or S-format
    int B, F, split_out;
};

// LDS byte offset of slab `slab` (0 .. 15) of weight stage G (see the kernel: stages land in the not-yet-written K / V images early on)
template <int G>
__device__ __forceinline__ constexpr int slab_off(int slab) {
    constexpr int VREG = QA_K_BYTES, RING = QA_K_BYTES + QA_V_BYTES, VTL = DH * QA_VLD * 2, ROW32 = 32 * QA_VLD * 2;
    if constexpr (G <= 2) return G * QA_STAGE + slab * 1024;
    else if constexpr (G <= 5) return VREG + (G - 3) * QA_STAGE + slab * 1024;
    else if constexpr (G == 6 || G == 9) return RING + slab * 1024;
    else if constexpr (G == 7 || G == 10) return RING + QA_STAGE + slab * 1024;
    else if constexpr (G == 8) return VREG + slab * 1024;
    else return slab < 8 ? VREG + ROW32 + slab * 1024 : VREG + VTL + ROW32 + (slab - 8) * 1024;
}
static_assert(3 * QA_STAGE <= QA_K_BYTES && 3 * QA_STAGE <= QA_V_BYTES && 32 * QA_VLD * 2 + 8 * 1024 <= DH * QA_VLD * 2, "weight stages inside the images");

typedef unsigned u32x4_q __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ void frag_read(u32x4_q& v, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536 * 4, "");
    if constexpr (OFF < 65536) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    else if constexpr (OFF < 2 * 65536) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr + 65536u), "n"(OFF - 65536) : "memory");
    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr + 131072u), "n"(OFF - 131072) : "memory");
}

__device__ __forceinline__ void lds_put16(unsigned addr, const u32x4_q& v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void wait_vm4(u32x4_q& a, u32x4_q& b, u32x4_q& c, u32x4_q& d) { asm volatile("s_waitcnt vmcnt(4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
__device__ __forceinline__ void wait_vm0(u32x4_q& a, u32x4_q& b, u32x4_q& c, u32x4_q& d) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
__device__ __forceinline__ void wait_lgkm0(u32x4_q& a, u32x4_q& b, u32x4_q& c, u32x4_q& d, u32x4_q& e, u32x4_q& f, u32x4_q& g, u32x4_q& h) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
}

// two 8-byte reads (offsets OFF8 and OFF8 + 2, in units of 8 bytes) into one register quad
template <int OFF8>
__device__ __forceinline__ void v_read2(u32x4_q& v, unsigned addr) {
    static_assert(OFF8 >= 0 && OFF8 + 2 < 256, "");
    asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(OFF8), "n"(OFF8 + 2) : "memory");
}

__device__ __forceinline__ void split8v(const float (&v)[8], s16x8& hi, s16x8& lo) {
    split8(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]}, hi, lo);
}

}  // namespace

// DIAG (timing experiments, garbage results): 1 = the stage loop does not wait for its LDS-DMA, 2 = no score / softmax / output core,
// 3 = no projection MFMAs, 4 = no projection at all (no weight stages), 5 = no x row loads, 6 = nothing but the x row loads and one store
template <int DIAG = 0>
__global__ __launch_bounds__(512) void dec_qkv_attn_kernel(const QkvAttnArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    s16* const Kh = reinterpret_cast<s16*>(lds);
    s16* const Kl = Kh + QA_FMAX * DH;
    s16* const Vth = reinterpret_cast<s16*>(lds + QA_K_BYTES);
    s16* const Vtl = Vth + DH * QA_VLD;
    char* const ring = lds + QA_K_BYTES + QA_V_BYTES;
    float* const bsm = reinterpret_cast<float*>(ring + 2 * QA_STAGE);          // [q 64 | k 64 | v 64] of this head
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroup n runs on XCD n % 8 (round-robin dispatch) and each XCD has its own L2: the four heads of a sample read the SAME x
    // rows, so they are given block ids of one residue class, 8 apart - sample 8 k + r, head j = block 8 (4 k + j) + r - and only the
    // first of them fetches the rows from memory (with b = n / 4 the four heads sat on four XCDs: 4 x 25.7 MB from the memory side
    // per layer, 24 us of the kernel)
    const int xr = blockIdx.x & 7, xm = blockIdx.x >> 3;
    const int b = 8 * (xm / H) + xr, h = xm % H;
    if (b >= p.B) return;
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

    // ---- LDS-DMA of stage g: slab (k-step s, plane) = 1 KiB = [half 2][row 32][16 B]; wave w brings slabs 2 w, 2 w + 1.
    // DMA lane l lands at byte 16 l of the slab: it must fetch weight row (l & 31), half (l >> 5).
    // WHERE a stage lands (slab_off): the K and V images are not written before stages 5 and 9, so the weight stages use them as
    // ring space early on - stages 0 .. 7 are all requested before the loop starts (K region 0-2, V region 3-5, the two ring
    // buffers 6, 7), stage 8 goes to the V region again, 9 / 10 to the ring buffers, 11 to the rows of the V planes that only ITS
    // OWN epilogue writes (d 32 .. 63: a barrier stands between its products and that epilogue) - every stage is requested at
    // least two stages ahead.  With two buffers and one stage of look-ahead every stage stood ~1.5 us waiting for its 16 KiB.
    // WHO requests a stage: stages 0 .. 7 (all requested before the loop and all landed before it starts: the wait for the x rows
    // below is a `vmcnt(0)`) every wave its slabs 2 w, 2 w + 1; stages 8 .. 11 (requested inside the loop, one stage ahead) the EVEN ones
    // waves 0 - 3 and the ODD ones waves 4 - 7, four slabs each - so a wave never has two stages in flight and its wait for one is
    // `vmcnt(0)`: LDS-DMA requests of a wave do not complete in issue order when their latencies differ (gemm_big.hip, header), and
    // the counted `vmcnt(2)` of rounds 3 - 4 could be satisfied by the younger stage's two requests.
    auto issue = [&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value, T = g >> 1, part = T >> 1, hf = T & 1;
        const char* wrow = reinterpret_cast<const char*>(p.w) + (size_t)(part * D + h * DH + 32 * hf + (lane & 31)) * 1024 + (lane >> 5) * 16;
        constexpr int NI = g < 8 ? 2 : 4;
        if (g >= 8 && (wave >> 2) != (g & 1)) return;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int slab = g < 8 ? 2 * wave + i : 4 * (wave & 3) + i, s = slab >> 1, pl = slab & 1, sa = 8 * (g & 1) + s;
            const char* src = wrow + (sa >> 2) * 256 + pl * 128 + (sa & 3) * 32;
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src), (__attribute__((address_space(3))) void*)(lds + slab_off<g>(slab)), 16, 0, 0);
        }
    };

    s16x8 qh[4], ql[4];
    f32x16 acc;
    const unsigned lds_lane = lds_addr(lds) + lane * 16;
    if constexpr (DIAG != 4) static_for<8>([&](auto gc) { issue(gc); });

    // ---- x rows as operand fragments: k-step s (16 columns) -> lane (row q, half h2) holds columns 16 s + 8 h2 .. + 7, hi and lo.
    // Round 5: loaded a 128-byte LINE of every row at a time and turned into the fragment layout through a 4-KiB LDS scratch per wave.
    // A fragment load straight from memory takes 16 bytes per lane from 32 different rows - 32 bytes of a row per instruction, a
    // quarter of every line it touches (20 of the kernel's 66 us, timing builds of round 3); here one instruction fetches whole lines
    // (8 rows x 128 B: lane l -> row 8 i + (l >> 3), piece l & 7), writes them to the scratch (row r at 128 r, piece p at slot
    // p ^ ((r >> 1) & 7)) and four reads per line give lane (q, h2) its pieces 2 e + h2 = k-steps 4 blk + e of that line (hi / lo plane).
    // The scratch areas are LDS nobody else uses before the stage loop: the 8-KiB tail of the K image (waves 0, 1), 8 KiB of the V
    // image's tail (2, 3), 12 KiB behind the bias table (4 - 6; wave 7 never has rows: QA_FMAX = 7 x 32).  All LDS traffic is asm
    // (the compiler would put a `vmcnt(0)` - the weight stages in flight - in front of every plain access), same wave writes and reads.
    s16x8 xh[16], xl[16];
    {
        const bool live = qrow < F;
        unsigned scr = lds_addr(lds) + (wave < 2 ? 3 * QA_STAGE + wave * 4096
                                        : wave < 4 ? QA_K_BYTES + 3 * QA_STAGE + (wave - 2) * 4096
                                                   : QA_LDS_BASE + ((wave - 4) % 3) * 4096);
        const unsigned wr = scr + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 4) & 3)) << 4);      // + 1024 i: row r = 8 i + (l >> 3), (r >> 1) & 7 = 4 (i & 1) + (l >> 4) - see below
        const unsigned rd = scr + q * 128;
        const int sw = (q >> 1) & 7;
        u32x4_q t[2][4], xq[2][16];                                          // lines in flight; the fragments as the asm reads deliver them (hi, lo)
        if (!(active && DIAG != 5)) {                                        // (defined registers on the path that skips the loads)
#pragma unroll
            for (int s = 0; s < 16; ++s) { xq[0][s] = u32x4_q{0u, 0u, 0u, 0u}; xq[1][s] = xq[0][s]; }
        }
        auto fetch_line = [&](int j, int bf) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r = qt * 32 + 8 * i + (lane >> 3);
                r = r < F ? r : 0;                                           // rows past the sample's frames: any valid row (their fragments are zeroed)
                t[bf][i] = *reinterpret_cast<const u32x4_q*>(reinterpret_cast<const char*>(p.xs) + (row0 + r) * 1024 + (j >> 1) * 256 + (j & 1) * 128 + (lane & 7) * 16);
            }
        };
        if (active && DIAG != 5) {
            fetch_line(0, 0);
            static_for<8>([&](auto jc) {
                constexpr int j = decltype(jc)::value, bf = j & 1;
                if constexpr (j + 1 < 8) fetch_line(j + 1, bf ^ 1);
                // line j has landed: everything but the four loads just issued (the first time: the weight stages requested above as well)
                if constexpr (j + 1 < 8) wait_vm4(t[bf][0], t[bf][1], t[bf][2], t[bf][3]);
                else wait_vm0(t[bf][0], t[bf][1], t[bf][2], t[bf][3]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // row r = 8 i + (l >> 3): (r >> 1) & 7 = (4 i + (l >> 4)) & 7 - the lane part is in `wr`, the i part flips slot bit 2
                    lds_put16((wr + 1024 * i) ^ ((i & 1) ? 64u : 0u), t[bf][i]);
                }
                static_for<4>([&](auto ec) {
                    constexpr int e = decltype(ec)::value, ks = 4 * (j >> 1) + e;
                    frag_read<0>(xq[j & 1][ks], rd + ((((2 * e) | h2) ^ sw) << 4));
                });
            });
            // the reads have returned (asm reads: the compiler does not know they are in flight - the registers are tied to the wait)
            static_for<4>([&](auto gc) {
                constexpr int g4 = 4 * decltype(gc)::value;
                wait_lgkm0(xq[0][g4], xq[0][g4 + 1], xq[0][g4 + 2], xq[0][g4 + 3], xq[1][g4], xq[1][g4 + 1], xq[1][g4 + 2], xq[1][g4 + 3]);
            });
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (live && active && DIAG != 5) { xh[s] = __builtin_bit_cast(s16x8, xq[0][s]); xl[s] = __builtin_bit_cast(s16x8, xq[1][s]); }
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { xh[s][e] = s16_of(0.f); xl[s][e] = s16_of(0.f); }
            }
        }
    }

    // every x register is used here once, so the compiler waits for their loads HERE (vmcnt(0): the eight weight stages requested
    // above land under the same wait): left to the first MFMA that reads a register, it put a `vmcnt(0)` - which also waits for
    // every weight stage requested since - in front of each stage's first MFMA
#pragma unroll
    for (int s = 0; s < 16; ++s) asm volatile("" ::"v"(xh[s]), "v"(xl[s]));
    if constexpr (DIAG == 4 || DIAG == 6) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { qh[ks] = xh[ks]; ql[ks] = xl[4 + ks]; }
        if constexpr (DIAG == 6) {
            float sacc = 0.f;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) sacc += (float)xh[ks][0] + (float)xl[ks][1];
            if (qrow < F && active) p.out[(row0 + qrow) * D + h * DH + h2] = sacc;
            return;
        }
    }
    if constexpr (DIAG != 4) static_for<QA_NSTAGE>([&](auto gc) {
        constexpr int g = decltype(gc)::value, T = g >> 1, khalf = g & 1;
        // stage g has landed: stages 0 .. 7 before the loop (with the x rows), a later one when the four waves that requested it have
        // waited for it - their ONLY requests in flight (see `issue`)
        if constexpr (g == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if constexpr (g >= 8 && DIAG != 1) {
            if ((wave >> 2) == (g & 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // lgkmcnt: this wave's LDS writes (bias table, image rows)
        } else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                   // every wave's have; and every wave is done with stage g - 1
        asm volatile("" ::: "memory");
        if constexpr (g >= 6 && g + 2 < QA_NSTAGE) issue(IntC<g + 2>{});
        if (active) {
            if constexpr (khalf == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            }
            // The weight fragments are read by asm (`ds_read_b128`, counted `lgkmcnt` waits tied to the registers): a plain LDS load
            // makes the compiler wait for EVERY LDS-DMA in flight first (`vmcnt(0)`: the DMA writes LDS, it may alias) - which
            // undid the look-ahead - and it read all sixteen fragments of a stage through one register quad, a full LDS round trip
            // in front of every one or two MFMAs.  Two k-steps of fragments in registers, k-step s + 1 requested before the MFMAs of s.
            u32x4_q fh[2], fl[2];
            frag_read<slab_off<g>(0)>(fh[0], lds_lane);
            frag_read<slab_off<g>(1)>(fl[0], lds_lane);
            static_for<8>([&](auto sc) {
                constexpr int s = decltype(sc)::value, cur = s & 1;
                if constexpr (s + 1 < 8) {
                    frag_read<slab_off<g>(2 * s + 2)>(fh[cur ^ 1], lds_lane);
                    frag_read<slab_off<g>(2 * s + 3)>(fl[cur ^ 1], lds_lane);
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fh[cur]), "+v"(fl[cur]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fh[cur]), "+v"(fl[cur]));
                }
                __builtin_amdgcn_sched_barrier(0);
                const s16x8 wh = __builtin_bit_cast(s16x8, fh[cur]), wl = __builtin_bit_cast(s16x8, fl[cur]);
                const s16x8 bh = xh[8 * khalf + s], bl = xl[8 * khalf + s];
                if constexpr (DIAG == 3) { acc[s] += (float)wh[0] + (float)wl[1] + (float)bh[2] + (float)bl[3]; }
                else if constexpr (T < 4) {        // q^T / k^T tile: [32 d x 32 rows] += W . x^T
                    acc = MFMA32_S16(wl, bh, acc, 0, 0, 0);
                    acc = MFMA32_S16(wh, bl, acc, 0, 0, 0);
                    acc = MFMA32_S16(wh, bh, acc, 0, 0, 0);
                } else {                           // v tile: [32 rows x 32 d] += x . W^T
                    acc = MFMA32_S16(bl, wh, acc, 0, 0, 0);
                    acc = MFMA32_S16(bh, wl, acc, 0, 0, 0);
                    acc = MFMA32_S16(bh, wh, acc, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        if constexpr (g == QA_NSTAGE - 1) {       // stage 11's slabs sit where its own epilogue writes: every wave must be done reading
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        if (active) {
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
                        s16x8 hi, lo;
                        split8v(v, hi, lo);
                        const int ks = 2 * Tk + j;
                        const int off = qrow * DH + (((2 * ks + h2) ^ ((qrow >> 1) & 7)) << 3);
                        *reinterpret_cast<s16x8*>(Kh + off) = hi;
                        *reinterpret_cast<s16x8*>(Kl + off) = lo;
                    }
                } else {                           // v: lane = d, four consecutive keys per register quad
                    constexpr int Tv = T - 4;
                    const int d = 32 * Tv + q;
                    const float bv = bsm[2 * DH + d];
#pragma unroll
                    for (int grp = 0; grp < 4; ++grp) {
                        s16x4 hi, lo;
                        split4(acc[4 * grp] + bv, acc[4 * grp + 1] + bv, acc[4 * grp + 2] + bv, acc[4 * grp + 3] + bv, hi, lo);
                        const int key0 = 32 * qt + 8 * grp + 4 * h2;
                        *reinterpret_cast<s16x4*>(Vth + d * QA_VLD + key0) = hi;
                        *reinterpret_cast<s16x4*>(Vtl + d * QA_VLD + key0) = lo;
                    }
                }
            }
        }
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                    // the K / V images are complete
    if (!active) return;
    if constexpr (DIAG == 2) {
        float sacc = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) sacc += (float)qh[ks][0] + (float)ql[ks][1];
        if (qrow < F) p.out[(row0 + qrow) * D + h * DH + h2] = sacc + (float)Kh[qrow * DH] + (float)Vth[q * QA_VLD];
        return;
    }

    // ---- scores, softmax, output: the arithmetic of self_attn_split_kernel (attention.hip).  The K / V fragments are read by asm
    // one step ahead of the MFMAs that use them (the compiler read every fragment through one register quad and waited for it in
    // front of each MFMA group: ~280 exposed LDS round trips per wave, most of this phase).  A read past the last valid key tile
    // fetches LDS bytes nobody uses.
    f32x16 sT[QA_NKT];
    float m = -INFINITY;
    u32x4_q fa[2], fb[2];                          // [step parity]: K hi / lo, then V^T hi / lo
    {
        const int sw = (q >> 1) & 7;
        unsigned ka[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ka[ks] = lds_addr(Kh) + q * (DH * 2) + ((((2 * ks + h2) ^ sw)) << 4);
        constexpr int KLO = QA_FMAX * DH * 2;      // the lo plane
        frag_read<0>(fa[0], ka[0]);
        frag_read<KLO>(fb[0], ka[0]);
        static_for<QA_NKT>([&](auto ktc) {
            constexpr int kt = decltype(ktc)::value;
            if (kt < nkt) {
                f32x16 a;
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = 0.f;
                static_for<4>([&](auto ksc) {
                    constexpr int ks = decltype(ksc)::value, n = 4 * kt + ks, cur = n & 1;
                    constexpr int kt2 = (n + 1) >> 2, ks2 = (n + 1) & 3;        // the step after this one (kt2 == 7: bytes behind the image)
                    frag_read<kt2 * 32 * DH * 2>(fa[cur ^ 1], ka[ks2]);
                    frag_read<kt2 * 32 * DH * 2 + KLO>(fb[cur ^ 1], ka[ks2]);
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[cur]), "+v"(fb[cur]));
                    __builtin_amdgcn_sched_barrier(0);
                    const s16x8 kh = __builtin_bit_cast(s16x8, fa[cur]), kl = __builtin_bit_cast(s16x8, fb[cur]);
                    a = MFMA32_S16(kl, qh[ks], a, 0, 0, 0);
                    a = MFMA32_S16(kh, ql[ks], a, 0, 0, 0);
                    a = MFMA32_S16(kh, qh[ks], a, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                });
                if (kb[kt] == 0xFFFFFFFFu) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) m = fmaxf(m, a[i]);
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int kin = (i & 3) + 8 * (i >> 2) + 4 * h2;
                        const float sc = ((kb[kt] >> kin) & 1u) ? a[i] : -INFINITY;
                        a[i] = sc;
                        m = fmaxf(m, sc);
                    }
                }
                sT[kt] = a;
            }
        });
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0]), "+v"(fb[0]), "+v"(fa[1]), "+v"(fb[1]));     // the read behind the last step
    }
    if (p.split_out) __syncthreads();              // every wave still running has read its keys: the K image becomes the output staging area
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
    {
        // V^T fragment of (key tile kt, half mm, d tile t): keys k1 .. k1 + 3 and k1 + 8 .. k1 + 11 (k1 = 32 kt + 16 mm + 4 h2) of row
        // d = 32 t + q: one ds_read2_b64 per plane
        constexpr int VLO = DH * QA_VLD * 2;
        unsigned va[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) va[t] = lds_addr(Vth) + ((32 * t + q) * QA_VLD + 4 * h2) * 2;
        const unsigned vl0 = va[0] + VLO, vl1 = va[1] + VLO;
        v_read2<0>(fa[0], va[0]);
        v_read2<0>(fb[0], vl0);
        static_for<QA_NKT>([&](auto ktc) {
            constexpr int kt = decltype(ktc)::value;
            if (kt < nkt) {
                static_for<2>([&](auto mc) {
                    constexpr int mm = decltype(mc)::value;
                    float pv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) pv[j] = sT[kt][8 * mm + j];
                    s16x8 ph, pl;
                    split8v(pv, ph, pl);
                    static_for<2>([&](auto tc) {
                        constexpr int t = decltype(tc)::value, n = (2 * kt + mm) * 2 + t, cur = n & 1;
                        constexpr int n2 = n + 1, t2 = n2 & 1, km2 = n2 >> 1;       // next step: d tile t2 of (kt, mm) pair km2
                        v_read2<4 * km2>(fa[cur ^ 1], t2 == 0 ? va[0] : va[1]);
                        v_read2<4 * km2>(fb[cur ^ 1], t2 == 0 ? vl0 : vl1);
                        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[cur]), "+v"(fb[cur]));
                        __builtin_amdgcn_sched_barrier(0);
                        const s16x8 vh = __builtin_bit_cast(s16x8, fa[cur]), vl = __builtin_bit_cast(s16x8, fb[cur]);
                        f32x16& o = t == 0 ? o0 : o1;
                        o = MFMA32_S16(vl, ph, o, 0, 0, 0);
                        o = MFMA32_S16(vh, pl, o, 0, 0, 0);
                        o = MFMA32_S16(vh, ph, o, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
            }
        });
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0]), "+v"(fb[0]), "+v"(fa[1]), "+v"(fb[1]));
    }

    const float inv = 1.f / l;
    if (p.split_out) {
        // S-format result: this head's 64 columns of a row are ONE 256-byte block of the row (128 B hi | 128 B lo).  Written from the
        // accumulator layout that is sixteen 8-byte stores per lane, each instruction 64 pieces of 32 rows (~150 cycles of issue each:
        // with the stores removed the phase was 27 us per layer shorter).  Instead the tile goes through the K image - dead once every
        // wave has its scores (the barrier) - and leaves as whole 256-byte blocks, four rows per store instruction.
        // LDS image of a wave's tile: row r at 256 r, 16-byte chunk c (0 .. 7 hi, 8 .. 15 lo) at (c & 8) | ((c ^ r) & 7).
        char* const stg = lds + wave * 8192;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                s16x4 hi, lo;
                {
                    const f32x16& o = t == 0 ? o0 : o1;
                    split4(o[4 * rg] * inv, o[4 * rg + 1] * inv, o[4 * rg + 2] * inv, o[4 * rg + 3] * inv, hi, lo);
                }
                const int d = 32 * t + 8 * rg + 4 * h2;                     // four consecutive columns d .. d + 3 of row q
                char* at = stg + q * 256 + ((((d >> 3) ^ q) & 7) << 4) + (d & 4) * 2;
                *reinterpret_cast<s16x4*>(at) = hi;
                *reinterpret_cast<s16x4*>(at + 128) = lo;
            }
        // (same wave writes and reads: program order; the compiler counts its own LDS operations)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = 4 * i + (lane >> 4), c = lane & 15;
            const f32x4 blk = *reinterpret_cast<const f32x4*>(stg + r * 256 + (c & 8) * 16 + (((c ^ r) & 7) << 4));
            if (qt * 32 + r < F) st4(p.out + (row0 + qt * 32 + r) * D + h * DH + 4 * c, blk);
        }
    } else if (qrow < F) {
        float* rowp = p.out + (row0 + qrow) * D;
        const int c0 = h * DH + 4 * h2;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            f32x4 v0, v1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] = o0[4 * rg + e] * inv; v1[e] = o1[4 * rg + e] * inv; }
            st4(rowp + c0 + 8 * rg, v0);
            st4(rowp + c0 + 32 + 8 * rg, v1);
        }
    }
}

// per-device kernel attributes: once, under a mutex, outside any stream capture (dec_mlp_prepare)
int dec_qkv_attn_prepare() {
    static std::mutex mu;
    static bool attr_set[64] = {};
    int dev = 0;
    LADIFF_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return LADIFF_ERR_ARG;
    std::lock_guard<std::mutex> lock(mu);
    if (!attr_set[dev]) {
        LADIFF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dec_qkv_attn_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, QA_LDS));
#ifdef LADIFF_STAMPS
        const void* k[6] = {reinterpret_cast<const void*>(dec_qkv_attn_kernel<1>), reinterpret_cast<const void*>(dec_qkv_attn_kernel<2>),
                            reinterpret_cast<const void*>(dec_qkv_attn_kernel<3>), reinterpret_cast<const void*>(dec_qkv_attn_kernel<4>),
                            reinterpret_cast<const void*>(dec_qkv_attn_kernel<5>), reinterpret_cast<const void*>(dec_qkv_attn_kernel<6>)};
        for (int i = 0; i < 6; ++i) LADIFF_HIP(hipFuncSetAttribute(k[i], hipFuncAttributeMaxDynamicSharedMemorySize, QA_LDS));
#endif
        attr_set[dev] = true;
    }
    return 0;
}

// out [M,256] = self-attention(x W_in^T + b_in) per sample over its frames, keys >= lengths[b] masked; xs / w S-format
int launch_dec_qkv_attn(const float* xs, const float* w, const float* bias, const int32_t* lengths, const int32_t* row_off, float* out,
                        int B, int F, int split_out, hipStream_t s) {
    if (F > QA_FMAX || F < 1) return LADIFF_ERR_SHAPE;
    if (B == 0) return 0;
    LADIFF_TRY(dec_qkv_attn_prepare());
    QkvAttnArgs a{xs, w, bias, lengths, row_off, out, B, F, split_out};
    void* args[] = {&a};
    const void* kern = reinterpret_cast<const void*>(dec_qkv_attn_kernel<0>);
#ifdef LADIFF_STAMPS
    // diagnostic twin only: timing builds with garbage results, ladiff_debug_set_mlp_variant(21 .. 26)
    const void* k[7] = {reinterpret_cast<const void*>(dec_qkv_attn_kernel<0>), reinterpret_cast<const void*>(dec_qkv_attn_kernel<1>),
                        reinterpret_cast<const void*>(dec_qkv_attn_kernel<2>), reinterpret_cast<const void*>(dec_qkv_attn_kernel<3>),
                        reinterpret_cast<const void*>(dec_qkv_attn_kernel<4>), reinterpret_cast<const void*>(dec_qkv_attn_kernel<5>),
                        reinterpret_cast<const void*>(dec_qkv_attn_kernel<6>)};
    const int v = g_mlp_variant;
    kern = k[v >= 21 && v <= 26 ? v - 20 : 0];
#endif
    LADIFF_HIP(hipLaunchKernel(kern, dim3(8 * H * ((B + 7) / 8)), dim3(512), args, QA_LDS, s));
    LADIFF_LAUNCH_CHECK();
    return 0;
}

}  // namespace ladiff
