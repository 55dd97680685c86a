// The denoiser loop as ONE persistent, weight-stationary pipeline over the whole chip (f16x3 mode, guidance on).
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
// Numerics: the same arithmetic as the launch-per-stage f16x3 path (S-format operands, hi*hi + hi*lo + lo*hi on
// v_mfma_f32_16x16x32_bf16, fp32 accumulation and fp32 everything else); only the summation order of the split products
// differs (8 hidden slices instead of 4 K-slices).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "model.h"
#include "tile_mma.h"

namespace ladiff {

std::atomic<int> g_stage_plan{0};   // measurement switch: ladiff_debug_set_stage_plan (red_plan below)
std::atomic<int> g_poll_pause{0};   // measurement switch: ladiff_debug_set_poll_pause (mask | len << 8)
std::atomic<int> g_stage_delay{-1}; // ladiff_debug_set_stage_delay (mask | len << 8); -1: by block count (launch_systolic_loop)
std::atomic<int> g_look_ahead_from{-1}, g_small_upto{-1};   // ladiff_debug_set_loop_thresholds (-1: the built-in block counts)
std::atomic<int> g_pace{-1};   // ladiff_debug_set_pacing (eighths | mask << 8); -1: by block count - STYL sleeps half of its last observed wait before polling in launches above SMALL_LAUNCH_BLOCKS (measured: -3 % at 128 / 256 prompts, round 4; -1.8 / -2.9 % at the round-6 kernels), nobody below (there pacing costs 0.4 %: profiles/r6/12_*)

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef LADIFF_PF1
#define LADIFF_PF1 2
#endif
#ifndef LADIFF_PF2
#define LADIFF_PF2 1
#endif
constexpr int PF1 = LADIFF_PF1;           // fragment fetch distance of the one-column-tile products (tile_mma.h, mma): LIN / FFN first product, QKV's third tile
constexpr int PF2 = LADIFF_PF2;           // ... of the two-tile products of the eight-wave roles
constexpr int NSLICE = 8;                 // hidden slices of the two MLPs (128 columns each)
constexpr int HS = FF / NSLICE;           // 128
constexpr int NRED = 3;                   // workgroups per layer of each of the two reduce stages (how they share the work: red_parts())
constexpr int NTAIL = 4;                  // tail workgroups (block b belongs to tail b % NTAIL)
constexpr int FLAG_SLOTS = 16;
// The eight partial planes of a layer's two K-split matrices (LIN -> RED2, FFN -> STYL) are RINGS of PRING block slots, not NB:
// with a buffer set per layer (sys_layout) they would otherwise be most of a working set larger than the 256 MiB memory-side
// cache.  A producer may therefore run at most PRING blocks ahead of its consumer: every PRING / 2 blocks it waits for the
// consumer's flag of the block PRING / 2 back (MlpRole::backpressure; the stages visit their blocks in order).  The slot of
// block b of local step s is (s NB + b) % PRING: blocks are counted THROUGH the steps, so that the reuse distance is PRING
// blocks at the wrap from one step to the next as well.
constexpr int PRING = 16;
constexpr int SMALL_LAUNCH_BLOCKS = 60;     // launches up to this many blocks: LIN / FFN rest after every block (launch_systolic_loop)
constexpr int PACED_ROLES = 4;              // stage types (R::PAUSE_BIT) whose polling is paced (tag_loop): STYL
constexpr int FLAG_STRIDE = 32;             // words between the flags of two producers: every flag on a 128-byte line of its own
constexpr int SYS_LDS_BYTES = 100 * 1024;   // > 80 KiB: one workgroup per CU, so the <= 256 workgroups sit on distinct CUs
constexpr int GROUPS_PER_LAYER = 7;
enum Group : int { G_XIN = 0, G_ATT = 1, G_X1 = 2, G_PC = 3, G_X2 = 4, G_PE = 5, G_XO = 6 };
enum Role : int { R_QKV = 0, R_OUT = 1, R_LIN = 2, R_RED2 = 3, R_FFN = 4, R_STYL = 5, R_SKIP = 6, R_TAIL = 7 };

struct Stage {                            // one per workgroup
    int role, layer, slice, act;
    int wait_group, wait_n, out_group, out_slot;
    int blk0, blkstride;                  // the blocks this workgroup visits: blk0, blk0 + blkstride, ...
    // A flag with many consumer workgroups is REPLICATED, one 128-byte line per consumer: the producer raises out_rep flags (slots
    // out_slot + i out_rep_stride) with one store instruction, a consumer polls wait_n slots from wait_slot0.  Sixty-four waves
    // polling one line made every poll of that line slow - and those were the inputs of the two busiest stage types (LIN, FFN).
    int wait_slot0, out_rep, out_rep_stride;
    // XCD placement (sys_place_stages): workgroup i runs on XCD i % 8 and each XCD has an L2 of its own.  out_local = every reader
    // of this stage's output (rows and flags) sits on the SAME XCD: the stage then stores plainly - the rows stay in that L2, the
    // store is acknowledged by the L2 instead of the memory side (a hop of 0.5 us instead of 0.9 - 1.2 us,
    // scripts/ubench_xcd_handoff.hip) - otherwise it writes through (sc1).  Loads are sc1 either way: they are served by the
    // reader's L2 when the line is there.  xcd = the XCD the plan put this workgroup on (-1: not checked).
    int out_local, xcd, pad0;
    int bp_group, bp_slot0, bp_n, bp_blocks;   // LIN / FFN: the consumer's flags (group, first slot, count) and how many consecutive blocks make "all consumers"
    const float *w0, *w1;                 // S-format matrices
    const float *b0, *b1;                 // biases
    const float *g, *be;                  // LayerNorm gamma / beta
    const float *in0, *in1, *in2;         // block-layout activations: [NB][RT][256] (partials: [8][NB][RT][256])
    float* out;
    const float* bp_buf;                  // LIN / FFN, tagged hand-off: the consumer's OUTPUT rows (their tags are the ring's back-pressure)
};

// Geometry of one block, built on the host (sys_pack_blocks).  32-row tiles: both guidance branches of P prompts, T rows each
// (the latent count masks keys).  16-row tiles: ONE guidance branch of as many prompts as fit with only their count[b] valid
// latent rows (length-aware: padded latent rows never influence valid ones - they are masked as keys, every other op is
// per row, and ladiff.py:559-566 zeroes them at the end - so they are not computed at all).
struct BlockDesc {
    int nrows, nsb, pad0, pad1;
    int b2[16];                           // per sample-branch sx: text-cache row (-1: absent)
    // attention stage, per tile row: sx | first tile row of sx << 8 | valid latent keys << 16 (0xff: counts[] at run time); the
    // row's cross-attention / counts row (-1: padding)
    int row_pk[32], row_b2[32];
    int row_lat[32], row_t[32];           // latents row (prompt * T + t, -1: padding) and latent index of a tile row
    // reduce stages: part q of NRED handles slot k = wave + 4 i -> tile row | latent index << 8 | latent count << 16 (0xff: run
    // time), -1: no row; and the row's cross-attention table row.  The parts cover ALL rows of the tile: a row past nrows
    // (PART_PAD | row) is stored as zeros - under the tagged hand-off every row of a tile carries the step's parity, so that a
    // consumer checks whole tiles and needs no geometry.
    int part_pk[3][12], part_b2[3][12];
    // tail: (prompt, latent) pair k = wave + 4 i -> latents row (-1: none), latent index, tile row of the conditional branch.
    // A slot without a pair zeroes two padding rows instead: row pair_pad of the unconditional block and row pair_rc of the
    // conditional one (-1: nothing left to pad)
    int pair_lat[16], pair_t[16], pair_rc[16], pair_pad[16];
};
constexpr int PART_PAD = 0x40000000;
static_assert(NRED == 3, "BlockDesc::part_pk");
// How the reduce workgroups of a layer share a block's rows.  32-row blocks: NRED row parts each (<= 11 rows, 3 per wave).
// 16-row blocks: a part can take 8 rows (2 per wave), so RED2 needs only two workgroups and the freed one goes to STYL, the
// busiest stage of that plan: two GROUPS of two parts, group g visiting the blocks b = g (mod 2) - it sees every other block.
struct RedPlan { int red2_parts, styl_parts, styl_groups, out_groups; };
// how a layer's workgroups that hold no MLP slice are dealt (16-row blocks): variant 0 = one OUT workgroup, STYL as two groups on
// alternating blocks x two row parts; variant 1 = OUT as two groups on alternating blocks, STYL as one group x two row parts
inline RedPlan red_plan(int MR) {
    if (MR != 1) return RedPlan{NRED, NRED, 1, 1};
    return g_stage_plan.load() == 1 ? RedPlan{2, 2, 1, 2} : RedPlan{2, 2, 2, 1};
}

struct SysArgs {
    const Stage* stages;
    const BlockDesc* blocks;              // [NB]
    unsigned* flags;                      // [groups][NB][FLAG_SLOTS]
    unsigned* status;                     // [0] abort code (0 = ok), [1] diagnostic
    const float* tables;                  // time tables [n_total][9][1536]
    const float* tkv;                     // text K|V [9][2B][512]
    const float* ctab;                    // hoisted cross-attention [9][n_chunk][2B+1][256]
    const float* coef; const float* noise; const float* pe; const float* ng; const float* nb;
    float* lat;                           // latents [B][T][256]
    const int32_t* counts;
    float gscale;
    int B, B2, T, P, NB, step_lo, n_steps, n_ctab;          // B2 = sample-branches: 2 B with guidance (uncond | cond), B without
    int force_mismatch;                   // test aid: one workgroup reports a placement that disagrees (ladiff_debug_set_xcd_local(2))
    int split;                            // 1: a block holds ONE guidance branch of its P prompts (block 2g + br), 0: both
    int fault_wg;                         // test aid: this workgroup leaves right after the start-up handshake and never publishes (-1: none)
    unsigned long long timeout_ticks;     // bound of every spin, in s_memrealtime ticks (100 MHz)
    int look_ahead;                       // tagged hand-off: stages may request the next block's rows early (see `settle`)
    NoiseGen gen;                         // on: the TAIL stage draws the per-step noise itself (noise_gen.h) instead of reading `noise`
    int probe;                            // diagnostic twin build only: timing probes with garbage results (ladiff_debug_set_probe)
    int pace;                             // low byte: eighths of its last observed wait a tag_loop stage sleeps before the first poll of a block; bits 8 ..: the stage types (R::PAUSE_BIT) that do
    int delay_mask, delay_len;            // measurement: stage types (R::PAUSE_BIT) that idle delay_len x ~60 ns after every block (pacing experiment)
    int pause_mask, pause_len;            // stage types (R::PAUSE_BIT) that rest pause_len x ~60 ns between two polls of rows that are not there yet
    unsigned long long* stamps;           // diagnostic twin build only (-DLADIFF_STAMPS): [workgroup][step][block][8] realtime ticks
};

// In-kernel timeline (diagnostic twin build; the shipped library executes no stamp): s_memrealtime is one 100 MHz counter
// for the whole chip, so stamps of different workgroups order the hand-offs between CUs.
#if defined(LADIFF_STAMPS) && LADIFF_STAMPS + 0 >= 2   // per-block timeline: -DLADIFF_STAMPS=2 (each stamp costs ~0.1 us: it distorts the busy / blocked totals)
#define SYS_STAMP(i)                                                                                                   \
    do {                                                                                                               \
        if (p.stamps != nullptr && threadIdx.x == 0) {                                                                 \
            if (s < 4 && b < 4) p.stamps[(((size_t)blockIdx.x * 4 + s) * 4 + b) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
            const int sm_ = p.n_steps / 2, bm_ = p.NB / 2;       /* steady state: four consecutive blocks mid-run */         \
            if (s == sm_ && b >= bm_ && b < bm_ + 4)                                                                   \
                p.stamps[(size_t)256 * 4 * 4 * 8 + 256 * 4 + ((size_t)blockIdx.x * 4 + (b - bm_)) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
        }                                                                                                              \
    } while (0)
// the same from the first loader thread of a wave-group loop (threads 256 ..)
#define SYS_STAMP_L(i)                                                                                                 \
    do {                                                                                                               \
        if (p.stamps != nullptr && threadIdx.x == 256) {                                                               \
            if (s < 4 && b < 4) p.stamps[(((size_t)blockIdx.x * 4 + s) * 4 + b) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
        }                                                                                                              \
    } while (0)
#elif defined(LADIFF_STAMPS)
#define SYS_STAMP(i) do { } while (0)
#define SYS_STAMP_L(i) do { } while (0)
#endif
#ifdef LADIFF_STAMPS
// per-workgroup totals behind the timeline: ticks blocked in wait_epoch, prefetch hits, blocks processed
#define SYS_STAT_DECL unsigned long long st_wait = 0, st_hit = 0, st_n = 0, st_t = 0
#define SYS_STAT_T0 st_t = __builtin_amdgcn_s_memrealtime()
#define SYS_STAT_WAIT st_wait += __builtin_amdgcn_s_memrealtime() - st_t
#define SYS_STAT_ITER(h) do { st_hit += (h) ? 1 : 0; st_n += 1; } while (0)
#define SYS_STAT_END                                                                                                   \
    do {                                                                                                               \
        if (p.stamps != nullptr && threadIdx.x == 0) {                                                                 \
            unsigned long long* o = p.stamps + (size_t)256 * 4 * 4 * 8 + (size_t)blockIdx.x * 4;                       \
            o[0] = st_wait; o[1] = st_hit; o[2] = st_n;                                                                \
        }                                                                                                              \
    } while (0)
// wave-group loops (QKV, OUT): [0] ticks the loader waves were blocked on flags, [2] blocks, [3] ticks the other waves stood at the
// tile barrier waiting for the loaders
#define SYS_SPLIT_DECL unsigned long long sp_wait = 0, sp_idle = 0, sp_n = 0, sp_t = 0
#define SYS_SPLIT_T0 sp_t = __builtin_amdgcn_s_memrealtime()
#define SYS_SPLIT_WAIT sp_wait += __builtin_amdgcn_s_memrealtime() - sp_t
#define SYS_SPLIT_IDLE do { sp_idle += __builtin_amdgcn_s_memrealtime() - sp_t; sp_n += 1; } while (0)
#define SYS_SPLIT_END                                                                                                  \
    do {                                                                                                               \
        if (p.stamps != nullptr && (threadIdx.x == 0 || threadIdx.x == 256)) {                                         \
            unsigned long long* o = p.stamps + (size_t)256 * 4 * 4 * 8 + (size_t)blockIdx.x * 4;                       \
            if (threadIdx.x == 256) { o[0] = sp_wait; o[1] = 0; } else { o[2] = sp_n; o[3] = sp_idle; }                \
        }                                                                                                              \
    } while (0)
#else
#define SYS_SPLIT_DECL do { } while (0)
#define SYS_SPLIT_T0 do { } while (0)
#define SYS_SPLIT_WAIT do { } while (0)
#define SYS_SPLIT_IDLE do { } while (0)
#define SYS_SPLIT_END do { } while (0)
#define SYS_STAMP_L(i) do { } while (0)
#define SYS_STAMP(i) do { } while (0)
#define SYS_STAT_DECL do { } while (0)
#define SYS_STAT_T0 do { } while (0)
#define SYS_STAT_WAIT do { } while (0)
#define SYS_STAT_ITER(h) do { } while (0)
#define SYS_STAT_END do { } while (0)
#endif

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

// ---- tagged hand-off (template switch HO = 1; HO = 0 is the flag protocol above).  The DATA carries the epoch: the last
// mantissa bit of EVERY fp32 word of a handed-off row is replaced by a parity bit - parity of the local step for a buffer that is
// written once per step, parity of the slot's use count for the rings of partial planes.  A consumer loads the rows it needs
// (sc1, 16 bytes per lane as before), looks at the parity bit of every word it loaded and loads again until all of them show
// the parity it expects; then it clears the bits and computes.  What this removes from every hop: the producer's drain
// (s_waitcnt vmcnt(0)), its barrier and flag store, and the consumer's separate poll round trip - the poll IS the load.
// What it rests on: an aligned 4-byte word is written whole (each word carries its own bit, so nothing is assumed about the
// 16-byte store being seen as one unit); a slot is written exactly once per use and not again before every reader of that use
// has loaded it (the dependency chain through the tail stage for the per-step buffers, MlpRole::backpressure_tag for the
// rings); AND, on the reader's side, nobody ever looks at a slot that is more than ONE use old - one bit cannot tell the use before
// last from the one awaited.  Every stage with a barrier and every live row has that by construction (it needed the previous use
// to get here); a wave of a barrier-free stage whose row is padding did not until round 6 (Red2Role::issue: it now waits for its
// ring's producers as a live row does); the hand-off buffers are set to parity 1 before every launch (bytes 0x01) and the first use expects parity 0.
// The value a consumer computes with has the bit CLEARED, whatever the parity was: results do not depend on the block plan, and
// the flag protocol stores and clears the same way (parity 0), so the two protocols give the same bits.
#ifndef LADIFF_TAG_BITS
#define LADIFF_TAG_BITS 1                   // diagnostic builds: a wider tag (step / ring use modulo 2^bits) tells an A-B-A apart from a lost write
#endif
constexpr unsigned TAG_MASK = (1u << LADIFF_TAG_BITS) - 1u;
__device__ __forceinline__ f32x4 tag4(const f32x4 v, unsigned par) {
    return __builtin_bit_cast(f32x4, (__builtin_bit_cast(u32x4, v) & ~TAG_MASK) | par);
}
__device__ __forceinline__ f32x4 untag4(const f32x4 v) { return __builtin_bit_cast(f32x4, __builtin_bit_cast(u32x4, v) & ~TAG_MASK); }
// bad |= (parity bit of any of the four words) != par   (only bit 0 of `bad` means anything)
__device__ __forceinline__ void tag_acc(unsigned& bad, const f32x4 v, unsigned par) {
    const u32x4 w = __builtin_bit_cast(u32x4, v) ^ par;
    bad |= (w[0] | w[1]) | (w[2] | w[3]);
}
__device__ __forceinline__ unsigned ring_use_par(const SysArgs& p, int s, int b) { return (unsigned)(((s * p.NB + b) / PRING) & TAG_MASK); }

// a stage's output rows: plain when every reader shares this XCD's L2, write-through otherwise; par = the parity tag (0 under
// the flag protocol).  HO as the roles' template argument: 0 = flag protocol (the store form is picked at run time), 1 / 2 = tagged
// hand-off with write-through / plain stores - there the form is a compile-time property of the role's loop, so that the loop
// body is straight-line code and the compiler can COUNT the stores behind the next block's loads (`s_waitcnt vmcnt(n)`) instead of
// draining them (`vmcnt(0)`) where those loads are first used.
// Which roles run their products transposed (tile_mma.h, mma<..., TR>): compile-time switches so that a variant build can measure each
// (scripts/build_variant.sh <name> -DLADIFF_TR_OUT=0 ...); same bits either way
#ifndef LADIFF_TR_OUT
#define LADIFF_TR_OUT 1
#endif
#ifndef LADIFF_TR_STYL
#define LADIFF_TR_STYL 1
#endif
#ifndef LADIFF_TR_SKIP
#define LADIFF_TR_SKIP 1
#endif
#ifndef LADIFF_TR_QKV
#define LADIFF_TR_QKV 1
#endif
template <bool TR, int MR, int NT, class ColOf>
__device__ __forceinline__ void stage_c_sel(float* ct, const f32x4 (&acc)[MR][NT], ColOf col_of) {
    if constexpr (TR) stage_c_t(ct, acc, col_of); else stage_c(ct, acc, col_of);
}

template <int HO>
__device__ __forceinline__ void st_out(const Stage& st, __amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v, unsigned par) {
    v = tag4(v, par);
    if constexpr (HO == 2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, 0);
    else if constexpr (HO == 1) st_sc1(r, off, v);
    else {
        if (st.out_local) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, 0);
        else st_sc1(r, off, v);
    }
}

constexpr unsigned long long TIMEOUT_TICKS = 150000000ull;     // default bound of a wait: s_memrealtime runs at 100 MHz, 1.5 s (SysArgs::timeout_ticks)

// LDS words of the hand-off protocol (last 16 bytes of the dynamic region)
struct Ctl { int abort, ready; unsigned arrive; int local_ok; };      // arrive: attention waves of the QKV stage that have drained, over all blocks

// Every wave polls flags[0 .. n) by itself until all are >= epoch and goes on to its own loads at once (no barrier, no LDS
// round trip after the flag is seen).  ONE poll in flight per wave: the memory side serves flags and data alike, and two polls in
// flight per wave made every hand-off of the chip slower (4 % on the loop), four much slower; relaying wave 0's poll to the other
// waves through LDS was no faster than letting them poll.  Returns false on abort; waves that leave end the kernel, and a
// barrier only counts the waves still running.
// `first`: a sample of this lane's flag taken earlier (stage_loop polls the next block's flags while the current block's stores
// drain), 0 = none: when the flags were already up then, the wait costs no round trip at all.
__device__ __forceinline__ bool wait_epoch(const unsigned* flags, int n, unsigned epoch, unsigned* status, Ctl*, unsigned first,
                                           unsigned long long timeout) {
    const int lane = threadIdx.x & 63;
    if (__all((lane < n ? first : epoch) >= epoch)) return true;
    const gu32* f = (const gu32*)flags + (lane < n ? lane : 0) * FLAG_STRIDE;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned spins = 1;; ++spins) {
        const unsigned v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(v >= epoch)) return true;
        if ((spins & 63u) == 0u) {
            const unsigned a = __hip_atomic_load((const gu32*)status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a != 0u) return false;
            if (__builtin_amdgcn_s_memrealtime() - t0 > timeout) {
                if (lane == 0) {
                    __hip_atomic_store((gu32*)status + 1, (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store((gu32*)status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// all rows of this workgroup are stored: drain (every wave), meet, publish.  YOUNGER = vector-memory operations the wave has issued
// after its stores (vmcnt counts in issue order: they are left in flight)
// raise the out_rep replicas of a stage's flag (flag = replica 0): lanes of ONE wave call this
__device__ __forceinline__ void raise(const Stage& st, unsigned* flag, unsigned epoch) {
    const int lane = threadIdx.x & 63;
    if (lane < st.out_rep) {
        if (st.out_local) __builtin_amdgcn_raw_buffer_store_b32(epoch, rsrc_of(flag), (unsigned)(lane * st.out_rep_stride * FLAG_STRIDE * 4), 0, 0);
        else __hip_atomic_store((gu32*)flag + lane * st.out_rep_stride * FLAG_STRIDE, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <int YOUNGER = 0>
__device__ __forceinline__ void publish(const Stage& st, unsigned* flag, unsigned epoch) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
    __syncthreads();
    if (threadIdx.x < 64) raise(st, flag, epoch);
}

__device__ __forceinline__ unsigned* flag_of(const SysArgs& p, int group, int b, int slot) {      // group = layer * 7 + Group
    return p.flags + ((size_t)(group * p.NB + b) * FLAG_SLOTS + slot) * FLAG_STRIDE;
}

// ---------------------------------------------------------------- LDS images (tile_mma.h) of hand-off rows
// A [16 MR][256] fp32 block of rows held in registers between its loads and its LDS image: thread t of the 256 WS threads owns
// the 8-column units id = t + 256 WS u (row id / 32, columns 8 (id % 32) ..).  WS = waves per SIMD of the stage workgroup.
template <int MR, int WS> struct Rows256 { f32x4 v[2 * MR / WS][2]; };
template <int MR, int WS>
__device__ __forceinline__ void issue_rows(Rows256<MR, WS>& x, __amdgpu_buffer_rsrc_t r, unsigned base) {
#pragma unroll
    for (int u = 0; u < 2 * MR / WS; ++u) {
        const int id = threadIdx.x + 256 * WS * u, row = id >> 5, c8 = id & 31;
        x.v[u][0] = ld_sc1(r, base + row * 1024 + c8 * 32);
        x.v[u][1] = ld_sc1(r, base + row * 1024 + c8 * 32 + 16);
    }
}
template <int AR, int KB, int MR, int WS>
__device__ __forceinline__ void commit_rows(char* tile, int kb0, const Rows256<MR, WS>& x) {
#pragma unroll
    for (int u = 0; u < 2 * MR / WS; ++u) {
        const int id = threadIdx.x + 256 * WS * u, row = id >> 5, c8 = id & 31;
        const f32x4 v0 = untag4(x.v[u][0]), v1 = untag4(x.v[u][1]);
        if constexpr (AR == 0) {
            s16x8 hi, lo;
            split8(v0, v1, hi, lo);
            *reinterpret_cast<s16x8*>(a_slot<KB>(tile, row, kb0 + (c8 >> 3), c8 & 7)) = hi;
            *reinterpret_cast<s16x8*>(a_slot<KB>(tile, row, kb0 + (c8 >> 3), 8 + (c8 & 7))) = lo;
        } else {
            tile_put4<1, KB>(tile, row, kb0 * 64 + c8 * 8, v0);
            tile_put4<1, KB>(tile, row, kb0 * 64 + c8 * 8 + 4, v1);
        }
    }
}
// tagged hand-off: does every word of the tile show parity `par`?  (every row of a tile is written, padding rows as zeros)
template <int MR, int WS>
__device__ __forceinline__ void rows_bad(unsigned& bad, const Rows256<MR, WS>& x, unsigned par) {
#pragma unroll
    for (int u = 0; u < 2 * MR / WS; ++u) { tag_acc(bad, x.v[u][0], par); tag_acc(bad, x.v[u][1], par); }
}

// LayerNorm statistics of a 256-column row, two-pass, fp32 (as rowops.hip).  The per-lane parts are shared by the two layouts
// below so that both sum in the same order: a lane's four consecutive columns first, then 16 lanes (row16_sum), then the four
// 64-column quarters as (q3 + q2) + (q1 + q0) - what wave_sum does with its two row broadcasts.
// Every multiply-add here is spelled out (__fmul_rn / __fsub_rn / __fmaf_rn): left to -ffp-contract the compiler fuses `mean = s / 256`
// into the subtractions that use it in one inlining context and not in another, and the two layouts then differ in the last bit
// of a few deviations - which the f16x3 splits of nine layers turn into 1e-4 on the latents.
__device__ __forceinline__ float part_sum4(const f32x4 v) { return v[0] + v[1] + v[2] + v[3]; }
__device__ __forceinline__ float part_sq4(const f32x4 v, float mean) {
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float d = __fsub_rn(v[i], mean); q = __fmaf_rn(d, d, q); }
    return q;
}
__device__ __forceinline__ float ln_apply(float v, float mean, float rstd, float g, float b) {
    return __fmaf_rn(__fmul_rn(__fsub_rn(v, mean), rstd), g, b);
}
__device__ __forceinline__ void row_stats4(const f32x4 v, float& mean, float& rstd) {   // a row on 64 lanes: lane l holds columns 4 l ..
    const float s = wave_sum(part_sum4(v));
    mean = __fmul_rn(s, 1.f / 256.f);
    const float q = wave_sum(part_sq4(v, mean));
    rstd = rsqrtf(__fmaf_rn(q, 1.f / 256.f, LN_EPS));
}
// a row on 16 lanes (four rows per wave at once): lane l16 holds columns 64 k + 4 l16 .. of quarter k in v[k]; same bits as row_stats4
__device__ __forceinline__ void row_stats16(const f32x4 (&v)[4], float& mean, float& rstd) {
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = row16_sum(part_sum4(v[k]));
    mean = __fmul_rn((r[3] + r[2]) + (r[1] + r[0]), 1.f / 256.f);
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = row16_sum(part_sq4(v[k], mean));
    rstd = rsqrtf(__fmaf_rn((r[3] + r[2]) + (r[1] + r[0]), 1.f / 256.f, LN_EPS));
}
__device__ __forceinline__ f32x4 sum8(const f32x4 (&pl)[NSLICE]) {       // fixed summation tree of the eight hidden slices
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        v[i] = ((pl[0][i] + pl[1][i]) + (pl[2][i] + pl[3][i])) + ((pl[4][i] + pl[5][i]) + (pl[6][i] + pl[7][i]));
    return v;
}

// ---------------------------------------------------------------- the pipelined stage loop
// A stage visits its blocks in the order (step, block); block order is the same everywhere, so the dependency graph is a
// lattice and no stage can wait on something that waits on it.  While a block is being computed the NEXT block's flags are
// polled once and, if they are up, its hand-off loads are already in flight (registers `nxt`): a stage's time per block is
// then its compute + store time, not the whole store -> flag -> poll -> load chain.
//   R::Pay            register image of one block's inputs
//   r.issue(s, b, pay) every load of block (s, b)                r.commit(pay)  registers -> LDS operand images
//   r.compute(s, b, pay) the stage's arithmetic and its sc1 stores
// What a stage does at the FIRST barrier inside its compute phase (about a microsecond after the phase started): by
// then the previous block's write-through stores have drained for free, so its flag is published there, and only then are
// the next block's loads and the early poll issued (a `vmcnt(0)` placed after them would wait for them as well).
template <class R>
struct Mid {
    const SysArgs& p; const Stage& st; R& r; typename R::Pay& nxt; typename R::Geo& gnxt; typename R::Geo& gnn; Ctl* ctl;
    unsigned* pending; unsigned pending_epoch; bool has_next; int s2, b2, s3, b3;
    unsigned early;                       // wave 0, lanes < wait_n: the next block's flags, polled at the top of this iteration
    bool have;                            // out: the next block's loads have been issued
    __device__ __forceinline__ void before_barrier() {
        if (pending != nullptr) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (R::PREFETCH) {
            if (threadIdx.x < 64) {       // is the next block's input there?  (a poll about a microsecond old: it came back under this block's first phase)
                const int ok = has_next && __all(early >= (unsigned)(s2 + 1));
                if ((threadIdx.x & 63) == 0) ctl->ready = ok;
            }
        }
    }
    __device__ __forceinline__ void before_stores() {}
    __device__ __forceinline__ void after_barrier() {
        if (pending != nullptr) {
            if (threadIdx.x < 64) raise(st, pending, pending_epoch);
            pending = nullptr;
        }
        have = false;
        if constexpr (R::PREFETCH) have = ctl->ready != 0;
        // block geometry (descriptor words) is fetched TWO blocks ahead: what the next block's loads need arrived an
        // iteration ago, so nothing here waits on a load issued in this phase
        r.geo_fix(gnxt);
        if constexpr (R::PREFETCH) { if (have) r.issue(s2, b2, gnxt, nxt); }
        if (s3 < p.n_steps) r.geo(b3, gnn);
    }
};

template <class R>
__device__ __forceinline__ void stage_loop(const SysArgs& p, const Stage& st, R& r, Ctl* ctl, int b0, int bstride) {
    typename R::Pay cur, nxt;
    typename R::Geo gcur, gnxt, gnn;
    bool have = false;
    const int lane = threadIdx.x & 63;
    if (b0 < p.NB) {
        r.geo(b0, gcur);
        r.geo(b0 + bstride < p.NB ? b0 + bstride : b0, gnxt);
        r.geo_fix(gcur);
        gnn = gnxt;
    }
    unsigned* pending = nullptr;                                         // flag of the previous block, its stores still draining
    unsigned pending_epoch = 0;
    unsigned pre = 0u;                                                   // an early sample of the next wait's flags (0: none)
    // Whether the NEXT block can be prefetched is decided by a poll issued at the top of the iteration and read at the block's first
    // compute barrier (a flag load is a ~1 us round trip to the memory side: it comes back under the commit and the first MFMA
    // phase).  A miss costs nothing: the next iteration then waits for its flags the normal way.
    SYS_STAT_DECL;
    for (int s = 0; s < p.n_steps; ++s)
        for (int b = b0; b < p.NB; b += bstride) {
            SYS_STAMP(0);
            SYS_STAT_ITER(have);
            if constexpr (R::BACKP) { if (!r.backpressure(s, b, ctl)) return; }
            if (!have) {
                SYS_STAT_T0;
                if (!wait_epoch(flag_of(p, st.wait_group, b, st.wait_slot0), st.wait_n, s + 1, p.status, ctl, pre, p.timeout_ticks)) return;
                SYS_STAT_WAIT;
                r.issue(s, b, gcur, cur);
            }
            SYS_STAMP(1);
            int s2 = s, b2 = b + bstride;
            if (b2 >= p.NB) { s2 = s + 1; b2 = b0; }
            int s3 = s2, b3 = b2 + bstride;
            if (b3 >= p.NB) { s3 = s2 + 1; b3 = b0; }
            const bool has_next = R::PREFETCH && s2 < p.n_steps;
            unsigned early = 0xffffffffu;
            if (has_next && threadIdx.x < 64 && lane < st.wait_n)
                early = __hip_atomic_load((const gu32*)flag_of(p, st.wait_group, b2, st.wait_slot0) + lane * FLAG_STRIDE, __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT);
            r.commit(cur);
            __syncthreads();
            SYS_STAMP(2);
            Mid<R> mid{p, st, r, nxt, gnxt, gnn, ctl, pending, pending_epoch, has_next, s2, b2, s3, b3, early, false};
            r.compute(s, b, gcur, cur, mid);
            have = mid.have;
            gcur = gnxt; gnxt = gnn;
            SYS_STAMP(4);
            if (have) {
                // backlogged (the next block is already here): throughput counts, so this block's flag goes out at the next
                // block's first compute barrier instead of stalling the stage on the write-through drain now
                pending = flag_of(p, st.out_group, b, st.out_slot);
                pending_epoch = s + 1;
                pre = 0u;
            } else {
                pending = nullptr;
                // not prefetched: every wave samples the next block's flags now, under the drain of this block's stores (always exactly
                // one load, so that the drain can leave it in flight; past the last block it reads a flag nobody waits for)
                if constexpr (R::PREPOLL) {
                    const unsigned sample = __hip_atomic_load((const gu32*)flag_of(p, st.wait_group, b2, st.wait_slot0) + (lane < st.wait_n ? lane : 0) * FLAG_STRIDE,
                                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    publish<1>(st, flag_of(p, st.out_group, b, st.out_slot), s + 1);
                    pre = s2 < p.n_steps && lane < st.wait_n ? sample : 0u;
                } else {                 // four-wave workgroups (32-row plan): the extra poll per block cost more than it saved
                    pre = 0u;
                    publish(st, flag_of(p, st.out_group, b, st.out_slot), s + 1);
                }
            }
            SYS_STAMP(5);
            if constexpr (R::PREFETCH) { if (have) cur = nxt; }
        }
    if (pending != nullptr) publish(st, pending, pending_epoch);
    SYS_STAT_END;
}

// ---------------------------------------------------------------- the stage loop of the tagged hand-off (HO = 1)
// Same lattice order of (step, block).  No flags: a block's loads are ISSUED - mid-way through the previous block's compute, always,
// whether its rows are there or not - and their tags are looked at when the block's turn comes (`settle`): a wave whose words do
// not all show the expected parity loads them again.  Every wave settles by itself on exactly the words IT consumes; the waves
// meet at the barrier behind the operand tile.  Nothing is drained and nothing is published: the stores of `compute` carry the
// parity of their step.
//   r.bad(s, b, geo, pay)  bit 0 set when a handed-off word of `pay` does not show the parity of (s, b)
// A spin that has lasted `timeout` raises the abort word (as wait_epoch does) and every poller looks at that word every 64 turns.
__device__ __forceinline__ bool spin_give_up(const SysArgs& p, unsigned spins, unsigned long long t0) {
    if ((spins & 63u) != 0u) return false;
    if (__hip_atomic_load((const gu32*)p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return true;
    if (__builtin_amdgcn_s_memrealtime() - t0 > p.timeout_ticks) {
        if ((threadIdx.x & 63) == 0) {
            __hip_atomic_store((gu32*)p.status + 1, (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store((gu32*)p.status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return true;
    }
    return false;
}
#ifdef LADIFF_SELFCHECK
// Diagnostic build (scripts/race_hunt.py, never shipped): a stage that has accepted a block's rows loads them AGAIN and compares - the
// buffer of (step, block) must not change before the next step's write, which this block's own result gates.  The first mismatch of a
// launch is recorded in status[16 ..]: [17] role bit | layer << 8 | slice << 16 | where << 24, [18] step, [19] block, [20] wave | lane << 8,
// [21] mask of the 16-byte units that differ, [22] / [23] first differing unit's word 0 (used / reloaded); [24] counts all of them.
// The same build with -DLADIFF_TAG_BITS=4: a word whose low tag bit is right while the wider tag is not is a word a one-bit parity would have
// ACCEPTED from another generation (A-B-A).  status[32 ..]: [32] count, first: [33] role | layer << 8 | slice << 16 | where << 24, [34] step,
// [35] block, [36] wave | lane << 8, [37] the xor of expected and seen tag bits.
__device__ __forceinline__ void aba_note(const SysArgs& p, const Stage& st, int role, int where, int s, int b, unsigned t) {
    if (TAG_MASK > 1u && (t & 1u) == 0u && (t & TAG_MASK) != 0u) {
        const int ri = 31 - __builtin_clz((unsigned)role);                // per role: [40 + ri] count, [48 + 2 ri] / [49 + 2 ri] its first event
        atomicAdd(p.status + 32, 1u);
        if (atomicAdd(p.status + 40 + ri, 1u) == 0u) {
            p.status[48 + 2 * ri] = (unsigned)(t & TAG_MASK) | (unsigned)st.layer << 8 | (unsigned)st.slice << 16 | (unsigned)where << 24;
            p.status[49 + 2 * ri] = (unsigned)s << 24 | (unsigned)b << 8 | (threadIdx.x >> 6);
        }
    }
}
template <int N>
__device__ __forceinline__ void selfcheck_record(const SysArgs& p, const Stage& st, int role, int where, int s, int b, const f32x4 (&a)[N], const f32x4 (&c)[N]) {
    unsigned m = 0u, wa = 0u, wc = 0u;
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        const u32x4 d = __builtin_bit_cast(u32x4, a[i]) ^ __builtin_bit_cast(u32x4, c[i]);
        if ((d[0] | d[1] | d[2] | d[3]) != 0u) { m |= 1u << (i & 31); wa = __builtin_bit_cast(u32x4, a[i])[0]; wc = __builtin_bit_cast(u32x4, c[i])[0]; }
    }
    if (m != 0u) {
        atomicAdd(p.status + 24, 1u);
        if (atomicCAS(p.status + 16, 0u, 1u) == 0u) {
            p.status[17] = (unsigned)role | (unsigned)st.layer << 8 | (unsigned)st.slice << 16 | (unsigned)where << 24;
            p.status[18] = (unsigned)s; p.status[19] = (unsigned)b; p.status[20] = (threadIdx.x >> 6) | (threadIdx.x & 63) << 8;
            p.status[21] = m; p.status[22] = wa; p.status[23] = wc;
        }
    }
}
template <class R>
__device__ __forceinline__ void selfcheck_pay(const SysArgs& p, const Stage& st, R& r, int s, int b, const typename R::Geo& g, const typename R::Pay& used) {
    typedef typename R::Pay Pay;
    constexpr int N = sizeof(Pay) / 16;
    static_assert(sizeof(Pay) % 16 == 0, "row images are 16-byte units");
    Pay again = used;
    r.issue(s, b, g, again);
    f32x4 a[N], c[N];
    __builtin_memcpy(a, &used, sizeof(Pay));
    __builtin_memcpy(c, &again, sizeof(Pay));
    selfcheck_record<N>(p, st, R::PAUSE_BIT, 0, s, b, a, c);
}
#endif
// Roles that request a block's rows ahead (R::PREFETCH): the rows of the NEXT block are requested at the start of this block's
// compute phase (the operand tile is committed, the row registers are free) and their tags are looked at for the first time just
// BEFORE this block's output stores (MidTag::before_stores): at that point the wave has no store in flight, so the wait for the
// loads is exact whatever the compiler makes of it (merged with other paths it waits `vmcnt(0)`; behind the stores that would be
// the drain this protocol exists to avoid).  Rows that were not there yet are loaded again when the block's turn comes (settle).
// The other roles (STYL: 256 registers, no second image) load inside the loop: ONE copy of the load and check code (two - one in
// front of the loop, one inside - kept both address sets alive and spilled weights).
// Polling policy (measured, profiles/r4: scripts/handoff_knobs.py; loop kernel at 64 / 128 / 128 mixed-length / 256 prompts):
//   * a poll IS the load of the rows (16 bytes per lane, every word checked); a wave polls back to back, one poll in flight.
//     A separate sentinel word per producer wave in front of the 9-KiB row loads of the fan-in stages cost one more round trip
//     per hop and lost everywhere (8.22 / 10.57 / 8.64 / 21.6 ms against 7.82 / 10.47 / 8.06 / 20.7), and so did a pause
//     between polls (10.47 against 10.38 at 128 prompts);
//   * a stage with a second row image (R::PREFETCH) may request the NEXT block's rows when this block's operand tile is complete
//     and look at their tags just before this block's stores - at that point the wave has no store in flight, so the wait for
//     those loads is exact whatever the compiler makes of it (`vmcnt(0)`; behind the stores that would be the drain this
//     protocol exists to avoid).  That pays when rows queue up in front of the stage (128 prompts and more: 10.38 against 11.11
//     ms) and costs when they arrive just in time (the speculative load misses, and the look at it stands in front of this
//     block's stores: 64 prompts 7.84 against 7.40 ms, mixed lengths 8.06 against 7.64) - so it is ADAPTIVE: the next block's
//     rows are requested early exactly when this block's rows were complete at the first look (10.20 ms at 128 prompts, 20.3 at
//     256), and only in launches with enough blocks to queue at all (SysArgs::look_ahead: from LOOK_AHEAD_BLOCKS up; below, one
//     block's trip through the stages bounds the step and a stage that looks ahead after a lucky hit loses: 7.62 against 7.40);
//   * requesting the rows a second time behind the stores after a miss lost everywhere (more load traffic: 11.17 ms at 128).
template <class R>
__device__ __forceinline__ bool settle(const SysArgs& p, R& r, int s, int b, const typename R::Geo& g, typename R::Pay& y, bool issued,
                                       bool& first_look) {
    first_look = false;
    unsigned long long t0 = 0ull;
    for (unsigned spins = 0;; ++spins) {
        if (spins != 0u || !issued) r.issue(s, b, g, y);
#ifdef LADIFF_SELFCHECK
        aba_note(p, r.st, R::PAUSE_BIT, 2, s, b, r.bad(s, b, g, y));
        if constexpr (R::PAUSE_BIT == 2) {                                // RED2: the residual / ticket row's words as seen
            const unsigned t_ = r.bad(s, b, g, y);
            if (TAG_MASK > 1u && (t_ & 1u) == 0u && (t_ & TAG_MASK) != 0u && (threadIdx.x & 63) == 5 && p.status[60] == 0u) {
                const u32x4 w_ = __builtin_bit_cast(u32x4, y.rs[0]);
                p.status[60] = w_[0]; p.status[61] = w_[1]; p.status[62] = w_[2]; p.status[63] = (unsigned)g.row[0];
            }
        }
#endif
        if (__all((r.bad(s, b, g, y) & TAG_MASK) == 0u)) { first_look = spins == 0u; return true; }
        if (spins == 0u) t0 = __builtin_amdgcn_s_memrealtime();
        else if (spin_give_up(p, spins, t0)) return false;
        if ((p.pause_mask & R::PAUSE_BIT) != 0)
            for (int i = 0; i < p.pause_len; ++i) __builtin_amdgcn_s_sleep(2);
    }
}
template <class R>
struct MidTag {
    const SysArgs& p; R& r; typename R::Pay& nxt; typename R::Geo& gnxt; typename R::Geo& gnn;
    int s2, b2, s3, b3;
    bool ahead;                           // in: request the next block's rows during this block's compute phase
    bool settled;                         // out: every word of the next block's rows already shows its parity (this wave's words)
    __device__ __forceinline__ void start() {                            // behind the barrier that completes the operand tile
        if constexpr (R::PREFETCH) { r.geo_fix(gnxt); if (ahead) r.issue(s2, b2, gnxt, nxt); }
    }
    __device__ __forceinline__ void before_barrier() {}
    __device__ __forceinline__ void after_barrier() {
        // geometry words are fetched two blocks ahead (as stage_loop)
        if constexpr (!R::PREFETCH) r.geo_fix(gnxt);
        if (s3 < p.n_steps) r.geo(b3, gnn);
    }
    __device__ __forceinline__ void before_stores() {
        if constexpr (R::PREFETCH) {
#ifdef LADIFF_SELFCHECK
            if (ahead) aba_note(p, r.st, R::PAUSE_BIT, 3, s2, b2, r.bad(s2, b2, gnxt, nxt));
#endif
            if (ahead) settled = __all((r.bad(s2, b2, gnxt, nxt) & TAG_MASK) == 0u);
        }
    }
};
template <class R>
__device__ __forceinline__ void tag_loop(const SysArgs& p, const Stage& st, R& r, int b0, int bstride) {
#ifdef LADIFF_SELFCHECK
    typename R::Pay cur{}, nxt{};         // (the diagnostics look at row images of padding slots too)
#else
    typename R::Pay cur, nxt;
#endif
    typename R::Geo gcur, gnxt, gnn;
    if (b0 >= p.NB || p.n_steps < 1) return;
    r.geo(b0, gcur);
    r.geo(b0 + bstride < p.NB ? b0 + bstride : b0, gnxt);
    r.geo_fix(gcur);
    gnn = gnxt;
    bool settled = false, issued = false, ahead = false;
    unsigned est = 0u;                    // the wait this wave saw for its last block's rows (10-ns ticks): pacing, see below
    SYS_STAT_DECL;
    for (int s = 0; s < p.n_steps; ++s)
        for (int b = b0; b < p.NB; b += bstride) {
            SYS_STAMP(0);
            SYS_STAT_ITER(settled);
            if constexpr (R::BACKP) { if (!r.backpressure_tag(s, b)) return; }
            SYS_STAT_T0;
            if (!settled) {
                // Pacing (stage types with R::PAUSE_BIT in PACED_ROLES): a stage that waited W for its previous block's rows will most
                // likely wait about as long again (the pipeline is periodic) - it sleeps a fraction of W before it starts to poll
                // instead of polling through the whole wait.  A first look that hits means the rows may have been there for a while: the
                // estimate decays.  Compiled into the paced stage types only: the others keep their registers.
                constexpr bool PACED = (R::PAUSE_BIT & PACED_ROLES) != 0;
                unsigned long long te = 0ull;
                if constexpr (PACED) {
                    te = __builtin_amdgcn_s_memrealtime();
                    if ((p.pace >> 8 & R::PAUSE_BIT) != 0 && !issued && est > 8u) {
                        const unsigned long long until = te + (unsigned long long)((est * (unsigned)(p.pace & 0xff)) >> 3);
                        while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(1);
                    }
                }
                bool first = false;
                if (!settle(p, r, s, b, gcur, cur, issued, first)) return;
                if constexpr (PACED) {
                    const unsigned obs = (unsigned)(__builtin_amdgcn_s_memrealtime() - te);
                    est = first ? (est * 3u) >> 2 : (obs < 400u ? obs : 400u);  // ticks of 10 ns; never more than 4 us
                }
                ahead = first && p.look_ahead != 0;                      // rows were waiting: the stage is behind - look ahead
            } else est >>= 1;
            SYS_STAT_WAIT;
            SYS_STAMP(1);
            int s2 = s, b2 = b + bstride;
            if (b2 >= p.NB) { s2 = s + 1; b2 = b0; }
            int s3 = s2, b3 = b2 + bstride;
            if (b3 >= p.NB) { s3 = s2 + 1; b3 = b0; }
            const bool has_next = s2 < p.n_steps;                       // behind the last block: that block again (never used)
#ifdef LADIFF_SELFCHECK
            selfcheck_pay(p, st, r, s, b, gcur, cur);
#endif
            r.commit(cur);
            if constexpr (R::TILE) __syncthreads();                     // the operand tile is complete (roles without one: no barrier)
            SYS_STAMP(2);
            // A role whose row image is dead once the tile is built (R::IMAGE_DEAD: LIN, FFN, SKIP) takes the early request INTO THE SAME
            // registers.  With a second image the compiler copied the freshly requested rows into the first one at once - behind an
            // `s_waitcnt vmcnt(0)` at the head of the first product: the whole load round trip (0.3 us per block in the busiest stage,
            // FFN) stood exactly where the early request was meant to hide it (round 5, ISA of the loop body).
            constexpr bool ALIAS = R::PREFETCH && R::IMAGE_DEAD;
            MidTag<R> mid{p, r, ALIAS ? cur : nxt, gnxt, gnn, has_next ? s2 : s, has_next ? b2 : b, s3, b3, R::PREFETCH && ahead, false};
            mid.start();
            r.compute(s, b, gcur, cur, mid);
            settled = mid.settled; issued = mid.ahead;
            gcur = gnxt; gnxt = gnn;
            if ((p.delay_mask & R::PAUSE_BIT) != 0)
                for (int i = 0; i < p.delay_len; ++i) __builtin_amdgcn_s_sleep(2);
            SYS_STAMP(4);
            if constexpr (R::PREFETCH && !ALIAS) { if (issued) cur = nxt; }
            SYS_STAMP(5);
        }
    SYS_STAT_END;
}

// ---------------------------------------------------------------- roles
// QKV: one head.  in_proj rows {q,k,v} x 64 of head `slice` on the block, then softmax(q k^T / 8) v over the valid latent
// keys of the row's sample-branch, the text token and the time token.  Geometry comes from the block descriptor (LDS copy).
// WS = waves per SIMD.  The head's 192 columns are 12 MFMA tiles, three per SIMD: with one wave per SIMD the wave takes all three;
// with two, wave s (< 4) takes the first two of SIMD s and wave s + 4 the third - two instruction streams per SIMD cover each
// other's LDS / DPP / barrier latencies.
template <int MR, int AR, int WS, int HO>
struct QkvRole {
    static constexpr int RT = 16 * MR, QLD = 196, TK = LADIFF_MAX_LATENTS + 2;   // keys of a row: <= 8 latents, text, time
    static constexpr int NTH = 256 * WS, NTW = WS == 1 ? 3 : 2, NX = 2 / WS;      // threads; column tiles a wave can hold; 16-byte text K|V units per thread
    static constexpr bool PREFETCH = true;
    static constexpr bool IMAGE_DEAD = false;                            // the row image is not read again once commit() has built the tile
    static constexpr bool PREPOLL = WS == 2;
    static constexpr bool BACKP = false;
    static constexpr int PAUSE_BIT = 16;
    static constexpr bool TILE = true;
    struct Geo { int gw, rb2, b2[NX]; };                                 // descriptor word `tid` (+ its row's sample-branch); sample-branch of this thread's text slots
    struct Pay { Rows256<MR, WS> x; f32x4 xk[NX]; };
    const SysArgs& p; const Stage& st;
    char* atile; float *qt, *xt; int* gd;
    WFrag<AR, NTW, 8> wf;
    f32x4 bcol[NTW];                                                     // in_proj bias at this lane's four columns of tile j (the projection runs transposed)
    __amdgpu_buffer_rsrc_t rin, rout, rtab, rtkv;
    const float* tkv;
    int h, T, nkeys, nvt;                                                // nvt: column tiles this wave really has
    // tile column of this wave's tile j (j >= nvt: a duplicate of a valid one, loaded but never used)
    __device__ __forceinline__ int tile_col(int j) const {
        const int wave = threadIdx.x >> 6;
        if constexpr (WS == 1) return 48 * wave + 16 * j;
        else return 48 * (wave & 3) + (wave < 4 ? 16 * j : 32);
    }

    __device__ __forceinline__ QkvRole(const SysArgs& p_, const Stage& st_, char* lds) : p(p_), st(st_) {
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, frow = lane & 15;
        h = st.slice; T = p.T; nkeys = T + 2;
        atile = lds;                                                     // [RT] x K=256 operand tile
        qt = reinterpret_cast<float*>(lds + tile_bytes<AR, 4>(RT));      // [RT][QLD] q | k | v (fp32)
        xt = qt + RT * QLD;                                              // [16][128]: text k|v per sample-branch, slot 15: time k|v
        gd = reinterpret_cast<int*>(xt + 16 * 128);                      // [0] rows of this block, [1 + r] row_pk (counts resolved)
        nvt = WS == 1 ? 3 : (wave < 4 ? 2 : 1);
        // tile column tc (+ frow): part tc / 64 (q, k, v), matrix row part * 256 + h * 64 + tc % 64
        load_w(wf, st.w0, D, 0, [&](int j) { const int tc = tile_col(j); return (tc >> 6) * 256 + h * 64 + (tc & 63); });
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            if constexpr (LADIFF_TR_QKV) { const int tc = tile_col(j) + 4 * (lane >> 4); bcol[j] = ld4(st.b0 + (tc >> 6) * 256 + h * 64 + (tc & 63)); }
            else { const int tc = tile_col(j) + frow; const float bb = st.b0[(tc >> 6) * 256 + h * 64 + (tc & 63)]; bcol[j] = f32x4{bb, bb, bb, bb}; }
        }
        landed();
        rin = rsrc_of(st.in0); rout = rsrc_of(st.out);
        tkv = p.tkv + (size_t)st.layer * p.B2 * 512;
        rtab = rsrc_of(p.tables); rtkv = rsrc_of(tkv);
    }
    __device__ __forceinline__ void geo(int b, Geo& g) {
        const int tid = threadIdx.x;
        const BlockDesc* d = p.blocks + b;
#pragma unroll
        for (int u = 0; u < NX; ++u) { const int sx = (tid + NTH * u) >> 5; g.b2[u] = sx < 15 ? d->b2[sx] : -1; }
        // one load per value (two loads into one register from different branches would make the second wait for the first -
        // and for every load issued before it)
        const int* src = tid == 0 ? &d->nrows : &d->row_pk[tid <= RT ? tid - 1 : 0];
        g.gw = *src;
        g.rb2 = d->row_b2[tid >= 1 && tid <= RT ? tid - 1 : 0];
    }
    __device__ __forceinline__ unsigned bad(int s, int, const Geo&, const Pay& y) const {
        unsigned t = 0u;
        rows_bad<MR, WS>(t, y.x, (unsigned)(s & TAG_MASK));
        return t;
    }
    __device__ __forceinline__ void geo_fix(Geo& g) {                    // a latent count that lives on the device only (0xff)
        const int tid = threadIdx.x;
        if (tid >= 1 && tid <= RT && ((g.gw >> 16) & 0xff) == 0xff) {
            int c = T;
            if (g.rb2 >= 0 && p.counts != nullptr) { c = p.counts[g.rb2 % p.B]; c = c > T ? T : c; }
            g.gw = (g.gw & 0xffff) | (c << 16);
        }
    }
    __device__ __forceinline__ void issue(int s, int b, const Geo& g, Pay& y) {
        const int tid = threadIdx.x;
        const float* timekv = p.tables + (size_t)(p.step_lo + s) * DEN_STEP_STRIDE + st.layer * DEN_LAYER_STRIDE + DEN_OFF_TIME_KV;
#pragma unroll
        for (int u = 0; u < NX; ++u) {                                   // text K|V slices of this head per sample-branch, slot 15: time
            const int f4 = tid + NTH * u, sx = f4 >> 5, c4 = (f4 & 31) * 4;
            y.xk[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (sx == 15) y.xk[u] = ld4(timekv + (c4 < 64 ? c4 : 192 + c4) + h * 64);
            else if (g.b2[u] >= 0) y.xk[u] = ld4(tkv + (size_t)g.b2[u] * 512 + (c4 < 64 ? c4 : 192 + c4) + h * 64);
        }
        issue_rows<MR, WS>(y.x, rin, (unsigned)b * RT * 1024);
    }
    __device__ __forceinline__ void commit(const Pay& y) { commit_rows<AR, 4, MR, WS>(atile, 0, y.x); }
    // in_proj of this wave's column tiles on the operand tile -> q | k | v tile (fp32, q scaled)
    __device__ __forceinline__ void project() {
        const int lane = threadIdx.x & 63, frow = lane & 15;
        f32x4 acc[MR][NTW];
        zero_acc(acc);
#ifdef LADIFF_STAMPS
        bool probe = false;                       // timing probe (garbage results), bit 4: the projection at a quarter of its MFMAs
        if constexpr (AR == 0) {
            probe = (p.probe & 16) != 0;
            if (probe) {
                WFrag<0, NTW, 2> hw;
#pragma unroll
                for (int j = 0; j < NTW; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q) { hw.hi[j][q] = wf.hi[j][q]; hw.lo[j][q] = wf.lo[j][q]; }
                mma<0, 4, NTW, 2, MR, NTW, 1, LADIFF_TR_QKV>(atile, hw, acc);
            }
        }
        if (probe) {} else
#endif
        if (nvt == NTW) {
            mma<AR, 4, NTW, 8, MR, NTW, (MR == 1 && WS == 2 ? PF2 : 1), LADIFF_TR_QKV>(atile, wf, acc);
        } else {                                                         // the SIMD's second wave: one tile (the fragment set's first)
            f32x4 a1[MR][1];
            zero_acc(a1);
            mma<AR, 4, 1, 8, MR, NTW, (MR == 1 ? PF1 : 1), LADIFF_TR_QKV>(atile, wf, a1);
#pragma unroll
            for (int i = 0; i < MR; ++i) acc[i][0] = a1[i][0];
        }
        // (transposed products: a lane holds columns tile_col(j) + 4 (lane >> 4) .. + 3 of row frow - one 16-byte write per tile)
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                if (j < nvt) {
                    const float scl = tile_col(j) < 64 ? 0.125f : 1.f;  // q / sqrt(64), exact (a tile is all q or all k | v)
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (acc[i][j][r] + bcol[j][r]) * scl;
                    if constexpr (LADIFF_TR_QKV) *reinterpret_cast<f32x4*>(qt + (16 * i + frow) * QLD + tile_col(j) + 4 * (lane >> 4)) = v;
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) qt[(16 * i + 4 * (lane >> 4) + r) * QLD + tile_col(j) + frow] = v[r];
                    }
                }
            }
    }
    template <class M>
    __device__ __forceinline__ void compute(int s, int b, const Geo& g, const Pay& y, M& mid) {
        const int tid = threadIdx.x;
        // text / time K|V and the descriptor are read by the attention phases below, up to the END of the previous block's
        // compute: they go to LDS here, behind the stage loop's barrier, not in commit()
#pragma unroll
        for (int u = 0; u < NX; ++u) st4(xt + (tid + NTH * u) * 4, y.xk[u]);
        if (tid <= RT) gd[tid] = g.gw;
        project();
        SYS_STAMP(3);
        mid.before_barrier();
        __syncthreads();
        mid.after_barrier();
        SYS_STAMP(6);
        // the shipped models have T = 5 latent tokens (7 keys): the loops are unrolled over the keys, so the bound is compile time
        const unsigned par = HO ? (unsigned)(s & TAG_MASK) : 0u;
        mid.before_stores();
        if (T <= 5) attention<7>(b, par); else attention<TK>(b, par);
    }
    // ---- two waves per SIMD: the role runs as TWO WAVE GROUPS instead of the generic stage loop.  Waves 4-7 ("loaders") wait for
    // a block's flags, load it and write its operand tile while waves 0-3 still run the attention of the PREVIOUS block (the
    // operand tile is free once every wave is past the projection); all eight waves then do the projection; waves 0-3 do the
    // attention, drain their stores, count themselves in (LDS) and the last one publishes.  Per block the stage then costs
    // projection + max(attention + drain, wait + load + commit) instead of their sum.  A lone block's latency is unchanged.
    // EARLY (tagged hand-off, launches in which rows queue up in front of the stage: SysArgs::look_ahead): the loader waves request the
    // NEXT block's rows as soon as this block's tile is built - their row registers are free from then on and they never store to global
    // memory, so the load round trip (0.5 us) runs under the projection instead of standing between two tiles; the attention waves had
    // been waiting 0.6 us per block at the tile barrier.  Loop kernel 10.13 -> 9.80 ms at 128 prompts, 20.13 -> 19.25 at 256
    // (profiles/r5/08_*).  A loop of its own: in the launches that gain nothing (a block's trip bounds them) the row registers must not
    // stay live across the projection.
    template <bool EARLY>
    __device__ __forceinline__ void split_loop(Ctl* ctl) {
        static_assert(WS == 2 && MR == 1, "wave groups: the eight-wave, 16-row form only");
        const int tid = threadIdx.x, lane = tid & 63, tl = tid - 256;
        const bool loader = tid >= 256;
        typedef __attribute__((address_space(3))) unsigned lu32;
        auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
        f32x4 x[2][2];                                                   // loader waves: a block's rows between their loads and the tile
        bool xissued = false, first = true;                              // ... already requested (behind the previous block's tile); complete at the first look
        // The loader threads' geometry words of a block (one load per value, QkvRole::geo) are fetched ONE BLOCK AHEAD (round 6): read at the
        // top of the block's own iteration they stood in front of the text / time K|V loads, whose addresses they are - the loader
        // waves' chain per block was three dependent round trips (geometry, K|V slot 0, K|V slot 1: the compiler waits vmcnt(0) at the
        // head of every divergent region) before the rows were even looked at.  They wait in LDS, not in registers (the role sits at
        // 256 VGPRs: four loop-carried words per loader thread spilled a weight fragment, reloaded inside the block loop): a thread
        // requests the NEXT block's words beside this block's K|V and row loads, writes them to its own 16-byte slot once the rows'
        // tags are right (the same vmcnt(0) covers them), and reads the slot back at the top of the next iteration.
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        i32x4* const gslot = reinterpret_cast<i32x4*>(gd + 64) + (loader ? tl : 0);
        f32x4* const xstage = reinterpret_cast<f32x4*>(gd + 64 + 256 * 4) + (loader ? tl : 0);      // [2][256]
        auto load_geo = [&](int bb, int (&b2o)[2], int& gwo, int& rb2o) __attribute__((always_inline)) {
            const BlockDesc* d = p.blocks + bb;
#pragma unroll
            for (int u = 0; u < 2; ++u) { const int sx = (tl + 256 * u) >> 5; b2o[u] = sx < 15 ? d->b2[sx] : -1; }
            const int* src = tl == 0 ? &d->nrows : &d->row_pk[tl <= RT ? tl - 1 : 0];
            gwo = *src;
            rb2o = d->row_b2[tl >= 1 && tl <= RT ? tl - 1 : 0];
        };
        if (loader && st.blk0 < p.NB) {
            int b2f[2], gwf, rb2f;
            load_geo(st.blk0, b2f, gwf, rb2f);
            *gslot = i32x4{b2f[0], b2f[1], gwf, rb2f};
        }
        SYS_SPLIT_DECL;
        for (int s = 0; s < p.n_steps; ++s)
            for (int b = st.blk0; b < p.NB; b += st.blkstride) {
                f32x4 xk[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
                int gw = 0;
                if (loader) {
                    const i32x4 g4 = *gslot;                             // this thread's own slot: no barrier
                    int b2[2] = {g4[0], g4[1]};
                    gw = g4[2];
                    const int rb2 = g4[3];
                    int nb2[2], ngw, nrb2;
                    {
                        int bn = b + st.blkstride;
                        if (bn >= p.NB) bn = st.blk0;
                        load_geo(bn, nb2, ngw, nrb2);
                    }
                    // the text / time K|V rows do not depend on the block's flags (the per-call tables): they are requested BEFORE the
                    // wait - these are plain loads of rows nobody touched since the prologue, often a trip to the memory side
                    // (buffer loads with the table's base in scalar registers: as 64-bit per-lane pointers these addresses were spilled)
                    const unsigned timeoff = (unsigned)(((p.step_lo + s) * DEN_STEP_STRIDE + st.layer * DEN_LAYER_STRIDE + DEN_OFF_TIME_KV + h * 64) * 4);
#pragma unroll
                    for (int u = 0; u < 2; ++u) {                        // text K|V slices of this head per sample-branch, slot 15: time
                        const int f4 = tl + 256 * u, sx = f4 >> 5, c4 = (f4 & 31) * 4;
                        const unsigned col = (unsigned)(c4 < 64 ? c4 : 192 + c4) * 4u;
                        if (sx == 15) xk[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rtab, col, timeoff, 0));
                        else if (b2[u] >= 0) xk[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rtkv, (unsigned)b2[u] * 2048u + col, (unsigned)h * 256u, 0));
                    }
                    SYS_SPLIT_T0;
                    if constexpr (!HO) { if (!wait_epoch(flag_of(p, st.wait_group, b, st.wait_slot0), st.wait_n, s + 1, p.status, ctl, 0u, p.timeout_ticks)) return; }
                    const unsigned base = (unsigned)b * RT * 1024;
                    auto load_x = [&](unsigned bs) {
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int id = tl + 256 * u, row = id >> 5, c8 = id & 31;
                            x[u][0] = ld_sc1(rin, bs + row * 1024 + c8 * 32);
                            x[u][1] = ld_sc1(rin, bs + row * 1024 + c8 * 32 + 16);
                        }
                    };
                    if (!xissued) load_x(base);
                    first = true;
                    if constexpr (HO) {              // tagged hand-off: the rows are loaded until every word shows this step's parity
                        const unsigned par = (unsigned)(s & TAG_MASK);
                        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                        for (unsigned spins = 1;; ++spins) {
                            unsigned t = 0u;
#pragma unroll
                            for (int u = 0; u < 2; ++u) { tag_acc(t, x[u][0], par); tag_acc(t, x[u][1], par); }
#ifdef LADIFF_SELFCHECK
                            aba_note(p, st, 16, 4, s, b, t);
#endif
                            if (__all((t & TAG_MASK) == 0u)) break;
                            first = false;
                            if (spin_give_up(p, spins, t0)) return;
                            __builtin_amdgcn_s_sleep(1);
                            if ((p.pause_mask & 16) != 0) for (int i_ = 0; i_ < p.pause_len; ++i_) __builtin_amdgcn_s_sleep(2);
                            load_x(base);
                        }
                    }
                    SYS_SPLIT_WAIT;
                    SYS_STAMP_L(1);
                    *gslot = i32x4{nb2[0], nb2[1], ngw, nrb2};           // the next block's words (requested above, landed with the rows)
                    // EARLY (rows queue up: the row registers stay in flight across the projection): the text / time K|V units wait in LDS
                    // too - eight registers less across the tile build and the projection, where this instantiation reloaded a spilled
                    // weight fragment in every block (loop -0.65 % at 128 prompts; the other instantiation loses 0.3 % with it: not there)
                    if constexpr (EARLY) { xstage[0] = xk[0]; xstage[256] = xk[1]; }
                    if (tl >= 1 && tl <= RT && ((gw >> 16) & 0xff) == 0xff) {        // a latent count that lives on the device only
                        int c = T;
                        if (rb2 >= 0 && p.counts != nullptr) { c = p.counts[rb2 % p.B]; c = c > T ? T : c; }
                        gw = (gw & 0xffff) | (c << 16);
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {                        // rows -> operand tile (as commit_rows)
                        const int id = tl + 256 * u, row = id >> 5, c8 = id & 31;
                        const f32x4 v0 = untag4(x[u][0]), v1 = untag4(x[u][1]);
                        if constexpr (AR == 0) {
                            s16x8 hi, lo;
                            split8(v0, v1, hi, lo);
                            *reinterpret_cast<s16x8*>(a_slot<4>(atile, row, c8 >> 3, c8 & 7)) = hi;
                            *reinterpret_cast<s16x8*>(a_slot<4>(atile, row, c8 >> 3, 8 + (c8 & 7))) = lo;
                        } else {
                            tile_put4<1, 4>(atile, row, c8 * 8, v0);
                            tile_put4<1, 4>(atile, row, c8 * 8 + 4, v1);
                        }
                    }
#ifdef LADIFF_SELFCHECK
                    {
                        f32x4 a[8], c[8];
                        a[0] = x[0][0]; a[1] = x[0][1]; a[2] = x[1][0]; a[3] = x[1][1];
                        a[4] = EARLY ? xstage[0] : xk[0]; a[5] = EARLY ? xstage[256] : xk[1];
                        a[6] = __builtin_bit_cast(f32x4, g4); a[7] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int id = tl + 256 * u, row = id >> 5, c8 = id & 31;
                            c[2 * u] = ld_sc1(rin, base + row * 1024 + c8 * 32);
                            c[2 * u + 1] = ld_sc1(rin, base + row * 1024 + c8 * 32 + 16);
                        }
                        int cb2[2], cgw, crb2;
                        load_geo(b, cb2, cgw, crb2);
                        c[6] = __builtin_bit_cast(f32x4, i32x4{cb2[0], cb2[1], cgw, crb2}); c[7] = a[7];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int f4 = tl + 256 * u, sx = f4 >> 5, c4 = (f4 & 31) * 4;
                            const unsigned col = (unsigned)(c4 < 64 ? c4 : 192 + c4) * 4u;
                            c[4 + u] = f32x4{0.f, 0.f, 0.f, 0.f};
                            if (sx == 15) c[4 + u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rtab, col, timeoff, 0));
                            else if (cb2[u] >= 0) c[4 + u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rtkv, (unsigned)cb2[u] * 2048u + col, (unsigned)h * 256u, 0));
                        }
                        selfcheck_record<8>(p, st, 16, 1, s, b, a, c);
                    }
#endif
                    // the next block's rows are requested now when this block's were complete at the first look (rows queue up in front of the stage)
                    xissued = false;
                    if constexpr (HO != 0 && EARLY) {
                        int s2 = s, b2n = b + st.blkstride;
                        if (b2n >= p.NB) { s2 = s + 1; b2n = st.blk0; }
                        xissued = first && s2 < p.n_steps;
                        if (xissued) {
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                const int id = tl + 256 * u, row = id >> 5, c8 = id & 31;
                                x[u][0] = ld_sc1(rin, (unsigned)b2n * RT * 1024 + row * 1024 + c8 * 32);
                                x[u][1] = ld_sc1(rin, (unsigned)b2n * RT * 1024 + row * 1024 + c8 * 32 + 16);
                            }
                        }
                    }
                }
                if (!loader) SYS_SPLIT_T0;
                lds_barrier();                                           // the operand tile is there; the previous block's attention is over
                if (!loader) SYS_SPLIT_IDLE;
                SYS_STAMP(2);
                if (loader) {                                            // text / time K|V and the geometry words of THIS block
#pragma unroll
                    for (int u = 0; u < 2; ++u) st4(xt + (tl + 256 * u) * 4, EARLY ? xstage[256 * u] : xk[u]);
                    if (tl <= RT) gd[tl] = gw;
                }
                project();
                lds_barrier();
                SYS_STAMP(3);
                if (!loader) {
                    const unsigned par = HO ? (unsigned)(s & TAG_MASK) : 0u;
                    if (T <= 5) attention<7>(b, par); else attention<TK>(b, par);
                    SYS_STAMP(4);
                    if constexpr (!HO) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's rows have landed
                        unsigned old = 0u;
                        if (lane == 0) old = __hip_atomic_fetch_add((lu32*)&ctl->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        old = __builtin_amdgcn_readfirstlane(old);
                        if ((old & 3u) == 3u) raise(st, flag_of(p, st.out_group, b, st.out_slot), (unsigned)(s + 1));   // the last of the four attention waves
                    }
                    SYS_STAMP(5);
                }
            }
        SYS_SPLIT_END;
    }
    template <int NKEY>
    __device__ __forceinline__ void attention(int b, unsigned par) {
        const int tid = threadIdx.x;
        const int nrows = gd[0];
        // attention of a row on 16 lanes (4 of the head's 64 columns each): the scores are reduced across the lanes with DPP,
        // softmax and the weighted sum of the values stay in registers - no barrier, no score tile
#pragma unroll
        for (int u = 0; u < (256 * MR + NTH - 1) / NTH; ++u) {           // 16 lanes per row: 256 MR items over NTH threads (whole waves may have none)
            const int id = tid + NTH * u, row = id >> 4, c4 = (id & 15) * 4;
            if (id >= 256 * MR) continue;
            const bool live = row < nrows;
            const int pk = live ? gd[1 + row] : 0, sx = pk & 0xff, r0 = (pk >> 8) & 0xff, nk = pk >> 16;
            const f32x4 q4 = ld4(qt + (live ? row : 0) * QLD + c4);
            const float* kt = qt + r0 * QLD + 64 + c4;                   // latent keys of the row's sample-branch; values 64 further
            const float* xs = xt + sx * 128 + c4;                        // its text token (k | v)
            const float* xm = xt + 15 * 128 + c4;                        // the time token
            // every load below is unconditional (a masked or absent key reads a valid dummy address): straight-line code, all the
            // LDS reads of a row in flight at once
            float e[NKEY];
            f32x4 k4[NKEY], v4[NKEY];
#pragma unroll
            for (int j = 0; j < NKEY; ++j) {
                const bool on = live && j < nkeys && (j >= T || j < nk);     // keys >= the latent count: masked
                const float* kp = j < T ? kt + j * QLD : (j == T ? xs : xm);
                k4[j] = ld4(on ? kp : qt + c4);
                if constexpr (WS == 1) v4[j] = ld4(on ? kp + 64 : qt + c4);
            }
#pragma unroll
            for (int j = 0; j < NKEY; ++j)
                e[j] = row16_sum(fmaf(q4[0], k4[j][0], fmaf(q4[1], k4[j][1], fmaf(q4[2], k4[j][2], q4[3] * k4[j][3]))));
            if constexpr (WS == 2) {                                     // 256 registers per wave: the values are fetched once the keys are done with
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NKEY; ++j) {
                    const bool on = live && j < nkeys && (j >= T || j < nk);
                    const float* kp = j < T ? kt + j * QLD : (j == T ? xs : xm);
                    v4[j] = ld4(on ? kp + 64 : qt + c4);
                }
            }
            float m = -INFINITY;
#pragma unroll
            for (int j = 0; j < NKEY; ++j) {
                const bool on = j < nkeys && (j >= T || j < nk);
                e[j] = on ? e[j] : -INFINITY;
                m = fmaxf(m, e[j]);
            }
            float l = 0.f;
#pragma unroll
#ifdef LADIFF_PROBE_QKV_NOSOFTMAX      // variant build (garbage results): QKV's attention without the exponentials
            for (int j = 0; j < NKEY; ++j) { e[j] = e[j] - m; l += e[j]; }
#else
            for (int j = 0; j < NKEY; ++j) { e[j] = __builtin_amdgcn_exp2f((e[j] - m) * 1.4426950408889634f); l += e[j]; }   // masked: exp2(-inf) = 0
#endif
            const float inv = 1.f / l;
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NKEY; ++j) {
                const float pj = e[j] * inv;                            // exactly 0 for a masked key
                o[0] = fmaf(pj, v4[j][0], o[0]); o[1] = fmaf(pj, v4[j][1], o[1]); o[2] = fmaf(pj, v4[j][2], o[2]); o[3] = fmaf(pj, v4[j][3], o[3]);
            }
            if (!live) o = f32x4{0.f, 0.f, 0.f, 0.f};
            st_out<HO>(st, rout, ((unsigned)b * RT + row) * 1024 + (h * 64 + c4) * 4, o, par);
        }
    }
};

// OUT: X1 = LN1(x + out_proj(att))
template <int MR, int AR, int WS, int HO>
struct OutRole {
    static constexpr int RT = 16 * MR, NW = 4 * WS, RPW = RT / NW, NTW = 16 / NW;     // rows / column tiles per wave
    static constexpr bool PREFETCH = true;
    static constexpr bool IMAGE_DEAD = false;                            // the row image is not read again once commit() has built the tile
    static constexpr bool PREPOLL = WS == 2;
    static constexpr bool BACKP = false;
    static constexpr int PAUSE_BIT = 32;
    static constexpr bool TILE = true;
    struct Geo {};
    struct Pay { Rows256<MR, WS> att; f32x4 res[RPW]; };
    const SysArgs& p; const Stage& st;
    char* atile; float* ct;
    WFrag<AR, NTW, 8> wf;
    f32x4 bias, gg, bb;
    __amdgpu_buffer_rsrc_t ratt, rx, rout;
    __device__ __forceinline__ OutRole(const SysArgs& p_, const Stage& st_, char* lds) : p(p_), st(st_) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        atile = lds; ct = reinterpret_cast<float*>(lds + tile_bytes<AR, 4>(RT));
        load_w(wf, st.w0, D, 0, [&](int j) { return 16 * NTW * wave + 16 * j; });
        bias = ld4(st.b0 + 4 * lane); gg = ld4(st.g + 4 * lane); bb = ld4(st.be + 4 * lane);
        landed();
        ratt = rsrc_of(st.in0); rx = rsrc_of(st.in1); rout = rsrc_of(st.out);
    }
    __device__ __forceinline__ void geo(int, Geo&) {}
    __device__ __forceinline__ void geo_fix(Geo&) {}
    __device__ __forceinline__ unsigned bad(int s, int, const Geo&, const Pay& y) const {
        unsigned t = 0u;
        rows_bad<MR, WS>(t, y.att, (unsigned)(s & TAG_MASK));
#pragma unroll
        for (int q = 0; q < RPW; ++q) tag_acc(t, y.res[q], (unsigned)(s & TAG_MASK));
        return t;
    }
    __device__ __forceinline__ void issue(int, int b, const Geo&, Pay& y) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const unsigned base = (unsigned)b * RT * 1024;
        issue_rows<MR, WS>(y.att, ratt, base);
#pragma unroll
        for (int q = 0; q < RPW; ++q) y.res[q] = ld_sc1(rx, base + (wave + NW * q) * 1024 + lane * 16);
    }
    __device__ __forceinline__ void commit(const Pay& y) { commit_rows<AR, 4, MR, WS>(atile, 0, y.att); }
    template <class M>
    __device__ __forceinline__ void compute(int s, int b, const Geo&, const Pay& y, M& mid) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = 4 * lane;
        const unsigned base = (unsigned)b * RT * 1024;
        const unsigned par = HO ? (unsigned)(s & TAG_MASK) : 0u;
        f32x4 acc[MR][NTW];
        zero_acc(acc);
        mma<AR, 4, NTW, 8, MR, NTW, (MR == 1 && WS == 2 ? PF2 : 1), LADIFF_TR_OUT>(atile, wf, acc);
        SYS_STAMP(3);
        stage_c_sel<LADIFF_TR_OUT>(ct, acc, [&](int j) { return 16 * NTW * wave + 16 * j; });
        mid.before_barrier();
        __syncthreads();
        mid.after_barrier();
        mid.before_stores();
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int row = wave + NW * q;
            f32x4 v = ld4(ct + row * CLD + c);
            const f32x4 res = untag4(y.res[q]);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = v[i] + bias[i] + res[i];
            float mean, rstd;
            row_stats4(v, mean, rstd);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ln_apply(v[i], mean, rstd, gg[i], bb[i]);
            st_out<HO>(st, rout, base + row * 1024 + c * 4, v, par);
        }
    }
    // ---- two waves per SIMD: two wave groups, as QkvRole::split_loop.  Waves 4-7 wait for a block's flags, load the attention rows
    // and write the operand tile while waves 0-3 still run the PREVIOUS block's residual + LayerNorm epilogue, drain their stores
    // and publish; all eight waves do the out-projection.  Per block: projection + max(epilogue + drain, wait + load + commit).
    __device__ __forceinline__ void split_loop(Ctl* ctl) {
        static_assert(WS == 2 && MR == 1, "wave groups: the eight-wave, 16-row form only");
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tl = tid - 256;
        const bool loader = tid >= 256;
        f32x4 ebias[4], egg[4], ebb[4];                                  // bias / LayerNorm gamma, beta in the epilogue layout
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cc = 64 * k + 4 * (lane & 15);
            ebias[k] = ld4(st.b0 + cc); egg[k] = ld4(st.g + cc); ebb[k] = ld4(st.be + cc);
        }
        typedef __attribute__((address_space(3))) unsigned lu32;
        auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
        SYS_SPLIT_DECL;
        for (int s = 0; s < p.n_steps; ++s)
            for (int b = st.blk0; b < p.NB; b += st.blkstride) {
                const unsigned base = (unsigned)b * RT * 1024;
                const unsigned par = HO ? (unsigned)(s & TAG_MASK) : 0u;
                if (loader) {
                    SYS_SPLIT_T0;
                    if constexpr (!HO) { if (!wait_epoch(flag_of(p, st.wait_group, b, st.wait_slot0), st.wait_n, s + 1, p.status, ctl, 0u, p.timeout_ticks)) return; }
                    f32x4 x[2][2];
                    auto load_x = [&] {
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int id = tl + 256 * u, row = id >> 5, c8 = id & 31;
                            x[u][0] = ld_sc1(ratt, base + row * 1024 + c8 * 32);
                            x[u][1] = ld_sc1(ratt, base + row * 1024 + c8 * 32 + 16);
                        }
                    };
                    load_x();
                    if constexpr (HO) {              // tagged hand-off: all 16 attention rows (the four heads' columns) show this step's parity
                        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                        for (unsigned spins = 1;; ++spins) {
                            unsigned t = 0u;
#pragma unroll
                            for (int u = 0; u < 2; ++u) { tag_acc(t, x[u][0], par); tag_acc(t, x[u][1], par); }
#ifdef LADIFF_SELFCHECK
                            aba_note(p, st, 32, 4, s, b, t);
#endif
                            if (__all((t & TAG_MASK) == 0u)) break;
                            if (spin_give_up(p, spins, t0)) return;
                            __builtin_amdgcn_s_sleep(1);
                            if ((p.pause_mask & 32) != 0) for (int i_ = 0; i_ < p.pause_len; ++i_) __builtin_amdgcn_s_sleep(2);
                            load_x();
                        }
                    }
                    SYS_SPLIT_WAIT;
                    SYS_STAMP_L(1);
#pragma unroll
                    for (int u = 0; u < 2; ++u) {                        // rows -> operand tile (as commit_rows)
                        const int id = tl + 256 * u, row = id >> 5, c8 = id & 31;
                        const f32x4 v0 = untag4(x[u][0]), v1 = untag4(x[u][1]);
                        if constexpr (AR == 0) {
                            s16x8 hi, lo;
                            split8(v0, v1, hi, lo);
                            *reinterpret_cast<s16x8*>(a_slot<4>(atile, row, c8 >> 3, c8 & 7)) = hi;
                            *reinterpret_cast<s16x8*>(a_slot<4>(atile, row, c8 >> 3, 8 + (c8 & 7))) = lo;
                        } else {
                            tile_put4<1, 4>(atile, row, c8 * 8, v0);
                            tile_put4<1, 4>(atile, row, c8 * 8 + 4, v1);
                        }
                    }
#ifdef LADIFF_SELFCHECK
                    {
                        f32x4 a[4] = {x[0][0], x[0][1], x[1][0], x[1][1]}, c[4];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int id = tl + 256 * u, row = id >> 5, c8 = id & 31;
                            c[2 * u] = ld_sc1(ratt, base + row * 1024 + c8 * 32);
                            c[2 * u + 1] = ld_sc1(ratt, base + row * 1024 + c8 * 32 + 16);
                        }
                        selfcheck_record<4>(p, st, 32, 1, s, b, a, c);
                    }
#endif
                }
                if (!loader) SYS_SPLIT_T0;
                lds_barrier();                                           // the operand tile is there (the loaders saw the block's flags); the previous epilogue is over
                if (!loader) SYS_SPLIT_IDLE;
                SYS_STAMP(2);
                // epilogue layout: a row on 16 lanes (lane l16: columns 64 k + 4 l16 .., k < 4), rows 4 wave .. 4 wave + 3 of waves 0-3
                // at once - the LayerNorm reductions of the four rows are then four-step DPP chains running side by side instead
                // of four six-step chains one after the other (same summation order, same bits: row_stats16)
                const int erow = 4 * (wave & 3) + (lane >> 4), ec = 4 * (lane & 15);
                f32x4 res[4];
                if (!loader) {                                           // the residual row comes in under the projection
#pragma unroll
                    for (int k = 0; k < 4; ++k) res[k] = ld_sc1(rx, base + erow * 1024 + (64 * k + ec) * 4);
                }
                f32x4 acc[MR][NTW];
                zero_acc(acc);
#ifdef LADIFF_STAMPS
                bool probe = false;               // timing probe (garbage results), bit 2: the out-projection at a quarter of its MFMAs
                if constexpr (AR == 0) {
                    probe = (p.probe & 4) != 0;
                    if (probe) {
                        WFrag<0, NTW, 2> hw;
#pragma unroll
                        for (int j = 0; j < NTW; ++j)
#pragma unroll
                            for (int q = 0; q < 2; ++q) { hw.hi[j][q] = wf.hi[j][q]; hw.lo[j][q] = wf.lo[j][q]; }
                        mma<0, 4, NTW, 2, MR, NTW, 1, LADIFF_TR_OUT>(atile, hw, acc);
                    }
                }
                if (!probe)
#endif
                mma<AR, 4, NTW, 8, MR, NTW, PF2, LADIFF_TR_OUT>(atile, wf, acc);
                stage_c_sel<LADIFF_TR_OUT>(ct, acc, [&](int j) { return 16 * NTW * wave + 16 * j; });
                lds_barrier();
                SYS_STAMP(3);
                if (!loader) {
                    if constexpr (HO) {
                        // the residual rows were consumed by this layer's QKV stages before any attention row existed: their tags are
                        // checked all the same (every handed-off word is), and a miss loads them again
                        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                        for (unsigned spins = 1;; ++spins) {
                            unsigned t = 0u;
#pragma unroll
                            for (int k = 0; k < 4; ++k) tag_acc(t, res[k], par);
                            if (__all((t & TAG_MASK) == 0u)) break;
                            if (spin_give_up(p, spins, t0)) return;
                            __builtin_amdgcn_s_sleep(1);
#pragma unroll
                            for (int k = 0; k < 4; ++k) res[k] = ld_sc1(rx, base + erow * 1024 + (64 * k + ec) * 4);
                        }
                    }
                    SYS_STAMP(6);
                    f32x4 v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[k] = ld4(ct + erow * CLD + 64 * k + ec);
                        const f32x4 rk = untag4(res[k]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[k][i] = v[k][i] + ebias[k][i] + rk[i];
                    }
                    float mean, rstd;
#ifdef LADIFF_PROBE_OUT_NOSTATS        // variant build (scripts/build_variant.sh, garbage results): OUT's epilogue without its LayerNorm statistics
                    mean = v[0][0]; rstd = 1.f;
#else
                    row_stats16(v, mean, rstd);
#endif
                    SYS_STAMP(7);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[k][i] = ln_apply(v[k][i], mean, rstd, egg[k][i], ebb[k][i]);
                        st_out<HO>(st, rout, base + erow * 1024 + (64 * k + ec) * 4, v[k], par);
                    }
                    if constexpr (!HO) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's rows have landed
                        unsigned old = 0u;
                        if (lane == 0) old = __hip_atomic_fetch_add((lu32*)&ctl->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        old = __builtin_amdgcn_readfirstlane(old);
                        if ((old & 3u) == 3u) raise(st, flag_of(p, st.out_group, b, st.out_slot), (unsigned)(s + 1));   // the last of the four epilogue waves
                    }
                    SYS_STAMP(5);
                }
            }
        SYS_SPLIT_END;
    }
};

// LIN / FFN: hidden slice = act(x W1_slice^T + b1_slice) (128 columns), partial = hidden . W2[:, slice]^T (256 columns)
template <int MR, int ACT, int AR, int WS, int HO>
struct MlpRole {
    static constexpr int RT = 16 * MR, NW = 4 * WS, NT1 = 8 / NW, NT2 = 16 / NW;       // hidden / output column tiles per wave
    static constexpr bool PREFETCH = true;
    static constexpr bool IMAGE_DEAD = true;                            // the row image is not read again once commit() has built the tile
    static constexpr bool PREPOLL = WS == 2;
    static constexpr bool BACKP = true;
    static constexpr int PAUSE_BIT = ACT == ACT_GELU ? 8 : 1;          // FFN : LIN
    static constexpr bool TILE = true;
    struct Geo {};
    struct Pay { Rows256<MR, WS> x; };
    const SysArgs& p; const Stage& st;
    char *atile, *htile; float* ct;
    WFrag<AR, NT1, 8> w1;
    WFrag<AR, NT2, 4> w2;
    f32x4 b1[NT1];                                                       // linear1's bias at this lane's four hidden columns of tile j (the products run transposed)
    __amdgpu_buffer_rsrc_t rin, rout;
    unsigned plane;
    __device__ __forceinline__ MlpRole(const SysArgs& p_, const Stage& st_, char* lds) : p(p_), st(st_) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fk = lane >> 4;
        atile = lds;                                                     // [RT] x K=256
        htile = lds + tile_bytes<AR, 4>(RT);                             // [RT] x K=128 (hidden slice, operand format)
        ct = reinterpret_cast<float*>(lds + tile_bytes<AR, 4>(RT) + tile_bytes<AR, 2>(RT));
        const int j0 = st.slice * HS;
        load_w(w1, st.w0, D, 0, [&](int j) { return j0 + 16 * NT1 * wave + 16 * j; });
        load_w(w2, st.w1, FF, 2 * st.slice, [&](int j) { return 16 * NT2 * wave + 16 * j; });
#pragma unroll
        for (int j = 0; j < NT1; ++j) b1[j] = ld4(st.b0 + j0 + 16 * NT1 * wave + 16 * j + 4 * fk);
        landed();
        rin = rsrc_of(st.in0); rout = rsrc_of(st.out);
        plane = (unsigned)st.slice * PRING * RT * 1024;
    }
    __device__ __forceinline__ void geo(int, Geo&) {}
    __device__ __forceinline__ void geo_fix(Geo&) {}
    __device__ __forceinline__ unsigned bad(int s, int, const Geo&, const Pay& y) const {
        unsigned t = 0u;
        rows_bad<MR, WS>(t, y.x, (unsigned)(s & TAG_MASK));
        return t;
    }
    __device__ __forceinline__ void issue(int, int b, const Geo&, Pay& y) { issue_rows<MR, WS>(y.x, rin, (unsigned)b * RT * 1024); }
    __device__ __forceinline__ void commit(const Pay& y) { commit_rows<AR, 4, MR, WS>(atile, 0, y.x); }
    // tagged hand-off: the same ring rule, read from the consumer's OUTPUT rows - a row of block x that shows the parity of x's
    // step was stored by a consumer wave that had loaded its partial rows of x before.  One word per row, all RT rows (every
    // wave of the consumer group stores rows of its own, padding rows included).  Needed only when the ring is shorter than a step: a block's next
    // step waits for its previous one through the tail stage, so no stage is ever more than NB blocks ahead of another.
    __device__ __forceinline__ bool backpressure_tag(int s, int b) {
        if (p.NB <= PRING) return true;
        const int g = s * p.NB + b, lane = threadIdx.x & 63;
        if (g % (PRING / 2) != 0) return true;
        const __amdgpu_buffer_rsrc_t rb = rsrc_of(st.bp_buf);
        for (int i = 0; i < st.bp_blocks; ++i) {
            const int gx = g - PRING / 2 - i;
            if (gx < 0) continue;
            const int sx = gx / p.NB, x = gx - sx * p.NB;
            const unsigned par = (unsigned)(sx & TAG_MASK);
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (unsigned spins = 1;; ++spins) {
                unsigned w = par;
                if (lane < RT) w = __builtin_amdgcn_raw_buffer_load_b32(rb, ((unsigned)x * RT + lane) * 1024, 0, 16);
#ifdef LADIFF_SELFCHECK
                aba_note(p, st, PAUSE_BIT, 5, sx, x, w ^ par);
#endif
                if (__all(((w ^ par) & TAG_MASK) == 0u)) break;
                if (spin_give_up(p, spins, t0)) return false;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        return true;
    }
    // the partial planes are rings of PRING slots: block b's slot was block b - PRING's.  Every PRING / 2 blocks: has the
    // consumer (both reduce parts / both block groups of it) finished block b - PRING / 2?  Its stages run in block order, so
    // it has then finished everything the next PRING / 2 blocks of this stage overwrite.
    __device__ __forceinline__ bool backpressure(int s, int b, Ctl* ctl) {
        const int g = s * p.NB + b;                                      // blocks counted through the steps
        if (g % (PRING / 2) != 0) return true;
        for (int i = 0; i < st.bp_blocks; ++i) {
            const int gx = g - PRING / 2 - i;
            if (gx < 0) continue;
            const int sx = gx / p.NB, x = gx - sx * p.NB;
            if (!wait_epoch(flag_of(p, st.bp_group, x, st.bp_slot0), st.bp_n, (unsigned)(sx + 1), p.status, ctl, 0u, p.timeout_ticks)) return false;
        }
        return true;
    }
    template <class M>
    __device__ __forceinline__ void compute(int s, int b, const Geo&, const Pay&, M& mid) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, frow = lane & 15, fk = lane >> 4;
        const unsigned base = (unsigned)((s * p.NB + b) % PRING) * RT * 1024;     // ring slot of this block
        const unsigned par = HO ? ring_use_par(p, s, b) : 0u;
        f32x4 acc1[MR][NT1];
        zero_acc(acc1);
#ifdef LADIFF_STAMPS
        // timing probe (garbage results): what would the loop do if this stage type were faster?  bit 0: FFN, bit 1: LIN - half of
        // the first product's k-steps, no activation, half of the second product's k-steps
        const bool probe = AR == 0 && (p.probe & (ACT == ACT_GELU ? 1 : 2)) != 0;
        if constexpr (AR == 0) {
            if (probe) {
                WFrag<0, NT1, 4> h1;
#pragma unroll
                for (int j = 0; j < NT1; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) { h1.hi[j][q] = w1.hi[j][q]; h1.lo[j][q] = w1.lo[j][q]; }
                mma<0, 4, NT1, 4, MR, NT1, 1, true>(atile, h1, acc1);
            }
        }
        if (!probe)
#endif
        mma<AR, 4, NT1, 8, MR, NT1, (NT1 == 1 && MR == 1 ? PF1 : 1), true>(atile, w1, acc1);
        SYS_STAMP(3);
        // hidden slice -> S-format operand tile (k = hidden column within the slice).  The product ran transposed: a lane holds the four
        // consecutive hidden columns 4 fk .. + 3 of tile row frow - one 8-byte write per plane (round 4: eight 2-byte writes)
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NT1; ++j) {
                const int k0 = 16 * NT1 * wave + 16 * j + 4 * fk;        // 0..124
                f32x4 h;
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = acc1[i][j][r] + b1[j][r];
#ifdef LADIFF_STAMPS
                if (!probe)
#endif
                {
                    if constexpr (ACT == ACT_GELU) {
#ifndef LADIFF_PROBE_FFN_NOGELU             // (defined: variant build, garbage results - FFN without its activation)
                        // GELU of the lane's four values as two packed pairs (gelu_erf2: the bits of gelu_erf)
                        const f32x2 g01 = gelu_erf2(f32x2{h[0], h[1]}), g23 = gelu_erf2(f32x2{h[2], h[3]});
                        h = f32x4{g01[0], g01[1], g23[0], g23[1]};
#endif
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) h[r] = act_c<ACT>(h[r]);
                    }
                }
                tile_put4<AR, 2>(htile, 16 * i + frow, k0, h);
            }
        SYS_STAMP(6);
        mid.before_barrier();
        __syncthreads();
        mid.after_barrier();
        SYS_STAMP(7);
        f32x4 acc2[MR][NT2];
        zero_acc(acc2);
#ifdef LADIFF_STAMPS
        if constexpr (AR == 0) {
            if (probe) {
                WFrag<0, NT2, 2> h2;
#pragma unroll
                for (int j = 0; j < NT2; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q) { h2.hi[j][q] = w2.hi[j][q]; h2.lo[j][q] = w2.lo[j][q]; }
                mma<0, 2, NT2, 2, MR, NT2, 1, true>(htile, h2, acc2);
            }
        }
        if (!probe)
#endif
        mma<AR, 2, NT2, 4, MR, NT2, (MR == 1 && WS == 2 ? PF2 : 1), true>(htile, w2, acc2);
        stage_c_t(ct, acc2, [&](int j) { return 16 * NT2 * wave + 16 * j; });
        mid.before_stores();
        // each wave stores the columns it staged itself (64 or 32 of them: 256 / 128 B per row, 4 / 8 rows per instruction): no
        // barrier - LDS serves a wave's accesses in order, and the next block's hidden tile is only written behind the stage loop's
        // barrier
        constexpr int CW = 16 * NT2, LPR = CW / 4, RPI = 64 / LPR;
#ifdef LADIFF_HALF_PLANES
        // traffic probe (a build of its own, garbage results: scripts/build_variant.sh): the partial planes at half width - hidden slices
        // 4 .. 7 store nothing, the reduce stages load planes 0 .. 3 only (57 % of the loop's hand-off bytes are these planes: is the
        // traffic what bounds the step?)
        if (st.slice >= NSLICE / 2) return;
#endif
#pragma unroll
        for (int q = 0; q < RT / RPI; ++q) {
            const int row = RPI * q + lane / LPR, cc = CW * wave + 4 * (lane % LPR);
            st_out<HO>(st, rout, plane + base + row * 1024 + cc * 4, ld4(ct + row * CLD + cc), par);
        }
    }
};

// RED2: X2 = LN2(X1 + sum_j partial_j + b2) + c[step, layer, sample | pad]
template <int MR, int WS, int HO>
struct Red2Role {
    static constexpr int RT = 16 * MR, NW = 4 * WS, PQ = ((MR == 1 ? 8 : 12) + NW - 1) / NW;   // rows per wave: a part has <= 8 (16-row blocks) / <= 11 rows
    // tagged hand-off: no operand tile, no barrier - every wave settles, reduces and stores its own rows; nothing to gain from
    // requesting the next block's partial rows early (they are produced just in time)
    static constexpr bool PREFETCH = HO == 0;
    static constexpr bool IMAGE_DEAD = false;                            // the row image is not read again once commit() has built the tile
    static constexpr bool PREPOLL = WS == 2;
    static constexpr bool BACKP = false;
    static constexpr int PAUSE_BIT = 2;
    static constexpr bool TILE = false;
    struct Geo { int pk[PQ], b2[PQ], row[PQ], t[PQ], cnt[PQ]; };          // slot wave + NW q of this part: raw words, then decoded
    struct Pay { f32x4 pl[PQ][NSLICE], rs[PQ], tv[PQ], tp[PQ]; };
    const SysArgs& p; const Stage& st;
    f32x4 bias, gg, bb;
    __amdgpu_buffer_rsrc_t rp, rx, rout;
    unsigned pstride;
    __device__ __forceinline__ Red2Role(const SysArgs& p_, const Stage& st_, char*) : p(p_), st(st_) {
        const int lane = threadIdx.x & 63;
        bias = ld4(st.b0 + 4 * lane); gg = ld4(st.g + 4 * lane); bb = ld4(st.be + 4 * lane);
        rp = rsrc_of(st.in0); rx = rsrc_of(st.in1); rout = rsrc_of(st.out);
        pstride = (unsigned)PRING * RT * 1024;
    }
    __device__ __forceinline__ void geo(int b, Geo& g) {
        const int wave = threadIdx.x >> 6;
        const BlockDesc* d = p.blocks + b;
#pragma unroll
        for (int q = 0; q < PQ; ++q) { g.pk[q] = d->part_pk[st.slice][wave + NW * q]; g.b2[q] = d->part_b2[st.slice][wave + NW * q]; }
    }
    __device__ __forceinline__ void geo_fix(Geo& g) {
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
            // a live row: its tile row; a padding row r: -2 - r (stored as zeros); no row: -1
            g.row[q] = g.pk[q] < 0 ? -1 : ((g.pk[q] & PART_PAD) ? -2 - (g.pk[q] & 0xff) : (g.pk[q] & 0xff));
            g.t[q] = (g.pk[q] >> 8) & 0xff;
            g.cnt[q] = (g.pk[q] >> 16) & 0xff;
            if (g.pk[q] < 0 || g.b2[q] < 0) g.cnt[q] = 0;
            else if (g.cnt[q] == 0xff) g.cnt[q] = p.counts != nullptr ? p.counts[g.b2[q] % p.B] : 0x7fffffff;   // device-only count
        }
    }
    __device__ __forceinline__ void issue(int s, int b, const Geo& g, Pay& y) {
        const int lane = threadIdx.x & 63, c = 4 * lane;
        const float* ct = p.ctab + ((size_t)st.layer * p.n_ctab + s) * (p.B2 + 1) * D;
        const unsigned base = (unsigned)b * RT * 1024, pbase = (unsigned)((s * p.NB + b) % PRING) * RT * 1024;   // partial planes: rings of PRING slots
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
            const int row = g.row[q];
            if (row >= 0) {
                // the row's cross-attention vector is its sample's table row while t < latent count, the pad row otherwise
                y.tp[q] = ld4(ct + (size_t)p.B2 * D + c);
                y.tv[q] = y.tp[q];
                if (g.b2[q] >= 0 && g.t[q] < g.cnt[q]) y.tv[q] = ld4(ct + (size_t)g.b2[q] * D + c);
#pragma unroll
                for (int j = 0; j < NSLICE; ++j) {
#ifdef LADIFF_HALF_PLANES
                    if (j >= NSLICE / 2) { y.pl[q][j] = y.pl[q][j - NSLICE / 2]; continue; }
#endif
                    y.pl[q][j] = ld_sc1(rp, j * pstride + pbase + row * 1024 + c * 4);
                }
                y.rs[q] = ld_sc1(rx, base + row * 1024 + c * 4);
            } else if (HO && row <= -2) {
                // A padding row is stored as zeros, but not before this step's input of the block exists: the same row of the
                // residual buffer (every producer writes all rows of its tiles) is the wave's ticket.  Without it a wave that owns
                // nothing but padding would run through the steps by itself and rewrite rows whose readers are a step behind.
                y.rs[q] = ld_sc1(rx, base + (-2 - row) * 1024 + c * 4);
                // ... and not before the partial planes of the block exist (round 6): this stage has no barrier, every wave runs by
                // itself, and a wave whose row is padding block after block (row 15 of every 15-row block) was held back by the
                // ticket only - i.e. by OUT, not by LIN.  With OUT more than a ring (PRING blocks) ahead of LIN, the wave's next LIVE
                // row found its ring slot still holding the block two uses back - whose one-bit tag is the one it expects - and
                // summed that block's partials (seen as run-to-run differences of single prompts once QKV / OUT got ahead of the
                // MLP stages; found with -DLADIFF_TAG_BITS=4 -DLADIFF_SELFCHECK, profiles/r6/22_*).  The padding rows of the planes are
                // written like every other row, so a padding row waits for them as a live one does: no wave of this stage is ever ahead
                // of the producer of its ring.
                // One word of the row per plane is enough to know that the plane's producer has reached this block: lane l looks at
                // plane l % 8 (one load instruction, eight lines - not the row's 8 KB).
                static_assert(NSLICE == 8, "a padding row's look at the planes: lane % 8 = plane");
                const f32x4 w = ld_sc1(rp, (unsigned)(lane & 7) * pstride + pbase + (unsigned)(-2 - row) * 1024 + (unsigned)(lane >> 3) * 16);
#pragma unroll
                for (int j = 0; j < NSLICE; ++j) y.pl[q][j] = w;
            }
        }
    }
    __device__ __forceinline__ void commit(const Pay&) {}
    __device__ __forceinline__ unsigned bad(int s, int b, const Geo& g, const Pay& y) const {
        const unsigned rpar = ring_use_par(p, s, b), spar = (unsigned)(s & TAG_MASK);
        unsigned t = 0u;
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
            unsigned tq = 0u, tr = 0u;
#pragma unroll
            for (int j = 0; j < NSLICE; ++j) tag_acc(tq, y.pl[q][j], rpar);
            tag_acc(tr, y.rs[q], spar);
            t |= g.row[q] != -1 ? (tq | tr) : 0u;                        // live and padding rows alike (issue)
        }
        return t;
    }
    template <class M>
    __device__ __forceinline__ void compute(int s, int b, const Geo& g, Pay& y, M& mid) {
        const int lane = threadIdx.x & 63, c = 4 * lane;
        const unsigned base = (unsigned)b * RT * 1024;
        const unsigned par = HO ? (unsigned)(s & TAG_MASK) : 0u;
        mid.before_barrier();
        if constexpr (!HO) __syncthreads();   // the stage's only barrier: the workgroup agrees on whether the next block is prefetched
        mid.after_barrier();
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
            const int row = g.row[q];
            if (row >= 0) {
#pragma unroll
                for (int j = 0; j < NSLICE; ++j) y.pl[q][j] = untag4(y.pl[q][j]);
                f32x4 v = sum8(y.pl[q]);
                const f32x4 rs = untag4(y.rs[q]);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = v[i] + bias[i] + rs[i];
                float mean, rstd;
                row_stats4(v, mean, rstd);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (v[i] - mean) * rstd * gg[i] + bb[i] + y.tv[q][i];
                st_out<HO>(st, rout, base + row * 1024 + c * 4, v, par);
            } else if (row <= -2) {
                st_out<HO>(st, rout, base + (-2 - row) * 1024 + c * 4, f32x4{0.f, 0.f, 0.f, 0.f}, par);
            }
        }
    }
};

// STYL: x' = X2 + out( SiLU( LN(sum_j partial_j + b2) * (1 + scale_t) + shift_t ) )
template <int MR, int AR, int WS, int HO>
struct StylRole {
    static constexpr int RT = 16 * MR, NW = 4 * WS, PQ = ((MR == 1 ? 8 : 12) + NW - 1) / NW, NTW = 16 / NW;
    static constexpr bool TILE = false;                                  // its operand tile is built inside compute(), behind barriers of its own
    // no prefetch image with 32-row blocks (256 weight registers + two images of 27 x 16 bytes per lane do not fit) nor with two
    // waves per SIMD (256 registers per wave: 128 of weights + two images of 11 x 16 bytes spilled; that plan runs STYL as two
    // groups on alternating blocks, which have the slack)
    static constexpr bool PREFETCH = MR == 1 && WS == 1;
    static constexpr bool IMAGE_DEAD = false;                            // the row image is not read again once commit() has built the tile
    static constexpr bool PREPOLL = WS == 2;
    static constexpr bool BACKP = false;
    static constexpr int PAUSE_BIT = 4;
    struct Geo { int pk[PQ], row[PQ]; };                                 // tile row of slot wave + NW q of this part (-1: none)
    struct Pay { f32x4 pl[PQ][NSLICE], rs[PQ], scl, shf; };
    const SysArgs& p; const Stage& st;
    char* atile; float *ct, *cst;
    WFrag<AR, NTW, 8> wf;
    // ffn.linear2 bias | out bias | norm beta: [3][256] in LDS, read where they are used (12 registers per lane otherwise: with the 128 weight registers
    // and the nine-row image of a block that was 7 dwords of scratch per block in the eight-wave kernel, and a scratch load waits behind
    // the stores in flight); the LayerNorm gamma stays in registers
    f32x4 gg;
    __amdgpu_buffer_rsrc_t rp, rx, rout, rtab;
    unsigned pstride;
    __device__ __forceinline__ StylRole(const SysArgs& p_, const Stage& st_, char* lds) : p(p_), st(st_) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        atile = lds; ct = reinterpret_cast<float*>(lds + tile_bytes<AR, 4>(16));
        cst = ct + 16 * CLD;
        load_w(wf, st.w0, D, 0, [&](int j) { return 16 * NTW * wave + 16 * j; });
        if (wave == 0) { st4(cst + 4 * lane, ld4(st.b1 + 4 * lane)); st4(cst + D + 4 * lane, ld4(st.b0 + 4 * lane)); st4(cst + 2 * D + 4 * lane, ld4(st.be + 4 * lane)); }
        gg = ld4(st.g + 4 * lane);
        landed();
        __syncthreads();
        rp = rsrc_of(st.in0); rx = rsrc_of(st.in1); rout = rsrc_of(st.out); rtab = rsrc_of(p.tables);
        pstride = (unsigned)PRING * RT * 1024;
    }
    __device__ __forceinline__ void geo(int b, Geo& g) {
        const int wave = threadIdx.x >> 6;
#pragma unroll
        for (int q = 0; q < PQ; ++q) g.pk[q] = p.blocks[b].part_pk[st.slice][wave + NW * q];
    }
    __device__ __forceinline__ void geo_fix(Geo& g) {
#pragma unroll
        for (int q = 0; q < PQ; ++q) g.row[q] = g.pk[q] < 0 ? -1 : ((g.pk[q] & PART_PAD) ? -2 - (g.pk[q] & 0xff) : (g.pk[q] & 0xff));
    }
    __device__ __forceinline__ void issue(int s, int b, const Geo& g, Pay& y) {
        const int lane = threadIdx.x & 63, c = 4 * lane;
        const unsigned base = (unsigned)b * RT * 1024, pbase = (unsigned)((s * p.NB + b) % PRING) * RT * 1024;
        // AdaLN scale | shift of this step (time tables): buffer loads with a SCALAR table offset - a per-lane 64-bit pointer held over the
        // loop was two dwords of scratch per block in the eight-wave kernel (reloaded behind the stores in flight)
        const unsigned mod = (unsigned)(((p.step_lo + s) * DEN_STEP_STRIDE + st.layer * DEN_LAYER_STRIDE + DEN_OFF_FFN_MOD) * 4);
        y.scl = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rtab, (unsigned)c * 4u, mod, 0));
        y.shf = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rtab, (unsigned)(D + c) * 4u, mod, 0));
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
            const int row = g.row[q];
            y.rs[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row >= 0) {
#pragma unroll
                for (int j = 0; j < NSLICE; ++j) {
#ifdef LADIFF_HALF_PLANES
                    if (j >= NSLICE / 2) { y.pl[q][j] = y.pl[q][j - NSLICE / 2]; continue; }
#endif
                    y.pl[q][j] = ld_sc1(rp, j * pstride + pbase + row * 1024 + c * 4);
                }
                y.rs[q] = ld_sc1(rx, base + row * 1024 + c * 4);
            } else if (HO && row <= -2) {
                y.rs[q] = ld_sc1(rx, base + (-2 - row) * 1024 + c * 4);     // a padding row's ticket (Red2Role::issue) ...
                // ... and one word of the row in each plane (never ahead of FFN, as there)
                const f32x4 w = ld_sc1(rp, (unsigned)(lane & 7) * pstride + pbase + (unsigned)(-2 - row) * 1024 + (unsigned)(lane >> 3) * 16);
#pragma unroll
                for (int j = 0; j < NSLICE; ++j) y.pl[q][j] = w;
            }
        }
    }
    __device__ __forceinline__ void commit(const Pay&) {}
    __device__ __forceinline__ unsigned bad(int s, int b, const Geo& g, const Pay& y) const {
        const unsigned rpar = ring_use_par(p, s, b), spar = (unsigned)(s & TAG_MASK);
        unsigned t = 0u;
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
            unsigned tq = 0u, tr = 0u;
#pragma unroll
            for (int j = 0; j < NSLICE; ++j) tag_acc(tq, y.pl[q][j], rpar);
            tag_acc(tr, y.rs[q], spar);
            t |= g.row[q] != -1 ? (tq | tr) : 0u;                        // live and padding rows alike (issue)
        }
        return t;
    }
    template <class M>
    __device__ __forceinline__ void compute(int s, int b, const Geo& g, Pay& y, M& mid) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = 4 * lane;
        const unsigned base = (unsigned)b * RT * 1024;
        const unsigned par = HO ? (unsigned)(s & TAG_MASK) : 0u;
        const f32x4 scl = y.scl, shf = y.shf;
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
#pragma unroll
            for (int j = 0; j < NSLICE; ++j) y.pl[q][j] = untag4(y.pl[q][j]);
        }
#pragma unroll
        for (int q = 0; q < 16 / NW; ++q) {                              // the 16 rows of the operand tile: local row wave + NW q
            const int lr = wave + NW * q;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (q < PQ && g.row[q < PQ ? q : 0] >= 0) {
                v = sum8(y.pl[q < PQ ? q : 0]);
                const f32x4 bias2 = ld4(cst + c), bb = ld4(cst + 2 * D + c);      // both reads fly under the first pass of the statistics
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] += bias2[i];
                float mean, rstd;
                row_stats4(v, mean, rstd);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = silu(((v[i] - mean) * rstd * gg[i] + bb[i]) * (1.f + scl[i]) + shf[i]);
            }
            tile_put4<AR, 4>(atile, lr, c, v);                           // u row -> operand tile (unused rows: zero)
        }
        SYS_STAMP(3);
        mid.before_barrier();
        __syncthreads();
        mid.after_barrier();
        SYS_STAMP(6);
        f32x4 acc[1][NTW];
        zero_acc(acc);
#ifdef LADIFF_STAMPS
        bool probe = false;                       // timing probe (garbage results), bit 3: the out product at a quarter of its MFMAs
        if constexpr (AR == 0) {
            probe = (p.probe & 8) != 0;
            if (probe) {
                WFrag<0, NTW, 2> hw;
#pragma unroll
                for (int j = 0; j < NTW; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q) { hw.hi[j][q] = wf.hi[j][q]; hw.lo[j][q] = wf.lo[j][q]; }
                mma<0, 4, NTW, 2, 1, NTW, 1, LADIFF_TR_STYL>(atile, hw, acc);
            }
        }
        if (!probe)
#endif
        mma<AR, 4, NTW, 8, 1, NTW, (WS == 2 ? PF2 : 1), LADIFF_TR_STYL>(atile, wf, acc);
        stage_c_sel<LADIFF_TR_STYL>(ct, acc, [&](int j) { return 16 * NTW * wave + 16 * j; });
        __syncthreads();
        SYS_STAMP(7);
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
            const int lr = wave + NW * q, row = g.row[q];
            if (row >= 0) {
                f32x4 v = ld4(ct + lr * CLD + c);
                const f32x4 rs = untag4(y.rs[q]), bias = ld4(cst + D + c);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = v[i] + bias[i] + rs[i];
                st_out<HO>(st, rout, base + row * 1024 + c * 4, v, par);
            } else if (row <= -2) {
                st_out<HO>(st, rout, base + (-2 - row) * 1024 + c * 4, f32x4{0.f, 0.f, 0.f, 0.f}, par);
            }
        }
    }
};

// SKIP: half of the 256 output columns of linear_blocks[i](cat(x, skip))
template <int MR, int AR, int WS, int HO>
struct SkipRole {
    static constexpr int RT = 16 * MR, NW = 4 * WS, NTH = 64 * NW, NTW = 8 / NW;      // column tiles per wave of this half's 128 columns
    static constexpr bool PREFETCH = true;
    static constexpr bool IMAGE_DEAD = true;                            // the row image is not read again once commit() has built the tile
    static constexpr bool PREPOLL = WS == 2;
    static constexpr bool BACKP = false;
    static constexpr int PAUSE_BIT = 64;
    static constexpr bool TILE = true;
    struct Geo {};
    struct Pay { Rows256<MR, WS> x, k; };
    const SysArgs& p; const Stage& st;
    char* atile; float* ct;
    WFrag<AR, NTW, 16> wf;
    __amdgpu_buffer_rsrc_t rx, rs, rout;
    f32x4 bias;                                                          // columns n0 + 4 (tid & 31) ..: the same for both of a thread's rows
    int n0;
    __device__ __forceinline__ SkipRole(const SysArgs& p_, const Stage& st_, char* lds) : p(p_), st(st_) {
        const int wave = threadIdx.x >> 6;
        atile = lds; ct = reinterpret_cast<float*>(lds + tile_bytes<AR, 8>(RT));     // [RT] x K=512 operand tile first
        n0 = st.slice * 128;
        load_w(wf, st.w0, 2 * D, 0, [&](int j) { return n0 + 16 * NTW * wave + 16 * j; });
        rx = rsrc_of(st.in0); rs = rsrc_of(st.in1); rout = rsrc_of(st.out);
        bias = ld4(st.b0 + n0 + (threadIdx.x & 31) * 4);
        landed();
    }
    __device__ __forceinline__ void geo(int, Geo&) {}
    __device__ __forceinline__ void geo_fix(Geo&) {}
    __device__ __forceinline__ unsigned bad(int s, int, const Geo&, const Pay& y) const {
        unsigned t = 0u;
        rows_bad<MR, WS>(t, y.x, (unsigned)(s & TAG_MASK));
        rows_bad<MR, WS>(t, y.k, (unsigned)(s & TAG_MASK));
        return t;
    }
    __device__ __forceinline__ void issue(int, int b, const Geo&, Pay& y) {
        issue_rows<MR, WS>(y.x, rx, (unsigned)b * RT * 1024);
        issue_rows<MR, WS>(y.k, rs, (unsigned)b * RT * 1024);
    }
    __device__ __forceinline__ void commit(const Pay& y) { commit_rows<AR, 8, MR, WS>(atile, 0, y.x); commit_rows<AR, 8, MR, WS>(atile, 4, y.k); }
    template <class M>
    __device__ __forceinline__ void compute(int s, int b, const Geo&, const Pay&, M& mid) {
        const int tid = threadIdx.x, wave = tid >> 6;
        const unsigned base = (unsigned)b * RT * 1024;
        const unsigned par = HO ? (unsigned)(s & TAG_MASK) : 0u;
        f32x4 acc[MR][NTW];
        zero_acc(acc);
        mma<AR, 8, NTW, 16, MR, NTW, (MR == 1 && WS == 2 ? PF2 : 1), LADIFF_TR_SKIP>(atile, wf, acc);
        stage_c_sel<LADIFF_TR_SKIP>(ct, acc, [&](int j) { return n0 + 16 * NTW * wave + 16 * j; });
        mid.before_barrier();
        __syncthreads();
        mid.after_barrier();
        mid.before_stores();
#pragma unroll
        for (int u = 0; u < 2 * MR / WS; ++u) {                          // (row, 4 columns) of this half
            const int id = tid + NTH * u, row = id >> 5, cc = n0 + (id & 31) * 4;
            f32x4 v = ld4(ct + row * CLD + cc);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += bias[i];
            st_out<HO>(st, rout, base + row * 1024 + cc * 4, v, par);
        }
    }
};

// TAIL: encoder.norm on both branches, guidance, scheduler step, latents, next step's network input (x = latents + pe).
// A tail works on UNITS: unit u = block u when a block holds both guidance branches (32-row tiles: unconditional rows first,
// the conditional row of (prompt, t) nrows / 2 further), or blocks 2u (unconditional) and 2u + 1 (conditional, same row) when
// blocks hold one branch (16-row tiles).  Tail workgroup k owns the units u = k (mod NTAIL).
template <int MR, int WS, int HO>
struct TailRole {
    static constexpr int RT = 16 * MR, NW = 4 * WS, NQ = 16 / NW;        // (prompt, latent) pairs per wave: <= 16 per unit
    struct Geo { int lat[NQ], t[NQ], rc[NQ], pad[NQ]; };                // latent row, position, conditional-branch row of pair q; padding row of a slot without a pair
    struct Pay { f32x4 eu[NQ], ec[NQ], lt[NQ], zz[NQ], pe[NQ]; };
    const SysArgs& p; const Stage& st;
    f32x4 gg, bb;
    __amdgpu_buffer_rsrc_t rin, rout;
    int T;
    __device__ __forceinline__ TailRole(const SysArgs& p_, const Stage& st_, char*) : p(p_), st(st_) {
        const int lane = threadIdx.x & 63, c = 4 * lane;
        T = p.T;
        gg = ld4(p.ng + c); bb = ld4(p.nb + c);
        rin = rsrc_of(st.in0); rout = rsrc_of(st.out);
    }
    __device__ __forceinline__ int blk_u(int u) const { return p.split ? 2 * u : u; }
    __device__ __forceinline__ int blk_c(int u) const { return p.split ? 2 * u + 1 : u; }
    __device__ __forceinline__ void publish_unit(int u, unsigned epoch) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_store((gu32*)flag_of(p, st.out_group, blk_u(u), st.out_slot), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p.split)
                __hip_atomic_store((gu32*)flag_of(p, st.out_group, blk_c(u), st.out_slot), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // local step 0: the first network input from the latents the prologue left (plain memory of earlier kernels); every row
    // of the blocks is written (padding rows: zero)
    __device__ __forceinline__ void prime(int nu) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = 4 * lane;
        for (int u = st.slice; u < nu; u += NTAIL) {
            for (int half = 0; half < (p.split ? 2 : 1); ++half) {
                const int b = p.split ? 2 * u + half : u;
                const BlockDesc* d = p.blocks + b;
                const unsigned base = (unsigned)b * RT * 1024;
                for (int q = wave; q < RT; q += NW) {
                    f32x4 xn = {0.f, 0.f, 0.f, 0.f};
                    const int lat = d->row_lat[q];
                    if (lat >= 0) {
                        const f32x4 l = ld4(p.lat + (size_t)lat * D + c), pe = ld4(p.pe + (size_t)d->row_t[q] * D + c);
#pragma unroll
                        for (int i = 0; i < 4; ++i) xn[i] = l[i] + pe[i];
                    }
                    st_out<HO>(st, rout, base + q * 1024 + c * 4, xn, 0u);    // the input of local step 0: parity 0
                }
            }
            if constexpr (!HO) publish_unit(u, 1);
        }
    }
    __device__ __forceinline__ unsigned bad(int s, const Geo& g, const Pay& y) const {
        unsigned t = 0u;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            unsigned tu = 0u, tc = 0u;
            tag_acc(tu, y.eu[i], (unsigned)(s & TAG_MASK)); tag_acc(tc, y.ec[i], (unsigned)(s & TAG_MASK));
            t |= (g.lat[i] >= 0 || g.pad[i] >= 0 ? tu : 0u) | (g.lat[i] >= 0 || g.rc[i] >= 0 ? tc : 0u);
        }
        return t;
    }
    __device__ __forceinline__ void geo(int u, Geo& g) {
        const int wave = threadIdx.x >> 6;
        const BlockDesc* d = p.blocks + blk_u(u);
#pragma unroll
        for (int i = 0; i < NQ; ++i) { const int q = wave + NW * i; g.lat[i] = d->pair_lat[q]; g.t[i] = d->pair_t[q]; g.rc[i] = d->pair_rc[q]; g.pad[i] = d->pair_pad[q]; }
    }
    __device__ __forceinline__ void issue(int s, int u, const Geo& g, Pay& y) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = 4 * lane;
        const unsigned bu = (unsigned)blk_u(u) * RT * 1024, bc = (unsigned)blk_c(u) * RT * 1024;
        const int step = p.step_lo + s;
        const float kn = p.coef[(size_t)step * LADIFF_COEF_STRIDE + 5];
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = wave + NW * i;
            y.zz[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (g.lat[i] >= 0) {
                y.eu[i] = ld_sc1(rin, bu + q * 1024 + c * 4);
                y.ec[i] = ld_sc1(rin, bc + g.rc[i] * 1024 + c * 4);
                y.lt[i] = ld4(p.lat + (size_t)g.lat[i] * D + c);
                y.pe[i] = ld4(p.pe + (size_t)g.t[i] * D + c);
                if (p.gen.on && kn != 0.f) {
                    float zz[4];
                    noise_normal4(p.gen, step, p.gen.prompt0 + (unsigned)(g.lat[i] / T), g.t[i], lane, zz);
                    y.zz[i] = f32x4{zz[0], zz[1], zz[2], zz[3]};
                } else if (p.noise != nullptr && kn != 0.f) y.zz[i] = ld4(p.noise + ((size_t)step * p.B * T + g.lat[i]) * D + c);
            } else if (HO) {                                             // the tickets of a slot that only pads (Red2Role::issue)
                if (g.pad[i] >= 0) y.eu[i] = ld_sc1(rin, bu + g.pad[i] * 1024 + c * 4);
                if (g.rc[i] >= 0) y.ec[i] = ld_sc1(rin, bc + g.rc[i] * 1024 + c * 4);
            }
        }
    }
    __device__ __forceinline__ void compute(int s, int u, const Geo& g, Pay& y) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = 4 * lane;
        const unsigned bu = (unsigned)blk_u(u) * RT * 1024, bc = (unsigned)blk_c(u) * RT * 1024;
        const float* cf = p.coef + (size_t)(p.step_lo + s) * LADIFF_COEF_STRIDE;
        const float sa = cf[0], sb = cf[1], kx0 = cf[2], kx = cf[3], ke = cf[4], kn = cf[5];
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = wave + NW * i;
            if (g.lat[i] >= 0) {
                f32x4 eu = untag4(y.eu[i]), ec = untag4(y.ec[i]), l = y.lt[i], xn;
                float mean, rstd;
                row_stats4(eu, mean, rstd);
#pragma unroll
                for (int k = 0; k < 4; ++k) eu[k] = (eu[k] - mean) * rstd * gg[k] + bb[k];
                row_stats4(ec, mean, rstd);
#pragma unroll
                for (int k = 0; k < 4; ++k) ec[k] = (ec[k] - mean) * rstd * gg[k] + bb[k];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float e = eu[k] + p.gscale * (ec[k] - eu[k]);
                    const float x0 = (l[k] - sb * e) / sa;
                    l[k] = kx0 * x0 + kx * l[k] + ke * e + kn * y.zz[i][k];
                    xn[k] = l[k] + y.pe[i][k];
                }
                st4(p.lat + (size_t)g.lat[i] * D + c, l);
                const unsigned par = HO ? (unsigned)((s + 1) & TAG_MASK) : 0u;     // the input of the NEXT local step
                st_out<HO>(st, rout, bu + q * 1024 + c * 4, xn, par);          // after the last step nobody reads it
                st_out<HO>(st, rout, bc + g.rc[i] * 1024 + c * 4, xn, par);
            } else {                                                       // padding rows of the unit's tiles: zeros, next step's parity
                const unsigned par = HO ? (unsigned)((s + 1) & TAG_MASK) : 0u;
                if (g.pad[i] >= 0) st_out<HO>(st, rout, bu + g.pad[i] * 1024 + c * 4, f32x4{0.f, 0.f, 0.f, 0.f}, par);
                if (g.rc[i] >= 0) st_out<HO>(st, rout, bc + g.rc[i] * 1024 + c * 4, f32x4{0.f, 0.f, 0.f, 0.f}, par);
            }
        }
    }
};

// The tail's loop differs from stage_loop: its output of step s is the input of step s + 1 (epoch s + 2), it may wait on two
// blocks, and a unit's latents are read in `issue` and written in `compute` of the SAME unit one step earlier - with other
// units in between, so a prefetch never overtakes the update it depends on as long as a tail owns more than one unit; with
// one unit it does not prefetch.
template <int MR, int WS>
__device__ __forceinline__ void tail_loop(const SysArgs& p, const Stage& st, TailRole<MR, WS, 0>& r, Ctl* ctl) {
    typename TailRole<MR, WS, 0>::Pay cur, nxt;
    typename TailRole<MR, WS, 0>::Geo gcur, gnxt;
    bool have = false;
    const int lane = threadIdx.x & 63, u0 = st.slice;
    const int nu = p.split ? p.NB / 2 : p.NB;
    const bool may_prefetch = u0 + NTAIL < nu;
    const int nflag = p.split ? 2 * st.wait_n : st.wait_n;          // lanes < wait_n: block blk_u, the next wait_n: block blk_c
    auto flag_ptr = [&](int u) {
        const int b = lane < st.wait_n ? r.blk_u(u) : r.blk_c(u);
        return (const gu32*)flag_of(p, st.wait_group, b, 0) + (lane < st.wait_n ? lane : lane - st.wait_n) * FLAG_STRIDE;
    };
    r.prime(nu);
    typename TailRole<MR, WS, 0>::Geo gnn;
    if (u0 < nu) { r.geo(u0, gcur); r.geo(u0 + NTAIL < nu ? u0 + NTAIL : u0, gnxt); gnn = gnxt; }
    for (int s = 0; s < p.n_steps; ++s)
        for (int u = u0; u < nu; u += NTAIL) {
            if (!have) {
                {                                                        // every wave of the four tails polls the flags of the unit's blocks
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    const gu32* f = flag_ptr(u);
                    for (unsigned spins = 1;; ++spins) {
                        unsigned v = 0xffffffffu;
                        if (lane < nflag) v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (__all(v >= (unsigned)(s + 1))) break;
                        if ((spins & 63u) == 0u) {
                            if (__hip_atomic_load((const gu32*)p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
                            if (__builtin_amdgcn_s_memrealtime() - t0 > p.timeout_ticks) {
                                if (lane == 0) {
                                    __hip_atomic_store((gu32*)p.status + 1, (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    __hip_atomic_store((gu32*)p.status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                }
                                return;
                            }
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                r.issue(s, u, gcur, cur);
            }
            int s2 = s, u2 = u + NTAIL;
            if (u2 >= nu) { s2 = s + 1; u2 = u0; }
            const bool has_next = may_prefetch && s2 < p.n_steps;
            unsigned fv = 0xffffffffu;
            if (has_next && threadIdx.x < 64 && lane < nflag) fv = __hip_atomic_load(flag_ptr(u2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (threadIdx.x < 64) {
                const int ok = has_next && __all(fv >= (unsigned)(s2 + 1));
                if (lane == 0) ctl->ready = ok;
            }
            __syncthreads();
            have = ctl->ready != 0;
            if (have) r.issue(s2, u2, gnxt, nxt);
            int s3 = s2, u3 = u2 + NTAIL;
            if (u3 >= nu) { s3 = s2 + 1; u3 = u0; }
            if (s3 < p.n_steps) r.geo(u3, gnn);                          // geometry two units ahead: its consumers never wait for it
            r.compute(s, u, gcur, cur);
            r.publish_unit(u, s + 2);
            if (have) cur = nxt;
            gcur = gnxt; gnxt = gnn;
        }
}

// Tagged hand-off: no barrier and no flag - every wave owns the (prompt, latent) pairs `wave + NW i` of its units, settles on the
// two rows of each pair, updates the latents and stores the next step's input rows with the next step's parity.
template <int MR, int WS, int HO>
__device__ __forceinline__ void tail_loop_tag(const SysArgs& p, const Stage& st, TailRole<MR, WS, HO>& r) {
    typename TailRole<MR, WS, HO>::Pay cur;
    typename TailRole<MR, WS, HO>::Geo gcur, gnxt, gnn;
    const int u0 = st.slice;
    const int nu = p.split ? p.NB / 2 : p.NB;
    r.prime(nu);
    if (u0 < nu) { r.geo(u0, gcur); r.geo(u0 + NTAIL < nu ? u0 + NTAIL : u0, gnxt); gnn = gnxt; }
    for (int s = 0; s < p.n_steps; ++s)
        for (int u = u0; u < nu; u += NTAIL) {
            r.issue(s, u, gcur, cur);
            int s2 = s, u2 = u + NTAIL;
            if (u2 >= nu) { s2 = s + 1; u2 = u0; }
            int s3 = s2, u3 = u2 + NTAIL;
            if (u3 >= nu) { s3 = s2 + 1; u3 = u0; }
            if (s3 < p.n_steps) r.geo(u3, gnn);                          // geometry two units ahead
#ifdef LADIFF_SELFCHECK
            aba_note(p, st, 128, 6, s, u, r.bad(s, gcur, cur));
#endif
            if (!__all((r.bad(s, gcur, cur) & TAG_MASK) == 0u)) {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                for (unsigned spins = 1;; ++spins) {
                    __builtin_amdgcn_s_sleep(1);
                    r.issue(s, u, gcur, cur);
#ifdef LADIFF_SELFCHECK
                    aba_note(p, st, 128, 7, s, u, r.bad(s, gcur, cur));
#endif
                    if (__all((r.bad(s, gcur, cur) & TAG_MASK) == 0u)) break;
                    if (spin_give_up(p, spins, t0)) return;
                }
            }
            r.compute(s, u, gcur, cur);
            gcur = gnxt; gnxt = gnn;
        }
}

}  // namespace

// WS = waves per SIMD of a stage workgroup: 1 = 256 threads (32-row blocks: 256 weight registers + two row tiles of everything else
// per wave), 2 = 512 threads with the stage's weight slice split over the two waves of a SIMD (16-row blocks)
// HO = hand-off protocol: 0 = flags (drain, barrier, epoch word, poll), 1 = parity tags in the data (see tag4)
// EQ = QKV's loader waves request the next block's rows early (QkvRole::split_loop<true>): the instantiation for launches in which rows
// queue up (SysArgs::look_ahead); a kernel of its own, so that the other launches run exactly the code without it
template <int MR, int AR, int WS, int HO, bool EQ = false>
__global__ __launch_bounds__(256 * WS, 1) void systolic_loop_kernel(const SysArgs p) {
    static_assert(WS == 1 || MR == 1, "two waves per SIMD: 16-row blocks only (the reduce parts' slot tables hold 12 slots: wave + 8 q needs q < 1)");
    // all LDS is dynamic (a static variable would shift the dynamic base off its 16-byte alignment, cdna_hip_programming.md G17)
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    Ctl* const ctl = reinterpret_cast<Ctl*>(lds + SYS_LDS_BYTES - 16);
    Stage st = p.stages[blockIdx.x];
    if (threadIdx.x == 0) {
        ctl->abort = 0; ctl->ready = 0; ctl->arrive = 0u; ctl->local_ok = 1;
        // the abort word is STICKY over the launches of one ladiff_diffusion_reverse call (a long schedule runs window by window:
        // the host clears it once per call, not per launch): after an abort the remaining windows end at once
        if (__hip_atomic_load((const gu32*)p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ctl->abort = 1;
        if (st.xcd >= 0 && !ctl->abort) {
            // The plan only needs workgroups i and j to share an XCD exactly when i = j (mod 8).  The dispatcher deals a launch's
            // workgroups to the XCDs round robin but starts where the previous launch stopped, so XCC_ID - i (mod 8) is one
            // number per launch: the first workgroup to get here records it, every other one compares - and ALL of them agree on
            // the outcome before anything is stored (the kernel needs all its workgroups resident anyway): were a single one
            // somewhere else (another queue's dispatch in between), the whole launch runs with write-through hand-offs, which
            // are right wherever a workgroup sits.  status[1] then reads -1 (ladiff_reverse_status: code 0, info -1).
            unsigned rot = ((__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf) + 8u - (blockIdx.x & 7u)) & 7u;
            if (p.force_mismatch && blockIdx.x == 5) rot = (rot + 1u) & 7u;
            unsigned seen = 0u;
            __hip_atomic_compare_exchange_strong((gu32*)p.status + 8, &seen, rot + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // one word carries both the head count (low half) and the number of workgroups that disagree (high half): a single
            // atomic per workgroup, relaxed polls (an acquire in the spin would invalidate the L2 on every turn)
            const unsigned mine = (seen != 0u && seen != rot + 1u) ? 0x10001u : 1u;
            __hip_atomic_fetch_add((gu32*)p.status + 9, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            unsigned v = 0u;
            while (((v = __hip_atomic_load((const gu32*)p.status + 9, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0xffffu) < gridDim.x) {
                if (__builtin_amdgcn_s_memrealtime() - t0 > p.timeout_ticks) {
                    ctl->abort = 1;
                    __hip_atomic_store((gu32*)p.status + 1, (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store((gu32*)p.status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if ((v >> 16) != 0u) {
                ctl->local_ok = 0;
                if (blockIdx.x == 0) __hip_atomic_store((gu32*)p.status + 1, 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    __syncthreads();
    if (ctl->abort) return;
    if ((int)blockIdx.x == p.fault_wg) return;       // injected fault (ladiff_debug_set_pipeline_fault): its consumers time out
    if (!ctl->local_ok) st.out_local = 0;
#ifdef LADIFF_STAMPS
    if (p.stamps != nullptr && threadIdx.x == 0)         // who runs here (the stage table is permuted by the XCD placement)
        p.stamps[(size_t)256 * 4 * 4 * 8 + 256 * 4 + 256 * 4 * 8 + blockIdx.x] =
            1ull | (unsigned long long)st.role << 8 | (unsigned long long)st.layer << 16 | (unsigned long long)st.slice << 24 |
            (unsigned long long)st.blk0 << 32 | (unsigned long long)(st.out_local & 1) << 40 | (unsigned long long)(st.xcd & 0xff) << 48;
#endif
    // HO of the roles: 0 = flags, 1 / 2 = tags with write-through / plain stores (st_out): the store form of a tagged stage is a
    // compile-time property of its loop
    auto with_ho = [&](auto&& f) {
        if constexpr (HO) { if (st.out_local) f(IntC<2>{}); else f(IntC<1>{}); }
        else f(IntC<0>{});
    };
    auto run = [&](auto& r) {
        if constexpr (HO) tag_loop(p, st, r, st.blk0, st.blkstride);
        else stage_loop(p, st, r, ctl, st.blk0, st.blkstride);
    };
    switch (st.role) {
        case R_QKV:
            with_ho([&](auto hc) {
                QkvRole<MR, AR, WS, decltype(hc)::value> r(p, st, lds);
                if constexpr (WS == 2) r.template split_loop<EQ>(ctl);
                else run(r);
            });
            break;
        case R_OUT:
            with_ho([&](auto hc) {
                OutRole<MR, AR, WS, decltype(hc)::value> r(p, st, lds);
                if constexpr (WS == 2) r.split_loop(ctl);
                else run(r);
            });
            break;
        case R_LIN: with_ho([&](auto hc) { MlpRole<MR, ACT_RELU, AR, WS, decltype(hc)::value> r(p, st, lds); run(r); }); break;
        case R_RED2: with_ho([&](auto hc) { Red2Role<MR, WS, decltype(hc)::value> r(p, st, lds); run(r); }); break;
        case R_FFN: with_ho([&](auto hc) { MlpRole<MR, ACT_GELU, AR, WS, decltype(hc)::value> r(p, st, lds); run(r); }); break;
        case R_STYL: with_ho([&](auto hc) { StylRole<MR, AR, WS, decltype(hc)::value> r(p, st, lds); run(r); }); break;
        case R_SKIP: with_ho([&](auto hc) { SkipRole<MR, AR, WS, decltype(hc)::value> r(p, st, lds); run(r); }); break;
        case R_TAIL:
            with_ho([&](auto hc) {
                TailRole<MR, WS, decltype(hc)::value> r(p, st, lds);
                if constexpr (HO) tail_loop_tag(p, st, r);
                else tail_loop(p, st, r, ctl);
            });
            break;
        default: break;
    }
}

// ================================================================== host side
#ifdef LADIFF_STAMPS
unsigned long long* g_sys_stamps = nullptr;
int g_sys_probe = 0;                                                   // diagnostic twin build: SysArgs::probe
#endif
namespace {
struct SysLayout {
    size_t blk, ring;             // floats of one [NB][RT][256] buffer / of one [PRING][RT][256] partial plane
    size_t off_stages, off_blocks, off_flags, off_status, off_xin0, off_xs, off_xo, off_att, off_x1, off_x2, off_pc, off_pe, total;
    int nwg, NB, split;
};
constexpr int NWG = NL * (4 + 1 + NSLICE + NRED + NSLICE + NRED) + 2 * NSKIP + NTAIL;     // the largest plan
int plan_nwg(int MR) {
    const RedPlan rp = red_plan(MR);
    return NL * (4 + rp.out_groups + NSLICE + rp.red2_parts + NSLICE + rp.styl_parts * rp.styl_groups) + 2 * NSKIP + NTAIL;
}
// 32-row tiles: P prompts per block, both guidance branches, T rows each
int prompts_per_block32(int T) { int P = 32 / (2 * T); return P > 7 ? 7 : (P < 1 ? 1 : P); }   // QKV parks <= 14 text K|V slots
int nb32(int B, int T) { const int P = prompts_per_block32(T); return (B + P - 1) / P; }
// 16-row tiles, worst case of the packing (every prompt with all T rows): floor(16 / T) prompts (<= 8) per branch block
int nb16_max(int B, int T) { int P = 16 / T; P = P > 8 ? 8 : (P < 1 ? 1 : P); return 2 * ((B + P - 1) / P); }

SysLayout sys_layout(int MR, int NB) {
    SysLayout L;
    const int RT = 16 * MR;
    L.split = MR == 1 ? 1 : 0;
    L.NB = NB;
    L.nwg = plan_nwg(MR);
    L.blk = (size_t)NB * RT * D;
    size_t off = 0;
    auto take = [&](size_t floats) { const size_t o = off; off += (floats + 63) / 64 * 64; return o; };
    L.off_stages = take((size_t)256 * sizeof(Stage) / sizeof(float));
    L.off_status = take(64);      // before everything sized by the block geometry: ladiff_reverse_status reads it at a fixed offset
    L.off_blocks = take((size_t)NB * sizeof(BlockDesc) / sizeof(float));
    L.off_flags = take((size_t)NL * GROUPS_PER_LAYER * NB * FLAG_SLOTS * FLAG_STRIDE);
    L.off_xin0 = take(L.blk);
    L.off_xs = take(NSKIP * L.blk);
    L.off_xo = take(NL * L.blk);
    // one set per LAYER: a buffer is then written by the workgroups of one stage group only - on one XCD when the group stores
    // plainly (Stage::out_local), so that no line is ever dirty in two L2s
    L.off_att = take(NL * L.blk);
    L.off_x1 = take(NL * L.blk);
    L.off_x2 = take(NL * L.blk);
    L.ring = (size_t)PRING * RT * D;                                  // floats of one partial plane: a ring of PRING block slots
    L.off_pc = take((size_t)NL * NSLICE * L.ring);
    L.off_pe = take((size_t)NL * NSLICE * L.ring);
    L.total = off;
    return L;
}
}  // namespace

size_t sys_ws_floats(int B, int T) {
    const size_t a = sys_layout(1, nb16_max(B, T)).total, b = sys_layout(2, nb32(B, T)).total;
    return a > b ? a : b;
}

bool sys_supported(int B, int T, int cfg, bool split) {
    (void)split; (void)cfg;       // both arithmetic modes and both guidance settings have a pipeline form
    if (B < 1 || T < 1 || T > LADIFF_MAX_LATENTS) return false;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    return cus >= NWG;            // every stage needs a CU of its own, all resident at once
}

// Block geometry for this call (host).  h_counts = the latent counts on the HOST (or NULL); masked = the call has device counts.
//   MR 2: blocks of P consecutive prompts, both branches, T rows per prompt; the count only masks keys (cnt = -1 when the host
//         does not know it: the kernel reads counts[] itself).
//   MR 1: LENGTH-AWARE packing - prompts sorted by latent count, a block = one guidance branch of as many prompts as fit in 16
//         rows with ONLY their count[b] valid rows (padded latent rows are never computed); needs the counts on the host.
// Returns the plan: `blocks`, and in `mr` the tile size actually planned (MR 1 falls back to 2 when the counts are device-only).
// cfg = false (no classifier-free guidance, ladiff.py:472-490: the network sees the B latents once): 16-row blocks of ONE branch, no
// partner block - the tail treats a block as a unit whose "conditional" row is the row itself (guidance then adds g * 0 exactly).
void sys_pack_blocks(int B, int T, int want_mr, const int32_t* h_counts, bool masked, bool cfg, std::vector<unsigned char>& out, int* mr, int* nb) {
    std::vector<BlockDesc> blocks;
    auto count_of = [&](int b) { int c = (masked && h_counts) ? h_counts[b] : T; return c > T ? T : (c < 1 ? 1 : c); };
    int MR = want_mr;
    if (!cfg) MR = 1;                                                  // the caller checked sys_plan_possible()
    if (MR == 1 && masked && h_counts == nullptr && cfg) MR = 2;
    auto fresh = [] { BlockDesc d; std::memset(&d, 0, sizeof(d)); for (int i = 0; i < 16; ++i) d.b2[i] = -1;
                      for (int r = 0; r < 32; ++r) { d.row_b2[r] = -1; d.row_lat[r] = -1; } return d; };
    // derived tables: the reduce parts' slots (the live rows split evenly over NRED parts) and the tail's (prompt, latent) pairs
    auto finish = [&](BlockDesc& d, const int* row_cnt, int npairs, int rc_off) {
        const int nparts = red_plan(MR).red2_parts;               // RED2 and STYL split a block's rows the same way
        const int RT = 16 * MR, per = (RT + nparts - 1) / nparts; // all rows of the tile, padding included (stored as zeros)
        for (int part = 0; part < NRED; ++part) {
            const int lo = part * per < RT ? part * per : RT, hi = lo + per < RT ? lo + per : RT;
            for (int k = 0; k < 12; ++k) {
                const int r = lo + k;
                d.part_pk[part][k] = -1; d.part_b2[part][k] = -1;
                if (part < nparts && r < hi) {
                    if (r < d.nrows) {
                        d.part_pk[part][k] = r | d.row_t[r] << 8 | (row_cnt[r] < 0 ? 0xff : row_cnt[r]) << 16;
                        d.part_b2[part][k] = d.row_b2[r];
                    } else {
                        d.part_pk[part][k] = PART_PAD | r;
                    }
                }
            }
        }
        // tail: pairs first; the slots behind them pad.  One-branch blocks (rc_off 0): slot q pads row q of both blocks of the unit;
        // two-branch blocks: the pad slots share the rows from nrows up, two each (both in the unit's one block)
        for (int q = 0; q < 16; ++q) {
            d.pair_lat[q] = -1; d.pair_t[q] = 0; d.pair_rc[q] = -1; d.pair_pad[q] = -1;
            if (q < npairs) { d.pair_lat[q] = d.row_lat[q]; d.pair_t[q] = d.row_t[q]; d.pair_rc[q] = q + rc_off; }
            else if (rc_off == 0) { if (q < RT) { d.pair_pad[q] = q; d.pair_rc[q] = q; } }
            else {
                const int r = d.nrows + 2 * (q - npairs);
                if (r < RT) d.pair_pad[q] = r;
                if (r + 1 < RT) d.pair_rc[q] = r + 1;
            }
        }
    };
    if (MR == 2) {
        const int P = prompts_per_block32(T);
        for (int p0 = 0; p0 < B; p0 += P) {
            const int Pb = B - p0 < P ? B - p0 : P;
            BlockDesc d = fresh();
            d.nsb = 2 * Pb; d.nrows = 2 * Pb * T;
            int row_cnt[32];
            for (int br = 0; br < 2; ++br)
                for (int pl = 0; pl < Pb; ++pl) {
                    const int sx = br * Pb + pl, prompt = p0 + pl;
                    const int cnt = masked ? (h_counts ? count_of(prompt) : -1) : T;
                    d.b2[sx] = br * B + prompt;
                    for (int t = 0; t < T; ++t) {
                        const int r = sx * T + t;
                        d.row_pk[r] = sx | (sx * T) << 8 | (cnt < 0 ? 0xff : cnt) << 16;
                        d.row_t[r] = t; d.row_b2[r] = br * B + prompt; row_cnt[r] = cnt; d.row_lat[r] = prompt * T + t;
                    }
                }
            finish(d, row_cnt, d.nrows / 2, d.nrows / 2);                // conditional row of a pair: nrows / 2 further
            blocks.push_back(d);
        }
    } else {
        std::vector<int> order(B);
        for (int b = 0; b < B; ++b) order[b] = b;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return count_of(a) > count_of(b); });
        size_t i = 0;
        while (i < order.size()) {
            std::vector<int> group;
            int rows = 0;
            while (i < order.size() && group.size() < 8 && rows + count_of(order[i]) <= 16) { rows += count_of(order[i]); group.push_back(order[i]); ++i; }
            for (int br = 0; br < (cfg ? 2 : 1); ++br) {
                BlockDesc d = fresh();
                d.nsb = (int)group.size(); d.nrows = rows;
                int r0 = 0, row_cnt[32];
                for (int sx = 0; sx < (int)group.size(); ++sx) {
                    const int prompt = group[sx], cnt = count_of(prompt);
                    d.b2[sx] = br * B + prompt;
                    for (int t = 0; t < cnt; ++t) {
                        const int r = r0 + t;
                        d.row_pk[r] = sx | r0 << 8 | cnt << 16;
                        d.row_t[r] = t; d.row_b2[r] = br * B + prompt; row_cnt[r] = cnt; d.row_lat[r] = prompt * T + t;
                    }
                    r0 += cnt;
                }
                finish(d, row_cnt, rows, 0);                             // the conditional branch is the next block, same row
                blocks.push_back(d);
            }
        }
    }
    out.resize(blocks.size() * sizeof(BlockDesc));
    std::memcpy(out.data(), blocks.data(), out.size());
    *mr = MR; *nb = (int)blocks.size();
}

// Builds the stage table (host) for this call's pointers.  `ws` = the systolic region of the reverse workspace.
// ---- XCD placement.  The dispatcher hands workgroup i of a launch to XCD i % 8 (from wherever the previous launch
// stopped; checked once per device by a probe launch of the same shape, and by every pipeline workgroup when it starts: status 3).  `st` comes in CHAIN order (the order a block
// flows through the stages); the stages are dealt to the XCDs in that order, 32 (31) to each, so that a layer's hand-offs stay
// inside one XCD's L2 and the chain crosses an XCD boundary only 7 times (+ the skip connections and the tail).  A stage whose
// readers all sit on its own XCD stores plainly (Stage::out_local); everything else works as before (write-through).
std::atomic<int> g_xcd_local{1};  // measurement switch: ladiff_debug_set_xcd_local

__global__ __launch_bounds__(512, 1) void xcd_probe_kernel(unsigned* xcc) {
    extern __shared__ char lds[];
    if (threadIdx.x == 0) xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;
}
static bool xcd_round_robin() {
    static std::mutex mu;
    static int known[64] = {};    // per device: 0 unknown, 1 round robin over 8 XCDs, 2 anything else
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    std::lock_guard<std::mutex> lock(mu);
    if (known[dev] == 0) {
        known[dev] = 2;
        unsigned* d = nullptr;
        unsigned h[NWG];
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess && hipMalloc(&d, sizeof(h)) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(xcd_probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SYS_LDS_BYTES) == hipSuccess) {
            hipLaunchKernelGGL(xcd_probe_kernel, dim3(NWG), dim3(512), SYS_LDS_BYTES, s, d);
            if (hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess) {
                bool ok = true;
                for (int i = 0; i < NWG; ++i) ok = ok && h[i] < 8u && h[i] == (h[0] + (unsigned)i) % 8u;   // any start, then round robin
                if (ok) known[dev] = 1;
            }
        }
        if (d) (void)hipFree(d);
        if (s) (void)hipStreamDestroy(s);
    }
    return known[dev] == 1;
}
static void sys_place_stages(std::vector<Stage>& st) {
    const int n = (int)st.size();
    std::vector<Stage> placed(n);
    int cnt[8] = {}, k = 0;
    for (int j = 0; j < n; ++j) {
        while (cnt[k] == (n - k + 7) / 8) ++k;                        // XCD k runs workgroups k, k + 8, ...
        st[j].xcd = k;
        placed[k + 8 * cnt[k]++] = st[j];
    }
    for (Stage& p : placed) {
        bool local = true;
        for (const Stage& c : placed) {
            const bool reads = c.in0 == p.out || c.in1 == p.out || c.in2 == p.out;                       // its rows
            const bool polls = c.wait_group == p.out_group || (c.bp_n > 0 && c.bp_group == p.out_group);  // its flags
            if ((reads || polls) && c.xcd != p.xcd) local = false;
        }
        p.out_local = local ? 1 : 0;
    }
    st.swap(placed);
}

int sys_build_stages(const DenoiserW& W, const DenoiserW& WS, float* ws, int MR, int NB, std::vector<unsigned char>& host) {
    // WS = the S-format weight table in f16x3 mode; in fp32 mode the caller passes the fp32 table twice
    const SysLayout L = sys_layout(MR, NB);
    const RedPlan rp = red_plan(MR);
    std::vector<Stage> st;
    float* xin0 = ws + L.off_xin0;
    auto XO = [&](int l) { return ws + L.off_xo + (size_t)l * L.blk; };
    auto XS = [&](int l) { return ws + L.off_xs + (size_t)(l - NSKIP - 1) * L.blk; };
    auto G = [&](int l, int g) { return l * GROUPS_PER_LAYER + g; };
    const bool x2_rep = rp.red2_parts * NSLICE <= FLAG_SLOTS;
    for (int l = 0; l < NL; ++l) {
        const DenLayerW& w = W.layer[l];
        const DenLayerW& ws_ = WS.layer[l];
        float* att = ws + L.off_att + (size_t)l * L.blk; float* x1 = ws + L.off_x1 + (size_t)l * L.blk; float* x2 = ws + L.off_x2 + (size_t)l * L.blk;
        float* pc = ws + L.off_pc + (size_t)l * NSLICE * L.ring; float* pe = ws + L.off_pe + (size_t)l * NSLICE * L.ring;
        const float* xin; int xg, xn;
        if (l == 0) { xin = xin0; xg = G(0, G_XIN); xn = 1; }
        else if (l <= NSKIP) { xin = XO(l - 1); xg = G(l - 1, G_XO); xn = rp.styl_parts; }
        else { xin = XS(l); xg = G(l, G_XIN); xn = 2; }
        if (l > NSKIP) {
            const int i = l - NSKIP - 1;
            for (int c = 0; c < 2; ++c) {
                Stage s{};
                s.role = R_SKIP; s.layer = l; s.slice = c; s.wait_group = G(l - 1, G_XO); s.wait_n = rp.styl_parts;
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
        for (int g = 0; g < rp.out_groups; ++g) {                         // groups: workgroups of their own on alternating blocks
            Stage s{};
            s.role = R_OUT; s.layer = l; s.wait_group = G(l, G_ATT); s.wait_n = H; s.out_group = G(l, G_X1); s.out_slot = 0;
            s.out_rep = NSLICE; s.out_rep_stride = 1;                     // one flag line per LIN workgroup
            s.blk0 = g; s.blkstride = rp.out_groups;
            s.w0 = ws_.sa_attn.out_w; s.b0 = w.sa_attn.out_b; s.g = w.sa_norm1.g; s.be = w.sa_norm1.b; s.in0 = att; s.in1 = xin; s.out = x1;
            st.push_back(s);
        }
        for (int j = 0; j < NSLICE; ++j) {
            Stage s{};
            s.role = R_LIN; s.layer = l; s.slice = j; s.wait_group = G(l, G_X1); s.wait_n = 1; s.out_group = G(l, G_PC); s.out_slot = j;
            s.wait_slot0 = j;
            s.bp_group = G(l, G_X2); s.bp_slot0 = x2_rep ? j * rp.red2_parts : 0; s.bp_n = rp.red2_parts; s.bp_blocks = 1; s.bp_buf = x2;
            s.w0 = ws_.sa_lin1.w; s.w1 = ws_.sa_lin2.w; s.b0 = w.sa_lin1.b; s.in0 = x1; s.out = pc;
            st.push_back(s);
        }
        for (int q = 0; q < rp.red2_parts; ++q) {
            Stage s{};
            s.role = R_RED2; s.layer = l; s.slice = q; s.wait_group = G(l, G_PC); s.wait_n = NSLICE; s.out_group = G(l, G_X2); s.out_slot = q;
            if (x2_rep) { s.out_rep = NSLICE; s.out_rep_stride = rp.red2_parts; }   // FFN workgroup j polls slots j parts + q
            s.b0 = w.sa_lin2.b; s.g = w.sa_norm2.g; s.be = w.sa_norm2.b; s.in0 = pc; s.in1 = x1; s.out = x2;
            st.push_back(s);
        }
        for (int j = 0; j < NSLICE; ++j) {
            Stage s{};
            s.role = R_FFN; s.layer = l; s.slice = j; s.wait_group = G(l, G_X2); s.wait_n = rp.red2_parts; s.out_group = G(l, G_PE); s.out_slot = j;
            if (x2_rep) s.wait_slot0 = j * rp.red2_parts;
            s.bp_group = G(l, G_XO); s.bp_slot0 = 0; s.bp_n = rp.styl_parts; s.bp_blocks = rp.styl_groups; s.bp_buf = XO(l);
            s.w0 = ws_.ffn1.w; s.w1 = ws_.ffn2.w; s.b0 = w.ffn1.b; s.in0 = x2; s.out = pe;
            st.push_back(s);
        }
        for (int g = 0; g < rp.styl_groups; ++g)
            for (int q = 0; q < rp.styl_parts; ++q) {
                Stage s{};
                s.role = R_STYL; s.layer = l; s.slice = q; s.wait_group = G(l, G_PE); s.wait_n = NSLICE; s.out_group = G(l, G_XO); s.out_slot = q;
                s.blk0 = g; s.blkstride = rp.styl_groups;
                s.w0 = ws_.ffn_proj.out.w; s.b0 = w.ffn_proj.out.b; s.b1 = w.ffn2.b; s.g = w.ffn_proj.norm.g; s.be = w.ffn_proj.norm.b;
                s.in0 = pe; s.in1 = x2; s.out = XO(l);
                st.push_back(s);
            }
    }
    for (int k = 0; k < NTAIL; ++k) {
        Stage s{};
        s.role = R_TAIL; s.layer = NL; s.slice = k; s.wait_group = G(NL - 1, G_XO); s.wait_n = rp.styl_parts; s.out_group = G(0, G_XIN); s.out_slot = 0;
        s.in0 = XO(NL - 1); s.out = xin0;
        st.push_back(s);
    }
    for (Stage& s : st) {
        if (s.blkstride == 0) s.blkstride = 1;                        // every other stage visits every block
        if (s.out_rep == 0) s.out_rep = 1;                            // one flag, one line
        s.xcd = -1;
    }
    if ((int)st.size() != L.nwg || st.size() > 256) return LADIFF_ERR_SHAPE;
    if (g_xcd_local && xcd_round_robin()) sys_place_stages(st);
    host.resize(st.size() * sizeof(Stage));
    std::memcpy(host.data(), st.data(), host.size());
    return 0;
}

// One launch = local steps [step_lo, step_lo + n) of the loop on the latents in `lat`.  The stage table must already be in
// the workspace (sys_upload_stages); `ctab` holds the hoisted cross-attention rows of n_ctab >= n steps starting at step_lo.
// waves per SIMD of the 16-row plan's stage workgroups (measurement switch: ladiff_debug_set_stage_waves)
std::atomic<int> g_waves16{2};
// hand-off protocol of the 16-row plan's eight-wave stages: 1 = parity tags in the data (default), 0 = flags (ladiff_debug_set_handoff)
std::atomic<int> g_handoff{1};
constexpr int LOOK_AHEAD_BLOCKS = 72;      // see `settle`: 58 blocks (128 prompts of mixed lengths) lose with it, 86 win

int sys_reset_status(float* ws, hipStream_t s) {
    const SysLayout L = sys_layout(2, 1);            // the status words sit at a fixed offset (before everything sized by the plan)
    LADIFF_HIP(hipMemsetAsync(ws + L.off_status, 0, 64 * sizeof(float), s));
    return 0;
}

int launch_systolic_loop(const DenoiserW& W, float* ws, const float* tables, const float* tkv, const float* ctab, int n_ctab,
                         const float* coef, const float* noise, float* lat, const int32_t* counts, float gscale, int B, int T,
                         int step_lo, int n, int fp32, int MR, int NB, hipStream_t s, int cfg, int fault_wg, unsigned long long timeout_ticks,
                         const NoiseGen& gen) {
    // fault_wg / timeout_ticks: the caller's SAMPLER's fault injection (ladiff_sampler_set_fault; -1 / 0 = none / default bound)
    const SysLayout L = sys_layout(MR, NB);
    SysArgs a;
    a.stages = reinterpret_cast<const Stage*>(ws + L.off_stages);
    a.blocks = reinterpret_cast<const BlockDesc*>(ws + L.off_blocks);
    a.flags = reinterpret_cast<unsigned*>(ws + L.off_flags);
    a.status = reinterpret_cast<unsigned*>(ws + L.off_status);
    a.tables = tables; a.tkv = tkv; a.ctab = ctab; a.coef = coef; a.noise = noise; a.pe = W.query_pe; a.ng = W.norm.g; a.nb = W.norm.b;
    a.lat = lat; a.counts = counts; a.gscale = gscale; a.B = B; a.B2 = cfg ? 2 * B : B; a.T = T; a.P = 0; a.NB = NB; a.step_lo = step_lo; a.n_steps = n;
    a.n_ctab = n_ctab;
    a.split = cfg ? L.split : 0;       // no guidance: every block is a unit of its own (sys_pack_blocks)
    a.force_mismatch = g_xcd_local == 2 ? 1 : 0;
    a.fault_wg = fault_wg;
    a.timeout_ticks = timeout_ticks > 0 ? timeout_ticks : TIMEOUT_TICKS;
    const int la_from = g_look_ahead_from.load() >= 0 ? g_look_ahead_from.load() : LOOK_AHEAD_BLOCKS;      // ladiff_debug_set_loop_thresholds
    const int small_upto = g_small_upto.load() >= 0 ? g_small_upto.load() : SMALL_LAUNCH_BLOCKS;
    a.look_ahead = NB >= la_from ? 1 : 0;
    a.gen = gen;
    a.pause_mask = g_poll_pause.load() & 0xff; a.pause_len = (g_poll_pause.load() >> 8) & 0xff;
    a.pace = g_pace.load() >= 0 ? g_pace.load() : (NB <= small_upto ? 0 : (4 | (4 << 8)));
    // Small launches (a block's trip through the stages bounds the step, the stages wait for rows most of the time): the LIN and FFN
    // workgroups idle ~0.25 us after every block before they start to poll for the next one's rows - those are being produced just then by
    // a stage on the critical path (OUT / RED2), and eight workgroups loading its lines do not make it faster.  Measured (profiles/r4/14_*):
    // loop -2.2 % at 32 ... 64 prompts and at 100 / 128 prompts of mixed lengths (<= 58 blocks), nothing from 65 blocks, a loss beyond.
    const int sd = g_stage_delay.load();
    if (sd < 0) { a.delay_mask = NB <= small_upto ? 1 | 8 : 0; a.delay_len = 4; }
    else { a.delay_mask = sd & 0xff; a.delay_len = (sd >> 8) & 0x7f; }
    a.probe = 0;
    a.stamps = nullptr;
#ifdef LADIFF_STAMPS
    a.stamps = g_sys_stamps;
    a.probe = g_sys_probe;
#endif
    // A pipeline kernel needs its 255 workgroups resident at once: two of them launched on different streams of one process
    // could each hold part of the chip and wait for the rest (they would time out, not hang, but the results are lost).  Launches are
    // therefore chained through one event per device: the next one, on whatever stream, starts after the previous one has ended.
    static std::mutex mu;
    static hipEvent_t done[64] = {};
    static bool attr_set[64] = {};             // hipFuncSetAttribute is per DEVICE: a process driving several GPUs opts in on each
    int dev = 0;
    LADIFF_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return LADIFF_ERR_ARG;
    std::lock_guard<std::mutex> lock(mu);
    if (!attr_set[dev]) {
        const void* k[10] = {reinterpret_cast<const void*>(systolic_loop_kernel<1, 0, 2, 0>), reinterpret_cast<const void*>(systolic_loop_kernel<2, 0, 1, 0>),
                             reinterpret_cast<const void*>(systolic_loop_kernel<1, 1, 2, 0>), reinterpret_cast<const void*>(systolic_loop_kernel<2, 1, 1, 0>),
                             reinterpret_cast<const void*>(systolic_loop_kernel<1, 0, 1, 0>), reinterpret_cast<const void*>(systolic_loop_kernel<1, 1, 1, 0>),
                             reinterpret_cast<const void*>(systolic_loop_kernel<1, 0, 2, 1>), reinterpret_cast<const void*>(systolic_loop_kernel<1, 1, 2, 1>),
                             reinterpret_cast<const void*>(systolic_loop_kernel<1, 0, 2, 1, true>), reinterpret_cast<const void*>(systolic_loop_kernel<1, 1, 2, 1, true>)};
        for (int i = 0; i < 10; ++i) LADIFF_HIP(hipFuncSetAttribute(k[i], hipFuncAttributeMaxDynamicSharedMemorySize, SYS_LDS_BYTES));
        attr_set[dev] = true;
    }
    if (done[dev] == nullptr) LADIFF_HIP(hipEventCreateWithFlags(&done[dev], hipEventDisableTiming));
    else LADIFF_HIP(hipStreamWaitEvent(s, done[dev], 0));
    // Per launch: the placement handshake words ([8], [9]) and the flags.  The abort / diagnostic words ([0], [1]) are cleared ONCE per
    // ladiff_diffusion_reverse call (sys_reset_status): a schedule longer than one window is several launches, and an abort in an
    // early window must neither be erased by the next launch's memset nor let the later windows iterate on corrupt latents.
    LADIFF_HIP(hipMemsetAsync(a.status + 8, 0, 2 * sizeof(unsigned), s));
    const bool tagged = MR == 1 && g_waves16 == 2 && g_handoff == 1;
    if (tagged) {
        // tagged hand-off: every word of the hand-off buffers starts with parity 1 (the first use of every slot expects 0)
        LADIFF_HIP(hipMemsetAsync(ws + L.off_xin0, 0x01, (L.total - L.off_xin0) * sizeof(float), s));
    } else {
        LADIFF_HIP(hipMemsetAsync(a.flags, 0, (L.off_xin0 - L.off_flags) * sizeof(float), s));
    }
    if (fp32) {
        if (tagged && a.look_ahead) hipLaunchKernelGGL((systolic_loop_kernel<1, 1, 2, 1, true>), dim3(L.nwg), dim3(512), SYS_LDS_BYTES, s, a);
        else if (tagged) hipLaunchKernelGGL((systolic_loop_kernel<1, 1, 2, 1>), dim3(L.nwg), dim3(512), SYS_LDS_BYTES, s, a);
        else if (MR == 1 && g_waves16 == 2) hipLaunchKernelGGL((systolic_loop_kernel<1, 1, 2, 0>), dim3(L.nwg), dim3(512), SYS_LDS_BYTES, s, a);
        else if (MR == 1) hipLaunchKernelGGL((systolic_loop_kernel<1, 1, 1, 0>), dim3(L.nwg), dim3(256), SYS_LDS_BYTES, s, a);
        else hipLaunchKernelGGL((systolic_loop_kernel<2, 1, 1, 0>), dim3(L.nwg), dim3(256), SYS_LDS_BYTES, s, a);
    } else {
        if (tagged && a.look_ahead) hipLaunchKernelGGL((systolic_loop_kernel<1, 0, 2, 1, true>), dim3(L.nwg), dim3(512), SYS_LDS_BYTES, s, a);
        else if (tagged) hipLaunchKernelGGL((systolic_loop_kernel<1, 0, 2, 1>), dim3(L.nwg), dim3(512), SYS_LDS_BYTES, s, a);
        else if (MR == 1 && g_waves16 == 2) hipLaunchKernelGGL((systolic_loop_kernel<1, 0, 2, 0>), dim3(L.nwg), dim3(512), SYS_LDS_BYTES, s, a);
        else if (MR == 1) hipLaunchKernelGGL((systolic_loop_kernel<1, 0, 1, 0>), dim3(L.nwg), dim3(256), SYS_LDS_BYTES, s, a);
        else hipLaunchKernelGGL((systolic_loop_kernel<2, 0, 1, 0>), dim3(L.nwg), dim3(256), SYS_LDS_BYTES, s, a);
    }
    LADIFF_LAUNCH_CHECK();
    LADIFF_HIP(hipEventRecord(done[dev], s));
    return 0;
}

size_t sys_blocks_offset_floats(int MR, int NB) { return sys_layout(MR, NB).off_blocks; }
size_t sys_status_offset_floats(int B, int T) { (void)B; (void)T; return sys_layout(2, 1).off_status; }

}  // namespace ladiff
