// K-resident fp32 MFMA GEMM for the small-M (denoiser) side of the path:  Y[M,N] = epi( pro(A)[M,K] . W[N,K]^T )
//
// The denoiser works on M = 2B*T = 1280 rows per step, so a GEMM is ~0.2-0.7 GFLOP: one tile per CU and a few
// microseconds of MFMA.  Measured on MI355X (profiles/r1): after every kernel boundary the operands come from the
// Infinity Cache at ~33 GB/s per CU, so a workgroup's life is launch (~2 us) + fill + MFMA, and the fill is as long as
// the MFMA.  This kernel therefore (1) moves the minimum bytes per workgroup - one 256-wide K slice of a tile that
// gives ~256 workgroups (80x64 for N=1024, 64x64 for N=768, 32x32 for N=256), with split-K over blockIdx.y for the
// K=1024/512 GEMMs (partials are combined by reduce_rows_kernel, which also applies the LayerNorm that follows them);
// (2) brings its operands by LDS-DMA (`global_load_lds_dwordx4`: no VGPRs), two of the four 64-wide K sub-slices in flight
// per workgroup, ONE per wave (each wait is `vmcnt(0)`: LDS-DMA requests do not complete in issue order, gemm_big.hip);
// (3) starts the MFMAs after the first sub-slice has landed (raw `s_barrier`), the others stream in underneath.
//
// LDS image: [4 sub-slices][BM + BN rows][64 floats] (256-byte rows = one LDS bank row).  One LDS-DMA instruction
// writes 1 KiB = 4 consecutive rows of one sub-slice, lane i -> row i/16, 16-byte slot i%16.  The slot is XOR-swizzled
// with (row & 15) ON THE SOURCE ADDRESS (the LDS side of an LDS-DMA is always linear), and the fragment reads apply the
// same XOR, which makes the ds_read_b128 of both MFMA shapes bank-conflict free.
//
#include "gemm_kr.h"

#include <type_traits>

namespace ladiff {

namespace {

template <int MT> struct Mf;
template <> struct Mf<32> {
    typedef f32x16 Acc;
    static constexpr int REGS = 16, KG = 8;
    __device__ static __forceinline__ Acc mma(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
    __device__ static __forceinline__ int frow(int l) { return l & 31; }
    __device__ static __forceinline__ int fk(int l) { return l >> 5; }
    __device__ static __forceinline__ int arow(int l, int r) { return (r & 3) + 8 * (r >> 2) + 4 * (l >> 5); }
    __device__ static __forceinline__ int acol(int l) { return l & 31; }
};
template <> struct Mf<16> {
    typedef f32x4 Acc;
    static constexpr int REGS = 4, KG = 16;
    __device__ static __forceinline__ Acc mma(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    __device__ static __forceinline__ int frow(int l) { return l & 15; }
    __device__ static __forceinline__ int fk(int l) { return l >> 4; }
    __device__ static __forceinline__ int arow(int l, int r) { return 4 * (l >> 4) + r; }
    __device__ static __forceinline__ int acol(int l) { return l & 15; }
};

#ifdef LADIFF_STAMPS
#define STAMP(i)                                                                                      \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        unsigned long long t_;                                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        if (p.stamps != nullptr && tid == 0) p.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = t_; \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void glds16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

}  // namespace

// fp32 mode (fp32-input MFMA); the f16x3 mode has its own kernel below.
template <int BM, int BN, int WM, int WN, int MT>
__global__ __launch_bounds__(WM * WN * 64) void gemm_kr_kernel(const KrArgs p) {
    typedef Mf<MT> MM;
    typedef typename MM::Acc Acc;
    constexpr int NW = WM * WN;
    constexpr int TMW = BM / WM, TNW = BN / WN;
    constexpr int RM = TMW / MT, RN = TNW / MT;
    constexpr int ROWS = BM + BN;                  // LDS rows of one sub-slice (A rows then W rows)
    constexpr int SUB = ROWS * 64;                 // floats per sub-slice
    constexpr int PCS = ROWS / 4;                  // 1-KiB pieces per sub-slice
    static_assert(NW == 4, "four waves per workgroup");
    static_assert(PCS % NW == 0, "pieces must split evenly over the waves");
    static_assert(PCS / 2 <= 63, "vmcnt is 6 bits (a wave has one sub-slice = PCS / 2 pieces in flight)");

    __shared__ __attribute__((aligned(1024))) float lds[4 * SUB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nbn = (p.N + BN - 1) / BN;
    const int bm = blockIdx.x / nbn, bn = blockIdx.x % nbn;
    const int ks = blockIdx.y;                     // K slice (split-K)
    const int row0 = bm * BM, col0 = bn * BN;
    const bool partial = gridDim.y > 1;            // split-K: raw partial sums, combined by reduce_rows_kernel
    float* const argY = pin_s(p.Y); float* const argYs = pin_s(p.Ys);      // epilogue arguments, read once (common.h)
    const int argM = pin_s(p.M), argN = pin_s(p.N), argLdy = pin_s(p.ldy), argAct = pin_s(p.act);

    // ---- residual tile first: these loads are OLDER than every LDS-DMA piece, so the counted waits below also
    // retire them and the epilogue never stalls on a dependent global load.  The epilogue works on 16-byte units
    // (row, 4 columns) spread over the 256 threads so that loads and stores are whole 256-byte row segments.
    constexpr int UPR = BN / 4;                    // 16-byte units per tile row
    constexpr int UNITS = BM * UPR / (NW * 64);    // units per thread
    static_assert((BM * UPR) % (NW * 64) == 0 && (NW * 64) % UPR == 0, "epilogue units must split evenly");
    constexpr int CLD = BN + 4;                    // LDS row stride of the staged C tile (floats)
    static_assert(BM * CLD <= 4 * SUB, "C tile must fit in the operand buffers");
    f32x4 rv[UNITS];
    const bool has_res = !partial && p.res != nullptr;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};               // all units of a thread share their 4 columns (256 % UPR == 0)
    if (!partial && p.bias != nullptr && col0 + 4 * (tid % UPR) < p.N) bv = ld4(p.bias + col0 + 4 * (tid % UPR));
#pragma unroll
    for (int u = 0; u < UNITS; ++u) {
        const int id = tid + u * NW * 64;
        const int gr = row0 + id / UPR, gc = col0 + 4 * (id % UPR);
        rv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (has_res && gr < p.M && gc < p.N) rv[u] = ld4(p.res + (size_t)gr * p.ldres + gc);
    }

    // ---- LDS-DMA pieces.  A piece = 4 rows x 64 floats of one sub-slice: rows 16 g + 4 q .. + 3 of 16-row group g, q < 4.
    // WHO requests WHAT (round 5): the waves form two pairs; pair 0 (waves 0, 1) brings sub-slices 0 and 2, pair 1 (waves 2, 3) brings
    // 1 and 3, wave `half` of a pair the row quads q = 2 half, 2 half + 1 - and a wave never has more than ONE sub-slice in flight, so
    // every wait is `vmcnt(0)`.  LDS-DMA requests of a wave do not complete in issue order when their latencies differ (gemm_big.hip,
    // header): the counted waits of rounds 1 - 4 (all four sub-slices requested at entry, `vmcnt(PW - PW_SUB)` ...) could be
    // satisfied by the pieces of a later sub-slice.
    constexpr int GA = BM / 16, GT = ROWS / 16;    // 16-row groups: A tile, total
    static_assert(BM % 16 == 0 && BN % 16 == 0, "tiles are multiples of 16 rows");
    const int pair = wave >> 1, half = wave & 1;
    int rlq[2], klq[2];
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
        rlq[qq] = 4 * (2 * half + qq) + (lane >> 4);                      // row within a 16-row group = LDS row & 15
        klq[qq] = (((lane & 15) ^ rlq[qq]) << 2) + (ks << 8);            // this lane's k: swizzled 16-byte slot + K slice
    }
    // Workgroups that share operand rows (same bm: A rows, same bn: W rows) start at the same time; rotating which
    // 64-wide k block lands in LDS sub-slice s keeps them from all missing on the same cache lines at once.
    const int rot = (bm + bn) & 3;
    const float* abase; int ald;
    if ((ks << 8) < p.K1) { abase = p.A; ald = p.lda; } else { abase = p.A2 - p.K1; ald = p.lda2; }
    float* const lbase = lds + 4 * (2 * half) * 64;
    auto issue = [&](int s, int g, int qq) __attribute__((always_inline)) {   // g, qq are compile-time at every call site
        const float* src;
        if (g < GA) {
            int gr = row0 + 16 * g + rlq[qq]; gr = gr < p.M ? gr : p.M - 1;
            src = abase + (size_t)gr * ald + klq[qq] + (((s + rot) & 3) << 6);
        } else {
            int gc = col0 + 16 * (g - GA) + rlq[qq]; gc = gc < p.N ? gc : p.N - 1;
            src = p.W + (size_t)gc * p.ldw + klq[qq] + (((s + rot) & 3) << 6);
        }
        glds16(src, lbase + s * SUB + (16 * g + 4 * qq) * 64);
    };
    auto issue_sub = [&](int s) __attribute__((always_inline)) {          // this wave's 2 GT pieces of sub-slice s
#pragma unroll
        for (int g = 0; g < GT; ++g)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) issue(s, g, qq);
    };

    Acc acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < MM::REGS; ++r) acc[i][j][r] = 0.f;

    const int frow = MM::frow(lane), fk = MM::fk(lane);
    auto compute_sub = [&](int s) __attribute__((always_inline)) {
        const float* sa = lds + s * SUB + (wm * TMW) * 64;
        const float* sb = lds + s * SUB + (BM + wn * TNW) * 64;
#pragma unroll
        for (int g = 0; g < 64 / MM::KG; ++g) {
            const int c = g * (MM::KG / 4) + fk;
            f32x4 fa[RM], fb[RN];
#pragma unroll
            for (int i = 0; i < RM; ++i) { const int r = i * MT + frow; fa[i] = ld4(sa + r * 64 + ((c ^ ((wm * TMW + r) & 15)) << 2)); }
#pragma unroll
            for (int j = 0; j < RN; ++j) { const int r = j * MT + frow; fb[j] = ld4(sb + r * 64 + ((c ^ ((BM + wn * TNW + r) & 15)) << 2)); }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < RM; ++i)
#pragma unroll
                    for (int j = 0; j < RN; ++j) acc[i][j] = MM::mma(fa[i][e], fb[j][e], acc[i][j]);
        }
    };

    {
        STAMP(0);
        issue_sub(pair);                            // pair 0: sub-slice 0, pair 1: sub-slice 1
        STAMP(1);
        if (pair == 0) wait_vmcnt<0>();             // (also retires this wave's residual / bias loads: they are older)
        __builtin_amdgcn_s_barrier();               // sub-slice 0 landed
        if (pair == 0) issue_sub(2);
        STAMP(2); compute_sub(0);
        STAMP(3);
        if (pair == 1) wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();               // 1
        if (pair == 1) issue_sub(3);
        compute_sub(1);
        if (pair == 0) wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();               // 2
        compute_sub(2);
        if (pair == 1) wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();               // 3
        reg_touch(bv);                              // everything has landed: keep compiler-made vmcnt(0) out of the store loop
#pragma unroll
        for (int u = 0; u < UNITS; ++u) reg_touch(rv[u]);
        STAMP(4); compute_sub(3);
        STAMP(5);
    }

    // ------------------------------------------------------------------ epilogue, staged through LDS
    __builtin_amdgcn_s_barrier();                  // every wave is done reading the operand tiles
    float* ct = lds;
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < MM::REGS; ++r)
                ct[(wm * TMW + i * MT + MM::arow(lane, r)) * CLD + wn * TNW + j * MT + MM::acol(lane)] = acc[i][j][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float* Y = argY == nullptr ? nullptr : argY + (partial ? (size_t)ks * argM * argLdy : 0);
    // all LDS reads first (one latency), then the arithmetic, then the stores back to back
    f32x4 cv[UNITS];
#pragma unroll
    for (int u = 0; u < UNITS; ++u) {
        const int id = tid + u * NW * 64;
        cv[u] = ld4(ct + (id / UPR) * CLD + 4 * (id % UPR));
    }
    if (!partial) {
        act_dispatch(argAct, [&](auto ACT) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < UNITS; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) cv[u][e] = act_c<decltype(ACT)::value>(cv[u][e] + bv[e]) + rv[u][e];
        });
    }
#pragma unroll
    for (int u = 0; u < UNITS; ++u) {
        const int id = tid + u * NW * 64;
        const int gr = row0 + id / UPR, gc = col0 + 4 * (id % UPR);
        if (gr < argM && gc < argN) {
            if (argY != nullptr) st4g(Y + (size_t)gr * argLdy + gc, cv[u]);
            if (!partial && argYs != nullptr) store_split4g(argYs + (size_t)gr * argLdy, gc, cv[u]);
        }
    }
    STAMP(6);
}

// ---- f16x3 kernel: producer / consumer, 8 waves.  Waves 4-7 only issue LDS-DMA: the texture path takes 64 B/clk per
// CU, so the 144 KiB of an 80x64 workgroup need ~2.3 k cycles of issue slots, and a wave that is issuing DMA cannot run
// MFMAs in the meantime.  Waves 0-3 (one per SIMD) only compute, arranged WN x WK: each owns BM x (BN / WN) outputs over
// 1 / WK of the K slice - with 16x16x32 MFMAs a wave tile of tm x tn reads (1/tm + 1/tn) * 4 bytes of LDS per MAC, and an
// 80 x 16 wave tile (four waves side by side) ran at LDS-read speed, not at matrix-pipe speed.  The K parts are summed in
// the staged epilogue.  The producers signal "phase landed" through the workgroup barrier (`vmcnt(0)` by the pair of producers that
// brought the sub-slice, the other pair's next one already in flight); the consumers take the four 64-wide sub-slices one by one (wave wk its 32-k
// half), so they trail the fill by one sub-slice.  Stamps: stores issued 6.5 k cycles after workgroup start (11.7 k for
// the previous all-waves-do-everything kernel, 6.9 k with two phases of two sub-slices); an empty kernel with this
// launch geometry costs 1.7 us per launch, the real one 5.4 us (scripts/ubench_empty.py).
template <int BM, int BN, int WN, int WK>
__global__ __launch_bounds__(512) void gemm_kp_kernel(const KrArgs p) {
    static_assert(WN == 2 && WK == 2, "four consumer waves: 2 (N halves) x 2 (K halves of every 64-wide sub-slice)");
    constexpr int MT = 16;
    constexpr int TNW = BN / WN;
    constexpr int RM = BM / MT, RN = TNW / MT;
    constexpr int ROWS = BM + BN;
    constexpr int SUB = ROWS * 64;
    constexpr int GA = BM / 16, GT = ROWS / 16;    // 16-row groups = DMA pieces per producer wave per sub-slice
    static_assert(2 * GT <= 63, "vmcnt is 6 bits (a producer has one sub-slice = 2 GT pieces in flight)");
    static_assert(BM % 16 == 0 && BN % (16 * WN) == 0, "tiles are multiples of 16 rows");

    __shared__ __attribute__((aligned(1024))) float lds[4 * SUB];
#ifdef LADIFF_STAMPS
    if (p.stamps == reinterpret_cast<unsigned long long*>(1)) { if (threadIdx.x == 9999) lds[0] = 0.f; return; }   // empty-launch probe
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbn = (p.N + BN - 1) / BN;
    const int bm = blockIdx.x / nbn, bn = blockIdx.x % nbn;
    const int ks = blockIdx.y;
    const int row0 = bm * BM, col0 = bn * BN;
    const bool partial = gridDim.y > 1;
    // epilogue: 16-byte units (row, 4 columns) spread over all 512 threads (the producers have nothing else left to do)
    constexpr int UPR = BN / 4;
    constexpr int TU = BM * UPR;
    constexpr int UNITS = (TU + 511) / 512;
    static_assert(512 % UPR == 0, "all units of a thread share their columns");
    constexpr int CLD = BN + 4;
    constexpr int CPL = BM * CLD;
    // the C planes go into sub-slices 0..WK-1, which nobody reads after phase 0: no barrier between the MFMAs and the staging
    static_assert(CPL <= SUB, "a C plane must fit in one phase-0 sub-slice");
    float* const argY = pin_s(p.Y); float* const argYs = pin_s(p.Ys);
    const int argM = pin_s(p.M), argN = pin_s(p.N), argLdy = pin_s(p.ldy), argAct = pin_s(p.act);
    f32x4 rv[UNITS];
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    auto prefetch_epilogue = [&]() __attribute__((always_inline)) {
        const bool has_res = !partial && p.res != nullptr;
        if (!partial && p.bias != nullptr && col0 + 4 * (tid % UPR) < p.N) bv = ld4(p.bias + col0 + 4 * (tid % UPR));
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            const int id = tid + u * 512;
            const int gr = row0 + id / UPR, gc = col0 + 4 * (id % UPR);
            rv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (has_res && id < TU && gr < p.M && gc < p.N) rv[u] = ld4(p.res + (size_t)gr * p.ldres + gc);
        }
    };
    auto store_tile = [&]() __attribute__((always_inline)) {          // after the barrier that follows the staging
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        reg_touch(bv);
#pragma unroll
        for (int u = 0; u < UNITS; ++u) reg_touch(rv[u]);
        float* Y = argY == nullptr ? nullptr : argY + (partial ? (size_t)ks * argM * argLdy : 0);
        f32x4 cv[UNITS];
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            int id = tid + u * 512; id = id < TU ? id : 0;
            cv[u] = ld4(lds + (id / UPR) * CLD + 4 * (id % UPR));
#pragma unroll
            for (int k = 1; k < WK; ++k) {
                const f32x4 t = ld4(lds + k * SUB + (id / UPR) * CLD + 4 * (id % UPR));
#pragma unroll
                for (int e = 0; e < 4; ++e) cv[u][e] += t[e];
            }
        }
        if (!partial) {
            act_dispatch(argAct, [&](auto ACT) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < UNITS; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) cv[u][e] = act_c<decltype(ACT)::value>(cv[u][e] + bv[e]) + rv[u][e];
            });
        }
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            const int id = tid + u * 512;
            const int gr = row0 + id / UPR, gc = col0 + 4 * (id % UPR);
            if (id < TU && gr < argM && gc < argN) {
                if (argY != nullptr) st4g(Y + (size_t)gr * argLdy + gc, cv[u]);
                if (!partial && argYs != nullptr) store_split4g(argYs + (size_t)gr * argLdy, gc, cv[u]);
            }
        }
    };

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        // Producer pair 0 (waves 4, 5) brings sub-slices 0 and 2, pair 1 (waves 6, 7) brings 1 and 3; wave `half` of a pair the row
        // quads q = 2 half, 2 half + 1 of every 16-row group.  A producer never has more than ONE sub-slice in flight and every wait
        // is `vmcnt(0)` (LDS-DMA requests of a wave do not complete in issue order: gemm_big.hip, header); the workgroup still has two
        // sub-slices in flight, requested at the points the counted form of rounds 1 - 4 requested them.
        const int pw = wave - 4, pair = pw >> 1, half = pw & 1;
        int rlq[2], klq[2];
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            rlq[qq] = 4 * (2 * half + qq) + (lane >> 4);
            klq[qq] = (((lane & 15) ^ rlq[qq]) << 2) + (ks << 8);
        }
        const float* abase; int ald;
        if ((ks << 8) < p.K1) { abase = p.A; ald = p.lda; } else { abase = p.A2 - p.K1; ald = p.lda2; }
        float* const lbase = lds + 4 * (2 * half) * 64;
        auto issue = [&](int s, int g, int qq) __attribute__((always_inline)) {
            const float* src;
            if (g < GA) {
                int gr = row0 + 16 * g + rlq[qq]; gr = gr < p.M ? gr : p.M - 1;
                src = abase + (size_t)gr * ald + klq[qq] + (s << 6);
            } else {
                int gc = col0 + 16 * (g - GA) + rlq[qq]; gc = gc < p.N ? gc : p.N - 1;
                src = p.W + (size_t)gc * p.ldw + klq[qq] + (s << 6);
            }
            glds16(src, lbase + s * SUB + (16 * g + 4 * qq) * 64);
        };
        auto issue_sub = [&](int s) __attribute__((always_inline)) {
#pragma unroll
            for (int g = 0; g < GT; ++g)
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) issue(s, g, qq);
        };
        // four phases = the four 64-wide sub-slices; a sub-slice is signalled when it has landed, with the next one already in flight
        // (the other pair's): the consumers trail the fill by one sub-slice instead of half the K slice
        issue_sub(pair);
        if (pair == 0) wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                                  // sub-slice 0 landed
        if (pair == 0) issue_sub(2);
        if (pair == 1) wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                                  // 1
        if (pair == 1) issue_sub(3);
        if (pair == 0) wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                                  // 2
        if (pair == 1) wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                                  // 3
        prefetch_epilogue();
        __builtin_amdgcn_s_barrier();                                  // C planes staged
        store_tile();
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int wn = wave % WN, wk = wave / WN;
    prefetch_epilogue();

    f32x4 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fk = lane >> 4;
    auto phase = [&](int ph) __attribute__((always_inline)) {
        // sub-slice ph, 32-k half wk: 16-byte slots 4 wk + fk (hi) and 8 + 4 wk + fk (lo)
        const float* sa = lds + ph * SUB;
        const float* sb = lds + ph * SUB + (BM + wn * TNW) * 64;
        const int ch = 4 * wk + fk, cl = 8 + 4 * wk + fk;
        s16x8 ah[RM], al[RM], bh[RN], bl[RN];
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            const int r = i * MT + frow, x = r & 15;
            ah[i] = __builtin_bit_cast(s16x8, ld4(sa + r * 64 + ((ch ^ x) << 2)));
            al[i] = __builtin_bit_cast(s16x8, ld4(sa + r * 64 + ((cl ^ x) << 2)));
        }
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const int r = j * MT + frow, x = (BM + wn * TNW + r) & 15;
            bh[j] = __builtin_bit_cast(s16x8, ld4(sb + r * 64 + ((ch ^ x) << 2)));
            bl[j] = __builtin_bit_cast(s16x8, ld4(sb + r * 64 + ((cl ^ x) << 2)));
        }
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j) acc[i][j] = MFMA16_S16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j) acc[i][j] = MFMA16_S16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j) acc[i][j] = MFMA16_S16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    };

    STAMP(0);
    __builtin_amdgcn_s_barrier();                  // sub-slice 0 landed
    STAMP(1);
    phase(0);
    STAMP(2);
    __builtin_amdgcn_s_barrier(); phase(1);
    STAMP(3);
    __builtin_amdgcn_s_barrier(); phase(2);
    __builtin_amdgcn_s_barrier(); phase(3);
    STAMP(4);
    float* ct = lds + wk * SUB;                    // phase-0 sub-slice of this K half: free since the phase-1 barrier
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                ct[(i * MT + 4 * (lane >> 4) + r) * CLD + wn * TNW + j * MT + (lane & 15)] = acc[i][j][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAMP(5);
    __builtin_amdgcn_s_barrier();
    STAMP(7);
    store_tile();
    STAMP(6);
}

template <int BM, int BN, int WN, int WK>
static int launch_kp_cfg(const KrArgs& a, int splits, hipStream_t s) {
    const int nbm = (a.M + BM - 1) / BM, nbn = (a.N + BN - 1) / BN;
    hipLaunchKernelGGL((gemm_kp_kernel<BM, BN, WN, WK>), dim3(nbm * nbn, splits), dim3(512), 0, s, a);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

template <int BM, int BN, int WM, int WN, int MT>
static int launch_kr_cfg(const KrArgs& a, int splits, hipStream_t s) {
    const int nbm = (a.M + BM - 1) / BM, nbn = (a.N + BN - 1) / BN;
    hipLaunchKernelGGL((gemm_kr_kernel<BM, BN, WM, WN, MT>), dim3(nbm * nbn, splits), dim3(WM * WN * 64), 0, s, a);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

// K == 256: one launch with the fused epilogue.  K == 512 / 1024: split-K, Y must hold K/256 partial planes [K/256][M][ldy].
int launch_gemm_kr(const KrArgs& a0, hipStream_t s) {
    KrArgs a = a0;
    if (a.A2 == nullptr) a.K1 = a.K;
    LADIFF_CHECK_ARG(a.A && a.W && (a.Y || a.Ys) && a.M > 0 && a.N > 0 && a.K > 0);
    if (a.K % 256 != 0 || a.K1 % 256 != 0 || (a.lda % 4) || (a.ldw % 4) || (a.A2 && (a.lda2 % 4))) return LADIFF_ERR_SHAPE;
    if ((a.N % 4) || (a.ldy % 4) || (a.res && (a.ldres % 4))) return LADIFF_ERR_SHAPE;   // 16-byte epilogue units
    const int splits = a.K / 256;
    if (splits > 1 && (a.Y == nullptr || a.Ys != nullptr)) return LADIFF_ERR_ARG;
    if (a.Ys != nullptr && (a.ldy % 64)) return LADIFF_ERR_SHAPE;                         // S-format rows are 64-column blocks
    if (a.split) {
        if (splits == 4 || (splits == 1 && a.N >= 1024)) return launch_kp_cfg<80, 64, 2, 2>(a, splits, s);
        if (splits > 1 || a.N >= 512) return launch_kp_cfg<64, 64, 2, 2>(a, splits, s);
        return launch_kp_cfg<32, 32, 2, 2>(a, 1, s);
    }
    if (splits > 1) {
        if (splits == 4) return launch_kr_cfg<80, 64, 1, 4, 16>(a, splits, s);   // 16 x 4 x 4 = 256 workgroups at M=1280, N=256
        return launch_kr_cfg<64, 64, 2, 2, 32>(a, splits, s);
    }
    if (a.N >= 1024) return launch_kr_cfg<80, 64, 1, 4, 16>(a, 1, s);
    if (a.N >= 512) return launch_kr_cfg<64, 64, 2, 2, 32>(a, 1, s);
    return launch_kr_cfg<32, 32, 2, 2, 16>(a, 1, s);
}

}  // namespace ladiff
