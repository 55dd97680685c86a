// Does `s_waitcnt vmcnt(N)` with N > 0 prove that the OLDEST load has landed when the younger ones take a different path through the
// cache hierarchy?  (The ISA manuals say vector memory loads return in order; the loop kernel's hand-off relied on that wherever the compiler
// counted a wait - e.g. QKV's loader waves in round 6: the rows of a block requested early with `sc1` (past this CU's L1), then six plain
// loads of geometry / text K|V words (L1 hits), then `s_waitcnt vmcnt(1)` in front of the first look at the rows.)
//
// One wave per probe: x = sentinel; ONE cold load into x (a line nobody touched: HBM or at least L2 miss); six hot loads (a 4 KB table this
// wave has just read: L1 hits); `s_waitcnt vmcnt(6)`; copy x; `s_waitcnt vmcnt(0)`; the copy must equal x.  A copy that still holds the
// sentinel = the counter reached 6 while the OLDEST load was still in flight.
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_vmcnt_order.bin scripts/ubench_vmcnt_order.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define PROBE(COLD_FLAGS, HOT_FLAGS)                                                                                               \
    asm volatile("v_mov_b32 %0, 0x7fc0dead\n\t"                                                                                    \
                 "s_waitcnt vmcnt(0)\n\t"                                                                                          \
                 "global_load_dword %0, %8, off " COLD_FLAGS "\n\t"                                                                \
                 "global_load_dword %2, %9, off " HOT_FLAGS "\n\t"                                                                 \
                 "global_load_dword %3, %9, off offset:256 " HOT_FLAGS "\n\t"                                                      \
                 "global_load_dword %4, %9, off offset:512 " HOT_FLAGS "\n\t"                                                      \
                 "global_load_dword %5, %9, off offset:768 " HOT_FLAGS "\n\t"                                                      \
                 "global_load_dword %6, %9, off offset:1024 " HOT_FLAGS "\n\t"                                                     \
                 "global_load_dword %7, %9, off offset:1280 " HOT_FLAGS "\n\t"                                                     \
                 "s_waitcnt vmcnt(6)\n\t"                                                                                          \
                 "v_mov_b32 %1, %0\n\t"                                                                                            \
                 "s_waitcnt vmcnt(0)"                                                                                              \
                 : "=&v"(x), "=&v"(e), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5)                            \
                 : "v"(ca), "v"(ha)                                                                                                \
                 : "memory")

template <int KIND>
__global__ void probe(const unsigned* cold, unsigned cold_lines, const unsigned* hot, unsigned long long* stats, int iters, unsigned seed) {
    const unsigned lane = threadIdx.x & 63, gwave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned state = seed ^ (gwave * 2654435761u);
    unsigned early = 0, wrong = 0;
    for (int it = 0; it < iters; ++it) {
        state = state * 1664525u + 1013904223u;
        const unsigned line = (state >> 3) % cold_lines;                 // 256 bytes per wave: one request, nobody else's line
        const unsigned* ca = cold + (size_t)line * 64 + lane;
        const unsigned* ha = hot + lane;
        // the hot table into this CU's L1 (and the L2)
        unsigned warm = 0;
        warm += hot[lane] + hot[lane + 64] + hot[lane + 128] + hot[lane + 192] + hot[lane + 256] + hot[lane + 320];
        asm volatile("" ::"v"(warm));
        unsigned x, e, t0, t1, t2, t3, t4, t5;
        if constexpr (KIND == 0) PROBE("", "");
        else if constexpr (KIND == 1) PROBE("sc1", "");
        else if constexpr (KIND == 2) PROBE("sc0 sc1", "");
        else if constexpr (KIND == 3) PROBE("nt", "");
        else if constexpr (KIND == 4) PROBE("sc1", "sc1");
        else PROBE("", "sc1");
        early += e != x;
        wrong += x != (line * 64 + lane) * 2u + 1u;
        wrong += (t0 != lane * 3u) + (t5 != (lane + 320) * 3u);
    }
    if (early) atomicAdd(stats, (unsigned long long)early);
    if (wrong) atomicAdd(stats + 1, (unsigned long long)wrong);
}

int main(int argc, char** argv) {
    const size_t cold_bytes = (size_t)4 << 30;
    const unsigned cold_lines = (unsigned)(cold_bytes / 256);
    unsigned *cold, *hot;
    unsigned long long* stats;
    hipMalloc(&cold, cold_bytes); hipMalloc(&hot, 4096); hipMalloc(&stats, 16);
    {
        std::vector<unsigned> h(1024);
        for (unsigned i = 0; i < 1024; ++i) h[i] = i * 3u;
        hipMemcpy(hot, h.data(), 4096, hipMemcpyHostToDevice);
        std::vector<unsigned> c(1 << 24);
        for (size_t base = 0; base < cold_bytes / 4; base += c.size()) {
            for (size_t i = 0; i < c.size(); ++i) c[i] = (unsigned)(base + i) * 2u + 1u;
            hipMemcpy(cold + base, c.data(), c.size() * 4, hipMemcpyHostToDevice);
        }
    }
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const char* names[6] = {"cold plain,   hot plain", "cold sc1,     hot plain", "cold sc0 sc1, hot plain", "cold nt,      hot plain",
                            "cold sc1,     hot sc1  ", "cold plain,   hot sc1  "};
    for (int waves = 1; waves <= 8; waves *= 2)
        for (int kind = 0; kind < 6; ++kind) {
            hipMemset(stats, 0, 16);
            const dim3 grid(256), block(64 * waves);
            const unsigned seed = 12345u + 77u * kind + waves;
            switch (kind) {
                case 0: probe<0><<<grid, block>>>(cold, cold_lines, hot, stats, iters, seed); break;
                case 1: probe<1><<<grid, block>>>(cold, cold_lines, hot, stats, iters, seed); break;
                case 2: probe<2><<<grid, block>>>(cold, cold_lines, hot, stats, iters, seed); break;
                case 3: probe<3><<<grid, block>>>(cold, cold_lines, hot, stats, iters, seed); break;
                case 4: probe<4><<<grid, block>>>(cold, cold_lines, hot, stats, iters, seed); break;
                default: probe<5><<<grid, block>>>(cold, cold_lines, hot, stats, iters, seed); break;
            }
            unsigned long long h[2];
            if (hipMemcpy(h, stats, 16, hipMemcpyDeviceToHost) != hipSuccess) { printf("launch failed\n"); return 1; }
            printf("%d waves/CU  %s : %llu lane-probes of %llu saw the oldest load still in flight behind vmcnt(6); wrong values %llu\n", waves, names[kind], h[0],
                   (unsigned long long)iters * 256ull * waves * 64ull, h[1]);
        }
    return 0;
}
