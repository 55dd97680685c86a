// What does one producer -> consumer hand-off (1 KiB of rows + a flag) cost between two workgroups of the SAME XCD through that
// XCD's L2 (sc0 loads, plain / sc0 stores) compared with the agent-scope path the pipeline loop uses today (sc1 loads, sc1
// write-through stores), and compared with two workgroups on DIFFERENT XCDs?
// Two single-wave workgroups play ping-pong: A stores a row, drains (vmcnt(0)), raises its flag; B polls the flag, loads the row,
// stores its own row, drains, raises its flag; ...  Time per round trip / 2 = one hop (store -> drain -> flag -> poll -> load).
// Every spin loop is bounded: a mode that never sees the other side's flag reports FAILED instead of hanging the box.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/ubench_xcd_handoff.bin scripts/ubench_xcd_handoff.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int LD, int ST>
__global__ __launch_bounds__(64, 1) void pingpong(float* data, unsigned* flags, unsigned* xcc, unsigned long long* ticks, int wa, int wb, int rounds,
                                                   unsigned* fail) {
    extern __shared__ char lds[];                                 // 100 KiB: one workgroup per CU
    const int lane = threadIdx.x;
    if (lane == 0) xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;   // XCC_ID
    if ((int)blockIdx.x != wa && (int)blockIdx.x != wb) return;
    const int me = (int)blockIdx.x == wa ? 0 : 1;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(data, 0, 0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(flags, 0, 0xffffffffu, 0x00020000);
    // row of side s: data + s * 4096 floats; flag of side s: flags + s * 64
    f32x4 v = {(float)lane, 1.f, 2.f, 3.f};
    float acc = 0.f;
    unsigned long long t0 = 0;
    for (int r = 1; r <= rounds; ++r) {
        {
            if (me == 1 || r > 1) {                               // wait for the other side's flag of this round (A of round 1 starts)
                const unsigned want = me == 1 ? (unsigned)r : (unsigned)(r - 1);
                unsigned spins = 0, f = 0;
                do {
                    asm volatile("" ::: "memory");                // a fresh load every turn
                    f = __builtin_amdgcn_raw_buffer_load_b32(rf, (unsigned)((1 - me) * 256), 0, LD);
                    if (++spins > 2000000u) { if (lane == 0) *fail = 1u; return; }
                } while (f < want);
                const f32x4 x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, (unsigned)((1 - me) * 16384 + lane * 16), 0, LD));
                acc += x[0] + x[1];
                v[1] = x[1] + 1.f;
            }
        }
        if (r == 2 && me == 0) t0 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rd, (unsigned)(me * 16384 + lane * 16), 0, ST);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __builtin_amdgcn_raw_buffer_store_b32((unsigned)r, rf, (unsigned)(me * 256), 0, ST);
    }
    if (me == 0 && lane == 0) ticks[0] = __builtin_amdgcn_s_memrealtime() - t0;
    if (acc == 12345.678f) data[8192 + lane] = acc;
}

template <int LD, int ST>
static void run(const char* name, float* data, unsigned* flags, unsigned* xcc, unsigned long long* ticks, unsigned* fail, int wa, int wb) {
    const int rounds = 2001;
    hipFuncSetAttribute(reinterpret_cast<const void*>(pingpong<LD, ST>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipMemset(flags, 0, 4096); hipMemset(fail, 0, 4); hipMemset(ticks, 0, 8);
    hipLaunchKernelGGL((pingpong<LD, ST>), dim3(64), dim3(64), 100 * 1024, 0, data, flags, xcc, ticks, wa, wb, rounds, fail);
    hipDeviceSynchronize();
    unsigned hx[64], hf = 0; unsigned long long ht = 0;
    hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost); hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost); hipMemcpy(&ht, ticks, 8, hipMemcpyDeviceToHost);
    if (hf) printf("%-44s workgroups %2d (XCC %u) <-> %2d (XCC %u): FAILED (flag never seen)\n", name, wa, hx[wa], wb, hx[wb]);
    else printf("%-44s workgroups %2d (XCC %u) <-> %2d (XCC %u): %.3f us per hop\n", name, wa, hx[wa], wb, hx[wb], ht * 0.01 / (2.0 * (rounds - 2)));
}

int main() {
    float* data; unsigned *flags, *xcc, *fail; unsigned long long* ticks;
    hipMalloc(&data, 1 << 20); hipMalloc(&flags, 4096); hipMalloc(&xcc, 4096); hipMalloc(&fail, 4); hipMalloc(&ticks, 8);
    hipMemset(data, 0, 1 << 20);
    for (int rep = 0; rep < 2; ++rep) {
        const int pairs[3][2] = {{0, 8}, {0, 1}, {3, 12}};
        for (auto& pr : pairs) {
            run<16, 16>("sc1 loads, sc1 write-through stores", data, flags, xcc, ticks, fail, pr[0], pr[1]);
            run<1, 0>("sc0 loads, plain stores", data, flags, xcc, ticks, fail, pr[0], pr[1]);
            run<1, 1>("sc0 loads, sc0 stores", data, flags, xcc, ticks, fail, pr[0], pr[1]);
            run<17, 16>("sc0 sc1 loads, sc1 stores", data, flags, xcc, ticks, fail, pr[0], pr[1]);
            run<16, 0>("sc1 loads, plain stores", data, flags, xcc, ticks, fail, pr[0], pr[1]);
        }
    }
    unsigned hx[64];
    hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost);
    printf("XCC_ID of workgroups 0..63:");
    for (int i = 0; i < 64; ++i) printf(" %u", hx[i]);
    printf("\n");
    return 0;
}
