// Would a hand-off whose DATA carries the epoch (every 16-byte unit = 3 payload floats + a tag word, the fourth floats of a lane's
// units packed in a side unit; the consumer polls the units themselves) be faster than the pipeline loop's store -> drain -> count-in
// -> flag -> poll -> load hand-off?  Two 512-thread workgroups (one per CU, as the stages of csrc/systolic.hip) play ping-pong with
// a 16 x 256 fp32 tile (16 KiB): time per round trip / 2 = one hop.  Modes: FLAG (what the loop does today) and TAGGED; stores
// plain (both workgroups on one XCD) or write-through (sc1), loads sc1.  Every spin is bounded.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/ubench_ll_handoff.bin scripts/ubench_ll_handoff.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int SPIN_MAX = 400000;

template <int TAGGED, int ST>
__global__ __launch_bounds__(512, 1) void pingpong(float* data, unsigned* flags, unsigned* xcc, unsigned long long* ticks, int wa, int wb,
                                                    int rounds, unsigned* fail, unsigned* retries) {
    extern __shared__ char lds[];                                 // 100 KiB: one workgroup per CU
    __shared__ unsigned count;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; count = 0; }
    if ((int)blockIdx.x != wa && (int)blockIdx.x != wb) return;
    __syncthreads();
    const int me = (int)blockIdx.x == wa ? 0 : 1;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(data, 0, 0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(flags, 0, 0xffffffffu, 0x00020000);
    // side s writes its tile at byte s * 65536: main units at +tid*16 and +8192+tid*16, side units at +16384 + tid*16
    f32x4 a = {(float)tid, 1.f, 2.f, 3.f}, b = {(float)tid, 5.f, 6.f, 7.f};
    float acc = 0.f;
    unsigned long long t0 = 0;
    unsigned nretry = 0;
    for (int r = 1; r <= rounds; ++r) {
        if (me == 1 || r > 1) {
            const unsigned want = me == 1 ? (unsigned)r : (unsigned)(r - 1);
            const unsigned src = (unsigned)((1 - me) * 65536);
            f32x4 x, y;
            if (TAGGED) {
                int spins = 0;
                f32x4 u0, u1, u2;                                  // whole-vector casts only (element-wise __builtin_bit_cast miscompiled here)
                for (;;) {
                    asm volatile("" ::: "memory");
                    u0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, src + tid * 16, 0, 16));
                    u1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, src + 8192 + tid * 16, 0, 16));
                    u2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, src + 16384 + tid * 16, 0, 16));
                    const bool ok = __float_as_uint(u0[3]) == want && __float_as_uint(u1[3]) == want && __float_as_uint(u2[2]) == want;
                    if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
                    ++nretry;
                    if (++spins > SPIN_MAX) { if (lane == 0) *fail = 1u; return; }
                }
                x = f32x4{u0[0], u0[1], u0[2], u2[0]};
                y = f32x4{u1[0], u1[1], u1[2], u2[1]};
            } else {
                int spins = 0;
                unsigned f;
                do {
                    asm volatile("" ::: "memory");
                    f = __builtin_amdgcn_raw_buffer_load_b32(rf, (unsigned)((1 - me) * 256), 0, 16);
                    if (++spins > SPIN_MAX) { if (lane == 0) *fail = 1u; return; }
                } while (f < want);
                x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, src + tid * 16, 0, 16));
                y = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, src + 8192 + tid * 16, 0, 16));
            }
            acc += x[0] + y[3];
            // (the chain counts modulo 2^20 so that a soak of many rounds stays exact in fp32)
            { const float nx = (float)(((int)x[1] + 1) & 0xfffff); a[1] = nx; a[3] = nx + 2.f; b[2] = nx + 5.f; b[3] = nx + 6.f; }
            // the chain: every hop adds one to these four words - a torn or stale unit shows here
            const float expect = me == 1 ? (float)((2 * r - 1) & 0xfffff) : (float)((2 * (r - 1)) & 0xfffff);
            if (x[1] != expect || x[3] != expect + 2.f || y[2] != expect + 5.f || y[3] != expect + 6.f) {
                if (atomicCAS(fail, 0u, 2u) == 0u) {              // first failure: what was read
                    float* dbg = data + 200000;
                    dbg[0] = (float)r; dbg[1] = (float)tid; dbg[2] = (float)me; dbg[3] = expect;
                    dbg[4] = x[0]; dbg[5] = x[1]; dbg[6] = x[2]; dbg[7] = x[3]; dbg[8] = y[0]; dbg[9] = y[1]; dbg[10] = y[2]; dbg[11] = y[3];
                }
            }
        }
        if (r == 2 && me == 0 && tid == 0) t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned dst = (unsigned)(me * 65536);
        if (TAGGED) {
            const float tag = __uint_as_float((unsigned)r);
            const f32x4 u0 = {a[0], a[1], a[2], tag}, u1 = {b[0], b[1], b[2], tag}, u2 = {a[3], b[3], tag, 0.f};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, u0), rd, dst + tid * 16, 0, ST);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, u1), rd, dst + 8192 + tid * 16, 0, ST);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, u2), rd, dst + 16384 + tid * 16, 0, ST);
        } else {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a), rd, dst + tid * 16, 0, ST);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, b), rd, dst + 8192 + tid * 16, 0, ST);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned c = 0;
            if (lane == 0) c = atomicAdd(&count, 1u) + 1u;
            c = __builtin_amdgcn_readfirstlane(c);
            if (c == 8u * (unsigned)r && lane == 0) __builtin_amdgcn_raw_buffer_store_b32((unsigned)r, rf, (unsigned)(me * 256), 0, ST);
        }
    }
    if (me == 0 && tid == 0) ticks[0] = __builtin_amdgcn_s_memrealtime() - t0;
    if (lane == 0) atomicAdd(retries, nretry);
    if (acc == 12345.678f) data[100000 + tid] = acc;
}

static int g_rounds = 2001;
template <int TAGGED, int ST>
static void run(const char* name, float* data, unsigned* flags, unsigned* xcc, unsigned long long* ticks, unsigned* fail, unsigned* retries, int wa, int wb) {
    const int rounds = g_rounds;
    hipFuncSetAttribute(reinterpret_cast<const void*>(pingpong<TAGGED, ST>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipMemset(flags, 0, 4096); hipMemset(fail, 0, 4); hipMemset(ticks, 0, 8); hipMemset(retries, 0, 4); hipMemset(data, 0, 1 << 20);
    hipLaunchKernelGGL((pingpong<TAGGED, ST>), dim3(64), dim3(512), 100 * 1024, 0, data, flags, xcc, ticks, wa, wb, rounds, fail, retries);
    hipDeviceSynchronize();
    unsigned hx[64], hf = 0, hr = 0; unsigned long long ht = 0;
    hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost); hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost); hipMemcpy(&ht, ticks, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&hr, retries, 4, hipMemcpyDeviceToHost);
    if (hf == 2) {
        float d[12]; hipMemcpy(d, data + 200000, sizeof(d), hipMemcpyDeviceToHost);
        printf("    first bad read: round %.0f thread %.0f side %.0f expected x1 = %.0f (x3 = +2, y2 = +5, y3 = +6); got x = %.0f %.0f %.0f %.0f  y = %.0f %.0f %.0f %.0f\n",
               d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9], d[10], d[11]);
    }
    if (hf) printf("%-34s workgroups %2d (XCC %u) <-> %2d (XCC %u): FAILED (%u: 1 = never seen, 2 = wrong data)\n", name, wa, hx[wa], wb, hx[wb], hf);
    else printf("%-34s workgroups %2d (XCC %u) <-> %2d (XCC %u): %.3f us per hop   (%.1f polls per wave-hop, every received word checked)\n", name, wa, hx[wa], wb, hx[wb],
                ht * 0.01 / (2.0 * (rounds - 2)), hr / (16.0 * rounds) + 1.0);
    fflush(stdout);
}

int main(int argc, char** argv) {
    if (argc > 1) g_rounds = atoi(argv[1]);     // rounds per configuration: a long run is the soak for torn units (every received word is checked)
    float* data; unsigned *flags, *xcc, *fail, *retries; unsigned long long* ticks;
    hipMalloc(&data, 1 << 20); hipMalloc(&flags, 4096); hipMalloc(&xcc, 4096); hipMalloc(&fail, 4); hipMalloc(&ticks, 8); hipMalloc(&retries, 4);
    for (int rep = 0; rep < 2; ++rep) {
        const int pairs[3][2] = {{0, 8}, {0, 1}, {3, 12}};
        for (auto& pr : pairs) {
            run<0, 16>("FLAG   write-through stores", data, flags, xcc, ticks, fail, retries, pr[0], pr[1]);
            run<1, 16>("TAGGED write-through stores", data, flags, xcc, ticks, fail, retries, pr[0], pr[1]);
            run<0, 0>("FLAG   plain stores", data, flags, xcc, ticks, fail, retries, pr[0], pr[1]);
            run<1, 0>("TAGGED plain stores", data, flags, xcc, ticks, fail, retries, pr[0], pr[1]);
        }
    }
    return 0;
}
