// Counter-based per-step noise of the stochastic schedulers (DDPM, DDIM with eta > 0): the reference draws it inside
// `scheduler.step` (diffusers' randn_tensor, ladiff.py:492 -> scheduler.step; configs/modules_novae/scheduler.yaml:16-29), one
// [B, T, 256] tensor per step.  Here a value is a pure function of (seed, schedule position, GLOBAL prompt index, latent, column):
// nothing is stored (a 1000-step schedule at 128 prompts would be 655 MB), a batch sharded over ranks or cut into chunks draws what
// the whole batch would have drawn, and the consumer (the TAIL stage of the pipeline kernel, the tail launch of the launch-per-stage
// loop, ladiff_noise_fill) computes it where it needs it.
//   bits:    Philox4x32-10 (Salmon et al., SC'11; the Random123 known-answer vectors are pinned in tests/test_noise.py)
//            counter = (latent * 64 + column / 4, global prompt, schedule position, 0), key = (seed low, seed high)
//   normals: Box-Muller on (x0, x1) and (x2, x3): u = ((x >> 9) + 0.5) 2^-23 in (0, 1) - exact in fp32 -,
//            r = sqrt(-2 ln u_a), (r cos 2 pi u_b, r sin 2 pi u_b) -> columns c, c + 1 (and c + 2, c + 3)
// oracle/ladiff_oracle.py:device_noise is the numpy restatement (same integers; ln / cos / sin differ by the libraries' last bits).
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace ladiff {

struct NoiseGen {              // by value in kernel arguments; on == 0: the caller's tensor (or none) is used
    unsigned seed_lo, seed_hi;
    unsigned prompt0;          // global index of this launch's prompt 0
    int on;
};

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float noise_unit(unsigned x) { return ((float)(x >> 9) + 0.5f) * 1.1920928955078125e-07f; }   // 2^-23

// the four normals of columns c .. c + 3 (c = 4 * chunk) of latent `t` of global prompt `prompt` at schedule position `step`
__device__ __forceinline__ void noise_normal4(const NoiseGen& g, int step, unsigned prompt, int t, int chunk, float (&z)[4]) {
    unsigned x[4];
    philox4x32_10((unsigned)(t * 64 + chunk), prompt, (unsigned)step, 0u, g.seed_lo, g.seed_hi, x);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float r = sqrtf(-2.0f * logf(noise_unit(x[2 * h])));
        const float th = 6.2831854820251465f * noise_unit(x[2 * h + 1]);
        z[2 * h] = r * cosf(th);
        z[2 * h + 1] = r * sinf(th);
    }
}

}  // namespace ladiff
