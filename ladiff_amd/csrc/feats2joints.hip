// feats2joints on the device: de-normalise the decoded features and recover joint positions from the
// rotation-invariant coordinates (HumanML3D 263-dim / KIT 251-dim layout).
//   reference: HumanML3DDataModule.feats2joints  (ladiff/data/HumanML3D.py:44-48, Kit.py:48-53)
//              recover_root_rot_pos / recover_from_ric (ladiff/data/humanml/scripts/motion_process.py:362-381, :415-430)
//              qinv / qrot (ladiff/data/humanml/common/quaternion.py:16-20, :54-73)
// It is the step right after the hot path: the reference copies [B,F,263] to the host and does this
// in PyTorch on the CPU (ladiff.py:307); here the frames never leave HBM and the result is 4x smaller ([B,F,J,3]).
//
// One workgroup per motion, one thread per frame.  The two prefix sums over frames (yaw angle, root XZ) are evaluated in
// frame order by single lanes, exactly as torch.cumsum does on the CPU, so results match the reference to rounding of
// sin/cos only; everything else is frame-parallel.  HBM-bound: 1052 B read + 264 B written per frame.
#include "kernels.h"

namespace ladiff {

constexpr int F2J_MAXF = 256;

__global__ __launch_bounds__(F2J_MAXF) void feats2joints_kernel(const float* __restrict__ feats, const float* __restrict__ mean,
                                                                const float* __restrict__ stdv, int F, int C, int J,
                                                                float* __restrict__ joints) {
    __shared__ float s_a[F2J_MAXF], s_x[F2J_MAXF], s_z[F2J_MAXF];
    const int b = blockIdx.x, t = threadIdx.x;
    const float* row = feats + ((size_t)b * F + t) * C;
    auto den = [&](int c) { return row[c] * stdv[c] + mean[c]; };   // features * std + mean

    // r_rot_ang[t] = sum_{s < t} rot_vel[s]                                    motion_process.py:363-367
    s_a[t] = (t < F) ? den(0) : 0.f;
    __syncthreads();
    if (t == 0) {
        float acc = 0.f;                       // torch.cumsum of [0, v0, v1, ...] in frame order
        for (int s = 0; s < F; ++s) { const float v = s_a[s]; s_a[s] = acc; acc += v; }
    }
    __syncthreads();
    const float ang = (t < F) ? s_a[t] : 0.f;
    const float qw = cosf(ang), qy = -sinf(ang);          // qinv(r_rot_quat) = (cos, 0, -sin, 0)   :369-371, quaternion.py:16-20
    // v + 2 (w (q x v) + q x (q x v)) with q_vec = (0, qy, 0)                  quaternion.py:54-73
    auto rot = [&](float vx, float vy, float vz, float& ox, float& oy, float& oz) {
        const float ux = qy * vz, uy = 0.f * vx - 0.f * vz, uz = -qy * vx;     // cross(qvec, v), qvec = (0, qy, 0)
        const float wx = qy * uz, wy = 0.f * ux - 0.f * uz, wz = -qy * ux;     // cross(qvec, uv)
        ox = vx + 2.f * (qw * ux + wx);
        oy = vy + 2.f * (qw * uy + wy);
        oz = vz + 2.f * (qw * uz + wz);
    };
    // r_pos[t] = cumsum_t qrot(qinv(q_t), [vel_x[t-1], 0, vel_z[t-1]]),  r_pos[0] = 0          :373-378
    float px = 0.f, py = 0.f, pz = 0.f;
    if (t >= 1 && t < F) {
        const float* prev = row - C;
        rot(prev[1] * stdv[1] + mean[1], 0.f, prev[2] * stdv[2] + mean[2], px, py, pz);
    }
    s_x[t] = px; s_z[t] = pz;
    __syncthreads();
    if (t < 2) {
        float* arr = t == 0 ? s_x : s_z;
        float acc = 0.f;
        for (int s = 0; s < F; ++s) { acc += arr[s]; arr[s] = acc; }
    }
    __syncthreads();
    if (t >= F) return;
    const float rx = s_x[t], rz = s_z[t], ry = den(3);                           // r_pos[..., 1] = data[..., 3]   :380
    float* out = joints + ((size_t)b * F + t) * J * 3;
    out[0] = rx; out[1] = ry; out[2] = rz;                                       // root                :427
    for (int j = 1; j < J; ++j) {                                                // local joints        :417-425
        float ox, oy, oz;
        rot(den(4 + 3 * (j - 1)), den(5 + 3 * (j - 1)), den(6 + 3 * (j - 1)), ox, oy, oz);
        out[3 * j] = ox + rx; out[3 * j + 1] = oy; out[3 * j + 2] = oz + rz;
    }
}

int launch_feats2joints(const float* feats, const float* mean, const float* stdv, int B, int F, int C, int J, float* joints,
                        hipStream_t s) {
    if (F < 1 || F > F2J_MAXF || J < 2 || C < 4 + 3 * (J - 1)) return LADIFF_ERR_SHAPE;
    if (B == 0) return 0;
    hipLaunchKernelGGL(feats2joints_kernel, dim3(B), dim3(F2J_MAXF), 0, s, feats, mean, stdv, F, C, J, joints);
    LADIFF_LAUNCH_CHECK();
    return 0;
}

}  // namespace ladiff
