// camera_math.hpp — SO3xR3 exponential map and pose composition shared by camera_opt.hip (k_camera_adjust,
// k_camera_pose_grad) and pixel_sampler.hip (fnr_train_prologue computes a ray's corrected camera in place).
// nerfstudio 0.3.2 CameraOptimizer(mode="SO3xR3") semantics, see camera_opt.hip.
#pragma once
#include "common.hpp"

namespace fnr {

struct SO3 {
  float R[9];
  float theta2_raw, theta, f1, f2;
};

__device__ __forceinline__ SO3 so3_exp(const float* w) {
  SO3 s;
  s.theta2_raw = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const float t2 = fmaxf(s.theta2_raw, 1e-4f);
  s.theta = sqrtf(t2);
  const float inv = 1.0f / s.theta;
  s.f1 = inv * sinf(s.theta);
  s.f2 = inv * inv * (1.0f - cosf(s.theta));
  // K = [[0,-wz,wy],[wz,0,-wx],[-wy,wx,0]];  K^2 = w w^T - |w|^2 I
  const float x = w[0], y = w[1], z = w[2];
  const float K[9] = {0.f, -z, y, z, 0.f, -x, -y, x, 0.f};
  const float K2[9] = {-(y * y + z * z), x * y, x * z, x * y, -(x * x + z * z), y * z, x * z, y * z, -(x * x + y * y)};
#pragma unroll
  for (int i = 0; i < 9; ++i) s.R[i] = s.f1 * K[i] + s.f2 * K2[i] + ((i % 4 == 0) ? 1.0f : 0.0f);
  return s;
}

// out [3,4] = multiply(M [3,4], exp_map_SO3xR3(tv [6])):  R' = R1 R,  t' = t1 + R1 t
__device__ __forceinline__ void adjusted_camera(const float* __restrict__ M, const float* __restrict__ tv, float (&out)[12]) {
  const SO3 s = so3_exp(tv + 3);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int b = 0; b < 3; ++b)
      out[4 * a + b] = M[4 * a] * s.R[b] + M[4 * a + 1] * s.R[3 + b] + M[4 * a + 2] * s.R[6 + b];
    out[4 * a + 3] = M[4 * a + 3] + (M[4 * a] * tv[0] + M[4 * a + 1] * tv[1] + M[4 * a + 2] * tv[2]);
  }
}

}  // namespace fnr
