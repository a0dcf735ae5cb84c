// sampler_math.hpp — the spaced sampler's bin arithmetic, shared by sampler.hip (k_sample_spaced) and
// pixel_sampler.hip (fnr_train_prologue samples level 0 in the launch that draws the rays).
#pragma once
#include "common.hpp"

namespace fnr {

// spacing <-> euclidean (SpacedSampler.generate_ray_samples; UniformLinDispPiecewiseSampler)
__device__ __forceinline__ float spacing_fn(int kind, float x) {
  if (kind == 0) return x;
  return (x < 1.0f) ? fdiv(x, 2.0f) : fsub(1.0f, fdiv(1.0f, fmul(2.0f, x)));
}
__device__ __forceinline__ float spacing_fn_inv(int kind, float x) {
  if (kind == 0) return x;
  return (x < 0.5f) ? fmul(2.0f, x) : fdiv(1.0f, fsub(2.0f, fmul(2.0f, x)));
}
__device__ __forceinline__ float spacing_to_euclid(int kind, float x, float s_near, float s_far) {
  return spacing_fn_inv(kind, fadd(fmul(x, s_far), fmul(fsub(1.0f, x), s_near)));
}

// bin edge j of a ray: base_bins[j] (eval) or jittered inside (lower, upper) — components/ray_samplers.py:79-87
__device__ __forceinline__ float spaced_bin_edge(const float* __restrict__ base_bins, int S, int j, bool jittered, float t) {
  float b = base_bins[j];
  if (jittered) {
    const float upper = (j < S) ? fdiv(fadd(base_bins[j + 1], base_bins[j]), 2.0f) : base_bins[S];
    const float lower = (j > 0) ? fdiv(fadd(base_bins[j], base_bins[j - 1]), 2.0f) : base_bins[0];
    b = fadd(lower, fmul(fsub(upper, lower), t));
  }
  return b;
}

}  // namespace fnr
