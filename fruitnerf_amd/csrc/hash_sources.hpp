// hash_sources.hpp — where sample positions come from (shared by the encode and scatter kernels).
#pragma once
#include "common.hpp"

namespace fnr {

// sample n = (ray n / S, bin n % S) at the bin midpoint of a RayBundle (Frustums.get_positions)
struct RaySource {
  RaysDev rays;
  const float* euclid;  // [R, S+1]
  int S;
  __device__ __forceinline__ void position(long long n, float& px, float& py, float& pz) const {
    long long r = n / S;
    int k = (int)(n - r * S);
    const float* b = euclid + r * (S + 1) + k;
    ray_position(rays.origins + 3 * r, rays.directions + 3 * r, b[0], b[1], px, py, pz);
  }
};

// sample n of an export batch on the orthographic lattice (data/fruit_datamanager.py:71-121)
struct LatticeSource {
  const float* xs;
  const float* ys;
  const float* zs;
  int n_y, n_z;
  long long ray_begin;
  __device__ __forceinline__ void position(long long n, float& px, float& py, float& pz) const {
    long long r = n / n_z;
    int k = (int)(n - r * n_z);
    r += ray_begin;
    long long ix = r / n_y;
    int iy = (int)(r - ix * n_y);
    px = xs[ix];
    py = ys[iy];
    pz = zs[k];
  }
};

// d(feature . g)/d(offset) for the oracle's blend (grid_interp): weights are products of o (ceil side) or 1 - o
// (floor side) per axis, corner order h0..h7 = ccc, cfc, ffc, fcc, ccf, cff, fff, fcf
__device__ __forceinline__ void blend_input_grad(const float (&d)[8], const float (&o)[3], float (&g)[3]) {
  const float ox = o[0], oy = o[1], oz = o[2];
  const float mx = 1.0f - ox, my = 1.0f - oy, mz = 1.0f - oz;
  g[0] = oz * (oy * (d[0] - d[3]) + my * (d[1] - d[2])) + mz * (oy * (d[4] - d[7]) + my * (d[5] - d[6]));
  g[1] = oz * (ox * (d[0] - d[1]) + mx * (d[3] - d[2])) + mz * (ox * (d[4] - d[5]) + mx * (d[7] - d[6]));
  g[2] = oy * (ox * (d[0] - d[4]) + mx * (d[3] - d[7])) + my * (ox * (d[1] - d[5]) + mx * (d[2] - d[6]));
}

// workgroup -> (level, sample block) of the (sample, level) kernels
__device__ __forceinline__ void decode_block(int b, int L, long long nsb, int& level, long long& sb) {
  if ((L & 7) == 0) {
    // XCD-aware: xcd = b % 8 owns L/8 levels, coarse levels paired with fine ones
    const int xcd = b & 7;
    const long long q = b >> 3;
    // one level at a time per XCD (all sample blocks of its coarse level, then its fine level): two 4 MiB tables
    // alternating in one 4 MiB L2 evicted each other
    const int li = (int)(q / nsb);
    sb = q - (long long)li * nsb;
    const int base = (li >> 1) * 8 + xcd;  // li even -> ascending from the coarse end
    level = (li & 1) ? (L - 1 - base) : base;
  } else {
    level = b % L;
    sb = b / L;
  }
}

}  // namespace fnr
