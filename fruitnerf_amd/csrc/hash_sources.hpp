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

// workgroup -> (level, sample block) of the (sample, level) kernels
__device__ __forceinline__ void decode_block(int b, int L, long long nsb, int& level, long long& sb) {
  if ((L & 7) == 0) {
    // XCD-aware: xcd = b % 8 owns L/8 levels, coarse levels paired with fine ones
    const int lpx = L >> 3;
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
