// hashgrid.hip — multiresolution hash-grid encode (HashEncoding torch semantics) and the fused
// proposal-network density kernel, for gfx950.
//
// Roofline: HBM/L2-bound random 8-byte gathers (8 corners x L levels per sample, 64 B/level
// algorithmic).  Layout decisions:
//   * features are written level-major [L][N] float2 so a wave's store is one contiguous 512 B run;
//   * the main-field encode launches one workgroup per (256 samples, level) and maps workgroup b to
//     XCD b%8 (observed dispatch order) so each XCD's private 4 MiB L2 only ever sees the table
//     slices of the two levels {x, L-1-x} it owns (one coarse = cache-friendly, one fine = 4 MiB);
//     a different placement changes speed only, never results.
#include "common.hpp"

namespace fnr {

// ------------------------------------------------------------------------------------------------
// position sources
// ------------------------------------------------------------------------------------------------
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

// ------------------------------------------------------------------------------------------------
// main-field encode: one thread per (sample, level)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void decode_block(int b, int L, long long nsb, int& level, long long& sb) {
  if ((L & 7) == 0) {
    // XCD-aware: xcd = b % 8 owns L/8 levels, coarse levels paired with fine ones
    const int lpx = L >> 3;
    const int xcd = b & 7;
    const long long q = b >> 3;
    const int li = (int)(q % lpx);
    sb = q / lpx;
    const int base = (li >> 1) * 8 + xcd;  // li even -> ascending from the coarse end
    level = (li & 1) ? (L - 1 - base) : base;
  } else {
    level = b % L;
    sb = b / L;
  }
}

template <class Source>
__global__ __launch_bounds__(256) void k_hash_encode(GridDev grid, Warp warp, Source src, long long N,
                                                     float2* __restrict__ feats, uint8_t* __restrict__ selector) {
  const long long nsb = (N + 255) / 256;
  int level;
  long long sb;
  decode_block(blockIdx.x, grid.n_levels, nsb, level, sb);
  const long long n = sb * 256 + threadIdx.x;
  if (n >= N) return;
  float px, py, pz, x[3];
  src.position(n, px, py, pz);
  const bool sel = warp_position(warp, px, py, pz, x);
  const uint32_t mask = (1u << grid.log2_T) - 1u;
  const float2* lt = grid.table + ((size_t)level << grid.log2_T);
  float2 f = grid_lookup(lt, x, grid.scalings[level], mask);
  feats[(size_t)level * N + n] = f;
  if (level == 0 && selector) selector[n] = sel ? 1 : 0;
}

template <class Source>
static int launch_encode(const fnr_grid* grid, const fnr_warp* warp, const Source& src, long long N, float* feats,
                         uint8_t* selector, void* stream) {
  FNR_CHECK_ARG(grid && warp && feats, "hash_encode: null argument");
  FNR_CHECK_ARG(grid->n_levels >= 1 && grid->n_levels <= FNR_MAX_LEVELS, "hash_encode: n_levels %d out of range",
                grid->n_levels);
  FNR_CHECK_ARG(grid->log2_hashmap_size >= 1 && grid->log2_hashmap_size <= 28, "hash_encode: log2_hashmap_size");
  if (N == 0) return FNR_OK;
  const long long nsb = (N + 255) / 256;
  const long long nblk = nsb * grid->n_levels;
  FNR_CHECK_ARG(nblk < (1ll << 31), "hash_encode: too many samples for one launch (%lld)", N);
  hipLaunchKernelGGL((k_hash_encode<Source>), dim3((unsigned)nblk), dim3(256), 0, as_stream(stream), make_grid(grid),
                     make_warp(warp), src, N, reinterpret_cast<float2*>(feats), selector);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

// ------------------------------------------------------------------------------------------------
// proposal network: hash(L levels) -> Linear(2L,H) ReLU Linear(H,1) -> trunc_exp * selector,
// one thread per sample, weights through scalar loads (uniform addresses), fp32 VALU.
// ------------------------------------------------------------------------------------------------
template <int L, int H>
__global__ __launch_bounds__(256) void k_prop_density(GridDev grid, Warp warp, RaySource src, long long N,
                                                      const float* __restrict__ w0, const float* __restrict__ b0,
                                                      const float* __restrict__ w1, const float* __restrict__ b1,
                                                      float* __restrict__ density, float2* __restrict__ feat_save) {
  const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float px, py, pz, x[3];
  src.position(n, px, py, pz);
  const bool sel = warp_position(warp, px, py, pz, x);
  const uint32_t mask = (1u << grid.log2_T) - 1u;
  float f[2 * L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    float2 v = grid_lookup(grid.table + ((size_t)l << grid.log2_T), x, grid.scalings[l], mask);
    f[2 * l] = v.x;
    f[2 * l + 1] = v.y;
    if (feat_save) feat_save[(size_t)l * N + n] = v;
  }
  float out = b1[0];
#pragma unroll
  for (int o = 0; o < H; ++o) {
    float a = b0[o];
#pragma unroll
    for (int k = 0; k < 2 * L; ++k) a = fmaf(w0[o * 2 * L + k], f[k], a);
    a = fmaxf(a, 0.0f);
    out = fmaf(w1[o], a, out);
  }
  density[n] = sel ? expf(out) : 0.0f;
}

// ------------------------------------------------------------------------------------------------
// backward: trilinear scatter-add of dL/dfeature into the gradient table (autograd of the 8-corner blend)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void grid_scatter(float* __restrict__ level_grad, const float (&x)[3], int scaling,
                                             uint32_t mask, float gx, float gy) {
  GridLevel g = grid_cell(x, scaling);
  uint32_t h[8];
  grid_corners(g, mask, h);
  const float ox = g.o[0], oy = g.o[1], oz = g.o[2];
  const float mx = 1.0f - ox, my = 1.0f - oy, mz = 1.0f - oz;
  // weights of f0..f7 in the oracle's blend
  const float wgt[8] = {ox * oy * oz, ox * my * oz, mx * my * oz, mx * oy * oz,
                        ox * oy * mz, ox * my * mz, mx * my * mz, mx * oy * mz};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float* dst = level_grad + 2 * (size_t)h[k];
    atomicAdd(dst, wgt[k] * gx);
    atomicAdd(dst + 1, wgt[k] * gy);
  }
}

template <class Source>
__global__ __launch_bounds__(256) void k_hash_scatter(GridDev grid, Warp warp, Source src, long long N,
                                                      const float2* __restrict__ d_feats) {
  const long long nsb = (N + 255) / 256;
  int level;
  long long sb;
  decode_block(blockIdx.x, grid.n_levels, nsb, level, sb);
  const long long n = sb * 256 + threadIdx.x;
  if (n >= N) return;
  const float2 gf = d_feats[(size_t)level * N + n];
  if (gf.x == 0.0f && gf.y == 0.0f) return;
  float px, py, pz, x[3];
  src.position(n, px, py, pz);
  warp_position(warp, px, py, pz, x);
  const uint32_t mask = (1u << grid.log2_T) - 1u;
  float* lg = reinterpret_cast<float*>(grid.table + ((size_t)level << grid.log2_T));
  grid_scatter(lg, x, grid.scalings[level], mask, gf.x, gf.y);
}

// ------------------------------------------------------------------------------------------------
// proposal network backward.  Persistent workgroups; per iteration 256 samples:
//   phase 1 (thread = sample): recompute the MLP from the saved features, d_out = d_sigma * trunc_exp'(out),
//            hidden gradients -> LDS, feature gradients -> scatter-add into the gradient table;
//   phase 2 (thread = weight): accumulate dW0[o][k], db0[o], dW1[o], db1 over the 256 samples from LDS.
// Weight gradients leave the workgroup once, at the end (one atomicAdd per weight per workgroup).
// ------------------------------------------------------------------------------------------------
template <int L, int H>
__global__ __launch_bounds__(256) void k_prop_bwd(GridDev grid_grad, Warp warp, RaySource src, long long N,
                                                  const float* __restrict__ w0, const float* __restrict__ b0,
                                                  const float* __restrict__ w1, const float* __restrict__ b1,
                                                  const float2* __restrict__ feat_save,
                                                  const float* __restrict__ d_density, float* __restrict__ g_w0,
                                                  float* __restrict__ g_b0, float* __restrict__ g_w1,
                                                  float* __restrict__ g_b1) {
  constexpr int K = 2 * L;
  __shared__ float s_dh[256][H + 1];   // d hidden (pre-activation)
  __shared__ float s_ha[256][H + 1];   // relu(hidden) * d_out  (for dW1)
  __shared__ float s_f[256][K + 1];    // input features
  __shared__ float s_do[256];          // d_out
  const int tid = threadIdx.x;
  constexpr int NW = H * K + H + H + 1;  // dW0, db0, dW1, db1
  float acc[2] = {0.0f, 0.0f};           // this thread's weight-gradient accumulators (tid, tid + 256)
  const uint32_t mask = (1u << grid_grad.log2_T) - 1u;
  const long long n_iter = (N + 255) / 256;
  for (long long it = blockIdx.x; it < n_iter; it += gridDim.x) {
    const long long n = it * 256 + tid;
    float f[K], dout = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) f[k] = 0.0f;
    float x[3] = {0.f, 0.f, 0.f};
    bool sel = false;
    if (n < N) {
      float px, py, pz;
      src.position(n, px, py, pz);
      sel = warp_position(warp, px, py, pz, x);
#pragma unroll
      for (int l = 0; l < L; ++l) {
        const float2 v = feat_save[(size_t)l * N + n];
        f[2 * l] = v.x;
        f[2 * l + 1] = v.y;
      }
    }
    float a[H];
    float out = b1[0];
#pragma unroll
    for (int o = 0; o < H; ++o) {
      float t = b0[o];
#pragma unroll
      for (int k = 0; k < K; ++k) t = fmaf(w0[o * K + k], f[k], t);
      a[o] = t;
      out = fmaf(w1[o], fmaxf(t, 0.0f), out);
    }
    if (n < N && sel) dout = d_density[n] * expf(fminf(fmaxf(out, -15.0f), 15.0f));  // trunc_exp backward
    float df[K];
#pragma unroll
    for (int k = 0; k < K; ++k) df[k] = 0.0f;
#pragma unroll
    for (int o = 0; o < H; ++o) {
      const float dh = (a[o] > 0.0f) ? dout * w1[o] : 0.0f;
      s_dh[tid][o] = dh;
      s_ha[tid][o] = fmaxf(a[o], 0.0f) * dout;
#pragma unroll
      for (int k = 0; k < K; ++k) df[k] = fmaf(dh, w0[o * K + k], df[k]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) s_f[tid][k] = f[k];
    s_do[tid] = dout;
    if (dout != 0.0f) {
#pragma unroll
      for (int l = 0; l < L; ++l) {
        float* lg = reinterpret_cast<float*>(grid_grad.table + ((size_t)l << grid_grad.log2_T));
        grid_scatter(lg, x, grid_grad.scalings[l], mask, df[2 * l], df[2 * l + 1]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int widx = tid + 256 * half;
      if (widx < NW) {
        float s = 0.0f;
        if (widx < H * K) {
          const int o = widx / K, k = widx - o * K;
          for (int q = 0; q < 256; ++q) s = fmaf(s_dh[q][o], s_f[q][k], s);
        } else if (widx < H * K + H) {
          const int o = widx - H * K;
          for (int q = 0; q < 256; ++q) s += s_dh[q][o];
        } else if (widx < H * K + 2 * H) {
          const int o = widx - H * K - H;
          for (int q = 0; q < 256; ++q) s += s_ha[q][o];
        } else {
          for (int q = 0; q < 256; ++q) s += s_do[q];
        }
        acc[half] += s;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int widx = tid + 256 * half;
    if (widx < NW && acc[half] != 0.0f) {
      if (widx < H * K) atomicAdd(&g_w0[widx], acc[half]);
      else if (widx < H * K + H) atomicAdd(&g_b0[widx - H * K], acc[half]);
      else if (widx < H * K + 2 * H) atomicAdd(&g_w1[widx - H * K - H], acc[half]);
      else atomicAdd(&g_b1[0], acc[half]);
    }
  }
}

}  // namespace fnr

using namespace fnr;

extern "C" int fnr_hash_encode_bwd(const fnr_grid* grid_grad, const fnr_warp* warp, const fnr_rays* rays,
                                   const float* euclid_bins, int S, const float* d_feats, void* stream) {
  FNR_CHECK_ARG(grid_grad && warp && rays && euclid_bins && d_feats && S > 0, "hash_encode_bwd: null argument");
  FNR_CHECK_ARG(grid_grad->n_levels >= 1 && grid_grad->n_levels <= FNR_MAX_LEVELS, "hash_encode_bwd: n_levels");
  const long long N = rays->n_rays * (long long)S;
  if (N == 0) return FNR_OK;
  RaySource src{make_rays(rays), euclid_bins, S};
  const long long nblk = ((N + 255) / 256) * grid_grad->n_levels;
  FNR_CHECK_ARG(nblk < (1ll << 31), "hash_encode_bwd: too many samples");
  hipLaunchKernelGGL((k_hash_scatter<RaySource>), dim3((unsigned)nblk), dim3(256), 0, as_stream(stream),
                     make_grid(grid_grad), make_warp(warp), src, N, reinterpret_cast<const float2*>(d_feats));
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

extern "C" int fnr_prop_density_bwd(const fnr_prop_net* net, const fnr_prop_net* grads, const fnr_warp* warp,
                                    const fnr_rays* rays, const float* euclid_bins, int S, const float* feat_save,
                                    const float* d_density, void* stream) {
  FNR_CHECK_ARG(net && grads && warp && rays && euclid_bins && feat_save && d_density && S > 0,
                "prop_density_bwd: null argument");
  FNR_UNSUPPORTED(net->hidden_dim == 16, "prop_density_bwd: hidden_dim %d not built (16 only)", net->hidden_dim);
  FNR_CHECK_ARG(grads->grid.table && grads->w0 && grads->b0 && grads->w1 && grads->b1,
                "prop_density_bwd: null gradient pointer");
  const long long N = rays->n_rays * (long long)S;
  if (N == 0) return FNR_OK;
  RaySource src{make_rays(rays), euclid_bins, S};
  long long blocks = (N + 255) / 256;
  const long long max_blocks = 8ll * device_cu_count();
  if (blocks > max_blocks) blocks = max_blocks;
  GridDev gg = make_grid(&grads->grid);
  Warp w = make_warp(warp);
  const float2* fs = reinterpret_cast<const float2*>(feat_save);
#define FNR_PROPB_CASE(LL)                                                                                       \
  case LL:                                                                                                       \
    hipLaunchKernelGGL((k_prop_bwd<LL, 16>), dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), gg, w, src, N, \
                       net->w0, net->b0, net->w1, net->b1, fs, d_density, grads->w0, grads->b0, grads->w1,       \
                       grads->b1);                                                                               \
    break;
  switch (net->grid.n_levels) {
    FNR_PROPB_CASE(1)
    FNR_PROPB_CASE(2)
    FNR_PROPB_CASE(3)
    FNR_PROPB_CASE(4)
    FNR_PROPB_CASE(5)
    FNR_PROPB_CASE(6)
    FNR_PROPB_CASE(7)
    FNR_PROPB_CASE(8)
    default:
      FNR_UNSUPPORTED(false, "prop_density_bwd: n_levels %d not built (1..8)", net->grid.n_levels);
  }
#undef FNR_PROPB_CASE
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

extern "C" int fnr_hash_encode_fwd(const fnr_grid* grid, const fnr_warp* warp, const fnr_rays* rays,
                                   const float* euclid_bins, int S, float* feats, uint8_t* selector, void* stream) {
  FNR_CHECK_ARG(rays && euclid_bins && S > 0, "hash_encode_fwd: null rays/bins or S<=0");
  RaySource src{make_rays(rays), euclid_bins, S};
  return launch_encode(grid, warp, src, rays->n_rays * (long long)S, feats, selector, stream);
}

extern "C" int fnr_hash_encode_lattice(const fnr_grid* grid, const fnr_warp* warp, const fnr_lattice* lat,
                                       int64_t ray_begin, int64_t n_rays, float* feats, uint8_t* selector,
                                       void* stream) {
  FNR_CHECK_ARG(lat && lat->xs && lat->ys && lat->zs, "hash_encode_lattice: null lattice");
  FNR_CHECK_ARG(ray_begin >= 0 && n_rays >= 0 && ray_begin + n_rays <= (int64_t)lat->n_x * lat->n_y,
                "hash_encode_lattice: ray range [%lld,+%lld) outside %d x %d lattice", (long long)ray_begin,
                (long long)n_rays, lat->n_x, lat->n_y);
  LatticeSource src{lat->xs, lat->ys, lat->zs, lat->n_y, lat->n_z, ray_begin};
  return launch_encode(grid, warp, src, n_rays * (long long)lat->n_z, feats, selector, stream);
}

extern "C" int fnr_prop_density_fwd(const fnr_prop_net* net, const fnr_warp* warp, const fnr_rays* rays,
                                    const float* euclid_bins, int S, float* density, float* feat_save, void* stream) {
  FNR_CHECK_ARG(net && warp && rays && euclid_bins && density && S > 0, "prop_density_fwd: null argument");
  FNR_UNSUPPORTED(net->hidden_dim == 16, "prop_density_fwd: hidden_dim %d not built (16 only)", net->hidden_dim);
  const long long N = rays->n_rays * (long long)S;
  if (N == 0) return FNR_OK;
  RaySource src{make_rays(rays), euclid_bins, S};
  const unsigned nblk = (unsigned)((N + 255) / 256);
  GridDev g = make_grid(&net->grid);
  Warp w = make_warp(warp);
  float2* fs = reinterpret_cast<float2*>(feat_save);
#define FNR_PROP_CASE(LL)                                                                                      \
  case LL:                                                                                                     \
    hipLaunchKernelGGL((k_prop_density<LL, 16>), dim3(nblk), dim3(256), 0, as_stream(stream), g, w, src, N,    \
                       net->w0, net->b0, net->w1, net->b1, density, fs);                                       \
    break;
  switch (net->grid.n_levels) {
    FNR_PROP_CASE(1)
    FNR_PROP_CASE(2)
    FNR_PROP_CASE(3)
    FNR_PROP_CASE(4)
    FNR_PROP_CASE(5)
    FNR_PROP_CASE(6)
    FNR_PROP_CASE(7)
    FNR_PROP_CASE(8)
    default:
      FNR_UNSUPPORTED(false, "prop_density_fwd: n_levels %d not built (1..8)", net->grid.n_levels);
  }
#undef FNR_PROP_CASE
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}
