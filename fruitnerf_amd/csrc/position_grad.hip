// position_grad.hip — dL/d(ray origins), dL/d(ray directions): the input gradient of the hash grids.
//
// In the reference the sample positions carry autograd history (fruit_field.py:171-182: Frustums.get_positions ->
// SceneContraction -> (x+2)/4 -> * selector -> HashEncoding), so loss.backward() also produces the gradient of the
// rays the camera-pose optimiser consumes (fruit_nerf_config.py:39-43).  The sampler's bins are detached
// (PDFSampler), SHEncoding runs under no_grad and the semantic branch sees detached geo features, so the ONLY path
// to the rays is  p = o + d (t0 + t1)/2  ->  x(p)  ->  trilinear offsets of every level.
//   main field: the forward encode (k_hash_encode) already holds the 8 corner rows of every (sample, level), so in
//     training it also stores the input Jacobian J = scaling * d(blend)/d(offset) (6 floats per sample and level);
//     k_position_from_jacobian (wave = ray, lane = sample) contracts it with d_feats, applies the transposed Jacobian
//     of the contraction (or AABB normalisation) and of p = o + d t, and reduces over the ray — 24 B written + 32 B
//     read per (sample, level) instead of re-gathering 64 B of random table rows (105 us -> ~25 us per step);
//   proposal nets: k_prop_bwd re-gathers (5 levels, tables L2-resident) and k_position_reduce finishes;
//   k_hash_input_grad + k_position_reduce remain as the gather-based path for callers without a saved Jacobian.
#include "hash_sources.hpp"
#include "sequencer.hpp"

namespace fnr {

constexpr int PG_SPT = 2;  // samples per thread (gathers in flight), as in k_hash_encode

template <class Source>
__global__ __launch_bounds__(256) void k_hash_input_grad(GridDev grid, Warp warp, Source src, long long N,
                                                         const float2* __restrict__ d_feats,
                                                         float4* __restrict__ partial) {
  const long long nsb = (N + 256 * PG_SPT - 1) / (256 * PG_SPT);
  int level;
  long long sb;
  decode_block(blockIdx.x, grid.n_levels, nsb, level, sb);
  const uint32_t mask = (1u << grid.log2_T) - 1u;
  const float2* lt = grid.table + ((size_t)level << grid.log2_T);
  const int scaling = grid.scalings[level];
  uint32_t h[PG_SPT][8];
  float o[PG_SPT][3];
  bool sel[PG_SPT];
  long long n[PG_SPT];
  float2 gf[PG_SPT];
#pragma unroll
  for (int u = 0; u < PG_SPT; ++u) {
    n[u] = sb * (256 * PG_SPT) + u * 256 + threadIdx.x;
    const long long nn = n[u] < N ? n[u] : N - 1;
    float px, py, pz, x[3];
    src.position(nn, px, py, pz);
    sel[u] = warp_position(warp, px, py, pz, x);
    const GridLevel g = grid_cell(x, scaling);
    grid_corners(g, mask, h[u]);
    o[u][0] = g.o[0], o[u][1] = g.o[1], o[u][2] = g.o[2];
    gf[u] = d_feats[(size_t)level * N + nn];
  }
  const bool odd = threadIdx.x & 1;  // gathers by lane pairs (common.hpp)
  PairedRows rows[PG_SPT];
#pragma unroll
  for (int u = 0; u < PG_SPT; ++u) rows[u] = paired_rows(h[u], odd);
  float2 va[PG_SPT][4], vb[PG_SPT][4];
#pragma unroll
  for (int u = 0; u < PG_SPT; ++u) {
#pragma unroll
    for (int q = 0; q < 4; ++q) va[u][q] = lt[rows[u].a[q]];
#pragma unroll
    for (int q = 0; q < 4; ++q) vb[u][q] = lt[rows[u].b[q]];
  }
  float2 v[PG_SPT][8];
#pragma unroll
  for (int u = 0; u < PG_SPT; ++u) paired_values(va[u], vb[u], odd, v[u]);
#pragma unroll
  for (int u = 0; u < PG_SPT; ++u) {
    if (n[u] >= N) continue;
    float d[8], g[3];
#pragma unroll
    for (int k = 0; k < 8; ++k) d[k] = fmaf(gf[u].x, v[u][k].x, gf[u].y * v[u][k].y);
    blend_input_grad(d, o[u], g);
    const float s = sel[u] ? (float)scaling : 0.0f;  // positions * selector: no gradient outside the unit cube
    partial[(size_t)level * N + n[u]] = make_float4(s * g[0], s * g[1], s * g[2], 0.0f);
  }
}

// transposed Jacobian of warp_position applied to g (gradient w.r.t. the unit-cube position) at world position p
__device__ __forceinline__ void warp_backward(const Warp& w, const float (&p)[3], float (&g)[3]) {
  if (w.mode == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) g[a] *= 0.25f;  // x = (x' + 2) / 4
    const float ax = fabsf(p[0]), ay = fabsf(p[1]), az = fabsf(p[2]);
    const float m = fmaxf(ax, fmaxf(ay, az));
    if (!(m < 1.0f)) {
      // x' = p c(m), c = 2/m - 1/m^2, m = |p|_inf:  dL/dp_b = c g_b + [b == argmax] sign(p_b) c'(m) (g . p)
      const float inv = 1.0f / m;
      const float c = 2.0f * inv - inv * inv;
      const float dc = -2.0f * inv * inv + 2.0f * inv * inv * inv;
      const float dot = g[0] * p[0] + g[1] * p[1] + g[2] * p[2];
      const int j = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
#pragma unroll
      for (int a = 0; a < 3; ++a) g[a] *= c;
      const float e = copysignf(1.0f, p[j]) * dc * dot;
      g[0] += (j == 0) ? e : 0.0f;
      g[1] += (j == 1) ? e : 0.0f;
      g[2] += (j == 2) ? e : 0.0f;
    }
  } else {
#pragma unroll
    for (int a = 0; a < 3; ++a) g[a] /= w.len[a];
  }
}

// wave = ray.  partial: [n_levels][N] float4.  d_origins / d_directions [R,3] are accumulated (+=).
__global__ __launch_bounds__(256) void k_position_reduce(Warp warp, RaysDev rays, const float* __restrict__ euclid, int S,
                                                         int n_levels, const float4* __restrict__ partial,
                                                         float* __restrict__ d_origins, float* __restrict__ d_directions) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long r = (long long)blockIdx.x * 4 + wave;
  if (r >= rays.n_rays) return;
  const long long N = rays.n_rays * (long long)S;
  const float* o = rays.origins + 3 * r;
  const float* d = rays.directions + 3 * r;
  float go[3] = {0.f, 0.f, 0.f}, gd[3] = {0.f, 0.f, 0.f};
  for (int k = lane; k < S; k += 64) {
    const long long n = r * S + k;
    float g[3] = {0.f, 0.f, 0.f};
    for (int l = 0; l < n_levels; ++l) {
      const float4 q = partial[(size_t)l * N + n];
      g[0] += q.x, g[1] += q.y, g[2] += q.z;
    }
    const float* b = euclid + r * (S + 1) + k;
    const float tm = (b[0] + b[1]) * 0.5f;
    float p[3];
    ray_position(o, d, b[0], b[1], p[0], p[1], p[2]);
    warp_backward(warp, p, g);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      go[a] += g[a];
      gd[a] += tm * g[a];
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    go[a] = wave_sum(go[a]);
    gd[a] = wave_sum(gd[a]);
  }
  if (lane < 3) {
    const float vo = (lane == 0) ? go[0] : (lane == 1) ? go[1] : go[2];
    const float vd = (lane == 0) ? gd[0] : (lane == 1) ? gd[1] : gd[2];
    d_origins[3 * r + lane] += vo;
    d_directions[3 * r + lane] += vd;
  }
}

// k_position_reduce over several sources in ONE launch (a training step with a camera optimiser has three: the two
// proposal levels' d_position and the main field's): wave = ray, the sources' per-ray sums are formed one after the
// other in source order and WRITTEN (accumulate = 0: the [R,3] buffers need no zero fill) or added.
struct PosSource {
  Warp warp;
  const float* euclid;
  const float4* partial;
  int S, n_levels;
};
struct PosSources {
  int n;
  PosSource s[FNR_MAX_POSITION_SOURCES];
};
__global__ __launch_bounds__(256) void k_position_reduce_multi(PosSources src, RaysDev rays, int accumulate,
                                                               float* __restrict__ d_origins,
                                                               float* __restrict__ d_directions) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long r = (long long)blockIdx.x * 4 + wave;
  if (r >= rays.n_rays) return;
  const float* o = rays.origins + 3 * r;
  const float* d = rays.directions + 3 * r;
  float tot_o[3] = {0.f, 0.f, 0.f}, tot_d[3] = {0.f, 0.f, 0.f};
  for (int q = 0; q < src.n; ++q) {
    const PosSource& ps = src.s[q];
    const int S = ps.S;
    const long long N = rays.n_rays * (long long)S;
    float go[3] = {0.f, 0.f, 0.f}, gd[3] = {0.f, 0.f, 0.f};
    for (int k = lane; k < S; k += 64) {
      const long long n = r * S + k;
      float g[3] = {0.f, 0.f, 0.f};
      for (int l = 0; l < ps.n_levels; ++l) {
        const float4 v = ps.partial[(size_t)l * N + n];
        g[0] += v.x, g[1] += v.y, g[2] += v.z;
      }
      const float* b = ps.euclid + r * (S + 1) + k;
      const float tm = (b[0] + b[1]) * 0.5f;
      float p[3];
      ray_position(o, d, b[0], b[1], p[0], p[1], p[2]);
      warp_backward(ps.warp, p, g);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        go[a] += g[a];
        gd[a] += tm * g[a];
      }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {  // the same per-source sums k_position_reduce forms, added in source order
      tot_o[a] += wave_sum(go[a]);
      tot_d[a] += wave_sum(gd[a]);
    }
  }
  if (lane < 3) {
    const float vo = (lane == 0) ? tot_o[0] : (lane == 1) ? tot_o[1] : tot_o[2];
    const float vd = (lane == 0) ? tot_d[0] : (lane == 1) ? tot_d[1] : tot_d[2];
    if (accumulate) {
      d_origins[3 * r + lane] += vo;
      d_directions[3 * r + lane] += vd;
    } else {
      d_origins[3 * r + lane] = vo;
      d_directions[3 * r + lane] = vd;
    }
  }
}

// g[a] = sum over levels l (in level order) of gf_l . J_l[a].  The loads of 8 levels are issued before the first use: the
// plain loop waited for one level's four loads at a time (SQ_WAIT_ANY 0.91 of the kernel's wave-cycles).
__device__ __forceinline__ void contract_jacobian(const float2* __restrict__ jac, const float2* __restrict__ d_feats,
                                                  long long N, long long n, int n_levels, float (&g)[3]) {
  g[0] = g[1] = g[2] = 0.0f;
  int l = 0;
  for (; l + 8 <= n_levels; l += 8) {
    float2 gf[8], j[8][3];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      gf[u] = d_feats[(size_t)(l + u) * N + n];
#pragma unroll
      for (int a = 0; a < 3; ++a) j[u][a] = ntc_load<NT_JAC_LD>(&jac[((size_t)(l + u) * 3 + a) * N + n]);   // the Jacobian's only use
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int a = 0; a < 3; ++a) g[a] += gf[u].x * j[u][a].x + gf[u].y * j[u][a].y;
  }
  for (; l < n_levels; ++l) {
    const float2 gf = d_feats[(size_t)l * N + n];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float2 j = ntc_load<NT_JAC_LD>(&jac[((size_t)l * 3 + a) * N + n]);
      g[a] += gf.x * j.x + gf.y * j.y;
    }
  }
}

// thread = sample: d_pos [N] float4 = the contraction alone (fnr_field_mlp_bwd_rays in fp32 mode; the bf16-pipe base
// kernel forms it from the dL/dfeats it holds in registers)
__global__ __launch_bounds__(256) void k_position_contract(long long N, int n_levels, const float2* __restrict__ jac,
                                                           const float2* __restrict__ d_feats, float4* __restrict__ d_pos) {
  const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float g[3];
  contract_jacobian(jac, d_feats, N, n, n_levels, g);
  d_pos[n] = make_float4(g[0], g[1], g[2], 0.0f);
}

int position_contract(long long N, int n_levels, const float2* jac, const float2* d_feats, float4* d_pos, hipStream_t st) {
  hipLaunchKernelGGL(k_position_contract, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, N, n_levels, jac, d_feats, d_pos);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

// wave = ray, lane = sample: d(loss)/d(unit-cube position) = sum over levels and features of gf * J, with the input
// Jacobian J [L][3][N] float2 (axis-major, feature pair) saved by the forward encode — no table gathers in the
// backward pass; then the same warp / frustum chain as k_position_reduce.
__global__ __launch_bounds__(256) void k_position_from_jacobian(Warp warp, RaysDev rays, const float* __restrict__ euclid,
                                                                int S, int n_levels, const float2* __restrict__ jac,
                                                                const float2* __restrict__ d_feats,
                                                                float* __restrict__ d_origins,
                                                                float* __restrict__ d_directions) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long r = (long long)blockIdx.x * 4 + wave;
  if (r >= rays.n_rays) return;
  const long long N = rays.n_rays * (long long)S;
  const float* o = rays.origins + 3 * r;
  const float* d = rays.directions + 3 * r;
  float go[3] = {0.f, 0.f, 0.f}, gd[3] = {0.f, 0.f, 0.f};
  for (int k = lane; k < S; k += 64) {
    const long long n = r * S + k;
    float g[3];
    contract_jacobian(jac, d_feats, N, n, n_levels, g);
    const float* b = euclid + r * (S + 1) + k;
    const float tm = (b[0] + b[1]) * 0.5f;
    float p[3];
    ray_position(o, d, b[0], b[1], p[0], p[1], p[2]);
    warp_backward(warp, p, g);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      go[a] += g[a];
      gd[a] += tm * g[a];
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    go[a] = wave_sum(go[a]);
    gd[a] = wave_sum(gd[a]);
  }
  if (lane < 3) {
    const float vo = (lane == 0) ? go[0] : (lane == 1) ? go[1] : go[2];
    const float vd = (lane == 0) ? gd[0] : (lane == 1) ? gd[1] : gd[2];
    d_origins[3 * r + lane] += vo;
    d_directions[3 * r + lane] += vd;
  }
}

}  // namespace fnr

using namespace fnr;

extern "C" int fnr_position_grad_from_jacobian(const fnr_warp* warp, const fnr_rays* rays, const float* euclid_bins, int S,
                                               int n_levels, const float* jacobian, const float* d_feats,
                                               float* d_origins, float* d_directions, void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_position_grad_from_jacobian");
  FNR_CHECK_ARG(warp && rays && euclid_bins && jacobian && d_feats && d_origins && d_directions && S > 0 && n_levels >= 1,
                "position_grad_from_jacobian: null argument");
  if (rays->n_rays == 0) return FNR_OK;
  FNR_PROF(OP_POSITION_GRAD, rays->n_rays * (long long)S);
  hipLaunchKernelGGL(k_position_from_jacobian, dim3((unsigned)((rays->n_rays + 3) / 4)), dim3(256), 0, as_stream(stream),
                     make_warp(warp), make_rays(rays), euclid_bins, S, n_levels, reinterpret_cast<const float2*>(jacobian),
                     reinterpret_cast<const float2*>(d_feats), d_origins, d_directions);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}


extern "C" int fnr_hash_encode_input_grad(const fnr_grid* grid, const fnr_warp* warp, const fnr_rays* rays,
                                          const float* euclid_bins, int S, const float* d_feats, float* partial,
                                          void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_hash_encode_input_grad");
  FNR_CHECK_ARG(grid && warp && rays && euclid_bins && d_feats && partial && S > 0, "hash_encode_input_grad: null argument");
  FNR_CHECK_ARG(grid->n_levels >= 1 && grid->n_levels <= FNR_MAX_LEVELS, "hash_encode_input_grad: n_levels");
  const long long N = rays->n_rays * (long long)S;
  if (N == 0) return FNR_OK;
  RaySource src{make_rays(rays), euclid_bins, S};
  const long long nsb = (N + 256 * PG_SPT - 1) / (256 * PG_SPT);
  const long long nblk = nsb * grid->n_levels;
  FNR_CHECK_ARG(nblk < (1ll << 31), "hash_encode_input_grad: too many samples for one launch (%lld)", N);
  FNR_PROF(OP_POSITION_GRAD, N);
  hipLaunchKernelGGL((k_hash_input_grad<RaySource>), dim3((unsigned)nblk), dim3(256), 0, as_stream(stream), make_grid(grid),
                     make_warp(warp), src, N, reinterpret_cast<const float2*>(d_feats), reinterpret_cast<float4*>(partial));
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

extern "C" int fnr_position_grad_reduce_multi(int n_sources, const fnr_warp* const* warps, const fnr_rays* rays,
                                              const float* const* euclid_bins, const int* S, const int* n_levels,
                                              const float* const* partials, int accumulate, float* d_origins,
                                              float* d_directions, void* stream) {
  if (seq::recording() && n_sources >= 1 && n_sources <= FNR_MAX_POSITION_SOURCES && warps && rays && euclid_bins && S &&
      n_levels && partials) {
    constexpr int M = FNR_MAX_POSITION_SOURCES;
    std::array<fnr_warp, M> warps_{};
    for (int q = 0; q < n_sources; ++q)
      if (warps[q]) warps_[q] = *warps[q];
    const fnr_rays rays_ = *rays;
    const auto eu_ = seq::copy_n<const float*, M>(euclid_bins, n_sources);
    const auto S_ = seq::copy_n<int, M>(S, n_sources);
    const auto lv_ = seq::copy_n<int, M>(n_levels, n_sources);
    const auto pa_ = seq::copy_n<const float*, M>(partials, n_sources);
    seq::push("fnr_position_grad_reduce_multi", [=](const fnr_step_scalars*) {
      const fnr_warp* w_[M];
      for (int q = 0; q < M; ++q) w_[q] = &warps_[q];
      return fnr_position_grad_reduce_multi(n_sources, w_, &rays_, eu_.data(), S_.data(), lv_.data(), pa_.data(), accumulate,
                                            d_origins, d_directions, stream);
    });
  }
  FNR_CHECK_ARG(n_sources >= 1 && n_sources <= FNR_MAX_POSITION_SOURCES && warps && rays && euclid_bins && S && n_levels &&
                    partials && d_origins && d_directions,
                "position_grad_reduce_multi: bad argument (1..%d sources)", FNR_MAX_POSITION_SOURCES);
  if (rays->n_rays == 0) return FNR_OK;
  PosSources src;
  src.n = n_sources;
  long long units = 0;
  for (int q = 0; q < n_sources; ++q) {
    FNR_CHECK_ARG(warps[q] && euclid_bins[q] && partials[q] && S[q] > 0 && n_levels[q] >= 1,
                  "position_grad_reduce_multi: source %d", q);
    src.s[q].warp = make_warp(warps[q]);
    src.s[q].euclid = euclid_bins[q];
    src.s[q].partial = reinterpret_cast<const float4*>(partials[q]);
    src.s[q].S = S[q];
    src.s[q].n_levels = n_levels[q];
    units += rays->n_rays * (long long)S[q];
  }
  FNR_PROF(OP_POSITION_GRAD, units);
  hipLaunchKernelGGL(k_position_reduce_multi, dim3((unsigned)((rays->n_rays + 3) / 4)), dim3(256), 0, as_stream(stream), src,
                     make_rays(rays), accumulate, d_origins, d_directions);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

extern "C" int fnr_position_grad_reduce(const fnr_warp* warp, const fnr_rays* rays, const float* euclid_bins, int S,
                                        int n_levels, const float* partial, float* d_origins, float* d_directions,
                                        void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_position_grad_reduce");
  FNR_CHECK_ARG(warp && rays && euclid_bins && partial && d_origins && d_directions && S > 0 && n_levels >= 1,
                "position_grad_reduce: null argument");
  if (rays->n_rays == 0) return FNR_OK;
  FNR_PROF(OP_POSITION_GRAD, rays->n_rays * (long long)S);
  hipLaunchKernelGGL(k_position_reduce, dim3((unsigned)((rays->n_rays + 3) / 4)), dim3(256), 0, as_stream(stream),
                     make_warp(warp), make_rays(rays), euclid_bins, S, n_levels, reinterpret_cast<const float4*>(partial),
                     d_origins, d_directions);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}
