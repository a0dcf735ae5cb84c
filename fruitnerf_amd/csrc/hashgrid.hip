// hashgrid.hip — multiresolution hash-grid encode (HashEncoding torch semantics) and the fused
// proposal-network density kernel, for gfx950 (forward; the backward lives in hash_scatter.hip).
//
// Roofline: HBM/L2-bound random 8-byte gathers (8 corners x L levels per sample, 64 B/level
// algorithmic).  Layout decisions:
//   * features are written level-major [L][N] float2 so a wave's store is one contiguous 512 B run;
//   * the main-field encode launches one workgroup per (256 samples, level) and maps workgroup b to
//     XCD b%8 (observed dispatch order) so each XCD's private 4 MiB L2 only ever sees the table
//     slices of the two levels {x, L-1-x} it owns (one coarse = cache-friendly, one fine = 4 MiB), one after
//     the other;
//     a different placement changes speed only, never results.
#include <type_traits>

#include "hash_sources.hpp"
#include "sequencer.hpp"

namespace fnr {

// ------------------------------------------------------------------------------------------------
// main-field encode: one thread per (sample, level)
// ------------------------------------------------------------------------------------------------
constexpr int ENC_SPT = 2;  // samples per thread: 16 independent 8-byte gathers in flight hide the L2-hit latency

template <class Source>
__global__ __launch_bounds__(256) void k_hash_encode(GridDev grid, Warp warp, Source src, long long N,
                                                     float2* __restrict__ feats, uint8_t* __restrict__ selector,
                                                     float2* __restrict__ jac) {
  const long long nsb = (N + 256 * ENC_SPT - 1) / (256 * ENC_SPT);
  int level;
  long long sb;
  decode_block(blockIdx.x, grid.n_levels, nsb, level, sb);
  const uint32_t mask = (1u << grid.log2_T) - 1u;
  const float2* lt = grid.table + ((size_t)level << grid.log2_T);
  const int scaling = grid.scalings[level];
  uint32_t h[ENC_SPT][8];
  float o[ENC_SPT][3];
  bool sel[ENC_SPT];
  long long n[ENC_SPT];
#pragma unroll
  for (int u = 0; u < ENC_SPT; ++u) {
    n[u] = sb * (256 * ENC_SPT) + u * 256 + threadIdx.x;
    const long long nn = n[u] < N ? n[u] : N - 1;
    float px, py, pz, x[3];
    src.position(nn, px, py, pz);
    sel[u] = warp_position(warp, px, py, pz, x);
    const GridLevel g = grid_cell(x, scaling);
    grid_corners(g, mask, h[u]);
    o[u][0] = g.o[0], o[u][1] = g.o[1], o[u][2] = g.o[2];
  }
  // lane pairs gather the x-neighbour corners side by side (common.hpp: corner gathers by lane pairs)
  const bool odd = threadIdx.x & 1;
  PairedRows rows[ENC_SPT];
#pragma unroll
  for (int u = 0; u < ENC_SPT; ++u) rows[u] = paired_rows(h[u], odd);
  float2 va[ENC_SPT][4], vb[ENC_SPT][4];
#pragma unroll
  for (int u = 0; u < ENC_SPT; ++u) {
#pragma unroll
    for (int j = 0; j < 4; ++j) va[u][j] = lt[rows[u].a[j]];
#pragma unroll
    for (int j = 0; j < 4; ++j) vb[u][j] = lt[rows[u].b[j]];
  }
  float2 v[ENC_SPT][8];
#pragma unroll
  for (int u = 0; u < ENC_SPT; ++u) paired_values(va[u], vb[u], odd, v[u]);
#pragma unroll
  for (int u = 0; u < ENC_SPT; ++u) {
    if (n[u] >= N) continue;
    feats[(size_t)level * N + n[u]] = grid_interp(v[u], o[u]);
    if (level == 0 && selector) selector[n[u]] = sel[u] ? 1 : 0;
    if (jac) {
      // input Jacobian for the ray gradients (position_grad.hip): d feat_f / d x_a = scaling * d(blend)/d(offset_a)
      // of the corner values, zero outside the unit cube (positions * selector)
      float dx[8], dy[8], gx[3], gy[3];
#pragma unroll
      for (int k = 0; k < 8; ++k) dx[k] = v[u][k].x, dy[k] = v[u][k].y;
      blend_input_grad(dx, o[u], gx);
      blend_input_grad(dy, o[u], gy);
      const float s = sel[u] ? (float)scaling : 0.0f;
#pragma unroll
      // (streaming stores: the 75 MB Jacobian is read once, by the MLP backward's base branch — as plain stores it pushed
      //  the table's own rows out of the caches: k_hash_encode 82 -> 72 us with the hint here and on the other
      //  write-once / read-once streams of the step, common.hpp)
      for (int a = 0; a < 3; ++a) ntc_store<NT_JAC_ST>(&jac[((size_t)level * 3 + a) * N + n[u]], make_float2(s * gx[a], s * gy[a]));
    }
  }
}

template <class Source>
static int launch_encode(const fnr_grid* grid, const fnr_warp* warp, const Source& src, long long N, float* feats,
                         uint8_t* selector, float* jacobian, void* stream) {
  FNR_CHECK_ARG(grid && warp && feats, "hash_encode: null argument");
  FNR_CHECK_ARG(grid->n_levels >= 1 && grid->n_levels <= FNR_MAX_LEVELS, "hash_encode: n_levels %d out of range",
                grid->n_levels);
  FNR_CHECK_ARG(grid->log2_hashmap_size >= 1 && grid->log2_hashmap_size <= 28, "hash_encode: log2_hashmap_size");
  if (N == 0) return FNR_OK;
  const long long nsb = (N + 256 * ENC_SPT - 1) / (256 * ENC_SPT);
  const long long nblk = nsb * grid->n_levels;
  FNR_CHECK_ARG(nblk < (1ll << 31), "hash_encode: too many samples for one launch (%lld)", N);
  constexpr int prof_op = std::is_same<Source, LatticeSource>::value ? OP_ENCODE_LATTICE : OP_ENCODE_FWD;
  FNR_PROF(prof_op, N);
  hipLaunchKernelGGL((k_hash_encode<Source>), dim3((unsigned)nblk), dim3(256), 0, as_stream(stream), make_grid(grid),
                     make_warp(warp), src, N, reinterpret_cast<float2*>(feats), selector,
                     reinterpret_cast<float2*>(jacobian));
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
    if (feat_save) ntc_store<NT_PROP_FEATS>(&feat_save[(size_t)l * N + n], v);   // read once, by k_prop_bwd on the steps that train the network
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

}  // namespace fnr

using namespace fnr;

extern "C" int fnr_hash_encode_fwd(const fnr_grid* grid, const fnr_warp* warp, const fnr_rays* rays,
                                   const float* euclid_bins, int S, float* feats, uint8_t* selector, float* jacobian,
                                   void* stream) {
  if (seq::recording() && grid && warp && rays) {
    const fnr_grid grid_ = *grid;
    const fnr_warp warp_ = *warp;
    const fnr_rays rays_ = *rays;
    seq::push("fnr_hash_encode_fwd", [=](const fnr_step_scalars*) {
      return fnr_hash_encode_fwd(&grid_, &warp_, &rays_, euclid_bins, S, feats, selector, jacobian, stream);
    });
  }
  FNR_CHECK_ARG(rays && euclid_bins && S > 0, "hash_encode_fwd: null rays/bins or S<=0");
  RaySource src{make_rays(rays), euclid_bins, S};
  return launch_encode(grid, warp, src, rays->n_rays * (long long)S, feats, selector, jacobian, stream);
}

extern "C" int fnr_hash_encode_lattice(const fnr_grid* grid, const fnr_warp* warp, const fnr_lattice* lat,
                                       int64_t ray_begin, int64_t n_rays, float* feats, uint8_t* selector,
                                       void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_hash_encode_lattice");
  FNR_CHECK_ARG(lat && lat->xs && lat->ys && lat->zs, "hash_encode_lattice: null lattice");
  FNR_CHECK_ARG(ray_begin >= 0 && n_rays >= 0 && ray_begin + n_rays <= (int64_t)lat->n_x * lat->n_y,
                "hash_encode_lattice: ray range [%lld,+%lld) outside %d x %d lattice", (long long)ray_begin,
                (long long)n_rays, lat->n_x, lat->n_y);
  LatticeSource src{lat->xs, lat->ys, lat->zs, lat->n_y, lat->n_z, ray_begin};
  return launch_encode(grid, warp, src, n_rays * (long long)lat->n_z, feats, selector, nullptr, stream);
}

extern "C" int fnr_prop_density_fwd(const fnr_prop_net* net, const fnr_warp* warp, const fnr_rays* rays,
                                    const float* euclid_bins, int S, float* density, float* feat_save, void* stream) {
  if (seq::recording() && net && warp && rays) {
    const fnr_prop_net net_ = *net;
    const fnr_warp warp_ = *warp;
    const fnr_rays rays_ = *rays;
    seq::push("fnr_prop_density_fwd", [=](const fnr_step_scalars*) {
      return fnr_prop_density_fwd(&net_, &warp_, &rays_, euclid_bins, S, density, feat_save, stream);
    });
  }
  FNR_CHECK_ARG(net && warp && rays && euclid_bins && density && S > 0, "prop_density_fwd: null argument");
  FNR_UNSUPPORTED(net->hidden_dim == 16, "prop_density_fwd: hidden_dim %d not built (16 only)", net->hidden_dim);
  const long long N = rays->n_rays * (long long)S;
  if (N == 0) return FNR_OK;
  RaySource src{make_rays(rays), euclid_bins, S};
  const unsigned nblk = (unsigned)((N + 255) / 256);
  GridDev g = make_grid(&net->grid);
  Warp w = make_warp(warp);
  float2* fs = reinterpret_cast<float2*>(feat_save);
  FNR_PROF(OP_PROP_FWD, N);
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
