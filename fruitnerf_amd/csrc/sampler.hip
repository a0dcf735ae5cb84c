// sampler.hip — hierarchical ray sampling for gfx950: SpacedSampler bins, RaySamples.get_weights and the
// inverse-CDF PDFSampler (nerfstudio 0.3.2 semantics, driven from fruit_nerf.py:151-158,318).
// One 64-lane wave owns one ray: per-ray scans are wave scans, the CDF lives in LDS, searchsorted is a
// per-lane binary search.  These kernels are latency-trivial next to the field queries (HBM-bound reads
// of S floats per ray); they exist so the whole sampling chain stays on the device without launches of
// dozens of small elementwise ops.
#include "sampler_math.hpp"
#include "sequencer.hpp"

namespace fnr {

__global__ __launch_bounds__(256) void k_sample_spaced(RaysDev rays, int kind, int S,
                                                       const float* __restrict__ base_bins,
                                                       const float* __restrict__ t_rand, int t_rand_per_bin,
                                                       float* __restrict__ spacing, float* __restrict__ euclid) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = rays.n_rays * (long long)(S + 1);
  if (idx >= total) return;
  const long long r = idx / (S + 1);
  const int j = (int)(idx - r * (S + 1));
  // single_jitter: one number per ray; otherwise one per bin edge, t_rand [R, S+1] (ray_samplers.py:79-83)
  const float b = spaced_bin_edge(base_bins, S, j, t_rand != nullptr, t_rand ? t_rand[t_rand_per_bin ? idx : r] : 0.0f);
  const float s_near = spacing_fn(kind, rays.nears[r]), s_far = spacing_fn(kind, rays.fars[r]);
  spacing[idx] = b;
  euclid[idx] = spacing_to_euclid(kind, b, s_near, s_far);
}

constexpr int PDF_MAXE = 8;        // elements per lane -> S_prev <= 512
constexpr int PDF_MAX_PREV = 512;  // LDS cdf capacity per wave

__global__ __launch_bounds__(256) void k_weights_pdf(RaysDev rays, int kind, int S_prev, int S_new,
                                                     const float* __restrict__ density,
                                                     const float* __restrict__ spacing_prev,
                                                     const float* __restrict__ euclid_prev, float anneal,
                                                     const float* __restrict__ u_base, const float* __restrict__ rand,
                                                     float* __restrict__ weights, float* __restrict__ median_depth,
                                                     float* __restrict__ spacing_new, float* __restrict__ euclid_new) {
  __shared__ float s_cdf[4][PDF_MAX_PREV + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long r = (long long)blockIdx.x * 4 + wave;
  if (r >= rays.n_rays) return;  // whole wave exits together
  float* cdf = s_cdf[wave];
  const int E = (S_prev + 63) >> 6;
  const float* eb = euclid_prev + r * (S_prev + 1);
  const float* dn = density + r * S_prev;

  // ---- RaySamples.get_weights ----------------------------------------------------------------
  float dd[PDF_MAXE], w[PDF_MAXE];
  float local = 0.0f;
#pragma unroll
  for (int e = 0; e < PDF_MAXE; ++e) {
    const int k = lane * E + e;
    dd[e] = 0.0f;
    if (e < E && k < S_prev) dd[e] = fmul(fsub(eb[k + 1], eb[k]), dn[k]);
    local += dd[e];
  }
  float excl = wave_excl_scan(local, lane);  // sum of delta*sigma before this lane's chunk
  float wsum_local = 0.0f;
#pragma unroll
  for (int e = 0; e < PDF_MAXE; ++e) {
    const int k = lane * E + e;
    const float T = expf(-excl);
    const float alpha = 1.0f - expf(-dd[e]);
    w[e] = nan_to_num(alpha * T);
    excl += dd[e];
    if (e < E && k < S_prev) {
      weights[r * S_prev + k] = w[e];
      wsum_local += w[e];
    } else {
      w[e] = 0.0f;
    }
  }

  // ---- DepthRenderer("median") of this level (fruit_nerf.py:299-300) ----------------------------
  if (median_depth) {
    float cw = wave_excl_scan(wsum_local, lane);
    int first = 0x7fffffff;
#pragma unroll
    for (int e = 0; e < PDF_MAXE; ++e) {
      const int k = lane * E + e;
      cw += w[e];
      if (e < E && k < S_prev && cw >= 0.5f && first == 0x7fffffff) first = k;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) first = min(first, __shfl_xor(first, d, 64));
    if (first > S_prev - 1) first = S_prev - 1;
    if (lane == 0) median_depth[r] = fdiv(fadd(eb[first], eb[first + 1]), 2.0f);
  }
  if (S_new <= 0) return;

  // ---- PDFSampler: annealed, padded histogram -> CDF in LDS --------------------------------------
  float wa[PDF_MAXE];
  float sum_local = 0.0f;
#pragma unroll
  for (int e = 0; e < PDF_MAXE; ++e) {
    const int k = lane * E + e;
    wa[e] = 0.0f;
    if (e < E && k < S_prev) {
      const float p = (anneal == 1.0f) ? w[e] : powf(w[e], anneal);
      wa[e] = fadd(p, 0.01f);  // histogram_padding
    }
    sum_local += wa[e];
  }
  float wsum = wave_sum(sum_local);
  const float padding = fmaxf(fsub(1e-5f, wsum), 0.0f);
  const float pad_each = fdiv(padding, (float)S_prev);
  wsum = fadd(wsum, padding);
  float pdf_local = 0.0f;
#pragma unroll
  for (int e = 0; e < PDF_MAXE; ++e) {
    const int k = lane * E + e;
    if (e < E && k < S_prev) {
      wa[e] = fdiv(fadd(wa[e], pad_each), wsum);
      pdf_local += wa[e];
    }
  }
  float run = wave_excl_scan(pdf_local, lane);
  if (lane == 0) cdf[0] = 0.0f;
#pragma unroll
  for (int e = 0; e < PDF_MAXE; ++e) {
    const int k = lane * E + e;
    if (e < E && k < S_prev) {
      run += wa[e];
      cdf[k + 1] = fminf(1.0f, run);
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes done (in-order LDS queue)
  __builtin_amdgcn_wave_barrier();

  // ---- inverse-CDF resampling ----------------------------------------------------------------------
  const int nb = S_new + 1;
  const float u_shift = rand ? fdiv(rand[r], (float)nb) : (float)(1.0 / (2.0 * (double)nb));
  const float s_near = spacing_fn(kind, rays.nears[r]), s_far = spacing_fn(kind, rays.fars[r]);
  const float* ex = spacing_prev + r * (S_prev + 1);
  for (int jj = lane; jj < nb; jj += 64) {
    const float u = fadd(u_base[jj], u_shift);
    // searchsorted(cdf, u, side="right"): number of entries <= u
    int lo = 0, hi = S_prev + 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] <= u) lo = mid + 1;
      else hi = mid;
    }
    const int below = min(max(lo - 1, 0), S_prev), above = min(max(lo, 0), S_prev);
    const float c0 = cdf[below], c1 = cdf[above];
    float t = fdiv(fsub(u, c0), fsub(c1, c0));
    t = nan_to_num(t);
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    const float b0 = ex[below], b1 = ex[above];
    const float b = fadd(b0, fmul(t, fsub(b1, b0)));
    spacing_new[r * nb + jj] = b;
    euclid_new[r * nb + jj] = spacing_to_euclid(kind, b, s_near, s_far);
  }
}

}  // namespace fnr

using namespace fnr;

extern "C" int fnr_sample_spaced(const fnr_rays* rays, int spacing_kind, int S, const float* base_bins,
                                 const float* t_rand, int t_rand_per_bin, float* spacing_bins, float* euclid_bins,
                                 void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_sample_spaced");
  FNR_CHECK_ARG(rays && base_bins && spacing_bins && euclid_bins && S > 0, "sample_spaced: null argument");
  FNR_CHECK_ARG(rays->nears && rays->fars, "sample_spaced: rays.nears/fars must be set (collider, fruit_nerf.py:382)");
  FNR_CHECK_ARG(spacing_kind == 0 || spacing_kind == 1, "sample_spaced: spacing_kind %d", spacing_kind);
  const long long total = rays->n_rays * (long long)(S + 1);
  if (total == 0) return FNR_OK;
  FNR_PROF(OP_SAMPLE_SPACED, total);
  hipLaunchKernelGGL(k_sample_spaced, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                     make_rays(rays), spacing_kind, S, base_bins, t_rand, t_rand_per_bin, spacing_bins, euclid_bins);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

extern "C" int fnr_weights_pdf(const fnr_rays* rays, int spacing_kind, int S_prev, int S_new, const float* density,
                               const float* spacing_prev, const float* euclid_prev, float anneal, const float* u_base,
                               const float* rand, float* weights, float* median_depth, float* spacing_new,
                               float* euclid_new, void* stream) {
  if (seq::recording() && rays) {
    const fnr_rays rays_ = *rays;
    seq::push("fnr_weights_pdf", [=](const fnr_step_scalars* sc) {
      return fnr_weights_pdf(&rays_, spacing_kind, S_prev, S_new, density, spacing_prev, euclid_prev, sc ? sc->anneal : anneal,
                             u_base, rand, weights, median_depth, spacing_new, euclid_new, stream);
    });
  }
  FNR_CHECK_ARG(rays && density && euclid_prev && weights, "weights_pdf: null argument");
  FNR_CHECK_ARG(S_prev > 0 && S_prev <= PDF_MAX_PREV, "weights_pdf: S_prev %d out of range (1..%d)", S_prev,
                PDF_MAX_PREV);
  FNR_CHECK_ARG(S_new == 0 || (spacing_prev && u_base && spacing_new && euclid_new),
                "weights_pdf: resampling needs spacing_prev/u_base/outputs");
  FNR_CHECK_ARG(spacing_kind == 0 || spacing_kind == 1, "weights_pdf: spacing_kind %d", spacing_kind);
  if (rays->n_rays == 0) return FNR_OK;
  FNR_PROF(OP_WEIGHTS_PDF, rays->n_rays * (long long)S_prev);
  hipLaunchKernelGGL(k_weights_pdf, dim3((unsigned)((rays->n_rays + 3) / 4)), dim3(256), 0, as_stream(stream),
                     make_rays(rays), spacing_kind, S_prev, S_new, density, spacing_prev, euclid_prev, anneal, u_base,
                     rand, weights, median_depth, spacing_new, euclid_new);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}
