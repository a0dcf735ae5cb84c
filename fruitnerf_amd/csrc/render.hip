// render.hip — alpha compositing for gfx950: RaySamples.get_weights + RGBRenderer("last_sample") +
// AccumulationRenderer + DepthRenderer("median") + SemanticRenderer (fruit_nerf.py:325-348), one wave per ray.
#include "common.hpp"
#include "sequencer.hpp"

namespace fnr {

constexpr int CMP_MAXE = 8;  // S <= 512

__global__ __launch_bounds__(256) void k_composite_fwd(RaysDev rays, int S, const float* __restrict__ euclid,
                                                       const float* __restrict__ density,
                                                       const float* __restrict__ rgb, const float* __restrict__ logit,
                                                       int training, float* __restrict__ weights,
                                                       float* __restrict__ out_rgb, float* __restrict__ out_acc,
                                                       float* __restrict__ out_depth, float* __restrict__ out_sem,
                                                       long long* __restrict__ out_label) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long r = (long long)blockIdx.x * 4 + wave;
  if (r >= rays.n_rays) return;
  const int E = (S + 63) >> 6;
  const float* eb = euclid + r * (S + 1);
  const float* dn = density + r * S;
  const float* cs = rgb + r * S * 3;
  const float* lg = logit + r * S;

  float dd[CMP_MAXE];
  float local = 0.0f;
#pragma unroll
  for (int e = 0; e < CMP_MAXE; ++e) {
    const int k = lane * E + e;
    dd[e] = 0.0f;
    if (e < E && k < S) dd[e] = fmul(fsub(eb[k + 1], eb[k]), dn[k]);
    local += dd[e];
  }
  float excl = wave_excl_scan(local, lane);
  float w[CMP_MAXE];
  float acc_l = 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f, sm = 0.0f;
#pragma unroll
  for (int e = 0; e < CMP_MAXE; ++e) {
    const int k = lane * E + e;
    const float T = expf(-excl);
    const float alpha = 1.0f - expf(-dd[e]);
    w[e] = nan_to_num(alpha * T);
    excl += dd[e];
    if (e < E && k < S) {
      weights[r * S + k] = w[e];
      float c0 = cs[3 * k], c1 = cs[3 * k + 1], c2 = cs[3 * k + 2];
      if (!training) {
        c0 = nan_to_num(c0);
        c1 = nan_to_num(c1);
        c2 = nan_to_num(c2);
      }
      acc_l += w[e];
      cr = fmaf(w[e], c0, cr);
      cg = fmaf(w[e], c1, cg);
      cb = fmaf(w[e], c2, cb);
      sm = fmaf(w[e], lg[k], sm);
    } else {
      w[e] = 0.0f;
    }
  }
  // median depth before the reductions destroy the per-lane partials
  float cw = wave_excl_scan(acc_l, lane);
  int first = 0x7fffffff;
#pragma unroll
  for (int e = 0; e < CMP_MAXE; ++e) {
    const int k = lane * E + e;
    cw += w[e];
    if (e < E && k < S && cw >= 0.5f && first == 0x7fffffff) first = k;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) first = min(first, __shfl_xor(first, d, 64));
  if (first > S - 1) first = S - 1;

  const float acc = wave_sum(acc_l);
  cr = wave_sum(cr);
  cg = wave_sum(cg);
  cb = wave_sum(cb);
  sm = wave_sum(sm);
  if (lane == 0) {
    // background_color = "last_sample": rgb[..., -1, :] * (1 - accumulated_weight)
    float l0 = cs[3 * (S - 1)], l1 = cs[3 * (S - 1) + 1], l2 = cs[3 * (S - 1) + 2];
    if (!training) {
      l0 = nan_to_num(l0);
      l1 = nan_to_num(l1);
      l2 = nan_to_num(l2);
    }
    const float bgw = 1.0f - acc;
    float o0 = cr + l0 * bgw, o1 = cg + l1 * bgw, o2 = cb + l2 * bgw;
    if (!training) {
      o0 = fminf(fmaxf(o0, 0.0f), 1.0f);
      o1 = fminf(fmaxf(o1, 0.0f), 1.0f);
      o2 = fminf(fmaxf(o2, 0.0f), 1.0f);
    }
    out_rgb[3 * r] = o0;
    out_rgb[3 * r + 1] = o1;
    out_rgb[3 * r + 2] = o2;
    out_acc[r] = acc;
    out_sem[r] = sm;
    // heaviside(sigmoid(semantics) - 0.9, 0) (fruit_nerf.py:309-311, 351-353)
    if (out_label) out_label[r] = (fsub(1.0f / (1.0f + expf(-sm)), 0.9f) > 0.0f) ? 1 : 0;
    out_depth[r] = fdiv(fadd(eb[first], eb[first + 1]), 2.0f);
  }
}

}  // namespace fnr

using namespace fnr;

extern "C" int fnr_composite_fwd(const fnr_rays* rays, int S, const float* euclid_bins, const float* density,
                                 const float* rgb, const float* logit, int training, float* weights, float* out_rgb,
                                 float* out_accumulation, float* out_depth, float* out_semantics,
                                 int64_t* out_label, void* stream) {
  if (seq::recording() && rays) {
    const fnr_rays rays_ = *rays;
    seq::push("fnr_composite_fwd", [=](const fnr_step_scalars*) {
      return fnr_composite_fwd(&rays_, S, euclid_bins, density, rgb, logit, training, weights, out_rgb, out_accumulation,
                               out_depth, out_semantics, out_label, stream);
    });
  }
  FNR_CHECK_ARG(rays && euclid_bins && density && rgb && logit && weights && out_rgb && out_accumulation &&
                    out_depth && out_semantics,
                "composite_fwd: null argument");
  FNR_CHECK_ARG(S > 0 && S <= 64 * CMP_MAXE, "composite_fwd: S %d out of range", S);
  if (rays->n_rays == 0) return FNR_OK;
  FNR_PROF(OP_COMPOSITE_FWD, rays->n_rays * (long long)S);
  hipLaunchKernelGGL(k_composite_fwd, dim3((unsigned)((rays->n_rays + 3) / 4)), dim3(256), 0, as_stream(stream),
                     make_rays(rays), S, euclid_bins, density, rgb, logit, training, weights, out_rgb,
                     out_accumulation, out_depth, out_semantics, reinterpret_cast<long long*>(out_label));
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}
