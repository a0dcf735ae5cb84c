// train.hip — losses, their gradients, compositing backward and the fused Adam step for gfx950.
//   get_loss_dict (fruit_nerf.py:359-372): MSELoss, BCEWithLogitsLoss, interlevel_loss
//   get_metrics_dict (fruit_nerf.py:396-401): distortion_loss (metric only)
//   backward of RaySamples.get_weights + renderers (fruit_nerf.py:325-348)
//   optimiser: torch.optim.Adam semantics (fruit_nerf_config.py:47-56)
// One wave per ray for the per-ray scans; everything here is bandwidth-trivial next to the field kernels.
#include "common.hpp"
#include "sequencer.hpp"

namespace fnr {

// ---------------------------------------------------------------------------------------------------
// rgb MSE + semantic BCE-with-logits: values and unit gradients
// The per-ray gradients of the two image losses, unit upstream: d MSELoss(mean over 3 R values) / d rgb_c and
// d (w * BCEWithLogitsLoss(mean over R)) / d logit.  ONE definition for every kernel that forms them — losses_block,
// k_train_losses and the composite backward that takes the targets itself (k_weights_bwd<.., TARGETS>): the bit-identity of
// those paths (tests/test_gpu_training_parity.py::test_composite_bwd_targets_is_losses_then_composite_bwd) rests on it.
__device__ __forceinline__ float mse_grad(float diff, float inv3r) { return 2.0f * diff * inv3r; }
__device__ __forceinline__ float bce_logit_grad(float x, float y, float sem_weight, float invr) {
  const float sg = 1.0f / (1.0f + expf(-x));
  return sem_weight * (sg - y) * invr;
}

// ---------------------------------------------------------------------------------------------------
// One workgroup: a training batch is a few thousand rays, and finishing inside the kernel (block reduction, PSNR)
// saves the memset, the atomics and three follow-up elementwise launches per step.
template <int THREADS>
__device__ __forceinline__ void losses_block(long long R, const float* __restrict__ rgb,
                                             const float* __restrict__ image, const float* __restrict__ sem,
                                             const float* __restrict__ mask, float sem_weight,
                                             float* __restrict__ losses, float* __restrict__ d_rgb,
                                             float* __restrict__ d_sem) {
  __shared__ float red[2][16];
  float l_rgb = 0.0f, l_sem = 0.0f;
  const float inv3r = 1.0f / (float)(3 * R), invr = 1.0f / (float)R;
  for (long long r = threadIdx.x; r < R; r += THREADS) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float d = rgb[3 * r + c] - image[3 * r + c];
      l_rgb += d * d;
      d_rgb[3 * r + c] = mse_grad(d, inv3r);
    }
    const float x = sem[r], y = mask[r];
    // BCEWithLogits: max(x,0) - x*y + log1p(exp(-|x|))
    l_sem += fmaxf(x, 0.0f) - x * y + log1pf(expf(-fabsf(x)));
    d_sem[r] = bce_logit_grad(x, y, sem_weight, invr);
  }
  l_rgb = wave_sum(l_rgb);
  l_sem = wave_sum(l_sem);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = l_rgb;
    red[1][threadIdx.x >> 6] = l_sem;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.0f, b = 0.0f;
    for (int w = 0; w < THREADS / 64; ++w) {
      a += red[0][w];
      b += red[1][w];
    }
    const float mse = a * inv3r;
    losses[0] = mse;
    losses[1] = sem_weight * b * invr;
    losses[2] = -10.0f * log10f(mse);  // PeakSignalNoiseRatio(data_range=1.0) of the same batch (fruit_nerf.py:398)
  }
}
__global__ __launch_bounds__(1024) void k_losses(long long R, const float* __restrict__ rgb,
                                                 const float* __restrict__ image, const float* __restrict__ sem,
                                                 const float* __restrict__ mask, float sem_weight,
                                                 float* __restrict__ losses, float* __restrict__ d_rgb,
                                                 float* __restrict__ d_sem) {
  losses_block<1024>(R, rgb, image, sem, mask, sem_weight, losses, d_rgb, d_sem);
}

// ---------------------------------------------------------------------------------------------------
// interlevel loss of one proposal level against the final level (nerfstudio losses.interlevel_loss):
// value (accumulated) + unit gradient w.r.t. the proposal weights.
// ---------------------------------------------------------------------------------------------------
constexpr int IL_MAX_P = 512;
constexpr int WB_MAXE = 8;   // samples per lane in the per-ray weight kernels (S <= 512)

// k_weights_bwd<false> for one ray with the upstream gradient d(loss)/d(w_k) already in registers (gw[e] belongs to
// sample lane * E + e, zero beyond S): same operations in the same order -> bit-identical d_density.
__device__ __forceinline__ void weights_bwd_ray_from_regs(long long r, int S, int lane, const float* __restrict__ euclid,
                                                          const float* __restrict__ density,
                                                          const float* __restrict__ weights,
                                                          const float (&gw)[WB_MAXE], float* __restrict__ d_density) {
  const int E = (S + 63) >> 6;
  const float* eb = euclid + r * (S + 1);
  const float* dn = density + r * S;
  const float* w = weights + r * S;
  float delta[WB_MAXE], wk[WB_MAXE];
  float dd_local = 0.0f, gww_local = 0.0f;
#pragma unroll
  for (int e = 0; e < WB_MAXE; ++e) {
    const int k = lane * E + e;
    delta[e] = 0.0f;
    wk[e] = 0.0f;
    if (e < E && k < S) {
      delta[e] = eb[k + 1] - eb[k];
      wk[e] = w[k];
      dd_local += delta[e] * dn[k];
      gww_local += gw[e] * wk[e];
    }
  }
  float cum_dd = wave_excl_scan(dd_local, lane);
  float suffix = wave_bcast_lane(wave_excl_scan(wave_bcast_lane(gww_local, 63 - lane), lane), 63 - lane);
  float suf[WB_MAXE];
#pragma unroll
  for (int e = WB_MAXE - 1; e >= 0; --e) {
    suf[e] = suffix;
    suffix += gw[e] * wk[e];
  }
#pragma unroll
  for (int e = 0; e < WB_MAXE; ++e) {
    const int k = lane * E + e;
    if (e < E && k < S) {
      cum_dd += delta[e] * dn[k];
      const float T_next = expf(-cum_dd);
      d_density[r * S + k] = delta[e] * (gw[e] * T_next - suf[e]);
    }
  }
}
constexpr int IL_LDS_FLOATS = 4 * (3 * IL_MAX_P + 4);   // per wave: cp [P+1], cy [P+1], dd [P+2]

__device__ __forceinline__ int searchsorted_right(const float* a, int n, float v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] <= v) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// A wave's contribution to a loss accumulator row (FNR_LOSS_SLOTS floats spread over 32 cache lines: same-line atomics
// serialise).  FIXED (fnr_train_losses): the row is read as 512 64-bit words and the value is added as two's-complement
// fixed point (2^-34 resolution) — integer adds commute, so the logged losses and metrics are bit-identical run to run like
// the gradients.  A contribution that is not finite or above its row's bound raises `flag` instead (the result is then
// NaN): per-ray contributions (interlevel, distortion: already divided by R) |v| <= 2^12, up to 2^17 of them — more than
// any built method launches per row (fruit_nerf_huge: 16 384 rays x 2 proposal levels); the rgb / semantic rows' raw
// per-wave sums |v| <= 2^28 / waves.  At a scale of 2^34 neither a slot nor the total can reach 2^63 and wrap
// (at 2^44 a total of ~5e5, e.g. a diverging semantic BCE sum over a large batch, wrapped silently).
constexpr double TL_FIX_SCALE = 17179869184.0;  // 2^34
constexpr float TL_FIX_MAX = 4096.0f;
template <bool FIXED>
__device__ __forceinline__ void slot_add(float* __restrict__ row, int block, int wave, float v, unsigned* flag = nullptr,
                                         float vmax = TL_FIX_MAX) {
  if constexpr (FIXED) {
    if (!(fabsf(v) <= vmax)) {  // inf / nan / absurd
      if (flag) atomicOr(flag, 1u);
      return;
    }
    const long long q = __double2ll_rn((double)v * TL_FIX_SCALE);
    atomicAdd(reinterpret_cast<unsigned long long*>(row) + (block & 31) * 16 + wave, (unsigned long long)q);
  } else {
    atomicAdd(&row[(block & 31) * 32 + wave], v);
  }
}

template <bool FIXED = false>
__device__ __forceinline__ void interlevel_block(long long R, int S_f, const float* __restrict__ spacing_f,
                                                 const float* __restrict__ w_f, int S_p,
                                                 const float* __restrict__ spacing_p, const float* __restrict__ w_p,
                                                 float mult, float* __restrict__ loss, float* __restrict__ d_wp,
                                                 int block, float* lds, int p_cap,  // lds: 4 * (3 * p_cap + 4) floats
                                                 const float* __restrict__ euclid_p = nullptr,
                                                 const float* __restrict__ density_p = nullptr,
                                                 float* __restrict__ d_density_p = nullptr, unsigned* flag = nullptr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long r = (long long)block * 4 + wave;
  if (r >= R) return;
  float* cp = lds + wave * (3 * p_cap + 4);   // p_cap >= S_p: the launch sizes the buffer for its largest level
  float* cy = cp + p_cap + 1;
  float* dd = cy + p_cap + 1;
  const float* sp = spacing_p + r * (S_p + 1);
  const float* wp = w_p + r * S_p;
  // cy1 = [0, cumsum(wp)] : lane-chunked scan
  const int E = (S_p + 63) >> 6;
  float local = 0.0f;
  for (int e = 0; e < E; ++e) {
    const int k = lane * E + e;
    if (k < S_p) local += wp[k];
  }
  float run = wave_excl_scan(local, lane);
  if (lane == 0) cy[0] = 0.0f;
  for (int e = 0; e < E; ++e) {
    const int k = lane * E + e;
    if (k < S_p) {
      run += wp[k];
      cy[k + 1] = run;
    }
  }
  for (int k = lane; k <= S_p; k += 64) cp[k] = sp[k];
  for (int k = lane; k <= S_p + 1; k += 64) dd[k] = 0.0f;
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();

  const float* c = spacing_f + r * (S_f + 1);
  const float* w = w_f + r * S_f;
  const float scale = mult / (float)(R * (long long)S_f);
  float lsum = 0.0f;
  for (int i = lane; i < S_f; i += 64) {
    int lo = searchsorted_right(cp, S_p, c[i]) - 1;          // t1_starts = cp[:-1]
    lo = min(max(lo, 0), S_p - 1);
    int hi = searchsorted_right(cp + 1, S_p, c[i + 1]);      // t1_ends = cp[1:]
    hi = min(max(hi, 0), S_p - 1);
    const float w_outer = cy[hi + 1] - cy[lo];
    const float wi = w[i];
    const float diff = fmaxf(wi - w_outer, 0.0f);
    const float den = wi + 1.0e-7f;
    lsum += diff * diff / den;
    const float g = -2.0f * diff / den * scale;  // d loss / d w_outer
    if (g != 0.0f) {
      atomicAdd(&dd[lo], g);
      atomicAdd(&dd[hi + 1], -g);
    }
  }
  lsum = wave_sum(lsum);
  // atomics to one L2 line serialise at ~12 ns each (4096 rays = ~35-50 us): 32 lines, 4 words (waves) per line
  if (lane == 0) slot_add<FIXED>(loss, block, wave, lsum * scale, flag);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  // d_wp = inclusive scan of the difference array
  float l2 = 0.0f;
  for (int e = 0; e < E; ++e) {
    const int k = lane * E + e;
    if (k < S_p) l2 += dd[k];
  }
  float run2 = wave_excl_scan(l2, lane);
  if (!d_density_p) {
    for (int e = 0; e < E; ++e) {
      const int k = lane * E + e;
      if (k < S_p) {
        run2 += dd[k];
        d_wp[r * S_p + k] = run2;
      }
    }
    return;
  }
  // the proposal level's weights backward right here (the ray is in this wave's hands; d_wp stays in registers)
  float gw[WB_MAXE];
#pragma unroll
  for (int e = 0; e < WB_MAXE; ++e) {
    const int k = lane * E + e;
    gw[e] = 0.0f;
    if (e < E && k < S_p) {
      run2 += dd[k];
      gw[e] = run2;
      if (d_wp) d_wp[r * S_p + k] = run2;
    }
  }
  weights_bwd_ray_from_regs(r, S_p, lane, euclid_p, density_p, w_p, gw, d_density_p);
}
__global__ __launch_bounds__(256) void k_interlevel(long long R, int S_f, const float* __restrict__ spacing_f,
                                                    const float* __restrict__ w_f, int S_p,
                                                    const float* __restrict__ spacing_p,
                                                    const float* __restrict__ w_p, float mult,
                                                    float* __restrict__ loss, float* __restrict__ d_wp) {
  __shared__ float lds[IL_LDS_FLOATS];
  interlevel_block(R, S_f, spacing_f, w_f, S_p, spacing_p, w_p, mult, loss, d_wp, blockIdx.x, lds, IL_MAX_P);
}

// distortion_loss (metric, fruit_nerf.py:400): mean over rays of sum_ij w_i w_j |m_i - m_j| + sum_i w_i^2 ds_i / 3
template <bool FIXED = false>
__device__ __forceinline__ void distortion_block(long long R, int S, const float* __restrict__ spacing,
                                                 const float* __restrict__ weights, float* __restrict__ out,
                                                 int block, float* lds, int s_cap, unsigned* flag = nullptr) {  // lds: 8 * s_cap floats, s_cap >= S
  float* s_m_all = lds;
  float* s_w_all = lds + 4 * s_cap;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long r = (long long)block * 4 + wave;
  if (r >= R) return;
  float* s_m = s_m_all + wave * s_cap;
  float* s_w = s_w_all + wave * s_cap;
  const float* t = spacing + r * (S + 1);
  const float* w = weights + r * S;
  for (int k = lane; k < S; k += 64) {
    s_m[k] = (t[k + 1] + t[k]) / 2.0f;
    s_w[k] = w[k];
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  float acc = 0.0f;
  for (int i = lane; i < S; i += 64) {
    const float mi = s_m[i], wi = s_w[i];
    float inner = 0.0f;
    for (int j = 0; j < S; ++j) inner += s_w[j] * fabsf(mi - s_m[j]);
    acc += wi * inner + wi * wi * (t[i + 1] - t[i]) / 3.0f;
  }
  acc = wave_sum(acc);
  if (lane == 0) slot_add<FIXED>(out, block, wave, acc / (float)R, flag);
}
__global__ __launch_bounds__(256) void k_distortion(long long R, int S, const float* __restrict__ spacing,
                                                    const float* __restrict__ weights, float* __restrict__ out) {
  __shared__ float lds[4096];
  distortion_block(R, S, spacing, weights, out, blockIdx.x, lds, 512);
}

// Every loss of a training step in ONE launch (fnr_train_losses; five dependent 4-10 us launches before): workgroups
// [0, nbl) = rgb + semantic losses and their gradients (256 rays each), then (R + 3) / 4 workgroups per proposal level
// = interlevel loss, then as many for the distortion metric.  Sums go to accumulator slots in 32 different 128-byte
// lines (same-line atomics serialise at ~12 ns each); the last workgroup to finish — found with a two-level
// completion count for the same reason — adds the slots up and writes the five scalars.
struct LevelLossArgs {
  int n_levels;
  int S_p[FNR_MAX_PROPOSAL_LEVELS];
  const float* spacing_p[FNR_MAX_PROPOSAL_LEVELS];
  const float* w_p[FNR_MAX_PROPOSAL_LEVELS];
  float* d_wp[FNR_MAX_PROPOSAL_LEVELS];
  const float* euclid_p[FNR_MAX_PROPOSAL_LEVELS];   // with d_density_p: the level's weights backward is fused in
  const float* density_p[FNR_MAX_PROPOSAL_LEVELS];
  float* d_density_p[FNR_MAX_PROPOSAL_LEVELS];
};
constexpr int TL_ROWS = 4;                                       // slot rows: interlevel, distortion, rgb, semantic
constexpr int TL_ACCUM_FLOATS = TL_ROWS * FNR_LOSS_SLOTS + 33 * 32;  // + 32 group counters and the top one, a line each
static_assert(TL_ACCUM_FLOATS == FNR_TRAIN_LOSSES_ACCUM_FLOATS, "fruitnerf_hip.h: FNR_TRAIN_LOSSES_ACCUM_FLOATS");
__global__ __launch_bounds__(256) void k_train_losses(long long R, const float* __restrict__ rgb,
                                                      const float* __restrict__ image, const float* __restrict__ sem,
                                                      const float* __restrict__ mask, float sem_weight,
                                                      float* __restrict__ d_rgb, float* __restrict__ d_sem, int S_f,
                                                      const float* __restrict__ spacing_f,
                                                      const float* __restrict__ w_f, LevelLossArgs lv, float mult,
                                                      int want_distortion, float* __restrict__ accum,
                                                      float* __restrict__ losses, int p_cap) {
  extern __shared__ float lds[];   // max(4 * (3 * p_cap + 4), 8 * S_f) floats: sized by the launch, not for 512 samples
  __shared__ bool s_last;
  __shared__ long long s_red[TL_ROWS][4];
  float* il_slots = accum;
  float* di_slots = accum + FNR_LOSS_SLOTS;
  float* rgb_slots = accum + 2 * FNR_LOSS_SLOTS;
  float* sem_slots = accum + 3 * FNR_LOSS_SLOTS;
  unsigned* cnt = reinterpret_cast<unsigned*>(accum + TL_ROWS * FNR_LOSS_SLOTS);
  unsigned* flag = cnt + 32 * 32 + 1;  // a non-finite contribution was seen (second word of the top counter's line)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nbl = (int)((R + 255) / 256);
  const int per = (int)((R + 3) / 4);
  const float inv3r = 1.0f / (float)(3 * R), invr = 1.0f / (float)R;
  if ((int)blockIdx.x < nbl) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    float l_rgb = 0.0f, l_sem = 0.0f;
    if (r < R) {  // per ray exactly what losses_block does
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float d = rgb[3 * r + c] - image[3 * r + c];
        l_rgb += d * d;
        if (d_rgb) d_rgb[3 * r + c] = mse_grad(d, inv3r);     // (NULL: the composite backward forms them itself)
      }
      const float x = sem[r], y = mask[r];
      l_sem = fmaxf(x, 0.0f) - x * y + log1pf(expf(-fabsf(x)));
      if (d_sem) d_sem[r] = bce_logit_grad(x, y, sem_weight, invr);
    }
    l_rgb = wave_sum(l_rgb);
    l_sem = wave_sum(l_sem);
    if (lane == 0) {
      // these two rows take raw sums over a wave's 64 rays (not divided by R yet), 4 nbl of them: the bound that keeps
      // their total below 2^28 (x 2^34 < 2^63) is 2^28 / waves — 4e6 per wave at 4096 rays
      const float wave_cap = 268435456.0f / (float)(4 * nbl);
      slot_add<true>(rgb_slots, blockIdx.x, wave, l_rgb, flag, wave_cap);
      slot_add<true>(sem_slots, blockIdx.x, wave, l_sem, flag, wave_cap);
    }
  } else {
    const int b = (int)blockIdx.x - nbl;
    const int role = b / per, local = b - role * per;
    if (role < lv.n_levels)
      interlevel_block<true>(R, S_f, spacing_f, w_f, lv.S_p[role], lv.spacing_p[role], lv.w_p[role], mult, il_slots,
                             lv.d_wp[role], local, lds, p_cap, lv.euclid_p[role], lv.density_p[role],
                             lv.d_density_p[role], flag);
    else
      distortion_block<true>(R, S_f, spacing_f, w_f, di_slots, local, lds, S_f, flag);
  }
  // Completion count WITHOUT __threadfence(): an agent-scope release fence writes the XCD's L2 back (this kernel's
  // outputs are dirty there) and 3000 workgroups doing that cost 100 us.  The slot sums are agent-scope atomics, performed
  // at the memory side; every wave waits until its own have been acknowledged (vmcnt(0)) before the workgroup's barrier,
  // then one thread counts the workgroup in, and the last workgroup reads the slots back with agent-scope loads.
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned g = blockIdx.x & 31u;
    const unsigned gsize = (gridDim.x - g + 31u) / 32u, ngroups = gridDim.x < 32u ? gridDim.x : 32u;
    bool last = false;
    if (atomicAdd(&cnt[g * 32], 1u) == gsize - 1u) last = atomicAdd(&cnt[32 * 32], 1u) == ngroups - 1u;
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  // the rows hold 64-bit fixed point (slot_add<true>): integer sums, any order
  long long acc[TL_ROWS] = {0, 0, 0, 0};
  for (int i = threadIdx.x; i < FNR_LOSS_SLOTS / 2; i += 256)
#pragma unroll
    for (int q = 0; q < TL_ROWS; ++q)
      acc[q] += (long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(accum + q * FNR_LOSS_SLOTS) + i,
                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int q = 0; q < TL_ROWS; ++q) {
#pragma unroll
    for (int dsh = 32; dsh >= 1; dsh >>= 1) acc[q] += __shfl_xor(acc[q], dsh, 64);
    if (lane == 0) s_red[q][wave] = acc[q];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const bool bad = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    float t[TL_ROWS];
#pragma unroll
    for (int q = 0; q < TL_ROWS; ++q) {
      const long long tot = (s_red[q][0] + s_red[q][1]) + (s_red[q][2] + s_red[q][3]);
      t[q] = bad ? __builtin_nanf("") : (float)((double)tot / TL_FIX_SCALE);
    }
    const float mse = t[2] * inv3r;
    losses[0] = mse;
    losses[1] = sem_weight * t[3] * invr;
    losses[2] = -10.0f * log10f(mse);  // PeakSignalNoiseRatio(data_range=1.0) of the same batch (fruit_nerf.py:398)
    losses[3] = t[0];
    losses[4] = want_distortion ? t[1] : 0.0f;
  }
  // self-cleaning: this (last) workgroup has read every slot and every other workgroup is done, so the accumulator goes
  // back to zero here — a caller that keeps the buffer needs no fill launch before the next step
  __syncthreads();
  for (int i = threadIdx.x; i < TL_ACCUM_FLOATS; i += 256) accum[i] = 0.0f;
}

// ---------------------------------------------------------------------------------------------------
// backward of w_i = (1 - exp(-d_i s_i)) exp(-sum_{j<i} d_j s_j):
//   dL/ds_k = d_k [ g_k T_{k+1} - sum_{i>k} g_i w_i ],  T_{k+1} = exp(-sum_{j<=k} d_j s_j)
// ---------------------------------------------------------------------------------------------------
// TARGETS (fnr_composite_bwd_targets): the per-ray loss gradients are not read from memory but formed here from the
// composited outputs and the batch — g_rgb = 2 (rgb - image) / (3 R), g_sem = w_sem (sigmoid(sem) - mask) / R: the very
// expressions of k_train_losses, so the same bits — which makes this launch independent of the losses launch: a training
// step runs the MLP backward straight behind the forward while the loss VALUES (and the interlevel loss with the proposal
// levels' weights backward) are summed on the second stream.
struct LossTargets {
  const float* out_rgb;    // [R,3] composited colour
  const float* image;      // [R,3]
  const float* out_sem;    // [R]   composited logit
  const float* mask;       // [R]
  float sem_weight;
};
template <bool COMPOSITE, bool TARGETS = false>
__global__ __launch_bounds__(256) void k_weights_bwd(long long R, int S, const float* __restrict__ euclid,
                                                     const float* __restrict__ density,
                                                     const float* __restrict__ weights,
                                                     const float* __restrict__ d_w_in,    // !COMPOSITE: [R,S]
                                                     const float* __restrict__ upstream,  // optional device scalar
                                                     const float* __restrict__ rgb,       // COMPOSITE: samples [N,3]
                                                     const float* __restrict__ g_rgb,     // COMPOSITE: [R,3]
                                                     const float* __restrict__ g_sem,     // COMPOSITE: [R]
                                                     float* __restrict__ d_density, float* __restrict__ d_rgb,
                                                     float* __restrict__ d_logit, LossTargets tg = LossTargets{}) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long r = (long long)blockIdx.x * 4 + wave;
  if (r >= R) return;
  const int E = (S + 63) >> 6;
  const float* eb = euclid + r * (S + 1);
  const float* dn = density + r * S;
  const float* w = weights + r * S;
  const float up = upstream ? upstream[0] : 1.0f;
  float gr = 0.0f, gg = 0.0f, gb = 0.0f, gs = 0.0f, l0 = 0.0f, l1 = 0.0f, l2 = 0.0f, bgw = 0.0f;
  if (COMPOSITE) {
    if constexpr (TARGETS) {
      const float inv3r = 1.0f / (float)(3 * R), invr = 1.0f / (float)R;   // (k_train_losses, rgb / semantic role)
      const float d0 = tg.out_rgb[3 * r] - tg.image[3 * r], d1 = tg.out_rgb[3 * r + 1] - tg.image[3 * r + 1],
                  d2 = tg.out_rgb[3 * r + 2] - tg.image[3 * r + 2];
      gr = mse_grad(d0, inv3r);
      gg = mse_grad(d1, inv3r);
      gb = mse_grad(d2, inv3r);
      gs = bce_logit_grad(tg.out_sem[r], tg.mask[r], tg.sem_weight, invr);
    } else {
      gr = g_rgb[3 * r];
      gg = g_rgb[3 * r + 1];
      gb = g_rgb[3 * r + 2];
      gs = g_sem[r];
    }
    const float* cl = rgb + (r * S + (S - 1)) * 3;
    l0 = cl[0];
    l1 = cl[1];
    l2 = cl[2];
  }
  float delta[WB_MAXE], gw[WB_MAXE], wk[WB_MAXE];
  float dd_local = 0.0f, gww_local = 0.0f, wsum_local = 0.0f;
#pragma unroll
  for (int e = 0; e < WB_MAXE; ++e) {
    const int k = lane * E + e;
    delta[e] = 0.0f;
    gw[e] = 0.0f;
    wk[e] = 0.0f;
    if (e < E && k < S) {
      delta[e] = eb[k + 1] - eb[k];
      wk[e] = w[k];
      if (COMPOSITE) {
        const float* c = rgb + (r * S + k) * 3;
        gw[e] = gr * (c[0] - l0) + gg * (c[1] - l1) + gb * (c[2] - l2);  // semantic weights are detached
      } else {
        gw[e] = d_w_in[r * S + k] * up;
      }
      dd_local += delta[e] * dn[k];
      gww_local += gw[e] * wk[e];
      wsum_local += wk[e];
    }
  }
  float cum_dd = wave_excl_scan(dd_local, lane);     // sum_{j<first k of lane}
  // sum_{i>k} g_i w_i as a true SUFFIX sum (reverse scan), not `total - prefix`: behind a surface the suffix is a sum of
  // tiny terms while total and prefix agree to 7 digits, and dL/dsigma there is multiplied by sigma (up to e^15) in the
  // trunc_exp backward — the cancellation error was comparable to the real gradient (autograd's cumsum backward is a
  // reverse cumsum, i.e. exact in this sense).
  float suffix = wave_bcast_lane(wave_excl_scan(wave_bcast_lane(gww_local, 63 - lane), lane), 63 - lane);
  if (COMPOSITE) bgw = 1.0f - wave_sum(wsum_local);
  float suf[WB_MAXE];   // suffix AFTER element e of this lane's chunk
#pragma unroll
  for (int e = WB_MAXE - 1; e >= 0; --e) {
    suf[e] = suffix;
    suffix += gw[e] * wk[e];   // gw / wk are 0 for e >= E or k >= S
  }
#pragma unroll
  for (int e = 0; e < WB_MAXE; ++e) {
    const int k = lane * E + e;
    if (e < E && k < S) {
      cum_dd += delta[e] * dn[k];   // inclusive of k
      const float T_next = expf(-cum_dd);
      d_density[r * S + k] = delta[e] * (gw[e] * T_next - suf[e]);
      if (COMPOSITE) {
        float f = wk[e];
        if (k == S - 1) f += bgw;  // background = last sample's colour (RGBRenderer "last_sample")
        float* o = d_rgb + (r * S + k) * 3;
        o[0] = gr * f;
        o[1] = gg * f;
        o[2] = gb * f;
        d_logit[r * S + k] = gs * wk[e];
      }
    }
  }
}

// k_composite_fwd (render.hip, training mode) followed by k_weights_bwd<true, true> for the same ray, in one wave: everything the
// backward read back from memory (weights, composited colour and logit) stays in registers.
__global__ __launch_bounds__(256) void k_composite_fwd_bwd(RaysDev rays, int S, const float* __restrict__ euclid,
                                                           const float* __restrict__ density, const float* __restrict__ rgb,
                                                           const float* __restrict__ logit, const float* __restrict__ image,
                                                           const float* __restrict__ mask, float sem_weight,
                                                           float* __restrict__ weights, float* __restrict__ out_rgb,
                                                           float* __restrict__ out_acc, float* __restrict__ out_depth,
                                                           float* __restrict__ out_sem, long long* __restrict__ out_label,
                                                           float* __restrict__ d_density, float* __restrict__ d_rgb,
                                                           float* __restrict__ d_logit) {
  static_assert(WB_MAXE == 8, "the forward and the backward chunk a ray the same way");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long R = rays.n_rays;
  const long long r = (long long)blockIdx.x * 4 + wave;
  if (r >= R) return;
  const int E = (S + 63) >> 6;
  const float* eb = euclid + r * (S + 1);
  const float* dn = density + r * S;
  const float* cs = rgb + r * S * 3;
  const float* lg = logit + r * S;
  // ---- forward: k_composite_fwd with training = 1 -----------------------------------------------------------------------
  float dd[WB_MAXE];
  float local = 0.0f;
#pragma unroll
  for (int e = 0; e < WB_MAXE; ++e) {
    const int k = lane * E + e;
    dd[e] = 0.0f;
    if (e < E && k < S) dd[e] = fmul(fsub(eb[k + 1], eb[k]), dn[k]);
    local += dd[e];
  }
  float excl = wave_excl_scan(local, lane);
  float w[WB_MAXE];
  float acc_l = 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f, sm = 0.0f;
#pragma unroll
  for (int e = 0; e < WB_MAXE; ++e) {
    const int k = lane * E + e;
    const float T = expf(-excl);
    const float alpha = 1.0f - expf(-dd[e]);
    w[e] = nan_to_num(alpha * T);
    excl += dd[e];
    if (e < E && k < S) {
      weights[r * S + k] = w[e];
      const float c0 = cs[3 * k], c1 = cs[3 * k + 1], c2 = cs[3 * k + 2];
      acc_l += w[e];
      cr = fmaf(w[e], c0, cr);
      cg = fmaf(w[e], c1, cg);
      cb = fmaf(w[e], c2, cb);
      sm = fmaf(w[e], lg[k], sm);
    } else {
      w[e] = 0.0f;
    }
  }
  float cw = wave_excl_scan(acc_l, lane);
  int first = 0x7fffffff;
#pragma unroll
  for (int e = 0; e < WB_MAXE; ++e) {
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
  // (every lane: the backward wants the composited colour; wave_sum leaves the same total in all of them)
  const float l0 = cs[3 * (S - 1)], l1 = cs[3 * (S - 1) + 1], l2 = cs[3 * (S - 1) + 2];
  const float o0 = cr + l0 * (1.0f - acc), o1 = cg + l1 * (1.0f - acc), o2 = cb + l2 * (1.0f - acc);
  if (lane == 0) {
    out_rgb[3 * r] = o0;
    out_rgb[3 * r + 1] = o1;
    out_rgb[3 * r + 2] = o2;
    out_acc[r] = acc;
    out_sem[r] = sm;
    if (out_label) out_label[r] = (fsub(1.0f / (1.0f + expf(-sm)), 0.9f) > 0.0f) ? 1 : 0;
    out_depth[r] = fdiv(fadd(eb[first], eb[first + 1]), 2.0f);
  }
  // ---- backward: k_weights_bwd<true, true> --------------------------------------------------------------------------------
  const float inv3r = 1.0f / (float)(3 * R), invr = 1.0f / (float)R;
  const float gr = mse_grad(o0 - image[3 * r], inv3r), gg = mse_grad(o1 - image[3 * r + 1], inv3r),
              gb = mse_grad(o2 - image[3 * r + 2], inv3r);
  const float gs = bce_logit_grad(sm, mask[r], sem_weight, invr);
  float delta[WB_MAXE], gw[WB_MAXE], wk[WB_MAXE];
  float dd_local = 0.0f, gww_local = 0.0f, wsum_local = 0.0f;
#pragma unroll
  for (int e = 0; e < WB_MAXE; ++e) {
    const int k = lane * E + e;
    delta[e] = 0.0f;
    gw[e] = 0.0f;
    wk[e] = 0.0f;
    if (e < E && k < S) {
      delta[e] = eb[k + 1] - eb[k];
      wk[e] = w[e];
      const float* c = cs + 3 * k;
      gw[e] = gr * (c[0] - l0) + gg * (c[1] - l1) + gb * (c[2] - l2);  // semantic weights are detached
      dd_local += delta[e] * dn[k];
      gww_local += gw[e] * wk[e];
      wsum_local += wk[e];
    }
  }
  float cum_dd = wave_excl_scan(dd_local, lane);
  float suffix = wave_bcast_lane(wave_excl_scan(wave_bcast_lane(gww_local, 63 - lane), lane), 63 - lane);
  const float bgw = 1.0f - wave_sum(wsum_local);
  float suf[WB_MAXE];
#pragma unroll
  for (int e = WB_MAXE - 1; e >= 0; --e) {
    suf[e] = suffix;
    suffix += gw[e] * wk[e];
  }
#pragma unroll
  for (int e = 0; e < WB_MAXE; ++e) {
    const int k = lane * E + e;
    if (e < E && k < S) {
      cum_dd += delta[e] * dn[k];
      const float T_next = expf(-cum_dd);
      d_density[r * S + k] = delta[e] * (gw[e] * T_next - suf[e]);
      float f = wk[e];
      if (k == S - 1) f += bgw;
      float* o = d_rgb + (r * S + k) * 3;
      o[0] = gr * f;
      o[1] = gg * f;
      o[2] = gb * f;
      d_logit[r * S + k] = gs * wk[e];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam incl. its L2 weight_decay, no amsgrad), whole arena in one launch; optionally zeroes grads
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_adam(float4* __restrict__ p, float4* __restrict__ g, float4* __restrict__ m,
                                              float4* __restrict__ v, long long n4, float lr, float b1, float b2,
                                              float eps, float bc1, float bc2_sqrt, float grad_scale, float weight_decay,
                                              int zero_grad) {
  const float step_size = lr / bc1;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    // moments and gradient: read once, written once per step — streaming, so that the sweep (the exchange path's 88 us over the
    // main table) leaves the table's parameters in the caches for the next encode (round 5's hints, now on the unfused sweep too)
    float4 P = p[i], G = ntc_load<NT_MOMENT_LD>(&g[i]), M = ntc_load<NT_MOMENT_LD>(&m[i]), V = ntc_load<NT_MOMENT_LD>(&v[i]);
    float* pp = reinterpret_cast<float*>(&P);
    float* gp = reinterpret_cast<float*>(&G);
    float* mp = reinterpret_cast<float*>(&M);
    float* vp = reinterpret_cast<float*>(&V);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float gr = gp[c] * grad_scale;
      if (weight_decay != 0.0f) gr = gr + weight_decay * pp[c];   // grad.add(param, alpha=weight_decay)
      mp[c] = mp[c] + (gr - mp[c]) * (1.0f - b1);           // exp_avg.lerp_(grad, 1 - beta1)
      vp[c] = vp[c] * b2 + (1.0f - b2) * gr * gr;           // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
      const float denom = sqrtf(vp[c]) / bc2_sqrt + eps;    // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
      pp[c] = pp[c] - step_size * (mp[c] / denom);          // param.addcdiv_(exp_avg, denom, value=-step_size)
    }
    p[i] = P;
    ntc_store<NT_MOMENT_ST>(&m[i], M);
    ntc_store<NT_MOMENT_ST>(&v[i], V);
    if (zero_grad) ntc_store<NT_MOMENT_ST>(&g[i], make_float4(0.f, 0.f, 0.f, 0.f));
  }
}

// torch.optim.RAdam (fruit_nerf_big / fruit_nerf_huge use it for every group, fruit_nerf_config.py:97-106,148-160):
// same moments as Adam; the update is rectified once the variance estimate is tractable (rho_t > 5), plain momentum
// before.  `rect` < 0 encodes "not rectified"; all step-dependent scalars are computed on the host in double.
__global__ __launch_bounds__(256) void k_radam(float4* __restrict__ p, float4* __restrict__ g, float4* __restrict__ m,
                                               float4* __restrict__ v, long long n4, float lr, float b1, float b2,
                                               float eps, float bc1, float bc2_sqrt, float rect, float grad_scale,
                                               float weight_decay, int zero_grad) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    // moments and gradient: read once, written once per step — streaming, so that the sweep (the exchange path's 88 us over the
    // main table) leaves the table's parameters in the caches for the next encode (round 5's hints, now on the unfused sweep too)
    float4 P = p[i], G = ntc_load<NT_MOMENT_LD>(&g[i]), M = ntc_load<NT_MOMENT_LD>(&m[i]), V = ntc_load<NT_MOMENT_LD>(&v[i]);
    float* pp = reinterpret_cast<float*>(&P);
    float* gp = reinterpret_cast<float*>(&G);
    float* mp = reinterpret_cast<float*>(&M);
    float* vp = reinterpret_cast<float*>(&V);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float gr = gp[c] * grad_scale;
      if (weight_decay != 0.0f) gr = gr + weight_decay * pp[c];
      mp[c] = mp[c] + (gr - mp[c]) * (1.0f - b1);
      vp[c] = vp[c] * b2 + (1.0f - b2) * gr * gr;
      const float mhat = mp[c] / bc1;                        // bias_corrected_exp_avg
      if (rect >= 0.0f) {
        const float adaptive = bc2_sqrt / (sqrtf(vp[c]) + eps);   // sqrt(bias_correction2) / (exp_avg_sq.sqrt() + eps)
        pp[c] = pp[c] - lr * (mhat * rect * adaptive);
      } else {
        pp[c] = pp[c] - lr * mhat;
      }
    }
    p[i] = P;
    ntc_store<NT_MOMENT_ST>(&m[i], M);
    ntc_store<NT_MOMENT_ST>(&v[i], V);
    if (zero_grad) ntc_store<NT_MOMENT_ST>(&g[i], make_float4(0.f, 0.f, 0.f, 0.f));
  }
}

// Adam / RAdam over several spans of one arena in one launch (fnr_adam_step_spans): the spans' float4 chunks are
// numbered consecutively, a thread finds its span by scanning the (<= 8) cumulative counts.  Per element the same
// operations in the same order as k_adam / k_radam.
struct AdamSpansDev {
  int n;
  long long cum4[FNR_MAX_ADAM_SPANS + 1];  // chunk index at which span k starts; cum4[n] = total
  long long off4[FNR_MAX_ADAM_SPANS];      // first float4 of span k in the arena
  float lr[FNR_MAX_ADAM_SPANS], bc1[FNR_MAX_ADAM_SPANS], bc2_sqrt[FNR_MAX_ADAM_SPANS], rect[FNR_MAX_ADAM_SPANS];
};
template <bool RADAM>
__global__ __launch_bounds__(256) void k_adam_spans(float4* __restrict__ p, float4* __restrict__ g,
                                                    float4* __restrict__ m, float4* __restrict__ v, AdamSpansDev sp,
                                                    float b1, float b2, float eps, float grad_scale,
                                                    float weight_decay, int zero_grad) {
  const long long total = sp.cum4[sp.n];
  for (long long c = (long long)blockIdx.x * 256 + threadIdx.x; c < total; c += (long long)gridDim.x * 256) {
    int k = 0;
#pragma unroll
    for (int q = 1; q < FNR_MAX_ADAM_SPANS; ++q)
      if (q < sp.n && c >= sp.cum4[q]) k = q;
    const long long i = sp.off4[k] + (c - sp.cum4[k]);
    const float lr = sp.lr[k], bc1 = sp.bc1[k], bc2_sqrt = sp.bc2_sqrt[k], rect = sp.rect[k];
    const float step_size = lr / bc1;
    // moments and gradient: read once, written once per step — streaming, so that the sweep (the exchange path's 88 us over the
    // main table) leaves the table's parameters in the caches for the next encode (round 5's hints, now on the unfused sweep too)
    float4 P = p[i], G = ntc_load<NT_MOMENT_LD>(&g[i]), M = ntc_load<NT_MOMENT_LD>(&m[i]), V = ntc_load<NT_MOMENT_LD>(&v[i]);
    float* pp = reinterpret_cast<float*>(&P);
    float* gp = reinterpret_cast<float*>(&G);
    float* mp = reinterpret_cast<float*>(&M);
    float* vp = reinterpret_cast<float*>(&V);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float gr = gp[e] * grad_scale;
      if (weight_decay != 0.0f) gr = gr + weight_decay * pp[e];
      mp[e] = mp[e] + (gr - mp[e]) * (1.0f - b1);
      vp[e] = vp[e] * b2 + (1.0f - b2) * gr * gr;
      if (!RADAM) {
        const float denom = sqrtf(vp[e]) / bc2_sqrt + eps;
        pp[e] = pp[e] - step_size * (mp[e] / denom);
      } else {
        const float mhat = mp[e] / bc1;
        if (rect >= 0.0f) {
          const float adaptive = bc2_sqrt / (sqrtf(vp[e]) + eps);
          pp[e] = pp[e] - lr * (mhat * rect * adaptive);
        } else {
          pp[e] = pp[e] - lr * mhat;
        }
      }
    }
    p[i] = P;
    ntc_store<NT_MOMENT_ST>(&m[i], M);
    ntc_store<NT_MOMENT_ST>(&v[i], V);
    if (zero_grad) ntc_store<NT_MOMENT_ST>(&g[i], make_float4(0.f, 0.f, 0.f, 0.f));
  }
}

}  // namespace fnr

using namespace fnr;

extern "C" int fnr_losses_fwd(int64_t n_rays, const float* rgb, const float* image, const float* semantics,
                              const float* fruit_mask, float semantic_loss_weight, float* losses, float* d_rgb,
                              float* d_semantics, void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_losses_fwd");
  FNR_CHECK_ARG(rgb && image && semantics && fruit_mask && losses && d_rgb && d_semantics && n_rays > 0,
                "losses_fwd: null argument");
  FNR_PROF(OP_LOSSES, n_rays);
  hipLaunchKernelGGL(k_losses, dim3(1), dim3(1024), 0, as_stream(stream), (long long)n_rays, rgb, image,
                     semantics, fruit_mask, semantic_loss_weight, losses, d_rgb, d_semantics);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

extern "C" int fnr_interlevel_fwd(int64_t n_rays, int S_f, const float* spacing_f, const float* weights_f, int S_p,
                                  const float* spacing_p, const float* weights_p, float mult, float* loss,
                                  float* d_weights_p, void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_interlevel_fwd");
  FNR_CHECK_ARG(spacing_f && weights_f && spacing_p && weights_p && loss && d_weights_p, "interlevel_fwd: null argument");
  FNR_CHECK_ARG(S_f > 0 && S_p > 0 && S_p <= IL_MAX_P, "interlevel_fwd: S_p %d out of range", S_p);
  if (n_rays == 0) return FNR_OK;
  FNR_PROF(OP_INTERLEVEL, n_rays * (long long)S_p);
  hipLaunchKernelGGL(k_interlevel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, as_stream(stream),
                     (long long)n_rays, S_f, spacing_f, weights_f, S_p, spacing_p, weights_p, mult, loss, d_weights_p);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

extern "C" int fnr_distortion(int64_t n_rays, int S, const float* spacing, const float* weights, float* out,
                              void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_distortion");
  FNR_CHECK_ARG(spacing && weights && out && S > 0 && S <= 512, "distortion: bad argument");
  if (n_rays == 0) return FNR_OK;
  FNR_PROF(OP_DISTORTION, n_rays * (long long)S);
  hipLaunchKernelGGL(k_distortion, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, as_stream(stream),
                     (long long)n_rays, S, spacing, weights, out);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

extern "C" int fnr_composite_bwd(const fnr_rays* rays, int S, const float* euclid_bins, const float* density,
                                 const float* rgb, const float* weights, const float* g_rgb, const float* g_semantics,
                                 float* d_density, float* d_rgb, float* d_logit, void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_composite_bwd");
  FNR_CHECK_ARG(rays && euclid_bins && density && rgb && weights && g_rgb && g_semantics && d_density && d_rgb &&
                    d_logit,
                "composite_bwd: null argument");
  FNR_CHECK_ARG(S > 0 && S <= 64 * WB_MAXE, "composite_bwd: S %d out of range", S);
  if (rays->n_rays == 0) return FNR_OK;
  FNR_PROF(OP_COMPOSITE_BWD, rays->n_rays * (long long)S);
  hipLaunchKernelGGL((k_weights_bwd<true>), dim3((unsigned)((rays->n_rays + 3) / 4)), dim3(256), 0, as_stream(stream),
                     (long long)rays->n_rays, S, euclid_bins, density, weights, (const float*)nullptr,
                     (const float*)nullptr, rgb, g_rgb, g_semantics, d_density, d_rgb, d_logit);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

extern "C" int fnr_composite_bwd_targets(const fnr_rays* rays, int S, const float* euclid_bins, const float* density,
                                         const float* rgb, const float* weights, const float* out_rgb, const float* image,
                                         const float* out_semantics, const float* mask, float semantic_loss_weight,
                                         float* d_density, float* d_rgb, float* d_logit, void* stream) {
  if (seq::recording() && rays) {
    const fnr_rays rays_ = *rays;
    seq::push("fnr_composite_bwd_targets", [=](const fnr_step_scalars*) {
      return fnr_composite_bwd_targets(&rays_, S, euclid_bins, density, rgb, weights, out_rgb, image, out_semantics, mask,
                                       semantic_loss_weight, d_density, d_rgb, d_logit, stream);
    });
  }
  FNR_CHECK_ARG(rays && euclid_bins && density && rgb && weights && out_rgb && image && out_semantics && mask && d_density &&
                    d_rgb && d_logit,
                "composite_bwd_targets: null argument");
  FNR_CHECK_ARG(S > 0 && S <= 64 * WB_MAXE, "composite_bwd_targets: S %d out of range", S);
  if (rays->n_rays == 0) return FNR_OK;
  FNR_PROF(OP_COMPOSITE_BWD, rays->n_rays * (long long)S);
  const LossTargets tg{out_rgb, image, out_semantics, mask, semantic_loss_weight};
  hipLaunchKernelGGL((k_weights_bwd<true, true>), dim3((unsigned)((rays->n_rays + 3) / 4)), dim3(256), 0, as_stream(stream),
                     (long long)rays->n_rays, S, euclid_bins, density, weights, (const float*)nullptr,
                     (const float*)nullptr, rgb, (const float*)nullptr, (const float*)nullptr, d_density, d_rgb, d_logit, tg);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

// fnr_composite_fwd (training) and fnr_composite_bwd_targets as ONE launch (ABI 13): a wave composites its ray and, with the
// outputs still in registers, forms the per-ray loss gradients and runs the backward — the same operations on the same values in
// the same order as the two kernels (tests/test_gpu_training_parity.py: bit-identical), one launch ramp less on the chain
// forward -> MLP backward of every training step.
extern "C" int fnr_composite_fwd_bwd_targets(const fnr_rays* rays, int S, const float* euclid_bins, const float* density,
                                             const float* rgb, const float* logit, const float* image, const float* mask,
                                             float semantic_loss_weight, float* weights, float* out_rgb,
                                             float* out_accumulation, float* out_depth, float* out_semantics,
                                             int64_t* out_label, float* d_density, float* d_rgb, float* d_logit,
                                             void* stream) {
  if (seq::recording() && rays) {
    const fnr_rays rays_ = *rays;
    seq::push("fnr_composite_fwd_bwd_targets", [=](const fnr_step_scalars*) {
      return fnr_composite_fwd_bwd_targets(&rays_, S, euclid_bins, density, rgb, logit, image, mask, semantic_loss_weight, weights,
                                           out_rgb, out_accumulation, out_depth, out_semantics, out_label, d_density, d_rgb,
                                           d_logit, stream);
    });
  }
  FNR_CHECK_ARG(rays && euclid_bins && density && rgb && logit && image && mask && weights && out_rgb && out_accumulation &&
                    out_depth && out_semantics && d_density && d_rgb && d_logit,
                "composite_fwd_bwd_targets: null argument");
  FNR_CHECK_ARG(S > 0 && S <= 64 * WB_MAXE, "composite_fwd_bwd_targets: S %d out of range", S);
  if (rays->n_rays == 0) return FNR_OK;
  FNR_PROF(OP_COMPOSITE_FWD, rays->n_rays * (long long)S);
  hipLaunchKernelGGL(k_composite_fwd_bwd, dim3((unsigned)((rays->n_rays + 3) / 4)), dim3(256), 0, as_stream(stream),
                     make_rays(rays), S, euclid_bins, density, rgb, logit, image, mask, semantic_loss_weight, weights, out_rgb,
                     out_accumulation, out_depth, out_semantics, reinterpret_cast<long long*>(out_label), d_density, d_rgb,
                     d_logit);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

extern "C" int fnr_weights_bwd(int64_t n_rays, int S, const float* euclid_bins, const float* density,
                               const float* weights, const float* d_weights, const float* upstream,
                               float* d_density, void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_weights_bwd");
  FNR_CHECK_ARG(euclid_bins && density && weights && d_weights && d_density, "weights_bwd: null argument");
  FNR_CHECK_ARG(S > 0 && S <= 64 * WB_MAXE, "weights_bwd: S %d out of range", S);
  if (n_rays == 0) return FNR_OK;
  FNR_PROF(OP_WEIGHTS_BWD, n_rays * (long long)S);
  hipLaunchKernelGGL((k_weights_bwd<false>), dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, as_stream(stream),
                     (long long)n_rays, S, euclid_bins, density, weights, d_weights, upstream, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, d_density, (float*)nullptr, (float*)nullptr);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

extern "C" int fnr_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                             float beta1, float beta2, float eps, int64_t step, float grad_scale, float weight_decay,
                             int zero_grad, void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_adam_step");
  FNR_CHECK_ARG(params && grads && exp_avg && exp_avg_sq, "adam_step: null argument");
  FNR_CHECK_ARG(n % 4 == 0 && step >= 1, "adam_step: n must be a multiple of 4 (arena is padded) and step >= 1");
  if (n == 0) return FNR_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  long long n4 = n / 4;
  long long blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  FNR_PROF(OP_ADAM, n);
  hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<float4*>(params), reinterpret_cast<float4*>(grads),
                     reinterpret_cast<float4*>(exp_avg), reinterpret_cast<float4*>(exp_avg_sq), n4, lr, beta1, beta2,
                     eps, (float)bc1, (float)sqrt(bc2), grad_scale, weight_decay, zero_grad);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

extern "C" int fnr_radam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                              float beta1, float beta2, float eps, int64_t step, float grad_scale, float weight_decay,
                              int zero_grad, void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_radam_step");
  FNR_CHECK_ARG(params && grads && exp_avg && exp_avg_sq, "radam_step: null argument");
  FNR_CHECK_ARG(n % 4 == 0 && step >= 1, "radam_step: n must be a multiple of 4 (arena is padded) and step >= 1");
  if (n == 0) return FNR_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double b2t = pow((double)beta2, (double)step);
  const double bc2 = 1.0 - b2t;
  const double rho_inf = 2.0 / (1.0 - (double)beta2) - 1.0;
  const double rho_t = rho_inf - 2.0 * (double)step * b2t / bc2;
  double rect = -1.0;  // not rectified
  if (rho_t > 5.0)
    rect = sqrt((rho_t - 4.0) * (rho_t - 2.0) * rho_inf / ((rho_inf - 4.0) * (rho_inf - 2.0) * rho_t));
  long long n4 = n / 4;
  long long blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  FNR_PROF(OP_ADAM, n);
  hipLaunchKernelGGL(k_radam, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), reinterpret_cast<float4*>(params),
                     reinterpret_cast<float4*>(grads), reinterpret_cast<float4*>(exp_avg),
                     reinterpret_cast<float4*>(exp_avg_sq), n4, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2),
                     (float)rect, grad_scale, weight_decay, zero_grad);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

extern "C" int fnr_adam_step_spans(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int n_spans,
                                   const fnr_adam_span* spans, int algorithm, float beta1, float beta2, float eps,
                                   float grad_scale, float weight_decay, int zero_grad, void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_adam_step_spans");
  FNR_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && spans, "adam_step_spans: null argument");
  FNR_CHECK_ARG(n_spans >= 1 && n_spans <= FNR_MAX_ADAM_SPANS, "adam_step_spans: %d spans (1..%d)", n_spans,
                FNR_MAX_ADAM_SPANS);
  FNR_CHECK_ARG(algorithm == 0 || algorithm == 1, "adam_step_spans: algorithm %d (0 = Adam, 1 = RAdam)", algorithm);
  AdamSpansDev sp;
  sp.n = n_spans;
  sp.cum4[0] = 0;
  long long total_floats = 0;
  for (int k = 0; k < FNR_MAX_ADAM_SPANS; ++k) {
    if (k >= n_spans) {
      sp.cum4[k + 1] = sp.cum4[k];
      sp.off4[k] = 0;
      sp.lr[k] = 0.0f, sp.bc1[k] = 1.0f, sp.bc2_sqrt[k] = 1.0f, sp.rect[k] = -1.0f;
      continue;
    }
    const fnr_adam_span& s = spans[k];
    FNR_CHECK_ARG(s.offset >= 0 && s.count >= 0 && s.offset % 4 == 0 && s.count % 4 == 0 && s.step >= 1,
                  "adam_step_spans: span %d: offset / count must be multiples of 4, step >= 1", k);
    // the step-dependent scalars exactly as fnr_adam_step / fnr_radam_step compute them (double)
    const double bc1 = 1.0 - pow((double)beta1, (double)s.step);
    const double b2t = pow((double)beta2, (double)s.step);
    const double bc2 = 1.0 - b2t;
    double rect = -1.0;
    if (algorithm == 1) {
      const double rho_inf = 2.0 / (1.0 - (double)beta2) - 1.0;
      const double rho_t = rho_inf - 2.0 * (double)s.step * b2t / bc2;
      if (rho_t > 5.0)
        rect = sqrt((rho_t - 4.0) * (rho_t - 2.0) * rho_inf / ((rho_inf - 4.0) * (rho_inf - 2.0) * rho_t));
    }
    sp.off4[k] = s.offset / 4;
    sp.cum4[k + 1] = sp.cum4[k] + s.count / 4;
    sp.lr[k] = s.lr, sp.bc1[k] = (float)bc1, sp.bc2_sqrt[k] = (float)sqrt(bc2), sp.rect[k] = (float)rect;
    total_floats += s.count;
  }
  const long long total4 = sp.cum4[n_spans];
  if (total4 == 0) return FNR_OK;
  long long blocks = (total4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  FNR_PROF(OP_ADAM, total_floats);
  float4 *p4 = reinterpret_cast<float4*>(params), *g4 = reinterpret_cast<float4*>(grads);
  float4 *m4 = reinterpret_cast<float4*>(exp_avg), *v4 = reinterpret_cast<float4*>(exp_avg_sq);
  if (algorithm == 0)
    hipLaunchKernelGGL(k_adam_spans<false>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), p4, g4, m4, v4, sp,
                       beta1, beta2, eps, grad_scale, weight_decay, zero_grad);
  else
    hipLaunchKernelGGL(k_adam_spans<true>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), p4, g4, m4, v4, sp,
                       beta1, beta2, eps, grad_scale, weight_decay, zero_grad);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

extern "C" int fnr_train_losses(int64_t n_rays, const float* rgb, const float* image, const float* semantics,
                                const float* fruit_mask, float semantic_loss_weight, float* d_rgb, float* d_semantics,
                                int S_f, const float* spacing_f, const float* weights_f, int n_levels, const int* S_p,
                                const float* const* spacing_p, const float* const* weights_p, float* const* d_weights_p,
                                const float* const* euclid_p, const float* const* density_p,
                                float* const* d_density_p, float interlevel_mult, int want_distortion, float* accum,
                                float* losses, void* stream) {
  if (seq::recording() && n_levels >= 0 && n_levels <= FNR_MAX_PROPOSAL_LEVELS) {
    constexpr int M = FNR_MAX_PROPOSAL_LEVELS;
    const auto S_ = seq::copy_n<int, M>(S_p, n_levels);
    const auto sp_ = seq::copy_n<const float*, M>(spacing_p, n_levels);
    const auto wp_ = seq::copy_n<const float*, M>(weights_p, n_levels);
    const auto dw_ = seq::copy_n<float*, M>(d_weights_p, n_levels);
    const auto eu_ = seq::copy_n<const float*, M>(euclid_p, n_levels);
    const auto dn_ = seq::copy_n<const float*, M>(density_p, n_levels);
    const auto dd_ = seq::copy_n<float*, M>(d_density_p, n_levels);
    const bool has_sp = spacing_p, has_wp = weights_p, has_dw = d_weights_p, has_eu = euclid_p, has_dn = density_p,
               has_dd = d_density_p;
    seq::push("fnr_train_losses", [=](const fnr_step_scalars* sc) {
      return fnr_train_losses(n_rays, rgb, image, semantics, fruit_mask, semantic_loss_weight, d_rgb, d_semantics, S_f,
                              spacing_f, weights_f, n_levels, S_.data(), has_sp ? sp_.data() : nullptr,
                              has_wp ? wp_.data() : nullptr, has_dw ? dw_.data() : nullptr, has_eu ? eu_.data() : nullptr,
                              has_dn ? dn_.data() : nullptr, has_dd ? dd_.data() : nullptr, interlevel_mult, want_distortion,
                              accum, (sc && sc->losses) ? sc->losses : losses, stream);
    });
  }
  FNR_CHECK_ARG(rgb && image && semantics && fruit_mask && spacing_f && weights_f && accum && losses && n_rays > 0,
                "train_losses: null argument");
  FNR_CHECK_ARG((d_rgb == nullptr) == (d_semantics == nullptr), "train_losses: d_rgb and d_semantics go together");
  FNR_CHECK_ARG(n_levels >= 0 && n_levels <= FNR_MAX_PROPOSAL_LEVELS, "train_losses: %d proposal levels (0..%d)",
                n_levels, FNR_MAX_PROPOSAL_LEVELS);
  FNR_CHECK_ARG(S_f > 0 && (!want_distortion || S_f <= 512), "train_losses: S_f %d out of range", S_f);
  LevelLossArgs lv{};
  lv.n_levels = n_levels;
  for (int l = 0; l < n_levels; ++l) {
    FNR_CHECK_ARG(S_p && spacing_p && weights_p && spacing_p[l] && weights_p[l], "train_losses: null proposal level %d", l);
    FNR_CHECK_ARG(S_p[l] > 0 && S_p[l] <= IL_MAX_P, "train_losses: S_p %d out of range", S_p[l]);
    lv.S_p[l] = S_p[l], lv.spacing_p[l] = spacing_p[l], lv.w_p[l] = weights_p[l];
    lv.d_wp[l] = d_weights_p ? d_weights_p[l] : nullptr;
    lv.d_density_p[l] = d_density_p ? d_density_p[l] : nullptr;
    if (lv.d_density_p[l]) {
      FNR_CHECK_ARG(euclid_p && density_p && euclid_p[l] && density_p[l],
                    "train_losses: level %d: d_density_p needs euclid_p and density_p", l);
      lv.euclid_p[l] = euclid_p[l], lv.density_p[l] = density_p[l];
    } else {
      FNR_CHECK_ARG(lv.d_wp[l], "train_losses: level %d: neither d_weights_p nor d_density_p", l);
    }
  }
  const long long per = (n_rays + 3) / 4;
  const long long blocks = (n_rays + 255) / 256 + per * (n_levels + (want_distortion ? 1 : 0));
  FNR_CHECK_ARG(blocks < (1ll << 31), "train_losses: too many rays");
  FNR_PROF(OP_LOSSES, n_rays);
  int p_cap = 1;
  for (int l = 0; l < n_levels; ++l) p_cap = S_p[l] > p_cap ? S_p[l] : p_cap;
  size_t lds_floats = 4 * (size_t)(3 * p_cap + 4);
  if (want_distortion && (size_t)8 * S_f > lds_floats) lds_floats = (size_t)8 * S_f;
  hipLaunchKernelGGL(k_train_losses, dim3((unsigned)blocks), dim3(256), lds_floats * sizeof(float), as_stream(stream),
                     (long long)n_rays, rgb, image, semantics, fruit_mask, semantic_loss_weight, d_rgb, d_semantics, S_f,
                     spacing_f, weights_f, lv, interlevel_mult, want_distortion, accum, losses, p_cap);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}
