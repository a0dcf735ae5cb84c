// field_layers.hpp — MFMA fragment layout of FruitField's tiny MLPs (fruit_field.py:132-166) on gfx950.
//
// Formulation (per 16-sample tile, one wave): every layer is computed TRANSPOSED,
//     Y^T[out][sample] = W[out][in] * X^T[in][sample]
// with v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain):  A = W fragment, B = X^T fragment.
//   A[i][k]: lane l supplies i = l&15, k = l>>4        B[k][j]: lane l supplies k = l>>4, j = l&15
//   D[row][col]: lane l holds col = l&15, rows 4*(l>>4)+r, r = 0..3
// With samples on j = l&15 the accumulator of layer n (lane holds features 16*ob + 4g + r of sample j)
// is *directly* the B operand of layer n+1 if K-step (ib, r) is defined to cover the four features
// {16*ib + 4g + r : g = 0..3}: activations never leave registers between layers, and the sum over k
// is only re-ordered (fp32 rounding differences ~1e-7 vs the oracle).
//
// LDS image ("P"): for layer l, block (ob, ib) is 64 slots x 4 floats; lane (i = l&15, g = l>>4)
// reads slot (i ^ g) + 16 g as one ds_read_b128 = {W[16 ob + i][kmap(ib, g, r)] : r = 0..3}.
// The XOR swizzle keeps that read conflict-free AND makes the transposed-weight read of the backward
// pass (lane wants W[16 ob + 4 kg + r][16 ib + i']) conflict-free as a broadcast b128 + select.
#pragma once
#include "common.hpp"

namespace fnr {

using f32x4 = __attribute__((ext_vector_type(4))) float;

enum KMap : int { KM_LINEAR = 0, KM_HASH = 1, KM_GEO = 2, KM_COLOR = 3 };

// Two field shapes are built, selected per call from the fnr_field_net dimensions:
//   FieldCfgBase  `fruit_nerf`      (fruit_nerf_config.py:27-61): hash 32 -> 64 -> 1+15; semantic
//                 15 -> 64 -> 64 (+ head 64 -> 1); colour 16+15+32 -> 64 -> 64 -> 3.
//   FieldCfgBig   `fruit_nerf_big` / `fruit_nerf_huge` (fruit_nerf_config.py:63-160): of the widths those configs set
//                 only geo_feat_dim = 30, num_layers_semantic = 3, hidden_dim_semantics = 128 reach FruitField
//                 (fruit_nerf.py:88-103): hash 32 -> 64 -> 1+30; semantic 30 -> 128 -> 128 -> 64 (+ head);
//                 colour 16+30+32 -> 64 -> 64 -> 3.
// A Cfg names its layers (L_*), gives every layer's block counts and nn.Linear dimensions, and orders the layers so
// that the set a kernel needs is a contiguous range of the fragment image (a kernel stages only its range into LDS:
// the Big image is 176 KB, more than the 160 KB of a CU).
//   HB   = 16-wide blocks of the base MLP's output h = [density logit | geo | zero padding]
//   NSEM = layers of mlp_semantics, SEMB = 16-wide blocks of its hidden layers
struct FieldCfgBase {
  static constexpr int GEO = 15, HB = 1, NSEM = 2, SEMB = 4;
  static constexpr int NLAYERS = 8;
  enum { L_BASE0 = 0, L_BASE1 = 1, L_SEM0 = 2, L_SEM1 = 3, L_SEM2 = -1, L_HEAD = 4, L_COL0 = 5, L_COL1 = 6, L_COL2 = 7 };
  // layer ids:                 base0 base1 sem0 sem1 head col0 col1 col2
  static constexpr int nob(int l) { constexpr int a[NLAYERS] = {4, 1, 4, 4, 1, 4, 4, 1}; return a[l]; }
  static constexpr int nib(int l) { constexpr int a[NLAYERS] = {2, 4, 1, 4, 4, 4, 4, 4}; return a[l]; }
  static constexpr int out_dim(int l) { constexpr int a[NLAYERS] = {64, 16, 64, 64, 1, 64, 64, 3}; return a[l]; }
  static constexpr int in_dim(int l) { constexpr int a[NLAYERS] = {32, 64, 15, 64, 64, 63, 64, 64}; return a[l]; }
  static constexpr int km(int l) {
    constexpr int a[NLAYERS] = {KM_HASH, KM_LINEAR, KM_GEO, KM_LINEAR, KM_LINEAR, KM_COLOR, KM_LINEAR, KM_LINEAR};
    return a[l];
  }
  static constexpr int woff(int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += nob(i) * nib(i) * 256;
    return o;
  }
  static constexpr int boff(int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += nob(i) * 16;
    return o;
  }
  static constexpr int W_TOTAL = 18432;  // = woff(NLAYERS), floats
  static constexpr int B_TOTAL = 368;    // = boff(NLAYERS)
  static constexpr int LDS_FLOATS = W_TOTAL + B_TOTAL;
  // packed buffer = fragment image + the ray-constant slice of mlp_head layer 0, transposed: Wt[k][out], k < 48
  static constexpr int PACKED_FLOATS = LDS_FLOATS + 48 * 64;
};
static_assert(FieldCfgBase::woff(FieldCfgBase::NLAYERS) == FieldCfgBase::W_TOTAL, "W_TOTAL");
static_assert(FieldCfgBase::boff(FieldCfgBase::NLAYERS) == FieldCfgBase::B_TOTAL, "B_TOTAL");
static_assert(FieldCfgBase::LDS_FLOATS % 4 == 0, "the transposed slice must stay 16-byte aligned");

struct FieldCfgBig {
  static constexpr int GEO = 30, HB = 2, NSEM = 3, SEMB = 8;
  static constexpr int NLAYERS = 9;
  // image order: [base0 base1 | col0 col1 col2 | sem0 sem1 sem2 head] — one contiguous range per kernel family
  enum { L_BASE0 = 0, L_BASE1 = 1, L_COL0 = 2, L_COL1 = 3, L_COL2 = 4, L_SEM0 = 5, L_SEM1 = 6, L_SEM2 = 7, L_HEAD = 8 };
  static constexpr int nob(int l) { constexpr int a[NLAYERS] = {4, 2, 4, 4, 1, 8, 8, 4, 1}; return a[l]; }
  static constexpr int nib(int l) { constexpr int a[NLAYERS] = {2, 4, 5, 4, 4, 2, 8, 8, 4}; return a[l]; }
  static constexpr int out_dim(int l) { constexpr int a[NLAYERS] = {64, 31, 64, 64, 3, 128, 128, 64, 1}; return a[l]; }
  static constexpr int in_dim(int l) { constexpr int a[NLAYERS] = {32, 64, 78, 64, 64, 30, 128, 128, 64}; return a[l]; }
  static constexpr int km(int l) {
    constexpr int a[NLAYERS] = {KM_HASH, KM_LINEAR, KM_COLOR, KM_LINEAR, KM_LINEAR, KM_GEO, KM_LINEAR, KM_LINEAR, KM_LINEAR};
    return a[l];
  }
  static constexpr int woff(int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += nob(i) * nib(i) * 256;
    return o;
  }
  static constexpr int boff(int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += nob(i) * 16;
    return o;
  }
  static constexpr int W_TOTAL = 44032;
  static constexpr int B_TOTAL = 576;
  static constexpr int LDS_FLOATS = W_TOTAL + B_TOTAL;  // size of the image in global memory (never all in LDS)
  static constexpr int PACKED_FLOATS = LDS_FLOATS + 48 * 64;
};
static_assert(FieldCfgBig::woff(FieldCfgBig::NLAYERS) == FieldCfgBig::W_TOTAL, "W_TOTAL");
static_assert(FieldCfgBig::boff(FieldCfgBig::NLAYERS) == FieldCfgBig::B_TOTAL, "B_TOTAL");
static_assert(FieldCfgBig::LDS_FLOATS % 4 == 0, "the transposed slice must stay 16-byte aligned");
constexpr int FIELD_MAX_LAYERS = 9;
constexpr int FIELD_MAX_PACKED_FLOATS = FieldCfgBig::PACKED_FLOATS;
constexpr int FIELD_MAX_IMAGE_FLOATS = FieldCfgBig::LDS_FLOATS;
constexpr int FIELD_MAX_HB = 2;

struct FieldPtrs {
  const float* w[FIELD_MAX_LAYERS];
  const float* b[FIELD_MAX_LAYERS];
};

// Global -> LDS copy of NVEC 16-byte vectors by a workgroup of THREADS threads.  All of a thread's loads are issued
// before its first LDS store (the trip count is a compile-time constant, batches of <= 12 loads): the plain
// `for (i = tid; i < n; i += blockDim.x) dst[i] = src[i]` loop exposes one L2/HBM round trip PER ITERATION
// (~1.5 us each; measured: ~15 us of prologue for the 75 KB image = most of an MLP kernel's FIXED cost, which was a
// third of the whole MLP backward at 4096 rays).
template <int THREADS, int NVEC>
__device__ __forceinline__ void stage_copy(f32x4* __restrict__ dst, const f32x4* __restrict__ src) {
  constexpr int PER = (NVEC + THREADS - 1) / THREADS;
  constexpr int BATCH = PER < 12 ? PER : 12;
#pragma unroll
  for (int b0 = 0; b0 < PER; b0 += BATCH) {
    f32x4 v[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int i = (int)threadIdx.x + (b0 + u) * THREADS;
      if (b0 + u < PER && i < NVEC) v[u] = src[i];
    }
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int i = (int)threadIdx.x + (b0 + u) * THREADS;
      if (b0 + u < PER && i < NVEC) dst[i] = v[u];
    }
  }
}

// The part of the fragment image a kernel keeps in LDS: layers [LA, LB) — weights first, then their biases.
template <class Cfg, int LA, int LB>
struct LdsRange {
  static constexpr int W0 = Cfg::woff(LA), W1 = Cfg::woff(LB), B0 = Cfg::boff(LA), B1 = Cfg::boff(LB);
  static constexpr int W_FLOATS = W1 - W0, B_FLOATS = B1 - B0, FLOATS = W_FLOATS + B_FLOATS;
  static_assert(W_FLOATS % 4 == 0 && B_FLOATS % 4 == 0 && W0 % 4 == 0 && (Cfg::W_TOTAL + B0) % 4 == 0, "float4 copies");
  __device__ static __forceinline__ const float* w(const float* lds, int l) { return lds + (Cfg::woff(l) - W0); }
  __device__ static __forceinline__ float* w(float* lds, int l) { return lds + (Cfg::woff(l) - W0); }
  __device__ static __forceinline__ const float* b(const float* lds, int l) { return lds + W_FLOATS + (Cfg::boff(l) - B0); }
  // fragment image (global, 16-byte aligned) -> LDS; THREADS = the workgroup size
  template <int THREADS>
  __device__ static __forceinline__ void stage(float* __restrict__ lds, const float* __restrict__ packed) {
    const f32x4* sw = reinterpret_cast<const f32x4*>(packed + W0);
    const f32x4* sb = reinterpret_cast<const f32x4*>(packed + Cfg::W_TOTAL + B0);
    f32x4* dst = reinterpret_cast<f32x4*>(lds);
    stage_copy<THREADS, W_FLOATS / 4>(dst, sw);
    stage_copy<THREADS, B_FLOATS / 4>(dst + W_FLOATS / 4, sb);
  }
};

// input column of nn.Linear weight for K-slot (ib, g, r); -1 -> structural zero
template <class Cfg>
__device__ __forceinline__ int kmap(int kind, int ib, int g, int r, int in_dim) {
  const int k = 4 * g + r;
  int col;
  switch (kind) {
    case KM_HASH: {  // hash features: slot s = 4 ib + r covers level 4 (s >> 1) + g, feature s & 1
      const int s = 4 * ib + r;
      col = 2 * (4 * (s >> 1) + g) + (s & 1);
      break;
    }
    case KM_GEO:  // input blocks = base-MLP output h (h[0] = density logit is not an input)
      col = 16 * ib + k - 1;
      break;
    case KM_COLOR:  // [h (HB blocks) | SH16 | emb32] -> nn.Linear columns [SH16, geo, emb]
      if (ib < Cfg::HB) {
        const int hk = 16 * ib + k;  // h index: 0 = density logit, 1..GEO = geo features, beyond = padding
        col = (hk >= 1 && hk <= Cfg::GEO) ? 16 + (hk - 1) : -1;
      } else if (ib == Cfg::HB) {
        col = k;
      } else {
        col = 16 + Cfg::GEO + 16 * (ib - Cfg::HB - 1) + k;
      }
      break;
    default:
      col = 16 * ib + k;
  }
  return (col >= 0 && col < in_dim) ? col : -1;
}

__device__ __forceinline__ int swz_slot(int i, int g) { return (i ^ g) + 16 * g; }

// nn.Linear column of mlp_head layer 0 for ray-constant input k: k < 16 -> SH component k, else embedding k - 16
template <class Cfg>
__device__ __forceinline__ int color_const_col(int k) { return k < 16 ? k : 16 + Cfg::GEO + (k - 16); }
constexpr int COLOR_CONST_K = 48;  // 16 SH + 32 embedding

// value of element `idx` of the fragment image (weights then biases) taken from the nn.Linear tensors
template <class Cfg>
__device__ __forceinline__ float packed_value(int idx, const FieldPtrs& p) {
  if (idx < Cfg::W_TOTAL) {
    int l = 0;
#pragma unroll
    for (int q = 1; q < Cfg::NLAYERS; ++q)
      if (idx >= Cfg::woff(q)) l = q;
    const int local = idx - Cfg::woff(l);
    const int r = local & 3, slot = (local >> 2) & 63, blk = local >> 8;
    const int nib = Cfg::nib(l);
    const int ib = blk % nib, ob = blk / nib;
    const int g = slot >> 4, i = (slot & 15) ^ g;
    const int out = 16 * ob + i;
    const int col = kmap<Cfg>(Cfg::km(l), ib, g, r, Cfg::in_dim(l));
    return (out < Cfg::out_dim(l) && col >= 0) ? p.w[l][out * Cfg::in_dim(l) + col] : 0.0f;
  }
  const int bi = idx - Cfg::W_TOTAL;
  int l = 0;
#pragma unroll
  for (int q = 1; q < Cfg::NLAYERS; ++q)
    if (bi >= Cfg::boff(q)) l = q;
  const int o = bi - Cfg::boff(l);
  return (o < Cfg::out_dim(l)) ? p.b[l][o] : 0.0f;
}

// nn.Linear weights -> fragment image in global memory, once per call (weights change every optimiser step).
// Packing inside every workgroup's prologue cost ~40 us per launch (18.8k scattered 4-byte loads + index math
// per workgroup); a 74-workgroup pack launch + linear float4 copies cost ~3 us.
template <class Cfg>
__device__ __forceinline__ void pack_field_weights_block(const FieldPtrs& p, float* __restrict__ packed, int block) {
  const int idx = block * 256 + threadIdx.x;
  if (idx < Cfg::LDS_FLOATS) {
    packed[idx] = packed_value<Cfg>(idx, p);
  } else if (idx < Cfg::PACKED_FLOATS) {
    const int k = (idx - Cfg::LDS_FLOATS) >> 6, o = (idx - Cfg::LDS_FLOATS) & 63;
    packed[idx] = p.w[Cfg::L_COL0][o * Cfg::in_dim(Cfg::L_COL0) + color_const_col<Cfg>(k)];
  }
}
template <class Cfg>
__global__ __launch_bounds__(256) void k_pack_field_weights(FieldPtrs p, float* __restrict__ packed) {
  pack_field_weights_block<Cfg>(p, packed, blockIdx.x);
}
template <class Cfg>
constexpr int pack_field_weights_blocks() { return (Cfg::PACKED_FLOATS + 255) / 256; }

template <class Cfg>
static inline void launch_pack_field_weights(const FieldPtrs& p, float* packed, hipStream_t st) {
  hipLaunchKernelGGL((k_pack_field_weights<Cfg>), dim3((Cfg::PACKED_FLOATS + 255) / 256), dim3(256), 0, st, p, packed);
}

// out^T[16 NOB][16] += W[:, input blocks 0..NIB-1] * in^T ; `in`/`out` are C-layout accumulators.
// NIB_STRIDE = number of input blocks of the layer in the image (> NIB when only the leading blocks are used).
template <int NOB, int NIB, int NIB_STRIDE = NIB>
__device__ __forceinline__ void mlp_layer_acc(const float* __restrict__ P, const f32x4 (&in)[NIB], f32x4 (&out)[NOB],
                                              int lane) {
  const int g = lane >> 4, i = lane & 15;
  const int slot = swz_slot(i, g);
#pragma unroll
  for (int ib = 0; ib < NIB; ++ib) {
    f32x4 w[NOB];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
      w[ob] = *reinterpret_cast<const f32x4*>(P + ((ob * NIB_STRIDE + ib) * 64 + slot) * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob)
        out[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ob][r], in[ib][r], out[ob], 0, 0, 0);
  }
}

// one layer: out^T[16 NOB][16] = W * in^T + b
template <int NOB, int NIB>
__device__ __forceinline__ void mlp_layer(const float* __restrict__ P, const float* __restrict__ B,
                                          const f32x4 (&in)[NIB], f32x4 (&out)[NOB], int lane) {
  const int g = lane >> 4;
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob) out[ob] = *reinterpret_cast<const f32x4*>(B + 16 * ob + 4 * g);
  mlp_layer_acc<NOB, NIB, NIB>(P, in, out, lane);
}

// first colour layer (mlp_head layer 0, fruit_field.py:150-158,258-262).  48 of its inputs — SH16(direction) and
// the appearance embedding — are constant along a ray, so their product with the weights is a per-RAY vector
// (k_color_ray_bias, 64 floats per ray) that enters here as the accumulator's initial value; only the h blocks
// (input blocks 0..HB-1 of the image) are multiplied per sample: 16 HB MFMAs instead of 64 (80).
// W0 = the layer's weight block in LDS.
template <class Cfg>
__device__ __forceinline__ void color_layer0(const float* __restrict__ W0, const float* __restrict__ ray_bias,
                                             long long ray, const f32x4 (&h)[Cfg::HB], f32x4 (&c1)[4], int lane) {
  const int g = lane >> 4;
#pragma unroll
  for (int ob = 0; ob < 4; ++ob)
    c1[ob] = *reinterpret_cast<const f32x4*>(ray_bias + (size_t)ray * 64 + 16 * ob + 4 * g);
  mlp_layer_acc<4, Cfg::HB, Cfg::HB + 3>(W0, h, c1, lane);
}

template <int N>
__device__ __forceinline__ void relu_(f32x4 (&a)[N]) {
#pragma unroll
  for (int b = 0; b < N; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) a[b][r] = fmaxf(a[b][r], 0.0f);
}

// SHEncoding(levels=4) torch path on the shifted direction d' = (d+1)/2 (fruit_field.py:208-210,243-245)
__device__ __forceinline__ void sh16_all(const float* __restrict__ dir, float (&c)[16]) {
  const float x = (dir[0] + 1.0f) / 2.0f, y = (dir[1] + 1.0f) / 2.0f, z = (dir[2] + 1.0f) / 2.0f;
  const float xx = x * x, yy = y * y, zz = z * z;
  c[0] = 0.28209479177387814f;
  c[1] = 0.4886025119029199f * y;
  c[2] = 0.4886025119029199f * z;
  c[3] = 0.4886025119029199f * x;
  c[4] = 1.0925484305920792f * x * y;
  c[5] = 1.0925484305920792f * y * z;
  c[6] = 0.9461746957575601f * zz - 0.31539156525251999f;
  c[7] = 1.0925484305920792f * x * z;
  c[8] = 0.5462742152960396f * (xx - yy);
  c[9] = 0.5900435899266435f * y * (3.0f * xx - yy);
  c[10] = 2.890611442640554f * x * y * z;
  c[11] = 0.4570457994644658f * y * (5.0f * zz - 1.0f);
  c[12] = 0.3731763325901154f * z * (5.0f * zz - 3.0f);
  c[13] = 0.4570457994644658f * x * (5.0f * zz - 1.0f);
  c[14] = 1.445305721320277f * z * (xx - yy);
  c[15] = 0.5900435899266435f * x * (xx - 3.0f * yy);
}

// ray_bias[ray][o] = b[o] + sum_k W[o][const col k] * c_ray[k],  c_ray = [SH16(direction) | embedding row]
// (embedding row = mean_embedding on the eval/export path, Embedding[camera] in training).  Wave per ray,
// lane = output feature; the 48 x 64 weight slice (packed buffer, already transposed) is copied to LDS.
// Two rays per wave and many small workgroups: the per-ray chain camera -> embedding row -> 48 FMAs is pure
// latency, so it is hidden by occupancy rather than by a long loop.
// FROM_RAW: the weight slice and the bias come straight from the nn.Linear tensors (same values as the packed image
// holds) so that the kernel does not depend on the pack launch and can share a launch with it (k_prepare_field).
template <class Cfg, bool FROM_RAW>
__device__ __forceinline__ void color_ray_bias_block(const FieldPtrs& p, const float* __restrict__ packed,
                                                     const RaysDev& rays, const float* __restrict__ embedding,
                                                     const float* __restrict__ mean_embedding,
                                                     float* __restrict__ ray_bias, int block, int n_blocks) {
  // row stride 65: the FROM_RAW fill walks k fastest (consecutive words of a weight row), reads walk the lane
  __shared__ __attribute__((aligned(16))) float Wt[COLOR_CONST_K][65];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float b;
  if constexpr (FROM_RAW) {
    constexpr int IN = Cfg::in_dim(Cfg::L_COL0);
    for (int i = threadIdx.x; i < COLOR_CONST_K * 64; i += 256) {
      const int o = i / COLOR_CONST_K, k = i - o * COLOR_CONST_K;
      Wt[k][o] = p.w[Cfg::L_COL0][o * IN + color_const_col<Cfg>(k)];
    }
    b = p.b[Cfg::L_COL0][lane];
  } else {
    const float* src = packed + Cfg::LDS_FLOATS;
    for (int i = threadIdx.x; i < COLOR_CONST_K * 64; i += 256) Wt[i >> 6][i & 63] = src[i];
    b = packed[Cfg::W_TOTAL + Cfg::boff(Cfg::L_COL0) + lane];
  }
  __syncthreads();
  for (long long ray = (long long)block * 4 + wave; ray < rays.n_rays; ray += (long long)n_blocks * 4) {
    float c[16];
    sh16_all(rays.directions + 3 * ray, c);
    const float* emb = mean_embedding ? mean_embedding : embedding + (size_t)rays.cam[ray] * 32;
    float e[32];
#pragma unroll
    for (int k = 0; k < 32; k += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(emb + k);
      e[k] = v[0], e[k + 1] = v[1], e[k + 2] = v[2], e[k + 3] = v[3];
    }
    float acc = b;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = fmaf(Wt[k][lane], c[k], acc);
#pragma unroll
    for (int k = 0; k < 32; ++k) acc = fmaf(Wt[16 + k][lane], e[k], acc);
    ray_bias[(size_t)ray * 64 + lane] = acc;
  }
}
template <class Cfg>
__global__ __launch_bounds__(256) void k_color_ray_bias(const float* __restrict__ packed, RaysDev rays,
                                                        const float* __restrict__ embedding,
                                                        const float* __restrict__ mean_embedding,
                                                        float* __restrict__ ray_bias) {
  FieldPtrs none{};
  color_ray_bias_block<Cfg, false>(none, packed, rays, embedding, mean_embedding, ray_bias, blockIdx.x, gridDim.x);
}
static inline long long color_ray_bias_blocks(const RaysDev& rays) {
  long long blocks = (rays.n_rays + 7) / 8;
  if (blocks > 8ll * device_cu_count()) blocks = 8ll * device_cu_count();
  return blocks < 1 ? 1 : blocks;
}

template <class Cfg>
static inline void launch_color_ray_bias(const float* packed, const RaysDev& rays, const float* embedding,
                                         const float* mean_embedding, float* ray_bias, hipStream_t st) {
  hipLaunchKernelGGL((k_color_ray_bias<Cfg>), dim3((unsigned)color_ray_bias_blocks(rays)), dim3(256), 0, st, packed, rays,
                     embedding, mean_embedding, ray_bias);
}

}  // namespace fnr
