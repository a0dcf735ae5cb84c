// field_mlp_bwd.hip — backward of FruitField's MLP stack on fp32 MFMA (v_mfma_f32_16x16x4_f32).
//
// Autograd of fruit_field.py:187-281 for the training path (get_outputs): rgb loss flows through mlp_head into
// the geo features, the appearance embedding and the base MLP; the semantic loss only reaches mlp_semantics
// + SemanticFieldHead (geo is detached, fruit_field.py:263-265); dL/dsigma enters through trunc_exp.
//
// Structure (per 16-sample tile, one wave; see field_layers.hpp for the forward layout):
//   * forward activations are RECOMPUTED from the saved hash features (128 B/sample) instead of being
//     stored (1.3 KB/sample) — MFMA time is cheaper than HBM traffic here;
//   * dX^T = W^T dY^T reuses the forward LDS weight image: lane (i', kg) reads W[16 ob + 4 kg + r][col(ib, i')]
//     with one conflict-free ds_read_b32 (that is what the XOR swizzle of the image is for), dY^T stays in
//     registers as the B operand;
//   * dW = dY^T X needs the samples on the K axis: both operands are transposed through a per-wave LDS
//     scratch (64 x 17 floats each) and accumulated in registers across all tiles of the (persistent) wave;
//   * weight gradients leave the workgroup once: LDS reduction over its waves -> one partial image per
//     workgroup -> k_reduce_dw sums the partials deterministically and un-permutes into nn.Linear layout.
// The three branches (colour, semantic, base) are separate instantiations so that the dW accumulators
// (144 / 96 / 48 registers) fit next to the recomputed activations.
#include <stdlib.h>

#include "field_layers.hpp"

namespace fnr {

enum { BR_COLOR = 0, BR_SEM = 1, BR_BASE = 2 };

// padded row length of the transpose scratch: 20 makes both the C-layout writes (bank 16 g + 20 r + j) and the
// fragment reads (bank 20 j + 4 ks + g) conflict-free; 17 had 2-way conflicts between lane groups on the writes
constexpr int SCR_LD = 20;
constexpr int SCR_FLOATS = 2 * 64 * SCR_LD;  // G^T and X^T, 64 feature rows each

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);  // phase boundary for the scheduler too (keeps register pressure local)
}

// out (C-layout blocks IB0..IB0+NIBO-1 of the layer INPUT) = W^T * G^T
// LDS reads are issued in batches of 4*NIBO (one output block of the layer) ahead of their MFMAs: with one
// ds_read_b32 per MFMA the un-batched loop exposed one LDS round trip (~100 clk) per 4 MFMAs (128 clk).
template <int NOB, int NIB_TOTAL, int IB0, int NIBO>
__device__ __forceinline__ void mlp_layer_T(const float* __restrict__ P, const f32x4 (&G)[NOB], f32x4 (&out)[NIBO],
                                            int lane) {
  const int ip = lane & 15, kg = lane >> 4;
  const int a = ip >> 2, b = ip & 3;
  constexpr int QC = (NIBO > 2) ? 2 : NIBO;  // input blocks per pass: 2 keeps the double buffer at 16 registers
  static_assert(NIBO % QC == 0, "input blocks must split evenly");
#pragma unroll
  for (int q0 = 0; q0 < NIBO; q0 += QC) {
#pragma unroll
    for (int q = 0; q < QC; ++q) out[q0 + q] = f32x4{0.f, 0.f, 0.f, 0.f};
    float w[2][4][QC];
    auto load_ob = [&](int ob, float (&dst)[4][QC]) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int slot = ((4 * kg + r) ^ a) + 16 * a;
#pragma unroll
        for (int q = 0; q < QC; ++q) dst[r][q] = P[((ob * NIB_TOTAL + (IB0 + q0 + q)) * 64 + slot) * 4 + b];
      }
    };
    load_ob(0, w[0]);
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
      if (ob + 1 < NOB) load_ob(ob + 1, w[(ob + 1) & 1]);  // next block's weights in flight during these MFMAs
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < QC; ++q)
          out[q0 + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ob & 1][r][q], G[ob][r], out[q0 + q], 0, 0, 0);
    }
  }
}

// acc[ob][ib] += sum over the tile's samples of G^T[16 ob + .][s] * X^T[16 ib + .][s]
template <int NOB, int NIB>
__device__ __forceinline__ void dw_accumulate(float* __restrict__ scr, const f32x4 (&G)[NOB], const f32x4 (&X)[NIB],
                                              f32x4 (&acc)[NOB][NIB], float& bsum, int lane) {
  const int j = lane & 15, g = lane >> 4;
  float* sG = scr;
  float* sX = scr + 64 * SCR_LD;
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
    for (int r = 0; r < 4; ++r) sG[(16 * ob + 4 * g + r) * SCR_LD + j] = G[ob][r];
#pragma unroll
  for (int ib = 0; ib < NIB; ++ib)
#pragma unroll
    for (int r = 0; r < 4; ++r) sX[(16 * ib + 4 * g + r) * SCR_LD + j] = X[ib][r];
  wave_lds_fence();
  // bias gradient of the layer: lane = feature row of G^T, summed over the tile's 16 samples (4 conflict-free
  // ds_read_b128); replaces 4 NOB (DPP row sum + branch + LDS atomic) sequences that cut the tile loop into
  // ~36 basic blocks
  {
    const f32x4* row = reinterpret_cast<const f32x4*>(sG + lane * SCR_LD);
    const f32x4 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];
    const float t = ((r0[0] + r0[1]) + (r0[2] + r0[3])) + ((r1[0] + r1[1]) + (r1[2] + r1[3])) +
                    ((r2[0] + r2[1]) + (r2[2] + r2[3])) + ((r3[0] + r3[1]) + (r3[2] + r3[3]));
    bsum += (lane < 16 * NOB) ? t : 0.0f;
  }
  __builtin_amdgcn_sched_barrier(0);
  // fragment reads in two batches of 2*(NOB+NIB), each followed by its 2*NOB*NIB MFMAs: one LDS round trip per
  // batch instead of per k-step, at half the registers of a single batch (the colour branch was spilling)
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    float av[2][NOB], bv[2][NIB];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      const int ks = 2 * half + k2;
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob) av[k2][ob] = sG[(16 * ob + j) * SCR_LD + 4 * ks + g];
#pragma unroll
      for (int ib = 0; ib < NIB; ++ib) bv[k2][ib] = sX[(16 * ib + j) * SCR_LD + 4 * ks + g];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int ib = 0; ib < NIB; ++ib)
          acc[ob][ib] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k2][ob], bv[k2][ib], acc[ob][ib], 0, 0, 0);
  }
  wave_lds_fence();
}

// sum over the 16 lanes of a DPP row (= the 16 samples of the tile) without touching the LDS crossbar:
// quad xor 1, quad xor 2, row_half_mirror, row_mirror — every lane ends up with the row total
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
  return v;
}

template <int N>
__device__ __forceinline__ void relu_mask_(f32x4 (&G)[N], const f32x4 (&act)[N]) {
#pragma unroll
  for (int b = 0; b < N; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) G[b][r] = (act[b][r] > 0.0f) ? G[b][r] : 0.0f;
}

// add this wave's dW accumulators of layer `l` into the workgroup's LDS image (same index space as "P").
// Plain read-add-write: the caller serialises the waves (ds_add_f32 retires ~1 lane per 3 clocks on gfx950 —
// 74k float atomics per workgroup cost 92 us here; 8 barrier-separated rounds cost ~4 us).
template <class Cfg, int NOB, int NIB, int NIB_STRIDE = NIB>
__device__ __forceinline__ void flush_dw(float* __restrict__ lds_acc, int l, const f32x4 (&acc)[NOB][NIB], int lane) {
  const int jn = lane & 15, g = lane >> 4;
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
    for (int ib = 0; ib < NIB; ++ib)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int slot = swz_slot(4 * g + r, jn >> 2);
        lds_acc[Cfg::woff(l) + ((ob * NIB_STRIDE + ib) * 64 + slot) * 4 + (jn & 3)] += acc[ob][ib][r];
      }
}

template <class Cfg, int BRANCH, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void k_field_mlp_bwd(
    const float* __restrict__ packed, const float* __restrict__ ray_bias, RaysDev rays, int S, long long N,
    const float2* __restrict__ feats, const float* __restrict__ h_saved, const uint8_t* __restrict__ selector,
    const float* __restrict__ d_density, const float* __restrict__ d_rgb, const float* __restrict__ d_logit,
    float* __restrict__ d_h, float2* __restrict__ d_feats, float* __restrict__ gsum_tile,
    float* __restrict__ gsum_extra, float* __restrict__ partials) {
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS + WAVES * SCR_FLOATS + Cfg::B_TOTAL];
  float* scr_all = lds + Cfg::LDS_FLOATS;
  float* lds_bias = scr_all + WAVES * SCR_FLOATS;  // bias-gradient accumulators (whole workgroup)
  stage_field_weights<Cfg>(lds, packed);
  for (int i = threadIdx.x; i < Cfg::B_TOTAL; i += blockDim.x) lds_bias[i] = 0.0f;
  __syncthreads();
  const float* Bv = lds + Cfg::W_TOTAL;
  const int lane0 = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* scr = scr_all + wave * SCR_FLOATS;

  // dW accumulators of this branch
  constexpr int A0 = (BRANCH == BR_COLOR) ? 4 : (BRANCH == BR_SEM) ? 4 : 4;  // first layer of the branch: NOB
  f32x4 accA[4][(BRANCH == BR_BASE) ? 2 : 1];                                 // col0 (h block) / sem0 / base0
  f32x4 accB[(BRANCH == BR_BASE) ? 1 : 4][4];                                 // col1 / sem1 / base1
  f32x4 accC[1][(BRANCH == BR_BASE) ? 1 : 4];                                 // col2 / head / (unused)
  (void)A0;
  float bsA = 0.0f, bsB = 0.0f, bsC = 0.0f;  // bias-gradient sums of the same layers (lane = output feature)
#pragma unroll
  for (auto& row : accA)
#pragma unroll
    for (auto& v : row) v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (auto& row : accB)
#pragma unroll
    for (auto& v : row) v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (auto& row : accC)
#pragma unroll
    for (auto& v : row) v = f32x4{0.f, 0.f, 0.f, 0.f};

  const long long n_tiles = (N + 15) / 16;
  for (long long tile = (long long)blockIdx.x * WAVES + wave; tile < n_tiles;
       tile += (long long)gridDim.x * WAVES) {
    asm volatile("" ::: "memory");  // keep the LDS weight reads inside the loop (see field_mlp.hip)
    // ... and their addresses: hoisted out of the loop, the ~50 lane-dependent LDS offsets of the layers stayed
    // live across the whole body and were spilled; an opaque copy of the lane id makes them per-use VALU ops
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int j = lane & 15, g = lane >> 4;
    const long long n = tile * 16 + j;
    const bool valid = n < N;
    const long long nn = valid ? n : N - 1;
    const long long ray = nn / S;

    // ---- h = base MLP output: COLOR / SEM read the copy the forward pass saved (64 B/sample); BASE needs the
    // hidden layer too and recomputes it from the hash features ----
    f32x4 x0[2], a1[4], h[1];
    if constexpr (BRANCH == BR_BASE) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float2 v = feats[(size_t)(4 * m + g) * N + nn];
        x0[m >> 1][2 * (m & 1)] = v.x;
        x0[m >> 1][2 * (m & 1) + 1] = v.y;
      }
      mlp_layer<4, 2>(lds + Cfg::woff(0), Bv + Cfg::boff(0), x0, a1, lane);
      relu_(a1);
      mlp_layer<1, 4>(lds + Cfg::woff(1), Bv + Cfg::boff(1), a1, h, lane);
    } else {
      h[0] = *reinterpret_cast<const f32x4*>(h_saved + (size_t)nn * 16 + 4 * g);
    }

    if constexpr (BRANCH == BR_COLOR) {
      // colour MLP; its first layer only multiplies the h block, the ray-constant inputs come in as ray_bias
      // (field_layers.hpp: color_layer0)
      f32x4 c1[4], c2[4], c3[1];
      color_layer0<Cfg>(lds, ray_bias, ray, h, c1, lane);
      relu_(c1);
      mlp_layer<4, 4>(lds + Cfg::woff(6), Bv + Cfg::boff(6), c1, c2, lane);
      relu_(c2);
      mlp_layer<1, 4>(lds + Cfg::woff(7), Bv + Cfg::boff(7), c2, c3, lane);
      // d(pre-sigmoid) = d_rgb * rgb * (1 - rgb) on rows 0..2 (lane group 0), zero elsewhere
      f32x4 G3[1];
      G3[0] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (g == 0 && valid) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const float s = 1.0f / (1.0f + expf(-c3[0][r]));
          G3[0][r] = d_rgb[3 * n + r] * s * (1.0f - s);
        }
      }
      dw_accumulate<1, 4>(scr, G3, c2, accC, bsC, lane);
      f32x4 G2[4];
      mlp_layer_T<1, 4, 0, 4>(lds + Cfg::woff(7), G3, G2, lane);
      relu_mask_(G2, c2);
      dw_accumulate<4, 4>(scr, G2, c1, accB, bsB, lane);
      f32x4 G1[4];
      mlp_layer_T<4, 4, 0, 4>(lds + Cfg::woff(6), G2, G1, lane);
      relu_mask_(G1, c1);
      // layer 0: dW of the h block here; for the 48 ray-constant inputs (and the bias) the gradient is the outer
      // product (sum over the ray's samples of G1) x c_ray, so only the tile's 64 row sums of G1 leave the kernel
      // and k_color_ray_grads finishes the job per ray (weights, bias, appearance embedding).
      float gs = 0.0f;
      dw_accumulate<4, 1>(scr, G1, h, accA, gs, lane);
      const long long ray0 = __shfl(ray, lane & 48, 64);
      const bool uniform = __all(ray == ray0);  // invalid lanes were clamped to the last sample's ray
      if (uniform) {
        gsum_tile[(size_t)tile * 64 + lane] = gs;
      } else if (valid) {  // tile straddles rays (S % 16 != 0): per-sample contributions
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
#pragma unroll
          for (int r = 0; r < 4; ++r) atomicAdd(&gsum_extra[(size_t)ray * 64 + 16 * ob + 4 * g + r], G1[ob][r]);
      }
      f32x4 Gh[1];
      mlp_layer_T<4, 4, 0, 1>(lds + Cfg::woff(5), G1, Gh, lane);
      if (valid) *reinterpret_cast<f32x4*>(d_h + (size_t)n * 16 + 4 * g) = Gh[0];
    } else if constexpr (BRANCH == BR_SEM) {
      f32x4 s1[4], s2[4];
      mlp_layer<4, 1>(lds + Cfg::woff(2), Bv + Cfg::boff(2), h, s1, lane);
      relu_(s1);
      mlp_layer<4, 4>(lds + Cfg::woff(3), Bv + Cfg::boff(3), s1, s2, lane);
      f32x4 Gl[1];
      Gl[0] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (g == 0 && valid) Gl[0][0] = d_logit[n];
      dw_accumulate<1, 4>(scr, Gl, s2, accC, bsC, lane);   // SemanticFieldHead
      f32x4 Gs2[4];
      mlp_layer_T<1, 4, 0, 4>(lds + Cfg::woff(4), Gl, Gs2, lane);  // no activation on mlp_semantics' last layer
      dw_accumulate<4, 4>(scr, Gs2, s1, accB, bsB, lane);
      f32x4 Gs1[4];
      mlp_layer_T<4, 4, 0, 4>(lds + Cfg::woff(3), Gs2, Gs1, lane);
      relu_mask_(Gs1, s1);
      dw_accumulate<4, 1>(scr, Gs1, h, accA, bsA, lane);   // input = detached geo: no dX
    } else {
      // ---- base: dL/dh = colour-branch gradient (+ density through trunc_exp on row 0) ----
      f32x4 Gh[1];
      Gh[0] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (valid) {
        Gh[0] = *reinterpret_cast<const f32x4*>(d_h + (size_t)n * 16 + 4 * g);
        if (g == 0) {
          const bool sel = selector ? (selector[n] != 0) : true;
          const float te = expf(fminf(fmaxf(h[0][0], -15.0f), 15.0f));  // trunc_exp backward (fruit_field.py:191)
          Gh[0][0] = sel ? d_density[n] * te : 0.0f;                   // colour block has a zero row 0
        }
      }
      dw_accumulate<1, 4>(scr, Gh, a1, accB, bsB, lane);
      f32x4 Ga[4];
      mlp_layer_T<1, 4, 0, 4>(lds + Cfg::woff(1), Gh, Ga, lane);
      relu_mask_(Ga, a1);
      dw_accumulate<4, 2>(scr, Ga, x0, accA, bsA, lane);
      f32x4 Gx[2];
      mlp_layer_T<4, 2, 0, 2>(lds + Cfg::woff(0), Ga, Gx, lane);
      if (valid) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
          d_feats[(size_t)(4 * m + g) * N + n] = make_float2(Gx[m >> 1][2 * (m & 1)], Gx[m >> 1][2 * (m & 1) + 1]);
      }
    }
  }

  // ---- workgroup reduction of the weight gradients, then one partial image per workgroup ----
  const int lane = lane0;
  __syncthreads();  // every wave is done with the weight image
  constexpr int L0 = (BRANCH == BR_COLOR) ? 5 : (BRANCH == BR_SEM) ? 2 : 0;
  constexpr int L1 = (BRANCH == BR_COLOR) ? 8 : (BRANCH == BR_SEM) ? 5 : 2;
  for (int i = Cfg::woff(L0) + threadIdx.x; i < Cfg::woff(L1); i += blockDim.x) lds[i] = 0.0f;
  __syncthreads();
  for (int turn = 0; turn < WAVES; ++turn) {
    if (wave == turn) {
      if constexpr (BRANCH == BR_COLOR) {
        flush_dw<Cfg, 4, 1, 4>(lds, 5, accA, lane);
        flush_dw<Cfg, 4, 4>(lds, 6, accB, lane);
        flush_dw<Cfg, 1, 4>(lds, 7, accC, lane);
        lds_bias[Cfg::boff(6) + lane] += bsB;
        if (lane < 16) lds_bias[Cfg::boff(7) + lane] += bsC;
      } else if constexpr (BRANCH == BR_SEM) {
        flush_dw<Cfg, 4, 1>(lds, 2, accA, lane);
        flush_dw<Cfg, 4, 4>(lds, 3, accB, lane);
        flush_dw<Cfg, 1, 4>(lds, 4, accC, lane);
        lds_bias[Cfg::boff(2) + lane] += bsA;
        lds_bias[Cfg::boff(3) + lane] += bsB;
        if (lane < 16) lds_bias[Cfg::boff(4) + lane] += bsC;
      } else {
        flush_dw<Cfg, 4, 2>(lds, 0, accA, lane);
        flush_dw<Cfg, 1, 4>(lds, 1, accB, lane);
        lds_bias[Cfg::boff(0) + lane] += bsA;
        if (lane < 16) lds_bias[Cfg::boff(1) + lane] += bsB;
      }
    }
    __syncthreads();
  }
  float* part = partials + (size_t)blockIdx.x * (Cfg::W_TOTAL + Cfg::B_TOTAL);
  for (int i = Cfg::woff(L0) + threadIdx.x; i < Cfg::woff(L1); i += blockDim.x) part[i] = lds[i];
  for (int i = Cfg::boff(L0) + threadIdx.x; i < Cfg::boff(L1); i += blockDim.x)
    part[Cfg::W_TOTAL + i] = lds_bias[i];
}

// sum the per-workgroup partial images and add them into the nn.Linear-layout gradients
template <class Cfg>
__global__ __launch_bounds__(256) void k_reduce_dw(const float* __restrict__ partials, int nblocks, FieldPtrs grads) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  constexpr int TOT = Cfg::W_TOTAL + Cfg::B_TOTAL;
  if (idx >= TOT) return;
  // blockIdx.y takes every gridDim.y-th group of 8 partial images; the <= gridDim.y results meet with atomics
  float s = 0.0f;
  int b = 8 * blockIdx.y;
  for (; b + 8 <= nblocks; b += 8 * gridDim.y) {  // 8 independent loads in flight (the plain loop is one HBM latency per row)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partials[(size_t)(b + u) * TOT + idx];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  if (blockIdx.y == gridDim.y - 1)
    for (b = nblocks & ~7; b < nblocks; ++b) s += partials[(size_t)b * TOT + idx];
  if (s == 0.0f) return;
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
    if (out < Cfg::out_dim(l) && col >= 0) {
      float* dst = const_cast<float*>(grads.w[l]) + out * Cfg::in_dim(l) + col;
      atomicAdd(dst, s);
    }
  } else {
    const int bi = idx - Cfg::W_TOTAL;
    int l = 0;
#pragma unroll
    for (int q = 1; q < Cfg::NLAYERS; ++q)
      if (bi >= Cfg::boff(q)) l = q;
    const int o = bi - Cfg::boff(l);
    if (o < Cfg::out_dim(l)) {
      float* dst = const_cast<float*>(grads.b[l]) + o;
      atomicAdd(dst, s);
    }
  }
}

// Per-ray finish of mlp_head layer 0 (see the colour branch above).  For every ray: g = sum of its tiles' G1 row
// sums (+ the per-sample contributions of tiles that straddle rays), written to g_ray [n_rays, 64] for the
// embedding gradient; c = [SH16(direction) | Embedding[camera]].  The workgroup accumulates g (x) c (64 x 48) and
// sum g (bias) over its 16 rays and stores them into ITS partial weight-gradient image (layer 5, input blocks
// 1..3 and the bias, which the colour kernel left zero), so k_reduce_dw adds them like any other partial.
constexpr int RAYG_RB = 16;  // rays per workgroup pass
template <class Cfg>
__global__ __launch_bounds__(256) void k_color_ray_grads(RaysDev rays, int S, long long N,
                                                         const float* __restrict__ embedding,
                                                         const float* __restrict__ gsum_tile,
                                                         const float* __restrict__ gsum_extra,
                                                         float* __restrict__ g_ray, float* __restrict__ partials) {
  __shared__ float Gs[RAYG_RB][64];
  __shared__ __attribute__((aligned(16))) float Cs[RAYG_RB][COLOR_CONST_K];
  const int t = threadIdx.x, o = t & 63, kq = t >> 6;  // thread owns output o, constant inputs 12 kq .. 12 kq + 11
  const long long R = rays.n_rays;
  f32x4 acc[3];
  float accb = 0.0f;
#pragma unroll
  for (int q = 0; q < 3; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (long long base = (long long)blockIdx.x * RAYG_RB; base < R; base += (long long)gridDim.x * RAYG_RB) {
#pragma unroll
    for (int i = 0; i < RAYG_RB / 4; ++i) {
      const int rr = kq + 4 * i;
      const long long ray = base + rr;
      float v = 0.0f;
      if (ray < R) {
        const long long s_lo = ray * S, s_hi = s_lo + S;          // the ray's samples [s_lo, s_hi)
        for (long long tl = s_lo >> 4; tl <= (s_hi - 1) >> 4; ++tl) {  // tiles that overlap them
          const long long first = 16 * tl, last = (16 * tl + 15 < N) ? 16 * tl + 15 : N - 1;
          if (first >= s_lo && last < s_hi) v += gsum_tile[(size_t)tl * 64 + o];  // tile inside the ray
        }
        if (gsum_extra) v += gsum_extra[(size_t)ray * 64 + o];
        g_ray[(size_t)ray * 64 + o] = v;
      }
      Gs[rr][o] = v;
    }
#pragma unroll
    for (int u = 0; u < RAYG_RB * COLOR_CONST_K / 256; ++u) {
      const int idx = t + 256 * u;
      const int rr = idx / COLOR_CONST_K, k = idx - rr * COLOR_CONST_K;
      const long long ray = base + rr;
      float v = 0.0f;
      if (ray < R) {
        if (k < 16) {
          float c[16];
          sh16_all(rays.directions + 3 * ray, c);
#pragma unroll
          for (int q = 0; q < 16; ++q) v = (k == q) ? c[q] : v;
        } else {
          v = embedding[(size_t)rays.cam[ray] * 32 + (k - 16)];
        }
      }
      Cs[rr][k] = v;
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < RAYG_RB; ++rr) {
      const float gv = Gs[rr][o];
      const f32x4* cr = reinterpret_cast<const f32x4*>(&Cs[rr][12 * kq]);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const f32x4 cv = cr[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[q][e] = fmaf(gv, cv[e], acc[q][e]);
      }
      accb += gv;
    }
    __syncthreads();
  }
  float* part = partials + (size_t)blockIdx.x * (Cfg::W_TOTAL + Cfg::B_TOTAL);
  const int ob = o >> 4, i = o & 15;
#pragma unroll
  for (int q = 0; q < 12; ++q) {
    const int k = 12 * kq + q, ib = 1 + (k >> 4), kk = k & 15;
    part[Cfg::woff(5) + ((ob * 4 + ib) * 64 + swz_slot(i, kk >> 2)) * 4 + (kk & 3)] = acc[q >> 2][q & 3];
  }
  if (kq == 0) part[Cfg::W_TOTAL + Cfg::boff(5) + o] = accb;
}
static_assert(RAYG_RB * COLOR_CONST_K % 256 == 0, "staging loop covers the batch exactly");

// appearance-embedding gradient (fruit_field.py:251 Embedding lookup): one workgroup per camera gathers the g rows
// of its rays (each wave tests 64 rays per ballot), then g_embedding[c][k] += sum_o W5[o][emb col k] * gcam[o].
// No atomics: direct adds into the [n_images, 32] table serialise at ~12 ns per same-address add (393k adds on
// 90 rows made the colour branch 4x slower than its MFMA time).
template <class Cfg>
__global__ __launch_bounds__(1024) void k_embedding_grad(RaysDev rays, const float* __restrict__ g_ray,
                                                         const float* __restrict__ packed,
                                                         float* __restrict__ g_embedding) {
  __shared__ float red[16][64];
  const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.0f;
  for (long long base = 64 * wave; base < rays.n_rays; base += 1024) {
    const long long r = base + lane;
    unsigned long long match = __ballot(r < rays.n_rays && rays.cam[r] == c);
    while (match) {
      const int bit = __builtin_ctzll(match);
      match &= match - 1;
      acc += g_ray[(size_t)(base + bit) * 64 + lane];
    }
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (threadIdx.x < 64) {
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += red[w][lane];
    red[0][lane] = s;
  }
  __syncthreads();
  // thread (k = t >> 5, oo = t & 31): two of the 64 products of column k, then a 32-lane butterfly
  const int k = threadIdx.x >> 5, oo = threadIdx.x & 31;
  const float* Wt = packed + Cfg::LDS_FLOATS + (16 + k) * 64;  // transposed slice, row = embedding column k
  float s = fmaf(Wt[oo], red[0][oo], Wt[oo + 32] * red[0][oo + 32]);
#pragma unroll
  for (int m = 16; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
  if (oo == 0 && s != 0.0f) g_embedding[(size_t)c * 32 + k] += s;
}

int field_ptrs(const fnr_field_net* net, FieldPtrs& p);  // field_mlp.hip

}  // namespace fnr

using namespace fnr;

namespace {
struct BwdWorkspace {
  float *partials, *d_h, *packed, *ray_bias, *gsum_tile, *g_ray, *gsum_extra;
  size_t bytes;
};
// carve the workspace: per-workgroup partial weight-gradient images (<= one workgroup per CU), dL/dh [N,16], the
// fragment image, and the per-ray colour terms
BwdWorkspace bwd_workspace(void* base, long long n_rays, int S) {
  const long long N = n_rays * (long long)S, n_tiles = (N + 15) / 16;
  uintptr_t p = reinterpret_cast<uintptr_t>(base);
  auto take = [&](size_t floats) {
    p = (p + 255) & ~(uintptr_t)255;
    float* r = reinterpret_cast<float*>(p);
    p += floats * sizeof(float);
    return r;
  };
  BwdWorkspace w;
  w.partials = take((size_t)device_cu_count() * (FieldCfgBase::W_TOTAL + FieldCfgBase::B_TOTAL));
  w.d_h = take((size_t)N * 16);
  w.packed = take(FieldCfgBase::PACKED_FLOATS);
  w.ray_bias = take((size_t)n_rays * 64);
  w.gsum_tile = take((size_t)n_tiles * 64);
  w.g_ray = take((size_t)n_rays * 64);
  w.gsum_extra = take((size_t)n_rays * 64);
  w.bytes = p - reinterpret_cast<uintptr_t>(base) + 256;
  return w;
}
}  // namespace

extern "C" size_t fnr_field_mlp_bwd_workspace_bytes(int64_t n_rays, int S) {
  if (n_rays < 0 || S <= 0) return 0;
  return bwd_workspace(nullptr, n_rays, S).bytes;
}

extern "C" int fnr_field_mlp_bwd(const fnr_field_net* net, const fnr_field_net* grads, const fnr_rays* rays, int S,
                                 const float* feats, const float* h_saved, const float* ray_bias_saved,
                                 const float* packed_saved, const uint8_t* selector, const float* d_density, const float* d_rgb, const float* d_logit, float* d_feats,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  FNR_CHECK_ARG(net && grads && rays && feats && h_saved && d_density && d_rgb && d_logit && d_feats && workspace && S > 0,
                "field_mlp_bwd: null argument");
  FNR_CHECK_ARG(rays->directions && rays->camera_indices && net->embedding && grads->embedding,
                "field_mlp_bwd: training path needs directions, camera indices and the embedding (+ its gradient)");
  FieldPtrs p, gp;
  int rc = field_ptrs(net, p);
  if (rc) return rc;
  rc = field_ptrs(grads, gp);
  if (rc) return rc;
  const long long N = rays->n_rays * (long long)S;
  if (N == 0) return FNR_OK;
  FNR_CHECK_ARG(workspace_bytes >= fnr_field_mlp_bwd_workspace_bytes(rays->n_rays, S),
                "field_mlp_bwd: workspace too small");
  const long long n_tiles = (N + 15) / 16;
  const long long max_blocks = device_cu_count();
  const BwdWorkspace ws = bwd_workspace(workspace, rays->n_rays, S);
  float* partials = ws.partials;
  float* d_h = ws.d_h;
  float* packed = ws.packed;
  float* gsum_extra = (S % 16 != 0) ? ws.gsum_extra : nullptr;  // only tiles that straddle rays use it
  hipStream_t st = as_stream(stream);
  const RaysDev rd = make_rays(rays);
  const float2* f2 = reinterpret_cast<const float2*>(feats);
  float2* df2 = reinterpret_cast<float2*>(d_feats);
  static const int color_waves = [] {
    const char* e = getenv("FNR_COLOR_WAVES");
    return (e && atoi(e) == 4) ? 4 : 8;
  }();
  FNR_PROF(OP_MLP_BWD, N);
  if (packed_saved) {
    packed = const_cast<float*>(packed_saved);  // the forward pass's fragment image of the same weights
  } else {
    launch_pack_field_weights<FieldCfgBase>(p, packed, st);
    FNR_LAUNCH_CHECK();
  }
  const float* ray_bias = ray_bias_saved;
  if (!ray_bias) {
    launch_color_ray_bias<FieldCfgBase>(packed, rd, net->embedding, nullptr, ws.ray_bias, st);
    FNR_LAUNCH_CHECK();
    ray_bias = ws.ray_bias;
  }
  if (gsum_extra) FNR_HIP(hipMemsetAsync(gsum_extra, 0, (size_t)rays->n_rays * 64 * sizeof(float), st));
  // every branch uses the same number of workgroups so that they share one partial-image buffer
  long long blocks = (n_tiles + 3) / 4;
  if (blocks > max_blocks) blocks = max_blocks;
#define FNR_BWD_LAUNCH(BR, WV)                                                                                       \
  hipLaunchKernelGGL((k_field_mlp_bwd<FieldCfgBase, BR, WV>), dim3((unsigned)blocks), dim3(64 * WV), 0, st, packed,     \
                     ray_bias, rd, S, N, f2, h_saved, selector, d_density, d_rgb, d_logit, d_h, df2, ws.gsum_tile,  \
                     gsum_extra, partials);                                                                                       \
  FNR_LAUNCH_CHECK();
  if (color_waves == 4) {
    FNR_BWD_LAUNCH(BR_COLOR, 4)
  } else {
    FNR_BWD_LAUNCH(BR_COLOR, 8)
  }
  {
    // per-ray finish of mlp_head layer 0: every workgroup owns one partial image row that exists
    long long rb = (rays->n_rays + RAYG_RB - 1) / RAYG_RB;
    if (rb > blocks) rb = blocks;
    hipLaunchKernelGGL((k_color_ray_grads<FieldCfgBase>), dim3((unsigned)rb), dim3(256), 0, st, rd, S, N, net->embedding,
                       ws.gsum_tile, gsum_extra, ws.g_ray, partials);
    FNR_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_embedding_grad<FieldCfgBase>), dim3((unsigned)net->n_images), dim3(1024), 0, st, rd, ws.g_ray,
                       packed, grads->embedding);
    FNR_LAUNCH_CHECK();
  }
  FNR_BWD_LAUNCH(BR_SEM, 8)
  FNR_BWD_LAUNCH(BR_BASE, 8)
#undef FNR_BWD_LAUNCH
  constexpr int TOT = FieldCfgBase::W_TOTAL + FieldCfgBase::B_TOTAL;
  hipLaunchKernelGGL((k_reduce_dw<FieldCfgBase>), dim3((TOT + 255) / 256, (unsigned)(blocks >= 64 ? 8 : 1)), dim3(256), 0, st,
                     partials, (int)blocks, gp);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}
