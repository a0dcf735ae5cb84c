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

constexpr int SCR_LD = 17;                 // padded row length of the transpose scratch
constexpr int SCR_FLOATS = 2 * 64 * SCR_LD;  // G^T and X^T, 64 feature rows each
constexpr int BWD_WAVES = 8;

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

// out (C-layout blocks IB0..IB0+NIBO-1 of the layer INPUT) = W^T * G^T
// LDS reads are issued in batches of 4*NIBO (one output block of the layer) ahead of their MFMAs: with one
// ds_read_b32 per MFMA the un-batched loop exposed one LDS round trip (~100 clk) per 4 MFMAs (128 clk).
template <int NOB, int NIB_TOTAL, int IB0, int NIBO>
__device__ __forceinline__ void mlp_layer_T(const float* __restrict__ P, const f32x4 (&G)[NOB], f32x4 (&out)[NIBO],
                                            int lane) {
  const int ip = lane & 15, kg = lane >> 4;
  const int a = ip >> 2, b = ip & 3;
#pragma unroll
  for (int q = 0; q < NIBO; ++q) out[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  float w[2][4][NIBO];
  auto load_ob = [&](int ob, float (&dst)[4][NIBO]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int slot = ((4 * kg + r) ^ a) + 16 * a;
#pragma unroll
      for (int q = 0; q < NIBO; ++q) dst[r][q] = P[((ob * NIB_TOTAL + (IB0 + q)) * 64 + slot) * 4 + b];
    }
  };
  load_ob(0, w[0]);
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob) {
    if (ob + 1 < NOB) load_ob(ob + 1, w[(ob + 1) & 1]);  // next block's weights in flight during these MFMAs
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < NIBO; ++q)
        out[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ob & 1][r][q], G[ob][r], out[q], 0, 0, 0);
  }
}

// acc[ob][ib] += sum over the tile's samples of G^T[16 ob + .][s] * X^T[16 ib + .][s]
template <int NOB, int NIB>
__device__ __forceinline__ void dw_accumulate(float* __restrict__ scr, const f32x4 (&G)[NOB], const f32x4 (&X)[NIB],
                                              f32x4 (&acc)[NOB][NIB], int lane) {
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
  // all 4*(NOB+NIB) fragment reads first, then the 4*NOB*NIB MFMAs: one LDS round trip per layer, not per k-step
  float av[4][NOB], bv[4][NIB];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) av[ks][ob] = sG[(16 * ob + j) * SCR_LD + 4 * ks + g];
#pragma unroll
    for (int ib = 0; ib < NIB; ++ib) bv[ks][ib] = sX[(16 * ib + j) * SCR_LD + 4 * ks + g];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
      for (int ib = 0; ib < NIB; ++ib)
        acc[ob][ib] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks][ob], bv[ks][ib], acc[ob][ib], 0, 0, 0);
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

// bias gradient of one layer for this tile: row-reduce over the 16 samples, one LDS atomic per feature
template <int NOB>
__device__ __forceinline__ void db_accumulate(float* __restrict__ lds_bias, const f32x4 (&G)[NOB], int lane) {
  const int j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = row16_sum(G[ob][r]);
      if (j == 0) atomicAdd(&lds_bias[16 * ob + 4 * g + r], v);
    }
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
template <class Cfg, int NOB, int NIB>
__device__ __forceinline__ void flush_dw(float* __restrict__ lds_acc, int l, const f32x4 (&acc)[NOB][NIB], int lane) {
  const int jn = lane & 15, g = lane >> 4;
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
    for (int ib = 0; ib < NIB; ++ib)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int slot = swz_slot(4 * g + r, jn >> 2);
        lds_acc[Cfg::woff(l) + ((ob * NIB + ib) * 64 + slot) * 4 + (jn & 3)] += acc[ob][ib][r];
      }
}

template <class Cfg, int BRANCH, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void k_field_mlp_bwd(
    const float* __restrict__ packed, RaysDev rays, int S, long long N, const float2* __restrict__ feats,
    const float* __restrict__ h_saved, const uint8_t* __restrict__ selector, const float* __restrict__ embedding, const float* __restrict__ d_density,
    const float* __restrict__ d_rgb, const float* __restrict__ d_logit, float* __restrict__ d_h,
    float2* __restrict__ d_feats, float* __restrict__ g_embedding, float* __restrict__ partials) {
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS + WAVES * SCR_FLOATS + Cfg::B_TOTAL];
  float* scr_all = lds + Cfg::LDS_FLOATS;
  float* lds_bias = scr_all + WAVES * SCR_FLOATS;  // bias-gradient accumulators (whole workgroup)
  stage_field_weights<Cfg>(lds, packed);
  for (int i = threadIdx.x; i < Cfg::B_TOTAL; i += blockDim.x) lds_bias[i] = 0.0f;
  __syncthreads();
  const float* Bv = lds + Cfg::W_TOTAL;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  float* scr = scr_all + wave * SCR_FLOATS;

  // dW accumulators of this branch
  constexpr int A0 = (BRANCH == BR_COLOR) ? 4 : (BRANCH == BR_SEM) ? 4 : 4;  // first layer of the branch: NOB
  f32x4 accA[4][(BRANCH == BR_COLOR) ? 4 : (BRANCH == BR_SEM) ? 1 : 2];      // col0 / sem0 / base0
  f32x4 accB[(BRANCH == BR_BASE) ? 1 : 4][4];                                 // col1 / sem1 / base1
  f32x4 accC[1][(BRANCH == BR_BASE) ? 1 : 4];                                 // col2 / head / (unused)
  (void)A0;
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
      f32x4 cin[4], c1[4], c2[4], c3[1];
      cin[0] = h[0];
      cin[1] = sh16_fragment(rays.directions + 3 * ray, g);
      const int cam = rays.cam[ray];
      const float* emb = embedding + (size_t)cam * 32;
      cin[2] = *reinterpret_cast<const f32x4*>(emb + 4 * g);
      cin[3] = *reinterpret_cast<const f32x4*>(emb + 16 + 4 * g);
      mlp_layer<4, 4>(lds + Cfg::woff(5), Bv + Cfg::boff(5), cin, c1, lane);
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
      dw_accumulate<1, 4>(scr, G3, c2, accC, lane);
      db_accumulate<1>(lds_bias + Cfg::boff(7), G3, lane);
      f32x4 G2[4];
      mlp_layer_T<1, 4, 0, 4>(lds + Cfg::woff(7), G3, G2, lane);
      relu_mask_(G2, c2);
      dw_accumulate<4, 4>(scr, G2, c1, accB, lane);
      db_accumulate<4>(lds_bias + Cfg::boff(6), G2, lane);
      f32x4 G1[4];
      mlp_layer_T<4, 4, 0, 4>(lds + Cfg::woff(6), G2, G1, lane);
      relu_mask_(G1, c1);
      dw_accumulate<4, 4>(scr, G1, cin, accA, lane);
      db_accumulate<4>(lds_bias + Cfg::boff(5), G1, lane);
      // dL/d[h] (block 0) and dL/d[embedding] (blocks 2,3); the SH block gets no gradient (no_grad encoding)
      f32x4 Gh[1], Ge[2];
      mlp_layer_T<4, 4, 0, 1>(lds + Cfg::woff(5), G1, Gh, lane);
      mlp_layer_T<4, 4, 2, 2>(lds + Cfg::woff(5), G1, Ge, lane);
      if (valid) *reinterpret_cast<f32x4*>(d_h + (size_t)n * 16 + 4 * g) = Gh[0];
      // appearance-embedding gradient: one row per camera.  Tiles usually sit inside one ray (S % 16 == 0).
      const int cam0 = __shfl(cam, lane & 48, 64);  // camera of sample 0 of this tile (same in all 4 lane groups)
      const bool uniform = __all((cam == cam0) || !valid);
      if (uniform) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = valid ? Ge[q][r] : 0.0f;
            v += __shfl_xor(v, 8, 64);
            v += __shfl_xor(v, 4, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 1, 64);
            if (j == 0) atomicAdd(&g_embedding[(size_t)cam0 * 32 + 16 * q + 4 * g + r], v);
          }
      } else if (valid) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int r = 0; r < 4; ++r) atomicAdd(&g_embedding[(size_t)cam * 32 + 16 * q + 4 * g + r], Ge[q][r]);
      }
    } else if constexpr (BRANCH == BR_SEM) {
      f32x4 s1[4], s2[4];
      mlp_layer<4, 1>(lds + Cfg::woff(2), Bv + Cfg::boff(2), h, s1, lane);
      relu_(s1);
      mlp_layer<4, 4>(lds + Cfg::woff(3), Bv + Cfg::boff(3), s1, s2, lane);
      f32x4 Gl[1];
      Gl[0] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (g == 0 && valid) Gl[0][0] = d_logit[n];
      dw_accumulate<1, 4>(scr, Gl, s2, accC, lane);   // SemanticFieldHead
      db_accumulate<1>(lds_bias + Cfg::boff(4), Gl, lane);
      f32x4 Gs2[4];
      mlp_layer_T<1, 4, 0, 4>(lds + Cfg::woff(4), Gl, Gs2, lane);  // no activation on mlp_semantics' last layer
      dw_accumulate<4, 4>(scr, Gs2, s1, accB, lane);
      db_accumulate<4>(lds_bias + Cfg::boff(3), Gs2, lane);
      f32x4 Gs1[4];
      mlp_layer_T<4, 4, 0, 4>(lds + Cfg::woff(3), Gs2, Gs1, lane);
      relu_mask_(Gs1, s1);
      dw_accumulate<4, 1>(scr, Gs1, h, accA, lane);   // input = detached geo: no dX
      db_accumulate<4>(lds_bias + Cfg::boff(2), Gs1, lane);
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
      dw_accumulate<1, 4>(scr, Gh, a1, accB, lane);
      db_accumulate<1>(lds_bias + Cfg::boff(1), Gh, lane);
      f32x4 Ga[4];
      mlp_layer_T<1, 4, 0, 4>(lds + Cfg::woff(1), Gh, Ga, lane);
      relu_mask_(Ga, a1);
      dw_accumulate<4, 2>(scr, Ga, x0, accA, lane);
      db_accumulate<4>(lds_bias + Cfg::boff(0), Ga, lane);
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
  __syncthreads();  // every wave is done with the weight image
  constexpr int L0 = (BRANCH == BR_COLOR) ? 5 : (BRANCH == BR_SEM) ? 2 : 0;
  constexpr int L1 = (BRANCH == BR_COLOR) ? 8 : (BRANCH == BR_SEM) ? 5 : 2;
  for (int i = Cfg::woff(L0) + threadIdx.x; i < Cfg::woff(L1); i += blockDim.x) lds[i] = 0.0f;
  __syncthreads();
  for (int turn = 0; turn < WAVES; ++turn) {
    if (wave == turn) {
      if constexpr (BRANCH == BR_COLOR) {
        flush_dw<Cfg, 4, 4>(lds, 5, accA, lane);
        flush_dw<Cfg, 4, 4>(lds, 6, accB, lane);
        flush_dw<Cfg, 1, 4>(lds, 7, accC, lane);
      } else if constexpr (BRANCH == BR_SEM) {
        flush_dw<Cfg, 4, 1>(lds, 2, accA, lane);
        flush_dw<Cfg, 4, 4>(lds, 3, accB, lane);
        flush_dw<Cfg, 1, 4>(lds, 4, accC, lane);
      } else {
        flush_dw<Cfg, 4, 2>(lds, 0, accA, lane);
        flush_dw<Cfg, 1, 4>(lds, 1, accB, lane);
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
  float s = 0.0f;
  int b = 0;
  for (; b + 8 <= nblocks; b += 8) {  // 8 independent loads in flight (the plain loop is one HBM latency per row)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partials[(size_t)(b + u) * TOT + idx];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; b < nblocks; ++b) s += partials[(size_t)b * TOT + idx];
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
      *dst += s;
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
      *dst += s;
    }
  }
}

int field_ptrs(const fnr_field_net* net, FieldPtrs& p);  // field_mlp.hip

}  // namespace fnr

using namespace fnr;

extern "C" size_t fnr_field_mlp_bwd_workspace_bytes(int64_t n_samples) {
  (void)n_samples;
  // per-workgroup partial weight-gradient images (<= one workgroup per CU) + dL/dh [N,16]
  const size_t blocks = (size_t)device_cu_count();
  return blocks * (FieldCfgBase::W_TOTAL + FieldCfgBase::B_TOTAL) * sizeof(float) +
         (size_t)(n_samples > 0 ? n_samples : 0) * 16 * sizeof(float) + 256 + FieldCfgBase::LDS_FLOATS * sizeof(float) +
         256;
}

extern "C" int fnr_field_mlp_bwd(const fnr_field_net* net, const fnr_field_net* grads, const fnr_rays* rays, int S,
                                 const float* feats, const float* h_saved, const uint8_t* selector,
                                 const float* d_density, const float* d_rgb, const float* d_logit, float* d_feats,
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
  FNR_CHECK_ARG(workspace_bytes >= fnr_field_mlp_bwd_workspace_bytes(N), "field_mlp_bwd: workspace too small");
  const long long n_tiles = (N + 15) / 16;
  const long long max_blocks = device_cu_count();
  float* partials = reinterpret_cast<float*>(workspace);
  float* d_h = partials + (size_t)max_blocks * (FieldCfgBase::W_TOTAL + FieldCfgBase::B_TOTAL);
  d_h = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(d_h) + 63) & ~(uintptr_t)63);
  float* packed = d_h + (size_t)N * 16;
  packed = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(packed) + 63) & ~(uintptr_t)63);
  hipStream_t st = as_stream(stream);
  const RaysDev rd = make_rays(rays);
  const float2* f2 = reinterpret_cast<const float2*>(feats);
  float2* df2 = reinterpret_cast<float2*>(d_feats);
  static const int color_waves = [] {
    const char* e = getenv("FNR_COLOR_WAVES");
    return (e && atoi(e) == 4) ? 4 : 8;
  }();
  FNR_PROF(OP_MLP_BWD, N);
  hipLaunchKernelGGL((k_pack_field_weights<FieldCfgBase>), dim3((FieldCfgBase::LDS_FLOATS + 255) / 256), dim3(256), 0,
                     st, p, packed);
  FNR_LAUNCH_CHECK();
  // every branch uses the same number of workgroups so that they share one partial-image buffer
  long long blocks = (n_tiles + 3) / 4;
  if (blocks > max_blocks) blocks = max_blocks;
#define FNR_BWD_LAUNCH(BR, WV)                                                                                       \
  hipLaunchKernelGGL((k_field_mlp_bwd<FieldCfgBase, BR, WV>), dim3((unsigned)blocks), dim3(64 * WV), 0, st, packed, rd, \
                     S, N, f2, h_saved, selector, net->embedding, d_density, d_rgb, d_logit, d_h, df2, grads->embedding,          \
                     partials);                                                                                       \
  FNR_LAUNCH_CHECK();
  if (color_waves == 4) {
    FNR_BWD_LAUNCH(BR_COLOR, 4)
  } else {
    FNR_BWD_LAUNCH(BR_COLOR, 8)
  }
  FNR_BWD_LAUNCH(BR_SEM, 8)
  FNR_BWD_LAUNCH(BR_BASE, 8)
#undef FNR_BWD_LAUNCH
  constexpr int TOT = FieldCfgBase::W_TOTAL + FieldCfgBase::B_TOTAL;
  hipLaunchKernelGGL((k_reduce_dw<FieldCfgBase>), dim3((TOT + 255) / 256), dim3(256), 0, st, partials, (int)blocks, gp);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}
