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
//     scratch (2 x 64 rows x 20 floats) and accumulated in registers across all tiles of the (persistent) wave;
//   * weight gradients leave the workgroup once: LDS reduction over its waves -> one partial image per
//     workgroup -> k_reduce_dw sums the partials deterministically and un-permutes into nn.Linear layout.
// The branches are separate instantiations so that the dW accumulators fit next to the recomputed activations:
//   FieldCfgBase (`fruit_nerf`):      colour / semantic / base with 144 / 96 / 48 accumulator registers, 8 waves each;
//   FieldCfgBig  (`fruit_nerf_big`):  colour (160) and base (64) as above with a 32-wide h; the semantic branch
//     30 -> 128 -> 128 -> 64 -> 1 needs 464 accumulator registers and 119 KB of weights, so it runs as TWO launches
//     of 4 waves x <= 512 registers with all four layers in LDS: SEM_A owns dW of sem0, sem2 and the head (208
//     registers), SEM_B owns the 128 x 128 layer (256 registers); SEM_B repeats the forward up to s2 and the dX chain
//     down to Gs2 (1.35x the algorithmic MACs of the branch, no activations through HBM).
#include <stdlib.h>

#include "field_layers.hpp"
#include "sequencer.hpp"

namespace fnr {

enum { BR_COLOR = 0, BR_SEM = 1, BR_BASE = 2, BR_SEM_A = 3, BR_SEM_B = 4 };

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

// acc[OB0 + ob][IB0 + ib] += sum over the tile's samples of G^T[16 ob + .][s] * X^T[16 ib + .][s], ob < NOB <= 4,
// ib < NIB <= 4 (G, X point at the first of NOB / NIB blocks; wider layers call this per 4 x 4 group of blocks so
// that the scratch stays at 2 x 64 rows per wave).  BIAS: also add the tile's row sums of G to bsum (lane = row).
template <int NOB, int NIB, int NOBT, int NIBT, int OB0, int IB0, bool BIAS>
__device__ __forceinline__ void dw_accumulate_sub(float* __restrict__ scr, const f32x4* __restrict__ G,
                                                  const f32x4* __restrict__ X, f32x4 (&acc)[NOBT][NIBT], float& bsum,
                                                  int lane) {
  static_assert(NOB <= 4 && NIB <= 4 && OB0 + NOB <= NOBT && IB0 + NIB <= NIBT, "block group out of range");
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
  if constexpr (BIAS) {
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
          acc[OB0 + ob][IB0 + ib] =
              __builtin_amdgcn_mfma_f32_16x16x4f32(av[k2][ob], bv[k2][ib], acc[OB0 + ob][IB0 + ib], 0, 0, 0);
  }
  wave_lds_fence();
}

template <int NOB, int NIB>
__device__ __forceinline__ void dw_accumulate(float* __restrict__ scr, const f32x4 (&G)[NOB], const f32x4 (&X)[NIB],
                                              f32x4 (&acc)[NOB][NIB], float& bsum, int lane) {
  dw_accumulate_sub<NOB, NIB, NOB, NIB, 0, 0, true>(scr, G, X, acc, bsum, lane);
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

// add this wave's dW accumulators of a layer into the workgroup's LDS image (`W` = the layer's block in LDS, same
// index space as the weights).  Plain read-add-write: the caller serialises the waves (ds_add_f32 retires ~1 lane per
// 3 clocks on gfx950 — 74k float atomics per workgroup cost 92 us here; 8 barrier-separated rounds cost ~4 us).
template <int NOB, int NIB, int NIB_STRIDE = NIB>
__device__ __forceinline__ void flush_dw(float* __restrict__ W, const f32x4 (&acc)[NOB][NIB], int lane) {
  const int jn = lane & 15, g = lane >> 4;
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
    for (int ib = 0; ib < NIB; ++ib)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int slot = swz_slot(4 * g + r, jn >> 2);
        W[((ob * NIB_STRIDE + ib) * 64 + slot) * 4 + (jn & 3)] += acc[ob][ib][r];
      }
}

template <class A>
__device__ __forceinline__ void zero_acc(A& acc) {
#pragma unroll
  for (auto& row : acc)
#pragma unroll
    for (auto& v : row) v = f32x4{0.f, 0.f, 0.f, 0.f};
}

// the layers a branch keeps in LDS (forward recompute + transposed reads) = the layers whose gradients it owns: a
// contiguous range of the fragment image for both shapes (FieldCfgBase used to stage the whole 75 KB image per branch)
template <class Cfg, int BRANCH>
struct BwdRange;
template <class Cfg>
struct BwdRange<Cfg, BR_COLOR> { using type = LdsRange<Cfg, Cfg::L_COL0, Cfg::L_COL2 + 1>; };
template <class Cfg>
struct BwdRange<Cfg, BR_BASE> { using type = LdsRange<Cfg, Cfg::L_BASE0, Cfg::L_BASE1 + 1>; };
template <class Cfg>
struct BwdRange<Cfg, BR_SEM> { using type = LdsRange<Cfg, Cfg::L_SEM0, Cfg::L_HEAD + 1>; };
template <class Cfg>
struct BwdRange<Cfg, BR_SEM_A> { using type = LdsRange<Cfg, Cfg::L_SEM0, Cfg::L_HEAD + 1>; };
template <class Cfg>
struct BwdRange<Cfg, BR_SEM_B> { using type = LdsRange<Cfg, Cfg::L_SEM0, Cfg::L_HEAD + 1>; };

// ---- colour branch: mlp_head (fruit_field.py:158-166,270-281) -----------------------------------------------------
template <class Cfg, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void k_field_mlp_bwd_color(
    const float* __restrict__ packed, const float* __restrict__ ray_bias, RaysDev rays, int S, long long N,
    const float* __restrict__ h_saved, const float* __restrict__ d_rgb, float* __restrict__ d_h,
    float* __restrict__ gsum_tile, float* __restrict__ gsum_extra, float* __restrict__ partials) {
  using R = typename BwdRange<Cfg, BR_COLOR>::type;
  constexpr int HB = Cfg::HB;
  constexpr int LC0 = Cfg::L_COL0, LC1 = Cfg::L_COL1, LC2 = Cfg::L_COL2;
  __shared__ __attribute__((aligned(16))) float lds[R::FLOATS + WAVES * SCR_FLOATS + 80];
  float* scr_all = lds + R::FLOATS;
  float* lds_bias = scr_all + WAVES * SCR_FLOATS;  // bias-gradient accumulators of col1 (64) and col2 (16)
  R::template stage<64 * WAVES>(lds, packed);
  for (int i = threadIdx.x; i < 80; i += blockDim.x) lds_bias[i] = 0.0f;
  __syncthreads();
  const int lane0 = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* scr = scr_all + wave * SCR_FLOATS;

  f32x4 accA[4][HB];  // col0, h blocks
  f32x4 accB[4][4];   // col1
  f32x4 accC[1][4];   // col2
  float bsB = 0.0f, bsC = 0.0f;
  zero_acc(accA);
  zero_acc(accB);
  zero_acc(accC);

  const long long n_tiles = (N + 15) / 16;
  for (long long tile = (long long)blockIdx.x * WAVES + wave; tile < n_tiles; tile += (long long)gridDim.x * WAVES) {
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

    f32x4 h[HB];
#pragma unroll
    for (int b = 0; b < HB; ++b) h[b] = *reinterpret_cast<const f32x4*>(h_saved + (size_t)nn * (16 * HB) + 16 * b + 4 * g);

    // colour MLP; its first layer only multiplies the h blocks, the ray-constant inputs come in as ray_bias
    // (field_layers.hpp: color_layer0)
    f32x4 c1[4], c2[4], c3[1];
    color_layer0<Cfg>(R::w(lds, LC0), ray_bias, ray, h, c1, lane);
    relu_(c1);
    mlp_layer<4, 4>(R::w(lds, LC1), R::b(lds, LC1), c1, c2, lane);
    relu_(c2);
    mlp_layer<1, 4>(R::w(lds, LC2), R::b(lds, LC2), c2, c3, lane);
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
    mlp_layer_T<1, 4, 0, 4>(R::w(lds, LC2), G3, G2, lane);
    relu_mask_(G2, c2);
    dw_accumulate<4, 4>(scr, G2, c1, accB, bsB, lane);
    f32x4 G1[4];
    mlp_layer_T<4, 4, 0, 4>(R::w(lds, LC1), G2, G1, lane);
    relu_mask_(G1, c1);
    // layer 0: dW of the h blocks here; for the 48 ray-constant inputs (and the bias) the gradient is the outer
    // product (sum over the ray's samples of G1) x c_ray, so only the tile's 64 row sums of G1 leave the kernel
    // and k_color_ray_grads finishes the job per ray (weights, bias, appearance embedding).
    float gs = 0.0f;
    dw_accumulate<4, HB>(scr, G1, h, accA, gs, lane);
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
    f32x4 Gh[HB];
    mlp_layer_T<4, HB + 3, 0, HB>(R::w(lds, LC0), G1, Gh, lane);
    if (valid) {
#pragma unroll
      for (int b = 0; b < HB; ++b) *reinterpret_cast<f32x4*>(d_h + (size_t)n * (16 * HB) + 16 * b + 4 * g) = Gh[b];
    }
  }

  // ---- workgroup reduction of the weight gradients, then this branch's part of the workgroup's partial image ----
  const int lane = lane0;
  __syncthreads();  // every wave is done with the weight image
  float* acc_lds = R::w(lds, LC0);
  constexpr int ACC_FLOATS = Cfg::woff(LC2 + 1) - Cfg::woff(LC0);
  for (int i = threadIdx.x; i < ACC_FLOATS; i += blockDim.x) acc_lds[i] = 0.0f;
  __syncthreads();
  for (int turn = 0; turn < WAVES; ++turn) {
    if (wave == turn) {
      flush_dw<4, HB, HB + 3>(R::w(lds, LC0), accA, lane);
      flush_dw<4, 4>(R::w(lds, LC1), accB, lane);
      flush_dw<1, 4>(R::w(lds, LC2), accC, lane);
      lds_bias[lane] += bsB;
      if (lane < 16) lds_bias[64 + lane] += bsC;
    }
    __syncthreads();
  }
  float* part = partials + (size_t)blockIdx.x * (Cfg::W_TOTAL + Cfg::B_TOTAL);
  for (int i = threadIdx.x; i < ACC_FLOATS; i += blockDim.x) part[Cfg::woff(LC0) + i] = acc_lds[i];
  // col0's bias slot belongs to k_color_ray_grads (which owns some workgroups' images only): zero it here
  for (int i = threadIdx.x; i < 64; i += blockDim.x) part[Cfg::W_TOTAL + Cfg::boff(LC0) + i] = 0.0f;
  for (int i = threadIdx.x; i < 64; i += blockDim.x) part[Cfg::W_TOTAL + Cfg::boff(LC1) + i] = lds_bias[i];
  for (int i = threadIdx.x; i < 16; i += blockDim.x) part[Cfg::W_TOTAL + Cfg::boff(LC2) + i] = lds_bias[64 + i];
}

// ---- base branch: mlp_base_mlp (fruit_field.py:132-140,187-193) ----------------------------------------------------
template <class Cfg, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void k_field_mlp_bwd_base(
    const float* __restrict__ packed, long long N, const float2* __restrict__ feats,
    const uint8_t* __restrict__ selector, const float* __restrict__ d_density, const float* __restrict__ d_h,
    float2* __restrict__ d_feats, float* __restrict__ partials) {
  using R = typename BwdRange<Cfg, BR_BASE>::type;
  constexpr int HB = Cfg::HB;
  constexpr int LB0 = Cfg::L_BASE0, LB1 = Cfg::L_BASE1;
  __shared__ __attribute__((aligned(16))) float lds[R::FLOATS + WAVES * SCR_FLOATS + 64 + 16 * HB];
  float* scr_all = lds + R::FLOATS;
  float* lds_bias = scr_all + WAVES * SCR_FLOATS;
  R::template stage<64 * WAVES>(lds, packed);
  for (int i = threadIdx.x; i < 64 + 16 * HB; i += blockDim.x) lds_bias[i] = 0.0f;
  __syncthreads();
  const int lane0 = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* scr = scr_all + wave * SCR_FLOATS;

  f32x4 accA[4][2];   // base0
  f32x4 accB[HB][4];  // base1
  float bsA = 0.0f, bsB = 0.0f;
  zero_acc(accA);
  zero_acc(accB);

  const long long n_tiles = (N + 15) / 16;
  for (long long tile = (long long)blockIdx.x * WAVES + wave; tile < n_tiles; tile += (long long)gridDim.x * WAVES) {
    asm volatile("" ::: "memory");
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int j = lane & 15, g = lane >> 4;
    const long long n = tile * 16 + j;
    const bool valid = n < N;
    const long long nn = valid ? n : N - 1;

    // the hidden layer is recomputed from the hash features
    f32x4 x0[2], a1[4], h[HB];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float2 v = feats[(size_t)(4 * m + g) * N + nn];
      x0[m >> 1][2 * (m & 1)] = v.x;
      x0[m >> 1][2 * (m & 1) + 1] = v.y;
    }
    mlp_layer<4, 2>(R::w(lds, LB0), R::b(lds, LB0), x0, a1, lane);
    relu_(a1);
    mlp_layer<HB, 4>(R::w(lds, LB1), R::b(lds, LB1), a1, h, lane);

    // dL/dh = colour-branch gradient (+ density through trunc_exp on row 0)
    f32x4 Gh[HB];
#pragma unroll
    for (int b = 0; b < HB; ++b) Gh[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (valid) {
#pragma unroll
      for (int b = 0; b < HB; ++b) Gh[b] = *reinterpret_cast<const f32x4*>(d_h + (size_t)n * (16 * HB) + 16 * b + 4 * g);
      if (g == 0) {
        const bool sel = selector ? (selector[n] != 0) : true;
        const float te = expf(fminf(fmaxf(h[0][0], -15.0f), 15.0f));  // trunc_exp backward (fruit_field.py:191)
        Gh[0][0] = sel ? d_density[n] * te : 0.0f;                   // colour block has a zero row 0
      }
    }
    dw_accumulate<HB, 4>(scr, Gh, a1, accB, bsB, lane);
    f32x4 Ga[4];
    mlp_layer_T<HB, 4, 0, 4>(R::w(lds, LB1), Gh, Ga, lane);
    relu_mask_(Ga, a1);
    dw_accumulate<4, 2>(scr, Ga, x0, accA, bsA, lane);
    f32x4 Gx[2];
    mlp_layer_T<4, 2, 0, 2>(R::w(lds, LB0), Ga, Gx, lane);
    if (valid) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
        d_feats[(size_t)(4 * m + g) * N + n] = make_float2(Gx[m >> 1][2 * (m & 1)], Gx[m >> 1][2 * (m & 1) + 1]);
    }
  }

  const int lane = lane0;
  __syncthreads();
  float* acc_lds = R::w(lds, LB0);
  constexpr int ACC_FLOATS = Cfg::woff(LB1 + 1) - Cfg::woff(LB0);
  for (int i = threadIdx.x; i < ACC_FLOATS; i += blockDim.x) acc_lds[i] = 0.0f;
  __syncthreads();
  for (int turn = 0; turn < WAVES; ++turn) {
    if (wave == turn) {
      flush_dw<4, 2>(R::w(lds, LB0), accA, lane);
      flush_dw<HB, 4>(R::w(lds, LB1), accB, lane);
      lds_bias[lane] += bsA;
      if (lane < 16 * HB) lds_bias[64 + lane] += bsB;
    }
    __syncthreads();
  }
  float* part = partials + (size_t)blockIdx.x * (Cfg::W_TOTAL + Cfg::B_TOTAL);
  for (int i = threadIdx.x; i < ACC_FLOATS; i += blockDim.x) part[Cfg::woff(LB0) + i] = acc_lds[i];
  for (int i = threadIdx.x; i < 64; i += blockDim.x) part[Cfg::W_TOTAL + Cfg::boff(LB0) + i] = lds_bias[i];
  for (int i = threadIdx.x; i < 16 * HB; i += blockDim.x) part[Cfg::W_TOTAL + Cfg::boff(LB1) + i] = lds_bias[64 + i];
}

// ---- semantic branch, `fruit_nerf` shape: 15 -> 64 -> 64 -> head (fruit_field.py:144-156,263-268) ----------------------
template <class Cfg, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void k_field_mlp_bwd_sem(
    const float* __restrict__ packed, long long N, const float* __restrict__ h_saved,
    const float* __restrict__ d_logit, float* __restrict__ partials) {
  static_assert(Cfg::NSEM == 2 && Cfg::HB == 1, "the single-launch semantic branch is the fruit_nerf shape");
  using R = typename BwdRange<Cfg, BR_SEM>::type;
  constexpr int LS0 = Cfg::L_SEM0, LS1 = Cfg::L_SEM1, LH = Cfg::L_HEAD;
  __shared__ __attribute__((aligned(16))) float lds[R::FLOATS + WAVES * SCR_FLOATS + 144];
  float* scr_all = lds + R::FLOATS;
  float* lds_bias = scr_all + WAVES * SCR_FLOATS;  // sem0 (64), sem1 (64), head (16)
  R::template stage<64 * WAVES>(lds, packed);
  for (int i = threadIdx.x; i < 144; i += blockDim.x) lds_bias[i] = 0.0f;
  __syncthreads();
  const int lane0 = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* scr = scr_all + wave * SCR_FLOATS;

  f32x4 accA[4][1], accB[4][4], accC[1][4];
  float bsA = 0.0f, bsB = 0.0f, bsC = 0.0f;
  zero_acc(accA);
  zero_acc(accB);
  zero_acc(accC);

  const long long n_tiles = (N + 15) / 16;
  for (long long tile = (long long)blockIdx.x * WAVES + wave; tile < n_tiles; tile += (long long)gridDim.x * WAVES) {
    asm volatile("" ::: "memory");
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int j = lane & 15, g = lane >> 4;
    const long long n = tile * 16 + j;
    const bool valid = n < N;
    const long long nn = valid ? n : N - 1;
    f32x4 h[1];
    h[0] = *reinterpret_cast<const f32x4*>(h_saved + (size_t)nn * 16 + 4 * g);
    f32x4 s1[4], s2[4];
    mlp_layer<4, 1>(R::w(lds, LS0), R::b(lds, LS0), h, s1, lane);
    relu_(s1);
    mlp_layer<4, 4>(R::w(lds, LS1), R::b(lds, LS1), s1, s2, lane);
    f32x4 Gl[1];
    Gl[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (g == 0 && valid) Gl[0][0] = d_logit[n];
    dw_accumulate<1, 4>(scr, Gl, s2, accC, bsC, lane);   // SemanticFieldHead
    f32x4 Gs2[4];
    mlp_layer_T<1, 4, 0, 4>(R::w(lds, LH), Gl, Gs2, lane);  // no activation on mlp_semantics' last layer
    dw_accumulate<4, 4>(scr, Gs2, s1, accB, bsB, lane);
    f32x4 Gs1[4];
    mlp_layer_T<4, 4, 0, 4>(R::w(lds, LS1), Gs2, Gs1, lane);
    relu_mask_(Gs1, s1);
    dw_accumulate<4, 1>(scr, Gs1, h, accA, bsA, lane);   // input = detached geo: no dX
  }

  const int lane = lane0;
  __syncthreads();
  float* acc_lds = R::w(lds, LS0);
  constexpr int ACC_FLOATS = Cfg::woff(LH + 1) - Cfg::woff(LS0);
  static_assert(LS1 == LS0 + 1 && LH == LS1 + 1, "the branch's layers are adjacent in the image");
  for (int i = threadIdx.x; i < ACC_FLOATS; i += blockDim.x) acc_lds[i] = 0.0f;
  __syncthreads();
  for (int turn = 0; turn < WAVES; ++turn) {
    if (wave == turn) {
      flush_dw<4, 1>(R::w(lds, LS0), accA, lane);
      flush_dw<4, 4>(R::w(lds, LS1), accB, lane);
      flush_dw<1, 4>(R::w(lds, LH), accC, lane);
      lds_bias[lane] += bsA;
      lds_bias[64 + lane] += bsB;
      if (lane < 16) lds_bias[128 + lane] += bsC;
    }
    __syncthreads();
  }
  float* part = partials + (size_t)blockIdx.x * (Cfg::W_TOTAL + Cfg::B_TOTAL);
  for (int i = threadIdx.x; i < ACC_FLOATS; i += blockDim.x) part[Cfg::woff(LS0) + i] = acc_lds[i];
  for (int i = threadIdx.x; i < 144; i += blockDim.x) part[Cfg::W_TOTAL + Cfg::boff(LS0) + i] = lds_bias[i];
}

// ---- semantic branch, `fruit_nerf_big` shape: 30 -> 128 -> 128 -> 64 -> head, two launches (see the file header) ----
// PHASE_A: dW/db of sem0, sem2 and the head;  PHASE_B: dW/db of sem1 (the 128 x 128 layer).
template <class Cfg, bool PHASE_A, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_field_mlp_bwd_sem_big(
    const float* __restrict__ packed, long long N, const float* __restrict__ h_saved,
    const float* __restrict__ d_logit, float* __restrict__ partials) {
  static_assert(Cfg::NSEM == 3 && Cfg::HB == 2 && Cfg::SEMB == 8, "fruit_nerf_big semantic shape");
  using R = typename BwdRange<Cfg, PHASE_A ? BR_SEM_A : BR_SEM_B>::type;
  constexpr int LS0 = Cfg::L_SEM0, LS1 = Cfg::L_SEM1, LS2 = Cfg::L_SEM2, LH = Cfg::L_HEAD;
  constexpr int NBIAS = PHASE_A ? 128 + 64 + 16 : 128;
  __shared__ __attribute__((aligned(16))) float lds[R::FLOATS + WAVES * SCR_FLOATS + NBIAS];
  float* scr_all = lds + R::FLOATS;
  float* lds_bias = scr_all + WAVES * SCR_FLOATS;
  R::template stage<64 * WAVES>(lds, packed);
  for (int i = threadIdx.x; i < NBIAS; i += blockDim.x) lds_bias[i] = 0.0f;
  __syncthreads();
  const int lane0 = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* scr = scr_all + wave * SCR_FLOATS;

  // PHASE_A: acc0 = sem0 [8][2], acc2 = sem2 [4][8], accH = head [1][4];  PHASE_B: acc1 = sem1 [8][8]
  f32x4 acc0[PHASE_A ? 8 : 1][PHASE_A ? 2 : 1];
  f32x4 acc2[PHASE_A ? 4 : 1][PHASE_A ? 8 : 1];
  f32x4 accH[1][PHASE_A ? 4 : 1];
  f32x4 acc1[PHASE_A ? 1 : 8][PHASE_A ? 1 : 8];
  float bs_lo = 0.0f, bs_hi = 0.0f;  // rows 0..63 / 64..127 of the 128-wide layer this phase owns (sem0 | sem1)
  float bs2 = 0.0f, bsH = 0.0f;
  zero_acc(acc0);
  zero_acc(acc1);
  zero_acc(acc2);
  zero_acc(accH);

  const long long n_tiles = (N + 15) / 16;
  for (long long tile = (long long)blockIdx.x * WAVES + wave; tile < n_tiles; tile += (long long)gridDim.x * WAVES) {
    asm volatile("" ::: "memory");
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int j = lane & 15, g = lane >> 4;
    const long long n = tile * 16 + j;
    const bool valid = n < N;
    const long long nn = valid ? n : N - 1;
    f32x4 h[2];
    h[0] = *reinterpret_cast<const f32x4*>(h_saved + (size_t)nn * 32 + 4 * g);
    h[1] = *reinterpret_cast<const f32x4*>(h_saved + (size_t)nn * 32 + 16 + 4 * g);
    f32x4 s1[8], s2[8];
    mlp_layer<8, 2>(R::w(lds, LS0), R::b(lds, LS0), h, s1, lane);
    relu_(s1);
    mlp_layer<8, 8>(R::w(lds, LS1), R::b(lds, LS1), s1, s2, lane);
    relu_(s2);
    f32x4 Gl[1];
    Gl[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (g == 0 && valid) Gl[0][0] = d_logit[n];
    f32x4 Gs3[4];
    mlp_layer_T<1, 4, 0, 4>(R::w(lds, LH), Gl, Gs3, lane);  // no activation on mlp_semantics' last layer
    if constexpr (PHASE_A) {
      f32x4 s3[4];
      mlp_layer<4, 8>(R::w(lds, LS2), R::b(lds, LS2), s2, s3, lane);
      dw_accumulate<1, 4>(scr, Gl, s3, accH, bsH, lane);  // SemanticFieldHead
      dw_accumulate_sub<4, 4, 4, 8, 0, 0, true>(scr, Gs3, s2, acc2, bs2, lane);
      dw_accumulate_sub<4, 4, 4, 8, 0, 4, false>(scr, Gs3, s2 + 4, acc2, bs2, lane);
    }
    f32x4 Gs2[8];
    mlp_layer_T<4, 8, 0, 8>(R::w(lds, LS2), Gs3, Gs2, lane);
    relu_mask_(Gs2, s2);
    if constexpr (PHASE_A) {
      f32x4 Gs1[8];
      mlp_layer_T<8, 8, 0, 8>(R::w(lds, LS1), Gs2, Gs1, lane);
      relu_mask_(Gs1, s1);
      dw_accumulate_sub<4, 2, 8, 2, 0, 0, true>(scr, Gs1, h, acc0, bs_lo, lane);  // input = detached geo: no dX
      dw_accumulate_sub<4, 2, 8, 2, 4, 0, true>(scr, Gs1 + 4, h, acc0, bs_hi, lane);
    } else {
      dw_accumulate_sub<4, 4, 8, 8, 0, 0, true>(scr, Gs2, s1, acc1, bs_lo, lane);
      dw_accumulate_sub<4, 4, 8, 8, 0, 4, false>(scr, Gs2, s1 + 4, acc1, bs_lo, lane);
      dw_accumulate_sub<4, 4, 8, 8, 4, 0, true>(scr, Gs2 + 4, s1, acc1, bs_hi, lane);
      dw_accumulate_sub<4, 4, 8, 8, 4, 4, false>(scr, Gs2 + 4, s1 + 4, acc1, bs_hi, lane);
    }
  }

  const int lane = lane0;
  __syncthreads();
  float* part = partials + (size_t)blockIdx.x * (Cfg::W_TOTAL + Cfg::B_TOTAL);
  if constexpr (PHASE_A) {
    for (int i = threadIdx.x; i < Cfg::nob(LS0) * Cfg::nib(LS0) * 256; i += blockDim.x) R::w(lds, LS0)[i] = 0.0f;
    for (int i = threadIdx.x; i < Cfg::woff(LH + 1) - Cfg::woff(LS2); i += blockDim.x) R::w(lds, LS2)[i] = 0.0f;
    static_assert(LH == LS2 + 1, "sem2 and the head are adjacent in the image");
    __syncthreads();
    for (int turn = 0; turn < WAVES; ++turn) {
      if (wave == turn) {
        flush_dw<8, 2>(R::w(lds, LS0), acc0, lane);
        flush_dw<4, 8>(R::w(lds, LS2), acc2, lane);
        flush_dw<1, 4>(R::w(lds, LH), accH, lane);
        lds_bias[lane] += bs_lo;
        lds_bias[64 + lane] += bs_hi;
        lds_bias[128 + lane] += bs2;
        if (lane < 16) lds_bias[192 + lane] += bsH;
      }
      __syncthreads();
    }
    for (int i = threadIdx.x; i < Cfg::nob(LS0) * Cfg::nib(LS0) * 256; i += blockDim.x)
      part[Cfg::woff(LS0) + i] = R::w(lds, LS0)[i];
    for (int i = threadIdx.x; i < Cfg::woff(LH + 1) - Cfg::woff(LS2); i += blockDim.x)
      part[Cfg::woff(LS2) + i] = R::w(lds, LS2)[i];
    for (int i = threadIdx.x; i < 128; i += blockDim.x) part[Cfg::W_TOTAL + Cfg::boff(LS0) + i] = lds_bias[i];
    for (int i = threadIdx.x; i < 64; i += blockDim.x) part[Cfg::W_TOTAL + Cfg::boff(LS2) + i] = lds_bias[128 + i];
    for (int i = threadIdx.x; i < 16; i += blockDim.x) part[Cfg::W_TOTAL + Cfg::boff(LH) + i] = lds_bias[192 + i];
  } else {
    for (int i = threadIdx.x; i < Cfg::nob(LS1) * Cfg::nib(LS1) * 256; i += blockDim.x) R::w(lds, LS1)[i] = 0.0f;
    __syncthreads();
    for (int turn = 0; turn < WAVES; ++turn) {
      if (wave == turn) {
        flush_dw<8, 8>(R::w(lds, LS1), acc1, lane);
        lds_bias[lane] += bs_lo;
        lds_bias[64 + lane] += bs_hi;
      }
      __syncthreads();
    }
    for (int i = threadIdx.x; i < Cfg::nob(LS1) * Cfg::nib(LS1) * 256; i += blockDim.x)
      part[Cfg::woff(LS1) + i] = R::w(lds, LS1)[i];
    for (int i = threadIdx.x; i < 128; i += blockDim.x) part[Cfg::W_TOTAL + Cfg::boff(LS1) + i] = lds_bias[i];
  }
}

// sum the per-workgroup partial images and add them into the nn.Linear-layout gradients.  FIXED summation order and a
// single writer per gradient entry (no float atomics: training is bit-reproducible run to run): a workgroup is
// RD_IDX entries x RD_Y slices; slice y sums the images y, y + RD_Y, ... (8 independent loads in flight), the slices
// meet in LDS and are added up in slice order by the thread of slice 0.
// ADAM (single-process training, fnr_field_mlp_bwd_adam): the thread that owns a gradient entry also takes that
// parameter's optimiser step (torch.optim.Adam / RAdam exactly as fnr_adam_step / fnr_radam_step: same operations in the
// same order) and leaves the gradient entry zero — the ~20 k weights of the field's MLPs no longer need a launch of
// their own, and a step that does not train the proposal networks ends without any optimiser launch.
constexpr int RD_IDX = 128, RD_Y = 8;
template <class Cfg, bool ADAM>
__device__ __forceinline__ void reduce_dw_block(int block, const float* __restrict__ partials, int nblocks, const FieldPtrs& grads,
                                                const WeightAdam& wa) {
  __shared__ float s_part[RD_Y][RD_IDX];
  const int t = threadIdx.x % RD_IDX, y = threadIdx.x / RD_IDX;
  const int idx = block * RD_IDX + t;
  constexpr int TOT = Cfg::W_TOTAL + Cfg::B_TOTAL;
  float s = 0.0f;
  if (idx < TOT) {
    int b = y;
    for (; b + 7 * RD_Y < nblocks; b += 8 * RD_Y) {  // 8 independent loads in flight (a plain loop is one latency per row)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partials[(size_t)(b + u * RD_Y) * TOT + idx];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < nblocks; b += RD_Y) s += partials[(size_t)b * TOT + idx];
  }
  s_part[y][t] = s;
  __syncthreads();
  if (y != 0 || idx >= TOT) return;
#pragma unroll
  for (int q = 1; q < RD_Y; ++q) s += s_part[q][t];
  if (!ADAM && s == 0.0f) return;
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
      if constexpr (ADAM) weight_adam_entry(wa, dst, s);
      else *dst += s;  // sole writer of this entry
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
      if constexpr (ADAM) weight_adam_entry(wa, dst, s);
      else *dst += s;
    }
  }
}

template <class Cfg, bool ADAM>
__global__ __launch_bounds__(RD_IDX * RD_Y) void k_reduce_dw(const float* __restrict__ partials, int nblocks, FieldPtrs grads,
                                                             WeightAdam wa) {
  reduce_dw_block<Cfg, ADAM>((int)blockIdx.x, partials, nblocks, grads, wa);
}

// Per-ray finish of mlp_head layer 0 (see the colour branch above).  For every ray: g = sum of its tiles' G1 row
// sums (+ the per-sample contributions of tiles that straddle rays), written to g_ray [n_rays, 64] for the
// embedding gradient; c = [SH16(direction) | Embedding[camera]].  The workgroup accumulates g (x) c (64 x 48) and
// sum g (bias) over its 16 rays and stores them into ITS partial weight-gradient image (col0: input blocks
// HB..HB+2 and the bias, which the colour kernel left zero), so k_reduce_dw adds them like any other partial.
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
  constexpr int NIB0 = Cfg::HB + 3;
#pragma unroll
  for (int q = 0; q < 12; ++q) {
    const int k = 12 * kq + q, ib = Cfg::HB + (k >> 4), kk = k & 15;
    part[Cfg::woff(Cfg::L_COL0) + ((ob * NIB0 + ib) * 64 + swz_slot(i, kk >> 2)) * 4 + (kk & 3)] = acc[q >> 2][q & 3];
  }
  if (kq == 0) part[Cfg::W_TOTAL + Cfg::boff(Cfg::L_COL0) + o] = accb;
}
static_assert(RAYG_RB * COLOR_CONST_K % 256 == 0, "staging loop covers the batch exactly");

// appearance-embedding gradient (fruit_field.py:251 Embedding lookup): one workgroup per camera gathers the g rows
// of its rays (each wave tests 64 rays per ballot), then g_embedding[c][k] += sum_o W[o][emb col k] * gcam[o].
// No atomics: direct adds into the [n_images, 32] table serialise at ~12 ns per same-address add (393k adds on
// 90 rows made the colour branch 4x slower than its MFMA time).
template <class Cfg, bool ADAM>
__device__ __forceinline__ void embedding_grad_block(int c, const RaysDev& rays, const float* __restrict__ g_ray,
                                                     const float* __restrict__ packed, float* __restrict__ g_embedding,
                                                     const WeightAdam& wa) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
  if constexpr (ADAM) {
    if (oo == 0) weight_adam_entry(wa, g_embedding + (size_t)c * 32 + k, s);  // every row takes its step (moment decay)
  } else {
    if (oo == 0 && s != 0.0f) g_embedding[(size_t)c * 32 + k] += s;
  }
}

template <class Cfg, bool ADAM>
__global__ __launch_bounds__(1024) void k_embedding_grad(RaysDev rays, const float* __restrict__ g_ray,
                                                         const float* __restrict__ packed,
                                                         float* __restrict__ g_embedding, WeightAdam wa) {
  embedding_grad_block<Cfg, ADAM>((int)blockIdx.x, rays, g_ray, packed, g_embedding, wa);
}

// The two finishing launches of the MLP backward as ONE (round 6): workgroups [0, n_images) take the appearance embedding's
// rows (k_embedding_grad), the rest the weight-gradient images (k_reduce_dw) — both only need the colour kernel's per-ray
// sums (k_color_ray_grads) and every branch's partial images, neither reads what the other writes, both are 1024 threads.
// One launch ramp less on the launch stream's chain and the two roles next to each other instead of one after the other
// (7.4 + 9.7 us -> ~10); the same sums in the same order by the same single writers.
static_assert(RD_IDX * RD_Y == 1024, "k_finish_weights: both roles are 1024-thread workgroups");
template <class Cfg, bool ADAM>
__global__ __launch_bounds__(1024) void k_finish_weights(int n_images, RaysDev rays, const float* __restrict__ g_ray,
                                                         const float* __restrict__ packed, float* __restrict__ g_embedding,
                                                         const float* __restrict__ partials, int nblocks, FieldPtrs grads,
                                                         WeightAdam wa) {
  if ((int)blockIdx.x < n_images) embedding_grad_block<Cfg, ADAM>((int)blockIdx.x, rays, g_ray, packed, g_embedding, wa);
  else reduce_dw_block<Cfg, ADAM>((int)blockIdx.x - n_images, partials, nblocks, grads, wa);
}

int field_ptrs(const fnr_field_net* net, FieldPtrs& p, int* cfg_id);  // field_mlp.hip
size_t field_fwd_ws_image_offset();                                     // field_mlp.hip
size_t field_bf16_image_bytes();                                        // field_mlp_bf16.hip
int field_mlp_bwd_bf16(int cfg, int mode, int branch, const FieldPtrs& p, bool pack, const float* packed, void* image_ws,
                       const float* ray_bias, const RaysDev& rd, int S, long long N, const float2* feats,
                       const float* h_saved, const uint8_t* selector, const float* d_density, const float* d_rgb,
                       const float* d_logit, float2* d_feats, float* d_h, float* gsum_tile, float* gsum_extra,
                       float* partials, long long blocks, hipStream_t st, const float2* jac = nullptr,
                       float4* d_pos = nullptr);
int position_contract(long long N, int n_levels, const float2* jac, const float2* d_feats, float4* d_pos, hipStream_t st);
int field_mlp_bwd_sem_big_bf16(int mode, const FieldPtrs& p, void* image_ws, const float* packed, long long N,
                               const float* h_saved, const float* d_logit, float* partials, long long blocks,
                               hipStream_t st);

}  // namespace fnr

using namespace fnr;

namespace {
struct BwdWorkspace {
  float *partials, *d_h, *packed, *ray_bias, *gsum_tile, *g_ray, *gsum_extra;
  void* bf16_image;
  size_t bytes;
};
// carve the workspace: per-workgroup partial weight-gradient images (<= one workgroup per CU), dL/dh [N, 16 HB], the
// fragment image, and the per-ray colour terms — sized for the larger of the two built shapes
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
  w.partials = take((size_t)device_cu_count() * FIELD_MAX_IMAGE_FLOATS);
  w.d_h = take((size_t)N * 16 * FIELD_MAX_HB);
  w.packed = take(FIELD_MAX_PACKED_FLOATS);
  w.ray_bias = take((size_t)n_rays * 64);
  w.gsum_tile = take((size_t)n_tiles * 64);
  w.g_ray = take((size_t)n_rays * 64);
  w.gsum_extra = take((size_t)n_rays * 64);
  w.bf16_image = take((field_bf16_image_bytes() + 3) / 4);
  w.bytes = p - reinterpret_cast<uintptr_t>(base) + 256;
  return w;
}

template <class Cfg>
int field_mlp_bwd_launch(const FieldPtrs& p, const FieldPtrs& gp, const fnr_field_net* net, const fnr_field_net* grads,
                         const RaysDev& rd, int S, long long N, const float* feats, const float* h_saved,
                         const float* ray_bias_saved, const float* packed_saved, const uint8_t* selector,
                         const float* d_density, const float* d_rgb, const float* d_logit, float* d_feats,
                         const BwdWorkspace& ws, hipStream_t st, const float* jacobian = nullptr,
                         float* d_position = nullptr, const WeightAdam* wadam = nullptr) {
  const float2* jac = reinterpret_cast<const float2*>(jacobian);
  float4* d_pos = reinterpret_cast<float4*>(d_position);
  const long long n_tiles = (N + 15) / 16;
  const long long max_blocks = device_cu_count();
  float* partials = ws.partials;
  float* packed = ws.packed;
  float* gsum_extra = (S % 16 != 0) ? ws.gsum_extra : nullptr;  // only tiles that straddle rays use it
  const float2* f2 = reinterpret_cast<const float2*>(feats);
  float2* df2 = reinterpret_cast<float2*>(d_feats);
  static const int color_waves = [] {
    const char* e = getenv("FNR_COLOR_WAVES");
    return (e && atoi(e) == 4) ? 4 : 8;
  }();
  void* bf16_image = ws.bf16_image;
  const bool bf16_pack = !packed_saved;
  if (packed_saved) {
    packed = const_cast<float*>(packed_saved);  // the forward pass's fragment image of the same weights
    bf16_image = reinterpret_cast<char*>(packed) + field_fwd_ws_image_offset();  // ... and its bf16 pieces
  } else {
    launch_pack_field_weights<Cfg>(p, packed, st);
    FNR_LAUNCH_CHECK();
  }
  const float* ray_bias = ray_bias_saved;
  if (!ray_bias) {
    launch_color_ray_bias<Cfg>(packed, rd, net->embedding, nullptr, ws.ray_bias, st);
    FNR_LAUNCH_CHECK();
    ray_bias = ws.ray_bias;
  }
  if (gsum_extra) FNR_HIP(hipMemsetAsync(gsum_extra, 0, (size_t)rd.n_rays * 64 * sizeof(float), st));
  // every branch uses the same number of workgroups so that they share one partial-image buffer
  long long blocks = (n_tiles + 3) / 4;
  if (blocks > max_blocks) blocks = max_blocks;
  const dim3 grid((unsigned)blocks);
  const int mode = net->mlp_mode;
  // bf16-pipe modes: every branch of both shapes (field_mlp_bf16.hip: cooperative dW; the `fruit_nerf_big` semantic
  // branch additionally weight-streamed)
  constexpr int cfg_id = Cfg::NSEM == 2 ? 0 : 1;
  const bool bf_all = mode != FNR_MLP_FP32;
  const bool bf_sem_big = mode != FNR_MLP_FP32 && Cfg::NSEM == 3;
  if (bf_all) {
    const int rc = field_mlp_bwd_bf16(cfg_id, mode, 0, p, bf16_pack, packed, bf16_image, ray_bias, rd, S, N, f2, h_saved, selector,
                                      d_density, d_rgb, d_logit, df2, ws.d_h, ws.gsum_tile, gsum_extra, partials, blocks, st);
    if (rc) return rc;
  } else if (color_waves == 4) {
    hipLaunchKernelGGL((k_field_mlp_bwd_color<Cfg, 4>), grid, dim3(256), 0, st, packed, ray_bias, rd, S, N, h_saved, d_rgb,
                       ws.d_h, ws.gsum_tile, gsum_extra, partials);
  } else {
    hipLaunchKernelGGL((k_field_mlp_bwd_color<Cfg, 8>), grid, dim3(512), 0, st, packed, ray_bias, rd, S, N, h_saved, d_rgb,
                       ws.d_h, ws.gsum_tile, gsum_extra, partials);
  }
  FNR_LAUNCH_CHECK();
  {
    // per-ray finish of mlp_head layer 0: every workgroup owns one partial image row that exists
    long long rb = (rd.n_rays + RAYG_RB - 1) / RAYG_RB;
    if (rb > blocks) rb = blocks;
    hipLaunchKernelGGL((k_color_ray_grads<Cfg>), dim3((unsigned)rb), dim3(256), 0, st, rd, S, N, net->embedding,
                       ws.gsum_tile, gsum_extra, ws.g_ray, partials);
    FNR_LAUNCH_CHECK();
    // (the appearance embedding's rows, which only need g_ray, are taken by the LAST launch: k_finish_weights)
  }
  if (bf_sem_big) {
    int rc = field_mlp_bwd_sem_big_bf16(mode, p, bf16_image, packed, N, h_saved, d_logit, partials, blocks, st);
    if (rc) return rc;
    rc = field_mlp_bwd_bf16(cfg_id, mode, 2, p, false, packed, bf16_image, ray_bias, rd, S, N, f2, h_saved, selector, d_density,
                            d_rgb, d_logit, df2, ws.d_h, ws.gsum_tile, gsum_extra, partials, blocks, st, jac, d_pos);
    if (rc) return rc;
  } else if (bf_all) {
    for (int branch = 1; branch <= 2; ++branch) {
      const int rc = field_mlp_bwd_bf16(cfg_id, mode, branch, p, false, packed, bf16_image, ray_bias, rd, S, N, f2, h_saved, selector,
                                        d_density, d_rgb, d_logit, df2, ws.d_h, ws.gsum_tile, gsum_extra, partials, blocks, st,
                                        branch == 2 ? jac : nullptr, branch == 2 ? d_pos : nullptr);
      if (rc) return rc;
    }
  } else if constexpr (Cfg::NSEM == 2) {
    hipLaunchKernelGGL((k_field_mlp_bwd_sem<Cfg, 8>), grid, dim3(512), 0, st, packed, N, h_saved, d_logit, partials);
    FNR_LAUNCH_CHECK();
  } else {
    hipLaunchKernelGGL((k_field_mlp_bwd_sem_big<Cfg, true, 4>), grid, dim3(256), 0, st, packed, N, h_saved, d_logit,
                       partials);
    FNR_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_field_mlp_bwd_sem_big<Cfg, false, 4>), grid, dim3(256), 0, st, packed, N, h_saved, d_logit,
                       partials);
    FNR_LAUNCH_CHECK();
  }
  if (!bf_all) {
    hipLaunchKernelGGL((k_field_mlp_bwd_base<Cfg, 8>), grid, dim3(512), 0, st, packed, N, f2, selector, d_density, ws.d_h,
                       df2, partials);
    FNR_LAUNCH_CHECK();
  }
  constexpr int TOT = Cfg::W_TOTAL + Cfg::B_TOTAL;
  const unsigned finish_blocks = (unsigned)net->n_images + (unsigned)((TOT + RD_IDX - 1) / RD_IDX);
  if (wadam)
    hipLaunchKernelGGL((k_finish_weights<Cfg, true>), dim3(finish_blocks), dim3(1024), 0, st, (int)net->n_images, rd, ws.g_ray,
                       packed, grads->embedding, partials, (int)blocks, gp, *wadam);
  else
    hipLaunchKernelGGL((k_finish_weights<Cfg, false>), dim3(finish_blocks), dim3(1024), 0, st, (int)net->n_images, rd,
                       ws.g_ray, packed, grads->embedding, partials, (int)blocks, gp, WeightAdam{});
  FNR_LAUNCH_CHECK();
  // fp32 chains: their base-branch kernel does not carry the contraction with the encode's Jacobian
  if (jac && d_pos && !bf_all) return position_contract(N, net->grid.n_levels, jac, df2, d_pos, st);
  return FNR_OK;
}
}  // namespace

extern "C" size_t fnr_field_mlp_bwd_workspace_bytes(int64_t n_rays, int S) {
  if (n_rays < 0 || S <= 0) return 0;
  return bwd_workspace(nullptr, n_rays, S).bytes;
}

static int field_mlp_bwd_entry(const fnr_field_net* net, const fnr_field_net* grads, const fnr_rays* rays, int S,
                                 const float* feats, const float* h_saved, const float* ray_bias_saved,
                                 const float* packed_saved, const uint8_t* selector, const float* d_density, const float* d_rgb, const float* d_logit, float* d_feats,
                                 void* workspace, size_t workspace_bytes, void* stream, const float* jacobian,
                               float* d_position, const fnr_table_adam* weight_adam = nullptr,
                               const float* grad_arena = nullptr) {
  WeightAdam wa{};
  if (weight_adam) {
    FNR_CHECK_ARG(grad_arena, "field_mlp_bwd_adam: grad_arena missing");
    const int rca = make_table_adam(weight_adam, wa.t);
    if (rca) return rca;
    wa.p_off = weight_adam->params - grad_arena;
    wa.m_off = weight_adam->exp_avg - grad_arena;
    wa.v_off = weight_adam->exp_avg_sq - grad_arena;
  }
  FNR_CHECK_ARG(net && grads && rays && feats && h_saved && d_density && d_rgb && d_logit && d_feats && workspace && S > 0,
                "field_mlp_bwd: null argument");
  FNR_CHECK_ARG(rays->directions && rays->camera_indices && net->embedding && grads->embedding,
                "field_mlp_bwd: training path needs directions, camera indices and the embedding (+ its gradient)");
  FieldPtrs p, gp;
  int cfg = 0, gcfg = 0;
  int rc = field_ptrs(net, p, &cfg);
  if (rc) return rc;
  rc = field_ptrs(grads, gp, &gcfg);
  if (rc) return rc;
  FNR_CHECK_ARG(cfg == gcfg, "field_mlp_bwd: net and grads describe different field shapes");
  FNR_CHECK_ARG(net->mlp_mode == FNR_MLP_FP32 || net->mlp_mode == FNR_MLP_BF16 || net->mlp_mode == FNR_MLP_BF16X3,
                "field_mlp_bwd: mlp_mode %d (FNR_MLP_FP32 0 | FNR_MLP_BF16 1 | FNR_MLP_BF16X3 3)", net->mlp_mode);
  const long long N = rays->n_rays * (long long)S;
  if (N == 0) return FNR_OK;
  FNR_CHECK_ARG(workspace_bytes >= fnr_field_mlp_bwd_workspace_bytes(rays->n_rays, S),
                "field_mlp_bwd: workspace too small");
  const BwdWorkspace ws = bwd_workspace(workspace, rays->n_rays, S);
  const RaysDev rd = make_rays(rays);
  FNR_PROF(OP_MLP_BWD, N);
  if (cfg == 0)
    return field_mlp_bwd_launch<FieldCfgBase>(p, gp, net, grads, rd, S, N, feats, h_saved, ray_bias_saved, packed_saved,
                                              selector, d_density, d_rgb, d_logit, d_feats, ws, as_stream(stream), jacobian, d_position,
                                              weight_adam ? &wa : nullptr);
  return field_mlp_bwd_launch<FieldCfgBig>(p, gp, net, grads, rd, S, N, feats, h_saved, ray_bias_saved, packed_saved,
                                           selector, d_density, d_rgb, d_logit, d_feats, ws, as_stream(stream), jacobian, d_position,
                                              weight_adam ? &wa : nullptr);
}

extern "C" int fnr_field_mlp_bwd(const fnr_field_net* net, const fnr_field_net* grads, const fnr_rays* rays, int S,
                                 const float* feats, const float* h_saved, const float* ray_bias_saved,
                                 const float* packed_saved, const uint8_t* selector, const float* d_density,
                                 const float* d_rgb, const float* d_logit, float* d_feats, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_field_mlp_bwd");
  return field_mlp_bwd_entry(net, grads, rays, S, feats, h_saved, ray_bias_saved, packed_saved, selector, d_density, d_rgb,
                             d_logit, d_feats, workspace, workspace_bytes, stream, nullptr, nullptr);
}

extern "C" int fnr_field_mlp_bwd_adam(const fnr_field_net* net, const fnr_field_net* grads, const fnr_rays* rays, int S,
                                      const float* feats, const float* h_saved, const float* ray_bias_saved,
                                      const float* packed_saved, const uint8_t* selector, const float* d_density,
                                      const float* d_rgb, const float* d_logit, float* d_feats, const float* jacobian,
                                      float* d_position, const fnr_table_adam* weight_adam, const float* grad_arena,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  if (seq::recording() && net && grads && rays && weight_adam) {
    const fnr_field_net net_ = *net, grads_ = *grads;
    const fnr_rays rays_ = *rays;
    const fnr_table_adam adam_ = *weight_adam;
    seq::push("fnr_field_mlp_bwd_adam", [=](const fnr_step_scalars* sc) {
      const fnr_table_adam a = seq::patched(adam_, sc);
      return fnr_field_mlp_bwd_adam(&net_, &grads_, &rays_, S, feats, h_saved, ray_bias_saved, packed_saved, selector, d_density,
                                    d_rgb, d_logit, d_feats, jacobian, d_position, &a, grad_arena, workspace, workspace_bytes,
                                    stream);
    });
  }
  FNR_CHECK_ARG(weight_adam && grad_arena, "field_mlp_bwd_adam: weight_adam / grad_arena missing");
  FNR_CHECK_ARG((jacobian == nullptr) == (d_position == nullptr), "field_mlp_bwd_adam: jacobian and d_position go together");
  return field_mlp_bwd_entry(net, grads, rays, S, feats, h_saved, ray_bias_saved, packed_saved, selector, d_density, d_rgb,
                             d_logit, d_feats, workspace, workspace_bytes, stream, jacobian, d_position, weight_adam,
                             grad_arena);
}

extern "C" int fnr_field_mlp_bwd_rays(const fnr_field_net* net, const fnr_field_net* grads, const fnr_rays* rays, int S,
                                      const float* feats, const float* h_saved, const float* ray_bias_saved,
                                      const float* packed_saved, const uint8_t* selector, const float* d_density,
                                      const float* d_rgb, const float* d_logit, float* d_feats, const float* jacobian,
                                      float* d_position, void* workspace, size_t workspace_bytes, void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_field_mlp_bwd_rays");
  FNR_CHECK_ARG(jacobian && d_position, "field_mlp_bwd_rays: jacobian / d_position missing");
  return field_mlp_bwd_entry(net, grads, rays, S, feats, h_saved, ray_bias_saved, packed_saved, selector, d_density, d_rgb,
                             d_logit, d_feats, workspace, workspace_bytes, stream, jacobian, d_position);
}
