// field_mlp_bf16.hip — FruitField's MLP stack, forward and backward, on the bf16 matrix pipe of gfx950
// (v_mfma_f32_16x16x32_bf16).  Two modes of fnr_field_net.mlp_mode, see field_bf16.hpp:
//   FNR_MLP_BF16X3 (3)  exact 3-way bf16 split of every fp32 operand (FruitField's default): 6 piece products per
//                       product in the forward pass and in the backward's forward recompute, 3 in dX / dW — fp32-grade
//                       results at 16/6 resp. 16/3 of the fp32-MFMA issue rate;
//   FNR_MLP_BF16   (1)  plain bf16 operands, fp32 accumulate — throughput mode, not parity grade.
// Same entry points, buffers and partial-gradient image as the fp32 kernels (field_mlp.hip, field_mlp_bwd.hip):
// the per-ray colour bias, the per-ray / per-camera finish of mlp_head layer 0 and k_reduce_dw are shared.
// Replaces the same reference code: fruit_field.py:132-166,187-281 and its autograd.
// Forward (`fruit_nerf`): one wave = one PAIR of 16-sample tiles sharing every LDS fragment.  Backward (both shapes)
// and the `fruit_nerf_big` semantic branch: one tile per wave, 8 waves = a 128-sample batch, COOPERATIVE dW (every
// 16 x 16 block of a layer's weight gradient owned by one wave over the batch), weight streaming where the fragments
// exceed the LDS.
#include <stdlib.h>

#include <type_traits>

#include "field_bf16.hpp"

namespace fnr {

// ---- segment lists ------------------------------------------------------------------------------------------------
template <class Cfg>
struct SegsFwdAll {  // every layer, forward only
  static constexpr int N = Cfg::NLAYERS;
  static constexpr int layer(int i) { return i; }
  static constexpr bool isT(int) { return false; }
};

template <class Cfg>
struct SegsFwdBaseColor {  // base + colour MLPs, forward only (`fruit_nerf_big`: its semantic branch is weight-streamed)
  static constexpr int N = 5;
  static constexpr int layer(int i) {
    return i == 0 ? Cfg::L_BASE0 : i == 1 ? Cfg::L_BASE1 : i == 2 ? Cfg::L_COL0 : i == 3 ? Cfg::L_COL1 : Cfg::L_COL2;
  }
  static constexpr bool isT(int) { return false; }
};

// the 16 x 2 hash features of sample nn as the single K-block of mlp_base layer 0 (slot (g, e): level 4 (e>>1) + g,
// feature e & 1 — the KM_HASH column map of field_layers.hpp with s = e)
__device__ __forceinline__ void load_hash_block(const float2* __restrict__ feats, long long N, long long nn, int g,
                                                f32x4 (&x0)[2]) {
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const float2 v = feats[(size_t)(4 * m + g) * N + nn];
    x0[m >> 1][2 * (m & 1)] = v.x;
    x0[m >> 1][2 * (m & 1) + 1] = v.y;
  }
}

template <int N>
__device__ __forceinline__ void relu2_(f32x4 (&a)[N], f32x4 (&b)[N]) {
  relu_(a);
  relu_(b);
}

// =====================================================================================================================
// forward
// =====================================================================================================================
// WITH_SEM: the `fruit_nerf` shape, every branch in this kernel.  !WITH_SEM: base + colour only (`fruit_nerf_big`, whose
// 128-wide semantic branch runs in k_field_mlp_fwd_sem_big_bf16 from the h this kernel saves).
template <class Cfg, int NS, int WAVES, bool WITH_SEM>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void k_field_mlp_fwd_bf16(
    const float* __restrict__ packed, const __bf16* __restrict__ image, const float* __restrict__ ray_bias, RaysDev rays,
    int S, long long N, const float2* __restrict__ feats, const uint8_t* __restrict__ selector,
    float* __restrict__ density, float* __restrict__ rgb, float* __restrict__ logit, float* __restrict__ geo_out,
    float* __restrict__ h_buf) {
  constexpr int HB = Cfg::HB;
  static_assert(HB == 1 || HB == 2, "built shapes");
  static_assert(!WITH_SEM || (HB == 1 && Cfg::NSEM == 2), "all-in-one kernel: `fruit_nerf` shape");
  using Segs = typename std::conditional<WITH_SEM, SegsFwdAll<Cfg>, SegsFwdBaseColor<Cfg>>::type;
  using Lds = BfLds<Cfg, Segs, NS>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16x8* lds = reinterpret_cast<bf16x8*>(smem);
  float* bias = reinterpret_cast<float*>(smem + Lds::BYTES);  // Cfg::B_TOTAL floats, the fp32 image's bias block
  Lds::template stage<64 * WAVES>(lds, image);
  for (int i = threadIdx.x; i < Cfg::B_TOTAL; i += blockDim.x) bias[i] = packed[Cfg::W_TOTAL + i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const long long n_pairs = (N + 31) / 32;
  for (long long pr = (long long)blockIdx.x * WAVES + wave; pr < n_pairs; pr += (long long)gridDim.x * WAVES) {
    asm volatile("" ::: "memory");  // keep the LDS fragment reads inside the loop (field_mlp.hip)
    const long long na = pr * 32 + j, nb = na + 16;
    const bool va = na < N, vb = nb < N;
    const long long nna = va ? na : N - 1, nnb = vb ? nb : N - 1;
    const long long raya = nna / S, rayb = nnb / S;

    f32x4 ha[HB], hb[HB];
    {
      f32x4 xa[2], xb[2], a1a[4], a1b[4];
      load_hash_block(feats, N, nna, g, xa);
      load_hash_block(feats, N, nnb, g, xb);
      bf_layer<NS, 4, 2>(Lds::template seg<Cfg::L_BASE0, false>(lds), bias + Cfg::boff(Cfg::L_BASE0), xa, xb, a1a, a1b, lane);
      relu2_(a1a, a1b);
      bf_layer<NS, HB, 4>(Lds::template seg<Cfg::L_BASE1, false>(lds), bias + Cfg::boff(Cfg::L_BASE1), a1a, a1b, ha, hb, lane);
    }
    if constexpr (WITH_SEM) {  // semantic branch
      f32x4 s1a[4], s1b[4], s2a[4], s2b[4], hda[1], hdb[1];
      bf_layer<NS, 4, 1>(Lds::template seg<Cfg::L_SEM0, false>(lds), bias + Cfg::boff(Cfg::L_SEM0), ha, hb, s1a, s1b, lane);
      relu2_(s1a, s1b);
      bf_layer<NS, 4, 4>(Lds::template seg<Cfg::L_SEM1, false>(lds), bias + Cfg::boff(Cfg::L_SEM1), s1a, s1b, s2a, s2b, lane);
      bf_layer<NS, 1, 4>(Lds::template seg<Cfg::L_HEAD, false>(lds), bias + Cfg::boff(Cfg::L_HEAD), s2a, s2b, hda, hdb, lane);
      if (g == 0 && va) logit[na] = hda[0][0];
      if (g == 0 && vb) logit[nb] = hdb[0][0];
    }
    {  // colour branch; the ray-constant inputs of layer 0 arrive as the per-ray bias (fp32, k_color_ray_bias)
      f32x4 c1a[4], c1b[4], c2a[4], c2b[4], c3a[1], c3b[1];
#pragma unroll
      for (int ob = 0; ob < 4; ++ob) {
        c1a[ob] = *reinterpret_cast<const f32x4*>(ray_bias + (size_t)raya * 64 + 16 * ob + 4 * g);
        c1b[ob] = *reinterpret_cast<const f32x4*>(ray_bias + (size_t)rayb * 64 + 16 * ob + 4 * g);
      }
      {
        bf16x8 xa[1][NS], xb[1][NS];   // h: one K-block of 32 holds both 16-wide blocks of the big shape
        bf_operand<NS, HB>(ha, xa);
        bf_operand<NS, HB>(hb, xb);
        bf_layer_acc<NS, 4, 1>(Lds::template seg<Cfg::L_COL0, false>(lds), xa, xb, c1a, c1b, lane);
      }
      relu2_(c1a, c1b);
      bf_layer<NS, 4, 4>(Lds::template seg<Cfg::L_COL1, false>(lds), bias + Cfg::boff(Cfg::L_COL1), c1a, c1b, c2a, c2b, lane);
      relu2_(c2a, c2b);
      bf_layer<NS, 1, 4>(Lds::template seg<Cfg::L_COL2, false>(lds), bias + Cfg::boff(Cfg::L_COL2), c2a, c2b, c3a, c3b, lane);
      auto finish = [&](long long n, bool valid, const f32x4 (&h)[HB], const f32x4 (&c3)[1]) {
        if (!valid) return;
#pragma unroll
        for (int b = 0; b < HB; ++b) {
          if (h_buf) *reinterpret_cast<f32x4*>(h_buf + (size_t)n * (16 * HB) + 16 * b + 4 * g) = h[b];
          if (geo_out) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int k = 16 * b + 4 * g + r;
              if (k >= 1 && k <= Cfg::GEO) geo_out[(size_t)n * Cfg::GEO + (k - 1)] = h[b][r];
            }
          }
        }
        if (g == 0) {
          const bool sel = selector ? (selector[n] != 0) : true;
          density[n] = sel ? expf(h[0][0]) : 0.0f;  // trunc_exp forward * selector (fruit_field.py:191-192)
          rgb[3 * n + 0] = 1.0f / (1.0f + expf(-c3[0][0]));
          rgb[3 * n + 1] = 1.0f / (1.0f + expf(-c3[0][1]));
          rgb[3 * n + 2] = 1.0f / (1.0f + expf(-c3[0][2]));
        }
      };
      finish(na, va, ha, c3a);
      finish(nb, vb, hb, c3b);
    }
  }
}

// ---- helpers of the backward kernels ---------------------------------------------------------------------------------
// sum over the 16 lanes of a DPP row (= the 16 samples of a tile); every lane ends up with the total
__device__ __forceinline__ float row16_sum_bf(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
  return v;
}
template <int N>
__device__ __forceinline__ void zero_vec_bf(f32x4 (&a)[N]) {
#pragma unroll
  for (int b = 0; b < N; ++b) a[b] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// =====================================================================================================================
// `fruit_nerf_big` semantic branch, backward: mlp_semantics 30 -> 128 -> 128 -> 64 + SemanticFieldHead
// (fruit_field.py:144-156,263-268 with fruit_nerf_config.py:82-95).  Its 119 KB of fp32 weights, 464 accumulator
// registers of dW and 128-wide activations made the fp32 path two launches of 4 waves x 512 registers that each repeat
// the forward and most of the dX chain (field_mlp_bwd.hip: 2.26 ms of a 7.06 ms step at 8192 rays).  Here:
//   * WEIGHT STREAMING: a workgroup of 8 waves takes a batch of 8 tiles (128 samples) through the branch layer by layer;
//     the bf16 fragment pieces of the layer(s) in use are re-staged from L2 into the SAME LDS region per phase
//     (forward sem0+sem1 at three pieces, 120 KB | transposed head+sem2 | transposed sem1 at two: ~1.8 KB of L2 reads per sample);
//   * COOPERATIVE dW: dW = dY^T X is not accumulated per wave.  All waves write their tile's dY and X columns (bf16
//     pieces) into one [feature row][128 samples] LDS scratch (it overlays the weight region between phases), and every
//     output block (16 x 16 weights) of the layer is owned by ONE wave, which sums it over the batch's 128 samples (four
//     K = 32 MFMAs per piece product).  116 blocks / 8 waves = 15 accumulator blocks (60 registers) per wave instead of
//     464, no cross-wave reduction at the end, and the whole branch runs once, at two waves per SIMD;
//   * the head's dW needs s3 = sem2(s2), which is not recomputed: dW_head = sum_s dlogit_s s3_s = W2 (sum_s dlogit_s s2_s)
//     + b2 sum_s dlogit_s is linear in a 128-vector that falls out of the sem2 round as one more block per wave.
// =====================================================================================================================
template <class Cfg>
struct SegsBigP1 {  // forward sem0, sem1
  static constexpr int N = 2;
  static constexpr int layer(int i) { return i == 0 ? Cfg::L_SEM0 : Cfg::L_SEM1; }
  static constexpr bool isT(int) { return false; }
};
template <class Cfg>
struct SegsBigP2 {  // transposed head, sem2
  static constexpr int N = 2;
  static constexpr int layer(int i) { return i == 0 ? Cfg::L_HEAD : Cfg::L_SEM2; }
  static constexpr bool isT(int) { return true; }
};
template <class Cfg>
struct SegsBigP4 {  // transposed sem1
  static constexpr int N = 1;
  static constexpr int layer(int) { return Cfg::L_SEM1; }
  static constexpr bool isT(int) { return true; }
};

constexpr int CS_LD = 72;     // words per scratch row: 64 hold the batch's 128 bf16 samples; 72 keeps the b128 reads conflict-free
constexpr int CS_ROWS = 128;  // feature rows per operand
template <int NS, int ROWS = CS_ROWS>
constexpr int cs_words() { return 2 * NS * ROWS * CS_LD; }  // [G | X][piece][row][CS_LD]

// this wave's tile (NB accumulator blocks) -> rows ROW0.. of operand array `arr`, columns 16 wave + j, as bf16 pieces
template <int NS, int NB, int ROWS = CS_ROWS>
__device__ __forceinline__ void cs_write(uint32_t* __restrict__ arr, int row0, const f32x4 (&a)[NB], int lane, int wave) {
  const int j = lane & 15, g = lane >> 4;
  __bf16* base = reinterpret_cast<__bf16*>(arr);
#pragma unroll
  for (int blk = 0; blk < NB; ++blk)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = a[blk][r];
#pragma unroll
      for (int pc = 0; pc < NS; ++pc) {
        const __bf16 pv = (__bf16)v;
        base[(size_t)((pc * ROWS + row0 + 16 * blk + 4 * g + r) * (2 * CS_LD)) + 16 * wave + j] = pv;
        if (pc + 1 < NS) v -= (float)pv;
      }
    }
}

// acc[s] += (G rows of block ob0 + s ob_step)^T (X rows of block ib) over the batch's 128 samples
template <int NS, int NOBW, int ROWS = CS_ROWS>
__device__ __forceinline__ void cs_dw(const uint32_t* __restrict__ sG, const uint32_t* __restrict__ sX, int ob0,
                                      int ob_step, int ib, f32x4 (&acc)[NOBW], int lane) {
  const int i = lane & 15, g = lane >> 4;
  // steps (kk, s), K-block outermost (one X fragment set live at a time); explicit double buffer of the G fragments
  // pinned by scheduling barriers (see bf_layer_acc1)
  constexpr int T = 4 * NOBW;
  bf16x8 ga[2][NS], xb[NS];
  auto load_g = [&](int t, bf16x8 (&dst)[NS]) {
    const int kk = t / NOBW, ob = ob0 + (t % NOBW) * ob_step;
#pragma unroll
    for (int pc = 0; pc < NS; ++pc)
      dst[pc] = *reinterpret_cast<const bf16x8*>(sG + (pc * ROWS + 16 * ob + i) * CS_LD + 16 * kk + 4 * g);
  };
  load_g(0, ga[0]);
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int kk = t / NOBW, s = t % NOBW;
    if (s == 0) {
#pragma unroll
      for (int pc = 0; pc < NS; ++pc)
        xb[pc] = *reinterpret_cast<const bf16x8*>(sX + (pc * ROWS + 16 * ib + i) * CS_LD + 16 * kk + 4 * g);
    }
    if (t + 1 < T) load_g(t + 1, ga[(t + 1) & 1]);
#pragma unroll
    for (int sp = NS - 1; sp >= 0; --sp)
#pragma unroll
      for (int pg = 0; pg <= sp; ++pg)
        acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga[t & 1][pg], xb[sp - pg], acc[s], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// bacc += (G rows of block ob)^T 1: every column of the block ends up with the rows' sums over the batch
template <int NS, int ROWS = CS_ROWS>
__device__ __forceinline__ void cs_bias(const uint32_t* __restrict__ sG, int ob, f32x4& bacc, int lane) {
  const int i = lane & 15, g = lane >> 4;
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
#pragma unroll
    for (int pc = NS - 1; pc >= 0; --pc) {
      const bf16x8 ga = *reinterpret_cast<const bf16x8*>(sG + (pc * ROWS + 16 * ob + i) * CS_LD + 16 * kk + 4 * g);
      bacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga, ones, bacc, 0, 0, 0);
    }
}
// one owned block of dW -> the workgroup's fp32 partial image (index space of field_layers.hpp / flush_dw)
__device__ __forceinline__ void store_dw_block(float* __restrict__ W, int ob, int ib, int nib_stride, const f32x4& acc,
                                               int lane) {
  const int jn = lane & 15, g = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    W[((ob * nib_stride + ib) * 64 + swz_slot(4 * g + r, jn >> 2)) * 4 + (jn & 3)] = acc[r];
}
__device__ __forceinline__ void store_bias_block(float* __restrict__ B, int ob, const f32x4& bacc, int lane) {
  if ((lane & 15) == 0) {
    const int g = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) B[16 * ob + 4 * g + r] = bacc[r];
  }
}

// NSF = pieces of the forward recompute, NS = pieces of the dX chain and of dW.  bf16x3 mode: NSF = 3, NS = 2 — the
// recomputed activations decide the ReLU gates, and a gate that flips against the forward pass
// changes a whole sample's contribution to a dW row: with two pieces (2^-17) that happened ~1000x more often
// than with three (2^-27) and single flips showed as ~5e-3 of max |g| under random zero-mean upstream gradients.
template <class Cfg, int NSF, int NS>
__global__ __launch_bounds__(512, 2) void k_field_mlp_bwd_sem_big_bf16(
    const float* __restrict__ packed, const __bf16* __restrict__ image, const float* __restrict__ w_sem2,
    const float* __restrict__ b_sem2, long long N, const float* __restrict__ h_saved, const float* __restrict__ d_logit,
    float* __restrict__ partials) {
  static_assert(Cfg::NSEM == 3 && Cfg::HB == 2 && Cfg::SEMB == 8, "fruit_nerf_big semantic shape");
  constexpr int WAVES = 8, THREADS = 64 * WAVES;
  constexpr int LS0 = Cfg::L_SEM0, LS1 = Cfg::L_SEM1, LS2 = Cfg::L_SEM2, LH = Cfg::L_HEAD;
  static_assert(LS1 == LS0 + 1 && LS2 == LS1 + 1 && LH == LS2 + 1, "the branch's layers are adjacent in the fp32 image");
  using P1 = BfLds<Cfg, SegsBigP1<Cfg>, NSF>;
  using P2 = BfLds<Cfg, SegsBigP2<Cfg>, NS>;
  using P4 = BfLds<Cfg, SegsBigP4<Cfg>, NS>;
  constexpr int REGION = cs_words<NS>() * 4;  // weights of a phase and the dW scratch share this region
  static_assert(P1::BYTES <= REGION && P2::BYTES <= REGION && P4::BYTES <= REGION, "phase weights exceed the shared region");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16x8* wl = reinterpret_cast<bf16x8*>(smem);
  uint32_t* sG = reinterpret_cast<uint32_t*>(smem);
  uint32_t* sX = sG + NS * CS_ROWS * CS_LD;
  float* fbias = reinterpret_cast<float*>(smem + REGION);  // forward biases: sem0 [128] | sem1 [128]
  float* vhead = fbias + 256;                              // epilogue: sum_s dlogit_s s2_s [128] and sum_s dlogit_s
  for (int i = threadIdx.x; i < 256; i += THREADS) fbias[i] = packed[Cfg::W_TOTAL + Cfg::boff(LS0) + i];
  const int lane0 = threadIdx.x & 63, wave = threadIdx.x >> 6;

  // dW blocks owned by this wave: sem0 (ob = (w>>1) + 4 s, ib = w & 1), sem1 (ob = s, ib = w), sem2 (ob = s, ib = w),
  // the virtual head block (dlogit^T s2, ib = w); bias blocks: sem0 / sem1 ob = w, sem2 ob = w (waves 0..3), head (wave 4)
  f32x4 acc0[2], acc1[8], acc2[4], accV[1];
  f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0, b2 = b0;
  zero_vec_bf(acc0);
  zero_vec_bf(acc1);
  zero_vec_bf(acc2);
  zero_vec_bf(accV);

  const long long n_batches = (N + 127) / 128;
  for (long long batch = blockIdx.x; batch < n_batches; batch += gridDim.x) {
    asm volatile("" ::: "memory");
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int j = lane & 15, g = lane >> 4;
    const long long n = batch * 128 + 16 * wave + j;
    const bool valid = n < N;
    const long long nn = valid ? n : N - 1;
    f32x4 h[2];
    h[0] = *reinterpret_cast<const f32x4*>(h_saved + (size_t)nn * 32 + 4 * g);
    h[1] = *reinterpret_cast<const f32x4*>(h_saved + (size_t)nn * 32 + 16 + 4 * g);
    f32x4 Gl[1];
    Gl[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (g == 0 && valid) Gl[0][0] = d_logit[n];

    // phase 1: forward sem0, sem1
    __syncthreads();  // the previous batch's last scratch reads
    P1::template stage<THREADS>(wl, image);
    __syncthreads();
    f32x4 s1[8], s2[8];
    bf_layer1<NSF, 8, 2>(P1::template seg<LS0, false>(wl), fbias, h, s1, lane);
    relu_(s1);
    bf_layer1<NSF, 8, 8>(P1::template seg<LS1, false>(wl), fbias + 128, s1, s2, lane);
    relu_(s2);
    // phase 2: dlogit -> Gs3 (mlp_semantics' output has no activation) -> Gs2.  The transposed head / sem2 fragments are
    // staged into the X half of the region, so that Gs3 and dlogit can go to the G half of the scratch as soon as they
    // exist (16 registers less at the kernel's register peak)
    __syncthreads();
    bf16x8* wl2 = reinterpret_cast<bf16x8*>(sX);
    static_assert(P2::BYTES <= NS * CS_ROWS * CS_LD * 4, "phase-2 fragments must fit the X half");
    P2::template stage<THREADS>(wl2, image);
    __syncthreads();
    f32x4 Gs2[8];
    {
      f32x4 Gs3[4];
      bf_layer_T1<NS, 4, 1>(P2::template seg<LH, true>(wl2), Gl, Gs3, lane);
      // round 1, G side: [Gs3 (blocks 0..3) | dlogit (block 4)]
      cs_write<NS, 4>(sG, 0, Gs3, lane, wave);
      cs_write<NS, 1>(sG, 64, Gl, lane, wave);
      bf_layer_T1<NS, 8, 4>(P2::template seg<LS2, true>(wl2), Gs3, Gs2, lane);
    }
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) Gs2[b][r] = (s2[b][r] > 0.0f) ? Gs2[b][r] : 0.0f;
    // round 1, X side: s2 -> dW sem2, the head's 128-vector, db sem2, db head
    __syncthreads();  // the fragments of phase 2 are dead
    cs_write<NS, 8>(sX, 0, s2, lane, wave);
    __syncthreads();
    cs_dw<NS, 4>(sG, sX, 0, 1, wave, acc2, lane);
    cs_dw<NS, 1>(sG, sX, 4, 0, wave, accV, lane);
    if (wave < 4) cs_bias<NS>(sG, wave, b2, lane);
    if (wave == 4) cs_bias<NS>(sG, 4, b2, lane);  // wave 4's b2 is the head's bias gradient
    // round 2: G = Gs2, X = s1 -> dW sem1, db sem1
    __syncthreads();
    cs_write<NS, 8>(sG, 0, Gs2, lane, wave);
    cs_write<NS, 8>(sX, 0, s1, lane, wave);
    __syncthreads();
    cs_dw<NS, 8>(sG, sX, 0, 1, wave, acc1, lane);
    cs_bias<NS>(sG, wave, b1, lane);
    // phase 4: Gs2 -> Gs1
    __syncthreads();
    P4::template stage<THREADS>(wl, image);
    __syncthreads();
    f32x4 Gs1[8];
    bf_layer_T1<NS, 8, 8>(P4::template seg<LS1, true>(wl), Gs2, Gs1, lane);
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) Gs1[b][r] = (s1[b][r] > 0.0f) ? Gs1[b][r] : 0.0f;
    // round 3: G = Gs1, X = h -> dW sem0, db sem0 (the input is the detached geo: no dX)
    __syncthreads();
    cs_write<NS, 8>(sG, 0, Gs1, lane, wave);
    {
      f32x4 hx[2];  // re-read (L2-resident) instead of keeping 8 registers live across the whole batch
      hx[0] = *reinterpret_cast<const f32x4*>(h_saved + (size_t)nn * 32 + 4 * g);
      hx[1] = *reinterpret_cast<const f32x4*>(h_saved + (size_t)nn * 32 + 16 + 4 * g);
      cs_write<NS, 2>(sX, 0, hx, lane, wave);
    }
    __syncthreads();
    cs_dw<NS, 2>(sG, sX, wave >> 1, 4, wave & 1, acc0, lane);
    cs_bias<NS>(sG, wave, b0, lane);
  }

  // ---- this workgroup's partial image: every block of the branch's layers is written by its owner -------------------
  const int lane = lane0;
  float* part = partials + (size_t)blockIdx.x * (Cfg::W_TOTAL + Cfg::B_TOTAL);
  float* pb = part + Cfg::W_TOTAL;
#pragma unroll
  for (int s = 0; s < 2; ++s) store_dw_block(part + Cfg::woff(LS0), (wave >> 1) + 4 * s, wave & 1, 2, acc0[s], lane);
#pragma unroll
  for (int s = 0; s < 8; ++s) store_dw_block(part + Cfg::woff(LS1), s, wave, 8, acc1[s], lane);
#pragma unroll
  for (int s = 0; s < 4; ++s) store_dw_block(part + Cfg::woff(LS2), s, wave, 8, acc2[s], lane);
  store_bias_block(pb + Cfg::boff(LS0), wave, b0, lane);
  store_bias_block(pb + Cfg::boff(LS1), wave, b1, lane);
  if (wave < 4) store_bias_block(pb + Cfg::boff(LS2), wave, b2, lane);
  // SemanticFieldHead: dW = W2 v + b2 sum(dlogit), v = row 0 of the virtual blocks, sum(dlogit) = row 0 of wave 4's b2
  __syncthreads();
  if ((lane >> 4) == 0) vhead[16 * wave + (lane & 15)] = accV[0][0];
  if (wave == 4 && lane == 0) vhead[128] = b2[0];
  for (int i = threadIdx.x; i < Cfg::nob(LH) * Cfg::nib(LH) * 256; i += THREADS) part[Cfg::woff(LH) + i] = 0.0f;
  for (int i = threadIdx.x; i < 16; i += THREADS) pb[Cfg::boff(LH) + i] = 0.0f;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int k = threadIdx.x;  // head input k = output k of mlp_semantics' last layer
    float dw = b_sem2[k] * vhead[128];
    for (int m = 0; m < 128; ++m) dw = fmaf(w_sem2[k * 128 + m], vhead[m], dw);
    const int ib = k >> 4, kk = k & 15;
    part[Cfg::woff(LH) + (ib * 64 + swz_slot(0, kk >> 2)) * 4 + (kk & 3)] = dw;
    if (k == 0) pb[Cfg::boff(LH)] = vhead[128];
  }
}

// ---- `fruit_nerf_big` semantic branch, forward (h [N,32] -> logit): the same weight streaming in two phases ------------
template <class Cfg>
struct SegsBigF2 {  // forward sem2, head
  static constexpr int N = 2;
  static constexpr int layer(int i) { return i == 0 ? Cfg::L_SEM2 : Cfg::L_HEAD; }
  static constexpr bool isT(int) { return false; }
};

template <class Cfg, int NS, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void k_field_mlp_fwd_sem_big_bf16(
    const float* __restrict__ packed, const __bf16* __restrict__ image, long long N, const float* __restrict__ h_buf,
    float* __restrict__ logit) {
  static_assert(Cfg::NSEM == 3 && Cfg::HB == 2 && Cfg::SEMB == 8, "fruit_nerf_big semantic shape");
  constexpr int THREADS = 64 * WAVES;
  constexpr int LS0 = Cfg::L_SEM0, LS1 = Cfg::L_SEM1, LS2 = Cfg::L_SEM2, LH = Cfg::L_HEAD;
  using F1 = BfLds<Cfg, SegsBigP1<Cfg>, NS>;
  using F2 = BfLds<Cfg, SegsBigF2<Cfg>, NS>;
  constexpr int REGION = F1::BYTES > F2::BYTES ? F1::BYTES : F2::BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16x8* wl = reinterpret_cast<bf16x8*>(smem);
  float* fbias = reinterpret_cast<float*>(smem + REGION);  // sem0 [128] | sem1 [128] | sem2 [64] | head [16]
  for (int i = threadIdx.x; i < 336; i += THREADS) fbias[i] = packed[Cfg::W_TOTAL + Cfg::boff(LS0) + i];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const long long per_batch = 16 * WAVES;
  const long long n_batches = (N + per_batch - 1) / per_batch;
  for (long long batch = blockIdx.x; batch < n_batches; batch += gridDim.x) {
    asm volatile("" ::: "memory");
    const long long n = batch * per_batch + 16 * wave + j;
    const bool valid = n < N;
    const long long nn = valid ? n : N - 1;
    f32x4 h[2];
    h[0] = *reinterpret_cast<const f32x4*>(h_buf + (size_t)nn * 32 + 4 * g);
    h[1] = *reinterpret_cast<const f32x4*>(h_buf + (size_t)nn * 32 + 16 + 4 * g);
    __syncthreads();
    F1::template stage<THREADS>(wl, image);
    __syncthreads();
    f32x4 s1[8], s2[8];
    bf_layer1<NS, 8, 2>(F1::template seg<LS0, false>(wl), fbias, h, s1, lane);
    relu_(s1);
    bf_layer1<NS, 8, 8>(F1::template seg<LS1, false>(wl), fbias + 128, s1, s2, lane);
    relu_(s2);
    __syncthreads();
    F2::template stage<THREADS>(wl, image);
    __syncthreads();
    f32x4 s3[4], hd[1];
    bf_layer1<NS, 4, 8>(F2::template seg<LS2, false>(wl), fbias + 256, s2, s3, lane);
    bf_layer1<NS, 1, 4>(F2::template seg<LH, false>(wl), fbias + 320, s3, hd, lane);
    if (g == 0 && valid) logit[n] = hd[0][0];
  }
}

int field_mlp_fwd_sem_big_bf16(int mode, const FieldPtrs& p, void* image_ws, const float* packed, long long N,
                               const float* h_buf, float* logit, hipStream_t st) {
  using Cfg = FieldCfgBig;
  __bf16* image = reinterpret_cast<__bf16*>(image_ws);  // packed by the caller (k_prepare_field)
  constexpr int WAVES = 8;
  const long long n_batches = (N + 16 * WAVES - 1) / (16 * WAVES);
  long long blocks = n_batches;
  if (blocks > (long long)device_cu_count()) blocks = device_cu_count();
  auto launch = [&](auto kern, int bytes) -> int {
    const int rc = ensure_dyn_lds(kern, bytes);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * WAVES), bytes, st, packed, image, N, h_buf, logit);
    FNR_LAUNCH_CHECK();
    return FNR_OK;
  };
  if (mode == MLP_BF16) {
    constexpr int bytes = BfLds<Cfg, SegsBigP1<Cfg>, 1>::BYTES + 336 * 4;
    return launch(k_field_mlp_fwd_sem_big_bf16<Cfg, 1, WAVES>, bytes);
  }
  constexpr int bytes = BfLds<Cfg, SegsBigP1<Cfg>, 3>::BYTES + 336 * 4;
  static_assert(bytes <= 160 * 1024, "fruit_nerf_big forward fragments exceed the LDS");
  return launch(k_field_mlp_fwd_sem_big_bf16<Cfg, 3, WAVES>, bytes);
}

int field_mlp_bwd_sem_big_bf16(int mode, const FieldPtrs& p, void* image_ws, const float* packed, long long N,
                               const float* h_saved, const float* d_logit, float* partials, long long blocks,
                               hipStream_t st) {
  using Cfg = FieldCfgBig;
  __bf16* image = reinterpret_cast<__bf16*>(image_ws);  // packed by the colour branch's call (or the forward pass)
  auto launch = [&](auto kern, int bytes) -> int {
    const int rc = ensure_dyn_lds(kern, bytes);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), bytes, st, packed, image, p.w[Cfg::L_SEM2], p.b[Cfg::L_SEM2],
                       N, h_saved, d_logit, partials);
    FNR_LAUNCH_CHECK();
    return FNR_OK;
  };
  if (mode == MLP_BF16) {
    constexpr int bytes = cs_words<1>() * 4 + (256 + 144) * 4;
    return launch(k_field_mlp_bwd_sem_big_bf16<Cfg, 1, 1>, bytes);
  }
  constexpr int bytes = cs_words<2>() * 4 + (256 + 144) * 4;
  static_assert(bytes <= 160 * 1024, "fruit_nerf_big semantic branch exceeds the LDS");
  return launch(k_field_mlp_bwd_sem_big_bf16<Cfg, 3, 2>, bytes);
}

// =====================================================================================================================
// `fruit_nerf` shape, backward, COOPERATIVE form (the kernels the bf16 modes launch).  Same building blocks as the
// fruit_nerf_big semantic kernel above: one 16-sample tile per wave, 8 waves = a 128-sample batch per workgroup, dW
// blocks owned by single waves over the batch (<= 6 accumulator blocks per wave instead of 144-164 registers), two waves
// per SIMD.  The branch's forward (NSF pieces) and transposed (NS pieces) fragments stay resident next to the scratch
// (no streaming at these sizes).  NSF = 3 / NS = 2 in the bf16x3 mode: the recomputed activations gate the ReLUs and
// must reproduce the forward pass's signs.
// =====================================================================================================================
constexpr int CB_ROWS = 64;  // feature rows per scratch operand: layers are <= 64 wide
template <int NS>
constexpr int cb_words() { return cs_words<NS, CB_ROWS>(); }

template <class Cfg, class SegsF, class SegsT, int NSF, int NS>
struct CoopLds {
  using F = BfLds<Cfg, SegsF, NSF>;
  using T = BfLds<Cfg, SegsT, NS>;
  static constexpr int SCR_OFF = F::BYTES + T::BYTES;
  static constexpr int FB_OFF = SCR_OFF + cb_words<NS>() * 4;  // floats after the scratch
};
template <class Cfg>
struct SegsColF {
  static constexpr int N = 3;
  static constexpr int layer(int i) { return i == 0 ? Cfg::L_COL0 : i == 1 ? Cfg::L_COL1 : Cfg::L_COL2; }
  static constexpr bool isT(int) { return false; }
};
template <class Cfg>
struct SegsColT {
  static constexpr int N = 3;
  static constexpr int layer(int i) { return i == 0 ? Cfg::L_COL0 : i == 1 ? Cfg::L_COL1 : Cfg::L_COL2; }
  static constexpr bool isT(int) { return true; }
};
template <class Cfg>
struct SegsSemF {
  static constexpr int N = 2;
  static constexpr int layer(int i) { return i == 0 ? Cfg::L_SEM0 : Cfg::L_SEM1; }
  static constexpr bool isT(int) { return false; }
};
template <class Cfg>
struct SegsSemT {
  static constexpr int N = 2;
  static constexpr int layer(int i) { return i == 0 ? Cfg::L_SEM1 : Cfg::L_HEAD; }
  static constexpr bool isT(int) { return true; }
};
template <class Cfg>
struct SegsBaseF {
  static constexpr int N = 2;
  static constexpr int layer(int i) { return i == 0 ? Cfg::L_BASE0 : Cfg::L_BASE1; }
  static constexpr bool isT(int) { return false; }
};
template <class Cfg>
struct SegsBaseT {
  static constexpr int N = 2;
  static constexpr int layer(int i) { return i == 0 ? Cfg::L_BASE0 : Cfg::L_BASE1; }
  static constexpr bool isT(int) { return true; }
};

template <int N>
__device__ __forceinline__ void relu_mask1_(f32x4 (&G)[N], const f32x4 (&act)[N]) {
#pragma unroll
  for (int b = 0; b < N; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) G[b][r] = (act[b][r] > 0.0f) ? G[b][r] : 0.0f;
}
__device__ __forceinline__ void zero_blocks(float* __restrict__ dst, int floats, int threads) {
  for (int i = threadIdx.x; i < floats; i += threads) dst[i] = 0.0f;
}

// ---- colour branch ---------------------------------------------------------------------------------------------------
template <class Cfg, int NSF, int NS>
__global__ __launch_bounds__(512, 2) void k_field_mlp_bwd_color_coop(
    const float* __restrict__ packed, const __bf16* __restrict__ image, const float* __restrict__ ray_bias, RaysDev rays,
    int S, long long N, const float* __restrict__ h_saved, const float* __restrict__ d_rgb, float* __restrict__ d_h,
    float* __restrict__ gsum_tile, float* __restrict__ gsum_extra, float* __restrict__ partials) {
  constexpr int HB = Cfg::HB;  // 16-wide blocks of h: 1 (`fruit_nerf`) or 2 (`fruit_nerf_big`)
  static_assert(HB == 1 || HB == 2, "built shapes");
  constexpr int THREADS = 512;  // 8 waves
  constexpr int LC0 = Cfg::L_COL0, LC1 = Cfg::L_COL1, LC2 = Cfg::L_COL2;
  using CL = CoopLds<Cfg, SegsColF<Cfg>, SegsColT<Cfg>, NSF, NS>;
  using F = typename CL::F;
  using T = typename CL::T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16x8* wf = reinterpret_cast<bf16x8*>(smem);
  bf16x8* wt = reinterpret_cast<bf16x8*>(smem + F::BYTES);
  uint32_t* sG = reinterpret_cast<uint32_t*>(smem + CL::SCR_OFF);
  uint32_t* sX = sG + NS * CB_ROWS * CS_LD;
  float* fbias = reinterpret_cast<float*>(smem + CL::FB_OFF);  // col1 [64] | col2 [16]
  F::template stage<THREADS>(wf, image);
  T::template stage<THREADS>(wt, image);
  for (int i = threadIdx.x; i < 64; i += THREADS) fbias[i] = packed[Cfg::W_TOTAL + Cfg::boff(LC1) + i];
  for (int i = threadIdx.x; i < 16; i += THREADS) fbias[64 + i] = packed[Cfg::W_TOTAL + Cfg::boff(LC2) + i];
  const int lane0 = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // owned blocks: col2 (ob 0, ib = w; waves 0..3), col1 (ob = (w >> 2) + 2 s, ib = w & 3), col0's h blocks (HB = 1: ob = w on
  // waves 0..3; HB = 2: ob = w >> 1, ib = w & 1 on all waves); bias blocks: col1 ob = w (waves 0..3), col2 (wave 4)
  const bool ownA = HB == 2 || wave < 4;
  const int obA = HB == 2 ? (wave >> 1) : wave, ibA = HB == 2 ? (wave & 1) : 0;
  f32x4 accC[1], accB[2], accA[1];
  f32x4 bB = {0.f, 0.f, 0.f, 0.f}, bC = bB;
  zero_vec_bf(accC);
  zero_vec_bf(accB);
  zero_vec_bf(accA);

  const long long n_batches = (N + 127) / 128, n_tiles = (N + 15) / 16;
  for (long long batch = blockIdx.x; batch < n_batches; batch += gridDim.x) {
    asm volatile("" ::: "memory");
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int j = lane & 15, g = lane >> 4;
    const long long tile = batch * 8 + wave;
    const long long n = tile * 16 + j;
    const bool valid = n < N;
    const long long nn = valid ? n : N - 1;
    const long long ray = nn / S;
    f32x4 h[HB], c1[4], c2[4], c3[1];
#pragma unroll
    for (int b = 0; b < HB; ++b) h[b] = *reinterpret_cast<const f32x4*>(h_saved + (size_t)nn * (16 * HB) + 16 * b + 4 * g);
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) c1[ob] = *reinterpret_cast<const f32x4*>(ray_bias + (size_t)ray * 64 + 16 * ob + 4 * g);
    __syncthreads();  // fragments staged (first batch) / the previous batch's last scratch reads
    {
      bf16x8 x[1][NSF];
      bf_operand<NSF, HB>(h, x);
      bf_layer_acc1<NSF, 4, 1>(F::template seg<LC0, false>(wf), x, c1, lane);
    }
    relu_(c1);
    bf_layer1<NSF, 4, 4>(F::template seg<LC1, false>(wf), fbias, c1, c2, lane);
    relu_(c2);
    bf_layer1<NSF, 1, 4>(F::template seg<LC2, false>(wf), fbias + 64, c2, c3, lane);
    f32x4 G3[1];
    G3[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (g == 0 && valid) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float sg = 1.0f / (1.0f + expf(-c3[0][r]));
        G3[0][r] = d_rgb[3 * n + r] * sg * (1.0f - sg);
      }
    }
    // round 1: G = G3, X = c2 -> dW / db of col2
    cs_write<NS, 1, CB_ROWS>(sG, 0, G3, lane, wave);
    cs_write<NS, 4, CB_ROWS>(sX, 0, c2, lane, wave);
    __syncthreads();
    if (wave < 4) cs_dw<NS, 1, CB_ROWS>(sG, sX, 0, 0, wave, accC, lane);
    if (wave == 4) cs_bias<NS, CB_ROWS>(sG, 0, bC, lane);
    f32x4 G2[4];
    bf_layer_T1<NS, 4, 1>(T::template seg<LC2, true>(wt), G3, G2, lane);
    relu_mask1_(G2, c2);
    // round 2: G = G2, X = c1 -> col1
    __syncthreads();
    cs_write<NS, 4, CB_ROWS>(sG, 0, G2, lane, wave);
    cs_write<NS, 4, CB_ROWS>(sX, 0, c1, lane, wave);
    __syncthreads();
    cs_dw<NS, 2, CB_ROWS>(sG, sX, wave >> 2, 2, wave & 3, accB, lane);
    if (wave < 4) cs_bias<NS, CB_ROWS>(sG, wave, bB, lane);
    f32x4 G1[4];
    bf_layer_T1<NS, 4, 4>(T::template seg<LC1, true>(wt), G2, G1, lane);
    relu_mask1_(G1, c1);
    // round 3: G = G1, X = h -> the h block of col0; the 48 ray-constant inputs and the bias are finished per ray by
    // k_color_ray_grads from the tiles' 64 row sums of G1 (exact fp32 DPP sums)
    __syncthreads();
    cs_write<NS, 4, CB_ROWS>(sG, 0, G1, lane, wave);
    cs_write<NS, HB, CB_ROWS>(sX, 0, h, lane, wave);
    __syncthreads();
    if (ownA) cs_dw<NS, 1, CB_ROWS>(sG, sX, obA, 0, ibA, accA, lane);
    if (tile < n_tiles) {
      const long long ray0 = __shfl(ray, lane & 48, 64);
      const bool uniform = __all(ray == ray0);  // invalid lanes were clamped to the last sample's ray
      if (uniform) {
        float mine = 0.0f;  // lane (g, j) keeps feature 16 (j >> 2) + 4 g + (j & 3)
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float t = row16_sum_bf(G1[ob][r]);
            mine = (j == 4 * ob + r) ? t : mine;
          }
        gsum_tile[(size_t)tile * 64 + 16 * (j >> 2) + 4 * g + (j & 3)] = mine;
      } else if (valid) {  // tile straddles rays (S % 16 != 0): per-sample contributions
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
#pragma unroll
          for (int r = 0; r < 4; ++r) atomicAdd(&gsum_extra[(size_t)ray * 64 + 16 * ob + 4 * g + r], G1[ob][r]);
      }
    }
    f32x4 Gh[HB];
    bf_layer_T1<NS, HB, 4>(T::template seg<LC0, true>(wt), G1, Gh, lane);
    if (valid) {
#pragma unroll
      for (int b = 0; b < HB; ++b) *reinterpret_cast<f32x4*>(d_h + (size_t)n * (16 * HB) + 16 * b + 4 * g) = Gh[b];
    }
  }
  const int lane = lane0;
  float* part = partials + (size_t)blockIdx.x * (Cfg::W_TOTAL + Cfg::B_TOTAL);
  float* pb = part + Cfg::W_TOTAL;
  // col0: this kernel owns the h input blocks; the three blocks of ray-constant inputs and the bias belong to
  // k_color_ray_grads, which only overwrites SOME workgroups' images: zero them here
  constexpr int NIB0 = HB + 3;
  for (int i = threadIdx.x; i < 4 * 3 * 256; i += THREADS) {
    const int blk = i >> 8, ob = blk / 3, ib = HB + blk % 3;
    part[Cfg::woff(LC0) + (ob * NIB0 + ib) * 256 + (i & 255)] = 0.0f;
  }
  for (int i = threadIdx.x; i < 64; i += THREADS) pb[Cfg::boff(LC0) + i] = 0.0f;
  if (ownA) store_dw_block(part + Cfg::woff(LC0), obA, ibA, NIB0, accA[0], lane);
  if (wave < 4) {
    store_dw_block(part + Cfg::woff(LC2), 0, wave, 4, accC[0], lane);
    store_bias_block(pb + Cfg::boff(LC1), wave, bB, lane);
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) store_dw_block(part + Cfg::woff(LC1), (wave >> 2) + 2 * s, wave & 3, 4, accB[s], lane);
  if (wave == 4) store_bias_block(pb + Cfg::boff(LC2), 0, bC, lane);
}

// ---- semantic branch -------------------------------------------------------------------------------------------------
template <class Cfg, int NSF, int NS>
__global__ __launch_bounds__(512, 2) void k_field_mlp_bwd_sem_coop(
    const float* __restrict__ packed, const __bf16* __restrict__ image, long long N, const float* __restrict__ h_saved,
    const float* __restrict__ d_logit, float* __restrict__ partials) {
  static_assert(Cfg::NSEM == 2 && Cfg::HB == 1, "`fruit_nerf` shape");
  constexpr int THREADS = 512;  // 8 waves
  constexpr int LS0 = Cfg::L_SEM0, LS1 = Cfg::L_SEM1, LH = Cfg::L_HEAD;
  using CL = CoopLds<Cfg, SegsSemF<Cfg>, SegsSemT<Cfg>, NSF, NS>;
  using F = typename CL::F;
  using T = typename CL::T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16x8* wf = reinterpret_cast<bf16x8*>(smem);
  bf16x8* wt = reinterpret_cast<bf16x8*>(smem + F::BYTES);
  uint32_t* sG = reinterpret_cast<uint32_t*>(smem + CL::SCR_OFF);
  uint32_t* sX = sG + NS * CB_ROWS * CS_LD;
  float* fbias = reinterpret_cast<float*>(smem + CL::FB_OFF);  // sem0 [64] | sem1 [64]
  F::template stage<THREADS>(wf, image);
  T::template stage<THREADS>(wt, image);
  for (int i = threadIdx.x; i < 128; i += THREADS) fbias[i] = packed[Cfg::W_TOTAL + Cfg::boff(LS0) + i];
  const int lane0 = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // owned blocks: head (ob 0, ib = w; waves 0..3), sem1 (ob = (w >> 2) + 2 s, ib = w & 3), sem0 (ob = w, ib 0; waves 0..3);
  // bias blocks: sem1 ob = w (waves 0..3), head (wave 4), sem0 ob = w - 4 (waves 4..7)
  f32x4 accH[1], accB[2], accA[1];
  f32x4 bB = {0.f, 0.f, 0.f, 0.f}, bX = bB, bA = bB;  // bX: head's bias on wave 4
  zero_vec_bf(accH);
  zero_vec_bf(accB);
  zero_vec_bf(accA);
  const long long n_batches = (N + 127) / 128;
  for (long long batch = blockIdx.x; batch < n_batches; batch += gridDim.x) {
    asm volatile("" ::: "memory");
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int j = lane & 15, g = lane >> 4;
    const long long n = (batch * 8 + wave) * 16 + j;
    const bool valid = n < N;
    const long long nn = valid ? n : N - 1;
    f32x4 h[1], s1[4], s2[4];
    h[0] = *reinterpret_cast<const f32x4*>(h_saved + (size_t)nn * 16 + 4 * g);
    __syncthreads();
    bf_layer1<NSF, 4, 1>(F::template seg<LS0, false>(wf), fbias, h, s1, lane);
    relu_(s1);
    bf_layer1<NSF, 4, 4>(F::template seg<LS1, false>(wf), fbias + 64, s1, s2, lane);
    f32x4 Gl[1];
    Gl[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (g == 0 && valid) Gl[0][0] = d_logit[n];
    // round 1: G = dlogit, X = s2 -> SemanticFieldHead
    cs_write<NS, 1, CB_ROWS>(sG, 0, Gl, lane, wave);
    cs_write<NS, 4, CB_ROWS>(sX, 0, s2, lane, wave);
    __syncthreads();
    if (wave < 4) cs_dw<NS, 1, CB_ROWS>(sG, sX, 0, 0, wave, accH, lane);
    if (wave == 4) cs_bias<NS, CB_ROWS>(sG, 0, bX, lane);
    f32x4 Gs2[4];
    bf_layer_T1<NS, 4, 1>(T::template seg<LH, true>(wt), Gl, Gs2, lane);  // no activation on mlp_semantics' last layer
    // round 2: G = Gs2, X = s1 -> sem1
    __syncthreads();
    cs_write<NS, 4, CB_ROWS>(sG, 0, Gs2, lane, wave);
    cs_write<NS, 4, CB_ROWS>(sX, 0, s1, lane, wave);
    __syncthreads();
    cs_dw<NS, 2, CB_ROWS>(sG, sX, wave >> 2, 2, wave & 3, accB, lane);
    if (wave < 4) cs_bias<NS, CB_ROWS>(sG, wave, bB, lane);
    f32x4 Gs1[4];
    bf_layer_T1<NS, 4, 4>(T::template seg<LS1, true>(wt), Gs2, Gs1, lane);
    relu_mask1_(Gs1, s1);
    // round 3: G = Gs1, X = h -> sem0 (input = detached geo: no dX)
    __syncthreads();
    cs_write<NS, 4, CB_ROWS>(sG, 0, Gs1, lane, wave);
    cs_write<NS, 1, CB_ROWS>(sX, 0, h, lane, wave);
    __syncthreads();
    if (wave < 4) cs_dw<NS, 1, CB_ROWS>(sG, sX, wave, 0, 0, accA, lane);
    if (wave >= 4) cs_bias<NS, CB_ROWS>(sG, wave - 4, bA, lane);
  }
  const int lane = lane0;
  float* part = partials + (size_t)blockIdx.x * (Cfg::W_TOTAL + Cfg::B_TOTAL);
  float* pb = part + Cfg::W_TOTAL;
  if (wave < 4) {
    store_dw_block(part + Cfg::woff(LH), 0, wave, 4, accH[0], lane);
    store_dw_block(part + Cfg::woff(LS0), wave, 0, 1, accA[0], lane);
    store_bias_block(pb + Cfg::boff(LS1), wave, bB, lane);
  } else {
    store_bias_block(pb + Cfg::boff(LS0), wave - 4, bA, lane);
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) store_dw_block(part + Cfg::woff(LS1), (wave >> 2) + 2 * s, wave & 3, 4, accB[s], lane);
  if (wave == 4) store_bias_block(pb + Cfg::boff(LH), 0, bX, lane);
}

// ---- base branch -----------------------------------------------------------------------------------------------------
// POSGRAD: the input gradient of the hash grid rides along (camera-pose optimisation, fruit_nerf_config.py:39-43).  The
// forward encode saved J = d feats / d(unit-cube position) [L][3][N] float2; this kernel holds dL/dfeats of its samples
// in registers (lane (g, j): levels g, 4 + g, 8 + g, 12 + g of sample j), so it contracts them with J right here —
// d_pos [N] float4 = sum over levels and features of dL/dfeat * J — instead of a separate launch that re-reads d_feats
// and waits 91 % of its cycles on 64 dependent loads per lane (k_position_from_jacobian: 32 us per 196 608 samples).
template <class Cfg, int NSF, int NS, bool POSGRAD>
__global__ __launch_bounds__(512, 2) void k_field_mlp_bwd_base_coop(
    const float* __restrict__ packed, const __bf16* __restrict__ image, long long N, const float2* __restrict__ feats,
    const uint8_t* __restrict__ selector, const float* __restrict__ d_density, const float* __restrict__ d_h,
    float2* __restrict__ d_feats, float* __restrict__ partials, const float2* __restrict__ jac,
    float4* __restrict__ d_pos) {
  constexpr int HB = Cfg::HB;
  static_assert(HB == 1 || HB == 2, "built shapes");
  constexpr int THREADS = 512;  // 8 waves
  constexpr int LB0 = Cfg::L_BASE0, LB1 = Cfg::L_BASE1;
  using CL = CoopLds<Cfg, SegsBaseF<Cfg>, SegsBaseT<Cfg>, NSF, NS>;
  using F = typename CL::F;
  using T = typename CL::T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16x8* wf = reinterpret_cast<bf16x8*>(smem);
  bf16x8* wt = reinterpret_cast<bf16x8*>(smem + F::BYTES);
  uint32_t* sG = reinterpret_cast<uint32_t*>(smem + CL::SCR_OFF);
  uint32_t* sX = sG + NS * CB_ROWS * CS_LD;
  float* fbias = reinterpret_cast<float*>(smem + CL::FB_OFF);  // base0 [64] | base1 [16 HB]
  F::template stage<THREADS>(wf, image);
  T::template stage<THREADS>(wt, image);
  for (int i = threadIdx.x; i < 64 + 16 * HB; i += THREADS) fbias[i] = packed[Cfg::W_TOTAL + Cfg::boff(LB0) + i];
  const int lane0 = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // owned blocks: base1 (HB = 1: ob 0, ib = w on waves 0..3; HB = 2: ob = w >> 2, ib = w & 3 on all waves), base0 (ob = w >> 1,
  // ib = w & 1); bias: base1 ob = w - 4 (waves 4..4+HB-1), base0 ob = w - 4 (waves 4..7)
  const bool ownB = HB == 2 || wave < 4;
  const int obB = HB == 2 ? (wave >> 2) : 0, ibB = wave & 3;
  f32x4 accB[1], accA[1];
  f32x4 bB = {0.f, 0.f, 0.f, 0.f}, bA = bB;
  zero_vec_bf(accB);
  zero_vec_bf(accA);
  const long long n_batches = (N + 127) / 128;
  for (long long batch = blockIdx.x; batch < n_batches; batch += gridDim.x) {
    asm volatile("" ::: "memory");
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int j = lane & 15, g = lane >> 4;
    const long long n = (batch * 8 + wave) * 16 + j;
    const bool valid = n < N;
    const long long nn = valid ? n : N - 1;
    f32x4 x0[2], a1[4], h[HB];
    load_hash_block(feats, N, nn, g, x0);
    f32x4 Gh[HB];
#pragma unroll
    for (int b = 0; b < HB; ++b) {
      Gh[b] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (valid) Gh[b] = *reinterpret_cast<const f32x4*>(d_h + (size_t)n * (16 * HB) + 16 * b + 4 * g);
    }
    __syncthreads();
    bf_layer1<NSF, 4, 2>(F::template seg<LB0, false>(wf), fbias, x0, a1, lane);
    relu_(a1);
    bf_layer1<NSF, HB, 4>(F::template seg<LB1, false>(wf), fbias + 64, a1, h, lane);
    if (valid && g == 0) {
      const bool sel = selector ? (selector[n] != 0) : true;
      const float te = expf(fminf(fmaxf(h[0][0], -15.0f), 15.0f));  // trunc_exp backward (fruit_field.py:191)
      Gh[0][0] = sel ? d_density[n] * te : 0.0f;                   // colour block has a zero row 0
    }
    // round 1: G = Gh, X = a1 -> base1
    cs_write<NS, HB, CB_ROWS>(sG, 0, Gh, lane, wave);
    cs_write<NS, 4, CB_ROWS>(sX, 0, a1, lane, wave);
    __syncthreads();
    if (ownB) cs_dw<NS, 1, CB_ROWS>(sG, sX, obB, 0, ibB, accB, lane);
    if (wave >= 4 && wave < 4 + HB) cs_bias<NS, CB_ROWS>(sG, wave - 4, bB, lane);
    f32x4 Ga[4];
    bf_layer_T1<NS, 4, HB>(T::template seg<LB1, true>(wt), Gh, Ga, lane);
    relu_mask1_(Ga, a1);
    // round 2: G = Ga, X = hash features -> base0
    __syncthreads();
    cs_write<NS, 4, CB_ROWS>(sG, 0, Ga, lane, wave);
    cs_write<NS, 2, CB_ROWS>(sX, 0, x0, lane, wave);
    __syncthreads();
    float2 jv[POSGRAD ? 12 : 1];
    if constexpr (POSGRAD) {  // issued before the dW round and the last dX layer: consumed after them
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int a = 0; a < 3; ++a) jv[3 * m + a] = ntc_load<NT_JAC_LD>(&jac[((size_t)(4 * m + g) * 3 + a) * N + nn]);   // its only use
#ifdef FNR_JAC_WAIT0   // (round 6 hunt, tests/diagnostics/base_coop_repeat.py: every load of this wave complete right here)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    }
    cs_dw<NS, 1, CB_ROWS>(sG, sX, wave >> 1, 0, wave & 1, accA, lane);
    if (wave >= 4) cs_bias<NS, CB_ROWS>(sG, wave - 4, bA, lane);
    f32x4 Gx[2];
    bf_layer_T1<NS, 2, 4>(T::template seg<LB0, true>(wt), Ga, Gx, lane);
    if (valid) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
        d_feats[(size_t)(4 * m + g) * N + n] = make_float2(Gx[m >> 1][2 * (m & 1)], Gx[m >> 1][2 * (m & 1) + 1]);
    }
    if constexpr (POSGRAD) {
      // EVERY PARTIAL SUM IS PINNED IN ITS OWN REGISTER (the empty asm statements).  Left to itself hipcc pairs the x / y sums
      // into packed-FP32 instructions with cross-half operand selects and threads the six ds_bpermute shuffles through them
      // (v_pk_add_f32 .. op_sel:[0,1] op_sel_hi:[1,0] ; ds_bpermute_b32 ; v_pk_add_f32 ..); in the schedule it picks when
      // the Jacobian's loads are `nt`, ~10 of 12 288 waves per launch then end with a WRONG y component — 5 % off, only waves
      // 0..3 (those that reach the sequence while the others still use the LDS pipe), never d_feats — which is what broke
      // run-to-run reproducibility in round 5 (NT_JAC_LD).  Not the loads: `nt` and plain accesses complete in issue order and
      // see earlier kernels' stores (tools/microbench/nt_load_order.hip, nt_visibility.hip: 0 events in 1e11); the same
      // instructions in isolation do not fail either (pk_forward_hazard.hip) — the defect needs this kernel's context and is not
      // understood beyond that (profiles/r06_raw/nt_hunt.md has the ISA and the probes).  Pinned, the sums compile to scalar
      // v_fma / v_add in every build (tests/test_isa_invariants.py keeps packed instructions out of this reduction) and
      // tests/diagnostics/base_coop_repeat.py gives identical, correct d_position on repeated calls with either load policy.
      float gp[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float gx = Gx[m >> 1][2 * (m & 1)], gy = Gx[m >> 1][2 * (m & 1) + 1];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          gp[a] += gx * jv[3 * m + a].x + gy * jv[3 * m + a].y;
          asm volatile("" : "+v"(gp[a]));
        }
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {  // the four level groups of a sample sit 16 lanes apart
        gp[a] += __shfl_xor(gp[a], 16, 64);
        asm volatile("" : "+v"(gp[a]));
        gp[a] += __shfl_xor(gp[a], 32, 64);
        asm volatile("" : "+v"(gp[a]));
      }
      if (valid && g == 0) d_pos[n] = make_float4(gp[0], gp[1], gp[2], 0.0f);
    }
  }
  const int lane = lane0;
  float* part = partials + (size_t)blockIdx.x * (Cfg::W_TOTAL + Cfg::B_TOTAL);
  float* pb = part + Cfg::W_TOTAL;
  if (ownB) store_dw_block(part + Cfg::woff(LB1), obB, ibB, 4, accB[0], lane);
  store_dw_block(part + Cfg::woff(LB0), wave >> 1, wave & 1, 2, accA[0], lane);
  if (wave >= 4 && wave < 4 + HB) store_bias_block(pb + Cfg::boff(LB1), wave - 4, bB, lane);
  if (wave >= 4) store_bias_block(pb + Cfg::boff(LB0), wave - 4, bA, lane);
}

template <class Cfg, int NSF, int NS>
static int bwd_launch_coop(const float* packed, const __bf16* image, const float* ray_bias, const RaysDev& rd, int S,
                           long long N, const float2* feats, const float* h_saved, const uint8_t* selector,
                           const float* d_density, const float* d_rgb, const float* d_logit, float2* d_feats, float* d_h,
                           float* gsum_tile, float* gsum_extra, float* partials, long long blocks, int branch,
                           hipStream_t st, const float2* jac = nullptr, float4* d_pos = nullptr) {
  // exactly `blocks` workgroups: every one of the caller's partial images must receive this branch's blocks (a
  // workgroup without a batch stores zeros)
  if (branch == 0) {
    using CL = CoopLds<Cfg, SegsColF<Cfg>, SegsColT<Cfg>, NSF, NS>;
    constexpr int bytes = CL::FB_OFF + 80 * 4;
    static_assert(bytes <= 160 * 1024, "colour branch exceeds the LDS");
    auto kern = k_field_mlp_bwd_color_coop<Cfg, NSF, NS>;
    const int once = ensure_dyn_lds(kern, bytes);
    if (once) return once;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), bytes, st, packed, image, ray_bias, rd, S, N, h_saved, d_rgb,
                       d_h, gsum_tile, gsum_extra, partials);
  } else if (branch == 1) {
    if constexpr (Cfg::NSEM == 2) {
      using CL = CoopLds<Cfg, SegsSemF<Cfg>, SegsSemT<Cfg>, NSF, NS>;
      constexpr int bytes = CL::FB_OFF + 128 * 4;
      static_assert(bytes <= 160 * 1024, "semantic branch exceeds the LDS");
      auto kern = k_field_mlp_bwd_sem_coop<Cfg, NSF, NS>;
      const int once = ensure_dyn_lds(kern, bytes);
      if (once) return once;
      hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), bytes, st, packed, image, N, h_saved, d_logit, partials);
    } else {
      FNR_CHECK_ARG(false, "the fruit_nerf_big semantic branch has its own kernel");
    }
  } else {
    using CL = CoopLds<Cfg, SegsBaseF<Cfg>, SegsBaseT<Cfg>, NSF, NS>;
    constexpr int bytes = CL::FB_OFF + (64 + 16 * Cfg::HB) * 4;
    static_assert(bytes <= 160 * 1024, "base branch exceeds the LDS");
    if (jac && d_pos) {
      auto kern = k_field_mlp_bwd_base_coop<Cfg, NSF, NS, true>;
      const int once = ensure_dyn_lds(kern, bytes);
      if (once) return once;
      hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), bytes, st, packed, image, N, feats, selector, d_density,
                         d_h, d_feats, partials, jac, d_pos);
    } else {
      auto kern = k_field_mlp_bwd_base_coop<Cfg, NSF, NS, false>;
      const int once = ensure_dyn_lds(kern, bytes);
      if (once) return once;
      hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), bytes, st, packed, image, N, feats, selector, d_density,
                         d_h, d_feats, partials, jac, d_pos);
    }
  }
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

// ---- launch helpers (called from field_mlp.hip / field_mlp_bwd.hip when fnr_field_net.mlp_mode != 0) --------------
int field_mlp_bwd_pw(int cfg, int mode, int branch, const float* packed, const __bf16* image, const float* ray_bias, const RaysDev& rd,
                     int S, long long N, const float2* feats, const float* h_saved, const uint8_t* selector,
                     const float* d_density, const float* d_rgb, const float* d_logit, float2* d_feats, float* d_h,
                     float* gsum_tile, float* gsum_extra, float* partials, long long blocks, hipStream_t st, const float2* jac,
                     float4* d_pos);  // field_mlp_bwd_pw.hip

template <class Cfg, int NS>
static int fwd_launch_bf16(const float* packed, const __bf16* image, const float* ray_bias, const RaysDev& rd, int S,
                           long long N, const float2* feats, const uint8_t* selector, float* density, float* rgb,
                           float* logit, float* geo_out, float* h_buf, hipStream_t st) {
  constexpr bool WITH_SEM = Cfg::NSEM == 2;
  using Segs = typename std::conditional<WITH_SEM, SegsFwdAll<Cfg>, SegsFwdBaseColor<Cfg>>::type;
  using Lds = BfLds<Cfg, Segs, NS>;
  constexpr int WAVES = 8;
  constexpr int bytes = Lds::BYTES + Cfg::B_TOTAL * 4;
  static_assert(bytes <= 160 * 1024, "forward fragments exceed the LDS");
  auto kern = k_field_mlp_fwd_bf16<Cfg, NS, WAVES, WITH_SEM>;
  const int once = ensure_dyn_lds(kern, bytes);
  if (once) return once;
  const long long n_pairs = (N + 31) / 32;
  long long blocks = (n_pairs + WAVES - 1) / WAVES;
  const long long max_blocks = (long long)device_cu_count() * (bytes <= 80 * 1024 ? 2 : 1);
  if (blocks > max_blocks) blocks = max_blocks;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * WAVES), bytes, st, packed, image, ray_bias, rd, S, N, feats,
                     selector, density, rgb, logit, geo_out, h_buf);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

// cfg 0: `fruit_nerf` (all branches); cfg 1: `fruit_nerf_big`'s base + colour MLPs (logit untouched: the caller runs
// field_mlp_fwd_sem_big_bf16 on the saved h afterwards)
int field_mlp_fwd_bf16(int cfg, int mode, const FieldPtrs& p, const float* packed, void* image_ws, const float* ray_bias,
                       const RaysDev& rd, int S, long long N, const float2* feats, const uint8_t* selector, float* density,
                       float* rgb, float* logit, float* geo_out, float* h_buf, hipStream_t st) {
  __bf16* image = reinterpret_cast<__bf16*>(image_ws);  // packed by the caller (k_prepare_field)
#define FNR_FWD16(C, NSV) \
  fwd_launch_bf16<C, NSV>(packed, image, ray_bias, rd, S, N, feats, selector, density, rgb, logit, geo_out, h_buf, st)
  if (cfg == 0) return mode == MLP_BF16 ? FNR_FWD16(FieldCfgBase, 1) : FNR_FWD16(FieldCfgBase, 3);
  return mode == MLP_BF16 ? FNR_FWD16(FieldCfgBig, 1) : FNR_FWD16(FieldCfgBig, 3);
#undef FNR_FWD16
}

// branch: 0 colour, 1 semantic, 2 base; cfg: 0 `fruit_nerf`, 1 `fruit_nerf_big` (its semantic branch is
// field_mlp_bwd_sem_big_bf16).  The bf16x3 mode runs dX / dW with two pieces (three products), the forward recompute with three.
int field_mlp_bwd_bf16(int cfg, int mode, int branch, const FieldPtrs& p, bool pack, const float* packed, void* image_ws,
                       const float* ray_bias, const RaysDev& rd, int S, long long N, const float2* feats,
                       const float* h_saved, const uint8_t* selector, const float* d_density, const float* d_rgb,
                       const float* d_logit, float2* d_feats, float* d_h, float* gsum_tile, float* gsum_extra,
                       float* partials, long long blocks, hipStream_t st, const float2* jac, float4* d_pos) {
  __bf16* image = reinterpret_cast<__bf16*>(image_ws);
  if (pack) {
    if (cfg == 0)
      launch_pack_field_weights_bf16<FieldCfgBase>(p, mode == MLP_BF16 ? 1 : 3, image, st);
    else
      launch_pack_field_weights_bf16<FieldCfgBig>(p, mode == MLP_BF16 ? 1 : 3, image, st);
    FNR_LAUNCH_CHECK();
  }
  // the per-wave kernels (field_mlp_bwd_pw.hip): every branch of `fruit_nerf`, colour and base of `fruit_nerf_big`.
  // FNR_MLP_BWD_PW=0: the cooperative ones below (A/B).
  static const bool per_wave = [] {
    const char* e = getenv("FNR_MLP_BWD_PW");
    return !(e && atoi(e) == 0);
  }();
  if (per_wave)
    return field_mlp_bwd_pw(cfg, mode, branch, packed, image, ray_bias, rd, S, N, feats, h_saved, selector, d_density, d_rgb, d_logit,
                            d_feats, d_h, gsum_tile, gsum_extra, partials, blocks, st, jac, d_pos);
  if (cfg == 1) {
    if (mode == MLP_BF16)
      return bwd_launch_coop<FieldCfgBig, 1, 1>(packed, image, ray_bias, rd, S, N, feats, h_saved, selector, d_density, d_rgb,
                                                d_logit, d_feats, d_h, gsum_tile, gsum_extra, partials, blocks, branch, st, jac, d_pos);
    return bwd_launch_coop<FieldCfgBig, 3, 2>(packed, image, ray_bias, rd, S, N, feats, h_saved, selector, d_density, d_rgb,
                                              d_logit, d_feats, d_h, gsum_tile, gsum_extra, partials, blocks, branch, st, jac, d_pos);
  }
  if (mode == MLP_BF16)
    return bwd_launch_coop<FieldCfgBase, 1, 1>(packed, image, ray_bias, rd, S, N, feats, h_saved, selector, d_density, d_rgb,
                                               d_logit, d_feats, d_h, gsum_tile, gsum_extra, partials, blocks, branch, st, jac, d_pos);
  return bwd_launch_coop<FieldCfgBase, 3, 2>(packed, image, ray_bias, rd, S, N, feats, h_saved, selector, d_density, d_rgb,
                                             d_logit, d_feats, d_h, gsum_tile, gsum_extra, partials, blocks, branch, st, jac, d_pos);
}

size_t field_bf16_image_bytes() {
  const size_t a = BfImage<FieldCfgBase>::bytes(BF_MAX_PIECES), b = BfImage<FieldCfgBig>::bytes(BF_MAX_PIECES);
  return (a > b ? a : b) + 256;
}

}  // namespace fnr
