// field_mlp_bwd_pw.hip — backward pass of FruitField's MLP stack, PER-WAVE form (round 6): every branch of the `fruit_nerf`
// shape, the colour and base branches of `fruit_nerf_big` (whose 128-wide semantic branch keeps its weight-streaming kernel).
// Replaces the same reference code as field_mlp_bf16.hip: fruit_field.py:132-166,187-281 and its autograd.
//
// The cooperative kernels (field_mlp_bf16.hip) give every 16 x 16 block of a layer's weight gradient to ONE wave over a
// 128-sample batch: eight waves meet at 6 - 7 workgroup barriers per batch, exchange their tiles' dY / X columns through a
// [feature row][128 samples] LDS scratch written with 2-byte stores, and run one 16-sample tile each — MFMA-busy 0.16 - 0.32,
// more than half the cycles waiting (profiles/r05_kernel_trace_pmc.md).  Here a wave is on its own:
//   * one wave = NT 16-sample tiles (NT = 2 or 4) through the whole branch: every LDS weight fragment feeds NT MFMA chains,
//     and 32 samples are exactly one K-block of v_mfma_f32_16x16x32_bf16, so dW[out block][in block] += G^T X is NT/2 MFMAs
//     per piece product, accumulated in the wave's own registers over its persistent loop (24 / 24 / 12 weight blocks + the
//     bias sums for the colour / semantic / base branch);
//   * NO workgroup barrier inside the loop.  The operands of dW need the sample index on the MFMA's K axis while the
//     accumulator layout has it on the lanes: the wave writes its tiles as [sample][16 features] bf16 blocks into a
//     PRIVATE LDS region (one ds_write_b64 per lane, block, piece and tile: the 4 features a lane holds of its sample) and
//     reads the fragments back with the transposing read ds_read_b64_tr_b16 (lane t of 16-lane group g receives feature t
//     of samples 4g .. 4g+3; checked on hardware by tools/microbench/lds_tr16_transpose.hip).  A wave's LDS operations
//     execute in order, so the region is reused X <- G <- X ... without any wait beyond the data dependence;
//   * activations leave the registers as bf16 pieces the moment they exist: the pieces are the next layer's MFMA operand,
//     the first two are what dW's transposition writes, and the sign of the first is the ReLU gate (x > 0 <=> bf16(x) > 0
//     for every x >= 2^-133);
//   * bias gradients are fp32 lane sums of G (reduced over the 16 sample lanes once, at the end);
//   * the waves' accumulators meet ONCE, after the loop: added into one fp32 image in LDS in wave order (fixed order:
//     bit-reproducible), copied out as the workgroup's partial image — same index space as before (field_layers.hpp), so
//     k_color_ray_grads / k_finish_weights are unchanged;
//   * the base branch takes the saved h (for trunc_exp') instead of recomputing mlp_base's second layer.
// Pieces as in the cooperative kernels: NSF = 3 for the forward recompute (the ReLU gates must reproduce the forward
// pass's signs), NS = 2 for dX / dW in the bf16x3 mode; 1 / 1 in the plain bf16 mode.
#include <stdlib.h>

#include <type_traits>

#include "field_bf16.hpp"

namespace fnr {
namespace pw {

using s16x4 = __attribute__((ext_vector_type(4))) short;
using s16x8 = __attribute__((ext_vector_type(8))) short;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
typedef s16x4 __attribute__((address_space(3))) lds_s16x4;

constexpr int TILE_BYTES = 512;  // [16 samples][16 features] bf16
// a wave's private region: up to 4 blocks x NS pieces x its NT tiles
template <int NS, int NT>
constexpr int scratch_bytes() { return 4 * NS * NT * TILE_BYTES; }

// the wave's NT tiles of one quantity, as accumulator blocks / as MFMA operand pieces (bf_operand: K-block kb = accumulator
// blocks 2 kb, 2 kb + 1, four values each per lane)
template <int NT, int NB>
struct Acts {
  f32x4 v[NT][NB];
};
template <int NT, int NB, int NS>
struct Ops {
  bf16x8 v[NT][(NB + 1) / 2][NS];
};
template <int NT, int NB, int NS>
__device__ __forceinline__ void to_ops(Ops<NT, NB, NS>& x, const Acts<NT, NB>& a) {
#pragma unroll
  for (int t = 0; t < NT; ++t) bf_operand<NS, NB>(a.v[t], x.v[t]);
}
template <int NT, int NB>
__device__ __forceinline__ void relu(Acts<NT, NB>& a) {
#pragma unroll
  for (int t = 0; t < NT; ++t) relu_(a.v[t]);
}
// G = 0 where the activation was not positive; the gate is the sign of the activation's first bf16 piece
template <int NT, int NB, int NS>
__device__ __forceinline__ void gate(Acts<NT, NB>& G, const Ops<NT, NB, NS>& act) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const u32x4 p = __builtin_bit_cast(u32x4, act.v[t][b >> 1][0]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned w = p[2 * (b & 1) + (r >> 1)];  // elements 4 (b&1) + r of the K-block: halves of dword 2 (b&1) + r/2
        const bool open = (r & 1) ? ((int)w >= 0x10000) : ((int)(w << 16) > 0);
        G.v[t][b][r] = open ? G.v[t][b][r] : 0.0f;
      }
    }
}

// out (every tile) += W * X: the products of bf_layer_acc (field_bf16.hpp) into every accumulator in the same order, so the
// recomputed activations are the forward kernel's bit for bit; explicit double buffer over the (kb, ob) steps, pinned by
// scheduling barriers (left alone, the scheduler hoists the LDS reads of many steps above the MFMAs)
template <int NT, int NS, int NOB, int NIB>
__device__ __forceinline__ void layer_acc(const bf16x8* __restrict__ seg, const Ops<NT, NIB, NS>& x, Acts<NT, NOB>& o, int lane) {
  constexpr int NKB = (NIB + 1) / 2, PB = NOB * NKB * 64, T = NOB * NKB;
  bf16x8 w[2][NS];
  auto load = [&](int t, bf16x8 (&dst)[NS]) {
    const int kb = t / NOB, ob = t % NOB;
#pragma unroll
    for (int pc = 0; pc < NS; ++pc) dst[pc] = seg[pc * PB + (ob * NKB + kb) * 64 + lane];
  };
  load(0, w[0]);
#pragma unroll
  for (int t = 0; t < T; ++t) {
    if (t + 1 < T) load(t + 1, w[(t + 1) & 1]);
    const int kb = t / NOB, ob = t % NOB;
#pragma unroll
    for (int s = NS - 1; s >= 0; --s)  // smallest terms first
#pragma unroll
      for (int pw = 0; pw <= s; ++pw)
#pragma unroll
        for (int tl = 0; tl < NT; ++tl)
          o.v[tl][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[t & 1][pw], x.v[tl][kb][s - pw], o.v[tl][ob], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// out = W in + the bias block `B` (zero without one); works for transposed segments (dX = W^T G) too
template <int NT, int NS, int NOB, int NIB>
__device__ __forceinline__ void layer(const bf16x8* __restrict__ seg, const float* __restrict__ B, const Ops<NT, NIB, NS>& x,
                                      Acts<NT, NOB>& o, int lane) {
  const int g = lane >> 4;
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob) {
    const f32x4 b0 = B ? *reinterpret_cast<const f32x4*>(B + 16 * ob + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) o.v[t][ob] = b0;
  }
  layer_acc<NT, NS, NOB, NIB>(seg, x, o, lane);
}

// ---- dW: operands through the wave's private transposition region ---------------------------------------------------------
// A 512-byte tile holds [16 samples][16 features] bf16 as 64 chunks of 8 bytes: chunk (s, c) = features 4c .. 4c+3 of sample s.
// The transposing read lets every lane name its own chunk (lane t of a 16-lane group must point at chunk (row t/4, column group
// t%4) of the 4-sample block it wants), so the chunk order is free, and it is chosen for the banks:
//     chunk (s, c) sits at 8 * (16 c + (s ^ 8 (c >> 1))) bytes.
// Writes (ds_write_b64, banks of 16 contiguous lanes mod 32 dwords): lane (j, g) holds chunk (s = j, c = g): a 16-lane group
// covers 128 contiguous bytes (permuted by the XOR) — every bank once.  Reads (ds_read_b64_tr_b16, two 32-lane groups, mod 64
// dwords): lanes 0..31 read samples 0..7 of all four column groups: dwords 2s, 32 + 2s, 80 + 2s, 112 + 2s — every bank once; lanes
// 32..63 likewise.  (The row-major order [s][c] is 4-way conflicted on the writes: half of this kernel's LDS cycles, round 6
// counters.)
__device__ __forceinline__ int chunk_offset(int s, int c) { return 8 * (16 * c + (s ^ ((c >> 1) << 3))); }
// [block][piece][tile] x 512-byte tile; lane (j, g) holds features 4g .. 4g+3 of sample j.  The first NS of the NSH pieces the
// operand holds are written (the forward recompute splits into three, dW uses two).
template <int NT, int NS, int NB, int NSH>
__device__ __forceinline__ void tr_write(unsigned char* __restrict__ scr, const Ops<NT, NB, NSH>& x, int lane) {
  static_assert(NS <= NSH, "pieces");
  unsigned char* dst = scr + chunk_offset(lane & 15, lane >> 4);
#pragma unroll
  for (int blk = 0; blk < NB; ++blk)
#pragma unroll
    for (int pc = 0; pc < NS; ++pc)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const s16x8 v = __builtin_bit_cast(s16x8, x.v[t][blk >> 1][pc]);
        const s16x4 h = (blk & 1) ? __builtin_shufflevector(v, v, 4, 5, 6, 7) : __builtin_shufflevector(v, v, 0, 1, 2, 3);
        *reinterpret_cast<s16x4*>(dst + ((blk * NS + pc) * NT + t) * TILE_BYTES) = h;
      }
}
// this lane's chunk for the transposing read of samples 4g .. 4g+3 (g = lane >> 4): row t/4, column group t%4
__device__ __forceinline__ int frag_offset(int lane) {
  const int t = lane & 15, g = lane >> 4;
  return chunk_offset(4 * g + (t >> 2), t & 3);
}
// MFMA operand of block blk, piece pc over the tile pair kp: lane (t, g) gets feature t of samples 4g .. 4g+3 of tile 2 kp
// (K-slots 0..3) and of tile 2 kp + 1 (4..7)
template <int NT, int NS>
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* __restrict__ scr, int blk, int pc, int kp, int lane) {
  const unsigned char* p = scr + ((blk * NS + pc) * NT + 2 * kp) * TILE_BYTES + frag_offset(lane);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + TILE_BYTES));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}
// one tile per wave: the K = 16 operand of v_mfma_f32_16x16x16_bf16 is exactly one transposing read
template <int NS>
__device__ __forceinline__ s16x4 tr_frag1(const unsigned char* __restrict__ scr, int blk, int pc, int lane) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(scr + (blk * NS + pc) * TILE_BYTES + frag_offset(lane)));
}
// acc[ob][ib] += (G block ob)^T (X block ib) over the wave's 16 NT samples.  The region is used twice (X, then G): the wave's
// LDS operations execute in order.
// ROWSUM: additionally rs[ob] += (G block ob)^T SEL, SEL [sample][16 columns] a 0/1 selector operand (see the colour branch):
// column c of rs[ob] = the sum of the block's rows over the samples column c selects — on the matrix pipe, from the fragments
// that are in registers anyway, instead of 64 dependent DPP additions (+ their hazard no-ops) per tile.
template <int NT>
struct SelOperand {
  typename std::conditional<NT == 1, s16x4, bf16x8>::type v[NT == 1 ? 1 : NT / 2];
};
template <int NT, int NS, int NOB, int NIB, int NSG, int NSX, bool ROWSUM = false>
__device__ __forceinline__ void dw_round(unsigned char* __restrict__ scr, const Ops<NT, NOB, NSG>& G, const Ops<NT, NIB, NSX>& X,
                                         f32x4 (&acc)[NOB * NIB], int lane, const SelOperand<NT>* sel = nullptr,
                                         f32x4 (*rs)[NOB] = nullptr) {
  tr_write<NT, NS, NIB, NSX>(scr, X, lane);
  if constexpr (NT == 1) {
    s16x4 xf[NIB][NS];
#pragma unroll
    for (int ib = 0; ib < NIB; ++ib)
#pragma unroll
      for (int pc = 0; pc < NS; ++pc) xf[ib][pc] = tr_frag1<NS>(scr, ib, pc, lane);
    tr_write<NT, NS, NOB, NSG>(scr, G, lane);
    s16x4 gf[2][NS];
    auto load = [&](int ob, s16x4 (&dst)[NS]) {
#pragma unroll
      for (int pc = 0; pc < NS; ++pc) dst[pc] = tr_frag1<NS>(scr, ob, pc, lane);
    };
    load(0, gf[0]);
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
      if (ob + 1 < NOB) load(ob + 1, gf[(ob + 1) & 1]);
#pragma unroll
      for (int ib = 0; ib < NIB; ++ib)
#pragma unroll
        for (int s = NS - 1; s >= 0; --s)  // smallest terms first
#pragma unroll
          for (int pg = 0; pg <= s; ++pg)
            acc[ob * NIB + ib] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(gf[ob & 1][pg], xf[ib][s - pg], acc[ob * NIB + ib], 0, 0, 0);
      if constexpr (ROWSUM) {
#pragma unroll
        for (int pc = NS - 1; pc >= 0; --pc)
          (*rs)[ob] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(gf[ob & 1][pc], sel->v[0], (*rs)[ob], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    constexpr int KP = NT / 2;
    bf16x8 xf[NIB][KP][NS];
#pragma unroll
    for (int ib = 0; ib < NIB; ++ib)
#pragma unroll
      for (int kp = 0; kp < KP; ++kp)
#pragma unroll
        for (int pc = 0; pc < NS; ++pc) xf[ib][kp][pc] = tr_frag<NT, NS>(scr, ib, pc, kp, lane);
    tr_write<NT, NS, NOB, NSG>(scr, G, lane);
    bf16x8 gf[2][KP][NS];
    auto load = [&](int ob, bf16x8 (&dst)[KP][NS]) {
#pragma unroll
      for (int kp = 0; kp < KP; ++kp)
#pragma unroll
        for (int pc = 0; pc < NS; ++pc) dst[kp][pc] = tr_frag<NT, NS>(scr, ob, pc, kp, lane);
    };
    load(0, gf[0]);
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
      if (ob + 1 < NOB) load(ob + 1, gf[(ob + 1) & 1]);
#pragma unroll
      for (int ib = 0; ib < NIB; ++ib)
#pragma unroll
        for (int kp = 0; kp < KP; ++kp)
#pragma unroll
          for (int s = NS - 1; s >= 0; --s)  // smallest terms first
#pragma unroll
            for (int pg = 0; pg <= s; ++pg)
              acc[ob * NIB + ib] =
                  __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[ob & 1][kp][pg], xf[ib][kp][s - pg], acc[ob * NIB + ib], 0, 0, 0);
      if constexpr (ROWSUM) {
#pragma unroll
        for (int kp = 0; kp < KP; ++kp)
#pragma unroll
          for (int pc = NS - 1; pc >= 0; --pc)
            (*rs)[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[ob & 1][kp][pc], sel->v[kp], (*rs)[ob], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}
// bias gradient: fp32 lane sums of G (each lane: its NT samples), reduced over the sample lanes after the loop
template <int NT, int NB>
__device__ __forceinline__ void bias_add(f32x4 (&bsum)[NB], const Acts<NT, NB>& G) {
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    f32x4 s = G.v[0][b];
#pragma unroll
    for (int t = 1; t < NT; ++t) s += G.v[t][b];
    bsum[b] += s;
  }
}

__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
  return v;
}
template <int N>
__device__ __forceinline__ void zero_vec(f32x4 (&a)[N]) {
#pragma unroll
  for (int b = 0; b < N; ++b) a[b] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// ---- the waves' accumulators -> the workgroup's partial image -----------------------------------------------------------
// Where one accumulator block goes in the partial image (index space of field_layers.hpp): a 16 x 16 weight block (ob, ib) of
// the layer at float offset `off` with `stride` input blocks, or (bias) rows 16 ob .. + 15 of the bias block at `off`.
struct Dest {
  int off, ob, ib, stride, bias;
};
// Every wave holds the same NB blocks (its own samples' sums).  Rounds of WAVES blocks: all waves park their copies of the
// round's blocks in LDS, then wave w adds up block w's WAVES copies IN WAVE ORDER (fixed order: bit-reproducible) and writes
// the total straight into the image.  Two barriers per round and no read-modify-write anywhere: adding into one LDS image
// wave after wave (the first form of this epilogue) serialised ~130 dependent LDS round trips per wave, eight waves in a
// row: 12 - 25 us per kernel, more than a third of its run time.
template <int NB, int WAVES, class Table>
__device__ __forceinline__ void reduce_and_store(unsigned char* __restrict__ smem, float* __restrict__ part, const f32x4 (&all)[NB],
                                                 int wave, int lane) {
  f32x4* slots = reinterpret_cast<f32x4*>(smem);  // [block of the round][wave][lane]
  constexpr int ROUNDS = (NB + WAVES - 1) / WAVES;
#pragma unroll
  for (int rd = 0; rd < ROUNDS; ++rd) {
    __syncthreads();  // the loop's last LDS reads / the previous round's
#pragma unroll
    for (int k = 0; k < WAVES; ++k)
      if (rd * WAVES + k < NB) slots[(k * WAVES + wave) * 64 + lane] = all[rd * WAVES + k];
    __syncthreads();
    const int b = rd * WAVES + wave;
    if (b < NB) {
      f32x4 s = slots[(wave * WAVES) * 64 + lane];
#pragma unroll
      for (int w = 1; w < WAVES; ++w) s += slots[(wave * WAVES + w) * 64 + lane];
      const Dest d = Table::get(b);
      const int jn = lane & 15, g = lane >> 4;
      if (d.bias) {
        if (jn == 0) *reinterpret_cast<f32x4*>(part + d.off + 16 * d.ob + 4 * g) = s;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) part[d.off + ((d.ob * d.stride + d.ib) * 64 + swz_slot(4 * g + r, jn >> 2)) * 4 + (jn & 3)] = s[r];
      }
    }
  }
}
constexpr int reduce_lds_bytes(int waves) { return waves * waves * 64 * 16; }
// lane sums of G -> the 16-sample-lane total (every lane of the row ends up with it)
__device__ __forceinline__ f32x4 bias_total(const f32x4& bsum) {
  f32x4 t;
#pragma unroll
  for (int r = 0; r < 4; ++r) t[r] = row16_sum(bsum[r]);
  return t;
}

// ---- segment lists ---------------------------------------------------------------------------------------------------
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
struct SegsBaseF {  // mlp_base layer 0 only: h comes from the forward pass
  static constexpr int N = 1;
  static constexpr int layer(int) { return Cfg::L_BASE0; }
  static constexpr bool isT(int) { return false; }
};
template <class Cfg>
struct SegsBaseT {
  static constexpr int N = 2;
  static constexpr int layer(int i) { return i == 0 ? Cfg::L_BASE0 : Cfg::L_BASE1; }
  static constexpr bool isT(int) { return true; }
};

template <class Cfg, class SegsF, class SegsT, int NSF, int NS, int NT, int WAVES, int BIAS_FLOATS>
struct Lds {
  using F = BfLds<Cfg, SegsF, NSF>;
  using T = BfLds<Cfg, SegsT, NS>;
  static constexpr int SCR_OFF = F::BYTES + T::BYTES;
  static constexpr int FB_OFF = SCR_OFF + WAVES * scratch_bytes<NS, NT>();
  static constexpr int LOOP_BYTES = FB_OFF + BIAS_FLOATS * 4;
  static constexpr int BYTES = LOOP_BYTES > reduce_lds_bytes(WAVES) ? LOOP_BYTES : reduce_lds_bytes(WAVES);  // the epilogue's slots overlay everything
};

// the wave's samples: tile t of group gr covers samples 16 (NT gr + t) .. + 15; n = this lane's sample of tile t
template <int NT>
struct Samples {
  int n[NT];     // clamped to N - 1 where invalid: loads stay in bounds, the upstream gradients of such lanes are zero
  bool ok[NT];
  __device__ __forceinline__ Samples(int gr, int j, int N) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int s = (gr * NT + t) * 16 + j;
      ok[t] = s < N;
      n[t] = ok[t] ? s : N - 1;
    }
  }
};

// ---- colour branch -----------------------------------------------------------------------------------------------------
template <class Cfg, int NSF, int NS, int NT, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void k_field_mlp_bwd_color_pw(
    const float* __restrict__ packed, const __bf16* __restrict__ image, const float* __restrict__ ray_bias, RaysDev rays,
    int S, int N, const float* __restrict__ h_saved, const float* __restrict__ d_rgb, float* __restrict__ d_h,
    float* __restrict__ gsum_tile, float* __restrict__ gsum_extra, float* __restrict__ partials) {
  constexpr int HB = Cfg::HB;  // 16-wide blocks of h: 1 (`fruit_nerf`) or 2 (`fruit_nerf_big`)
  static_assert(HB == 1 || HB == 2, "built shapes");
  constexpr int THREADS = 64 * WAVES;
  constexpr int LC0 = Cfg::L_COL0, LC1 = Cfg::L_COL1, LC2 = Cfg::L_COL2;
  using L = Lds<Cfg, SegsColF<Cfg>, SegsColT<Cfg>, NSF, NS, NT, WAVES, 80>;
  using F = typename L::F;
  using T = typename L::T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16x8* wf = reinterpret_cast<bf16x8*>(smem);
  bf16x8* wt = reinterpret_cast<bf16x8*>(smem + F::BYTES);
  float* fbias = reinterpret_cast<float*>(smem + L::FB_OFF);  // col1 [64] | col2 [16]
  F::template stage<THREADS>(wf, image);
  T::template stage<THREADS>(wt, image);
  for (int i = threadIdx.x; i < 64; i += THREADS) fbias[i] = packed[Cfg::W_TOTAL + Cfg::boff(LC1) + i];
  for (int i = threadIdx.x; i < 16; i += THREADS) fbias[64 + i] = packed[Cfg::W_TOTAL + Cfg::boff(LC2) + i];
  __syncthreads();
  const int lane0 = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char* scr = smem + L::SCR_OFF + wave * scratch_bytes<NS, NT>();
  f32x4 accC[4], accB[16], accA[4 * HB];  // col2 [ob 0][ib], col1 [ob][ib], col0's h blocks [ob][ib < HB]
  f32x4 bC[1], bB[4];                // (the bias of col0 belongs to k_color_ray_grads)
  zero_vec(accC);
  zero_vec(accB);
  zero_vec(accA);
  zero_vec(bC);
  zero_vec(bB);

  const int n_groups = (N + 16 * NT - 1) / (16 * NT), n_tiles = (N + 15) / 16;
  for (int gr = blockIdx.x * WAVES + wave; gr < n_groups; gr += gridDim.x * WAVES) {
    asm volatile("" ::: "memory");  // keep the LDS fragment reads inside the loop
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int j = lane & 15, g = lane >> 4;
    const Samples<NT> sm(gr, j, N);
    int ray[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) ray[t] = sm.n[t] / S;
    // forward recompute; every activation leaves the registers as bf16 pieces
    Ops<NT, 4, NSF> x1, x2;
    Acts<NT, 1> c3;
    {
      Acts<NT, HB> h;
      Acts<NT, 4> c1;
      Ops<NT, HB, NSF> hx;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) h.v[t][hb] = *reinterpret_cast<const f32x4*>(h_saved + (size_t)sm.n[t] * (16 * HB) + 16 * hb + 4 * g);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) c1.v[t][ob] = *reinterpret_cast<const f32x4*>(ray_bias + (size_t)ray[t] * 64 + 16 * ob + 4 * g);
      }
      to_ops(hx, h);
      layer_acc<NT, NSF, 4, HB>(F::template seg<LC0, false>(wf), hx, c1, lane);
      relu(c1);
      to_ops(x1, c1);
    }
    {
      Acts<NT, 4> c2;
      layer<NT, NSF, 4, 4>(F::template seg<LC1, false>(wf), fbias, x1, c2, lane);
      relu(c2);
      to_ops(x2, c2);
    }
    layer<NT, NSF, 1, 4>(F::template seg<LC2, false>(wf), fbias + 64, x2, c3, lane);
    Ops<NT, 1, NS> g3;
    {
      Acts<NT, 1> G3;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        G3.v[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (g == 0 && sm.ok[t]) {
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            const float sg = 1.0f / (1.0f + expf(-c3.v[t][0][r]));
            G3.v[t][0][r] = d_rgb[3 * (size_t)sm.n[t] + r] * sg * (1.0f - sg);
          }
        }
      }
      bias_add(bC, G3);
      to_ops(g3, G3);
    }
    // col2: G = G3, X = c2
    dw_round<NT, NS, 1, 4>(scr, g3, x2, accC, lane);
    Ops<NT, 4, NS> g2;
    {
      Acts<NT, 4> G2;
      layer<NT, NS, 4, 1>(T::template seg<LC2, true>(wt), nullptr, g3, G2, lane);
      gate(G2, x2);
      bias_add(bB, G2);
      to_ops(g2, G2);
    }
    // col1: G = G2, X = c1
    dw_round<NT, NS, 4, 4>(scr, g2, x1, accB, lane);
    Ops<NT, 4, NS> g1;
    {
      Acts<NT, 4> G1;
      layer<NT, NS, 4, 4>(T::template seg<LC1, true>(wt), nullptr, g2, G1, lane);
      gate(G1, x1);
      if (S < 16) {  // a tile holds more than two rays (the plugin API's per-sample queries: S = 1): per-sample contributions
#pragma unroll
        for (int t = 0; t < NT; ++t)
          if (sm.ok[t]) {
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
#pragma unroll
              for (int r = 0; r < 4; ++r) atomicAdd(&gsum_extra[(size_t)ray[t] * 64 + 16 * ob + 4 * g + r], G1.v[t][ob][r]);
          }
      }
      to_ops(g1, G1);
    }
    // col0: G = G1, X = h -> its h block.  The 48 ray-constant inputs and the bias are finished per ray by k_color_ray_grads
    // from per-ray row sums of G1: a tile inside one ray stores its 64 sums to gsum_tile; a tile that straddles two rays
    // (S % 16 != 0, S >= 16) adds its two parts to gsum_extra.  The sums ride on the round's MFMAs: selector column 2 t' = the samples of
    // tile t' (of a K-block's two) that belong to the tile's first ray, column 2 t' + 1 = the rest.
    int first_ray[NT], boundary[NT];
    SelOperand<NT> sel;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int s0 = (gr * NT + t) * 16;
      first_ray[t] = (s0 < N ? s0 : N - 1) / S;
      boundary[t] = (first_ray[t] + 1) * S;  // first sample of the next ray
    }
    {
      const unsigned short one = 0x3F80;  // bf16 1.0
      if constexpr (NT == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool lo = 16 * gr + 4 * g + e < boundary[0];
          sel.v[0][e] = (short)(((j == 0 && lo) || (j == 1 && !lo)) ? one : 0);
        }
      } else {
#pragma unroll
        for (int kp = 0; kp < NT / 2; ++kp) {
          s16x8 v;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int t = 2 * kp + (e >> 2);
            const bool lo = (gr * NT + t) * 16 + 4 * g + (e & 3) < boundary[t];
            v[e] = (short)(((j == 2 * (e >> 2) && lo) || (j == 2 * (e >> 2) + 1 && !lo)) ? one : 0);
          }
          sel.v[kp] = __builtin_bit_cast(bf16x8, v);
        }
      }
    }
    f32x4 rs[NT == 1 ? 1 : NT / 2][4];
    {
      Acts<NT, HB> h;  // (re-read, L2-resident, instead of its pieces living through the whole recompute)
      Ops<NT, HB, NS> hx;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) h.v[t][hb] = *reinterpret_cast<const f32x4*>(h_saved + (size_t)sm.n[t] * (16 * HB) + 16 * hb + 4 * g);
      to_ops(hx, h);
      static_assert(NT <= 2, "one K-block of row sums per round");
      zero_vec(rs[0]);
      dw_round<NT, NS, 4, HB, NS, NS, true>(scr, g1, hx, accA, lane, &sel, &rs[0]);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int tile = gr * NT + t;
      if (tile < n_tiles) {
        const int last = (16 * tile + 15 < N) ? 16 * tile + 15 : N - 1;
        const int col = (NT == 1) ? 0 : 2 * (t & 1);
        if (last < boundary[t]) {  // inside one ray (S < 16: only the last, partial tile can be; its samples went to gsum_extra)
          if (j == col) {
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
              *reinterpret_cast<f32x4*>(gsum_tile + (size_t)tile * 64 + 16 * ob + 4 * g) = (S < 16) ? f32x4{0.f, 0.f, 0.f, 0.f} : rs[0][ob];
          }
        } else if (S >= 16 && (j == col || j == col + 1)) {
          const int ray = first_ray[t] + (j - col);
#pragma unroll
          for (int ob = 0; ob < 4; ++ob)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(&gsum_extra[(size_t)ray * 64 + 16 * ob + 4 * g + r], rs[0][ob][r]);
        }
      }
    }
    Acts<NT, HB> Gh;
    layer<NT, NS, HB, 4>(T::template seg<LC0, true>(wt), nullptr, g1, Gh, lane);
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (sm.ok[t]) {
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) *reinterpret_cast<f32x4*>(d_h + (size_t)sm.n[t] * (16 * HB) + 16 * hb + 4 * g) = Gh.v[t][hb];
      }
  }
  const int lane = lane0;
  float* part = partials + (size_t)blockIdx.x * (Cfg::W_TOTAL + Cfg::B_TOTAL);
  constexpr int NA = 4 * HB, NALL = NA + 25;
  struct Table {  // accA [ob][ib < HB] | accB [ob][ib] | accC [ib] | bias col1 [ob] | bias col2
    __device__ static __forceinline__ Dest get(int b) {
      constexpr int NA = 4 * Cfg::HB;
      if (b < NA) return Dest{Cfg::woff(LC0), b / Cfg::HB, b % Cfg::HB, Cfg::HB + 3, 0};
      b -= NA;
      if (b < 16) return Dest{Cfg::woff(LC1), b >> 2, b & 3, 4, 0};
      if (b < 20) return Dest{Cfg::woff(LC2), 0, b - 16, 4, 0};
      if (b < 24) return Dest{Cfg::W_TOTAL + Cfg::boff(LC1), b - 20, 0, 0, 1};
      return Dest{Cfg::W_TOTAL + Cfg::boff(LC2), 0, 0, 0, 1};
    }
  };
  f32x4 all[NALL];
#pragma unroll
  for (int i = 0; i < NA; ++i) all[i] = accA[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) all[NA + i] = accB[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) all[NA + 16 + i] = accC[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) all[NA + 20 + i] = bias_total(bB[i]);
  all[NA + 24] = bias_total(bC[0]);
  reduce_and_store<NALL, WAVES, Table>(smem, part, all, wave, lane);
  // col0: this kernel owns the h input block; the three blocks of ray-constant inputs and the bias belong to
  // k_color_ray_grads, which only overwrites SOME workgroups' images: zero them here
  constexpr int NIB0 = Cfg::HB + 3;
  for (int i = threadIdx.x; i < 4 * 3 * 256; i += THREADS) {
    const int blk = i >> 8, ob = blk / 3, ib = Cfg::HB + blk % 3;
    part[Cfg::woff(LC0) + (ob * NIB0 + ib) * 256 + (i & 255)] = 0.0f;
  }
  for (int i = threadIdx.x; i < 64; i += THREADS) part[Cfg::W_TOTAL + Cfg::boff(LC0) + i] = 0.0f;
}

// ---- semantic branch ---------------------------------------------------------------------------------------------------
template <class Cfg, int NSF, int NS, int NT, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void k_field_mlp_bwd_sem_pw(
    const float* __restrict__ packed, const __bf16* __restrict__ image, int N, const float* __restrict__ h_saved,
    const float* __restrict__ d_logit, float* __restrict__ partials) {
  static_assert(Cfg::NSEM == 2 && Cfg::HB == 1, "`fruit_nerf` shape");
  constexpr int THREADS = 64 * WAVES;
  constexpr int LS0 = Cfg::L_SEM0, LS1 = Cfg::L_SEM1, LH = Cfg::L_HEAD;
  static_assert(LS1 == LS0 + 1 && LH == LS1 + 1, "the branch's layers are adjacent in the fp32 image");
  using L = Lds<Cfg, SegsSemF<Cfg>, SegsSemT<Cfg>, NSF, NS, NT, WAVES, 128>;
  using F = typename L::F;
  using T = typename L::T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16x8* wf = reinterpret_cast<bf16x8*>(smem);
  bf16x8* wt = reinterpret_cast<bf16x8*>(smem + F::BYTES);
  float* fbias = reinterpret_cast<float*>(smem + L::FB_OFF);  // sem0 [64] | sem1 [64]
  F::template stage<THREADS>(wf, image);
  T::template stage<THREADS>(wt, image);
  for (int i = threadIdx.x; i < 128; i += THREADS) fbias[i] = packed[Cfg::W_TOTAL + Cfg::boff(LS0) + i];
  __syncthreads();
  const int lane0 = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char* scr = smem + L::SCR_OFF + wave * scratch_bytes<NS, NT>();
  f32x4 accH[4], accB[16], accA[4];  // head [0][ib], sem1 [ob][ib], sem0 [ob][0]
  f32x4 bH[1], bB[4], bA[4];
  zero_vec(accH);
  zero_vec(accB);
  zero_vec(accA);
  zero_vec(bH);
  zero_vec(bB);
  zero_vec(bA);
  const int n_groups = (N + 16 * NT - 1) / (16 * NT);
  for (int gr = blockIdx.x * WAVES + wave; gr < n_groups; gr += gridDim.x * WAVES) {
    asm volatile("" ::: "memory");
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int j = lane & 15, g = lane >> 4;
    const Samples<NT> sm(gr, j, N);
    Ops<NT, 4, NSF> x1;
    Ops<NT, 4, NS> x2;
    {
      Ops<NT, 1, NSF> hx;
      Acts<NT, 1> h;
      Acts<NT, 4> s1;
#pragma unroll
      for (int t = 0; t < NT; ++t) h.v[t][0] = *reinterpret_cast<const f32x4*>(h_saved + (size_t)sm.n[t] * 16 + 4 * g);
      to_ops(hx, h);
      layer<NT, NSF, 4, 1>(F::template seg<LS0, false>(wf), fbias, hx, s1, lane);
      relu(s1);
      to_ops(x1, s1);
    }
    {
      Acts<NT, 4> s2;
      layer<NT, NSF, 4, 4>(F::template seg<LS1, false>(wf), fbias + 64, x1, s2, lane);
      to_ops(x2, s2);
    }
    Ops<NT, 1, NS> gl;
    {
      Acts<NT, 1> Gl;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        Gl.v[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (g == 0 && sm.ok[t]) Gl.v[t][0][0] = d_logit[sm.n[t]];
      }
      bias_add(bH, Gl);
      to_ops(gl, Gl);
    }
    // SemanticFieldHead: G = dlogit, X = s2
    dw_round<NT, NS, 1, 4>(scr, gl, x2, accH, lane);
    Ops<NT, 4, NS> g2;
    {
      Acts<NT, 4> Gs2;
      layer<NT, NS, 4, 1>(T::template seg<LH, true>(wt), nullptr, gl, Gs2, lane);  // no activation on mlp_semantics' last layer
      bias_add(bB, Gs2);
      to_ops(g2, Gs2);
    }
    // sem1: G = Gs2, X = s1
    dw_round<NT, NS, 4, 4>(scr, g2, x1, accB, lane);
    Ops<NT, 4, NS> g1;
    {
      Acts<NT, 4> Gs1;
      layer<NT, NS, 4, 4>(T::template seg<LS1, true>(wt), nullptr, g2, Gs1, lane);
      gate(Gs1, x1);
      bias_add(bA, Gs1);
      to_ops(g1, Gs1);
    }
    // sem0: G = Gs1, X = h (input = detached geo: no dX)
    {
      Acts<NT, 1> h;  // (re-read, L2-resident, instead of its pieces living through the whole recompute)
      Ops<NT, 1, NS> hx;
#pragma unroll
      for (int t = 0; t < NT; ++t) h.v[t][0] = *reinterpret_cast<const f32x4*>(h_saved + (size_t)sm.n[t] * 16 + 4 * g);
      to_ops(hx, h);
      dw_round<NT, NS, 4, 1>(scr, g1, hx, accA, lane);
    }
  }
  const int lane = lane0;
  float* part = partials + (size_t)blockIdx.x * (Cfg::W_TOTAL + Cfg::B_TOTAL);
  struct Table {  // accA [ob] | accB [ob][ib] | accH [ib] | bias sem0 [ob] | bias sem1 [ob] | bias head
    __device__ static __forceinline__ Dest get(int b) {
      if (b < 4) return Dest{Cfg::woff(LS0), b, 0, 1, 0};
      if (b < 20) return Dest{Cfg::woff(LS1), (b - 4) >> 2, (b - 4) & 3, 4, 0};
      if (b < 24) return Dest{Cfg::woff(LH), 0, b - 20, 4, 0};
      if (b < 28) return Dest{Cfg::W_TOTAL + Cfg::boff(LS0), b - 24, 0, 0, 1};
      if (b < 32) return Dest{Cfg::W_TOTAL + Cfg::boff(LS1), b - 28, 0, 0, 1};
      return Dest{Cfg::W_TOTAL + Cfg::boff(LH), 0, 0, 0, 1};
    }
  };
  f32x4 all[33];
#pragma unroll
  for (int i = 0; i < 4; ++i) all[i] = accA[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) all[4 + i] = accB[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) all[20 + i] = accH[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) all[24 + i] = bias_total(bA[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) all[28 + i] = bias_total(bB[i]);
  all[32] = bias_total(bH[0]);
  reduce_and_store<33, WAVES, Table>(smem, part, all, wave, lane);
}

// ---- base branch -------------------------------------------------------------------------------------------------------
// POSGRAD: the input gradient of the hash grid rides along (see k_field_mlp_bwd_base_coop): d_pos [N] float4 = dL/dfeats
// contracted with the encode's saved Jacobian [L][3][N] float2, from the registers that hold dL/dfeats.
template <class Cfg, int NSF, int NS, int NT, int WAVES, bool POSGRAD>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void k_field_mlp_bwd_base_pw(
    const float* __restrict__ packed, const __bf16* __restrict__ image, int N, const float2* __restrict__ feats,
    const float* __restrict__ h_saved, const uint8_t* __restrict__ selector, const float* __restrict__ d_density,
    const float* __restrict__ d_h, float2* __restrict__ d_feats, float* __restrict__ partials,
    const float2* __restrict__ jac, float4* __restrict__ d_pos) {
  constexpr int HB = Cfg::HB;
  static_assert(HB == 1 || HB == 2, "built shapes");
  constexpr int THREADS = 64 * WAVES;
  constexpr int LB0 = Cfg::L_BASE0, LB1 = Cfg::L_BASE1;
  static_assert(LB1 == LB0 + 1, "the branch's layers are adjacent in the fp32 image");
  using L = Lds<Cfg, SegsBaseF<Cfg>, SegsBaseT<Cfg>, NSF, NS, NT, WAVES, 64>;
  using F = typename L::F;
  using T = typename L::T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16x8* wf = reinterpret_cast<bf16x8*>(smem);
  bf16x8* wt = reinterpret_cast<bf16x8*>(smem + F::BYTES);
  float* fbias = reinterpret_cast<float*>(smem + L::FB_OFF);  // base0 [64]
  F::template stage<THREADS>(wf, image);
  T::template stage<THREADS>(wt, image);
  for (int i = threadIdx.x; i < 64; i += THREADS) fbias[i] = packed[Cfg::W_TOTAL + Cfg::boff(LB0) + i];
  __syncthreads();
  const int lane0 = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char* scr = smem + L::SCR_OFF + wave * scratch_bytes<NS, NT>();
  f32x4 accB[4 * HB], accA[8];  // base1 [ob < HB][ib], base0 [ob][ib]
  f32x4 bB[HB], bA[4];
  zero_vec(accB);
  zero_vec(accA);
  zero_vec(bB);
  zero_vec(bA);
  const int n_groups = (N + 16 * NT - 1) / (16 * NT);
  for (int gr = blockIdx.x * WAVES + wave; gr < n_groups; gr += gridDim.x * WAVES) {
    asm volatile("" ::: "memory");
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int j = lane & 15, g = lane >> 4;
    const Samples<NT> sm(gr, j, N);
    Ops<NT, 2, NSF> x0;
    Ops<NT, 4, NS> x1;
    {
      Acts<NT, 2> f;
      Acts<NT, 4> a1;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m) {  // slot (g, e): level 4 (e>>1) + g, feature e & 1 (KM_HASH)
          const float2 v = feats[(size_t)(4 * m + g) * N + sm.n[t]];
          f.v[t][m >> 1][2 * (m & 1)] = v.x, f.v[t][m >> 1][2 * (m & 1) + 1] = v.y;
        }
      to_ops(x0, f);
      layer<NT, NSF, 4, 2>(F::template seg<LB0, false>(wf), fbias, x0, a1, lane);
      relu(a1);
      to_ops(x1, a1);
    }
    Ops<NT, HB, NS> gh;
    {
      Acts<NT, HB> Gh;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) Gh.v[t][hb] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (sm.ok[t]) {
#pragma unroll
          for (int hb = 0; hb < HB; ++hb) Gh.v[t][hb] = *reinterpret_cast<const f32x4*>(d_h + (size_t)sm.n[t] * (16 * HB) + 16 * hb + 4 * g);
          if (g == 0) {  // trunc_exp backward (fruit_field.py:191) on the saved density logit; the colour block has a zero row 0
            const bool sel = selector ? (selector[sm.n[t]] != 0) : true;
            const float te = expf(fminf(fmaxf(h_saved[(size_t)sm.n[t] * (16 * HB)], -15.0f), 15.0f));
            Gh.v[t][0][0] = sel ? d_density[sm.n[t]] * te : 0.0f;
          }
        }
      }
      bias_add(bB, Gh);
      to_ops(gh, Gh);
    }
    // base1: G = Gh, X = a1
    dw_round<NT, NS, HB, 4>(scr, gh, x1, accB, lane);
    Ops<NT, 4, NS> ga;
    {
      Acts<NT, 4> Ga;
      layer<NT, NS, 4, HB>(T::template seg<LB1, true>(wt), nullptr, gh, Ga, lane);
      gate(Ga, x1);
      bias_add(bA, Ga);
      to_ops(ga, Ga);
    }
    float2 jv[POSGRAD ? NT : 1][POSGRAD ? 12 : 1];
    if constexpr (POSGRAD) {  // issued ahead of the dW round and the last dX layer: consumed after them
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int a = 0; a < 3; ++a) jv[t][3 * m + a] = ntc_load<NT_JAC_LD>(&jac[((size_t)(4 * m + g) * 3 + a) * N + sm.n[t]]);   // their only use
    }
    // base0: G = Ga, X = hash features
    dw_round<NT, NS, 4, 2>(scr, ga, x0, accA, lane);
    Acts<NT, 2> Gx;
    layer<NT, NS, 2, 4>(T::template seg<LB0, true>(wt), nullptr, ga, Gx, lane);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int m = 0; m < 4; ++m)
        if (sm.ok[t])
          d_feats[(size_t)(4 * m + g) * N + sm.n[t]] = make_float2(Gx.v[t][m >> 1][2 * (m & 1)], Gx.v[t][m >> 1][2 * (m & 1) + 1]);
    if constexpr (POSGRAD) {
      // EVERY PARTIAL SUM IS PINNED IN ITS OWN REGISTER (the empty asm statements), as in k_field_mlp_bwd_base_coop: left to
      // itself hipcc may pair the x / y sums into packed-FP32 instructions threaded through the ds_bpermute shuffles, the
      // sequence that produced wrong y components there (profiles/r06_raw/nt_hunt.md; tests/test_isa_invariants.py keeps
      // packed instructions out of this reduction).
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float gp[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const float gx = Gx.v[t][m >> 1][2 * (m & 1)], gy = Gx.v[t][m >> 1][2 * (m & 1) + 1];
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            gp[a] += gx * jv[t][3 * m + a].x + gy * jv[t][3 * m + a].y;
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
        if (sm.ok[t] && g == 0) d_pos[sm.n[t]] = make_float4(gp[0], gp[1], gp[2], 0.0f);
      }
    }
  }
  const int lane = lane0;
  float* part = partials + (size_t)blockIdx.x * (Cfg::W_TOTAL + Cfg::B_TOTAL);
  constexpr int NB1 = 4 * HB, NALL = 8 + NB1 + 4 + HB;
  struct Table {  // accA [ob][ib] | accB [ob < HB][ib] | bias base0 [ob] | bias base1 [ob < HB]
    __device__ static __forceinline__ Dest get(int b) {
      constexpr int NB1 = 4 * Cfg::HB;
      if (b < 8) return Dest{Cfg::woff(LB0), b >> 1, b & 1, 2, 0};
      b -= 8;
      if (b < NB1) return Dest{Cfg::woff(LB1), b >> 2, b & 3, 4, 0};
      b -= NB1;
      if (b < 4) return Dest{Cfg::W_TOTAL + Cfg::boff(LB0), b, 0, 0, 1};
      return Dest{Cfg::W_TOTAL + Cfg::boff(LB1), b - 4, 0, 0, 1};
    }
  };
  f32x4 all[NALL];
#pragma unroll
  for (int i = 0; i < 8; ++i) all[i] = accA[i];
#pragma unroll
  for (int i = 0; i < NB1; ++i) all[8 + i] = accB[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) all[8 + NB1 + i] = bias_total(bA[i]);
#pragma unroll
  for (int i = 0; i < HB; ++i) all[12 + NB1 + i] = bias_total(bB[i]);
  reduce_and_store<NALL, WAVES, Table>(smem, part, all, wave, lane);
}

// tiles per wave, by branch (the colour branch's 116 accumulator registers leave room for one tile's activations: two spill),
// and waves per workgroup (8 = two waves per SIMD, 256 registers each)
#ifndef FNR_PW_NT_COLOR
#define FNR_PW_NT_COLOR 1
#endif
#ifndef FNR_PW_NT_SEM
#define FNR_PW_NT_SEM 2
#endif
#ifndef FNR_PW_NT_BASE
#define FNR_PW_NT_BASE 2
#endif
#ifndef FNR_PW_NT_BASE_BIG
#define FNR_PW_NT_BASE_BIG 1
#endif
#ifndef FNR_PW_WAVES
#define FNR_PW_WAVES 8
#endif

template <class Cfg, int NSF, int NS>
static int launch(int branch, const float* packed, const __bf16* image, const float* ray_bias, const RaysDev& rd, int S, long long N,
                  const float2* feats, const float* h_saved, const uint8_t* selector, const float* d_density, const float* d_rgb,
                  const float* d_logit, float2* d_feats, float* d_h, float* gsum_tile, float* gsum_extra, float* partials,
                  long long blocks, hipStream_t st, const float2* jac, float4* d_pos) {
  constexpr int WAVES = FNR_PW_WAVES, THREADS = 64 * WAVES;
  FNR_CHECK_ARG(N < (1ll << 31) - 64, "field_mlp_bwd: %lld samples exceed the 32-bit sample index of the backward kernels", N);
  const int n = (int)N;
  // exactly `blocks` workgroups: every one of the caller's partial images receives this branch's blocks (a workgroup
  // without samples stores zeros)
  if (branch == 0) {
    constexpr int NT = FNR_PW_NT_COLOR;
    using L = Lds<Cfg, SegsColF<Cfg>, SegsColT<Cfg>, NSF, NS, NT, WAVES, 80>;
    static_assert(L::BYTES <= 160 * 1024, "colour branch exceeds the LDS");
    auto kern = k_field_mlp_bwd_color_pw<Cfg, NSF, NS, NT, WAVES>;
    const int once = ensure_dyn_lds(kern, L::BYTES);
    if (once) return once;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(THREADS), L::BYTES, st, packed, image, ray_bias, rd, S, n, h_saved,
                       d_rgb, d_h, gsum_tile, gsum_extra, partials);
  } else if (branch == 1) {
    if constexpr (Cfg::NSEM == 2) {
      constexpr int NT = NSF == 1 ? 1 : FNR_PW_NT_SEM;  // (plain bf16: hipcc's schedule of the two-tile form spills 8 registers)
      using L = Lds<Cfg, SegsSemF<Cfg>, SegsSemT<Cfg>, NSF, NS, NT, WAVES, 128>;
      static_assert(L::BYTES <= 160 * 1024, "semantic branch exceeds the LDS");
      auto kern = k_field_mlp_bwd_sem_pw<Cfg, NSF, NS, NT, WAVES>;
      const int once = ensure_dyn_lds(kern, L::BYTES);
      if (once) return once;
      hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(THREADS), L::BYTES, st, packed, image, n, h_saved, d_logit, partials);
    } else {
      FNR_CHECK_ARG(false, "the fruit_nerf_big semantic branch has its own kernel (field_mlp_bwd_sem_big_bf16)");
    }
  } else {
    constexpr int NT = Cfg::HB == 1 ? FNR_PW_NT_BASE : FNR_PW_NT_BASE_BIG;
    using L = Lds<Cfg, SegsBaseF<Cfg>, SegsBaseT<Cfg>, NSF, NS, NT, WAVES, 64>;
    static_assert(L::BYTES <= 160 * 1024, "base branch exceeds the LDS");
    if (jac && d_pos) {
      auto kern = k_field_mlp_bwd_base_pw<Cfg, NSF, NS, NT, WAVES, true>;
      const int once = ensure_dyn_lds(kern, L::BYTES);
      if (once) return once;
      hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(THREADS), L::BYTES, st, packed, image, n, feats, h_saved, selector,
                         d_density, d_h, d_feats, partials, jac, d_pos);
    } else {
      auto kern = k_field_mlp_bwd_base_pw<Cfg, NSF, NS, NT, WAVES, false>;
      const int once = ensure_dyn_lds(kern, L::BYTES);
      if (once) return once;
      hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(THREADS), L::BYTES, st, packed, image, n, feats, h_saved, selector,
                         d_density, d_h, d_feats, partials, jac, d_pos);
    }
  }
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

}  // namespace pw

// cfg: 0 `fruit_nerf` (branch: 0 colour, 1 semantic, 2 base), 1 `fruit_nerf_big` (colour and base; its 128-wide semantic branch
// is field_mlp_bwd_sem_big_bf16) — the backward in the bf16-pipe modes (called by field_mlp_bwd_bf16, which has packed the
// fragment image)
int field_mlp_bwd_pw(int cfg, int mode, int branch, const float* packed, const __bf16* image, const float* ray_bias, const RaysDev& rd,
                     int S, long long N, const float2* feats, const float* h_saved, const uint8_t* selector,
                     const float* d_density, const float* d_rgb, const float* d_logit, float2* d_feats, float* d_h,
                     float* gsum_tile, float* gsum_extra, float* partials, long long blocks, hipStream_t st, const float2* jac,
                     float4* d_pos) {
#define FNR_PW_LAUNCH(C, A, B)                                                                                                      \
  pw::launch<C, A, B>(branch, packed, image, ray_bias, rd, S, N, feats, h_saved, selector, d_density, d_rgb, d_logit, d_feats, d_h, \
                      gsum_tile, gsum_extra, partials, blocks, st, jac, d_pos)
  if (cfg == 0) return mode == MLP_BF16 ? FNR_PW_LAUNCH(FieldCfgBase, 1, 1) : FNR_PW_LAUNCH(FieldCfgBase, 3, 2);
  return mode == MLP_BF16 ? FNR_PW_LAUNCH(FieldCfgBig, 1, 1) : FNR_PW_LAUNCH(FieldCfgBig, 3, 2);
#undef FNR_PW_LAUNCH
}

}  // namespace fnr
