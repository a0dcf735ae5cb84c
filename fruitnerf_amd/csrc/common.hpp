// common.hpp — shared device/host helpers for the gfx950 FruitNeRF kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/fruitnerf_hip.h"

// hipcc defaults to -ffp-contract=fast and HIP's __fmul_rn/__fadd_rn are plain operators, so a*b+c (and
// x*s - floor(x*s)!) would be fused into FMAs.  The oracle (PyTorch CPU eager) rounds after every op and
// discrete decisions (cell index vs. offset, selector, thresholds) depend on those roundings: contraction
// is off for every translation unit; FMAs are written explicitly (fmaf / MFMA) where they are wanted.
#pragma clang fp contract(off)

namespace fnr {

// ---- error plumbing ---------------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define FNR_CHECK_ARG(cond, ...)          \
  do {                                    \
    if (!(cond)) {                        \
      ::fnr::set_error(__VA_ARGS__);      \
      return FNR_ERR_INVALID;             \
    }                                     \
  } while (0)

#define FNR_UNSUPPORTED(cond, ...)        \
  do {                                    \
    if (!(cond)) {                        \
      ::fnr::set_error(__VA_ARGS__);      \
      return FNR_ERR_UNSUPPORTED;         \
    }                                     \
  } while (0)

#define FNR_HIP(expr)                                                                       \
  do {                                                                                      \
    hipError_t e__ = (expr);                                                                \
    if (e__ != hipSuccess) {                                                                \
      ::fnr::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      return FNR_ERR_HIP;                                                                   \
    }                                                                                       \
  } while (0)

#define FNR_LAUNCH_CHECK() FNR_HIP(hipGetLastError())

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// number of CUs of the current device (cached)
int device_cu_count();

// ---- optional per-entry-point HIP-event timing (fnr_profile_enable / fnr_profile_collect) -------
// Brackets everything an entry point enqueues with two events on the SAME stream the kernels run on.
enum ProfOp : int {
  OP_SAMPLE_SPACED = 0, OP_WEIGHTS_PDF, OP_PROP_FWD, OP_ENCODE_FWD, OP_ENCODE_LATTICE, OP_MLP_FWD, OP_COMPOSITE_FWD,
  OP_LOSSES, OP_INTERLEVEL, OP_DISTORTION, OP_COMPOSITE_BWD, OP_WEIGHTS_BWD, OP_MLP_BWD, OP_ENCODE_BWD, OP_PROP_BWD,
  OP_ADAM, OP_EXPORT_COMPACT, OP_POSITION_GRAD, OP_CLOUD, OP_COUNT
};
// Dynamic LDS above 64 KB needs hipFuncAttributeMaxDynamicSharedMemorySize — a PER-DEVICE attribute of the kernel.
// Set once per (kernel instantiation, device); a failure is reported and retried on the next call, never cached.
template <class K>
static inline int ensure_dyn_lds(K kernel, int bytes) {
  static unsigned long long done_mask = 0ull;  // one bit per device id < 64 (one static per kernel instantiation)
  int dev = 0;
  FNR_HIP(hipGetDevice(&dev));
  if (dev < 64 && ((done_mask >> dev) & 1ull)) return FNR_OK;
  FNR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  if (dev < 64) done_mask |= 1ull << dev;
  return FNR_OK;
}

struct ProfScope {
  ProfScope(int op, long long units, void* stream);
  ~ProfScope();
  int slot;
  hipStream_t st;
};
#define FNR_PROF(op, units) ::fnr::ProfScope prof_scope__((op), (long long)(units), stream)

// ---- streaming (`nt`) accesses -------------------------------------------------------------------------------------
// For data a step writes once and reads once (the optimiser moments, the encode's input Jacobian, the record queues on
// their way back): without the hint those streams evict the hash table's parameters from the L2s / the Infinity Cache and
// the next encode's gathers go to HBM (round 5, profiles/r05_raw/kt_nt_call6.log: k_hash_encode 82 -> 65 us with the hints).
// NOT for SCATTERED narrow stores: the emit kernel's queue stores (an 8-byte value pair + a 2-byte row per record, a few
// records per bin and wave) as `nt` are one fabric write each: 75 -> 120 us.  Coalesced 8-byte `nt` stores — a wave writes one
// contiguous 512-byte run: NT_JAC_ST, NT_PROP_* — combine into full lines and are fine.
typedef float nt_f4 __attribute__((ext_vector_type(4)));
typedef float nt_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 nt_load(const float4* p) {
  const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float2 nt_load(const float2* p) {
  const nt_f2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f2*>(p));
  return make_float2(v.x, v.y);
}
__device__ __forceinline__ ushort2 nt_load(const ushort2* p) {
  const unsigned v = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(p));
  return make_ushort2((unsigned short)(v & 0xffffu), (unsigned short)(v >> 16));
}
__device__ __forceinline__ void nt_store(float4* p, const float4& v) {
  const nt_f4 t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<nt_f4*>(p));
}
__device__ __forceinline__ void nt_store(float2* p, const float2& v) {
  const nt_f2 t = {v.x, v.y};
  __builtin_nontemporal_store(t, reinterpret_cast<nt_f2*>(p));
}
// Class c is streaming iff bit c of FNR_NT_MASK is set.  NT_JAC_LD is off because it buys nothing (same-box A/B, round 6:
// 5.72 vs 5.74 M rays/s).  Round 5 switched it off for another reason — with the Jacobian's loads in
// k_field_mlp_bwd_base_coop streaming, training stopped being bit-reproducible — and suspected the loads ("may `nt` loads
// complete out of order with plain ones at a partial s_waitcnt vmcnt(n)?").  Round 6 measured: they may not
// (tools/microbench/nt_load_order.hip, nt_visibility.hip: 0 events in 1e11 partial waits / 300 buffer rewrites per mode, across
// policies, stores, address-register reuse, a streaming second stream); the defect sat in that kernel's position-gradient
// reduction, in an instruction selection hipcc only makes in the schedule the `nt` loads give it, and is fenced there
// (field_mlp_bf16.hip "EVERY PARTIAL SUM IS PINNED"; profiles/r06_raw/nt_hunt.md).  Partial waits with accesses of both
// policies in flight are everywhere in this library (tools/isa_nt_scan.py: 71 places) and are fine.
#ifndef FNR_NT_MASK
#define FNR_NT_MASK 0xef
#endif
enum NtClass { NT_QUEUE_LD = 0, NT_MOMENT_LD = 1, NT_MOMENT_ST = 2, NT_JAC_ST = 3, NT_JAC_LD = 4, NT_DFEATS_LD = 5, NT_PROP_FEATS = 6, NT_PROP_DFEATS_ST = 7 };
template <int CLASS, class T>
__device__ __forceinline__ T ntc_load(const T* p) {
  if constexpr ((FNR_NT_MASK >> CLASS) & 1) return nt_load(p);
  else return *p;
}
template <int CLASS, class T>
__device__ __forceinline__ void ntc_store(T* p, const T& v) {
  if constexpr ((FNR_NT_MASK >> CLASS) & 1) nt_store(p, v);
  else *p = v;
}

// ---- exact (non-contracted) fp32 arithmetic -----------------------------------------------------
// The oracle (PyTorch CPU eager) rounds after every elementwise op.  Wherever a discrete decision
// depends on the value (selector mask, floor/ceil cell, searchsorted) we use these so the HIP path
// produces the same bits.
__device__ __forceinline__ float fmul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fadd(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float fsub(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float fdiv(float a, float b) { return __fdiv_rn(a, b); }

// torch.nan_to_num defaults: nan->0, +inf->FLT_MAX, -inf->-FLT_MAX
__device__ __forceinline__ float nan_to_num(float x) {
  if (x != x) return 0.0f;
  if (x == __builtin_inff()) return 3.4028234663852886e38f;
  if (x == -__builtin_inff()) return -3.4028234663852886e38f;
  return x;
}

// ---- wave (64-lane) primitives -------------------------------------------------------------------
__device__ __forceinline__ float wave_incl_scan(float v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    float t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}
// exclusive prefix sum: the inclusive scan shifted by one lane.  NOT `inclusive - own`: when the own term dwarfs the
// prefix (delta * sigma ~ 1e8 behind a prefix of ~0.6 on a sharp surface) that subtraction cancels the prefix to 0,
// the transmittance becomes 1 and the sample gets weight 1 on top of the earlier weights — the gradient spike that
// blew training up after a few thousand steps.
__device__ __forceinline__ float wave_excl_scan(float v, int lane) {
  const float incl = wave_incl_scan(v, lane);
  const float up = __shfl_up(incl, 1, 64);
  return lane == 0 ? 0.0f : up;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
__device__ __forceinline__ float wave_bcast(float v, int src) { return __shfl(v, src, 64); }
// per-lane source (lane permutation), e.g. 63 - lane reverses the wave
__device__ __forceinline__ float wave_bcast_lane(float v, int src_lane) { return __shfl(v, src_lane, 64); }

// ---- position warp (fruit_field.py:168-179) -------------------------------------------------------
struct Warp {
  int mode;
  float lo[3];
  float len[3];
};
static inline Warp make_warp(const fnr_warp* w) {
  Warp r;
  r.mode = w->mode;
  for (int i = 0; i < 3; ++i) {
    r.lo[i] = w->aabb[i];
    r.len[i] = w->aabb[3 + i] - w->aabb[i];  // fp32, like aabb[1] - aabb[0]
  }
  return r;
}

// world position -> masked unit-cube position; returns selector
__device__ __forceinline__ bool warp_position(const Warp& w, float px, float py, float pz, float (&x)[3]) {
  if (w.mode == 0) {
    float mag = fmaxf(fabsf(px), fmaxf(fabsf(py), fabsf(pz)));
    if (!(mag < 1.0f)) {
      float s = fsub(2.0f, fdiv(1.0f, mag));
      px = fmul(s, fdiv(px, mag));
      py = fmul(s, fdiv(py, mag));
      pz = fmul(s, fdiv(pz, mag));
    }
    x[0] = fdiv(fadd(px, 2.0f), 4.0f);
    x[1] = fdiv(fadd(py, 2.0f), 4.0f);
    x[2] = fdiv(fadd(pz, 2.0f), 4.0f);
  } else {
    x[0] = fdiv(fsub(px, w.lo[0]), w.len[0]);
    x[1] = fdiv(fsub(py, w.lo[1]), w.len[1]);
    x[2] = fdiv(fsub(pz, w.lo[2]), w.len[2]);
  }
  bool sel = (x[0] > 0.0f) && (x[0] < 1.0f) && (x[1] > 0.0f) && (x[1] < 1.0f) && (x[2] > 0.0f) && (x[2] < 1.0f);
  if (!sel) {
    x[0] = 0.0f;
    x[1] = 0.0f;
    x[2] = 0.0f;
  }
  return sel;
}

// Frustums.get_positions(): origins + directions * (starts + ends) / 2
__device__ __forceinline__ void ray_position(const float* __restrict__ o, const float* __restrict__ d, float t0,
                                             float t1, float& px, float& py, float& pz) {
  float s = fadd(t0, t1);
  px = fadd(o[0], fdiv(fmul(d[0], s), 2.0f));
  py = fadd(o[1], fdiv(fmul(d[1], s), 2.0f));
  pz = fadd(o[2], fdiv(fmul(d[2], s), 2.0f));
}

// ---- hash grid (HashEncoding torch path) -------------------------------------
struct GridLevel {
  uint32_t c[3], f[3];
  float o[3];
};
__device__ __forceinline__ GridLevel grid_cell(const float (&x)[3], int scaling) {
  GridLevel g;
  const float s = (float)scaling;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float sc = fmul(x[a], s);
    float fl = floorf(sc);
    g.c[a] = (uint32_t)(int)ceilf(sc);
    g.f[a] = (uint32_t)(int)fl;
    g.o[a] = fsub(sc, fl);
  }
  return g;
}
__device__ __forceinline__ uint32_t grid_hash(uint32_t x, uint32_t y, uint32_t z, uint32_t mask) {
  return (x ^ (y * 2654435761u) ^ (z * 805459861u)) & mask;
}
// the 8 corner rows in the oracle's order h0..h7
__device__ __forceinline__ void grid_corners(const GridLevel& g, uint32_t mask, uint32_t (&h)[8]) {
  h[0] = grid_hash(g.c[0], g.c[1], g.c[2], mask);
  h[1] = grid_hash(g.c[0], g.f[1], g.c[2], mask);
  h[2] = grid_hash(g.f[0], g.f[1], g.c[2], mask);
  h[3] = grid_hash(g.f[0], g.c[1], g.c[2], mask);
  h[4] = grid_hash(g.c[0], g.c[1], g.f[2], mask);
  h[5] = grid_hash(g.c[0], g.f[1], g.f[2], mask);
  h[6] = grid_hash(g.f[0], g.f[1], g.f[2], mask);
  h[7] = grid_hash(g.f[0], g.c[1], g.f[2], mask);
}
// trilinear blend in the oracle's op order (mul, mul, add — no fma)
__device__ __forceinline__ float lerp_nf(float a, float b, float o, float om) { return fadd(fmul(a, o), fmul(b, om)); }
__device__ __forceinline__ float2 grid_interp(const float2 (&v)[8], const float (&o)[3]) {
  const float omx = fsub(1.0f, o[0]), omy = fsub(1.0f, o[1]), omz = fsub(1.0f, o[2]);
  float2 r;
  {
    float f03 = lerp_nf(v[0].x, v[3].x, o[0], omx), f12 = lerp_nf(v[1].x, v[2].x, o[0], omx);
    float f56 = lerp_nf(v[5].x, v[6].x, o[0], omx), f47 = lerp_nf(v[4].x, v[7].x, o[0], omx);
    float f0312 = lerp_nf(f03, f12, o[1], omy), f4756 = lerp_nf(f47, f56, o[1], omy);
    r.x = lerp_nf(f0312, f4756, o[2], omz);
  }
  {
    float f03 = lerp_nf(v[0].y, v[3].y, o[0], omx), f12 = lerp_nf(v[1].y, v[2].y, o[0], omx);
    float f56 = lerp_nf(v[5].y, v[6].y, o[0], omx), f47 = lerp_nf(v[4].y, v[7].y, o[0], omx);
    float f0312 = lerp_nf(f03, f12, o[1], omy), f4756 = lerp_nf(f47, f56, o[1], omy);
    r.y = lerp_nf(f0312, f4756, o[2], omz);
  }
  return r;
}
// ---- corner gathers by lane PAIRS ----------------------------------------------------------------
// A wave-wide 8-byte gather costs the texture path one tag lookup per distinct 128-byte line, and that (not HBM or L2
// bandwidth) bounds the encode kernels: with one sample per lane every instruction touches 64 lines.  The two
// x-neighbours of a cell edge — corners (0,3), (1,2), (4,7), (5,6) — sit in the same line 15 times out of 16 (dense
// levels: consecutive rows; hashed levels: x enters the hash with multiplier 1, so x -> x + 1 only changes low row
// bits).  So lanes 2i and 2i+1 gather TOGETHER: first the four pairs of the even lane's sample, then the four pairs of
// the odd lane's, each lane taking one side — 64 lanes, ~34 lines per instruction — and swap what belongs to the
// other (DPP quad_perm, no LDS).  Values and blend order are unchanged (bit-identical features).  Measured: k_hash_encode
// alone on spread-out samples 84 -> 71 us; in the training step nothing at initialisation (the table arrives cold from
// HBM there) and +2.6 % rays/s at a trained state, eval +3 %, 256^3 export +10 %.  The proposal networks' gathers
// (coarse levels, 256 closely spaced samples per ray: neighbouring lanes already share lines) got 5-10 % SLOWER with it
// and keep one sample per lane.  Needs every lane of the wave active (no early return before it).
__device__ __forceinline__ uint32_t lane_swap1(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float2 lane_swap1(float2 v) {
  return make_float2(__uint_as_float(lane_swap1(__float_as_uint(v.x))), __uint_as_float(lane_swap1(__float_as_uint(v.y))));
}
struct PairedRows {
  uint32_t a[4], b[4];  // rows this lane reads for the even lane's sample / for the odd lane's sample
};
__device__ __forceinline__ PairedRows paired_rows(const uint32_t (&h)[8], bool odd) {
  constexpr int KA[4] = {0, 1, 4, 5}, KB[4] = {3, 2, 7, 6};
  PairedRows r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t recv = lane_swap1(odd ? h[KA[j]] : h[KB[j]]);
    r.a[j] = odd ? recv : h[KA[j]];
    r.b[j] = odd ? h[KB[j]] : recv;
  }
  return r;
}
__device__ __forceinline__ void paired_values(const float2 (&va)[4], const float2 (&vb)[4], bool odd, float2 (&v)[8]) {
  constexpr int KA[4] = {0, 1, 4, 5}, KB[4] = {3, 2, 7, 6};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float2 recv = lane_swap1(odd ? va[j] : vb[j]);
    v[KA[j]] = odd ? recv : va[j];
    v[KB[j]] = odd ? vb[j] : recv;
  }
}
// one level: gather 8 corners and blend
__device__ __forceinline__ float2 grid_lookup(const float2* __restrict__ level_table, const float (&x)[3], int scaling,
                                              uint32_t mask) {
  GridLevel g = grid_cell(x, scaling);
  uint32_t h[8];
  grid_corners(g, mask, h);
  float2 v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = level_table[h[k]];
  return grid_interp(v, g.o);
}

struct GridDev {
  int n_levels;
  int log2_T;
  int scalings[FNR_MAX_LEVELS];
  float2* table;
};
static inline GridDev make_grid(const fnr_grid* g) {
  GridDev d;
  d.n_levels = g->n_levels;
  d.log2_T = g->log2_hashmap_size;
  for (int i = 0; i < FNR_MAX_LEVELS; ++i) d.scalings[i] = g->scalings[i];
  d.table = reinterpret_cast<float2*>(g->table);
  return d;
}

struct RaysDev {
  long long n_rays;
  const float* origins;
  const float* directions;
  const float* nears;
  const float* fars;
  const int32_t* cam;
};
static inline RaysDev make_rays(const fnr_rays* r) {
  RaysDev d{r->n_rays, r->origins, r->directions, r->nears, r->fars, r->camera_indices};
  return d;
}


// Optimiser step fused into the kernel that produces a gradient (single-process training).  Hash table: the accumulate
// workgroup that owns a bin has the bin's summed gradient in LDS, so it applies torch.optim.Adam / RAdam to those rows
// right there (hash_scatter.hip); camera poses: the workgroup of a camera finishes its 6 gradients (camera_opt.hip).  The gradient table
// is then neither read-modify-written here nor read and zeroed by the optimiser pass: 40 -> 24 bytes of HBM traffic
// per table parameter and step.  Same operations in the same order as k_adam / k_radam (train.hip): bit-identical
// parameters and moments.
struct TableAdam {
  float2 *p, *m, *v;  // the table's slices of the parameter / exp_avg / exp_avg_sq arenas, [L << log2_T] rows
  float lr, b1, b2, eps, bc1, bc2_sqrt, rect, grad_scale, weight_decay;
  int radam;
  unsigned* touched;  // one bit per pair of rows: ever received a gradient (nullable; fnr_table_adam.touched)
};
__device__ __forceinline__ void table_adam_update(const TableAdam& a, float g, float& P, float& M, float& V) {
  float gr = g * a.grad_scale;
  if (a.weight_decay != 0.0f) gr = gr + a.weight_decay * P;
  M = M + (gr - M) * (1.0f - a.b1);
  V = V * a.b2 + (1.0f - a.b2) * gr * gr;
  if (!a.radam) {
    const float step_size = a.lr / a.bc1;
    const float denom = sqrtf(V) / a.bc2_sqrt + a.eps;
    P = P - step_size * (M / denom);
  } else {
    const float mhat = M / a.bc1;
    if (a.rect >= 0.0f) {
      const float adaptive = a.bc2_sqrt / (sqrtf(V) + a.eps);
      P = P - a.lr * (mhat * a.rect * adaptive);
    } else {
      P = P - a.lr * mhat;
    }
  }
}
// The optimiser step of a parameter taken by the thread that finishes its gradient entry (k_reduce_dw,
// k_embedding_grad, k_prop_reduce: single writers): the parameter / moment arrays parallel the gradient arena.
struct WeightAdam {
  TableAdam t;               // hyper-parameters and step-dependent scalars (its p / m / v pointers are unused here)
  long long p_off, m_off, v_off;   // element offsets from a GRADIENT address to the parameter / exp_avg / exp_avg_sq entry
};
__device__ __forceinline__ void weight_adam_entry(const WeightAdam& wa, float* __restrict__ g_entry, float s) {
  const float g = *g_entry + s;
  float P = g_entry[wa.p_off], M = g_entry[wa.m_off], V = g_entry[wa.v_off];
  table_adam_update(wa.t, g, P, M, V);
  g_entry[wa.p_off] = P;
  g_entry[wa.m_off] = M;
  g_entry[wa.v_off] = V;
  *g_entry = 0.0f;
}
// host side of TableAdam: the step-dependent scalars exactly as fnr_adam_step / fnr_radam_step compute them (double)
static inline int make_table_adam(const fnr_table_adam* a, TableAdam& t) {
  FNR_CHECK_ARG(a && a->params && a->exp_avg && a->exp_avg_sq, "table adam: null argument");
  FNR_CHECK_ARG(a->algorithm == 0 || a->algorithm == 1, "table adam: algorithm %d (0 = Adam, 1 = RAdam)", a->algorithm);
  FNR_CHECK_ARG(a->step >= 1, "table adam: step must be >= 1");
  t.p = reinterpret_cast<float2*>(a->params);
  t.m = reinterpret_cast<float2*>(a->exp_avg);
  t.v = reinterpret_cast<float2*>(a->exp_avg_sq);
  t.lr = a->lr, t.b1 = a->beta1, t.b2 = a->beta2, t.eps = a->eps, t.grad_scale = a->grad_scale, t.weight_decay = a->weight_decay;
  t.radam = a->algorithm;
  t.touched = a->touched;
  FNR_CHECK_ARG(!a->touched || a->weight_decay == 0.0f, "table adam: sparse-touch skipping needs weight_decay = 0");
  const double bc1 = 1.0 - pow((double)a->beta1, (double)a->step);
  const double b2t = pow((double)a->beta2, (double)a->step);
  const double bc2 = 1.0 - b2t;
  t.bc1 = (float)bc1;
  t.bc2_sqrt = (float)sqrt(bc2);
  t.rect = -1.0f;
  if (a->algorithm == 1) {
    const double rho_inf = 2.0 / (1.0 - (double)a->beta2) - 1.0;
    const double rho_t = rho_inf - 2.0 * (double)a->step * b2t / bc2;
    if (rho_t > 5.0)
      t.rect = (float)sqrt((rho_t - 4.0) * (rho_t - 2.0) * rho_inf / ((rho_inf - 4.0) * (rho_inf - 2.0) * rho_t));
  }
  return FNR_OK;
}

}  // namespace fnr
