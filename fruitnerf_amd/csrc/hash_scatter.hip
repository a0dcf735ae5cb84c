// hash_scatter.hip — backward of the hash-grid lookup (autograd of HashEncoding's 8-corner blend) and of the
// proposal networks, for gfx950.
//
// Measured on MI355X: global fp32 atomic adds run at ~21 G atomics/s for the whole chip, independent of table
// size, scope or XCD locality.  A straight atomicAdd scatter (8 corners x 2 features per sample and level) needs
// 165 M atomics per 4096-ray training step = ~8 ms, 20x the whole forward pass.  So the scatter is BINNED:
//   1. emit       every contribution (row, w*gx, w*gy) is appended to the queue of the bin that owns its table
//                 row (a bin = E consecutive rows of one level, E*8 B <= 64 KiB); a workgroup counts its
//                 contributions per bin in LDS, reserves queue space with ONE global atomic per non-empty bin,
//                 then writes 10-byte records (value pair + 16-bit row inside the bin);
//   2. accumulate one workgroup per bin sums its queue into LDS (64-bit block fixed point, see below) and adds the
//                 dense E-row tile to the gradient table with plain coalesced read-modify-writes (it is the
//                 only writer of those rows).
// Queue overflow (a pathologically hot bin) falls back to global atomics in step 1, so results never depend on
// the capacity heuristic.  HBM-bound: 10 B written + read per contribution (value pair + 16-bit row) instead of two
// serialized atomics.
#include <stdlib.h>
#include <string.h>

#include "hash_sources.hpp"
#include "sequencer.hpp"

namespace fnr {

constexpr int SC_MAX_ROWS = 8192;       // rows per bin (64 KiB of float2 in LDS)
constexpr int SC_MAX_BINS = 256;        // per level (LDS histogram size): T = 2^21 (fruit_nerf_big) has 256 bins of 8192 rows
// samples per emit workgroup: 512 x 8 records x 12 B (value pair + packed row|bin) = 48 KiB of LDS, + 3 KiB of bin
// tables = 3 workgroups (24 waves) per CU; with 16-byte records it was 2, and the kernel spends half its wave-cycles
// waiting (reservation atomics, barriers between its phases) with only the other workgroup to fill in
constexpr int SC_CHUNK = 512;
constexpr int SC_EMIT_THREADS = 512;    // 8 waves, one sample per thread
constexpr int SC_PER_THREAD = SC_CHUNK / SC_EMIT_THREADS;
constexpr int SC_BINS_PER_THREAD = 1;    // thread t < bins owns bin t in the scan / reservation step
static_assert(SC_MAX_BINS <= SC_EMIT_THREADS, "one bin per thread");
// Every bin's queue counter (and its max-|v| word) sits in its own 128-byte line: all workgroups of a level hit the
// same 64 counters, and atomics to one L2 line serialise (~12 ns each) — packed 4 bytes apart, 64 counters shared
// two lines and the reservation step alone cost ~150 us per call.
constexpr int SC_CNT_STRIDE = 32;       // uint32 words between two bins' counters

__device__ unsigned long long g_scatter_overflow_records;   // records that went to the table through global atomics
// records summed by the accumulate kernels since the last reset, per kind of call (0: the field's table, 1: proposal
// tables); 64 slots per kind, each in its own 128-byte line (one atomic per accumulate workgroup, spread by workgroup id:
// same-line atomics serialise at ~12 ns).  fnr_debug_scatter_records: bench.py prices the record queue's round trip with it.
__device__ unsigned long long g_scatter_records[2][64][16];


// Streaming hints (common.hpp: nt_load / nt_store; round 5, +6.5 .. 7.5 % rays/s together with the encode's Jacobian,
// profiles/r05_raw/kt_nt_call7.log, ab_quick_call7.log) on what this file reads or writes exactly once per step: the
// accumulate kernel's loads of the record queues, the optimiser sweep's loads and stores of exp_avg / exp_avg_sq, the emit
// kernel's loads of d_feats, the proposal networks' saved features and d_feats.  The queue STORES stay plain: 2- and 8-byte
// `nt` stores are one fabric write each (emit 75 -> 120 us).

struct ScatterPlan {
  int log2_rows;         // log2(E)
  int bins_per_level;    // T / E
  long long cap;         // queue capacity per bin (records)
  size_t count_bytes, queue_bytes;  // queue = values float2 [nbins][cap] followed by rows uint16 [nbins][cap]
  size_t value_bytes;
};

static ScatterPlan scatter_plan(long long N, int n_levels, int log2_T) {
  ScatterPlan p;
  int log2_rows = log2_T - 5;  // at least 32 bins per level (measured best for the 2^17-row proposal tables) ...
  if (log2_rows > 13) log2_rows = 13;  // ... but at most 8192 rows per bin
  if (log2_rows < 0) log2_rows = 0;
  p.log2_rows = log2_rows;
  p.bins_per_level = 1 << (log2_T - log2_rows);
  const long long avg = (N * 8 + p.bins_per_level - 1) / p.bins_per_level;
  p.cap = (3 * avg + 1023) / 1024 * 1024;
  if (p.cap < 1024) p.cap = 1024;
  const size_t nbins = (size_t)n_levels * p.bins_per_level;
  // [nbins] queue counts, then per level: max |v| bits and a 'bins done' counter, each in its own 128-byte line
  p.count_bytes = (nbins + 2 * (size_t)n_levels) * SC_CNT_STRIDE * sizeof(unsigned);
  // 10 bytes per record in HBM (8-byte value pair + 16-bit row inside the bin; rows per bin <= 8192)
  p.value_bytes = nbins * (size_t)p.cap * sizeof(float2);
  p.queue_bytes = p.value_bytes + (nbins * (size_t)p.cap * sizeof(unsigned short) + 255) / 256 * 256;
  return p;
}

// the 8 (row, weight) pairs of one sample at one level, in the oracle's corner order
__device__ __forceinline__ GridLevel corner_weights(const float (&x)[3], int scaling, uint32_t mask, uint32_t (&h)[8],
                                                    float (&wgt)[8]) {
  GridLevel g = grid_cell(x, scaling);
  grid_corners(g, mask, h);
  const float ox = g.o[0], oy = g.o[1], oz = g.o[2];
  const float mx = 1.0f - ox, my = 1.0f - oy, mz = 1.0f - oz;
  wgt[0] = ox * oy * oz;
  wgt[1] = ox * my * oz;
  wgt[2] = mx * my * oz;
  wgt[3] = mx * oy * oz;
  wgt[4] = ox * oy * mz;
  wgt[5] = ox * my * mz;
  wgt[6] = mx * my * mz;
  wgt[7] = mx * oy * mz;
  return g;
}

// DPP lane movement (VALU only).  The emit kernel was LDS-bound on ds_bpermute: 20 shuffles per corner x 8 corners
// per thread kept the LDS crossbar ~80 % busy (SQ_LDS_IDX_ACTIVE), four times the traffic of the record staging.
//   row_shr:n = 0x110+n (lane i <- i-n inside its row of 16), row_bcast:15 = 0x142 / row_bcast:31 = 0x143
//   (last lane of a row -> the next row / the upper half), wave_shr:1 = 0x138, wave_shl:1 = 0x130.
// Lanes without a source keep `old`.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_f32(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                 CTRL, ROW_MASK, 0xf, false));
}

// Runs of consecutive samples (lanes) that sit in the same grid cell share all 8 corner rows, so their
// contributions are pre-summed and only the last lane of a run emits.  The run structure is found ONCE per
// (sample, level) from the exact cell coordinates and cut at rows of 16 lanes: a lane's distance to its run head
// gives four masks, and each of the 16 value streams (8 corners x 2 features) is a 4-step Kogge-Stone scan of
// row_shr DPP adds gated by those masks (3 VALU per step; the previous per-corner key/flag scan cost ~45 per
// corner and made this kernel VALU-bound).
struct RunMasks {
  float g1, g2, g4, g8;   // 1.0f where the lane takes its 1 / 2 / 4 / 8-lanes-left neighbour's partial sum, else 0.0f
  bool tail;
  bool any_run;  // wave-uniform: some lane continues its left neighbour's run
};
// lane i <- lane i - n inside its row of 16, 0 where there is no such lane (bound_ctrl: the combiner can then fold the
// move into the instruction that uses it)
template <int CTRL>
__device__ __forceinline__ float dpp_shr0_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ RunMasks run_structure(uint32_t key_a, uint32_t key_b, int lane) {
  const int j = lane & 15;
  const uint32_t pa = dpp_u32<0x111>(~key_a, key_a), pb = dpp_u32<0x111>(key_b, key_b);  // j == 0: pa != key_a
  const bool head = (pa != key_a) || (pb != key_b);
  const unsigned long long heads = __ballot(head);
  const uint32_t row = (uint32_t)(heads >> (lane & 48)) & 0xffffu;  // bit 0 (row start) is always set
  const uint32_t below = row & ((2u << j) - 1u);
  const int dist = j - (31 - __clz((int)below));
  RunMasks m;
  m.g1 = dist >= 1 ? 1.0f : 0.0f;
  m.g2 = dist >= 2 ? 1.0f : 0.0f;
  m.g4 = dist >= 4 ? 1.0f : 0.0f;
  m.g8 = dist >= 8 ? 1.0f : 0.0f;
  m.tail = (j == 15) || ((row >> (j + 1)) & 1u);
  m.any_run = ~heads != 0ull;
  return m;
}
__device__ __forceinline__ float run_sum(float v, const RunMasks& m) {
  // the DPP reads are executed by ALL lanes, then gated: inside a divergent branch the disabled source lanes
  // would read as invalid
  // one gated step = v + g * shifted(v): a 0 / 1 gate makes the product exact, so the sums are those of the select form
  // (finite values); one v_fmac_f32 with a DPP operand per step instead of mov + mov_dpp + cndmask + add
  v = fmaf(m.g1, dpp_shr0_f32<0x111>(v), v);
  v = fmaf(m.g2, dpp_shr0_f32<0x112>(v), v);
  v = fmaf(m.g4, dpp_shr0_f32<0x114>(v), v);
  v = fmaf(m.g8, dpp_shr0_f32<0x118>(v), v);
  return v;
}

// (Round 4 A/B, profiles/r04_raw/ab_emit.log: all sixteen streams as 64 hand-placed v_fmac_f32_dpp — 440 instead of 609
// VALU instructions per pair — was SLOWER, 194.5 -> 202.4 us for the main-field scatter: a DPP-modified VALU instruction
// does not issue at the plain rate, and the compiler's mov_dpp + fma form interleaves the streams better.  Removed.)

// (The phase timers of rounds 2 - 5 — -DFNR_EMIT_TIMING, fnr_debug_emit_phases, tools/microbench/scatter_phases.py — and the
// accumulate kernel's zero_early A/B variant left the tree in round 6; they are in the history at commit 48e9f7b.)

// PAIRS: the two x-neighbours of a cell edge — corners (0,3), (1,2), (4,7), (5,6) of the oracle's order — always fall
// into the same bin when every level's resolution is below the bin size (x only touches row bits below log2(res + 1),
// the bin is the row's high bits): they are counted and placed with ONE LDS atomic per pair, half the LDS atomics of
// the count and placement phases (a third of this kernel's time).
template <class Source, bool PAIRS>
__global__ __launch_bounds__(SC_EMIT_THREADS) void k_scatter_emit(GridDev grid, Warp warp, Source src, long long N,
                                                      const float2* __restrict__ d_feats, float2* __restrict__ queue_v,
                                                      unsigned short* __restrict__ queue_r, unsigned* __restrict__ qcount, unsigned* __restrict__ qmax,
                                                      long long cap, int log2_rows, int level0, int level_count, int lpb
                                                      ) {
  // LDS-staged multisplit: records are grouped by bin in LDS, then copied out as contiguous runs.
  // A workgroup takes its 512 samples through `lpb` consecutive levels: the sample position and its warp are computed
  // once, and the next level's feature gradient is loaded while the current level is processed (the wait for the first
  // loads + the warp was 22 % of a one-level workgroup's time).
  __shared__ float2 s_val[SC_CHUNK * 8];    // value pair of a record
  __shared__ unsigned s_key[SC_CHUNK * 8];  // row inside the bin | bin << 16
  __shared__ unsigned s_cnt[SC_MAX_BINS];   // per-bin count, then running cursor
  __shared__ unsigned s_off[SC_MAX_BINS];   // per-bin start inside s_val / s_key
  __shared__ unsigned s_base[SC_MAX_BINS];  // per-bin start inside the global queue
  __shared__ unsigned s_max;                // max |value| emitted by this workgroup (float bits; order-preserving for >= 0)
  __shared__ unsigned s_wsum[SC_EMIT_THREADS / 64];
  const int bins = 1 << (grid.log2_T - log2_rows);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t mask = (1u << grid.log2_T) - 1u;
  const uint32_t row_mask = (1u << log2_rows) - 1u;
  const long long n0 = (long long)blockIdx.x * SC_CHUNK;
  static_assert(SC_PER_THREAD == 1, "one sample per thread");
  const long long n = n0 + threadIdx.x;
  const bool valid = n < N;
  float x[3] = {0.f, 0.f, 0.f};
  float2 gf_next = make_float2(0.f, 0.f);
  const int lrel0 = blockIdx.y * lpb;
  if (valid) {
    gf_next = ntc_load<NT_DFEATS_LD>(&d_feats[(size_t)(level0 + lrel0) * N + n]);
    float px, py, pz;
    src.position(n, px, py, pz);
    warp_position(warp, px, py, pz, x);
  }
  for (int li = 0; li < lpb && lrel0 + li < level_count; ++li) {
  const int lrel = lrel0 + li;           // level inside this call's range: indexes the counters and queues
  const int level = level0 + lrel;       // level of the grid: indexes scalings, d_feats and the gradient table
  for (int i = threadIdx.x; i < bins; i += SC_EMIT_THREADS) {
    s_cnt[i] = 0;
  }
  if (threadIdx.x == 0) s_max = 0;
  __syncthreads();
  const int scaling = grid.scalings[level];
  const float2 gf = gf_next;
  if (valid && li + 1 < lpb && lrel + 1 < level_count) gf_next = ntc_load<NT_DFEATS_LD>(&d_feats[(size_t)(level + 1) * N + n]);

  // contributions of this thread's samples; equal rows in adjacent lanes (consecutive samples of a ray share
  // cells at coarse levels) are pre-summed so only the last lane of a run emits a record
  uint32_t hk[SC_PER_THREAD][8];
  float vxk[SC_PER_THREAD][8], vyk[SC_PER_THREAD][8];
  float tmax = 0.0f;  // largest |value| this thread emits
  unsigned emit_mask[SC_PER_THREAD];
#pragma unroll
  for (int q = 0; q < SC_PER_THREAD; ++q) {
    float wgt[8];
    const GridLevel g = corner_weights(x, scaling, mask, hk[q], wgt);
    // exact cell identity: floor coordinates + whether ceil differs (coordinates < 2^16, checked on the host)
    const uint32_t key_a = valid ? (g.f[0] | (g.f[1] << 16)) : 0xffffffffu;
    const uint32_t key_b = valid ? (g.f[2] | ((g.c[0] - g.f[0]) << 16) | ((g.c[1] - g.f[1]) << 17) |
                                    ((g.c[2] - g.f[2]) << 18))
                                 : (0x80000000u | (uint32_t)lane);
    const RunMasks rm = run_structure(key_a, key_b, lane);
    emit_mask[q] = 0;
    // fine levels: every sample of the wave sits in its own cell — no run to sum (wave-uniform branch around the
    // 16 x 4 DPP steps, a quarter of this kernel's VALU instructions)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      vxk[q][k] = valid ? wgt[k] * gf.x : 0.0f;
      vyk[q][k] = valid ? wgt[k] * gf.y : 0.0f;
    }
    if (rm.any_run) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        vxk[q][k] = run_sum(vxk[q][k], rm);
        vyk[q][k] = run_sum(vyk[q][k], rm);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (rm.tail && (vxk[q][k] != 0.0f || vyk[q][k] != 0.0f)) {
        emit_mask[q] |= 1u << k;
        tmax = fmaxf(tmax, fmaxf(fabsf(vxk[q][k]), fabsf(vyk[q][k])));
      }
    }
    if constexpr (PAIRS) {
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        const int ka = (pr == 0) ? 0 : (pr == 1) ? 1 : (pr == 2) ? 4 : 5, kb = (pr == 0) ? 3 : (pr == 1) ? 2 : (pr == 2) ? 7 : 6;
        const unsigned cnt = ((emit_mask[q] >> ka) & 1u) + ((emit_mask[q] >> kb) & 1u);
        if (cnt) atomicAdd(&s_cnt[hk[q][ka] >> log2_rows], cnt);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if ((emit_mask[q] >> k) & 1u) atomicAdd(&s_cnt[hk[q][k] >> log2_rows], 1u);
    }
  }
  // largest emitted |value| of the level (scale of the accumulate kernel's block fixed point): wave max by DPP-free
  // shuffles, one LDS atomic per wave, one global atomicMax per workgroup (per-bin maxima cost 8 LDS atomics per
  // thread and 64 global atomics per workgroup)
#pragma unroll
  for (int dsh = 32; dsh >= 1; dsh >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, dsh, 64));
  if (lane == 0 && tmax > 0.0f) atomicMax(&s_max, __float_as_uint(tmax));
  __syncthreads();
  if (threadIdx.x == 0 && s_max != 0u) atomicMax(&qmax[(size_t)lrel * SC_CNT_STRIDE], s_max);
  // exclusive scan of the per-bin counts (bins <= 1024 = 4 per thread) + one global reservation per non-empty bin
  unsigned c4[SC_BINS_PER_THREAD], tsum = 0;
#pragma unroll
  for (int t = 0; t < SC_BINS_PER_THREAD; ++t) {
    const int i = threadIdx.x * SC_BINS_PER_THREAD + t;
    c4[t] = (i < bins) ? s_cnt[i] : 0u;
    tsum += c4[t];
  }
  unsigned incl = tsum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned u = __shfl_up(incl, d, 64);
    if (lane >= d) incl += u;
  }
  if (lane == 63) s_wsum[wave] = incl;
  __syncthreads();
  unsigned woff = 0;
  for (int w = 0; w < wave; ++w) woff += s_wsum[w];
  unsigned total = 0;
#pragma unroll
  for (int w = 0; w < SC_EMIT_THREADS / 64; ++w) total += s_wsum[w];
  unsigned run = woff + incl - tsum;
#pragma unroll
  for (int t = 0; t < SC_BINS_PER_THREAD; ++t) {
    const int i = threadIdx.x * SC_BINS_PER_THREAD + t;
    if (i < bins) {
      s_off[i] = run;
      s_base[i] = c4[t] ? atomicAdd(&qcount[(size_t)(lrel * bins + i) * SC_CNT_STRIDE], c4[t]) : 0u;
      s_cnt[i] = 0;
      run += c4[t];
    }
  }
  __syncthreads();
  // place the records bin by bin in LDS
#pragma unroll
  for (int q = 0; q < SC_PER_THREAD; ++q) {
    if constexpr (PAIRS) {
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        const int ka = (pr == 0) ? 0 : (pr == 1) ? 1 : (pr == 2) ? 4 : 5, kb = (pr == 0) ? 3 : (pr == 1) ? 2 : (pr == 2) ? 7 : 6;
        const unsigned ea = (emit_mask[q] >> ka) & 1u, eb = (emit_mask[q] >> kb) & 1u;
        if (ea + eb) {
          const unsigned bin = hk[q][ka] >> log2_rows;
          const unsigned pos = s_off[bin] + atomicAdd(&s_cnt[bin], ea + eb);
          if (ea) {
            s_val[pos] = make_float2(vxk[q][ka], vyk[q][ka]);
            s_key[pos] = (hk[q][ka] & row_mask) | (bin << 16);
          }
          if (eb) {
            s_val[pos + ea] = make_float2(vxk[q][kb], vyk[q][kb]);
            s_key[pos + ea] = (hk[q][kb] & row_mask) | (bin << 16);
          }
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if ((emit_mask[q] >> k) & 1u) {
          const int bin = hk[q][k] >> log2_rows;
          const unsigned pos = s_off[bin] + atomicAdd(&s_cnt[bin], 1u);
          s_val[pos] = make_float2(vxk[q][k], vyk[q][k]);
          s_key[pos] = (hk[q][k] & row_mask) | ((unsigned)bin << 16);
        }
      }
    }
  }
  __syncthreads();
  // copy out: consecutive records of a bin go to consecutive queue slots (coalesced 8-byte + 2-byte stores)
  float* table = reinterpret_cast<float*>(grid.table + ((size_t)level << grid.log2_T));
  unsigned overflowed_here = 0;
  for (unsigned i = threadIdx.x; i < total; i += SC_EMIT_THREADS) {
    const float2 v = s_val[i];
    const unsigned key = s_key[i];
    const unsigned bin = key >> 16, row_in_bin = key & 0xffffu;
    const unsigned slot = s_base[bin] + (i - s_off[bin]);
    if ((long long)slot < cap) {
      const size_t q = ((size_t)lrel * bins + bin) * cap + slot;
      // (plain, one record per store: narrow `nt` stores are one fabric write each — emit 75 -> 120 us — and two records per
      //  store with even-sized reservations + zero pads measured 71 -> 79 us, profiles/r05_raw/kt_pair_call8.log)
      queue_v[q] = v;
      queue_r[q] = (unsigned short)row_in_bin;
    } else {  // hot bin: fall back to global atomics (rare; keeps the result independent of `cap`)
      qmax[(size_t)lrel * SC_CNT_STRIDE + 1] = 1u;  // tells the accumulate kernel that the table holds part of the sum
      ++overflowed_here;                             // fnr_debug_scatter_overflows: one atomic per thread, below
      const size_t row = ((size_t)bin << log2_rows) + row_in_bin;
      atomicAdd(table + 2 * row, v.x);
      atomicAdd(table + 2 * row + 1, v.y);
    }
  }
  if (overflowed_here) atomicAdd(&g_scatter_overflow_records, (unsigned long long)overflowed_here);
  __syncthreads();  // the next level re-uses the bin tables and the record staging
  }  // levels of this workgroup
}

// LDS fp32 atomics (ds_add_f32) retire ~1 lane every 3 clocks per CU on gfx950 (measured: 200 G/s chip-wide,
// 17x slower than ds_add_u32), so the per-bin sums are accumulated as 64-bit BLOCK FIXED POINT with
// ds_add_u64 (measured 10x faster for two adds per record).  The scale is chosen per bin from the largest
// |value| queued for it (tracked by the emit kernel) and the record count so that the sum cannot overflow:
// resolution = max|v| * 2^-41 or better, i.e. finer than fp32 rounding of any partial sum that contains the
// largest term; exact and order-independent (bitwise deterministic) above that resolution.
__device__ __forceinline__ void acc_record(unsigned long long* __restrict__ s_acc, unsigned row, const float2& v,
                                           double scale) {
  const long long ix = __double2ll_rn((double)v.x * scale), iy = __double2ll_rn((double)v.y * scale);
  atomicAdd(&s_acc[2 * row], (unsigned long long)ix);
  atomicAdd(&s_acc[2 * row + 1], (unsigned long long)iy);
}

// (TableAdam / table_adam_update: common.hpp — the optimiser step fused into the accumulate kernel)
// everything one accumulate launch needs about one scatter call (a launch can serve two calls: k_scatter_accumulate2)
struct AccArgs {
  GridDev grid;
  const float2* queue_v;
  const unsigned short* queue_r;
  unsigned *qcount, *qmax, *qdone;
  long long cap;
  int log2_rows, level0, nbins;   // nbins = level_count * bins per level = workgroups of this call
  int kind;                       // 0: the field's table, 1: a proposal network's (g_scatter_records)
  TableAdam adam;
};

// s_acc: [rows][2] two's-complement fixed point in DYNAMIC LDS, 16 bytes per row of the bin (128 KiB for the main
// table's 8192-row bins, 64 KiB for the proposal tables' 4096: two workgroups per CU there)
template <bool ADAM>
__device__ __forceinline__ void accumulate_bin(const AccArgs& A, int vblock, unsigned long long* __restrict__ s_acc) {
  const GridDev& grid = A.grid;
  const float2* __restrict__ queue_v = A.queue_v;
  const unsigned short* __restrict__ queue_r = A.queue_r;
  unsigned* __restrict__ qcount = A.qcount;
  unsigned* __restrict__ qmax = A.qmax;
  unsigned* __restrict__ qdone = A.qdone;
  const long long cap = A.cap;
  const int log2_rows = A.log2_rows, level0 = A.level0;
  const TableAdam& adam = A.adam;
  const int rows = 1 << log2_rows;
  const int bins = 1 << (grid.log2_T - log2_rows);
  // fine levels (long queues) are dispatched first, the short coarse-level bins fill the tail
  const int gbin = A.nbins - 1 - vblock;  // (level - level0) * bins + bin
  const int lrel = gbin / bins, bin = gbin - lrel * bins;
  const int level = level0 + lrel;
  long long n = qcount[(size_t)gbin * SC_CNT_STRIDE];
  const float vmax = __uint_as_float(qmax[(size_t)lrel * SC_CNT_STRIDE]);  // largest |value| of the whole level
  // some emit workgroup overflowed a queue of this level and added records to the gradient table with atomics
  const bool overflowed = qmax[(size_t)lrel * SC_CNT_STRIDE + 1] != 0u;
  // EVERY WAVE MUST HOLD ITS COPIES BEFORE ANYONE MAY RESET A COUNTER.  The loads above are uniform, and the compiler is free
  // to issue them as scalar loads or as vector loads; a workgroup barrier on gfx950 only waits for LDS / scalar traffic
  // (lgkmcnt), NOT for outstanding vector loads (the workgroup-scope fence of __syncthreads omits vmcnt(0)).  In
  // k_scatter_accumulate2<true> the FIRST call's copy of this function (proposal network 0) got VECTOR loads: its waves
  // passed the barrier with the count / maximum still in flight, thread 0 then reset the count and counted the workgroup in
  // on `qdone` — and the last workgroup of the level to arrive (any CU, any XCD) reset the level's maximum while loads of it
  // were still on their way somewhere else.  A wave whose load was overtaken read 0, took its share of the bin's queue
  // (a sixteenth) for empty and dropped it: the rare two-stream divergence of long fruit_nerf_big runs (DESIGN 2 round 4 (c);
  // found by reading the ISA at the end of round 4: `s_barrier` ahead of `s_waitcnt vmcnt(1)` in that one code path, scalar
  // loads + `s_waitcnt lgkmcnt(0)` ahead of the barrier in the second call's copy and in k_scatter_accumulate).  The
  // counters are inputs of the wait: the compiler has to have them in registers before it, whatever loads it chose.
  // CONFIRMED ON HARDWARE IN ROUND 5 (tools/microbench/barrier_load_race.hip, profiles/r05_raw/barrier_load_race.log): the
  // same load / barrier / reset pattern with vector loads and no wait returned 1264 stale level maxima in 200 000 rounds of
  // 160 workgroups once a second stream kept the memory system busy; with this wait, and with scalar loads, none.
  {
    const unsigned held_n = (unsigned)n, held_max = __float_as_uint(vmax), held_ovf = overflowed ? 1u : 0u;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::"v"(held_n), "v"(held_max), "v"(held_ovf) : "memory");
  }
  // now every thread has its copy: put the counters back to zero, so the NEXT call on this workspace needs no memset
  // launch (the caller says so with workspace_clean = 1).  The level's max is shared by its `bins` workgroups: the
  // last of them to have read it (its own increment of `qdone` follows its waves' loads) clears it.
  __syncthreads();
  if (threadIdx.x == 0) {
    qcount[(size_t)gbin * SC_CNT_STRIDE] = 0u;
    if (atomicAdd(&qdone[(size_t)lrel * SC_CNT_STRIDE], 1u) == (unsigned)bins - 1u) {
      qmax[(size_t)lrel * SC_CNT_STRIDE] = 0u;
      qmax[(size_t)lrel * SC_CNT_STRIDE + 1] = 0u;
      qdone[(size_t)lrel * SC_CNT_STRIDE] = 0u;
    }
  }
  const bool have = n != 0 && vmax > 0.0f;
  if (threadIdx.x == 0 && n > 0)
    atomicAdd(&g_scatter_records[A.kind & 1][vblock & 63][0], (unsigned long long)(n > cap ? cap : n));
  if (!ADAM && !have) return;  // (with the fused optimiser every row still takes its moment-decay step)
  if (n > cap) n = cap;
  if (n < 1) n = 1;
  // |v| < 2^e ; n < 2^nb  =>  |sum * 2^S| < 2^62 with S = 62 - nb - e
  int e;
  (void)frexpf(vmax, &e);
  const int nb = 64 - __clzll((unsigned long long)n);
  int S = 62 - nb - e;
  if (S > 1000) S = 1000;
  const double scale = ldexp(1.0, S), inv = ldexp(1.0, -S);
  for (int i = threadIdx.x; i < 2 * rows; i += blockDim.x) s_acc[i] = 0ull;
  __syncthreads();
  if (!have) n = 0;
  const float2* qv = queue_v + (size_t)gbin * cap;
  const unsigned short* qr = queue_r + (size_t)gbin * cap;
  // records in PAIRS: one 16-byte + one 4-byte load per two records (a bin's queue starts at a multiple of `cap`, a
  // multiple of 1024 records).  8-byte + 2-byte loads ran at ~0.6x the bytes per clock of a CU (MI355X_MICROARCH: narrow
  // accesses), and this kernel is bound by exactly that: one workgroup per CU streaming its queue.
  const float4* qv2 = reinterpret_cast<const float4*>(qv);
  const ushort2* qr2 = reinterpret_cast<const ushort2*>(qr);
  const long long np = n >> 1;  // full pairs
  long long i = threadIdx.x;
  for (; i + 3 * (long long)blockDim.x < np; i += 4 * (long long)blockDim.x) {  // 8 loads (8 records) in flight per thread
    float4 v[4];
    ushort2 row[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u] = ntc_load<NT_QUEUE_LD>(&qv2[i + u * (long long)blockDim.x]);
      row[u] = ntc_load<NT_QUEUE_LD>(&qr2[i + u * (long long)blockDim.x]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc_record(s_acc, row[u].x, make_float2(v[u].x, v[u].y), scale);
      acc_record(s_acc, row[u].y, make_float2(v[u].z, v[u].w), scale);
    }
  }
  for (; i < np; i += blockDim.x) {
    const float4 v = ntc_load<NT_QUEUE_LD>(&qv2[i]);
    const ushort2 row = ntc_load<NT_QUEUE_LD>(&qr2[i]);
    acc_record(s_acc, row.x, make_float2(v.x, v.y), scale);
    acc_record(s_acc, row.y, make_float2(v.z, v.w), scale);
  }
  if ((n & 1) && threadIdx.x == 0) {
    acc_record(s_acc, qr[n - 1], qv[n - 1], scale);
  }
  __syncthreads();
  float2* dst = grid.table + ((size_t)level << grid.log2_T) + (size_t)bin * rows;
  if constexpr (ADAM) {
    const size_t row0 = ((size_t)level << grid.log2_T) + (size_t)bin * rows;
    // Two rows per thread through 16-byte loads / stores: this phase moves two thirds of the kernel's bytes (24 B in +
    // 24 B out per row) and a CU streams 8-byte accesses at ~0.6 x the bytes per clock of 16-byte ones
    // (MI355X_MICROARCH: narrow accesses).  Same arithmetic per entry, so the results do not change.
    const bool wide = rows >= 2 && (((uintptr_t)(adam.p + row0) | (uintptr_t)(adam.m + row0) | (uintptr_t)(adam.v + row0)) & 15u) == 0 &&
                      !overflowed;
    if (wide) {
      float4* P4 = reinterpret_cast<float4*>(adam.p + row0);
      float4* M4 = reinterpret_cast<float4*>(adam.m + row0);
      float4* V4 = reinterpret_cast<float4*>(adam.v + row0);
      // sparse-touch skipping: a pair of rows that never received a gradient (bit clear) and receives none now has
      // m = v = 0 and a zero update — its 48 bytes in + 48 out are skipped.  Bins are >= 64 rows here (32 pairs a word).
      unsigned* tw = (adam.touched && rows >= 64) ? adam.touched + (row0 >> 6) : nullptr;
      // (Two iterations' parameters / moments in flight per thread — 96 B instead of 48 — changes nothing: 194.3 vs 194.9 us
      //  for the whole entry point, same-box A/B, profiles/r04_raw/ab_sweep.log.  The sweep is not latency-bound.)
      for (int e4 = threadIdx.x; e4 < (rows >> 1); e4 += blockDim.x) {
        const long long a0 = (long long)s_acc[4 * e4], a1 = (long long)s_acc[4 * e4 + 1],
                        a2 = (long long)s_acc[4 * e4 + 2], a3 = (long long)s_acc[4 * e4 + 3];
        if (tw) {
          const bool now = (a0 | a1 | a2 | a3) != 0;
          const bool ever = (tw[e4 >> 5] >> (e4 & 31)) & 1u;
          if (!now && !ever) continue;
          if (!ever) atomicOr(&tw[e4 >> 5], 1u << (e4 & 31));   // first gradient of this pair (once per pair, ever)
        }
        float4 P = P4[e4], M = ntc_load<NT_MOMENT_LD>(&M4[e4]), V = ntc_load<NT_MOMENT_LD>(&V4[e4]);
        const float g0 = (a0 != 0 || a1 != 0) ? 0.0f + (float)((double)a0 * inv) : 0.0f;
        const float g1 = (a0 != 0 || a1 != 0) ? 0.0f + (float)((double)a1 * inv) : 0.0f;
        const float g2 = (a2 != 0 || a3 != 0) ? 0.0f + (float)((double)a2 * inv) : 0.0f;
        const float g3 = (a2 != 0 || a3 != 0) ? 0.0f + (float)((double)a3 * inv) : 0.0f;
        table_adam_update(adam, g0, P.x, M.x, V.x);
        table_adam_update(adam, g1, P.y, M.y, V.y);
        table_adam_update(adam, g2, P.z, M.z, V.z);
        table_adam_update(adam, g3, P.w, M.w, V.w);
        P4[e4] = P;
        ntc_store<NT_MOMENT_ST>(&M4[e4], M);
        ntc_store<NT_MOMENT_ST>(&V4[e4], V);
      }
      return;
    }
    // (row-by-row sweep: unaligned spans, and levels some queue of which overflowed.  It visits every row, but it has to
    //  keep the sparse-touch bitmap valid for the wide sweeps of later steps: a row that gets its first gradient here
    //  leaves with non-zero moments, so its pair's bit is set — without it the next wide sweep would skip the pair and
    //  freeze its moments)
    unsigned* tw1 = (adam.touched && rows >= 64) ? adam.touched + (row0 >> 6) : nullptr;
    for (int e2 = threadIdx.x; e2 < rows; e2 += blockDim.x) {
      const long long ax = (long long)s_acc[2 * e2], ay = (long long)s_acc[2 * e2 + 1];
      // the gradient exactly as the unfused path leaves it in the table: existing entry (zero, or what overflowed
      // queues added with atomics) + this bin's sum
      float2 g = make_float2(0.0f, 0.0f);
      if (overflowed) {
        g = dst[e2];
        if (g.x != 0.0f || g.y != 0.0f) dst[e2] = make_float2(0.0f, 0.0f);
      }
      if (ax != 0 || ay != 0) {
        g.x += (float)((double)ax * inv);
        g.y += (float)((double)ay * inv);
      }
      if (tw1 && (ax != 0 || ay != 0 || g.x != 0.0f || g.y != 0.0f)) {
        const int pair = e2 >> 1;
        if (!((tw1[pair >> 5] >> (pair & 31)) & 1u)) atomicOr(&tw1[pair >> 5], 1u << (pair & 31));
      }
      float2 P = adam.p[row0 + e2], M = adam.m[row0 + e2], V = adam.v[row0 + e2];
      table_adam_update(adam, g.x, P.x, M.x, V.x);
      table_adam_update(adam, g.y, P.y, M.y, V.y);
      adam.p[row0 + e2] = P;
      adam.m[row0 + e2] = M;
      adam.v[row0 + e2] = V;
    }
  } else {
    for (int e2 = threadIdx.x; e2 < rows; e2 += blockDim.x) {
      const long long ax = (long long)s_acc[2 * e2], ay = (long long)s_acc[2 * e2 + 1];
      if (ax != 0 || ay != 0) {
        float2 t = dst[e2];
        t.x += (float)((double)ax * inv);
        t.y += (float)((double)ay * inv);
        dst[e2] = t;
      }
    }
  }
}

extern __shared__ __attribute__((aligned(16))) unsigned char acc_smem[];
template <bool ADAM>
__global__ __launch_bounds__(1024) void k_scatter_accumulate(AccArgs a) {
  accumulate_bin<ADAM>(a, (int)blockIdx.x, reinterpret_cast<unsigned long long*>(acc_smem));
}
// two scatter calls in one launch (the two proposal levels' tables: 160 workgroups each on 256 CUs — side by side they
// cost the longer of the two instead of their sum); the longer queues (call a) are dispatched first
template <bool ADAM>
__global__ __launch_bounds__(1024) void k_scatter_accumulate2(AccArgs a, AccArgs b) {
  unsigned long long* s_acc = reinterpret_cast<unsigned long long*>(acc_smem);
  if ((int)blockIdx.x < a.nbins) accumulate_bin<ADAM>(a, (int)blockIdx.x, s_acc);
  else accumulate_bin<ADAM>(b, (int)blockIdx.x - a.nbins, s_acc);
}


// emit of one scatter call -> the arguments its accumulate launch needs
template <class Source>
static int scatter_emit(const fnr_grid* grid_grad, const Warp& warp, const Source& src, long long N,
                        const float2* d_feats, int level0, int level_count, void* workspace, size_t workspace_bytes,
                        int workspace_clean, hipStream_t st, const TableAdam* adam, AccArgs& acc) {
  FNR_CHECK_ARG(level0 >= 0 && level_count >= 1 && level0 + level_count <= grid_grad->n_levels,
                "hash scatter: level range [%d,+%d) outside the %d levels", level0, level_count, grid_grad->n_levels);
  const ScatterPlan p = scatter_plan(N, level_count, grid_grad->log2_hashmap_size);
  FNR_CHECK_ARG(p.bins_per_level <= SC_MAX_BINS, "hash scatter: log2_hashmap_size %d too large for the bin histogram",
                grid_grad->log2_hashmap_size);
  FNR_CHECK_ARG(workspace && workspace_bytes >= p.count_bytes + p.queue_bytes, "hash scatter: workspace too small");
  unsigned* qcount = reinterpret_cast<unsigned*>(workspace);
  float2* queue_v = reinterpret_cast<float2*>(reinterpret_cast<char*>(workspace) + p.count_bytes);
  unsigned short* queue_r = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(workspace) + p.count_bytes + p.value_bytes);
  if (!workspace_clean) FNR_HIP(hipMemsetAsync(qcount, 0, p.count_bytes, st));
  const size_t nbins_all = (size_t)level_count * p.bins_per_level;
  const long long chunks = (N + SC_CHUNK - 1) / SC_CHUNK;
  FNR_CHECK_ARG(chunks < (1ll << 31), "hash scatter: too many samples");
  const GridDev gd = make_grid(grid_grad);
  for (int l = 0; l < grid_grad->n_levels; ++l)
    FNR_CHECK_ARG(gd.scalings[l] > 0 && gd.scalings[l] < 65535, "hash scatter: level resolution %d out of range", gd.scalings[l]);
  bool pairs = true;  // every level's resolution below the bin size: x-neighbours share their bin (see k_scatter_emit)
  for (int l = level0; l < level0 + level_count; ++l) pairs = pairs && gd.scalings[l] < (1 << p.log2_rows);
  // levels per workgroup: as many as leaves >= 4 workgroups per CU-slot (3 workgroups per CU) in flight
  int lpb = 1;
  {
    const char* e = getenv("FNR_EMIT_LPB");  // tests force the multi-level path at small sizes
    const int forced = e ? atoi(e) : 0;
    const long long slots = 3ll * device_cu_count();
    while (lpb < level_count && lpb < 8 && chunks * ((level_count + 2 * lpb - 1) / (2 * lpb)) >= 2 * slots) lpb *= 2;
    if (forced > 0) lpb = forced;
    if (lpb > level_count) lpb = level_count;
  }
  const unsigned gy = (unsigned)((level_count + lpb - 1) / lpb);
  if (chunks == 0) {
    // nothing to emit (the fused-optimiser call still runs the accumulate kernel: every row takes its step)
  } else if (pairs)
    hipLaunchKernelGGL((k_scatter_emit<Source, true>), dim3((unsigned)chunks, gy), dim3(SC_EMIT_THREADS),
                       0, st, gd, warp, src, N, d_feats, queue_v, queue_r, qcount, qcount + nbins_all * SC_CNT_STRIDE, p.cap,
                       p.log2_rows, level0, level_count, lpb);
  else
    hipLaunchKernelGGL((k_scatter_emit<Source, false>), dim3((unsigned)chunks, gy), dim3(SC_EMIT_THREADS),
                       0, st, gd, warp, src, N, d_feats, queue_v, queue_r, qcount, qcount + nbins_all * SC_CNT_STRIDE, p.cap,
                       p.log2_rows, level0, level_count, lpb);
  FNR_LAUNCH_CHECK();
  acc.grid = gd;
  acc.queue_v = queue_v, acc.queue_r = queue_r, acc.qcount = qcount;
  acc.qmax = qcount + nbins_all * SC_CNT_STRIDE;
  acc.qdone = qcount + (nbins_all + level_count) * SC_CNT_STRIDE;
  acc.cap = p.cap, acc.log2_rows = p.log2_rows, acc.level0 = level0;
  acc.nbins = level_count * p.bins_per_level;
  acc.kind = 0;
  acc.adam = adam ? *adam : TableAdam{};
  return FNR_OK;
}

static int acc_lds_bytes(const AccArgs& a) { return (2 << a.log2_rows) * (int)sizeof(unsigned long long); }

static int scatter_accumulate(const AccArgs& a, bool adam, hipStream_t st) {
  const int bytes = acc_lds_bytes(a);
  if (adam) {
    const int rc = ensure_dyn_lds(k_scatter_accumulate<true>, 2 * SC_MAX_ROWS * (int)sizeof(unsigned long long));
    if (rc) return rc;
    hipLaunchKernelGGL(k_scatter_accumulate<true>, dim3((unsigned)a.nbins), dim3(1024), bytes, st, a);
  } else {
    const int rc = ensure_dyn_lds(k_scatter_accumulate<false>, 2 * SC_MAX_ROWS * (int)sizeof(unsigned long long));
    if (rc) return rc;
    hipLaunchKernelGGL(k_scatter_accumulate<false>, dim3((unsigned)a.nbins), dim3(1024), bytes, st, a);
  }
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

// the accumulate launches of two scatter calls as one (a's queues are the longer ones)
static int scatter_accumulate2(const AccArgs& a, const AccArgs& b, bool adam, hipStream_t st) {
  const int bytes = acc_lds_bytes(a) > acc_lds_bytes(b) ? acc_lds_bytes(a) : acc_lds_bytes(b);
  if (adam) {
    const int rc = ensure_dyn_lds(k_scatter_accumulate2<true>, 2 * SC_MAX_ROWS * (int)sizeof(unsigned long long));
    if (rc) return rc;
    hipLaunchKernelGGL(k_scatter_accumulate2<true>, dim3((unsigned)(a.nbins + b.nbins)), dim3(1024), bytes, st, a, b);
  } else {
    const int rc = ensure_dyn_lds(k_scatter_accumulate2<false>, 2 * SC_MAX_ROWS * (int)sizeof(unsigned long long));
    if (rc) return rc;
    hipLaunchKernelGGL(k_scatter_accumulate2<false>, dim3((unsigned)(a.nbins + b.nbins)), dim3(1024), bytes, st, a, b);
  }
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

template <class Source>
static int binned_scatter(const fnr_grid* grid_grad, const Warp& warp, const Source& src, long long N,
                          const float2* d_feats, int level0, int level_count, void* workspace, size_t workspace_bytes,
                          int workspace_clean, hipStream_t st, const TableAdam* adam = nullptr) {
  AccArgs acc;
  const int rc = scatter_emit(grid_grad, warp, src, N, d_feats, level0, level_count, workspace, workspace_bytes,
                              workspace_clean, st, adam, acc);
  if (rc) return rc;
  return scatter_accumulate(acc, adam != nullptr, st);
}

// ------------------------------------------------------------------------------------------------
// proposal network backward.  Persistent workgroups; per iteration 256 samples:
//   phase 1 (thread = sample): recompute the MLP from the saved features, d_out = d_sigma * trunc_exp'(out),
//            hidden gradients -> LDS, feature gradients -> d_feats [L][N][2] (scattered by binned_scatter);
//   phase 2 (wave = 64 of the samples): dW0 = dh^T f, db0 = dh^T 1 and dW1 = (relu(h) d_out)^T 1 as 16x16x4 fp32 MFMAs
//            with the samples on the K axis (3 MFMAs per 4 samples; operands are single conflict-light LDS
//            reads).  A thread-per-weight loop over the 256 samples read two LDS words per FMA and was LDS-bound
//            (~45 us of the 73 us this kernel took for 1 M samples).
// Weight gradients leave the workgroup once, at the end (one atomicAdd per weight per workgroup).
// ------------------------------------------------------------------------------------------------
constexpr int PROP_PART = 320;   // floats per workgroup partial: dW0 tile 256 + dW1 16 + db0 16 + db1 (+ pad)

template <int L, int H, bool POSGRAD>
__global__ __launch_bounds__(256) void k_prop_bwd(GridDev grid, float4* __restrict__ d_xw, Warp warp, RaySource src,
                                                  long long N, const float* __restrict__ w0,
                                                  const float* __restrict__ b0, const float* __restrict__ w1,
                                                  const float* __restrict__ b1, const float2* __restrict__ feat_save,
                                                  const float* __restrict__ d_density, float2* __restrict__ d_feats,
                                                  float* __restrict__ partials) {
  constexpr int K = 2 * L;
  __shared__ float s_dh[256][H + 1];   // d hidden (pre-activation)
  __shared__ float s_ha[256][H + 1];   // relu(hidden) * d_out  (for dW1)
  __shared__ float s_f[256][K + 1];    // input features
  __shared__ float s_do[256];          // d_out
  static_assert(H == 16 && K <= 16, "phase 2 is a single 16x16 MFMA tile");
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  // accumulators (C layout: lane holds column j, rows 4 g + r): dW0[o][k = j], db0[o] in column 0 of d3,
  // dW1[o] in column 0 of d2; db1 per thread
  f32x4 d1 = {0.f, 0.f, 0.f, 0.f}, d2 = {0.f, 0.f, 0.f, 0.f}, d3 = {0.f, 0.f, 0.f, 0.f};
  float db1 = 0.0f;
  const float ones_col0 = (j == 0) ? 1.0f : 0.0f;  // B operand that sums over the samples into column 0
  const long long n_iter = (N + 255) / 256;
  for (long long it = blockIdx.x; it < n_iter; it += gridDim.x) {
    const long long n = it * 256 + tid;
    float f[K], dout = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) f[k] = 0.0f;
    bool sel = false;
    float x[3] = {0.f, 0.f, 0.f};
    if (n < N) {
      float px, py, pz;
      src.position(n, px, py, pz);
      sel = warp_position(warp, px, py, pz, x);
#pragma unroll
      for (int l = 0; l < L; ++l) {
        const float2 v = ntc_load<NT_PROP_FEATS>(&feat_save[(size_t)l * N + n]);
        f[2 * l] = v.x;
        f[2 * l + 1] = v.y;
      }
    }
    float a[H];
    float out = b1[0];
#pragma unroll
    for (int o = 0; o < H; ++o) {
      float t = b0[o];
#pragma unroll
      for (int k = 0; k < K; ++k) t = fmaf(w0[o * K + k], f[k], t);
      a[o] = t;
      out = fmaf(w1[o], fmaxf(t, 0.0f), out);
    }
    if (n < N && sel) dout = d_density[n] * expf(fminf(fmaxf(out, -15.0f), 15.0f));  // trunc_exp backward
    float df[K];
#pragma unroll
    for (int k = 0; k < K; ++k) df[k] = 0.0f;
#pragma unroll
    for (int o = 0; o < H; ++o) {
      const float dh = (a[o] > 0.0f) ? dout * w1[o] : 0.0f;
      s_dh[tid][o] = dh;
      s_ha[tid][o] = fmaxf(a[o], 0.0f) * dout;
#pragma unroll
      for (int k = 0; k < K; ++k) df[k] = fmaf(dh, w0[o * K + k], df[k]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) s_f[tid][k] = f[k];
    db1 += dout;
    if (n < N) {
#pragma unroll
      for (int l = 0; l < L; ++l) ntc_store<NT_PROP_DFEATS_ST>(&d_feats[(size_t)l * N + n], make_float2(df[2 * l], df[2 * l + 1]));
    }
    if constexpr (POSGRAD) {
      // gradient w.r.t. the unit-cube position (camera-pose optimisation): re-gather the corner rows of every level
      // and contract with d(blend weights)/d(offset) (position_grad.hip has the main-field version)
      if (n < N) {
        const uint32_t hmask = (1u << grid.log2_T) - 1u;
        float gx = 0.0f, gy = 0.0f, gz = 0.0f;
#pragma unroll
        for (int l = 0; l < L; ++l) {
          const int scaling = grid.scalings[l];
          const GridLevel gl = grid_cell(x, scaling);
          uint32_t hh[8];
          grid_corners(gl, hmask, hh);
          const float2* lt = grid.table + ((size_t)l << grid.log2_T);
          float dk[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float2 v = lt[hh[k]];
            dk[k] = fmaf(df[2 * l], v.x, df[2 * l + 1] * v.y);
          }
          const float ox = gl.o[0], oy = gl.o[1], oz = gl.o[2];
          const float mx = 1.0f - ox, my = 1.0f - oy, mz = 1.0f - oz;
          const float s = sel ? (float)scaling : 0.0f;
          gx += s * (oz * (oy * (dk[0] - dk[3]) + my * (dk[1] - dk[2])) + mz * (oy * (dk[4] - dk[7]) + my * (dk[5] - dk[6])));
          gy += s * (oz * (ox * (dk[0] - dk[1]) + mx * (dk[3] - dk[2])) + mz * (ox * (dk[4] - dk[5]) + mx * (dk[7] - dk[6])));
          gz += s * (oy * (ox * (dk[0] - dk[4]) + mx * (dk[3] - dk[7])) + my * (ox * (dk[1] - dk[5]) + mx * (dk[2] - dk[6])));
        }
        d_xw[n] = make_float4(gx, gy, gz, 0.0f);
      }
    }
    __syncthreads();
    // phase 2: this wave's 64 samples, 4 per MFMA step.  A[i = o][kk] = dh / ha of sample 4 step + kk,
    // B[kk][j = k] = feature k of that sample (0 beyond K)
    const int row0 = 64 * wave + g;
#pragma unroll 4
    for (int st = 0; st < 16; ++st) {
      const int row = row0 + 4 * st;
      const float a1 = s_dh[row][j], a2 = s_ha[row][j];
      const float bf = (j < K) ? s_f[row][j < K ? j : 0] : 0.0f;
      d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bf, d1, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, ones_col0, d2, 0, 0, 0);
      d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, ones_col0, d3, 0, 0, 0);
    }
    __syncthreads();
  }
  // combine the 4 waves through LDS (reusing s_dh), then one atomicAdd per weight per workgroup
  float* red = &s_dh[0][0];  // [4 waves][3][16 rows][16 cols]
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    red[((wave * 3 + 0) * 16 + 4 * g + r) * 16 + j] = d1[r];
    red[((wave * 3 + 1) * 16 + 4 * g + r) * 16 + j] = d2[r];
    red[((wave * 3 + 2) * 16 + 4 * g + r) * 16 + j] = d3[r];
  }
  db1 = wave_sum(db1);
  if (lane == 0) s_do[wave] = db1;
  __syncthreads();
  {
    // one partial vector per workgroup: [dW0 16x16 | dW1 16 | db0 16 | db1], summed by k_prop_reduce.  Atomics
    // from 768 workgroups onto the same 13 cache lines of the gradient serialised in L2 (~34 us per call).
    const int o = tid >> 4, k = tid & 15;  // 256 threads = the 16 x 16 tile
    auto total = [&](int which) {
      return (red[((0 * 3 + which) * 16 + o) * 16 + k] + red[((1 * 3 + which) * 16 + o) * 16 + k]) +
             (red[((2 * 3 + which) * 16 + o) * 16 + k] + red[((3 * 3 + which) * 16 + o) * 16 + k]);
    };
    float* part = partials + (size_t)blockIdx.x * PROP_PART;
    part[tid] = total(0);
    if (k == 0) {
      part[256 + o] = total(1);
      part[272 + o] = total(2);
    }
    if (tid == 0) part[288] = (s_do[0] + s_do[1]) + (s_do[2] + s_do[3]);
  }
}

// gradient += sum over workgroups of their partial vectors, in a FIXED order with one writer per entry (no float
// atomics: bit-reproducible training).  Workgroup = PRD_E entries x PRD_Y slices: slice y sums rows y, y + PRD_Y, ... (a
// few hundred rows: two or three batches of 8 independent loads per thread), the slices meet in LDS and the thread of
// slice 0 adds them up in slice order.
// ADAM (fnr_prop_density_bwd_adam): the owning thread also takes the parameter's optimiser step (weight_adam_entry).
constexpr int PRD_E = 16, PRD_Y = 64;
struct PropReduceArgs {
  const float* partials;
  int nblocks, K;
  float *g_w0, *g_b0, *g_w1, *g_b1;
  WeightAdam wa;
};
template <bool ADAM>
__device__ __forceinline__ void prop_reduce_block(int block, const PropReduceArgs& a) {
  __shared__ float s_part[PRD_Y][PRD_E];
  const float* __restrict__ partials = a.partials;
  const int nblocks = a.nblocks, K = a.K;
  const int t = threadIdx.x % PRD_E, y = threadIdx.x / PRD_E;
  const int e = block * PRD_E + t;
  float s = 0.0f;
  if (e <= 288) {
    int b = y;
    for (; b + 7 * PRD_Y < nblocks; b += 8 * PRD_Y) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partials[(size_t)(b + u * PRD_Y) * PROP_PART + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < nblocks; b += PRD_Y) s += partials[(size_t)b * PROP_PART + e];
  }
  s_part[y][t] = s;
  __syncthreads();
  if (y != 0 || e > 288) return;
#pragma unroll 8
  for (int q = 1; q < PRD_Y; ++q) s += s_part[q][t];
  if (!ADAM && s == 0.0f) return;
  float* dst = nullptr;
  if (e < 256) {
    const int o = e >> 4, k = e & 15;
    if (k < K) dst = &a.g_w0[o * K + k];
  } else if (e < 272) {
    dst = &a.g_w1[e - 256];
  } else if (e < 288) {
    dst = &a.g_b0[e - 272];
  } else {
    dst = &a.g_b1[0];
  }
  if (!dst) return;
  if constexpr (ADAM) weight_adam_entry(a.wa, dst, s);
  else *dst += s;
}
template <bool ADAM>
__global__ __launch_bounds__(PRD_E * PRD_Y) void k_prop_reduce(PropReduceArgs a) {
  prop_reduce_block<ADAM>((int)blockIdx.x, a);
}
// both proposal levels' weight reductions as one launch (fnr_prop_density_bwd_pair_split: level 1's MLP backward no longer
// waits for level 0's reduction to start and drain; same sums in the same order by the same single writers)
template <bool ADAM>
__global__ __launch_bounds__(PRD_E * PRD_Y) void k_prop_reduce2(PropReduceArgs a, PropReduceArgs b) {
  constexpr int NB = PROP_PART / PRD_E;
  if ((int)blockIdx.x < NB) prop_reduce_block<ADAM>((int)blockIdx.x, a);
  else prop_reduce_block<ADAM>((int)blockIdx.x - NB, b);
}

}  // namespace fnr

using namespace fnr;


extern "C" int fnr_debug_scatter_overflows(uint64_t* count_host, int reset) {
  FNR_CHECK_ARG(count_host, "debug_scatter_overflows: null argument");
  unsigned long long n = 0;
  FNR_HIP(hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_scatter_overflow_records), sizeof(n)));
  *count_host = n;
  if (reset) {
    n = 0;
    FNR_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_scatter_overflow_records), &n, sizeof(n)));
  }
  return FNR_OK;
}

extern "C" int fnr_debug_scatter_records(uint64_t* records_host, int reset) {
  FNR_CHECK_ARG(records_host, "debug_scatter_records: null argument");
  // the accumulate kernels run on the caller's (non-blocking) streams, which a symbol copy does not order against:
  // drain the device first.  Stack buffer (16 KB): callable from several host threads.
  FNR_HIP(hipDeviceSynchronize());
  unsigned long long host[2][64][16];
  FNR_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_scatter_records), sizeof(host)));
  for (int k = 0; k < 2; ++k) {
    records_host[k] = 0;
    for (int i = 0; i < 64; ++i) records_host[k] += host[k][i][0];
  }
  if (reset) {
    memset(host, 0, sizeof(host));
    FNR_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_scatter_records), host, sizeof(host)));
  }
  return FNR_OK;
}


extern "C" size_t fnr_hash_scatter_workspace_bytes(int64_t n_samples, int n_levels, int log2_hashmap_size) {
  const ScatterPlan p = scatter_plan(n_samples, n_levels, log2_hashmap_size);
  return p.count_bytes + p.queue_bytes;
}

extern "C" int fnr_hash_encode_bwd(const fnr_grid* grid_grad, const fnr_warp* warp, const fnr_rays* rays,
                                   const float* euclid_bins, int S, const float* d_feats, int level_begin,
                                   int level_count, void* workspace, size_t workspace_bytes, int workspace_clean,
                                   void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_hash_encode_bwd");
  FNR_CHECK_ARG(grid_grad && warp && rays && euclid_bins && d_feats && S > 0, "hash_encode_bwd: null argument");
  FNR_CHECK_ARG(grid_grad->n_levels >= 1 && grid_grad->n_levels <= FNR_MAX_LEVELS, "hash_encode_bwd: n_levels");
  const long long N = rays->n_rays * (long long)S;
  if (N == 0) return FNR_OK;
  RaySource src{make_rays(rays), euclid_bins, S};
  FNR_PROF(OP_ENCODE_BWD, N);
  return binned_scatter(grid_grad, make_warp(warp), src, N, reinterpret_cast<const float2*>(d_feats), level_begin,
                        level_count, workspace, workspace_bytes, workspace_clean, as_stream(stream));
}

extern "C" int fnr_hash_encode_bwd_adam(const fnr_grid* grid_grad, const fnr_warp* warp, const fnr_rays* rays,
                                        const float* euclid_bins, int S, const float* d_feats, void* workspace,
                                        size_t workspace_bytes, int workspace_clean, const fnr_table_adam* adam,
                                        void* stream) {
  if (seq::recording() && grid_grad && warp && rays && adam) {
    const fnr_grid grid_ = *grid_grad;
    const fnr_warp warp_ = *warp;
    const fnr_rays rays_ = *rays;
    const fnr_table_adam adam_ = *adam;
    seq::push("fnr_hash_encode_bwd_adam", [=](const fnr_step_scalars* sc) {
      const fnr_table_adam a = seq::patched(adam_, sc);
      return fnr_hash_encode_bwd_adam(&grid_, &warp_, &rays_, euclid_bins, S, d_feats, workspace, workspace_bytes,
                                      workspace_clean, &a, stream);
    });
  }
  FNR_CHECK_ARG(grid_grad && warp && rays && euclid_bins && d_feats && S > 0, "hash_encode_bwd_adam: null argument");
  FNR_CHECK_ARG(grid_grad->n_levels >= 1 && grid_grad->n_levels <= FNR_MAX_LEVELS, "hash_encode_bwd_adam: n_levels");
  TableAdam t;
  const int rc = make_table_adam(adam, t);
  if (rc) return rc;
  const long long N = rays->n_rays * (long long)S;
  RaySource src{make_rays(rays), euclid_bins, S};
  FNR_PROF(OP_ENCODE_BWD, N);
  // N == 0 still takes the optimiser step (every row decays its moments): the scatter runs with empty queues
  return binned_scatter(grid_grad, make_warp(warp), src, N, reinterpret_cast<const float2*>(d_feats), 0,
                        grid_grad->n_levels, workspace, workspace_bytes, workspace_clean, as_stream(stream), &t);
}

extern "C" size_t fnr_prop_density_bwd_workspace_bytes(int64_t n_samples, int n_levels, int log2_hashmap_size) {
  const size_t dfeat = ((size_t)n_levels * (size_t)n_samples * sizeof(float2) + 255) / 256 * 256;
  const size_t partials = (size_t)3 * device_cu_count() * PROP_PART * sizeof(float);
  return dfeat + partials + fnr_hash_scatter_workspace_bytes(n_samples, n_levels, log2_hashmap_size);
}

static int prop_density_bwd_entry(const fnr_prop_net* net, const fnr_prop_net* grads, const fnr_warp* warp,
                                    const fnr_rays* rays, const float* euclid_bins, int S, const float* feat_save,
                                    const float* d_density, float* d_position, void* workspace,
                                    size_t workspace_bytes, int workspace_clean, void* stream,
                                    const fnr_table_adam* table_adam, const fnr_table_adam* weight_adam,
                                    const float* grad_arena, AccArgs* defer_acc = nullptr, int phase = 0,
                                    bool own_scope = true, PropReduceArgs* defer_reduce = nullptr) {
  // own_scope false: the caller's ProfScope spans this call (the paired entry points open ONE scope over both levels and
  // their joint accumulate launch, which runs outside any per-level call)
  // phase 0: the whole backward; 1: MLP backward + weight reduction only (d_feats stay in the workspace, d_position is
  // complete); 2: the scatter of what phase 1 left (fnr_prop_density_bwd_pair_split)
  FNR_CHECK_ARG(net && grads && warp && rays && euclid_bins && feat_save && d_density && workspace && S > 0,
                "prop_density_bwd: null argument");
  WeightAdam wa{};
  TableAdam ta;
  if (table_adam) {
    FNR_CHECK_ARG(weight_adam && grad_arena, "prop_density_bwd_adam: weight_adam / grad_arena missing");
    int rca = make_table_adam(weight_adam, wa.t);
    if (rca) return rca;
    wa.p_off = weight_adam->params - grad_arena;
    wa.m_off = weight_adam->exp_avg - grad_arena;
    wa.v_off = weight_adam->exp_avg_sq - grad_arena;
    rca = make_table_adam(table_adam, ta);
    if (rca) return rca;
  }
  FNR_UNSUPPORTED(net->hidden_dim == 16, "prop_density_bwd: hidden_dim %d not built (16 only)", net->hidden_dim);
  FNR_CHECK_ARG(grads->grid.table && grads->w0 && grads->b0 && grads->w1 && grads->b1,
                "prop_density_bwd: null gradient pointer");
  const long long N = rays->n_rays * (long long)S;
  if (N == 0) return FNR_OK;
  const int L = net->grid.n_levels;
  FNR_CHECK_ARG(workspace_bytes >= fnr_prop_density_bwd_workspace_bytes(N, L, net->grid.log2_hashmap_size),
                "prop_density_bwd: workspace too small");
  RaySource src{make_rays(rays), euclid_bins, S};
  long long blocks = (N + 255) / 256;
  const long long max_blocks = 3ll * device_cu_count();  // 3 x 46 KiB LDS per CU; fewer workgroups = fewer dW atomics
  if (blocks > max_blocks) blocks = max_blocks;
  {
    // A/B knob (round 5): FNR_PROP_BWD_WGS_PER_CU = 1 | 2 caps the PERSISTENT workgroups of k_prop_bwd below the three a CU
    // holds.  On a second stream the kernel fills every CU for its whole 75 - 140 us and the launch stream's short kernels
    // wait for it to END (kernel trace of the two-stream step, profiles/r04_raw/prof_step_timeline_two_streams.txt:
    // k_color_ray_grads 11 -> 75 us, the base backward 50 -> 95 us next to it); with fewer resident workgroups it runs
    // longer itself but leaves wave slots and LDS to the other queue.  Same partial sums per workgroup, summed by
    // k_prop_reduce in workgroup order: the weight gradients change in the last bits with the workgroup count.
    static const int cap_per_cu = [] { const char* e = getenv("FNR_PROP_BWD_WGS_PER_CU"); const int v = e ? atoi(e) : 0; return (v >= 1 && v <= 3) ? v : 3; }();
    const long long capped = (long long)cap_per_cu * device_cu_count();
    if (blocks > capped) blocks = capped;
  }
  Warp w = make_warp(warp);
  const float2* fs = reinterpret_cast<const float2*>(feat_save);
  float2* d_feats = reinterpret_cast<float2*>(workspace);
  const size_t dfeat_bytes = ((size_t)L * (size_t)N * sizeof(float2) + 255) / 256 * 256;
  float* partials = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + dfeat_bytes);
  const size_t partial_bytes = (size_t)max_blocks * PROP_PART * sizeof(float);
  ::fnr::ProfScope prof_scope__(own_scope ? (int)OP_PROP_BWD : -1, N, stream);
  if (phase != 2) {
  const GridDev pgrid = make_grid(&net->grid);
  float4* d_xw = reinterpret_cast<float4*>(d_position);
#define FNR_PROPB_CASE(LL)                                                                                          \
  case LL:                                                                                                          \
    if (d_xw)                                                                                                       \
      hipLaunchKernelGGL((k_prop_bwd<LL, 16, true>), dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), pgrid, \
                         d_xw, w, src, N, net->w0, net->b0, net->w1, net->b1, fs, d_density, d_feats, partials);    \
    else                                                                                                            \
      hipLaunchKernelGGL((k_prop_bwd<LL, 16, false>), dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), pgrid, \
                         d_xw, w, src, N, net->w0, net->b0, net->w1, net->b1, fs, d_density, d_feats, partials);    \
    break;
  switch (L) {
    FNR_PROPB_CASE(1)
    FNR_PROPB_CASE(2)
    FNR_PROPB_CASE(3)
    FNR_PROPB_CASE(4)
    FNR_PROPB_CASE(5)
    FNR_PROPB_CASE(6)
    FNR_PROPB_CASE(7)
    FNR_PROPB_CASE(8)
    default:
      FNR_UNSUPPORTED(false, "prop_density_bwd: n_levels %d not built (1..8)", L);
  }
#undef FNR_PROPB_CASE
  FNR_LAUNCH_CHECK();
  const PropReduceArgs ra{partials, (int)blocks, 2 * L, grads->w0, grads->b0, grads->w1, grads->b1, table_adam ? wa : WeightAdam{}};
  if (defer_reduce) {   // the caller launches the reduction (together with the other level's)
    *defer_reduce = ra;
  } else {
    if (table_adam) hipLaunchKernelGGL(k_prop_reduce<true>, dim3(PROP_PART / PRD_E), dim3(PRD_E * PRD_Y), 0, as_stream(stream), ra);
    else hipLaunchKernelGGL(k_prop_reduce<false>, dim3(PROP_PART / PRD_E), dim3(PRD_E * PRD_Y), 0, as_stream(stream), ra);
    FNR_LAUNCH_CHECK();
  }
  }
  if (phase == 1) return FNR_OK;
  AccArgs acc;
  const int rce = scatter_emit(&grads->grid, w, src, N, d_feats, 0, L,
                               reinterpret_cast<char*>(workspace) + dfeat_bytes + partial_bytes,
                               workspace_bytes - dfeat_bytes - partial_bytes, workspace_clean, as_stream(stream),
                               table_adam ? &ta : nullptr, acc);
  if (rce) return rce;
  acc.kind = 1;
  if (defer_acc) {   // the caller launches the accumulate kernel (together with another call's)
    *defer_acc = acc;
    return FNR_OK;
  }
  return scatter_accumulate(acc, table_adam != nullptr, as_stream(stream));
}

extern "C" int fnr_prop_density_bwd(const fnr_prop_net* net, const fnr_prop_net* grads, const fnr_warp* warp,
                                    const fnr_rays* rays, const float* euclid_bins, int S, const float* feat_save,
                                    const float* d_density, float* d_position, void* workspace,
                                    size_t workspace_bytes, int workspace_clean, void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_prop_density_bwd");
  return prop_density_bwd_entry(net, grads, warp, rays, euclid_bins, S, feat_save, d_density, d_position, workspace,
                                workspace_bytes, workspace_clean, stream, nullptr, nullptr, nullptr);
}

extern "C" int fnr_prop_density_bwd_adam(const fnr_prop_net* net, const fnr_prop_net* grads, const fnr_warp* warp,
                                         const fnr_rays* rays, const float* euclid_bins, int S, const float* feat_save,
                                         const float* d_density, float* d_position, const fnr_table_adam* table_adam,
                                         const fnr_table_adam* weight_adam, const float* grad_arena, void* workspace,
                                         size_t workspace_bytes, int workspace_clean, void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_prop_density_bwd_adam");
  FNR_CHECK_ARG(table_adam && weight_adam && grad_arena, "prop_density_bwd_adam: optimiser descriptors missing");
  return prop_density_bwd_entry(net, grads, warp, rays, euclid_bins, S, feat_save, d_density, d_position, workspace,
                                workspace_bytes, workspace_clean, stream, table_adam, weight_adam, grad_arena);
}

// Both proposal levels of a training step (two networks, two sets of samples) as one entry point: their MLP backward,
// weight reduction and emit launches run one after the other, their accumulate launches as ONE (k_scatter_accumulate2:
// 160 workgroups of 64 KiB each per level on 256 CUs — side by side instead of one after the other).  table_adam /
// weight_adam all NULL (gradients are left in `grads`) or all set (fnr_prop_density_bwd_adam semantics per network).
extern "C" int fnr_prop_density_bwd_pair(const fnr_prop_net* const* nets, const fnr_prop_net* const* grads,
                                         const fnr_warp* const* warps, const fnr_rays* rays,
                                         const float* const* euclid_bins, const int* S, const float* const* feat_save,
                                         const float* const* d_density, float* const* d_position,
                                         const fnr_table_adam* const* table_adam, const fnr_table_adam* weight_adam,
                                         const float* grad_arena, void* const* workspace, const size_t* workspace_bytes,
                                         const int* workspace_clean, void* stream) {
  if (seq::recording() && nets && grads && warps && rays && euclid_bins && S && feat_save && d_density && d_position &&
      workspace && workspace_bytes && workspace_clean && nets[0] && nets[1] && grads[0] && grads[1] && warps[0] && warps[1]) {
    struct Pair {
      fnr_prop_net net[2], grad[2];
      fnr_warp warp[2];
      fnr_rays rays;
      const float* euclid[2];
      int S[2];
      const float* feat[2];
      const float* dd[2];
      float* dpos[2];
      fnr_table_adam tadam[2], wadam;
      bool has_adam;
      void* ws[2];
      size_t ws_bytes[2];
      int ws_clean[2];
    } p;
    for (int q = 0; q < 2; ++q) {
      p.net[q] = *nets[q], p.grad[q] = *grads[q], p.warp[q] = *warps[q], p.euclid[q] = euclid_bins[q], p.S[q] = S[q];
      p.feat[q] = feat_save[q], p.dd[q] = d_density[q], p.dpos[q] = d_position[q], p.ws[q] = workspace[q];
      p.ws_bytes[q] = workspace_bytes[q], p.ws_clean[q] = workspace_clean[q];
    }
    p.rays = *rays;
    p.has_adam = table_adam && table_adam[0] && table_adam[1] && weight_adam;
    if (p.has_adam) p.tadam[0] = *table_adam[0], p.tadam[1] = *table_adam[1], p.wadam = *weight_adam;
    seq::push("fnr_prop_density_bwd_pair", [=](const fnr_step_scalars* sc) {
      const fnr_prop_net* n_[2] = {&p.net[0], &p.net[1]};
      const fnr_prop_net* g_[2] = {&p.grad[0], &p.grad[1]};
      const fnr_warp* w_[2] = {&p.warp[0], &p.warp[1]};
      const fnr_table_adam t0 = seq::patched(p.tadam[0], sc), t1 = seq::patched(p.tadam[1], sc), wa = seq::patched(p.wadam, sc);
      const fnr_table_adam* t_[2] = {&t0, &t1};
      return fnr_prop_density_bwd_pair(n_, g_, w_, &p.rays, p.euclid, p.S, p.feat, p.dd, p.dpos, p.has_adam ? t_ : nullptr,
                      p.has_adam ? &wa : nullptr, grad_arena, p.ws, p.ws_bytes, p.ws_clean, stream);
    });
  }
  FNR_CHECK_ARG(nets && grads && warps && rays && euclid_bins && S && feat_save && d_density && d_position && workspace &&
                    workspace_bytes && workspace_clean,
                "prop_density_bwd_pair: null argument");
  FNR_CHECK_ARG(nets[0] != nets[1] && grads[0] != grads[1] && workspace[0] != workspace[1],
                "prop_density_bwd_pair: the two levels must have their own network, gradients and workspace");
  const bool adam = table_adam && table_adam[0] && table_adam[1];
  FNR_CHECK_ARG(adam || !(table_adam && (table_adam[0] || table_adam[1])), "prop_density_bwd_pair: one table_adam missing");
  if (rays->n_rays == 0) return FNR_OK;
  FNR_PROF(OP_PROP_BWD, rays->n_rays * ((long long)S[0] + (long long)S[1]));   // one scope: both levels + the joint accumulate
  AccArgs acc[2];
  for (int q = 0; q < 2; ++q) {
    const int rc = prop_density_bwd_entry(nets[q], grads[q], warps[q], rays, euclid_bins[q], S[q], feat_save[q], d_density[q],
                                          d_position[q], workspace[q], workspace_bytes[q], workspace_clean[q], stream,
                                          adam ? table_adam[q] : nullptr, adam ? weight_adam : nullptr,
                                          adam ? grad_arena : nullptr, &acc[q], 0, false);
    if (rc) return rc;
  }
  const bool first_longer = (long long)S[0] >= (long long)S[1];
  return scatter_accumulate2(first_longer ? acc[0] : acc[1], first_longer ? acc[1] : acc[0], adam, as_stream(stream));
}

// fnr_prop_density_bwd_pair with the launches in two groups: both levels' MLP backward + weight reduction first — after
// them d_position[0..1] are final and `position_ready_event` (a hipEvent_t, optional) is recorded on `stream` — then both
// levels' emit launches and the joint accumulate.  A caller with a second stream can finish the ray gradients and take
// the camera optimiser's step next to the ~210 us of scatter that follow.  Same launches, same results.
extern "C" int fnr_prop_density_bwd_pair_split(const fnr_prop_net* const* nets, const fnr_prop_net* const* grads,
                                               const fnr_warp* const* warps, const fnr_rays* rays,
                                               const float* const* euclid_bins, const int* S, const float* const* feat_save,
                                               const float* const* d_density, float* const* d_position,
                                               const fnr_table_adam* const* table_adam, const fnr_table_adam* weight_adam,
                                               const float* grad_arena, void* const* workspace, const size_t* workspace_bytes,
                                               const int* workspace_clean, void* stream, void* position_ready_event) {
  if (seq::recording() && nets && grads && warps && rays && euclid_bins && S && feat_save && d_density && d_position &&
      workspace && workspace_bytes && workspace_clean && nets[0] && nets[1] && grads[0] && grads[1] && warps[0] && warps[1]) {
    struct Pair {
      fnr_prop_net net[2], grad[2];
      fnr_warp warp[2];
      fnr_rays rays;
      const float* euclid[2];
      int S[2];
      const float* feat[2];
      const float* dd[2];
      float* dpos[2];
      fnr_table_adam tadam[2], wadam;
      bool has_adam;
      void* ws[2];
      size_t ws_bytes[2];
      int ws_clean[2];
    } p;
    for (int q = 0; q < 2; ++q) {
      p.net[q] = *nets[q], p.grad[q] = *grads[q], p.warp[q] = *warps[q], p.euclid[q] = euclid_bins[q], p.S[q] = S[q];
      p.feat[q] = feat_save[q], p.dd[q] = d_density[q], p.dpos[q] = d_position[q], p.ws[q] = workspace[q];
      p.ws_bytes[q] = workspace_bytes[q], p.ws_clean[q] = workspace_clean[q];
    }
    p.rays = *rays;
    p.has_adam = table_adam && table_adam[0] && table_adam[1] && weight_adam;
    if (p.has_adam) p.tadam[0] = *table_adam[0], p.tadam[1] = *table_adam[1], p.wadam = *weight_adam;
    seq::push("fnr_prop_density_bwd_pair_split", [=](const fnr_step_scalars* sc) {
      const fnr_prop_net* n_[2] = {&p.net[0], &p.net[1]};
      const fnr_prop_net* g_[2] = {&p.grad[0], &p.grad[1]};
      const fnr_warp* w_[2] = {&p.warp[0], &p.warp[1]};
      const fnr_table_adam t0 = seq::patched(p.tadam[0], sc), t1 = seq::patched(p.tadam[1], sc), wa = seq::patched(p.wadam, sc);
      const fnr_table_adam* t_[2] = {&t0, &t1};
      return fnr_prop_density_bwd_pair_split(n_, g_, w_, &p.rays, p.euclid, p.S, p.feat, p.dd, p.dpos, p.has_adam ? t_ : nullptr,
                      p.has_adam ? &wa : nullptr, grad_arena, p.ws, p.ws_bytes, p.ws_clean, stream, position_ready_event);
    });
  }
  FNR_CHECK_ARG(nets && grads && warps && rays && euclid_bins && S && feat_save && d_density && d_position && workspace &&
                    workspace_bytes && workspace_clean,
                "prop_density_bwd_pair_split: null argument");
  FNR_CHECK_ARG(nets[0] != nets[1] && grads[0] != grads[1] && workspace[0] != workspace[1],
                "prop_density_bwd_pair_split: the two levels must have their own network, gradients and workspace");
  const bool adam = table_adam && table_adam[0] && table_adam[1];
  FNR_CHECK_ARG(adam || !(table_adam && (table_adam[0] || table_adam[1])), "prop_density_bwd_pair_split: one table_adam missing");
  if (rays->n_rays == 0) {
    if (position_ready_event) FNR_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(position_ready_event), as_stream(stream)));
    return FNR_OK;
  }
  FNR_PROF(OP_PROP_BWD, rays->n_rays * ((long long)S[0] + (long long)S[1]));   // one scope: both phases + the joint accumulate
  AccArgs acc[2];
  PropReduceArgs red[2];
  for (int phase = 1; phase <= 2; ++phase) {
    for (int q = 0; q < 2; ++q) {
      const int rc = prop_density_bwd_entry(nets[q], grads[q], warps[q], rays, euclid_bins[q], S[q], feat_save[q], d_density[q],
                                            d_position[q], workspace[q], workspace_bytes[q], workspace_clean[q], stream,
                                            adam ? table_adam[q] : nullptr, adam ? weight_adam : nullptr,
                                            adam ? grad_arena : nullptr, &acc[q], phase, false, phase == 1 ? &red[q] : nullptr);
      if (rc) return rc;
    }
    if (phase == 1) {
      // d_position is final behind the two MLP backward launches; their weight reductions (+ optimiser steps) are ONE launch
      if (position_ready_event) FNR_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(position_ready_event), as_stream(stream)));
      constexpr unsigned NB = PROP_PART / PRD_E;
      if (adam) hipLaunchKernelGGL(k_prop_reduce2<true>, dim3(2 * NB), dim3(PRD_E * PRD_Y), 0, as_stream(stream), red[0], red[1]);
      else hipLaunchKernelGGL(k_prop_reduce2<false>, dim3(2 * NB), dim3(PRD_E * PRD_Y), 0, as_stream(stream), red[0], red[1]);
      FNR_LAUNCH_CHECK();
    }
  }
  const bool first_longer = (long long)S[0] >= (long long)S[1];
  return scatter_accumulate2(first_longer ? acc[0] : acc[1], first_longer ? acc[1] : acc[0], adam, as_stream(stream));
}
