// export.hip — sample_volume's threshold masks + boolean gathers (export/exporter_utils.py:111-153) as an
// ORDER-PRESERVING three-stream compaction on gfx950: flag/count -> scan -> scatter.  Order preservation
// makes the exported point lists identical (not just equal in count) to the reference's per-batch
// boolean-mask indexing.  HBM-bound: 20 B read per sample (density, rgb, logit), twice.
#include "common.hpp"
#include "sequencer.hpp"

namespace fnr {

constexpr int EXP_BLOCK = 256;
constexpr int EXP_PER_THREAD = 4;
constexpr int EXP_TILE = EXP_BLOCK * EXP_PER_THREAD;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// bit 0: semantic_colormap set, bit 1: semantic set, bit 2: density set
__device__ __forceinline__ unsigned export_flags(float den, float lg) {
  const bool m_den = den >= 70.0f;                           // exporter_utils.py:112
  const bool m_sem = lg >= 3.0f;                             // exporter_utils.py:111
  const bool m_cm = fsub(sigmoidf_(lg), 0.9f) > 0.0f;        // heaviside(sigmoid - 0.9, 0) (fruit_nerf.py:263-265) >= 0.999
  return (m_den && m_cm ? 1u : 0u) | (m_den && m_sem ? 2u : 0u) | (m_den ? 4u : 0u);
}

__global__ __launch_bounds__(EXP_BLOCK) void k_export_count(long long N, const float* __restrict__ density,
                                                            const float* __restrict__ logit,
                                                            unsigned long long* __restrict__ block_counts,
                                                            long long nblocks) {
  __shared__ unsigned s_cnt[3];
  if (threadIdx.x < 3) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const long long base = (long long)blockIdx.x * EXP_TILE + (long long)threadIdx.x * EXP_PER_THREAD;
  unsigned c0 = 0, c1 = 0, c2 = 0;
#pragma unroll
  for (int q = 0; q < EXP_PER_THREAD; ++q) {
    const long long n = base + q;
    if (n < N) {
      const unsigned f = export_flags(density[n], logit[n]);
      c0 += f & 1u;
      c1 += (f >> 1) & 1u;
      c2 += (f >> 2) & 1u;
    }
  }
  if (c0) atomicAdd(&s_cnt[0], c0);
  if (c1) atomicAdd(&s_cnt[1], c1);
  if (c2) atomicAdd(&s_cnt[2], c2);
  __syncthreads();
  if (threadIdx.x < 3) block_counts[(size_t)threadIdx.x * nblocks + blockIdx.x] = s_cnt[threadIdx.x];
}

// exclusive scan of block_counts per set (in place), offset by the running totals, which are advanced
__global__ __launch_bounds__(1024) void k_export_scan(unsigned long long* __restrict__ block_counts, long long nblocks,
                                                      unsigned long long* __restrict__ counts) {
  __shared__ unsigned long long s_wave[16];
  __shared__ unsigned long long s_carry;
  const int set = blockIdx.x;
  unsigned long long* bc = block_counts + (size_t)set * nblocks;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_carry = counts[set];
  __syncthreads();
  for (long long start = 0; start < nblocks; start += 1024) {
    const long long i = start + threadIdx.x;
    unsigned long long v = (i < nblocks) ? bc[i] : 0ull;
    unsigned long long incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      unsigned long long t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned long long wave_off = 0;
    for (int q = 0; q < wave; ++q) wave_off += s_wave[q];
    const unsigned long long carry = s_carry;
    if (i < nblocks) bc[i] = carry + wave_off + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = carry + wave_off + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) counts[set] = s_carry;
}

struct ExportOut {
  float* points[3];
  float* colors[3];
};

__global__ __launch_bounds__(EXP_BLOCK) void k_export_write(long long N, int n_y, int n_z, long long ray_begin,
                                                            const float* __restrict__ xs, const float* __restrict__ ys,
                                                            const float* __restrict__ zs,
                                                            const float* __restrict__ positions,
                                                            const float* __restrict__ density,
                                                            const float* __restrict__ rgb,
                                                            const float* __restrict__ logit, ExportOut out,
                                                            long long capacity,
                                                            const unsigned long long* __restrict__ block_offsets,
                                                            long long nblocks) {
  __shared__ unsigned s_wave[3][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long base = (long long)blockIdx.x * EXP_TILE + (long long)threadIdx.x * EXP_PER_THREAD;
  unsigned fl[EXP_PER_THREAD];
  unsigned cnt[3] = {0, 0, 0};
#pragma unroll
  for (int q = 0; q < EXP_PER_THREAD; ++q) {
    const long long n = base + q;
    fl[q] = (n < N) ? export_flags(density[n], logit[n]) : 0u;
#pragma unroll
    for (int s = 0; s < 3; ++s) cnt[s] += (fl[q] >> s) & 1u;
  }
  unsigned pre[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    unsigned incl = cnt[s];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      unsigned t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    pre[s] = incl - cnt[s];
    if (lane == 63) s_wave[s][wave] = incl;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 3; ++s)
    for (int q = 0; q < wave; ++q) pre[s] += s_wave[s][q];

  unsigned long long pos[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) pos[s] = block_offsets[(size_t)s * nblocks + blockIdx.x] + pre[s];

#pragma unroll
  for (int q = 0; q < EXP_PER_THREAD; ++q) {
    if (!fl[q]) continue;
    const long long n = base + q;
    float px, py, pz;
    if (positions) {
      px = positions[3 * n];
      py = positions[3 * n + 1];
      pz = positions[3 * n + 2];
    } else {
      long long r = n / n_z;
      const int k = (int)(n - r * n_z);
      r += ray_begin;
      const long long ix = r / n_y;
      const int iy = (int)(r - ix * n_y);
      px = xs[ix];
      py = ys[iy];
      pz = zs[k];
    }
    const float c0 = rgb[3 * n], c1 = rgb[3 * n + 1], c2 = rgb[3 * n + 2];
    const float lg = logit[n], den = density[n];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      if ((fl[q] >> s) & 1u) {
        const unsigned long long p = pos[s]++;
        if ((long long)p < capacity) {
          float* pt = out.points[s] + 3 * p;
          pt[0] = px;
          pt[1] = py;
          pt[2] = pz;
          float* cl = out.colors[s] + 4 * p;
          cl[0] = c0;
          cl[1] = c1;
          cl[2] = c2;
          cl[3] = sigmoidf_(s == 2 ? den : lg);  // exporter_utils.py:124-125,136-137,147-148
        }
      }
    }
  }
}

}  // namespace fnr

using namespace fnr;

extern "C" size_t fnr_export_workspace_bytes(int64_t n_samples) {
  const long long nblocks = (n_samples + EXP_TILE - 1) / EXP_TILE;
  return (size_t)(nblocks > 0 ? nblocks : 1) * 3 * sizeof(unsigned long long);
}

extern "C" int fnr_export_compact(const fnr_lattice* lat, int64_t ray_begin, int64_t n_rays,
                                  const float* positions, int64_t n_positions, const float* density,
                                  const float* rgb, const float* logit, float* const points[3],
                                  float* const colors[3], int64_t capacity, uint64_t* counts, void* workspace,
                                  void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_export_compact");
  FNR_CHECK_ARG(density && rgb && logit && points && colors && counts && workspace, "export_compact: null argument");
  FNR_CHECK_ARG((lat != nullptr) != (positions != nullptr), "export_compact: give either a lattice or positions");
  long long N;
  fnr_lattice nolat = {1, 1, 1, nullptr, nullptr, nullptr};
  if (lat) {
    FNR_CHECK_ARG(ray_begin >= 0 && n_rays >= 0 && ray_begin + n_rays <= (int64_t)lat->n_x * lat->n_y,
                  "export_compact: ray range outside lattice");
    N = n_rays * (long long)lat->n_z;
  } else {
    FNR_CHECK_ARG(n_positions >= 0, "export_compact: n_positions < 0");
    N = n_positions;
    lat = &nolat;
  }
  if (N == 0) return FNR_OK;
  const long long nblocks = (N + EXP_TILE - 1) / EXP_TILE;
  FNR_CHECK_ARG(nblocks < (1ll << 31), "export_compact: batch too large");
  ExportOut out;
  for (int s = 0; s < 3; ++s) {
    FNR_CHECK_ARG(points[s] && colors[s], "export_compact: null output stream %d", s);
    out.points[s] = points[s];
    out.colors[s] = colors[s];
  }
  unsigned long long* bc = reinterpret_cast<unsigned long long*>(workspace);
  hipStream_t st = as_stream(stream);
  FNR_PROF(OP_EXPORT_COMPACT, N);
  hipLaunchKernelGGL(k_export_count, dim3((unsigned)nblocks), dim3(EXP_BLOCK), 0, st, N, density, logit, bc, nblocks);
  FNR_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_export_scan, dim3(3), dim3(1024), 0, st, bc, nblocks,
                     reinterpret_cast<unsigned long long*>(counts));
  FNR_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_export_write, dim3((unsigned)nblocks), dim3(EXP_BLOCK), 0, st, N, lat->n_y, lat->n_z,
                     (long long)ray_begin, lat->xs, lat->ys, lat->zs, positions, density, rgb, logit, out, (long long)capacity,
                     bc, nblocks);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}
