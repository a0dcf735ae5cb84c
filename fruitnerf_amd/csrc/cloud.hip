// cloud.hip — point-cloud front-end of the fruit counting stage on gfx950 (fp64, integer/index work, no MFMA):
//   radius-outlier neighbour counts   (Open3D remove_radius_outlier, clustering/clustering_base.py:141-143)
//   voxel down-sampling               (Open3D voxel_down_sample,     clustering/clustering_base.py:138-139)
//   DBSCAN labels                     (sklearn.cluster.DBSCAN,       clustering/clustering_base.py:199-200)
//
// One search structure serves the two neighbourhood queries: points are sorted by the key of a uniform grid whose
// cell edge is (just above) the search radius, with x fastest in the key, so the 27 neighbour cells of a point are
// 9 CONTIGUOUS runs of the sorted array, each found with two binary searches.  A wave owns one occupied cell: its lanes
// hold the cell's points, the candidates are walked with a wave-uniform index (scalar loads) and tested in fp64 with
// the operation order of the CPU libraries (((dx*dx)+dy*dy)+dz*dz, no FMA), so the integer results (counts, masks,
// labels) are bit-exact.  The cost is the pair tests: fp64 VALU bound (10 instructions per candidate and wave).
//
// DBSCAN = (1) inclusive neighbour counts -> core flags, (2) lock-free union-find over core-core pairs in ORIGINAL
// index space, always hooking the larger root under the smaller, so a cluster's root is its first core point in input
// order, (3) cluster numbers = rank of the roots (prefix sum in input order) = scikit-learn's numbering,
// (4) border points take the smallest cluster number among the core points within eps (the first cluster to reach
// them in scikit-learn's index-ordered expansion), everything else is noise (-1).
#include <cstring>

#include "common.hpp"
#include "sequencer.hpp"

#include <rocprim/rocprim.hpp>

namespace fnr {

typedef unsigned long long u64;

// one sorted point: position + what the sweeps need to know about it as a CANDIDATE, so that a candidate is one 32-byte
// scalar load (input index and core flag ride in what would be padding)
struct alignas(32) CloudPoint {
  double x, y, z;
  int index;   // input index (order[s])
  int core;    // DBSCAN core flag, filled in after the count pass
};

struct CloudGrid {
  double lo[3];
  double cell;
  long long dim[3];
};

struct CloudWs {
  u64* keys_a;
  u64* keys;      // sorted
  int* idx_a;
  int* order;     // sorted position -> input index
  CloudPoint* sxyz;  // [n] points in sorted order
  int* aux0;      // per sorted position: core flag / head flag
  int* aux1;      // per input index: union-find parent
  int* aux2;      // per input index: core flag, then root flag
  int* aux3;      // scan output
  int* cell_first;  // [n_cells + 1] first sorted position of every occupied cell
  int* ctrl;      // [0] n_cells, [1..7] work counters of the cell sweeps
  void* temp;
  size_t temp_bytes;
};

static inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

static size_t cloud_fixed_bytes(int64_t n) {
  const size_t m = (size_t)(n < 1 ? 1 : n);
  return 2 * align256(m * 8) + 2 * align256(m * 4) + align256(m * 32) + 4 * align256(m * 4) + align256((m + 1) * 4) + 256;
}
static size_t cloud_temp_reserve(int64_t n) { return (size_t)(n < 1 ? 1 : n) * 16 + ((size_t)8 << 20); }

static bool carve_cloud(void* ws, size_t bytes, int64_t n, CloudWs* out) {
  if (!ws || bytes < cloud_fixed_bytes(n) + ((size_t)1 << 20)) return false;
  const size_t m = (size_t)(n < 1 ? 1 : n);
  char* p = static_cast<char*>(ws);
  out->keys_a = reinterpret_cast<u64*>(p);  p += align256(m * 8);
  out->keys = reinterpret_cast<u64*>(p);    p += align256(m * 8);
  out->idx_a = reinterpret_cast<int*>(p);   p += align256(m * 4);
  out->order = reinterpret_cast<int*>(p);   p += align256(m * 4);
  out->sxyz = reinterpret_cast<CloudPoint*>(p); p += align256(m * 32);
  out->aux0 = reinterpret_cast<int*>(p);    p += align256(m * 4);
  out->aux1 = reinterpret_cast<int*>(p);    p += align256(m * 4);
  out->aux2 = reinterpret_cast<int*>(p);    p += align256(m * 4);
  out->aux3 = reinterpret_cast<int*>(p);    p += align256(m * 4);
  out->cell_first = reinterpret_cast<int*>(p); p += align256((m + 1) * 4);
  out->ctrl = reinterpret_cast<int*>(p);    p += 256;
  out->temp = p;
  out->temp_bytes = bytes - (size_t)(p - static_cast<char*>(ws));
  return true;
}

// ---- axis-aligned bounds (Open3D GetMinBound / GetMaxBound) ---------------------------------------------------------
// doubles are mapped to unsigned keys with the same order, reduced per wave with shuffles, then one atomic per wave
__device__ inline u64 order_key(double v) {
  const u64 b = (u64)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ inline double key_value(u64 k) {
  const u64 b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}

__global__ __launch_bounds__(256) void k_cloud_bounds(const double* __restrict__ xyz, int n, u64* keys6) {
  __shared__ u64 part[4][6];
  u64 lo[3] = {~0ull, ~0ull, ~0ull}, hi[3] = {0ull, 0ull, 0ull};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const u64 key = order_key(xyz[3 * (size_t)i + k]);
      lo[k] = key < lo[k] ? key : lo[k];
      hi[k] = key > hi[k] ? key : hi[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    for (int off = 32; off > 0; off >>= 1) {
      const u64 a = __shfl_xor(lo[k], off), b = __shfl_xor(hi[k], off);
      lo[k] = a < lo[k] ? a : lo[k];
      hi[k] = b > hi[k] ? b : hi[k];
    }
    if ((threadIdx.x & 63u) == 0) {
      part[threadIdx.x >> 6][k] = lo[k];
      part[threadIdx.x >> 6][3 + k] = hi[k];
    }
  }
  __syncthreads();
  // the six results share one cache line, where atomics retire one at a time (~12 ns): one set per WORKGROUP, and
  // the launch is capped at one workgroup per CU
  if (threadIdx.x < 6) {
    const int k = (int)threadIdx.x;
    u64 v = part[0][k];
    for (int w = 1; w < 4; ++w) v = k < 3 ? (part[w][k] < v ? part[w][k] : v) : (part[w][k] > v ? part[w][k] : v);
    if (k < 3) atomicMin(&keys6[k], v); else atomicMax(&keys6[k], v);
  }
}

__global__ void k_cloud_bounds_decode(const u64* __restrict__ keys6, double* __restrict__ out6) {
  if (threadIdx.x < 6) out6[threadIdx.x] = key_value(keys6[threadIdx.x]);
}

__device__ inline long long cell_index(double p, double lo, double cell, long long dim) {
  long long c = (long long)floor((p - lo) / cell);
  return c < 0 ? 0 : (c >= dim ? dim - 1 : c);
}

__global__ __launch_bounds__(256) void k_cloud_keys(const double* __restrict__ xyz, int n, CloudGrid g,
                                                    u64* __restrict__ keys, int* __restrict__ idx) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long long cx = cell_index(xyz[3 * (size_t)i + 0], g.lo[0], g.cell, g.dim[0]);
  const long long cy = cell_index(xyz[3 * (size_t)i + 1], g.lo[1], g.cell, g.dim[1]);
  const long long cz = cell_index(xyz[3 * (size_t)i + 2], g.lo[2], g.cell, g.dim[2]);
  keys[i] = (u64)((cz * g.dim[1] + cy) * g.dim[0] + cx);
  idx[i] = i;
}

__global__ __launch_bounds__(256) void k_cloud_gather(const double* __restrict__ xyz, const int* __restrict__ order,
                                                      int n, CloudPoint* __restrict__ sxyz) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= n) return;
  const int o = order[s];
  const size_t i = (size_t)o;
  CloudPoint pt;
  pt.x = xyz[3 * i + 0];
  pt.y = xyz[3 * i + 1];
  pt.z = xyz[3 * i + 2];
  pt.index = o;
  pt.core = 0;
  sxyz[s] = pt;
}

__global__ __launch_bounds__(256) void k_cloud_pack_core(const int* __restrict__ core_sorted, int n,
                                                         CloudPoint* __restrict__ sxyz) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s < n) sxyz[s].core = core_sorted[s];
}

__device__ inline int lower_bound_key(const u64* __restrict__ keys, int lo, int hi, u64 k) {
  while (lo < hi) {
    const int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
    if (keys[mid] < k) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__device__ inline int upper_bound_key(const u64* __restrict__ keys, int lo, int hi, u64 k) {
  while (lo < hi) {
    const int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
    if (keys[mid] <= k) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// squared distance in the CPU libraries' operation order: ((dx*dx) + dy*dy) + dz*dz, every op rounded
__device__ inline double pair_d2(const CloudPoint& p, const CloudPoint& q) {
  const double dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
  return (dx * dx + dy * dy) + dz * dz;
}

// ---- cell-cooperative neighbourhood sweep ---------------------------------------------------------------------------
// A wave takes one occupied cell at a time (dynamic: one atomic per cell).  The cell's 9 candidate runs are found once
// (lanes 0..8 search one run each) and live in SGPRs; the lanes hold up to 64 of the cell's points (the targets) and
// the wave walks the candidates with a wave-uniform index, so each candidate is ONE scalar load (32 B through the
// scalar cache) feeding 64 fp64 tests whose second operand is an SGPR pair: no vector-memory or LDS traffic in the
// inner loop, the fp64 VALU is the only busy unit.
struct CellRuns {
  int b[9], e[9];
};

__device__ inline int wave_lane() { return (int)(threadIdx.x & 63u); }

__device__ inline bool next_cell(int* counter, int n_cells, int* c) {
  int v = 0;
  if (wave_lane() == 0) v = atomicAdd(counter, 1);
  *c = __builtin_amdgcn_readfirstlane(v);
  return *c < n_cells;
}

__device__ inline CellRuns cell_runs(const u64* __restrict__ keys, int n, const CloudGrid& g, u64 key) {
  const long long nx = g.dim[0], ny = g.dim[1], nz = g.dim[2];
  const long long cx = (long long)(key % (u64)nx);
  const long long cy = (long long)((key / (u64)nx) % (u64)ny);
  const long long cz = (long long)(key / (u64)(nx * ny));
  const long long x0 = cx > 0 ? cx - 1 : 0, x1 = cx + 1 < nx ? cx + 1 : nx - 1;
  const int k = wave_lane() < 9 ? wave_lane() : 8;
  const long long yy = cy + (k % 3) - 1, zz = cz + (k / 3) - 1;
  int b = 0, e = 0;
  if (yy >= 0 && yy < ny && zz >= 0 && zz < nz) {
    const u64 row = (u64)((zz * ny + yy) * nx);
    b = lower_bound_key(keys, 0, n, row + (u64)x0);
    e = upper_bound_key(keys, b, n, row + (u64)x1);
  }
  CellRuns r;
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    r.b[j] = __builtin_amdgcn_readlane(b, j);
    r.e[j] = __builtin_amdgcn_readlane(e, j);
  }
  return r;
}

// counts[input index] = neighbours with d2 < r2 (INCL: <=), the point itself included.
// CORE: additionally core_sorted[s] / core_input[i] = count >= min_samples and parent[i] = i.
template <bool INCL, bool CORE>
__global__ __launch_bounds__(256) void k_cloud_count(const u64* __restrict__ keys, const CloudPoint* __restrict__ sxyz,
                                                     const int* __restrict__ order, int n, CloudGrid g, double r2,
                                                     const int* __restrict__ cell_first, int* ctrl, int pass,
                                                     int* __restrict__ counts, int min_samples,
                                                     int* __restrict__ core_sorted, int* __restrict__ core_input,
                                                     int* __restrict__ parent) {
  const int n_cells = ctrl[0];
  int c;
  while (next_cell(&ctrl[1 + pass], n_cells, &c)) {
    const int s0 = cell_first[c], s1 = cell_first[c + 1];
    const CellRuns runs = cell_runs(keys, n, g, keys[s0]);
    for (int base = s0; base < s1; base += 64) {
      const int s = base + wave_lane();
      const bool act = s < s1;
      const CloudPoint p = sxyz[act ? s : s0];
      int cnt = 0;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
#pragma unroll 4
        for (int t = runs.b[k]; t < runs.e[k]; ++t) {
          const double d2 = pair_d2(p, sxyz[t]);
          cnt += INCL ? (d2 <= r2) : (d2 < r2);
        }
      }
      if (act) {
        const int i = order[s];
        if (counts) counts[i] = cnt;
        if (CORE) {
          const int is_core = cnt >= min_samples;
          core_sorted[s] = is_core;
          core_input[i] = is_core;
          parent[i] = i;
        }
      }
    }
  }
}

// ---- lock-free union-find (roots only ever hook under SMALLER indices) -------------------------------------------
// device scope: L2 is the coherence point for the kernel's own traffic (system scope would go past it)
__device__ inline int uf_load(int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void uf_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ inline int uf_find(int* parent, int x) {
  while (true) {
    const int p = uf_load(&parent[x]);
    if (p == x) return x;
    const int gp = uf_load(&parent[p]);
    if (gp == p) return p;
    uf_store(&parent[x], gp);                             // path halving: gp is an ancestor of x, x is not a root
    x = gp;
  }
}

__device__ inline void uf_unite(int* parent, int a, int b) {
  while (true) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }
    if (atomicCAS(&parent[a], a, b) == a) return;        // a was still a root: hooked under the smaller root
  }
}

__global__ __launch_bounds__(256) void k_cloud_union(const u64* __restrict__ keys, const CloudPoint* __restrict__ sxyz,
                                                     const int* __restrict__ order, int n, CloudGrid g, double r2,
                                                     const int* __restrict__ cell_first, int* ctrl, int pass,
                                                     const int* __restrict__ core_sorted, int* parent) {
  const int n_cells = ctrl[0];
  int c;
  while (next_cell(&ctrl[1 + pass], n_cells, &c)) {
    const int s0 = cell_first[c], s1 = cell_first[c + 1];
    bool runs_ready = false;
    CellRuns runs;
    for (int base = s0; base < s1; base += 64) {
      const int s = base + wave_lane();
      const bool core = s < s1 && core_sorted[s] != 0;
      if (__ballot(core) == 0) continue;                  // no core target in this chunk
      if (!runs_ready) {
        runs = cell_runs(keys, n, g, keys[s0]);
        runs_ready = true;
      }
      const CloudPoint p = sxyz[s < s1 ? s : s0];
      const int i = core ? p.index : -1;
      int mine = core ? uf_find(parent, i) : -1;
      int known = mine;   // a node already proven to be in this target's set (sets only ever merge): typically the
                          // previous root that most neighbours still hang under after a merge moved the root
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        for (int t0 = runs.b[k]; t0 < runs.e[k]; t0 += 4) {
          // four candidates per round: one 32-byte scalar load each, then their (wave-uniform) parent words together,
          // so the common case — the candidate already hangs under this target's root — has no dependent round trip
          const int m = runs.e[k] - t0;
          CloudPoint q[4];
          int pj[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) q[u] = sxyz[t0 + (u < m ? u : 0)];
#pragma unroll
          for (int u = 0; u < 4; ++u) pj[u] = uf_load(&parent[q[u].index]);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            // only core-core pairs connect; every pair is handled from its larger end
            const bool pair = (u < m) & (q[u].core != 0) & core & (q[u].index < i) & (pair_d2(p, q[u]) <= r2) &
                              (pj[u] != mine) & (pj[u] != known);
            if (pair) {
              const int j = q[u].index;
              const int rj = uf_find(parent, j);
              if (rj != mine) {
                uf_unite(parent, mine, rj);
                mine = uf_find(parent, i);
              } else {
                known = pj[u];                            // j's parent is in this set: so is everything under it
                if (pj[u] != j) uf_store(&parent[j], rj); // j is not a root: hang it directly under the common root
              }
            }
          }
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_cloud_roots(int* parent, int* __restrict__ core_input_to_root_flag, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (core_input_to_root_flag[i]) {
    const int r = uf_find(parent, i);
    uf_store(&parent[i], r);                              // i is a non-root or r == i: never races with a hook
    core_input_to_root_flag[i] = (r == i);
  } else {
    parent[i] = -1;
  }
}

__global__ __launch_bounds__(256) void k_cloud_labels(const u64* __restrict__ keys, const CloudPoint* __restrict__ sxyz,
                                                      const int* __restrict__ order, int n, CloudGrid g, double r2,
                                                      const int* __restrict__ cell_first, int* ctrl, int pass,
                                                      const int* __restrict__ core_sorted,
                                                      const int* __restrict__ root, const int* __restrict__ root_flag,
                                                      const int* __restrict__ ids, int* __restrict__ labels,
                                                      int* __restrict__ n_clusters) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && n_clusters) *n_clusters = ids[n - 1] + root_flag[n - 1];
  const int n_cells = ctrl[0];
  int c;
  while (next_cell(&ctrl[1 + pass], n_cells, &c)) {
    const int s0 = cell_first[c], s1 = cell_first[c + 1];
    for (int base = s0; base < s1; base += 64) {
      const int s = base + wave_lane();
      const bool act = s < s1;
      const bool core = act && core_sorted[s] != 0;
      const bool border = act && !core;
      int best = 0x7fffffff;                               // smallest root = smallest cluster number
      if (__ballot(border) != 0) {
        const CellRuns runs = cell_runs(keys, n, g, keys[s0]);
        const CloudPoint p = sxyz[act ? s : s0];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          for (int t = runs.b[k]; t < runs.e[k]; ++t) {
            const CloudPoint q = sxyz[t];
            if (!q.core) continue;                         // wave-uniform
            const int r = root[q.index];
            if (border && pair_d2(p, q) <= r2) best = r < best ? r : best;
          }
        }
      }
      if (act) {
        const int i = order[s];
        labels[i] = core ? ids[root[i]] : (best == 0x7fffffff ? -1 : ids[best]);
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_cell_list(const int* __restrict__ heads, const int* __restrict__ vid, int n,
                                                   int* __restrict__ cell_first, int* __restrict__ ctrl) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= n) return;
  if (heads[s]) cell_first[vid[s]] = s;
  if (s == n - 1) {
    const int n_cells = vid[s] + heads[s];
    cell_first[n_cells] = n;
    ctrl[0] = n_cells;
#pragma unroll
    for (int k = 1; k < 8; ++k) ctrl[k] = 0;              // the passes' work counters
  }
}

// ---- voxel down-sampling -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_voxel_keys(const double* __restrict__ xyz, int n, CloudGrid g,
                                                    u64* __restrict__ keys, int* __restrict__ idx) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  long long v[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    // Open3D: ref_coord = (point - voxel_min_bound) / voxel_size; index = int(floor(ref_coord))
    long long c = (long long)floor((xyz[3 * (size_t)i + k] - g.lo[k]) / g.cell);
    v[k] = c < 0 ? 0 : (c >= g.dim[k] ? g.dim[k] - 1 : c);
  }
  keys[i] = (u64)((v[2] * g.dim[1] + v[1]) * g.dim[0] + v[0]);
  idx[i] = i;
}

__global__ __launch_bounds__(256) void k_voxel_heads(const u64* __restrict__ keys, int n, int* __restrict__ heads) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= n) return;
  heads[s] = (s == 0) || (keys[s] != keys[s - 1]);
}

__global__ __launch_bounds__(256) void k_voxel_average(const u64* __restrict__ keys, const int* __restrict__ order,
                                                       const int* __restrict__ heads, const int* __restrict__ vid,
                                                       const double* __restrict__ xyz, const double* __restrict__ rgb,
                                                       int n, double* __restrict__ xyz_out,
                                                       double* __restrict__ rgb_out, int* __restrict__ n_out) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= n) return;
  if (s == n - 1) *n_out = vid[s] + heads[s];
  if (!heads[s]) return;
  const u64 key = keys[s];
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, c0 = 0.0, c1 = 0.0, c2 = 0.0;
  int cnt = 0;
  // the sort is stable: the points of a voxel follow in input order, summed sequentially like AccumulatedPoint
  for (int t = s; t < n && keys[t] == key; ++t, ++cnt) {
    const size_t i = (size_t)order[t];
    a0 = a0 + xyz[3 * i + 0];
    a1 = a1 + xyz[3 * i + 1];
    a2 = a2 + xyz[3 * i + 2];
    if (rgb) {
      c0 = c0 + rgb[3 * i + 0];
      c1 = c1 + rgb[3 * i + 1];
      c2 = c2 + rgb[3 * i + 2];
    }
  }
  const size_t v = (size_t)vid[s];
  const double d = (double)cnt;
  xyz_out[3 * v + 0] = a0 / d;
  xyz_out[3 * v + 1] = a1 / d;
  xyz_out[3 * v + 2] = a2 / d;
  if (rgb) {
    rgb_out[3 * v + 0] = c0 / d;
    rgb_out[3 * v + 1] = c1 / d;
    rgb_out[3 * v + 2] = c2 / d;
  }
}

// ---- host side -----------------------------------------------------------------------------------------------------
static int make_grid(const double* lo, const double* hi, double cell, CloudGrid* g, const char* what) {
  FNR_CHECK_ARG(lo && hi, "%s: null bounds", what);
  FNR_CHECK_ARG(cell > 0.0 && cell == cell, "%s: cell size must be positive", what);
  for (int k = 0; k < 3; ++k) {
    FNR_CHECK_ARG(hi[k] >= lo[k] && (hi[k] - lo[k]) < 1e300, "%s: bounds must be finite with hi >= lo", what);
    const double d = floor((hi[k] - lo[k]) / cell);
    FNR_UNSUPPORTED(d < 2097151.0, "%s: extent / cell = %.3g cells on axis %d exceeds 2^21", what, d, k);
    g->lo[k] = lo[k];
    g->dim[k] = (long long)d + 1;
  }
  g->cell = cell;
  return FNR_OK;
}

static int key_bits(const CloudGrid& g) {
  u64 cells = (u64)g.dim[0] * (u64)g.dim[1] * (u64)g.dim[2];
  int bits = 1;
  while (bits < 64 && ((u64)1 << bits) < cells) ++bits;
  return bits;
}

static int sort_by_key(const CloudWs& w, int n, int bits, hipStream_t st) {
  size_t need = 0;
  FNR_HIP(rocprim::radix_sort_pairs(nullptr, need, w.keys_a, w.keys, w.idx_a, w.order, (size_t)n, 0u, (unsigned)bits,
                                    st));
  FNR_CHECK_ARG(need <= w.temp_bytes, "cloud: workspace too small for the key sort (%zu > %zu bytes)", need,
                w.temp_bytes);
  size_t have = w.temp_bytes;
  FNR_HIP(rocprim::radix_sort_pairs(w.temp, have, w.keys_a, w.keys, w.idx_a, w.order, (size_t)n, 0u, (unsigned)bits,
                                    st));
  return FNR_OK;
}

static int scan_flags(const CloudWs& w, const int* flags, int* out, int n, hipStream_t st) {
  size_t need = 0;
  FNR_HIP(rocprim::exclusive_scan(nullptr, need, flags, out, 0, (size_t)n, rocprim::plus<int>(), st));
  FNR_CHECK_ARG(need <= w.temp_bytes, "cloud: workspace too small for the scan");
  size_t have = w.temp_bytes;
  FNR_HIP(rocprim::exclusive_scan(w.temp, have, flags, out, 0, (size_t)n, rocprim::plus<int>(), st));
  return FNR_OK;
}

// neighbour grid for radius r: the cell edge is a hair above r so that rounding in (p - lo) / cell can never put two
// points closer than r more than one cell apart
static int build_search_grid(const double* xyz, int n, const double* lo, const double* hi, double radius,
                             const CloudWs& w, CloudGrid* g, hipStream_t st, const char* what) {
  int rc = make_grid(lo, hi, radius * (1.0 + 1.0 / 1048576.0), g, what);
  if (rc != FNR_OK) return rc;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_cloud_keys, dim3(blocks), dim3(256), 0, st, xyz, n, *g, w.keys_a, w.idx_a);
  FNR_LAUNCH_CHECK();
  rc = sort_by_key(w, n, key_bits(*g), st);
  if (rc != FNR_OK) return rc;
  hipLaunchKernelGGL(k_cloud_gather, dim3(blocks), dim3(256), 0, st, xyz, w.order, n, w.sxyz);
  FNR_LAUNCH_CHECK();
  // occupied cells: head flags of the sorted keys -> ranks -> first position per cell
  hipLaunchKernelGGL(k_voxel_heads, dim3(blocks), dim3(256), 0, st, w.keys, n, w.aux0);
  FNR_LAUNCH_CHECK();
  rc = scan_flags(w, w.aux0, w.aux3, n, st);
  if (rc != FNR_OK) return rc;
  hipLaunchKernelGGL(k_cell_list, dim3(blocks), dim3(256), 0, st, w.aux0, w.aux3, n, w.cell_first, w.ctrl);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

// persistent launch of the cell sweeps: enough waves to fill the chip, never more than one per cell
static unsigned sweep_blocks(int n) {
  const long long cap = (long long)device_cu_count() * 8;
  const long long need = ((long long)n + 3) / 4;
  return (unsigned)(need < cap ? need : cap);
}

}  // namespace fnr

using namespace fnr;

extern "C" size_t fnr_cloud_workspace_bytes(int64_t n_points) {
  return cloud_fixed_bytes(n_points) + cloud_temp_reserve(n_points);
}

extern "C" int fnr_cloud_bounds(const double* xyz, int64_t n, double* lo_hi, void* workspace, size_t workspace_bytes,
                                void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_cloud_bounds");
  FNR_CHECK_ARG(n >= 1 && n < 2147483647LL, "cloud_bounds: n out of range (an empty cloud has no bounds)");
  FNR_CHECK_ARG(xyz && lo_hi && workspace && workspace_bytes >= 64, "cloud_bounds: null argument / workspace < 64 B");
  hipStream_t st = as_stream(stream);
  u64* keys6 = static_cast<u64*>(workspace);
  FNR_HIP(hipMemsetAsync(keys6, 0xff, 3 * sizeof(u64), st));
  FNR_HIP(hipMemsetAsync(keys6 + 3, 0, 3 * sizeof(u64), st));
  long long blocks = (n + 255) / 256;
  const long long cap = (long long)device_cu_count();
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(k_cloud_bounds, dim3((unsigned)blocks), dim3(256), 0, st, xyz, (int)n, keys6);
  FNR_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_cloud_bounds_decode, dim3(1), dim3(64), 0, st, keys6, lo_hi);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

extern "C" int fnr_cloud_radius_count(const double* xyz, int64_t n, const double* lo, const double* hi, double radius,
                                      int inclusive, int32_t* counts, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_cloud_radius_count");
  FNR_CHECK_ARG(n >= 0 && n < 2147483647LL, "cloud_radius_count: n out of range");
  if (n == 0) return FNR_OK;
  FNR_CHECK_ARG(xyz && counts, "cloud_radius_count: null argument");
  FNR_CHECK_ARG(radius > 0.0, "cloud_radius_count: radius must be positive");
  CloudWs w;
  FNR_CHECK_ARG(carve_cloud(workspace, workspace_bytes, n, &w), "cloud_radius_count: workspace too small (need %zu)",
                fnr_cloud_workspace_bytes(n));
  hipStream_t st = as_stream(stream);
  FNR_PROF(OP_CLOUD, n);
  CloudGrid g;
  int rc = build_search_grid(xyz, (int)n, lo, hi, radius, w, &g, st, "cloud_radius_count");
  if (rc != FNR_OK) return rc;
  const unsigned sweep = sweep_blocks((int)n);
  const double r2 = radius * radius;
  if (inclusive)
    hipLaunchKernelGGL((k_cloud_count<true, false>), dim3(sweep), dim3(256), 0, st, w.keys, w.sxyz, w.order, (int)n, g,
                       r2, w.cell_first, w.ctrl, 0, counts, 0, nullptr, nullptr, nullptr);
  else
    hipLaunchKernelGGL((k_cloud_count<false, false>), dim3(sweep), dim3(256), 0, st, w.keys, w.sxyz, w.order, (int)n,
                       g, r2, w.cell_first, w.ctrl, 0, counts, 0, nullptr, nullptr, nullptr);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

extern "C" int fnr_cloud_dbscan(const double* xyz, int64_t n, const double* lo, const double* hi, double eps,
                                int32_t min_samples, int32_t* labels, int32_t* n_clusters, void* workspace,
                                size_t workspace_bytes, void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_cloud_dbscan");
  FNR_CHECK_ARG(n >= 0 && n < 2147483647LL, "cloud_dbscan: n out of range");
  hipStream_t st = as_stream(stream);
  if (n == 0) {
    if (n_clusters) FNR_HIP(hipMemsetAsync(n_clusters, 0, sizeof(int32_t), st));
    return FNR_OK;
  }
  FNR_CHECK_ARG(xyz && labels, "cloud_dbscan: null argument");
  FNR_CHECK_ARG(eps > 0.0 && min_samples >= 1, "cloud_dbscan: eps must be positive and min_samples >= 1");
  CloudWs w;
  FNR_CHECK_ARG(carve_cloud(workspace, workspace_bytes, n, &w), "cloud_dbscan: workspace too small (need %zu)",
                fnr_cloud_workspace_bytes(n));
  FNR_PROF(OP_CLOUD, n);
  CloudGrid g;
  int rc = build_search_grid(xyz, (int)n, lo, hi, eps, w, &g, st, "cloud_dbscan");
  if (rc != FNR_OK) return rc;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  const double r2 = eps * eps;
  int* core_sorted = w.aux0;
  int* parent = w.aux1;
  int* core_input = w.aux2;   // becomes the root flag
  int* ids = w.aux3;
  const unsigned sweep = sweep_blocks((int)n);
  hipLaunchKernelGGL((k_cloud_count<true, true>), dim3(sweep), dim3(256), 0, st, w.keys, w.sxyz, w.order, (int)n, g, r2,
                     w.cell_first, w.ctrl, 0, (int*)nullptr, (int)min_samples, core_sorted, core_input, parent);
  FNR_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_cloud_pack_core, dim3(blocks), dim3(256), 0, st, core_sorted, (int)n, w.sxyz);
  FNR_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_cloud_union, dim3(sweep), dim3(256), 0, st, w.keys, w.sxyz, w.order, (int)n, g, r2,
                     w.cell_first, w.ctrl, 1, core_sorted, parent);
  FNR_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_cloud_roots, dim3(blocks), dim3(256), 0, st, parent, core_input, (int)n);
  FNR_LAUNCH_CHECK();
  rc = scan_flags(w, core_input, ids, (int)n, st);
  if (rc != FNR_OK) return rc;
  hipLaunchKernelGGL(k_cloud_labels, dim3(sweep), dim3(256), 0, st, w.keys, w.sxyz, w.order, (int)n, g, r2,
                     w.cell_first, w.ctrl, 2, core_sorted, parent, core_input, ids, labels, n_clusters);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

extern "C" int fnr_cloud_voxel_down_sample(const double* xyz, const double* rgb, int64_t n, const double* min_bound,
                                           const double* max_bound, double voxel_size, double* xyz_out,
                                           double* rgb_out, int32_t* n_out, void* workspace, size_t workspace_bytes,
                                           void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_cloud_voxel_down_sample");
  FNR_CHECK_ARG(n >= 0 && n < 2147483647LL, "cloud_voxel_down_sample: n out of range");
  FNR_CHECK_ARG(n_out, "cloud_voxel_down_sample: null n_out");
  hipStream_t st = as_stream(stream);
  if (n == 0) {
    FNR_HIP(hipMemsetAsync(n_out, 0, sizeof(int32_t), st));
    return FNR_OK;
  }
  FNR_CHECK_ARG(xyz && xyz_out && min_bound && max_bound, "cloud_voxel_down_sample: null argument");
  FNR_CHECK_ARG((rgb == nullptr) == (rgb_out == nullptr), "cloud_voxel_down_sample: rgb and rgb_out go together");
  FNR_CHECK_ARG(voxel_size > 0.0, "cloud_voxel_down_sample: voxel_size must be positive");   // Open3D: voxel_size <= 0 is an error
  CloudWs w;
  FNR_CHECK_ARG(carve_cloud(workspace, workspace_bytes, n, &w),
                "cloud_voxel_down_sample: workspace too small (need %zu)", fnr_cloud_workspace_bytes(n));
  // Open3D: voxel_min_bound = min_bound - voxel_size / 2, voxel_max_bound = max_bound + voxel_size / 2
  double lo[3], hi[3];
  for (int k = 0; k < 3; ++k) {
    lo[k] = min_bound[k] - voxel_size * 0.5;
    hi[k] = max_bound[k] + voxel_size * 0.5;
  }
  CloudGrid g;
  int rc = make_grid(lo, hi, voxel_size, &g, "cloud_voxel_down_sample");
  if (rc != FNR_OK) return rc;
  FNR_PROF(OP_CLOUD, n);
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_voxel_keys, dim3(blocks), dim3(256), 0, st, xyz, (int)n, g, w.keys_a, w.idx_a);
  FNR_LAUNCH_CHECK();
  rc = sort_by_key(w, (int)n, key_bits(g), st);
  if (rc != FNR_OK) return rc;
  hipLaunchKernelGGL(k_voxel_heads, dim3(blocks), dim3(256), 0, st, w.keys, (int)n, w.aux0);
  FNR_LAUNCH_CHECK();
  rc = scan_flags(w, w.aux0, w.aux3, (int)n, st);
  if (rc != FNR_OK) return rc;
  hipLaunchKernelGGL(k_voxel_average, dim3(blocks), dim3(256), 0, st, w.keys, w.order, w.aux0, w.aux3, xyz, rgb, (int)n,
                     xyz_out, rgb_out, n_out);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}
