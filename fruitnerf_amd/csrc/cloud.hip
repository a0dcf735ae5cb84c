// cloud.hip — point-cloud front-end of the fruit counting stage on gfx950 (fp64, integer/index work, no MFMA):
//   radius-outlier neighbour counts   (Open3D remove_radius_outlier, clustering/clustering_base.py:141-143)
//   voxel down-sampling               (Open3D voxel_down_sample,     clustering/clustering_base.py:138-139)
//   DBSCAN labels                     (sklearn.cluster.DBSCAN,       clustering/clustering_base.py:199-200)
//
// One search structure serves the two neighbourhood queries: points are sorted by the key of a uniform grid whose
// cell edge is (just above) the search radius, with x fastest in the key, so the 27 neighbour cells of a point are
// 9 CONTIGUOUS runs of the sorted array, each found with two binary searches (the lanes of a wave mostly sit in the
// same cell, so the searches and the candidate loads are wave-wide broadcasts out of L1/L2).  A thread owns one point
// and tests every candidate in fp64 with the operation order of the CPU libraries (((dx*dx)+dy*dy)+dz*dz, no FMA), so
// the integer results (counts, masks, labels) are bit-exact.  The cost is the pair tests: fp64 VALU bound.
//
// DBSCAN = (1) inclusive neighbour counts -> core flags, (2) lock-free union-find over core-core pairs in ORIGINAL
// index space, always hooking the larger root under the smaller, so a cluster's root is its first core point in input
// order, (3) cluster numbers = rank of the roots (prefix sum in input order) = scikit-learn's numbering,
// (4) border points take the smallest cluster number among the core points within eps (the first cluster to reach
// them in scikit-learn's index-ordered expansion), everything else is noise (-1).
#include <cstring>

#include "common.hpp"

#include <rocprim/rocprim.hpp>

namespace fnr {

typedef unsigned long long u64;

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
  double* sxyz;   // [n][3] positions in sorted order
  int* aux0;      // per sorted position: core flag / head flag
  int* aux1;      // per input index: union-find parent
  int* aux2;      // per input index: core flag, then root flag
  int* aux3;      // scan output
  void* temp;
  size_t temp_bytes;
};

static inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

static size_t cloud_fixed_bytes(int64_t n) {
  const size_t m = (size_t)(n < 1 ? 1 : n);
  return 2 * align256(m * 8) + 2 * align256(m * 4) + align256(m * 24) + 4 * align256(m * 4);
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
  out->sxyz = reinterpret_cast<double*>(p); p += align256(m * 24);
  out->aux0 = reinterpret_cast<int*>(p);    p += align256(m * 4);
  out->aux1 = reinterpret_cast<int*>(p);    p += align256(m * 4);
  out->aux2 = reinterpret_cast<int*>(p);    p += align256(m * 4);
  out->aux3 = reinterpret_cast<int*>(p);    p += align256(m * 4);
  out->temp = p;
  out->temp_bytes = bytes - (size_t)(p - static_cast<char*>(ws));
  return true;
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
                                                      int n, double* __restrict__ sxyz) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= n) return;
  const size_t i = (size_t)order[s];
  sxyz[3 * (size_t)s + 0] = xyz[3 * i + 0];
  sxyz[3 * (size_t)s + 1] = xyz[3 * i + 1];
  sxyz[3 * (size_t)s + 2] = xyz[3 * i + 2];
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

// f(t, d2) for every sorted position t in the 3x3x3 cell neighbourhood of sorted position s (s itself included)
template <class F>
__device__ inline void for_each_candidate(const u64* __restrict__ keys, const double* __restrict__ sxyz, int n,
                                          const CloudGrid& g, int s, F&& f) {
  const u64 key = keys[s];
  const long long nx = g.dim[0], ny = g.dim[1], nz = g.dim[2];
  const long long cx = (long long)(key % (u64)nx);
  const long long cy = (long long)((key / (u64)nx) % (u64)ny);
  const long long cz = (long long)(key / (u64)(nx * ny));
  const double px = sxyz[3 * (size_t)s + 0], py = sxyz[3 * (size_t)s + 1], pz = sxyz[3 * (size_t)s + 2];
  const long long x0 = cx > 0 ? cx - 1 : 0, x1 = cx + 1 < nx ? cx + 1 : nx - 1;
  for (int dz = -1; dz <= 1; ++dz) {
    const long long zz = cz + dz;
    if (zz < 0 || zz >= nz) continue;
    for (int dy = -1; dy <= 1; ++dy) {
      const long long yy = cy + dy;
      if (yy < 0 || yy >= ny) continue;
      const u64 row = (u64)((zz * ny + yy) * nx);
      const int b = lower_bound_key(keys, 0, n, row + (u64)x0);
      const int e = upper_bound_key(keys, b, n, row + (u64)x1);
      for (int t = b; t < e; ++t) {
        const double dx = px - sxyz[3 * (size_t)t + 0];
        const double dy_ = py - sxyz[3 * (size_t)t + 1];
        const double dz_ = pz - sxyz[3 * (size_t)t + 2];
        const double d2 = (dx * dx + dy_ * dy_) + dz_ * dz_;
        f(t, d2);
      }
    }
  }
}

// counts[input index] = neighbours with d2 < r2 (INCL: <=), the point itself included.
// CORE: additionally core_sorted[s] / core_input[i] = count >= min_samples and parent[i] = i.
template <bool INCL, bool CORE>
__global__ __launch_bounds__(256) void k_cloud_count(const u64* __restrict__ keys, const double* __restrict__ sxyz,
                                                     const int* __restrict__ order, int n, CloudGrid g, double r2,
                                                     int* __restrict__ counts, int min_samples,
                                                     int* __restrict__ core_sorted, int* __restrict__ core_input,
                                                     int* __restrict__ parent) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= n) return;
  int c = 0;
  for_each_candidate(keys, sxyz, n, g, s, [&](int, double d2) { c += INCL ? (d2 <= r2) : (d2 < r2); });
  const int i = order[s];
  if (counts) counts[i] = c;
  if (CORE) {
    const int is_core = c >= min_samples;
    core_sorted[s] = is_core;
    core_input[i] = is_core;
    parent[i] = i;
  }
}

// ---- lock-free union-find (roots only ever hook under SMALLER indices) -------------------------------------------
__device__ inline int uf_load(int* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }

__device__ inline int uf_find(int* parent, int x) {
  while (true) {
    const int p = uf_load(&parent[x]);
    if (p == x) return x;
    const int gp = uf_load(&parent[p]);
    if (gp == p) return p;
    __atomic_store_n(&parent[x], gp, __ATOMIC_RELAXED);   // path halving: gp is an ancestor of x, x is not a root
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

__global__ __launch_bounds__(256) void k_cloud_union(const u64* __restrict__ keys, const double* __restrict__ sxyz,
                                                     const int* __restrict__ order, int n, CloudGrid g, double r2,
                                                     const int* __restrict__ core_sorted, int* parent) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= n || !core_sorted[s]) return;
  const int i = order[s];
  int mine = uf_find(parent, i);
  for_each_candidate(keys, sxyz, n, g, s, [&](int t, double d2) {
    if (d2 <= r2 && core_sorted[t]) {
      const int j = order[t];
      if (j < i) {                                        // every core-core pair is handled from its larger end
        const int rj = uf_find(parent, j);
        if (rj != mine) {
          uf_unite(parent, mine, rj);
          mine = uf_find(parent, i);
        }
      }
    }
  });
}

__global__ __launch_bounds__(256) void k_cloud_roots(int* parent, int* __restrict__ core_input_to_root_flag, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (core_input_to_root_flag[i]) {
    const int r = uf_find(parent, i);
    __atomic_store_n(&parent[i], r, __ATOMIC_RELAXED);    // i is a non-root or r == i: never races with a hook
    core_input_to_root_flag[i] = (r == i);
  } else {
    parent[i] = -1;
  }
}

__global__ __launch_bounds__(256) void k_cloud_labels(const u64* __restrict__ keys, const double* __restrict__ sxyz,
                                                      const int* __restrict__ order, int n, CloudGrid g, double r2,
                                                      const int* __restrict__ core_sorted,
                                                      const int* __restrict__ root, const int* __restrict__ root_flag,
                                                      const int* __restrict__ ids, int* __restrict__ labels,
                                                      int* __restrict__ n_clusters) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s == 0 && n_clusters) *n_clusters = ids[n - 1] + root_flag[n - 1];
  if (s >= n) return;
  const int i = order[s];
  if (core_sorted[s]) {
    labels[i] = ids[root[i]];
    return;
  }
  int best = 0x7fffffff;                                   // smallest root = smallest cluster number
  for_each_candidate(keys, sxyz, n, g, s, [&](int t, double d2) {
    if (d2 <= r2 && core_sorted[t]) {
      const int r = root[order[t]];
      best = r < best ? r : best;
    }
  });
  labels[i] = best == 0x7fffffff ? -1 : ids[best];
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
  return FNR_OK;
}

}  // namespace fnr

using namespace fnr;

extern "C" size_t fnr_cloud_workspace_bytes(int64_t n_points) {
  return cloud_fixed_bytes(n_points) + cloud_temp_reserve(n_points);
}

extern "C" int fnr_cloud_radius_count(const double* xyz, int64_t n, const double* lo, const double* hi, double radius,
                                      int inclusive, int32_t* counts, void* workspace, size_t workspace_bytes,
                                      void* stream) {
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
  const unsigned blocks = (unsigned)((n + 255) / 256);
  const double r2 = radius * radius;
  if (inclusive)
    hipLaunchKernelGGL((k_cloud_count<true, false>), dim3(blocks), dim3(256), 0, st, w.keys, w.sxyz, w.order, (int)n, g,
                       r2, counts, 0, nullptr, nullptr, nullptr);
  else
    hipLaunchKernelGGL((k_cloud_count<false, false>), dim3(blocks), dim3(256), 0, st, w.keys, w.sxyz, w.order, (int)n,
                       g, r2, counts, 0, nullptr, nullptr, nullptr);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

extern "C" int fnr_cloud_dbscan(const double* xyz, int64_t n, const double* lo, const double* hi, double eps,
                                int32_t min_samples, int32_t* labels, int32_t* n_clusters, void* workspace,
                                size_t workspace_bytes, void* stream) {
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
  hipLaunchKernelGGL((k_cloud_count<true, true>), dim3(blocks), dim3(256), 0, st, w.keys, w.sxyz, w.order, (int)n, g, r2,
                     (int*)nullptr, (int)min_samples, core_sorted, core_input, parent);
  FNR_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_cloud_union, dim3(blocks), dim3(256), 0, st, w.keys, w.sxyz, w.order, (int)n, g, r2, core_sorted,
                     parent);
  FNR_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_cloud_roots, dim3(blocks), dim3(256), 0, st, parent, core_input, (int)n);
  FNR_LAUNCH_CHECK();
  rc = scan_flags(w, core_input, ids, (int)n, st);
  if (rc != FNR_OK) return rc;
  hipLaunchKernelGGL(k_cloud_labels, dim3(blocks), dim3(256), 0, st, w.keys, w.sxyz, w.order, (int)n, g, r2,
                     core_sorted, parent, core_input, ids, labels, n_clusters);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

extern "C" int fnr_cloud_voxel_down_sample(const double* xyz, const double* rgb, int64_t n, const double* min_bound,
                                           const double* max_bound, double voxel_size, double* xyz_out,
                                           double* rgb_out, int32_t* n_out, void* workspace, size_t workspace_bytes,
                                           void* stream) {
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
