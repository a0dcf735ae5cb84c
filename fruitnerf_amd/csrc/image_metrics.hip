// image_metrics.hip — eval-image metrics on the device: what FruitModel.get_image_metrics_and_images
// (/root/reference/fruit_nerf/fruit_nerf.py:403-458) computes with torchmetrics on [H, W, C] images:
//   * PSNR(data_range = 1): squared error over every pixel and channel of clamp(rgb, 0, 1) vs the image;
//   * SSIM (torchmetrics structural_similarity_index_measure defaults: 11 x 11 gaussian window, sigma 1.5, k1 0.01,
//     k2 0.03, data_range None — the reference calls `self.ssim(image, rgb)` (:176,:424) without one, so torchmetrics
//     takes max(preds.max() - preds.min(), target.max() - target.min()) of the two images it is given: found on the
//     device by k_image_range, c1 / c2 = (k * range)^2 read by the SSIM kernel from the workspace): reflect-pad by 5, filter a, b, a^2, b^2, ab with the window, SSIM map, CROP the padded
//     border away again and take the mean — i.e. the mean over the (H - 10) x (W - 10) interior pixels of the
//     valid-window SSIM map; the padding never reaches a kept pixel.  Separable here (row pass into LDS, column pass);
//   * BinaryJaccardIndex(threshold 0.5) of sigmoid(semantics) against the mask, and of the reference's
//     `F.softmax(outputs["semantics"])` WITHOUT a dim (:451): on the [H, W, 1] map torch's implicit dim is 0, a softmax
//     over image ROWS (per column) — reproduced as is.
// All sums leave the device as doubles (per-workgroup partials, summed in a fixed order by one workgroup): deterministic.
// HBM-bound and tiny (an 800 x 800 image: ~15 MB read): latency is the launches, so four kernels in one entry point.
#include "common.hpp"
#include "sequencer.hpp"

namespace fnr {

constexpr int IM_WIN = 11, IM_PAD = 5;
constexpr int IM_TILE = 32;                 // SSIM outputs per workgroup: 32 x 32 of one channel
constexpr int IM_IN = IM_TILE + 2 * IM_PAD;  // 42 x 42 inputs
constexpr int IM_THREADS = 256;

struct Gauss11 {
  float w[IM_WIN];
};

__device__ __forceinline__ double block_sum(double v, double* s_red) {
  // 256 threads = 4 waves: wave sums by shuffles (two 32-bit halves), then lane 0 of each wave through LDS
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const long long bits = __double_as_longlong(v);
    const int lo = __shfl_xor((int)(bits & 0xffffffffll), d, 64), hi = __shfl_xor((int)(bits >> 32), d, 64);
    v += __longlong_as_double(((long long)hi << 32) | (unsigned long long)(unsigned)lo);
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_red[w];
  return t;
}

// one workgroup: a 32 x 32 tile of the interior SSIM map of one channel -> partial[block] = sum of the tile's SSIM values
__global__ __launch_bounds__(IM_THREADS) void k_image_ssim(int H, int W, const float* __restrict__ rgb,
                                                           const float* __restrict__ image, Gauss11 g,
                                                           const float* __restrict__ c12, double* __restrict__ partial) {
  __shared__ float s_a[IM_IN][IM_IN + 1], s_b[IM_IN][IM_IN + 1];
  __shared__ float s_h[5][IM_IN][IM_TILE + 1];
  __shared__ double s_red[IM_THREADS / 64];
  const int c = blockIdx.z;
  const float c1 = c12[0], c2 = c12[1];                                // (k1 * data_range)^2, (k2 * data_range)^2: k_image_range_finish
  const int oy0 = blockIdx.y * IM_TILE, ox0 = blockIdx.x * IM_TILE;   // interior coordinates: pixel (oy + 5, ox + 5)
  const int OH = H - 2 * IM_PAD, OW = W - 2 * IM_PAD;
  for (int i = threadIdx.x; i < IM_IN * IM_IN; i += IM_THREADS) {
    const int r = i / IM_IN, q = i - r * IM_IN;
    const int y = oy0 + r, x = ox0 + q;                                // window row / column in image coordinates
    float a = 0.0f, b = 0.0f;
    if (y < H && x < W) {
      const size_t p = ((size_t)y * W + x) * 3 + c;
      a = image[p];
      b = fminf(fmaxf(rgb[p], 0.0f), 1.0f);                            // torch.clamp(rgb, 0, 1) (:408)
    }
    s_a[r][q] = a;
    s_b[r][q] = b;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < IM_IN * IM_TILE; i += IM_THREADS) {    // row pass
    const int r = i / IM_TILE, q = i - r * IM_TILE;
    float ma = 0.f, mb = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
    for (int k = 0; k < IM_WIN; ++k) {
      const float a = s_a[r][q + k], b = s_b[r][q + k], w = g.w[k];
      ma = fmaf(w, a, ma);
      mb = fmaf(w, b, mb);
      aa = fmaf(w, a * a, aa);
      bb = fmaf(w, b * b, bb);
      ab = fmaf(w, a * b, ab);
    }
    s_h[0][r][q] = ma, s_h[1][r][q] = mb, s_h[2][r][q] = aa, s_h[3][r][q] = bb, s_h[4][r][q] = ab;
  }
  __syncthreads();
  double acc = 0.0;
  for (int i = threadIdx.x; i < IM_TILE * IM_TILE; i += IM_THREADS) {  // column pass + SSIM map
    const int r = i / IM_TILE, q = i - r * IM_TILE;
    if (oy0 + r >= OH || ox0 + q >= OW) continue;
    float v[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < IM_WIN; ++k) t = fmaf(g.w[k], s_h[j][r + k][q], t);
      v[j] = t;
    }
    const float mu_a = v[0], mu_b = v[1];
    const float s_aa = v[2] - mu_a * mu_a, s_bb = v[3] - mu_b * mu_b, s_ab = v[4] - mu_a * mu_b;
    const float upper = 2.0f * s_ab + c2, lower = s_aa + s_bb + c2;
    acc += (double)(((2.0f * mu_a * mu_b + c1) * upper) / ((mu_a * mu_a + mu_b * mu_b + c1) * lower));
  }
  const double t = block_sum(acc, s_red);
  if (threadIdx.x == 0) partial[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = t;
}

// data_range = None of torchmetrics: value ranges of the image and of clamp(rgb, 0, 1), partial[block] = {min a, max a, min b, max b}
__global__ __launch_bounds__(IM_THREADS) void k_image_range(long long n_val, const float* __restrict__ rgb,
                                                            const float* __restrict__ image, float* __restrict__ partial) {
  __shared__ float s_mm[IM_THREADS / 64][4];
  __shared__ int s_nan[IM_THREADS / 64];
  float lo_a = __builtin_inff(), hi_a = -__builtin_inff(), lo_b = __builtin_inff(), hi_b = -__builtin_inff();
  // torch's max() / min() and clamp() PROPAGATE NaN (preds.max() - preds.min() of a render with a NaN is NaN, and so is the
  // SSIM); fminf / fmaxf drop it — a NaN anywhere is carried as a flag and written as NaN bounds
  int bad = 0;
  for (long long i = (long long)blockIdx.x * IM_THREADS + threadIdx.x; i < n_val; i += (long long)gridDim.x * IM_THREADS) {
    const float a = image[i], raw = rgb[i], b = fminf(fmaxf(raw, 0.0f), 1.0f);
    bad |= (a != a) | (raw != raw);
    lo_a = fminf(lo_a, a), hi_a = fmaxf(hi_a, a), lo_b = fminf(lo_b, b), hi_b = fmaxf(hi_b, b);
  }
  bad = __any(bad);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    lo_a = fminf(lo_a, __shfl_xor(lo_a, d, 64)), hi_a = fmaxf(hi_a, __shfl_xor(hi_a, d, 64));
    lo_b = fminf(lo_b, __shfl_xor(lo_b, d, 64)), hi_b = fmaxf(hi_b, __shfl_xor(hi_b, d, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    float* m = s_mm[threadIdx.x >> 6];
    m[0] = lo_a, m[1] = hi_a, m[2] = lo_b, m[3] = hi_b;
    s_nan[threadIdx.x >> 6] = bad;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < IM_THREADS / 64; ++w) {
      lo_a = fminf(lo_a, s_mm[w][0]), hi_a = fmaxf(hi_a, s_mm[w][1]), lo_b = fminf(lo_b, s_mm[w][2]), hi_b = fmaxf(hi_b, s_mm[w][3]);
      bad |= s_nan[w];
    }
    float* o = partial + 4 * blockIdx.x;
    const float nan = __builtin_nanf("");
    o[0] = bad ? nan : lo_a, o[1] = bad ? nan : hi_a, o[2] = bad ? nan : lo_b, o[3] = bad ? nan : hi_b;
  }
}

// one wave: the partial ranges -> c12 = {(0.01 * range)^2, (0.03 * range)^2}, c12[2] = range
__global__ __launch_bounds__(64) void k_image_range_finish(const float* __restrict__ partial, int n, float* __restrict__ c12) {
  float lo_a = __builtin_inff(), hi_a = -__builtin_inff(), lo_b = __builtin_inff(), hi_b = -__builtin_inff();
  int bad = 0;
  for (int i = threadIdx.x; i < n; i += 64) {
    bad |= partial[4 * i] != partial[4 * i];       // (a block that saw a NaN wrote NaN bounds)
    lo_a = fminf(lo_a, partial[4 * i]), hi_a = fmaxf(hi_a, partial[4 * i + 1]), lo_b = fminf(lo_b, partial[4 * i + 2]),
    hi_b = fmaxf(hi_b, partial[4 * i + 3]);
  }
  bad = __any(bad);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    lo_a = fminf(lo_a, __shfl_xor(lo_a, d, 64)), hi_a = fmaxf(hi_a, __shfl_xor(hi_a, d, 64));
    lo_b = fminf(lo_b, __shfl_xor(lo_b, d, 64)), hi_b = fmaxf(hi_b, __shfl_xor(hi_b, d, 64));
  }
  if (threadIdx.x == 0) {
    const float range = bad ? __builtin_nanf("") : fmaxf(hi_a - lo_a, hi_b - lo_b);
    const float k1r = 0.01f * range, k2r = 0.03f * range;
    c12[0] = k1r * k1r, c12[1] = k2r * k2r, c12[2] = range;
  }
}

// per pixel: squared error over the three channels; sigmoid(semantics) > 0.5 against mask > 0.5
// partial[block] = {sse, intersection, union}
__global__ __launch_bounds__(IM_THREADS) void k_image_pixels(long long n_pix, const float* __restrict__ rgb,
                                                             const float* __restrict__ image,
                                                             const float* __restrict__ semantics,
                                                             const float* __restrict__ mask, double* __restrict__ partial) {
  __shared__ double s_red[IM_THREADS / 64];
  double sse = 0.0, inter = 0.0, uni = 0.0;
  for (long long p = (long long)blockIdx.x * IM_THREADS + threadIdx.x; p < n_pix; p += (long long)gridDim.x * IM_THREADS) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float d = image[3 * p + c] - fminf(fmaxf(rgb[3 * p + c], 0.0f), 1.0f);
      sse += (double)(d * d);
    }
    if (semantics) {
      const bool pred = 1.0f / (1.0f + expf(-semantics[p])) > 0.5f, tgt = mask[p] > 0.5f;
      inter += (pred && tgt) ? 1.0 : 0.0;
      uni += (pred || tgt) ? 1.0 : 0.0;
    }
  }
  const double a = block_sum(sse, s_red), b = block_sum(inter, s_red), c = block_sum(uni, s_red);
  if (threadIdx.x == 0) partial[3 * blockIdx.x] = a, partial[3 * blockIdx.x + 1] = b, partial[3 * blockIdx.x + 2] = c;
}

// thread = image column: softmax over the rows of that column (torch's implicit dim 0 for a 3-D tensor), > 0.5 against the
// mask; partial[block] = {intersection, union}
__global__ __launch_bounds__(IM_THREADS) void k_image_row_softmax_iou(int H, int W, const float* __restrict__ semantics,
                                                                      const float* __restrict__ mask,
                                                                      double* __restrict__ partial) {
  __shared__ double s_red[IM_THREADS / 64];
  const int x = blockIdx.x * IM_THREADS + threadIdx.x;
  double inter = 0.0, uni = 0.0;
  if (x < W) {
    float m = -__builtin_inff();
    for (int y = 0; y < H; ++y) m = fmaxf(m, semantics[(size_t)y * W + x]);
    float z = 0.0f;
    for (int y = 0; y < H; ++y) z += expf(semantics[(size_t)y * W + x] - m);
    for (int y = 0; y < H; ++y) {
      const bool pred = expf(semantics[(size_t)y * W + x] - m) / z > 0.5f, tgt = mask[(size_t)y * W + x] > 0.5f;
      inter += (pred && tgt) ? 1.0 : 0.0;
      uni += (pred || tgt) ? 1.0 : 0.0;
    }
  }
  const double a = block_sum(inter, s_red), b = block_sum(uni, s_red);
  if (threadIdx.x == 0) partial[2 * blockIdx.x] = a, partial[2 * blockIdx.x + 1] = b;
}

// one workgroup: the partials in a fixed order -> out[8] = {sse, ssim sum, inter sigmoid, union sigmoid, inter row-softmax,
// union row-softmax, SSIM values summed, values under sse}
__global__ __launch_bounds__(IM_THREADS) void k_image_finish(const double* __restrict__ p_ssim, int n_ssim,
                                                             const double* __restrict__ p_pix, int n_pix_blocks,
                                                             const double* __restrict__ p_col, int n_col_blocks,
                                                             double ssim_count, double sse_count, double* __restrict__ out) {
  __shared__ double s_red[IM_THREADS / 64];
  double v[6] = {0, 0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < n_ssim; i += IM_THREADS) v[1] += p_ssim[i];
  for (int i = threadIdx.x; i < n_pix_blocks; i += IM_THREADS) v[0] += p_pix[3 * i], v[2] += p_pix[3 * i + 1], v[3] += p_pix[3 * i + 2];
  for (int i = threadIdx.x; i < n_col_blocks; i += IM_THREADS) v[4] += p_col[2 * i], v[5] += p_col[2 * i + 1];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const double t = block_sum(v[j], s_red);
    if (threadIdx.x == 0) out[j] = t;
  }
  if (threadIdx.x == 0) out[6] = ssim_count, out[7] = sse_count;
}

static int pixel_blocks(long long n_pix) {
  long long b = (n_pix + IM_THREADS - 1) / IM_THREADS;
  return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}

}  // namespace fnr

using namespace fnr;

extern "C" size_t fnr_image_metrics_workspace_bytes(int H, int W) {
  if (H <= 0 || W <= 0) return 0;
  const int OH = H - 2 * IM_PAD, OW = W - 2 * IM_PAD;
  const long long tiles = (OH > 0 && OW > 0) ? 3ll * ((OH + IM_TILE - 1) / IM_TILE) * ((OW + IM_TILE - 1) / IM_TILE) : 0;
  // + the value-range partials (4 floats per pixel block) and {c1, c2, range, -} of the SSIM
  return (size_t)(tiles + 3ll * pixel_blocks((long long)H * W) + 2ll * ((W + IM_THREADS - 1) / IM_THREADS) +
                  2ll * pixel_blocks((long long)H * W) + 2) * sizeof(double);
}

extern "C" int fnr_image_metrics(int H, int W, const float* rgb, const float* image, const float* semantics,
                                 const float* mask, const float* gauss11, double* out, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_image_metrics");
  FNR_CHECK_ARG(rgb && image && gauss11 && out && workspace, "image_metrics: null argument");
  FNR_CHECK_ARG((semantics == nullptr) == (mask == nullptr), "image_metrics: semantics and mask come together");
  FNR_CHECK_ARG(H > 2 * IM_PAD && W > 2 * IM_PAD, "image_metrics: image %d x %d smaller than the 11 x 11 SSIM window", H, W);
  FNR_CHECK_ARG(workspace_bytes >= fnr_image_metrics_workspace_bytes(H, W), "image_metrics: workspace too small");
  Gauss11 g;
  for (int k = 0; k < IM_WIN; ++k) g.w[k] = gauss11[k];   // host array (11 floats: the torchmetrics window, float32)
  const int OH = H - 2 * IM_PAD, OW = W - 2 * IM_PAD;
  const dim3 tiles((unsigned)((OW + IM_TILE - 1) / IM_TILE), (unsigned)((OH + IM_TILE - 1) / IM_TILE), 3);
  const int n_ssim = (int)(tiles.x * tiles.y * tiles.z), n_pb = pixel_blocks((long long)H * W);
  const int n_cb = semantics ? (W + IM_THREADS - 1) / IM_THREADS : 0;
  double* p_ssim = reinterpret_cast<double*>(workspace);
  double* p_pix = p_ssim + n_ssim;
  double* p_col = p_pix + 3 * n_pb;
  float* p_range = reinterpret_cast<float*>(p_col + 2 * ((W + IM_THREADS - 1) / IM_THREADS));   // [n_pb][4]
  float* c12 = p_range + 4 * n_pb;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(k_image_range, dim3((unsigned)n_pb), dim3(IM_THREADS), 0, st, 3ll * H * W, rgb, image, p_range);
  FNR_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_image_range_finish, dim3(1), dim3(64), 0, st, p_range, n_pb, c12);
  FNR_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_image_ssim, tiles, dim3(IM_THREADS), 0, st, H, W, rgb, image, g, c12, p_ssim);
  FNR_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_image_pixels, dim3((unsigned)n_pb), dim3(IM_THREADS), 0, st, (long long)H * W, rgb, image, semantics,
                     mask, p_pix);
  FNR_LAUNCH_CHECK();
  if (semantics) {
    hipLaunchKernelGGL(k_image_row_softmax_iou, dim3((unsigned)n_cb), dim3(IM_THREADS), 0, st, H, W, semantics, mask, p_col);
    FNR_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_image_finish, dim3(1), dim3(IM_THREADS), 0, st, p_ssim, n_ssim, p_pix, n_pb, p_col, n_cb,
                     3.0 * OH * OW, 3.0 * H * W, out);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}
