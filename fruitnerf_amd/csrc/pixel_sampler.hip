// pixel_sampler.hip — the caller side of a training step on the device (SURVEY §8f "next" row 1):
// Nerfstudio's PixelSampler + RayGenerator as driven by FruitDataManager.next_train
// (/root/reference/fruit_nerf/data/fruit_datamanager.py:188-197): uniform random (image, y, x) triples ->
// pinhole rays (pixel centre +0.5, camera looks along -z) + the rgb / fruit-mask targets of those pixels.
// One thread per ray; replaces ~15 small elementwise/index launches per step.  HBM-bound, 40 B written per ray.
#include "common.hpp"

namespace fnr {

struct ImageSetDev {
  int n_images, H, W;
  const uint8_t* images;  // [M,H,W,3]
  const uint8_t* masks;   // [M,H,W]
  const float* c2w;       // [M,3,4]
  float fx, fy, cx, cy;
};

__global__ __launch_bounds__(256) void k_sample_pixels(ImageSetDev s, const long long* __restrict__ train_ids,
                                                       int n_train, long long n_rays, const float* __restrict__ u,
                                                       const float* __restrict__ c2w_adjusted,
                                                       float* __restrict__ origins, float* __restrict__ directions,
                                                       int* __restrict__ cam_idx, float* __restrict__ image,
                                                       float* __restrict__ mask) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rays) return;
  // indices = floor(rand * [n_train, H, W]) (PixelSampler.sample_method), clamped like the torch mirror
  int k = (int)(u[3 * r] * (float)n_train);
  int y = (int)(u[3 * r + 1] * (float)s.H);
  int x = (int)(u[3 * r + 2] * (float)s.W);
  k = min(k, n_train - 1);
  y = min(y, s.H - 1);
  x = min(x, s.W - 1);
  const long long img = train_ids[k];
  const float dx = fdiv(fsub(fadd((float)x, 0.5f), s.cx), s.fx);
  const float dy = -fdiv(fsub(fadd((float)y, 0.5f), s.cy), s.fy);
  const float dz = -1.0f;
  // camera-pose optimisation: the slot's corrected camera (fnr_camera_adjust) replaces the dataset's
  const float* M = c2w_adjusted ? c2w_adjusted + (size_t)k * 12 : s.c2w + img * 12;
  float d[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) d[a] = fadd(fadd(fmul(M[4 * a], dx), fmul(M[4 * a + 1], dy)), fmul(M[4 * a + 2], dz));
  const float nrm = fmaxf(sqrtf(fadd(fadd(fmul(d[0], d[0]), fmul(d[1], d[1])), fmul(d[2], d[2]))), 1e-12f);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    directions[3 * r + a] = fdiv(d[a], nrm);
    origins[3 * r + a] = M[4 * a + 3];
  }
  cam_idx[r] = k;  // camera index = position in the training set (appearance-embedding row)
  const size_t pix = ((size_t)img * s.H + y) * s.W + x;
#pragma unroll
  for (int c = 0; c < 3; ++c) image[3 * r + c] = fdiv((float)s.images[3 * pix + c], 255.0f);
  mask[r] = (float)s.masks[pix];
}

}  // namespace fnr

using namespace fnr;

extern "C" int fnr_sample_pixels(const fnr_image_set* set, const int64_t* train_ids, int n_train, int64_t n_rays,
                                 const float* u, const float* c2w_adjusted, float* origins, float* directions,
                                 int32_t* camera_indices, float* image, float* fruit_mask, void* stream) {
  FNR_CHECK_ARG(set && train_ids && u && origins && directions && camera_indices && image && fruit_mask,
                "sample_pixels: null argument");
  FNR_CHECK_ARG(set->images && set->masks && set->c2w && set->n_images > 0 && set->H > 0 && set->W > 0 && n_train > 0,
                "sample_pixels: bad image set");
  if (n_rays == 0) return FNR_OK;
  ImageSetDev s{set->n_images, set->H, set->W, set->images, set->masks, set->c2w, set->fx, set->fy, set->cx, set->cy};
  hipLaunchKernelGGL(k_sample_pixels, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, as_stream(stream), s,
                     reinterpret_cast<const long long*>(train_ids), n_train, (long long)n_rays, u, c2w_adjusted, origins, directions,
                     camera_indices, image, fruit_mask);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}
