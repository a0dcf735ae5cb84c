// pixel_sampler.hip — the caller side of a training step on the device (SURVEY §8f "next" row 1):
// Nerfstudio's PixelSampler + RayGenerator as driven by FruitDataManager.next_train
// (/root/reference/fruit_nerf/data/fruit_datamanager.py:188-197): uniform random (image, y, x) triples ->
// pinhole rays (pixel centre +0.5, camera looks along -z) + the rgb / fruit-mask targets of those pixels.
// One thread per ray; replaces ~15 small elementwise/index launches per step.  HBM-bound, 40 B written per ray.
#include "camera_math.hpp"
#include "sampler_math.hpp"
#include "sequencer.hpp"

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

// ---- the start of a training step in ONE launch (fnr_train_prologue) ---------------------------------------------------
// Five launches of 4-7 us each before (random numbers for the pixels, camera adjust, pixel sampling, random numbers for
// the sampler's jitter, level-0 spaced sampling), all latency: the random numbers come from a counter-based generator
// (Philox4x32-10, Salmon et al. 2011; counter = (ray, word group, step offset), key = seed), so every role of the launch
// derives the numbers it needs itself.
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0, c[1] = lo1, c[2] = n2, c[3] = lo0;
    k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
  }
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }  // [0, 1), 24 bits
// the step's random numbers of ray r: words 0-2 = (image, y, x) of the pixel, words 3-5 = the samplers' single jitters
__device__ __forceinline__ void prologue_randoms(unsigned long long seed, unsigned long long offset, long long r, int group,
                                                 float (&out)[4]) {
  uint32_t c[4] = {(uint32_t)r, (uint32_t)((unsigned long long)r >> 32) ^ ((uint32_t)group << 24), (uint32_t)offset,
                   (uint32_t)(offset >> 32)};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = u01(c[i]);
}

struct PrologueArgs {
  ImageSetDev set;
  const long long* train_ids;
  int n_train;
  long long n_rays;
  unsigned long long seed, offset;
  const float* pose;        // [n_train, 6] SO3xR3 tangents, or NULL (cameras as they are)
  float* c2w_adjusted;      // [n_train, 3, 4] out (with pose)
  float *u, *jitter;        // [R, 3], [n_jitter, R] out
  int n_jitter;             // <= 5
  float *origins, *directions;
  int* cam_idx;
  float *image, *mask;
  float near, far;
  int kind, S0;
  const float* base_bins;   // [S0 + 1]
  float *spacing0, *euclid0;   // [R, S0 + 1] out
  int nb_rays;              // workgroups of the per-ray role
};

__global__ __launch_bounds__(256) void k_train_prologue(PrologueArgs a) {
  if ((int)blockIdx.x < a.nb_rays) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    // corrected cameras for the pose-gradient kernel: camera k by thread k of the launch (same arithmetic as per ray)
    if (a.pose && r < a.n_train) {
      float adj[12];
      adjusted_camera(a.set.c2w + a.train_ids[r] * 12, a.pose + 6 * r, adj);
#pragma unroll
      for (int i = 0; i < 12; ++i) a.c2w_adjusted[12 * r + i] = adj[i];
    }
    if (r >= a.n_rays) return;
    float w0[4], w1[4];
    prologue_randoms(a.seed, a.offset, r, 0, w0);
    prologue_randoms(a.seed, a.offset, r, 1, w1);
    const float rnd[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
#pragma unroll
    for (int i = 0; i < 3; ++i) a.u[3 * r + i] = rnd[i];
    for (int i = 0; i < a.n_jitter; ++i) a.jitter[(size_t)i * a.n_rays + r] = rnd[3 + i];
    // pixel sampling + ray generation: k_sample_pixels, with the ray's corrected camera formed in place
    int k = (int)(rnd[0] * (float)a.n_train);
    int y = (int)(rnd[1] * (float)a.set.H);
    int x = (int)(rnd[2] * (float)a.set.W);
    k = min(k, a.n_train - 1);
    y = min(y, a.set.H - 1);
    x = min(x, a.set.W - 1);
    const long long img = a.train_ids[k];
    const float dx = fdiv(fsub(fadd((float)x, 0.5f), a.set.cx), a.set.fx);
    const float dy = -fdiv(fsub(fadd((float)y, 0.5f), a.set.cy), a.set.fy);
    const float dz = -1.0f;
    float M[12];
    if (a.pose) {
      adjusted_camera(a.set.c2w + img * 12, a.pose + 6 * k, M);
    } else {
#pragma unroll
      for (int i = 0; i < 12; ++i) M[i] = a.set.c2w[img * 12 + i];
    }
    float d[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) d[q] = fadd(fadd(fmul(M[4 * q], dx), fmul(M[4 * q + 1], dy)), fmul(M[4 * q + 2], dz));
    const float nrm = fmaxf(sqrtf(fadd(fadd(fmul(d[0], d[0]), fmul(d[1], d[1])), fmul(d[2], d[2]))), 1e-12f);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      a.directions[3 * r + q] = fdiv(d[q], nrm);
      a.origins[3 * r + q] = M[4 * q + 3];
    }
    a.cam_idx[r] = k;
    const size_t pix = ((size_t)img * a.set.H + y) * a.set.W + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) a.image[3 * r + c] = fdiv((float)a.set.images[3 * pix + c], 255.0f);
    a.mask[r] = (float)a.set.masks[pix];
    return;
  }
  // level-0 bins of the proposal sampler (k_sample_spaced with one jitter per ray = word 3 of the ray's numbers)
  const long long idx = ((long long)blockIdx.x - a.nb_rays) * 256 + threadIdx.x;
  const long long total = a.n_rays * (long long)(a.S0 + 1);
  if (idx >= total) return;
  const long long r = idx / (a.S0 + 1);
  const int j = (int)(idx - r * (a.S0 + 1));
  float w0[4];
  prologue_randoms(a.seed, a.offset, r, 0, w0);
  const float b = spaced_bin_edge(a.base_bins, a.S0, j, true, w0[3]);
  const float s_near = spacing_fn(a.kind, a.near), s_far = spacing_fn(a.kind, a.far);
  a.spacing0[idx] = b;
  a.euclid0[idx] = spacing_to_euclid(a.kind, b, s_near, s_far);
}

}  // namespace fnr

using namespace fnr;

extern "C" int fnr_sample_pixels(const fnr_image_set* set, const int64_t* train_ids, int n_train, int64_t n_rays,
                                 const float* u, const float* c2w_adjusted, float* origins, float* directions,
                                 int32_t* camera_indices, float* image, float* fruit_mask, void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_sample_pixels");
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

extern "C" int fnr_train_prologue(const fnr_image_set* set, const int64_t* train_ids, int n_train, int64_t n_rays,
                                  uint64_t seed, uint64_t offset, const float* pose_adjustment, float* c2w_adjusted,
                                  float* u, float* jitter, int n_jitter, float* origins, float* directions,
                                  int32_t* camera_indices, float* image, float* fruit_mask, float near_plane,
                                  float far_plane, int spacing_kind, int S0, const float* base_bins, float* spacing0,
                                  float* euclid0, void* stream) {
  if (seq::recording() && set) {
    const fnr_image_set set_ = *set;
    seq::push("fnr_train_prologue", [=](const fnr_step_scalars* sc) {
      return fnr_train_prologue(&set_, train_ids, n_train, n_rays, seed, sc ? sc->prologue_offset : offset, pose_adjustment,
                                c2w_adjusted, u, jitter, n_jitter, origins, directions, camera_indices, image, fruit_mask,
                                near_plane, far_plane, spacing_kind, S0, base_bins, spacing0, euclid0, stream);
    });
  }
  FNR_CHECK_ARG(set && train_ids && u && jitter && origins && directions && camera_indices && image && fruit_mask &&
                    base_bins && spacing0 && euclid0,
                "train_prologue: null argument");
  FNR_CHECK_ARG(set->images && set->masks && set->c2w && set->n_images > 0 && set->H > 0 && set->W > 0 && n_train > 0,
                "train_prologue: bad image set");
  FNR_CHECK_ARG(n_jitter >= 1 && n_jitter <= FNR_TRAIN_PROLOGUE_MAX_JITTER && S0 >= 1, "train_prologue: n_jitter %d (1..%d), S0 %d",
                n_jitter, FNR_TRAIN_PROLOGUE_MAX_JITTER, S0);
  FNR_CHECK_ARG((pose_adjustment == nullptr) == (c2w_adjusted == nullptr), "train_prologue: pose and c2w_adjusted go together");
  FNR_CHECK_ARG(n_rays >= n_train || !pose_adjustment, "train_prologue: fewer rays than cameras");
  if (n_rays == 0) return FNR_OK;
  PrologueArgs a;
  a.set = ImageSetDev{set->n_images, set->H, set->W, set->images, set->masks, set->c2w, set->fx, set->fy, set->cx, set->cy};
  a.train_ids = reinterpret_cast<const long long*>(train_ids);
  a.n_train = n_train, a.n_rays = n_rays, a.seed = seed, a.offset = offset;
  a.pose = pose_adjustment, a.c2w_adjusted = c2w_adjusted, a.u = u, a.jitter = jitter, a.n_jitter = n_jitter;
  a.origins = origins, a.directions = directions, a.cam_idx = camera_indices, a.image = image, a.mask = fruit_mask;
  a.near = near_plane, a.far = far_plane, a.kind = spacing_kind, a.S0 = S0, a.base_bins = base_bins;
  a.spacing0 = spacing0, a.euclid0 = euclid0;
  a.nb_rays = (int)((n_rays + 255) / 256);
  const long long nb_bins = (n_rays * (long long)(S0 + 1) + 255) / 256;
  FNR_PROF(OP_SAMPLE_SPACED, n_rays * (long long)(S0 + 1));
  hipLaunchKernelGGL(k_train_prologue, dim3((unsigned)(a.nb_rays + nb_bins)), dim3(256), 0, as_stream(stream), a);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}
