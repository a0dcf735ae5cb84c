/*
 * fruitnerf_hip.h — C ABI of the MI355X-native (gfx950) FruitNeRF ray-marching hot path.
 *
 * The reference (meyerls/FruitNeRF) has no FFI: its hot path is the Nerfstudio Python plugin API
 * (fruit_nerf/fruit_field.py FruitField, fruit_nerf/fruit_nerf.py FruitModel) and the arithmetic
 * runs in nerfstudio==0.3.2 / tiny-cuda-nn.  This header is the C boundary our Python mirror of
 * that API (fruitnerf_amd/) binds with ctypes; every entry point cites the reference interface
 * (file:line relative to /root/reference) whose arithmetic it replaces.  INTEGRATION.md shows the
 * reference-side binding.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless a name ends in _host;
 *   - the caller owns every buffer (parameters, gradients, workspace, outputs);
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*, NULL = default stream);
 *     no entry point synchronises unless documented;
 *   - return value 0 = ok, negative = error (fnr_last_error() gives the message); no exceptions
 *     cross the ABI; nothing falls back to the CPU.
 *   - tensors are dense row-major fp32 unless stated; N = n_rays * S samples, sample n = ray*S + k.
 */
#ifndef FRUITNERF_HIP_H
#define FRUITNERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FNR_ABI_VERSION 13
#define FNR_MAX_LEVELS 16
#define FNR_MAX_SEM_LAYERS 4
/* floats in a loss accumulator buffer: per-ray partials are spread over 32 accumulators that sit in 32 different
 * 128-byte lines (atomics to one L2 line serialise at ~12 ns each on MI355X: 4096 rays on one line cost ~35 us);
 * the caller zeroes the buffer and sums all FNR_LOSS_SLOTS floats (see fnr_interlevel_fwd) */
#define FNR_LOSS_SLOTS 1024

enum {
  FNR_OK = 0,
  FNR_ERR_INVALID = -1,     /* bad argument */
  FNR_ERR_UNSUPPORTED = -2, /* configuration not built into this library */
  FNR_ERR_HIP = -3,         /* HIP runtime error (message in fnr_last_error) */
};

/* Multiresolution hash grid, nerfstudio-0.3.2 torch layout: HashEncoding (constructed at
 * fruit_field.py:124-131 and fruit_nerf.py:111-127).  Every level owns T = 2^log2_hashmap_size rows
 * of 2 floats; `scalings` are the integers floor(min_res * growth**level) computed on the HOST in
 * float32 exactly as HashEncoding.__init__ does. */
typedef struct fnr_grid {
  int32_t n_levels;
  int32_t log2_hashmap_size;
  int32_t scalings[FNR_MAX_LEVELS];
  float* table; /* [n_levels << log2_hashmap_size, 2] */
} fnr_grid;

/* RayBundle (nerfstudio.cameras.rays; built at data/fruit_datamanager.py:188-197 and
 * components/ray_generators.py:52-64), after the collider (fruit_nerf.py:161,382-383). */
typedef struct fnr_rays {
  int64_t n_rays;
  const float* origins;          /* [R,3] */
  const float* directions;       /* [R,3] */
  const float* nears;            /* [R]   */
  const float* fars;             /* [R]   */
  const int32_t* camera_indices; /* [R] or NULL */
} fnr_rays;

/* World position -> unit cube (fruit_field.py:168-179).
 * mode 0: SceneContraction(order=inf) then (x+2)/4     (training / rendering)
 * mode 1: SceneBox.get_normalized_positions(x, aabb)   (export: spatial_distortion=None, fruit_nerf.py:183) */
typedef struct fnr_warp {
  int32_t mode;
  float aabb[6]; /* min xyz, max xyz */
} fnr_warp;

/* HashMLPDensityField (fruit_nerf.py:104-129): hash grid -> Linear(2L,H) ReLU Linear(H,1) -> trunc_exp. */
typedef struct fnr_prop_net {
  fnr_grid grid;
  int32_t hidden_dim; /* 16 */
  float* w0;          /* [H, 2L] */
  float* b0;          /* [H] */
  float* w1;          /* [1, H] */
  float* b1;          /* [1] */
} fnr_prop_net;

/* FruitField (fruit_field.py:43-301).  Linear weights are nn.Linear layout [out, in].
 * A second instance of this struct with pointers into the gradient arena describes the gradients.
 * Two shapes are built (everything else returns FNR_ERR_UNSUPPORTED):
 *   `fruit_nerf`                       geo_feat_dim 15, num_layers_semantic 2, hidden_dim_semantics 64
 *   `fruit_nerf_big` / `fruit_nerf_huge` geo_feat_dim 30, num_layers_semantic 3, hidden_dim_semantics 128
 * (fruit_nerf_config.py:27-160; of the widths those configs set only these three reach FruitField,
 * fruit_nerf.py:88-103), both with base / colour width 64, semantic_out_dim 64, appearance 32, 16 hash levels. */
typedef struct fnr_field_net {
  fnr_grid grid;
  int32_t geo_feat_dim;         /* 15 | 30 */
  int32_t hidden_dim;           /* 64 (base MLP width)  */
  int32_t hidden_dim_color;     /* 64 */
  int32_t hidden_dim_semantics; /* 64 | 128 */
  int32_t num_layers_semantic;  /* 2 | 3 */
  int32_t semantic_out_dim;     /* 64 (hidden_dim_transient, fruit_field.py:150) */
  int32_t appearance_dim;       /* 32 */
  int32_t n_images;
  float* base_w0; float* base_b0;           /* [64, 2L], [64] */
  float* base_w1; float* base_b1;           /* [1+geo, 64], [1+geo] */
  float* sem_w[FNR_MAX_SEM_LAYERS];         /* mlp_semantics layers: [64,15],[64,64] | [128,30],[128,128],[64,128] */
  float* sem_b[FNR_MAX_SEM_LAYERS];
  float* head_w; float* head_b;             /* SemanticFieldHead: [1, 64], [1] (components/field_heads.py:29-40) */
  float* col_w[3]; float* col_b[3];         /* mlp_head: [64,16+geo+32],[64,64],[3,64] */
  float* embedding;                         /* embedding_appearance [n_images, 32] */
  int32_t mlp_mode;                         /* FNR_MLP_*: arithmetic of the MLP GEMMs (fnr_field_mlp_fwd / _bwd) */
} fnr_field_net;

/* fnr_field_net.mlp_mode.  The reference trains with mixed_precision=True (fp16 autocast on CUDA,
 * fruit_nerf_config.py:33,69) and runs pure fp32 on the CPU (nerfstudio disables autocast there); parity is judged
 * against the fp32 CPU path.
 *   FNR_MLP_FP32    v_mfma_f32_16x16x4_f32: exact fp32 FMA chains (default; the parity path, 157 TFLOP/s roofline)
 *   FNR_MLP_BF16    v_mfma_f32_16x16x32_bf16 on bf16-rounded operands, fp32 accumulate: throughput mode of BASELINE
 *                   config 2 (2.5 PFLOP/s roofline); outputs differ from fp32 by ~1e-2 relative — NOT parity grade
 *   FNR_MLP_BF16X3  the same instruction on an exact three-way bf16 split of every fp32 operand (6 piece products
 *                   forward, 3 backward): fp32-grade results (~1e-6) at 16/6 resp. 16/3 of the fp32-MFMA rate
 * The bf16 modes are built for the `fruit_nerf` shape; others return FNR_ERR_UNSUPPORTED. */
#define FNR_MLP_FP32 0
#define FNR_MLP_BF16 1
#define FNR_MLP_BF16X3 3

/* ---- library / device ----------------------------------------------------------------------- */
int fnr_abi_version(void);
const char* fnr_last_error(void);
/* Queries the current HIP device; fails loudly when no gfx950 device is present. */
int fnr_device_check(int* cu_count_out, char* name_out, int name_len);

/* Optional HIP-event timing of the entry points (bench.py's roofline leg).  While enabled, every entry
 * point whose op id is set in op_mask brackets what it enqueues with two events recorded on the caller's
 * stream.  fnr_profile_collect synchronises those events and returns up to `capacity` records
 * (op id, units = samples/rays/params the call processed, milliseconds); enable(…) clears old records.
 * Op ids: 0 sample_spaced, 1 weights_pdf, 2 prop_density_fwd, 3 hash_encode_fwd, 4 hash_encode_lattice,
 * 5 field_mlp_fwd, 6 composite_fwd, 7 losses_fwd, 8 interlevel_fwd, 9 distortion, 10 composite_bwd,
 * 11 weights_bwd, 12 field_mlp_bwd, 13 hash_encode_bwd, 14 prop_density_bwd, 15 adam_step, 16 export_compact. */
int fnr_profile_enable(int on, uint64_t op_mask);
/* paused != 0: entry points stop recording events but the records collected so far are kept (bench.py times every
 * tenth step of its timed window: each event pair costs the GPU a ~3 us bubble). */
int fnr_profile_pause(int paused);
int64_t fnr_profile_collect(int32_t* ops_host, int64_t* units_host, float* ms_host, int64_t capacity);
/* Diagnostics of the binned scatter (fnr_hash_encode_bwd*, fnr_prop_density_bwd*): records that did not fit their bin's
 * queue (capacity = 3 x the level's mean) and went to the gradient table through global atomics instead, summed over all
 * calls since the last reset (synchronises the device).  0 in a healthy configuration; such records make the
 * gradient's summation order, hence its last bits, depend on timing.  No counterpart in the reference (torch autograd
 * scatters with atomics throughout). */
int fnr_debug_scatter_overflows(uint64_t* count_host, int reset);
/* records_host[2] = records the accumulate kernels have summed since the last reset: [0] into the field's table
 * (fnr_hash_encode_bwd*), [1] into proposal tables (fnr_prop_density_bwd*) — i.e. the 8 L contributions per sample after
 * run pre-summing.  Each record is 10 bytes written by the emit kernel and read back by the accumulate kernel: the
 * design overhead bench.py reports as roofline.record_queue_bytes.  Synchronises the device.  No counterpart in the
 * reference. */
int fnr_debug_scatter_records(uint64_t* records_host, int reset);

/* ---- caller side: pixel sampling + ray generation --------------------------------------------- */
/* The on-device image batch of a datamanager: uint8 images [M,H,W,3], uint8 fruit masks [M,H,W] (1 = fruit),
 * pinhole cameras (camera-to-world [M,3,4]; fx, fy, cx, cy shared). */
typedef struct fnr_image_set {
  int32_t n_images, H, W;
  const uint8_t* images;
  const uint8_t* masks;
  const float* c2w;
  float fx, fy, cx, cy;
} fnr_image_set;

/* FruitDataManager.next_train (data/fruit_datamanager.py:188-197): PixelSampler (uniform (image, y, x) from
 * u [R,3] in [0,1)) + RayGenerator (pixel centre +0.5, -z forward, unit directions).  train_ids [n_train] maps
 * training slot -> dataset image; camera_indices receives the slot (the appearance-embedding row).
 * c2w_adjusted (optional) [n_train,3,4]: pose-corrected cameras from fnr_camera_adjust, indexed by slot.
 * Outputs: origins/directions [R,3], camera_indices [R], image [R,3] in [0,1], fruit_mask [R] in {0,1}. */
int fnr_sample_pixels(const fnr_image_set* set, const int64_t* train_ids, int n_train, int64_t n_rays, const float* u,
                      const float* c2w_adjusted, float* origins, float* directions, int32_t* camera_indices,
                      float* image, float* fruit_mask, void* stream);

#define FNR_TRAIN_PROLOGUE_MAX_JITTER 5   /* rows of `jitter` one fnr_train_prologue launch draws (level 0 + 4 PDF levels) */
/* The start of a training step in ONE launch (single jitter per ray, the Nerfacto default): draws the step's random
 * numbers itself — Philox4x32-10, counter = (ray, word group, offset), key = seed; the caller advances `offset` by one per
 * step — and performs fnr_camera_adjust (pose_adjustment / c2w_adjusted both set, or both NULL), fnr_sample_pixels and
 * fnr_sample_spaced(level 0, near / far as the collider sets them, one jitter per ray) on them.
 * Outputs besides those of the three entry points: u [R,3] (what fnr_camera_pose_grad needs again) and
 * jitter [n_jitter, R]: row 0 is the jitter level 0 was sampled with, rows 1.. are for the PDF samplers of the following
 * levels.  Given u and jitter, every output is bit-identical to the separate entry points (tests/test_gpu_properties.py). */
int fnr_train_prologue(const fnr_image_set* set, const int64_t* train_ids, int n_train, int64_t n_rays, uint64_t seed,
                       uint64_t offset, const float* pose_adjustment, float* c2w_adjusted, float* u, float* jitter,
                       int n_jitter, float* origins, float* directions, int32_t* camera_indices, float* image,
                       float* fruit_mask, float near_plane, float far_plane, int spacing_kind, int S0,
                       const float* base_bins, float* spacing0, float* euclid0, void* stream);

/* nerfstudio CameraOptimizer(mode="SO3xR3") (fruit_nerf_config.py:39-43): c2w_adjusted[k] =
 * pose_utils.multiply(c2w[train_ids[k]], exp_map_SO3xR3(pose_adjustment[k])), pose_adjustment [n_train,6] =
 * (translation, so3 log-rotation) per training camera. */
int fnr_camera_adjust(const float* c2w, const int64_t* train_ids, int n_train, const float* pose_adjustment,
                      float* c2w_adjusted, void* stream);
/* Backward of fnr_camera_adjust + the ray generation of fnr_sample_pixels: pose_grad [n_train,6] += d(loss)/d(pose)
 * from d_origins / d_directions [R,3] (fnr_position_grad_reduce) of the rays drawn with `u`. */
int fnr_camera_pose_grad(const fnr_image_set* set, const int64_t* train_ids, int n_train, int64_t n_rays,
                         const float* u, const int32_t* camera_indices, const float* pose_adjustment,
                         const float* c2w_adjusted, const float* d_origins, const float* d_directions,
                         float* pose_grad, void* stream);

/* ---- samplers ------------------------------------------------------------------------------- */
/* SpacedSampler.generate_ray_samples (components/ray_samplers.py:54-104; nerfstudio
 * UniformLinDispPiecewiseSampler for the proposal level 0, fruit_nerf.py:151-158).
 * spacing_kind 0: identity (UniformSamplerWithNoise, export); 1: lin/disparity piecewise.
 * base_bins: [S+1] = torch.linspace(0, 1, S+1) evaluated on the HOST (ray_samplers.py:76), so the bins
 *            carry ATen's exact linspace rounding.
 * t_rand: stratified jitter (training mode) or NULL (eval: the bins are base_bins).  t_rand_per_bin = 0: [R], one
 *         number per ray (single_jitter=True: the proposal sampler's level 0); != 0: [R,S+1], one per bin edge
 *         (single_jitter=False, components/ray_samplers.py:79-83 — what the reference's exporter runs, because
 *         FruitModel.setup_inference builds its UniformSamplerWithNoise after eval_setup(), fruit_nerf.py:179-183).
 * Outputs: spacing bins [R,S+1] and euclidean bins [R,S+1]. */
int fnr_sample_spaced(const fnr_rays* rays, int spacing_kind, int S, const float* base_bins, const float* t_rand,
                      int t_rand_per_bin, float* spacing_bins, float* euclid_bins, void* stream);

/* RaySamples.get_weights on the previous level + PDFSampler (nerfstudio; driven from
 * ProposalNetworkSampler, fruit_nerf.py:151-158,318): weights of the S_prev samples, annealed
 * (w^anneal), histogram padding 0.01, inverse-CDF resampling to S_new+1 bins.
 * u_base: [S_new+1] = torch.linspace(0, 1 - 1/(S_new+1), S_new+1) evaluated on the HOST.
 * rand: [R] single-jitter numbers (training) or NULL (eval: centred u).
 * S_new = 0: weights (and median depth) only.
 * median_depth (optional, [R]): DepthRenderer("median") of the S_prev samples (fruit_nerf.py:299-300). */
int fnr_weights_pdf(const fnr_rays* rays, int spacing_kind, int S_prev, int S_new, const float* density,
                    const float* spacing_prev, const float* euclid_prev, float anneal, const float* u_base,
                    const float* rand, float* weights, float* median_depth, float* spacing_new, float* euclid_new, void* stream);

/* ---- proposal networks ---------------------------------------------------------------------- */
/* HashMLPDensityField.density_fn at the S bin midpoints of every ray (fruit_nerf.py:111-129).
 * feat_save: optional [L][N][2] encoded features kept for fnr_prop_density_bwd. */
int fnr_prop_density_fwd(const fnr_prop_net* net, const fnr_warp* warp, const fnr_rays* rays, const float* euclid_bins,
                         int S, float* density, float* feat_save, void* stream);

/* ---- main field ----------------------------------------------------------------------------- */
/* HashEncoding forward of FruitField.get_density (fruit_field.py:168-186) at the bin midpoints.
 * feats: [L][N][2] (level-major), selector: [N] bytes (the 0<x<1 mask, fruit_field.py:178).
 * jacobian (optional) [L][3][N][2]: d feats / d(unit-cube position), kept for fnr_position_grad_from_jacobian when
 * the rays carry gradients (camera-pose optimisation). */
int fnr_hash_encode_fwd(const fnr_grid* grid, const fnr_warp* warp, const fnr_rays* rays, const float* euclid_bins,
                        int S, float* feats, uint8_t* selector, float* jacobian, void* stream);

/* Same on the orthographic export lattice (data/fruit_datamanager.py:71-121,157-172;
 * components/ray_generators.py:46-66; components/ray_samplers.py:76-94): sample n of the batch is
 * ray (ray_begin + n / n_z), depth index n % n_z, at (xs[ray / n_y], ys[ray % n_y], zs[n % n_z]). */
typedef struct fnr_lattice {
  int32_t n_x, n_y, n_z;
  const float* xs; /* [n_x] torch.linspace values */
  const float* ys; /* [n_y] */
  const float* zs; /* [n_z] sample-midpoint z coordinates */
} fnr_lattice;
int fnr_hash_encode_lattice(const fnr_grid* grid, const fnr_warp* warp, const fnr_lattice* lat, int64_t ray_begin,
                            int64_t n_rays, float* feats, uint8_t* selector, void* stream);

/* mlp_base_mlp + trunc_exp, mlp_semantics + SemanticFieldHead, SHEncoding + mlp_head
 * (fruit_field.py:187-193, 195-232, 234-281) on MFMA.
 * mean_embedding: [32] -> eval/export path (fruit_field.py:217-219,253-256); NULL -> training path,
 * per-ray Embedding[camera_indices] (fruit_field.py:251).
 * h_save [N, fnr_field_h_dim(net)]: the base MLP's raw output (density logit | geo | zero padding to a multiple of
 * 16: 16 floats for geo 15, 32 for geo 30), kept for fnr_field_mlp_bwd.  Optional for the `fruit_nerf` shape; REQUIRED
 * for the `fruit_nerf_big` shape, whose 176 KB of weights do not fit one CU's LDS: it runs as two launches (base +
 * colour, then semantic) that hand h over through this buffer.
 * ray_bias_save (optional) [n_rays,64]: the per-ray part of mlp_head's first layer, kept for fnr_field_mlp_bwd.
 * Outputs per sample: density [N], rgb [N,3], logit [N]; geo_out (optional) [N, geo_feat_dim] = the
 * density embedding `base_mlp_out` returned by get_density (fruit_field.py:187-193).
 * workspace: >= fnr_field_mlp_fwd_workspace_bytes(n_rays) bytes of device scratch: the MFMA fragment image of
 * the weights (packed once per call) and the per-ray part of mlp_head's first layer (SH + appearance
 * embedding are constant along a ray: [n_rays, 64]; with ray_bias_save the workspace only needs
 * fnr_field_mlp_fwd_workspace_bytes(0)).  The image sits at the start of the workspace. */
size_t fnr_field_mlp_fwd_workspace_bytes(int64_t n_rays);
/* Row length of h_save / h_saved for this field shape (16 or 32); -1 if the shape is not built. */
int fnr_field_h_dim(const fnr_field_net* net);
int fnr_field_mlp_fwd(const fnr_field_net* net, const fnr_rays* rays, int S, const float* feats,
                      const uint8_t* selector, const float* mean_embedding, float* density, float* rgb, float* logit,
                      float* geo_out, float* h_save, float* ray_bias_save, void* workspace, size_t workspace_bytes,
                      void* stream);

/* embedding_appearance.mean(dim=0) (fruit_field.py:219,256): out [appearance_dim]. */
int fnr_embedding_mean(const float* embedding, int n_images, int dim, float* out, void* stream);

/* ---- compositing ---------------------------------------------------------------------------- */
/* RaySamples.get_weights + RGBRenderer("last_sample") + AccumulationRenderer + DepthRenderer("median")
 * + SemanticRenderer (fruit_nerf.py:325-348).  training=0 additionally applies nan_to_num / clamp. */
int fnr_composite_fwd(const fnr_rays* rays, int S, const float* euclid_bins, const float* density, const float* rgb,
                      const float* logit, int training, float* weights, float* out_rgb, float* out_accumulation,
                      float* out_depth, float* out_semantics, int64_t* out_label, void* stream);

/* ---- training: losses, backward, optimiser --------------------------------------------------- */
/* get_loss_dict's rgb_loss = MSELoss(image, rgb) and semantics_loss = w * BCEWithLogitsLoss(semantics,
 * fruit_mask) (fruit_nerf.py:171-172,362-366).  losses[0..1] are overwritten with the two scalars and losses[2]
 * with PSNR(rgb, image) = -10 log10(rgb_loss) (get_metrics_dict, fruit_nerf.py:398) — `losses` holds 3 floats;
 * d_rgb [R,3] / d_semantics [R] receive dloss/drgb and dloss/dsemantics (unit upstream). */
int fnr_losses_fwd(int64_t n_rays, const float* rgb, const float* image, const float* semantics,
                   const float* fruit_mask, float semantic_loss_weight, float* losses, float* d_rgb,
                   float* d_semantics, void* stream);

/* nerfstudio interlevel_loss (fruit_nerf.py:368-370) of ONE proposal level against the final level:
 * sum(loss[0..FNR_LOSS_SLOTS)) += mult * mean(clip(w - outer(c, cp, wp), 0)^2 / (w + 1e-7)) — the caller adds the
 * slots up (same-address global atomics serialise at ~12 ns each on MI355X); d_weights_p [R,S_p] = d/d wp. */
int fnr_interlevel_fwd(int64_t n_rays, int S_f, const float* spacing_f, const float* weights_f, int S_p,
                       const float* spacing_p, const float* weights_p, float mult, float* loss, float* d_weights_p,
                       void* stream);

/* nerfstudio distortion_loss on the final level — a metric only (fruit_nerf.py:400):
 * sum(out[0..FNR_LOSS_SLOTS)) += value. */
int fnr_distortion(int64_t n_rays, int S, const float* spacing, const float* weights, float* out, void* stream);

/* Backward of fnr_composite_fwd (training): g_rgb [R,3], g_semantics [R] -> per-sample d_density [N],
 * d_rgb [N,3], d_logit [N].  Semantic compositing weights are detached (fruit_nerf.py:343-345). */
int fnr_composite_bwd(const fnr_rays* rays, int S, const float* euclid_bins, const float* density, const float* rgb,
                      const float* weights, const float* g_rgb, const float* g_semantics, float* d_density,
                      float* d_rgb, float* d_logit, void* stream);

/* fnr_composite_bwd with the per-ray loss gradients formed in the kernel (ABI 11): g_rgb = 2 (out_rgb - image) / (3 R),
 * g_semantics = semantic_loss_weight (sigmoid(out_semantics) - mask) / R — MSELoss / BCEWithLogitsLoss(mean) backward
 * with unit upstream (fruit_nerf.py:359-367), the expressions fnr_train_losses evaluates, so d_density / d_rgb / d_logit
 * are bit-identical to fnr_train_losses + fnr_composite_bwd.  It makes the backward independent of the losses launch: a
 * training step runs it straight behind the forward and sums the loss values on another stream.
 * out_rgb [R,3], out_semantics [R]: what fnr_composite_fwd returned; image [R,3], mask [R]: the batch. */
int fnr_composite_bwd_targets(const fnr_rays* rays, int S, const float* euclid_bins, const float* density,
                              const float* rgb, const float* weights, const float* out_rgb, const float* image,
                              const float* out_semantics, const float* mask, float semantic_loss_weight,
                              float* d_density, float* d_rgb, float* d_logit, void* stream);
/* fnr_composite_fwd (training = 1) + fnr_composite_bwd_targets as one launch (ABI 13): the same outputs and the same gradients,
 * bit for bit, one kernel boundary less between the field's forward and backward pass of a training step
 * (fruit_nerf.py:325-348 + the autograd of its renderers and of get_loss_dict's rgb / semantic terms, :365-391). */
int fnr_composite_fwd_bwd_targets(const fnr_rays* rays, int S, const float* euclid_bins, const float* density,
                                  const float* rgb, const float* logit, const float* image, const float* mask,
                                  float semantic_loss_weight, float* weights, float* out_rgb, float* out_accumulation,
                                  float* out_depth, float* out_semantics, int64_t* out_label, float* d_density,
                                  float* d_rgb, float* d_logit, void* stream);

/* Backward of RaySamples.get_weights for a proposal level: d_weights [R,S] (x *upstream if non-NULL, a
 * device scalar) -> d_density [R,S]. */
int fnr_weights_bwd(int64_t n_rays, int S, const float* euclid_bins, const float* density, const float* weights,
                    const float* d_weights, const float* upstream, float* d_density, void* stream);

/* Backward of fnr_field_mlp_fwd (training path): accumulates (+=) the Linear weight/bias/embedding gradients
 * into `grads` and writes dL/dfeatures d_feats [L][N][2] for fnr_hash_encode_bwd.
 * workspace: >= fnr_field_mlp_bwd_workspace_bytes(n_rays, S) bytes. */
size_t fnr_field_mlp_bwd_workspace_bytes(int64_t n_rays, int S);
int fnr_field_mlp_bwd(const fnr_field_net* net, const fnr_field_net* grads, const fnr_rays* rays, int S,
                      const float* feats, const float* h_saved, const float* ray_bias_saved /* optional */,
                      const float* packed_saved /* optional: the start of fnr_field_mlp_fwd's workspace (its fragment
                      image), valid if the weights have not changed since */,
                      const uint8_t* selector, const float* d_density,
                      const float* d_rgb, const float* d_logit, float* d_feats, void* workspace, size_t workspace_bytes,
                      void* stream);

/* fnr_field_mlp_bwd + the input gradient of the hash grid for rays that carry gradients (camera-pose optimisation,
 * fruit_nerf_config.py:39-43; the reference's positions are forced requires_grad, fruit_field.py:180-182):
 * jacobian [L][3][N][2] = what fnr_hash_encode_fwd saved (d feats / d unit-cube position), d_position [N][4] receives
 * dL/d(unit-cube position) of every sample (xyz, w = 0) — the contraction of d_feats with the Jacobian, formed by the
 * base-branch kernel from the dL/dfeats it holds in registers (bf16-pipe modes; one extra launch in fp32 mode).
 * fnr_position_grad_reduce(n_levels = 1, partial = d_position) finishes dL/d(origins, directions). */
int fnr_field_mlp_bwd_rays(const fnr_field_net* net, const fnr_field_net* grads, const fnr_rays* rays, int S,
                           const float* feats, const float* h_saved, const float* ray_bias_saved /* optional */,
                           const float* packed_saved /* optional */, const uint8_t* selector, const float* d_density,
                           const float* d_rgb, const float* d_logit, float* d_feats, const float* jacobian,
                           float* d_position, void* workspace, size_t workspace_bytes, void* stream);

/* fnr_field_mlp_bwd (+ optionally the input gradient, as fnr_field_mlp_bwd_rays: jacobian and d_position both NULL or
 * both set) with the optimiser step of every parameter this backward produces the gradient of — the Linear weights and
 * biases of mlp_base_mlp / mlp_semantics / field_head_semantics / mlp_head and the appearance embedding, i.e. the
 * "fields" group without the hash table (fruit_nerf_config.py:51-56) — fused in (single-process training).  The thread
 * that owns a gradient entry (k_reduce_dw, k_embedding_grad: fixed-order sums, single writer) applies torch.optim.Adam /
 * RAdam to that parameter and leaves the gradient entry ZERO.  grad_arena: base of the gradient buffer `grads` points
 * into; weight_adam->params / exp_avg / exp_avg_sq: bases of the buffers that parallel it element for element (the
 * caller's flat arenas).  Bit-identical to fnr_field_mlp_bwd followed by fnr_adam_step / fnr_radam_step(zero_grad = 1)
 * on those spans (tests/test_gpu_training_parity.py). */
int fnr_field_mlp_bwd_adam(const fnr_field_net* net, const fnr_field_net* grads, const fnr_rays* rays, int S,
                           const float* feats, const float* h_saved, const float* ray_bias_saved /* optional */,
                           const float* packed_saved /* optional */, const uint8_t* selector, const float* d_density,
                           const float* d_rgb, const float* d_logit, float* d_feats, const float* jacobian /* optional */,
                           float* d_position /* optional */, const struct fnr_table_adam* weight_adam,
                           const float* grad_arena, void* workspace, size_t workspace_bytes, void* stream);


/* Backward of fnr_hash_encode_fwd: adds (+=) the trilinear scatter of d_feats [L][N][2] into
 * grid_grad->table for the levels [level_begin, level_begin + level_count) (all levels: 0, n_levels; data-parallel
 * training calls it per group of levels so that a group's rows can be all-reduced while the next group is being
 * scattered).  Binned through per-bin queues in `workspace` (>= fnr_hash_scatter_workspace_bytes(n_samples,
 * level_count, log2_hashmap_size)) because global fp32 atomics top out at ~21 G/s on MI355X (hash_scatter.hip).
 * workspace_clean: 0 = the library zeroes the queue counters first (any buffer); 1 = the caller passes a buffer whose
 * last use was a completed call of the same entry point with the same sizes — the kernels leave the counters zeroed,
 * which saves one memset launch per call. */
size_t fnr_hash_scatter_workspace_bytes(int64_t n_samples, int n_levels, int log2_hashmap_size);
int fnr_hash_encode_bwd(const fnr_grid* grid_grad, const fnr_warp* warp, const fnr_rays* rays,
                        const float* euclid_bins, int S, const float* d_feats, int level_begin, int level_count,
                        void* workspace, size_t workspace_bytes, int workspace_clean, void* stream);

/* fnr_hash_encode_bwd over ALL levels with the optimiser step of the table ("fields" group, fruit_nerf_config.py:51-56)
 * fused in (single-process training: with
 * several ranks the gradient has to exist for the all-reduce).  The workgroup that owns a bin of rows holds their summed
 * gradient in LDS and applies torch.optim.Adam (algorithm 0) / RAdam (1) to those rows — parameters, exp_avg and
 * exp_avg_sq are the TABLE's slices [n_levels << log2_hashmap_size, 2] of the caller's arenas, `step` is the
 * optimiser's step count of this update (>= 1), grad_scale / weight_decay as in fnr_adam_step.  The gradient table
 * (grid_grad->table) is left as it was (zero): it is only read where an overflowed queue spilled into it.  Results are
 * bit-identical to fnr_hash_encode_bwd followed by fnr_adam_step / fnr_radam_step on the table's span.
 * touched (optional, table entry points only; weight_decay must be 0): persistent bitmap owned by the caller, one bit per
 * PAIR of table rows (4 floats; bit i of word i / 32, index relative to `params`), zeroed when the moments are zero: "some
 * row of this pair has ever received a gradient".  A pair whose bit is clear and which receives no gradient now has
 * exp_avg = exp_avg_sq = 0, so torch's update of it is exactly zero (m = v = 0 stay, p - lr / bc1 * (0 / eps) = p) and
 * its 24 B per parameter are neither read nor written ("sparse-touch skipping"): every level is hashed,
 * so the coarse levels only ever use (res + 1)^3 of their 2^T rows — 27 % of the `fruit_nerf` table never moves.  The
 * kernel sets the bits of the pairs it touches.  NULL: every row is swept.  Results are bit-identical either way. */
typedef struct fnr_table_adam {
  int32_t algorithm;
  float lr, beta1, beta2, eps;
  int32_t slot; /* step programs (below): 1..FNR_PROGRAM_ADAM_SLOTS = on replay `lr` and `step` come from
                   fnr_step_scalars.adam[slot - 1]; 0 = the recorded values.  Ignored outside a replay. */
  int64_t step;
  float grad_scale, weight_decay;
  float* params;
  float* exp_avg;
  float* exp_avg_sq;
  uint32_t* touched;
} fnr_table_adam;
int fnr_hash_encode_bwd_adam(const fnr_grid* grid_grad, const fnr_warp* warp, const fnr_rays* rays,
                             const float* euclid_bins, int S, const float* d_feats, void* workspace,
                             size_t workspace_bytes, int workspace_clean, const fnr_table_adam* adam, void* stream);

/* fnr_camera_pose_grad with the pose table's optimiser step (camera_optimizer group, fruit_nerf_config.py:39-43) fused in
 * (single-process training): adam->params is
 * pose_adjustment [n_train,6] (read for the gradient, then updated), exp_avg / exp_avg_sq its moments; pose_grad is
 * added to the new gradient and left ZERO.  Bit-identical to fnr_camera_pose_grad followed by fnr_adam_step /
 * fnr_radam_step(zero_grad = 1) over the 6 n_train floats. */
int fnr_camera_pose_grad_adam(const fnr_image_set* set, const int64_t* train_ids, int n_train, int64_t n_rays,
                              const float* u, const int32_t* camera_indices, const float* c2w_adjusted,
                              const float* d_origins, const float* d_directions, float* pose_grad,
                              const fnr_table_adam* adam, void* stream);

/* Backward of fnr_prop_density_fwd: d_density [R,S] -> += into grads (table, w0, b0, w1, b1).
 * d_position (optional) [N,4]: gradient w.r.t. each sample's unit-cube position (xyz, w = 0) for
 * fnr_position_grad_reduce(n_levels = 1) — only needed when the rays carry gradients (camera-pose optimiser).
 * workspace >= fnr_prop_density_bwd_workspace_bytes(N, L, log2_hashmap_size). */
size_t fnr_prop_density_bwd_workspace_bytes(int64_t n_samples, int n_levels, int log2_hashmap_size);
int fnr_prop_density_bwd(const fnr_prop_net* net, const fnr_prop_net* grads, const fnr_warp* warp,
                         const fnr_rays* rays, const float* euclid_bins, int S, const float* feat_save,
                         const float* d_density, float* d_position, void* workspace, size_t workspace_bytes,
                         int workspace_clean, void* stream);

/* fnr_prop_density_bwd with the optimiser step of THIS network's parameters fused in (single-process training; a network
 * that serves one proposal level — with use_same_proposal_network the levels' gradients have to be summed first):
 * table_adam = the hash table's slices of the caller's arenas (as in fnr_hash_encode_bwd_adam: the scatter's accumulate
 * kernel steps the rows it owns, the gradient table stays zero), weight_adam / grad_arena = the arenas' bases (as in
 * fnr_field_mlp_bwd_adam: k_prop_reduce steps w0 / b0 / w1 / b1 and leaves their gradient entries zero).  Bit-identical to
 * fnr_prop_density_bwd followed by fnr_adam_step / fnr_radam_step(zero_grad = 1) on the network's spans. */
int fnr_prop_density_bwd_adam(const fnr_prop_net* net, const fnr_prop_net* grads, const fnr_warp* warp,
                              const fnr_rays* rays, const float* euclid_bins, int S, const float* feat_save,
                              const float* d_density, float* d_position, const fnr_table_adam* table_adam,
                              const fnr_table_adam* weight_adam, const float* grad_arena, void* workspace,
                              size_t workspace_bytes, int workspace_clean, void* stream);

/* Both proposal levels of a training step through ONE entry point (arrays of two: network, gradients, warp, bins, S,
 * saved features, d_density, d_position, workspace, ...; the same rays): each level's MLP backward, weight reduction and
 * scatter emit as in fnr_prop_density_bwd, then ONE accumulate launch over both levels' bins (each level has 160
 * workgroups of 64 KiB LDS on 256 CUs: side by side they cost the longer of the two).  The two levels need their own
 * network, gradients and workspace.  table_adam [2] + weight_adam + grad_arena: all NULL, or all set for the fused
 * optimiser steps of fnr_prop_density_bwd_adam.  Results are those of the two separate calls, bit for bit. */
int fnr_prop_density_bwd_pair(const fnr_prop_net* const* nets, const fnr_prop_net* const* grads,
                              const fnr_warp* const* warps, const fnr_rays* rays, const float* const* euclid_bins,
                              const int* S, const float* const* feat_save, const float* const* d_density,
                              float* const* d_position, const fnr_table_adam* const* table_adam,
                              const fnr_table_adam* weight_adam, const float* grad_arena, void* const* workspace,
                              const size_t* workspace_bytes, const int* workspace_clean, void* stream);
/* fnr_prop_density_bwd_pair with its launches in two groups: both levels' MLP backward + weight reduction first — after
 * them d_position[0] and [1] are final, and position_ready_event (a hipEvent_t created by the caller; optional) is
 * recorded on `stream` — then both levels' emit launches and the joint accumulate (~210 us for `fruit_nerf`).  A caller
 * with a second stream finishes the ray gradients and takes the camera optimiser's step next to that scatter instead
 * of behind it.  Same launches on the same inputs: results are those of fnr_prop_density_bwd_pair, bit for bit. */
int fnr_prop_density_bwd_pair_split(const fnr_prop_net* const* nets, const fnr_prop_net* const* grads,
                                    const fnr_warp* const* warps, const fnr_rays* rays, const float* const* euclid_bins,
                                    const int* S, const float* const* feat_save, const float* const* d_density,
                                    float* const* d_position, const fnr_table_adam* const* table_adam,
                                    const fnr_table_adam* weight_adam, const float* grad_arena, void* const* workspace,
                                    const size_t* workspace_bytes, const int* workspace_clean, void* stream,
                                    void* position_ready_event);

/* All of get_loss_dict / get_metrics_dict (fruit_nerf.py:359-372, 396-401: rgb_loss, semantics_loss, interlevel_loss; psnr, distortion) for one training batch in ONE launch:
 * fnr_losses_fwd + fnr_interlevel_fwd for
 * each of the n_levels (<= FNR_MAX_PROPOSAL_LEVELS) proposal levels against the final level + (want_distortion)
 * fnr_distortion, and the sum of the accumulator slots.  losses [5] = rgb_loss, semantics_loss, psnr,
 * interlevel_loss, distortion (0 when not wanted); d_rgb [R,3], d_semantics [R] (both NULL: not written — a caller whose
 * composite backward forms them itself, fnr_composite_bwd_targets), d_weights_p[l] [R,S_p[l]] as the single calls give them.  accum: FNR_TRAIN_LOSSES_ACCUM_FLOATS floats (loss slots + completion counters), zeroed by
 * the caller before its FIRST use; every completed call leaves it zeroed again (the last workgroup cleans up), so a caller
 * that keeps the buffer launches no fill per step — after a FAILED call the caller must zero it again.  The slots hold
 * 64-bit fixed point (2^-34 resolution): a contribution that is not finite or beyond its row's bound (4096 for the per-ray
 * interlevel / distortion terms, 2^28 / waves for the per-wave squared-error and BCE sums) makes every loss of the call
 * NaN instead of wrapping the sum.
 * S_p / spacing_p / weights_p / d_weights_p are host arrays of n_levels entries.
 * d_density_p (optional, with euclid_p [R,S_p+1] and density_p [R,S_p]): level l's d(loss)/d(density) [R,S_p[l]] =
 * fnr_weights_bwd of that level with d_weights_p[l] as upstream, computed in the same pass (bit-identical); then
 * d_weights_p / d_weights_p[l] may be NULL. */
#define FNR_MAX_PROPOSAL_LEVELS 4
#define FNR_TRAIN_LOSSES_ACCUM_FLOATS (4 * FNR_LOSS_SLOTS + 33 * 32)
int fnr_train_losses(int64_t n_rays, const float* rgb, const float* image, const float* semantics,
                     const float* fruit_mask, float semantic_loss_weight, float* d_rgb, float* d_semantics, int S_f,
                     const float* spacing_f, const float* weights_f, int n_levels, const int* S_p,
                     const float* const* spacing_p, const float* const* weights_p, float* const* d_weights_p,
                     const float* const* euclid_p, const float* const* density_p, float* const* d_density_p,
                     float interlevel_mult, int want_distortion, float* accum, float* losses, void* stream);

/* ---- gradient of the rays (camera-pose optimisation, fruit_nerf_config.py:39-43) ----------------- */
/* Input gradient of fnr_hash_encode_fwd: partial [L][N][4] = per level d(loss)/d(unit-cube position) of every
 * sample, from d_feats [L][N][2] and the PARAMETER table (autograd of HashEncoding w.r.t. its input through the
 * trilinear offsets; `positions * selector` zeroes it outside the unit cube, fruit_field.py:178-179). */
int fnr_hash_encode_input_grad(const fnr_grid* grid, const fnr_warp* warp, const fnr_rays* rays,
                               const float* euclid_bins, int S, const float* d_feats, float* partial, void* stream);
/* Sums `partial` [n_levels][N][4] over the levels, applies the transposed Jacobian of the position warp
 * (SceneContraction + (x+2)/4, or AABB normalisation) and of Frustums.get_positions (p = o + d (t0+t1)/2, bins
 * detached), and ADDS the per-ray sums to d_origins [R,3] and d_directions [R,3]. */
int fnr_position_grad_reduce(const fnr_warp* warp, const fnr_rays* rays, const float* euclid_bins, int S,
                             int n_levels, const float* partial, float* d_origins, float* d_directions,
                             void* stream);
/* fnr_position_grad_reduce over up to FNR_MAX_POSITION_SOURCES sources in one launch: source q contributes the ray
 * gradient carried by partials[q] ([n_levels[q]][N_q][4], N_q = n_rays * S[q]) through warps[q] and the frustum chain of
 * euclid_bins[q] ([n_rays, S[q] + 1]).  accumulate = 0: d_origins / d_directions are WRITTEN (no zero fill needed);
 * 1: added to.  The per-source sums are those of fnr_position_grad_reduce, added in source order. */
#define FNR_MAX_POSITION_SOURCES 4
int fnr_position_grad_reduce_multi(int n_sources, const fnr_warp* const* warps, const fnr_rays* rays,
                                   const float* const* euclid_bins, const int* S, const int* n_levels,
                                   const float* const* partials, int accumulate, float* d_origins, float* d_directions,
                                   void* stream);

/* Same result from the Jacobian fnr_hash_encode_fwd saved: d_origins / d_directions [R,3] += the ray gradient of
 * d_feats [L][N][2] (no table gathers in the backward pass). */
int fnr_position_grad_from_jacobian(const fnr_warp* warp, const fnr_rays* rays, const float* euclid_bins, int S,
                                    int n_levels, const float* jacobian, const float* d_feats, float* d_origins,
                                    float* d_directions, void* stream);

/* torch.optim.Adam step (no amsgrad; fruit_nerf_config.py:47-56) over a flat arena of n floats (n % 4 == 0); the
 * gradient is multiplied by grad_scale first (1/world_size after an all-reduce(SUM)), weight_decay is torch's L2 form
 * (grad += weight_decay * param; 0 for the model's groups, 1e-2 for the camera optimiser, fruit_nerf_config.py:41),
 * and the gradient is zeroed afterwards when zero_grad != 0.  `step` is the 1-based step count for the bias
 * corrections. */
int fnr_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, int64_t step, float grad_scale, float weight_decay, int zero_grad,
                  void* stream);
/* torch.optim.RAdam step with the same conventions (RAdamOptimizerConfig of fruit_nerf_big / fruit_nerf_huge,
 * fruit_nerf_config.py:77-80,97-106,125-160): rectified adaptive update once rho_t > 5, plain bias-corrected momentum
 * before. */
int fnr_radam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                   float beta2, float eps, int64_t step, float grad_scale, float weight_decay, int zero_grad,
                   void* stream);

/* The optimisers of a method's parameter groups (fruit_nerf_config.py:47-56, 97-106, 148-160: one torch.optim instance and
 * scheduler per group) as ONE launch over several spans of one arena (the groups differ in learning rate and step count;
 * each is a few thousand to a few million floats and a launch of its own cost more than its traffic).  Span k updates
 * elements [offset, offset + count) (both multiples of 4) exactly as fnr_adam_step (algorithm 0) / fnr_radam_step (1)
 * with its own lr / step would.  n_spans <= FNR_MAX_ADAM_SPANS; `spans` is host memory. */
#define FNR_MAX_ADAM_SPANS 8
typedef struct fnr_adam_span {
  int64_t offset, count;
  int64_t step;
  float lr;
  int32_t reserved;
} fnr_adam_span;
int fnr_adam_step_spans(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int n_spans,
                        const fnr_adam_span* spans, int algorithm, float beta1, float beta2, float eps,
                        float grad_scale, float weight_decay, int zero_grad, void* stream);

/* ---- step programs: a training step's launch sequence, recorded once and replayed natively (ABI 12) ------------- */
/* The reference's loop is nerfstudio's Python Trainer (fruit_pipeline.py:120-146 under Trainer.train_iteration); ours
 * (fruitnerf_amd/training.py::TrainingSteps) is ~30 calls of this ABI per step on two HIP streams.  Every pointer of a
 * step is stable (parameter arenas, persistent workspaces, the caller's double-buffered step arena), so the sequence is
 * recorded ONCE per step shape and replayed by one call: between fnr_program_begin and fnr_program_end the recordable
 * entry points called on this thread — fnr_train_prologue, fnr_prop_density_fwd, fnr_weights_pdf, fnr_hash_encode_fwd,
 * fnr_field_mlp_fwd, fnr_composite_fwd, fnr_train_losses, fnr_composite_bwd_targets, fnr_field_mlp_bwd_adam,
 * fnr_hash_encode_bwd_adam, fnr_prop_density_bwd_pair(_split), fnr_position_grad_reduce_multi,
 * fnr_camera_pose_grad_adam and the stream operations below — run as usual AND append themselves (arguments by value,
 * host structs / host arrays copied) to the program; any other entry point that enqueues device work poisons the
 * recording (fnr_program_end then fails and the program stays empty).  fnr_program_replay calls them again in order, on
 * the streams they were recorded on, with the per-step scalars patched in.  The caller keeps every recorded buffer
 * alive and unchanged in place for as long as it replays the program. */
#define FNR_PROGRAM_ADAM_SLOTS 4
typedef struct fnr_step_scalars {
  uint64_t prologue_offset; /* fnr_train_prologue: `offset` (the step's random-number counter) */
  float anneal;             /* fnr_weights_pdf: `anneal` */
  int32_t reserved;
  struct {
    float lr;
    int32_t reserved;
    int64_t step;           /* 0 = keep the recorded lr / step of this slot */
  } adam[FNR_PROGRAM_ADAM_SLOTS]; /* by fnr_table_adam.slot - 1 */
  float* losses;            /* fnr_train_losses: `losses` (5 floats, device); NULL = the recorded buffer */
} fnr_step_scalars;
typedef struct fnr_program fnr_program;
int fnr_program_create(fnr_program** out);
int fnr_program_destroy(fnr_program* program);
/* begin: the program is emptied and this thread records into it; end: stops recording, fails (and empties the program)
 * if an unrecordable entry point ran in between; abort: stops recording and empties the program. */
int fnr_program_begin(fnr_program* program);
int fnr_program_end(fnr_program* program);
int fnr_program_abort(fnr_program* program);
int64_t fnr_program_size(const fnr_program* program);
/* name of the entry point behind operation i (a static string), NULL when out of range */
const char* fnr_program_op_name(const fnr_program* program, int64_t i);
/* scalars NULL: every operation with its recorded arguments.  Stops at the first failing operation and returns its code. */
int fnr_program_replay(const fnr_program* program, const fnr_step_scalars* scalars);
/* Stream dependencies (recordable): events are hipEvent_t created with timing disabled, owned by the caller.
 * fnr_stream_wait_stream: work enqueued on `waiting` after the call runs after everything enqueued on `signalling`
 * before it (one event of an internal pool is recorded and waited for; the host does not block). */
int fnr_event_create(void** event_out);
int fnr_event_destroy(void* event);
int fnr_event_record(void* event, void* stream);
int fnr_stream_wait_event(void* stream, void* event);
int fnr_stream_wait_stream(void* waiting, void* signalling);

/* ---- export --------------------------------------------------------------------------------- */
/* sample_volume's masks + gathers (export/exporter_utils.py:111-153) as an order-preserving stream
 * compaction.  Sets: 0 = semantic_colormap (sigmoid(logit) > 0.9 and density >= 70),
 * 1 = semantic (logit >= 3 and density >= 70), 2 = density (density >= 70).
 * points[s]: [capacity,3] world positions (fruit_nerf.py:259), colors[s]: [capacity,4] = rgb || sigmoid(x)
 * with x = logit (sets 0,1) or density (set 2).  counts: uint64[3] running totals (device), advanced by
 * this call; entries beyond `capacity` are counted but not written.
 * Sample positions come either from the lattice (lat != NULL: rays [ray_begin, ray_begin+n_rays) x n_z) or
 * from an explicit array positions[n_positions,3] (lat == NULL; outputs['point_location'], fruit_nerf.py:259).
 * workspace: >= fnr_export_workspace_bytes(n_samples) bytes. */
size_t fnr_export_workspace_bytes(int64_t n_samples);
int fnr_export_compact(const fnr_lattice* lat, int64_t ray_begin, int64_t n_rays, const float* positions,
                       int64_t n_positions, const float* density,
                       const float* rgb, const float* logit, float* const points[3], float* const colors[3],
                       int64_t capacity, uint64_t* counts, void* workspace, void* stream);

/* ---- point-cloud front-end of the counting stage ---------------------------------------------- */
/* FruitClustering.cluster (clustering/clustering_base.py:183-207) runs, on the exported cloud, Open3D's
 * remove_radius_outlier (:141-143), Open3D's voxel_down_sample (:138-139) and sklearn.cluster.DBSCAN (:199-200).
 * These three entry points replace those library calls; clouds are [n][3] float64 device arrays (Open3D / PLY
 * precision), every integer result (counts, labels) is bit-exact with the CPU libraries.  lo/hi/min_bound/max_bound
 * are HOST pointers to 3 doubles: the axis-aligned bounds of the cloud (Open3D's GetMinBound/GetMaxBound).
 * workspace: >= fnr_cloud_workspace_bytes(n) bytes, caller-owned, contents undefined afterwards. */
size_t fnr_cloud_workspace_bytes(int64_t n_points);
/* lo_hi[6] (device) = min x, y, z, max x, y, z of the cloud (n >= 1); workspace >= 64 bytes. */
int fnr_cloud_bounds(const double* xyz, int64_t n, double* lo_hi, void* workspace, size_t workspace_bytes,
                     void* stream);
/* counts[i] = #{j : |p_i - p_j|^2 < radius^2} (inclusive != 0: <=), the point itself included.
 * Open3D's RemoveRadiusOutliers keeps i when counts[i] > nb_points (strict search, inclusive = 0). */
int fnr_cloud_radius_count(const double* xyz, int64_t n, const double* lo, const double* hi, double radius,
                           int inclusive, int32_t* counts, void* workspace, size_t workspace_bytes, void* stream);
/* scikit-learn's DBSCAN(eps, min_samples).fit(X).labels_: labels[n] (-1 = noise, clusters numbered by their first core
 * point in input order, border points join the lowest-numbered adjacent cluster); n_clusters: device int32 (nullable). */
int fnr_cloud_dbscan(const double* xyz, int64_t n, const double* lo, const double* hi, double eps, int32_t min_samples,
                     int32_t* labels, int32_t* n_clusters, void* workspace, size_t workspace_bytes, void* stream);
/* Open3D's VoxelDownSample: one output point per occupied voxel (index = floor((p - (min_bound - voxel_size/2)) /
 * voxel_size)) = mean of its points (and colours; rgb / rgb_out nullable together), summed in input order.  Voxels are
 * emitted in ascending (iz, iy, ix) order (Open3D's order is unspecified).  xyz_out / rgb_out: capacity [n][3];
 * n_out: device int32 = number of occupied voxels. */
int fnr_cloud_voxel_down_sample(const double* xyz, const double* rgb, int64_t n, const double* min_bound,
                                const double* max_bound, double voxel_size, double* xyz_out, double* rgb_out,
                                int32_t* n_out, void* workspace, size_t workspace_bytes, void* stream);

/* ---- eval-image metrics ------------------------------------------------------------------------ */
/* FruitModel.get_image_metrics_and_images (fruit_nerf/fruit_nerf.py:403-458) on the device: everything the reference
 * computes with torchmetrics on a rendered [H, W, 3] image, as raw sums (the host forms the four ratios):
 *   PSNR(data_range 1, :427): squared error of clamp(rgb, 0, 1) (:408) against the image;
 *   SSIM (:428; torchmetrics defaults: 11 x 11 gaussian, sigma 1.5, k1 0.01, k2 0.03, data_range None = the larger of
 *        the value ranges of the image and of clamp(rgb, 0, 1), found on the device): the mean over the
 *        (H - 10) x (W - 10) x 3 interior values of the valid-window SSIM map (torchmetrics pads by 5 and crops the
 *        padded border away again); gauss11: HOST array of the 11 normalised float32 window weights;
 *   IoU  (:449-453): BinaryJaccardIndex(threshold 0.5) of `F.softmax(semantics)` — no dim given: on the [H, W, 1] map
 *        torch's implicit dim is 0, a softmax over image ROWS — and of sigmoid(semantics), each against mask > 0.5.
 * rgb / image: [H][W][3]; semantics (logits) / mask: [H][W], nullable together.
 * out: 8 doubles (device) = { squared error, SSIM sum, intersection | union of sigmoid > 0.5, intersection | union of the
 * row softmax > 0.5, number of SSIM values, number of squared-error terms }.
 * workspace: >= fnr_image_metrics_workspace_bytes(H, W) bytes, caller-owned.  H, W > 10. */
size_t fnr_image_metrics_workspace_bytes(int H, int W);
int fnr_image_metrics(int H, int W, const float* rgb, const float* image, const float* semantics, const float* mask,
                      const float* gauss11, double* out, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FRUITNERF_HIP_H */
