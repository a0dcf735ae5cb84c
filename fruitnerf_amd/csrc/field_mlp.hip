// field_mlp.hip — FruitField's MLP stack on fp32 MFMA (v_mfma_f32_16x16x4_f32), forward.
//   mlp_base_mlp (32->64->1+geo) + trunc_exp*selector, mlp_semantics (geo->64->64 | geo->128->128->64) +
//   SemanticFieldHead (64->1), SHEncoding(4) + appearance embedding + mlp_head (16+geo+32->64->64->3, sigmoid)
//   fruit_field.py:132-166,187-281; geo = 15 (`fruit_nerf`) or 30 (`fruit_nerf_big` / `fruit_nerf_huge`)
// One wave = one 16-sample tile per iteration; activations stay in registers between layers
// (see field_layers.hpp); weights live in LDS for the whole (persistent) workgroup.
// Roofline: MFMA fp32 (157.3 TF peak): 33 024 useful FLOP/sample, 36 864 issued (padding).
#include "field_layers.hpp"
#include "field_bf16.hpp"
#include "sequencer.hpp"

namespace fnr {

// which FieldCfg a net describes (0 = FieldCfgBase, 1 = FieldCfgBig) and its nn.Linear pointers in that Cfg's layer order
int field_ptrs(const fnr_field_net* net, FieldPtrs& p, int* cfg_id) {
  FNR_UNSUPPORTED(net->grid.n_levels == 16, "field_mlp: num_levels %d not built (16 only)", net->grid.n_levels);
  const bool common = net->hidden_dim == 64 && net->hidden_dim_color == 64 && net->semantic_out_dim == 64 &&
                      net->appearance_dim == 32;
  const bool base = common && net->geo_feat_dim == 15 && net->hidden_dim_semantics == 64 && net->num_layers_semantic == 2;
  const bool big = common && net->geo_feat_dim == 30 && net->hidden_dim_semantics == 128 && net->num_layers_semantic == 3;
  FNR_UNSUPPORTED(base || big,
                  "field_mlp: built shapes are `fruit_nerf` (geo 15, semantic 2 x 64) and `fruit_nerf_big`/`huge` (geo 30, "
                  "semantic 3 x 128), base/colour width 64, appearance 32; got geo %d hidden %d/%d/%d sem_layers %d",
                  net->geo_feat_dim, net->hidden_dim, net->hidden_dim_color, net->hidden_dim_semantics,
                  net->num_layers_semantic);
  for (int i = 0; i < FIELD_MAX_LAYERS; ++i) p.w[i] = p.b[i] = nullptr;
  auto fill = [&](auto cfg) {
    using C = decltype(cfg);
    p.w[C::L_BASE0] = net->base_w0; p.b[C::L_BASE0] = net->base_b0;
    p.w[C::L_BASE1] = net->base_w1; p.b[C::L_BASE1] = net->base_b1;
    p.w[C::L_SEM0] = net->sem_w[0]; p.b[C::L_SEM0] = net->sem_b[0];
    p.w[C::L_SEM1] = net->sem_w[1]; p.b[C::L_SEM1] = net->sem_b[1];
    if constexpr (C::NSEM == 3) { p.w[C::L_SEM2] = net->sem_w[2]; p.b[C::L_SEM2] = net->sem_b[2]; }
    p.w[C::L_HEAD] = net->head_w;  p.b[C::L_HEAD] = net->head_b;
    p.w[C::L_COL0] = net->col_w[0]; p.b[C::L_COL0] = net->col_b[0];
    p.w[C::L_COL1] = net->col_w[1]; p.b[C::L_COL1] = net->col_b[1];
    p.w[C::L_COL2] = net->col_w[2]; p.b[C::L_COL2] = net->col_b[2];
    return (int)C::NLAYERS;
  };
  const int nl = big ? fill(FieldCfgBig{}) : fill(FieldCfgBase{});
  for (int i = 0; i < nl; ++i) FNR_CHECK_ARG(p.w[i] && p.b[i], "field_mlp: null weight/bias pointer (layer %d)", i);
  *cfg_id = big ? 1 : 0;
  return FNR_OK;
}

// bf16 / bf16x3 modes (field_mlp_bf16.hip)
int field_mlp_fwd_bf16(int cfg, int mode, const FieldPtrs& p, const float* packed, void* image_ws, const float* ray_bias,
                       const RaysDev& rd, int S, long long N, const float2* feats, const uint8_t* selector, float* density,
                       float* rgb, float* logit, float* geo_out, float* h_buf, hipStream_t st);
size_t field_bf16_image_bytes();
int field_mlp_fwd_sem_big_bf16(int mode, const FieldPtrs& p, void* image_ws, const float* packed, long long N,
                               const float* h_buf, float* logit, hipStream_t st);
// workspace layout of fnr_field_mlp_fwd: [fp32 fragment image | bf16 fragment image (3 pieces) | per-ray colour bias]
static inline size_t fwd_ws_packed_bytes() { return ((size_t)(FIELD_MAX_PACKED_FLOATS + 64) * sizeof(float) + 255) / 256 * 256; }
size_t field_fwd_ws_image_offset() { return fwd_ws_packed_bytes(); }
static inline size_t fwd_ws_fixed_bytes() { return fwd_ws_packed_bytes() + (field_bf16_image_bytes() + 255) / 256 * 256; }

// ---- per-tile pieces shared by the forward and backward kernels ("R" = the LdsRange the kernel staged) ----------

// B operand of base layer 0 from the level-major features: lane group g covers levels {g, 4+g, 8+g, 12+g}
__device__ __forceinline__ void load_hash_features(const float2* __restrict__ feats, long long N, long long nn, int g,
                                                   f32x4 (&x0)[2]) {
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const float2 v = feats[(size_t)(4 * m + g) * N + nn];
    x0[m >> 1][2 * (m & 1)] = v.x;
    x0[m >> 1][2 * (m & 1) + 1] = v.y;
  }
}

// mlp_semantics + SemanticFieldHead (fruit_field.py:145-156,201-206,263-268); s_last = mlp_semantics' output (no
// activation on its last layer), hd = the logit block (row 0)
template <class Cfg, class R>
__device__ __forceinline__ void semantic_forward(const float* __restrict__ lds, const f32x4 (&h)[Cfg::HB],
                                                 f32x4 (&s1)[Cfg::SEMB], f32x4 (&s2)[Cfg::NSEM == 3 ? Cfg::SEMB : 4],
                                                 f32x4 (&s3)[4], f32x4 (&hd)[1], int lane) {
  mlp_layer<Cfg::SEMB, Cfg::HB>(R::w(lds, Cfg::L_SEM0), R::b(lds, Cfg::L_SEM0), h, s1, lane);
  relu_(s1);
  if constexpr (Cfg::NSEM == 3) {
    mlp_layer<Cfg::SEMB, Cfg::SEMB>(R::w(lds, Cfg::L_SEM1), R::b(lds, Cfg::L_SEM1), s1, s2, lane);
    relu_(s2);
    mlp_layer<4, Cfg::SEMB>(R::w(lds, Cfg::L_SEM2), R::b(lds, Cfg::L_SEM2), s2, s3, lane);
    mlp_layer<1, 4>(R::w(lds, Cfg::L_HEAD), R::b(lds, Cfg::L_HEAD), s3, hd, lane);
  } else {
    mlp_layer<4, 4>(R::w(lds, Cfg::L_SEM1), R::b(lds, Cfg::L_SEM1), s1, s2, lane);
    mlp_layer<1, 4>(R::w(lds, Cfg::L_HEAD), R::b(lds, Cfg::L_HEAD), s2, hd, lane);
  }
}

enum { PART_ALL = 0, PART_BASE_COLOR = 1, PART_SEM = 2 };
template <class Cfg, int PART>
struct FwdRange;
template <class Cfg>
struct FwdRange<Cfg, PART_ALL> { using type = LdsRange<Cfg, 0, Cfg::NLAYERS>; };
template <class Cfg>
struct FwdRange<Cfg, PART_BASE_COLOR> { using type = LdsRange<Cfg, Cfg::L_BASE0, Cfg::L_COL2 + 1>; };
template <class Cfg>
struct FwdRange<Cfg, PART_SEM> { using type = LdsRange<Cfg, Cfg::L_SEM0, Cfg::L_HEAD + 1>; };

// PART_ALL: the whole stack in one launch (FieldCfgBase: the 75 KB image fits twice in a CU's LDS).
// FieldCfgBig (176 KB image): PART_BASE_COLOR writes h [N, 16 HB] (h_buf), PART_SEM reads it back (128 B / sample).
template <class Cfg, int PART, int WAVES>
__global__ __launch_bounds__(64 * WAVES, (PART == PART_SEM) ? WAVES / 4 : 4) void k_field_mlp_fwd(
    const float* __restrict__ packed, const float* __restrict__ ray_bias, RaysDev rays, int S, long long N,
    const float2* __restrict__ feats, const uint8_t* __restrict__ selector, float* __restrict__ density,
    float* __restrict__ rgb, float* __restrict__ logit, float* __restrict__ geo_out, float* __restrict__ h_buf) {
  using R = typename FwdRange<Cfg, PART>::type;
  constexpr int HB = Cfg::HB;
  __shared__ __attribute__((aligned(16))) float lds[R::FLOATS];
  R::template stage<64 * WAVES>(lds, packed);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const long long n_tiles = (N + 15) / 16;
  for (long long tile = (long long)blockIdx.x * WAVES + wave; tile < n_tiles; tile += (long long)gridDim.x * WAVES) {
    // weights are loop-invariant LDS reads: without this barrier LICM hoists all 72 KiB of fragments
    // into registers and spills them to scratch.
    asm volatile("" ::: "memory");
    const long long n = tile * 16 + j;
    const bool valid = n < N;
    const long long nn = valid ? n : N - 1;
    const long long ray = nn / S;

    f32x4 h[HB];
    if constexpr (PART != PART_SEM) {
      f32x4 x0[2], a1[4];
      load_hash_features(feats, N, nn, g, x0);
      mlp_layer<4, 2>(R::w(lds, Cfg::L_BASE0), R::b(lds, Cfg::L_BASE0), x0, a1, lane);
      relu_(a1);
      mlp_layer<HB, 4>(R::w(lds, Cfg::L_BASE1), R::b(lds, Cfg::L_BASE1), a1, h, lane);
    } else {
#pragma unroll
      for (int b = 0; b < HB; ++b) h[b] = *reinterpret_cast<const f32x4*>(h_buf + (size_t)nn * (16 * HB) + 16 * b + 4 * g);
    }

    // semantic branch (input = geo features = h[1..GEO]; h[0] and the padding have structural-zero weight columns)
    if constexpr (PART != PART_BASE_COLOR) {
      f32x4 s1[Cfg::SEMB], s2[Cfg::NSEM == 3 ? Cfg::SEMB : 4], s3[4], hd[1];
      semantic_forward<Cfg, R>(lds, h, s1, s2, s3, hd, lane);
      if (g == 0 && valid) logit[n] = hd[0][0];
    }

    // colour branch: [h | SH16(d') | appearance embedding]; the ray-constant part arrives as ray_bias
    if constexpr (PART != PART_SEM) {
      f32x4 c1[4], c2[4], c3[1];
      color_layer0<Cfg>(R::w(lds, Cfg::L_COL0), ray_bias, ray, h, c1, lane);
      relu_(c1);
      mlp_layer<4, 4>(R::w(lds, Cfg::L_COL1), R::b(lds, Cfg::L_COL1), c1, c2, lane);
      relu_(c2);
      mlp_layer<1, 4>(R::w(lds, Cfg::L_COL2), R::b(lds, Cfg::L_COL2), c2, c3, lane);

      if (h_buf && valid) {
#pragma unroll
        for (int b = 0; b < HB; ++b) *reinterpret_cast<f32x4*>(h_buf + (size_t)n * (16 * HB) + 16 * b + 4 * g) = h[b];
      }
      if (geo_out && valid) {
#pragma unroll
        for (int b = 0; b < HB; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int k = 16 * b + 4 * g + r;  // h index; geo feature k-1
            if (k >= 1 && k <= Cfg::GEO) geo_out[(size_t)n * Cfg::GEO + (k - 1)] = h[b][r];
          }
      }
      if (g == 0 && valid) {
        const bool sel = selector ? (selector[n] != 0) : true;
        density[n] = sel ? expf(h[0][0]) : 0.0f;  // trunc_exp forward * selector (fruit_field.py:191-192)
        rgb[3 * n + 0] = 1.0f / (1.0f + expf(-c3[0][0]));
        rgb[3 * n + 1] = 1.0f / (1.0f + expf(-c3[0][1]));
        rgb[3 * n + 2] = 1.0f / (1.0f + expf(-c3[0][2]));
      }
    }
  }
}

__global__ void k_embedding_mean(const float* __restrict__ emb, int n, int dim, float* __restrict__ out) {
  // one wave per column; sequential-ish order is irrelevant at 1e-7
  const int c = blockIdx.x;
  const int lane = threadIdx.x;
  float s = 0.0f;
  for (int i = lane; i < n; i += 64) s += emb[(size_t)i * dim + c];
  s = wave_sum(s);
  if (lane == 0) out[c] = s / (float)n;
}

}  // namespace fnr

using namespace fnr;

extern "C" size_t fnr_field_mlp_fwd_workspace_bytes(int64_t n_rays) {
  // fragment images of the weights (fp32, sized for the larger of the two built shapes; bf16 pieces for the
  // FNR_MLP_BF16* modes) + per-ray colour bias [n_rays, 64]
  return fwd_ws_fixed_bytes() + 256 + (size_t)(n_rays > 0 ? n_rays : 0) * 64 * sizeof(float);
}

extern "C" int fnr_field_h_dim(const fnr_field_net* net) {
  FieldPtrs p;
  int cfg = 0;
  if (!net || field_ptrs(net, p, &cfg)) return -1;
  return cfg ? 16 * FieldCfgBig::HB : 16 * FieldCfgBase::HB;
}

namespace {
template <class Cfg>
int field_mlp_fwd_launch(const FieldPtrs& p, const fnr_field_net* net, const RaysDev& rd, int S, long long N,
                         const float* feats, const uint8_t* selector, const float* mean_embedding, float* density,
                         float* rgb, float* logit, float* geo_out, float* h_buf, float* packed, float* ray_bias,
                         hipStream_t st) {
  // fp32 fragment image + (bf16-pipe modes) the bf16 pieces + per-ray colour bias: one launch
  launch_prepare_field<Cfg>(p, packed, net->mlp_mode == FNR_MLP_FP32 ? 0 : (net->mlp_mode == FNR_MLP_BF16 ? 1 : 3),
                            reinterpret_cast<__bf16*>(reinterpret_cast<char*>(packed) + field_fwd_ws_image_offset()), rd,
                            net->embedding, mean_embedding, ray_bias, st);
  FNR_LAUNCH_CHECK();
  const long long n_tiles = (N + 15) / 16;
  const float2* f2 = reinterpret_cast<const float2*>(feats);
  // bf16-pipe modes: every layer of both shapes (field_mlp_bf16.hip); `fruit_nerf_big` = the tile-pair kernel for base +
  // colour, then the weight-streamed semantic branch
  if constexpr (Cfg::NSEM == 2) {
    if (net->mlp_mode != FNR_MLP_FP32)
      return field_mlp_fwd_bf16(0, net->mlp_mode, p, packed, reinterpret_cast<char*>(packed) + field_fwd_ws_image_offset(),
                                ray_bias, rd, S, N, f2, selector, density, rgb, logit, geo_out, h_buf, st);
  }
  if constexpr (Cfg::NSEM == 2) {
    long long blocks = (n_tiles + 7) / 8;
    const long long max_blocks = 2ll * device_cu_count();
    if (blocks > max_blocks) blocks = max_blocks;
    hipLaunchKernelGGL((k_field_mlp_fwd<Cfg, PART_ALL, 8>), dim3((unsigned)blocks), dim3(512), 0, st, packed, ray_bias, rd,
                       S, N, f2, selector, density, rgb, logit, geo_out, h_buf);
    FNR_LAUNCH_CHECK();
  } else {
    if (net->mlp_mode != FNR_MLP_FP32) {
      // bf16-pipe modes: base + colour on the tile-pair kernel, then the semantic branch weight-streamed from the saved h
      const int rc = field_mlp_fwd_bf16(1, net->mlp_mode, p, packed,
                                        reinterpret_cast<char*>(packed) + field_fwd_ws_image_offset(), ray_bias, rd, S, N, f2,
                                        selector, density, rgb, logit, geo_out, h_buf, st);
      if (rc) return rc;
      return field_mlp_fwd_sem_big_bf16(net->mlp_mode, p, reinterpret_cast<char*>(packed) + field_fwd_ws_image_offset(),
                                        packed, N, h_buf, logit, st);
    }
    long long blocks = (n_tiles + 7) / 8;
    if (blocks > 2ll * device_cu_count()) blocks = 2ll * device_cu_count();
    hipLaunchKernelGGL((k_field_mlp_fwd<Cfg, PART_BASE_COLOR, 8>), dim3((unsigned)blocks), dim3(512), 0, st, packed,
                       ray_bias, rd, S, N, f2, selector, density, rgb, logit, geo_out, h_buf);
    FNR_LAUNCH_CHECK();
    blocks = (n_tiles + 15) / 16;   // 120 KB of semantic weights: one 16-wave workgroup per CU
    if (blocks > (long long)device_cu_count()) blocks = device_cu_count();
    hipLaunchKernelGGL((k_field_mlp_fwd<Cfg, PART_SEM, 16>), dim3((unsigned)blocks), dim3(1024), 0, st, packed, ray_bias,
                       rd, S, N, f2, selector, density, rgb, logit, geo_out, h_buf);
    FNR_LAUNCH_CHECK();
  }
  return FNR_OK;
}
}  // namespace

extern "C" int fnr_field_mlp_fwd(const fnr_field_net* net, const fnr_rays* rays, int S, const float* feats,
                                 const uint8_t* selector, const float* mean_embedding, float* density, float* rgb,
                                 float* logit, float* geo_out, float* h_save, float* ray_bias_save,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  if (seq::recording() && net && rays) {
    const fnr_field_net net_ = *net;
    const fnr_rays rays_ = *rays;
    seq::push("fnr_field_mlp_fwd", [=](const fnr_step_scalars*) {
      return fnr_field_mlp_fwd(&net_, &rays_, S, feats, selector, mean_embedding, density, rgb, logit, geo_out, h_save,
                               ray_bias_save, workspace, workspace_bytes, stream);
    });
  }
  FNR_CHECK_ARG(net && rays && feats && density && rgb && logit && S > 0, "field_mlp_fwd: null argument");
  FNR_CHECK_ARG(rays->directions, "field_mlp_fwd: rays.directions is null");
  FNR_CHECK_ARG(mean_embedding || (rays->camera_indices && net->embedding),
                "field_mlp_fwd: training path needs rays.camera_indices and net.embedding "
                "(\"Camera indices are not provided.\", fruit_field.py:240-241)");
  FieldPtrs p;
  int cfg = 0;
  int rc = field_ptrs(net, p, &cfg);
  if (rc) return rc;
  const long long N = rays->n_rays * (long long)S;
  if (N == 0) return FNR_OK;
  FNR_CHECK_ARG(workspace && workspace_bytes >= fnr_field_mlp_fwd_workspace_bytes(ray_bias_save ? 0 : rays->n_rays),
                "field_mlp_fwd: workspace too small");
  FNR_CHECK_ARG(cfg == 0 || h_save, "field_mlp_fwd: the fruit_nerf_big shape runs as two launches that hand the base MLP's "
                "output over in h_save [N, fnr_field_h_dim()] — pass that buffer");
  FNR_CHECK_ARG(net->mlp_mode == FNR_MLP_FP32 || net->mlp_mode == FNR_MLP_BF16 || net->mlp_mode == FNR_MLP_BF16X3,
                "field_mlp_fwd: mlp_mode %d (FNR_MLP_FP32 0 | FNR_MLP_BF16 1 | FNR_MLP_BF16X3 3)", net->mlp_mode);
  float* packed = reinterpret_cast<float*>(workspace);
  float* ray_bias = ray_bias_save ? ray_bias_save
                                  : reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + fwd_ws_fixed_bytes());
  const RaysDev rd = make_rays(rays);
  FNR_PROF(OP_MLP_FWD, N);
  if (cfg == 0)
    return field_mlp_fwd_launch<FieldCfgBase>(p, net, rd, S, N, feats, selector, mean_embedding, density, rgb, logit, geo_out,
                                              h_save, packed, ray_bias, as_stream(stream));
  return field_mlp_fwd_launch<FieldCfgBig>(p, net, rd, S, N, feats, selector, mean_embedding, density, rgb, logit, geo_out,
                                           h_save, packed, ray_bias, as_stream(stream));
}

extern "C" int fnr_embedding_mean(const float* embedding, int n_images, int dim, float* out, void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_embedding_mean");
  FNR_CHECK_ARG(embedding && out && n_images > 0 && dim > 0, "embedding_mean: bad argument");
  hipLaunchKernelGGL(k_embedding_mean, dim3(dim), dim3(64), 0, as_stream(stream), embedding, n_images, dim, out);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}
