// field_mlp.hip — FruitField's MLP stack on fp32 MFMA (v_mfma_f32_16x16x4_f32), forward.
//   mlp_base_mlp (32->64->16) + trunc_exp*selector, mlp_semantics (15->64->64) + SemanticFieldHead (64->1),
//   SHEncoding(4) + appearance embedding + mlp_head (63->64->64->3, sigmoid)        fruit_field.py:132-166,187-281
// One wave = one 16-sample tile per iteration; activations stay in registers between layers
// (see field_layers.hpp); weights live in LDS for the whole (persistent) workgroup.
// Roofline: MFMA fp32 (157.3 TF peak): 33 024 useful FLOP/sample, 36 864 issued (padding).
#include "field_layers.hpp"

namespace fnr {

int field_ptrs(const fnr_field_net* net, FieldPtrs& p) {
  FNR_UNSUPPORTED(net->grid.n_levels == 16, "field_mlp: num_levels %d not built (16 only)", net->grid.n_levels);
  FNR_UNSUPPORTED(net->geo_feat_dim == 15 && net->hidden_dim == 64 && net->hidden_dim_color == 64 &&
                      net->hidden_dim_semantics == 64 && net->num_layers_semantic == 2 &&
                      net->semantic_out_dim == 64 && net->appearance_dim == 32,
                  "field_mlp: only the `fruit_nerf` MLP shape is built (geo 15, widths 64, 2 semantic layers, "
                  "appearance 32); got geo %d hidden %d/%d/%d sem_layers %d",
                  net->geo_feat_dim, net->hidden_dim, net->hidden_dim_color, net->hidden_dim_semantics,
                  net->num_layers_semantic);
  p.w[0] = net->base_w0; p.b[0] = net->base_b0;
  p.w[1] = net->base_w1; p.b[1] = net->base_b1;
  p.w[2] = net->sem_w[0]; p.b[2] = net->sem_b[0];
  p.w[3] = net->sem_w[1]; p.b[3] = net->sem_b[1];
  p.w[4] = net->head_w;  p.b[4] = net->head_b;
  p.w[5] = net->col_w[0]; p.b[5] = net->col_b[0];
  p.w[6] = net->col_w[1]; p.b[6] = net->col_b[1];
  p.w[7] = net->col_w[2]; p.b[7] = net->col_b[2];
  for (int i = 0; i < 8; ++i) FNR_CHECK_ARG(p.w[i] && p.b[i], "field_mlp: null weight/bias pointer (layer %d)", i);
  return FNR_OK;
}

template <class Cfg>
__global__ __launch_bounds__(512, 4) void k_field_mlp_fwd(const float* __restrict__ packed, const float* __restrict__ ray_bias,
                                                          RaysDev rays, int S, long long N,
                                                          const float2* __restrict__ feats,
                                                          const uint8_t* __restrict__ selector,
                                                          float* __restrict__ density, float* __restrict__ rgb,
                                                          float* __restrict__ logit, float* __restrict__ geo_out,
                                                          float* __restrict__ h_save) {
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  stage_field_weights<Cfg>(lds, packed);
  __syncthreads();
  const float* Bv = lds + Cfg::W_TOTAL;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const long long n_tiles = (N + 15) / 16;
  for (long long tile = (long long)blockIdx.x * 8 + wave; tile < n_tiles; tile += (long long)gridDim.x * 8) {
    // weights are loop-invariant LDS reads: without this barrier LICM hoists all 72 KiB of fragments
    // into registers and spills them to scratch.
    asm volatile("" ::: "memory");
    const long long n = tile * 16 + j;
    const bool valid = n < N;
    const long long nn = valid ? n : N - 1;
    const long long ray = nn / S;

    // B operand of base layer 0: lane group g covers levels {g, 4+g, 8+g, 12+g}
    f32x4 x0[2];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float2 v = feats[(size_t)(4 * m + g) * N + nn];
      x0[m >> 1][2 * (m & 1)] = v.x;
      x0[m >> 1][2 * (m & 1) + 1] = v.y;
    }
    f32x4 a1[4];
    mlp_layer<4, 2>(lds + Cfg::woff(0), Bv + Cfg::boff(0), x0, a1, lane);
    relu_(a1);
    f32x4 h[1];
    mlp_layer<1, 4>(lds + Cfg::woff(1), Bv + Cfg::boff(1), a1, h, lane);

    // semantic branch (input = geo features = h[1..15]; h[0] has a structural-zero weight column)
    f32x4 s1[4], s2[4], hd[1];
    mlp_layer<4, 1>(lds + Cfg::woff(2), Bv + Cfg::boff(2), h, s1, lane);
    relu_(s1);
    mlp_layer<4, 4>(lds + Cfg::woff(3), Bv + Cfg::boff(3), s1, s2, lane);
    mlp_layer<1, 4>(lds + Cfg::woff(4), Bv + Cfg::boff(4), s2, hd, lane);

    // colour branch: [h | SH16(d') | appearance embedding]; the ray-constant part arrives as ray_bias
    f32x4 c1[4], c2[4], c3[1];
    color_layer0<Cfg>(lds, ray_bias, ray, h, c1, lane);
    relu_(c1);
    mlp_layer<4, 4>(lds + Cfg::woff(6), Bv + Cfg::boff(6), c1, c2, lane);
    relu_(c2);
    mlp_layer<1, 4>(lds + Cfg::woff(7), Bv + Cfg::boff(7), c2, c3, lane);

    if (h_save && valid) *reinterpret_cast<f32x4*>(h_save + (size_t)n * 16 + 4 * g) = h[0];
    if (geo_out && valid) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 4 * g + r;  // h index; geo feature k-1
        if (k >= 1) geo_out[(size_t)n * Cfg::GEO + (k - 1)] = h[0][r];
      }
    }
    if (g == 0 && valid) {
      const bool sel = selector ? (selector[n] != 0) : true;
      density[n] = sel ? expf(h[0][0]) : 0.0f;  // trunc_exp forward * selector (fruit_field.py:191-192)
      logit[n] = hd[0][0];
      rgb[3 * n + 0] = 1.0f / (1.0f + expf(-c3[0][0]));
      rgb[3 * n + 1] = 1.0f / (1.0f + expf(-c3[0][1]));
      rgb[3 * n + 2] = 1.0f / (1.0f + expf(-c3[0][2]));
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
  // fragment image of the weights + per-ray colour bias [n_rays, 64]
  return (FieldCfgBase::PACKED_FLOATS + 64) * sizeof(float) + 256 + (size_t)(n_rays > 0 ? n_rays : 0) * 64 * sizeof(float);
}

extern "C" int fnr_field_mlp_fwd(const fnr_field_net* net, const fnr_rays* rays, int S, const float* feats,
                                 const uint8_t* selector, const float* mean_embedding, float* density, float* rgb,
                                 float* logit, float* geo_out, float* h_save, float* ray_bias_save,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  FNR_CHECK_ARG(net && rays && feats && density && rgb && logit && S > 0, "field_mlp_fwd: null argument");
  FNR_CHECK_ARG(rays->directions, "field_mlp_fwd: rays.directions is null");
  FNR_CHECK_ARG(mean_embedding || (rays->camera_indices && net->embedding),
                "field_mlp_fwd: training path needs rays.camera_indices and net.embedding "
                "(\"Camera indices are not provided.\", fruit_field.py:240-241)");
  FieldPtrs p;
  int rc = field_ptrs(net, p);
  if (rc) return rc;
  const long long N = rays->n_rays * (long long)S;
  if (N == 0) return FNR_OK;
  const long long n_tiles = (N + 15) / 16;
  long long blocks = (n_tiles + 7) / 8;
  const long long max_blocks = 2ll * device_cu_count();
  if (blocks > max_blocks) blocks = max_blocks;
  FNR_CHECK_ARG(workspace && workspace_bytes >= fnr_field_mlp_fwd_workspace_bytes(ray_bias_save ? 0 : rays->n_rays),
                "field_mlp_fwd: workspace too small");
  float* packed = reinterpret_cast<float*>(workspace);
  float* ray_bias = ray_bias_save ? ray_bias_save : packed + (FieldCfgBase::PACKED_FLOATS + 63) / 64 * 64;
  const RaysDev rd = make_rays(rays);
  FNR_PROF(OP_MLP_FWD, N);
  launch_pack_field_weights<FieldCfgBase>(p, packed, as_stream(stream));
  FNR_LAUNCH_CHECK();
  launch_color_ray_bias<FieldCfgBase>(packed, rd, net->embedding, mean_embedding, ray_bias, as_stream(stream));
  FNR_LAUNCH_CHECK();
  hipLaunchKernelGGL((k_field_mlp_fwd<FieldCfgBase>), dim3((unsigned)blocks), dim3(512), 0, as_stream(stream), packed,
                     ray_bias, rd, S, N, reinterpret_cast<const float2*>(feats), selector, density, rgb, logit,
                     geo_out, h_save);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

extern "C" int fnr_embedding_mean(const float* embedding, int n_images, int dim, float* out, void* stream) {
  FNR_CHECK_ARG(embedding && out && n_images > 0 && dim > 0, "embedding_mean: bad argument");
  hipLaunchKernelGGL(k_embedding_mean, dim3(dim), dim3(64), 0, as_stream(stream), embedding, n_images, dim, out);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}
