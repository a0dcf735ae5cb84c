// camera_opt.hip — nerfstudio CameraOptimizer(mode="SO3xR3") on the device: the pose corrections the reference's
// datamanager applies to the training cameras and learns from the ray gradients
// (/root/reference/fruit_nerf/fruit_nerf_config.py:39-43; metrics camera_opt_translation / camera_opt_rotation at
// fruit_pipeline.py:133-142).  nerfstudio 0.3.2 semantics (restated):
//   delta[k]  = exp_map_SO3xR3(pose_adjustment[k]):  t = tangent[:3],  R = I + f1 K + f2 K^2,  K = skew(tangent[3:]),
//               theta = sqrt(clamp(|w|^2, 1e-4)), f1 = sin(theta)/theta, f2 = (1 - cos(theta))/theta^2
//   c2w'[k]   = pose_utils.multiply(c2w[k], delta[k]):  R' = R1 R,  t' = t1 + R1 t
//   rays      : origins = t', directions = normalize(R' d_cam)      (Cameras._generate_rays_from_coords)
// k_camera_adjust builds c2w' (thread = camera); k_camera_pose_grad maps d(loss)/d(origins, directions) back to the
// [n_train, 6] tangent vectors: one workgroup per camera gathers its rays (ballot), so there are no atomics on the
// shared rows.
#include "camera_math.hpp"
#include "sequencer.hpp"

namespace fnr {

// c2w' = multiply(c2w[train_ids[k]], exp_map_SO3xR3(pose[k]))
__global__ __launch_bounds__(64) void k_camera_adjust(const float* __restrict__ c2w, const long long* __restrict__ train_ids,
                                                      int n_train, const float* __restrict__ pose,
                                                      float* __restrict__ c2w_adj) {
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= n_train) return;
  float out[12];
  adjusted_camera(c2w + train_ids[k] * 12, pose + 6 * k, out);
#pragma unroll
  for (int i = 0; i < 12; ++i) c2w_adj[12 * k + i] = out[i];
}

constexpr int CPG_ROUND = 4096;   // rays a pose-gradient workgroup compacts at a time (16 per thread: one round per batch)

struct PinholeDev {
  int H, W;
  float fx, fy, cx, cy;
};

// pose_grad[k] += d(loss)/d(tangent[k]) from the ray gradients of camera k's rays.  ADAM (single process): the
// camera's workgroup then takes the optimiser step of its 6 pose parameters itself (the gradient it has just finished
// + whatever pose_grad held) and leaves pose_grad zero — fnr_camera_pose_grad + fnr_adam_step / fnr_radam_step with
// zero_grad, one launch less per step.
template <bool ADAM>
__global__ __launch_bounds__(256) void k_camera_pose_grad(PinholeDev cam, const float* __restrict__ c2w,
                                                          const long long* __restrict__ train_ids, long long n_rays,
                                                          const float* __restrict__ u, const int* __restrict__ cam_idx,
                                                          const float* __restrict__ pose,
                                                          const float* __restrict__ c2w_adj,
                                                          const float* __restrict__ d_origins,
                                                          const float* __restrict__ d_directions,
                                                          float* __restrict__ pose_grad, TableAdam adam) {
  __shared__ float red[4][12];
  __shared__ int s_list[CPG_ROUND];        // rays of this camera among the round's CPG_ROUND rays, in ray order
  __shared__ int s_cnt[CPG_ROUND / 256][4];
  const int k = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* Ma = c2w_adj + 12 * k;  // adjusted camera: R' = rows of Ma[:, :3]
  float acc[12];                       // G (3x3, dL/dR') row-major, then dL/dt'
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = 0.0f;
  // what the single-thread tail needs (all workgroup-uniform) is fetched NOW: the kernel is a chain of dependent
  // memory round trips (~1.5 us each), and these were four more of them at its end
  float Mk[12], wk[3], g_old[6], Pk[6], Mm[6], Vk[6];
  {
    const float* Msrc = c2w + train_ids[k] * 12;
#pragma unroll
    for (int i = 0; i < 12; ++i) Mk[i] = Msrc[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) wk[i] = pose[6 * k + 3 + i];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      g_old[i] = pose_grad[6 * k + i];
      if constexpr (ADAM) {
        Pk[i] = reinterpret_cast<const float*>(adam.p)[6 * k + i];
        Mm[i] = reinterpret_cast<const float*>(adam.m)[6 * k + i];
        Vk[i] = reinterpret_cast<const float*>(adam.v)[6 * k + i];
      }
    }
  }
  // A camera owns ~n_rays / n_cameras of the batch's rays, scattered over it.  Walking the batch with a
  // load-compare-branch per ray, or even with the camera indices of 8 rays fetched at once and the matching rays
  // processed under a divergent branch (one dependent round trip per q in which ANY lane matched: ~12 us), left this
  // 90-workgroup kernel at 19 us.  So: compact first — ballots + a 32-entry count table give every matching ray its
  // slot in ray order (deterministic, no atomics) — then one ray per thread, all loads of a round in one round trip.
  for (long long base0 = 0; base0 < n_rays; base0 += CPG_ROUND) {
    bool mine[CPG_ROUND / 256];
    unsigned long long bal[CPG_ROUND / 256];
#pragma unroll
    for (int q = 0; q < CPG_ROUND / 256; ++q) {
      const long long r = base0 + q * 256 + threadIdx.x;
      mine[q] = (r < n_rays) && cam_idx[r] == k;
    }
#pragma unroll
    for (int q = 0; q < CPG_ROUND / 256; ++q) {
      bal[q] = __ballot(mine[q]);
      if (lane == 0) s_cnt[q][wave] = __popcll(bal[q]);
    }
    __syncthreads();
    int total = 0;
#pragma unroll
    for (int q = 0; q < CPG_ROUND / 256; ++q) {
      int before = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        if (w < wave) before += s_cnt[q][w];
        total += s_cnt[q][w];
      }
      // `total` so far counts q' <= q completely; slots of (q, wave) start after all of q' < q and waves < wave of q
      const int start = total - (s_cnt[q][0] + s_cnt[q][1] + s_cnt[q][2] + s_cnt[q][3]) + before;
      if (mine[q]) s_list[start + __popcll(bal[q] & ((1ull << lane) - 1ull))] = q * 256 + (int)threadIdx.x;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < total; i += 256) {
      const long long r = base0 + s_list[i];
      int y = (int)(u[3 * r + 1] * (float)cam.H);
      int x = (int)(u[3 * r + 2] * (float)cam.W);
      y = min(y, cam.H - 1);
      x = min(x, cam.W - 1);
      const float dc[3] = {((float)x + 0.5f - cam.cx) / cam.fx, -(((float)y + 0.5f - cam.cy) / cam.fy), -1.0f};
      float v[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) v[a] = Ma[4 * a] * dc[0] + Ma[4 * a + 1] * dc[1] + Ma[4 * a + 2] * dc[2];
      const float n = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);
      const float dir[3] = {v[0] / n, v[1] / n, v[2] / n};
      const float* gd = d_directions + 3 * r;
      const float dot = dir[0] * gd[0] + dir[1] * gd[1] + dir[2] * gd[2];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float gv = (gd[a] - dir[a] * dot) / n;  // backward of v / |v|
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[3 * a + b] += gv * dc[b];
        acc[9 + a] += d_origins[3 * r + a];
      }
    }
    __syncthreads();  // s_list / s_cnt are rewritten by the next round
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const float s = wave_sum(acc[i]);
    if (lane == 0) red[wave][i] = s;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  float G[9], gt[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) G[i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
#pragma unroll
  for (int a = 0; a < 3; ++a) gt[a] = (red[0][9 + a] + red[1][9 + a]) + (red[2][9 + a] + red[3][9 + a]);
  // R' = R1 R, t' = t1 + R1 t  =>  dL/dR = R1^T G,  dL/dt = R1^T gt
  const float* M = Mk;
  float GR[9], gtt[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int b = 0; b < 3; ++b) GR[3 * a + b] = M[a] * G[b] + M[4 + a] * G[3 + b] + M[8 + a] * G[6 + b];
    gtt[a] = M[a] * gt[0] + M[4 + a] * gt[1] + M[8 + a] * gt[2];
  }
  const float* w = wk;
  const SO3 s = so3_exp(w);
  // dR/dw_i = f1 K_i + f2 (K_i K + K K_i) + (df1/dw_i) K + (df2/dw_i) K^2;  d theta/d w_i = w_i / theta above the clamp
  const float th = s.theta, sn = sinf(th), cs = cosf(th);
  const float df1 = (th * cs - sn) / (th * th);                    // d(sin t / t)/dt
  const float df2 = (th * sn - 2.0f * (1.0f - cs)) / (th * th * th);  // d((1 - cos t)/t^2)/dt
  const bool above = s.theta2_raw > 1e-4f;  // clamp(nrms, 1e-4): zero gradient through theta below the threshold
  const float x = w[0], y = w[1], z = w[2];
  const float K[9] = {0.f, -z, y, z, 0.f, -x, -y, x, 0.f};
  const float K2[9] = {-(y * y + z * z), x * y, x * z, x * y, -(x * x + z * z), y * z, x * z, y * z, -(x * x + y * y)};
  float gk = 0.0f, gk2 = 0.0f;  // <GR, K>, <GR, K^2>
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    gk += GR[i] * K[i];
    gk2 += GR[i] * K2[i];
  }
  // <GR, K_i> with K_0 = skew(e_x) etc.;  <GR, K_i K + K K_i> = d<GR, K^2>/dw_i with K^2 = w w^T - |w|^2 I
  const float gK[3] = {GR[7] - GR[5], GR[2] - GR[6], GR[3] - GR[1]};
  const float tr = GR[0] + GR[4] + GR[8];
  const float Sw[3] = {(GR[0] + GR[0]) * x + (GR[1] + GR[3]) * y + (GR[2] + GR[6]) * z,
                       (GR[3] + GR[1]) * x + (GR[4] + GR[4]) * y + (GR[5] + GR[7]) * z,
                       (GR[6] + GR[2]) * x + (GR[7] + GR[5]) * y + (GR[8] + GR[8]) * z};
  float* out = pose_grad + 6 * k;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float wi = (i == 0) ? x : (i == 1) ? y : z;
    const float dth = above ? wi / th : 0.0f;
    const float g = s.f1 * gK[i] + s.f2 * (Sw[i] - 2.0f * wi * tr) + dth * (df1 * gk + df2 * gk2);
    if constexpr (!ADAM) {
      out[3 + i] = g_old[3 + i] + g;
      out[i] = g_old[i] + gtt[i];
    } else {
      float* P = reinterpret_cast<float*>(adam.p) + 6 * k;
      float* M2 = reinterpret_cast<float*>(adam.m) + 6 * k;
      float* V = reinterpret_cast<float*>(adam.v) + 6 * k;
      const float gr = g_old[3 + i] + g, gtr = g_old[i] + gtt[i];
      table_adam_update(adam, gr, Pk[3 + i], Mm[3 + i], Vk[3 + i]);
      table_adam_update(adam, gtr, Pk[i], Mm[i], Vk[i]);
      P[3 + i] = Pk[3 + i], M2[3 + i] = Mm[3 + i], V[3 + i] = Vk[3 + i];
      P[i] = Pk[i], M2[i] = Mm[i], V[i] = Vk[i];
      out[3 + i] = 0.0f;
      out[i] = 0.0f;
    }
  }
}

}  // namespace fnr

using namespace fnr;

extern "C" int fnr_camera_adjust(const float* c2w, const int64_t* train_ids, int n_train, const float* pose_adjustment,
                                 float* c2w_adjusted, void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_camera_adjust");
  FNR_CHECK_ARG(c2w && train_ids && pose_adjustment && c2w_adjusted && n_train > 0, "camera_adjust: null argument");
  hipLaunchKernelGGL(k_camera_adjust, dim3((unsigned)((n_train + 63) / 64)), dim3(64), 0, as_stream(stream), c2w,
                     reinterpret_cast<const long long*>(train_ids), n_train, pose_adjustment, c2w_adjusted);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

extern "C" int fnr_camera_pose_grad(const fnr_image_set* set, const int64_t* train_ids, int n_train, int64_t n_rays,
                                    const float* u, const int32_t* camera_indices, const float* pose_adjustment,
                                    const float* c2w_adjusted, const float* d_origins, const float* d_directions,
                                    float* pose_grad, void* stream) {
  FNR_SEQ_UNRECORDABLE("fnr_camera_pose_grad");
  FNR_CHECK_ARG(set && set->c2w && train_ids && u && camera_indices && pose_adjustment && c2w_adjusted && d_origins &&
                    d_directions && pose_grad && n_train > 0,
                "camera_pose_grad: null argument");
  if (n_rays == 0) return FNR_OK;
  PinholeDev cam{set->H, set->W, set->fx, set->fy, set->cx, set->cy};
  hipLaunchKernelGGL(k_camera_pose_grad<false>, dim3((unsigned)n_train), dim3(256), 0, as_stream(stream), cam, set->c2w,
                     reinterpret_cast<const long long*>(train_ids), (long long)n_rays, u, camera_indices, pose_adjustment,
                     c2w_adjusted, d_origins, d_directions, pose_grad, TableAdam{});
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}

extern "C" int fnr_camera_pose_grad_adam(const fnr_image_set* set, const int64_t* train_ids, int n_train, int64_t n_rays,
                                         const float* u, const int32_t* camera_indices, const float* c2w_adjusted,
                                         const float* d_origins, const float* d_directions, float* pose_grad,
                                         const fnr_table_adam* adam, void* stream) {
  if (seq::recording() && set && adam) {
    const fnr_image_set set_ = *set;
    const fnr_table_adam adam_ = *adam;
    seq::push("fnr_camera_pose_grad_adam", [=](const fnr_step_scalars* sc) {
      const fnr_table_adam a = seq::patched(adam_, sc);
      return fnr_camera_pose_grad_adam(&set_, train_ids, n_train, n_rays, u, camera_indices, c2w_adjusted, d_origins,
                                       d_directions, pose_grad, &a, stream);
    });
  }
  FNR_CHECK_ARG(set && set->c2w && train_ids && u && camera_indices && c2w_adjusted && d_origins && d_directions &&
                    pose_grad && adam && n_train > 0,
                "camera_pose_grad_adam: null argument");
  TableAdam t;
  const int rc = make_table_adam(adam, t);
  if (rc) return rc;
  PinholeDev cam{set->H, set->W, set->fx, set->fy, set->cx, set->cy};
  // n_rays == 0 still takes the step (every pose parameter decays its moments)
  hipLaunchKernelGGL(k_camera_pose_grad<true>, dim3((unsigned)n_train), dim3(256), 0, as_stream(stream), cam, set->c2w,
                     reinterpret_cast<const long long*>(train_ids), (long long)n_rays, u, camera_indices, adam->params,
                     c2w_adjusted, d_origins, d_directions, pose_grad, t);
  FNR_LAUNCH_CHECK();
  return FNR_OK;
}
