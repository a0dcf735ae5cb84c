// nt_visibility.hip — does a streaming (`nt`) load in kernel K2 always see what an EARLIER kernel K1 of the same stream
// stored (plain or `nt`) — on a part whose eight XCDs have private L2s?
//
// Second half of round 6's hunt for what `nt` loads of the encode's Jacobian do to k_field_mlp_bwd_base_coop
// (tools/microbench/nt_load_order.hip is the first half: completion ORDER is not the problem).  The probe run
// (tests/diagnostics/nt_jac_probe.py, profiles/r06_raw/nt_jac_probe.log) shows the camera poses — fed by the position
// gradient, i.e. by the Jacobian values the kernel loads — diverging first, and from step 0 on when the kernel waits for the
// loads right away: the signature of loads that return the buffer's PREVIOUS contents.  The pattern, in isolation:
//   K1  block b stores pattern(i, tag) to chunk b of a buffer         (policy of the stores: plain | nt)
//   [an unrelated kernel in between, as in the training step]
//   K2  block b loads chunk perm(b) — written by ANOTHER workgroup, most likely on another XCD — and compares
//                                                                      (policy of the loads: plain | nt)
// repeated with tag = 1, 2, ... on the SAME buffer.  A mismatch that equals pattern(i, tag - 1) is a STALE read.
//
// build: hipcc -O2 --offload-arch=gfx950 tools/microbench/nt_visibility.hip -o /tmp/nt_visibility
// run:   /tmp/nt_visibility [tags = 300]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float nt_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float2 pat(size_t i, unsigned tag) {
  const unsigned a = (unsigned)(i * 2654435761ull) ^ (tag * 0x9e3779b9u), b = (unsigned)(i * 40503ull) + tag;
  return make_float2((float)(a & 0xffffff), (float)(b & 0xffffff));
}

template <bool NT>
__global__ __launch_bounds__(256) void k_write(float2* __restrict__ buf, size_t per_block, unsigned tag) {
  const size_t base = (size_t)blockIdx.x * per_block;
  for (size_t k = threadIdx.x; k < per_block; k += 256) {
    const float2 v = pat(base + k, tag);
    if (NT) {
      const nt_f2 t = {v.x, v.y};
      __builtin_nontemporal_store(t, reinterpret_cast<nt_f2*>(buf + base + k));
    } else {
      buf[base + k] = v;
    }
  }
}

template <bool NT>
__global__ __launch_bounds__(256) void k_check(const float2* __restrict__ buf, size_t per_block, unsigned tag,
                                               unsigned long long* __restrict__ bad /* [2]: stale (previous tag), other */) {
  const size_t chunk = ((size_t)blockIdx.x * 37 + 11) % gridDim.x;   // another workgroup's chunk
  const size_t base = chunk * per_block;
  unsigned long long stale = 0, other = 0;
  for (size_t k = threadIdx.x; k < per_block; k += 256) {
    float2 v;
    if (NT) {
      const nt_f2 t = __builtin_nontemporal_load(reinterpret_cast<const nt_f2*>(buf + base + k));
      v = make_float2(t.x, t.y);
    } else {
      v = buf[base + k];
    }
    const float2 want = pat(base + k, tag);
    if (v.x != want.x || v.y != want.y) {
      const float2 prev = pat(base + k, tag - 1);
      if (v.x == prev.x && v.y == prev.y) ++stale;
      else ++other;
    }
  }
  if (stale) atomicAdd(&bad[0], stale);
  if (other) atomicAdd(&bad[1], other);
}

__global__ void k_stream(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = a[i];
    v.x += 1.0f;
    b[i] = v;
  }
}

int main(int argc, char** argv) {
  const int tags = argc > 1 ? atoi(argv[1]) : 300;
  const int blocks = 4096;
  const size_t per_block = 2304;                       // 75 MB of float2, the Jacobian's size
  float2* buf;
  unsigned long long* bad;
  (void)hipMalloc(&buf, (size_t)blocks * per_block * sizeof(float2));
  (void)hipMalloc(&bad, 16);
  const size_t nload = (size_t)1 << 23;
  float4 *la, *lb;
  (void)hipMalloc(&la, nload * sizeof(float4));
  (void)hipMalloc(&lb, nload * sizeof(float4));
  (void)hipMemset(la, 0, nload * sizeof(float4));
  hipStream_t s0, s1;
  (void)hipStreamCreate(&s0);
  (void)hipStreamCreate(&s1);
  for (int st = 0; st < 2; ++st)
    for (int ld = 0; ld < 2; ++ld)
      for (int between = 0; between < 2; ++between)
        for (int load = 0; load < 2; ++load) {
          (void)hipMemset(bad, 0, 16);
          (void)hipMemset(buf, 0, (size_t)blocks * per_block * sizeof(float2));
          (void)hipDeviceSynchronize();
          for (int tag = 1; tag <= tags; ++tag) {
            if (load && (tag % 4) == 0) hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, s1, la, lb, nload);
            if (st) hipLaunchKernelGGL(k_write<true>, dim3(blocks), dim3(256), 0, s0, buf, per_block, (unsigned)tag);
            else hipLaunchKernelGGL(k_write<false>, dim3(blocks), dim3(256), 0, s0, buf, per_block, (unsigned)tag);
            if (between) hipLaunchKernelGGL(k_stream, dim3(512), dim3(256), 0, s0, la, lb, nload / 8);
            if (ld) hipLaunchKernelGGL(k_check<true>, dim3(blocks), dim3(256), 0, s0, buf, per_block, (unsigned)tag, bad);
            else hipLaunchKernelGGL(k_check<false>, dim3(blocks), dim3(256), 0, s0, buf, per_block, (unsigned)tag, bad);
          }
          const hipError_t err = hipDeviceSynchronize();
          if (err != hipSuccess) return printf("HIP error: %s\n", hipGetErrorString(err)), 1;
          unsigned long long h[2];
          (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
          printf("stores %-5s loads %-5s kernel in between: %-3s second stream: %-3s  %d rewrites x %.1f M values: stale %llu, other %llu\n",
                 st ? "nt" : "plain", ld ? "nt" : "plain", between ? "yes" : "no", load ? "yes" : "no", tags,
                 blocks * (double)per_block / 1e6, h[0], h[1]);
          fflush(stdout);
        }
  return 0;
}
