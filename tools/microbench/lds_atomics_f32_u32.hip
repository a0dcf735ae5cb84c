#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ uint32_t hash32(uint32_t x){ x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
// MODE 0: ds_add_f32 random, 1: ds_add_u32 random, 2: ds_write_b32 random, 3: ds_add_f32 sequential (lane-linear), 4: ds_add_f32 x2 adjacent (row,row+1) like float2
template<int MODE> __global__ __launch_bounds__(1024) void k(float* out, int iters) {
  __shared__ float s[16384];
  for (int i = threadIdx.x; i < 16384; i += 1024) s[i] = 0.f;
  __syncthreads();
  uint32_t* su = (uint32_t*)s;
  uint32_t seed = blockIdx.x * 1024 + threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    uint32_t idx = hash32(seed + i * 7919u) & 16383u;
    if (MODE == 0) atomicAdd(&s[idx], 1.0f);
    else if (MODE == 1) atomicAdd(&su[idx], 1u);
    else if (MODE == 2) s[idx] = (float)i;
    else if (MODE == 3) atomicAdd(&s[(threadIdx.x + i * 1024) & 16383], 1.0f);
    else { idx &= ~1u; atomicAdd(&s[idx], 1.0f); atomicAdd(&s[idx + 1], 2.0f); }
  }
  __syncthreads();
  float acc = 0; for (int i = threadIdx.x; i < 16384; i += 1024) acc += s[i];
  out[blockIdx.x * 1024 + threadIdx.x] = acc;
}
template<int MODE> void run(const char* name, float* out) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  int blocks = 512, iters = 256; float best = 1e9;
  for (int r = 0; r < 3; ++r) { hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, out, iters); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
  double n = (double)blocks * 1024 * iters * (MODE == 4 ? 2 : 1);
  printf("%-32s %.3f ms  %.1f G lane-ops/s (%.2f per clk per CU @2.4GHz,256CU)\n", name, best, n / best / 1e6, n / best / 1e6 / 256 / 2.4);
}
int main() { float* out; hipMalloc(&out, 512 * 1024 * 4);
  run<0>("ds_add_f32 random", out); run<1>("ds_add_u32 random", out); run<2>("ds_write_b32 random", out); run<3>("ds_add_f32 lane-linear", out); run<4>("ds_add_f32 pair(row,row+1)", out);
  return 0; }
