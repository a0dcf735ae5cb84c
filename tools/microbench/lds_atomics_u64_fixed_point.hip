#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ uint32_t hash32(uint32_t x){ x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template<int MODE> __global__ __launch_bounds__(1024) void k(float* out, int iters) {
  __shared__ unsigned long long s[8192 * 2];  // 128 KB
  for (int i = threadIdx.x; i < 16384; i += 1024) s[i] = 0;
  __syncthreads();
  uint32_t seed = blockIdx.x * 1024 + threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    uint32_t idx = hash32(seed + i * 7919u) & 8191u;
    if (MODE == 0) { atomicAdd(&s[2*idx], (unsigned long long)(i+1)); atomicAdd(&s[2*idx+1], (unsigned long long)(2*i+1)); }
    else if (MODE == 1) { atomicAdd(&s[idx], (unsigned long long)(i+1)); }
    else { unsigned* su = (unsigned*)s; atomicAdd(&su[4*idx], (unsigned)i); atomicAdd(&su[4*idx+1], 1u); atomicAdd(&su[4*idx+2], (unsigned)i); atomicAdd(&su[4*idx+3], 1u);}
  }
  __syncthreads();
  float acc = 0; for (int i = threadIdx.x; i < 16384; i += 1024) acc += (float)s[i];
  out[blockIdx.x * 1024 + threadIdx.x] = acc;
}
template<int MODE> void run(const char* name, float* out, int per) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  int blocks = 512, iters = 256; float best = 1e9;
  for (int r = 0; r < 3; ++r) { hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, out, iters); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
  double n = (double)blocks * 1024 * iters;
  printf("%-40s %.3f ms  %.1f G records/s (%.2f records per clk per CU)\n", name, best, n / best / 1e6, n / best / 1e6 / 256 / 2.4);
}
int main() { float* out; hipMalloc(&out, 512 * 1024 * 4);
  run<0>("2x ds_add_u64 per record (random row)", out, 2); run<1>("1x ds_add_u64 per record", out, 1); run<2>("4x ds_add_u32 per record", out, 4);
  return 0; }
