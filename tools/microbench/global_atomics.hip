#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
// Microbenchmark: random fp32 atomic adds into a table of `tsize` floats, agent scope vs workgroup scope,
// plus an XCC_ID census.
__device__ __forceinline__ uint32_t hash32(uint32_t x){ x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template<int SCOPE> __global__ void k_atomic(float* t, uint32_t mask, int per_thread, int xcd_local) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t base = 0;
  if (xcd_local) { // each XCD (observed b%8) gets its own disjoint region
    base = (blockIdx.x & 7) * (mask + 1);
  }
  for (int i = 0; i < per_thread; ++i) {
    uint32_t idx = base + (hash32(tid * 131u + i) & mask);
    if (SCOPE == 0) __hip_atomic_fetch_add(t + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(t + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}
__global__ void k_census(int* out) {
  if (threadIdx.x == 0) { int x = __builtin_amdgcn_s_getreg((20 | (0 << 6) | (3 << 11))); out[blockIdx.x] = x & 0xf; }
}
int main() {
  float* t; size_t maxn = (size_t)8 * (64u << 20) / 4; hipMalloc(&t, maxn * 4); hipMemset(t, 0, maxn*4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  int blocks = 256 * 16, threads = 256, per = 64;
  for (int xl = 0; xl < 2; ++xl)
  for (uint32_t logn = 14; logn <= 24; logn += 2) {
    uint32_t mask = (1u << logn) - 1;
    for (int scope = 0; scope < 2; ++scope) {
      float best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        if (scope == 0) hipLaunchKernelGGL(k_atomic<0>, dim3(blocks), dim3(threads), 0, 0, t, mask, per, xl);
        else hipLaunchKernelGGL(k_atomic<1>, dim3(blocks), dim3(threads), 0, 0, t, mask, per, xl);
        hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
      }
      double n = (double)blocks * threads * per;
      printf("xcd_local=%d table=%7.2f MB scope=%s : %.3f ms  %.1f G atomics/s\n", xl, (mask+1)*4.0/1048576.0, scope==0?"agent":"wg   ", best, n / best / 1e6);
    }
  }
  // verify sums for wg-scope xcd_local run
  int* cen; hipMalloc(&cen, 64*4); hipLaunchKernelGGL(k_census, dim3(64), dim3(64), 0, 0, cen); int h[64]; hipMemcpy(h, cen, 256, hipMemcpyDeviceToHost);
  printf("xcc ids of blocks 0..15:"); for (int i=0;i<16;++i) printf(" %d", h[i]); printf("\n");
  // correctness: wg-scope, xcd_local, count total
  hipMemset(t, 0, maxn*4); uint32_t mask = (1u<<20)-1;
  hipLaunchKernelGGL(k_atomic<1>, dim3(blocks), dim3(threads), 0, 0, t, mask, per, 1); hipDeviceSynchronize();
  size_t n = (size_t)8 * (mask+1); float* hbuf = (float*)malloc(n*4); hipMemcpy(hbuf, t, n*4, hipMemcpyDeviceToHost);
  double s = 0; for (size_t i=0;i<n;++i) s += hbuf[i];
  printf("wg-scope xcd_local sum %.0f expected %.0f\n", s, (double)blocks*threads*per);
  hipMemset(t, 0, maxn*4);
  hipLaunchKernelGGL(k_atomic<1>, dim3(blocks), dim3(threads), 0, 0, t, mask, per, 0); hipDeviceSynchronize();
  hipMemcpy(hbuf, t, (mask+1)*4, hipMemcpyDeviceToHost); s = 0; for (size_t i=0;i<=mask;++i) s += hbuf[i];
  printf("wg-scope SHARED (all XCDs same region) sum %.0f expected %.0f\n", s, (double)blocks*threads*per);
  return 0;
}
