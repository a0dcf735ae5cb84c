// barrier_load_race.hip — does a workgroup barrier on gfx950 wait for outstanding VECTOR loads?  (It does not: neither
// `s_barrier` nor the workgroup-scope fence of __syncthreads() carries a vmcnt wait.)  The hazard behind round 4's rare
// divergence, in isolation, with the scatter's counter protocol:
//   set    (the emit kernel's role)  qcount[g] = g + 1, qmax[l] = 1.0f — with device-scope atomics, like the emit kernel
//   read   (accumulate_bin's role)   every thread of a 1024-thread workgroup loads its bin's count and its level's maximum
//          through the VECTOR path (an opaque zero in a VGPR; third mode: provably uniform address = scalar loads),
//          [WAIT: s_waitcnt vmcnt(0)], __syncthreads(), thread 0
//          resets the count and counts the workgroup in on qdone[l]; the level's last workgroup resets the maximum.
//          Every wave then checks what it read: a 0 where g + 1 / 1.0f was set = a reset overtook the load.
// A second stream streams a few GB through HBM meanwhile (the training step's other stream, in spirit).
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench/barrier_load_race.hip -o /tmp/barrier_load_race
// run:   /tmp/barrier_load_race [rounds = 20000]      -> stale reads without / with the wait
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int LEVELS = 5, BINS = 32, STRIDE = 32;   // one counter per 128-byte line, as in hash_scatter.hip

__global__ void k_set(unsigned* qcount, unsigned* qmax) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < LEVELS * BINS) atomicAdd(&qcount[(size_t)g * STRIDE], (unsigned)g + 1u);
  if (g < LEVELS) atomicMax(&qmax[(size_t)g * STRIDE], 0x3f800000u);
}

template <bool WAIT, bool VEC>
__global__ __launch_bounds__(1024) void k_read(unsigned* qcount, unsigned* qmax, unsigned* qdone,
                                               unsigned long long* stale /* [2]: counts, maxima */) {
  extern __shared__ unsigned long long lds[];   // 64 KiB like the proposal tables' accumulate workgroups: two per CU
  const int gbin = LEVELS * BINS - 1 - (int)blockIdx.x;
  const int lrel = gbin / BINS;
  unsigned vzero = 0u;
  if (VEC) asm volatile("" : "+v"(vzero));      // the address is not uniform for the compiler: vector loads
  // (!VEC: a provably uniform address — hipcc issues scalar loads, which the barrier's lgkmcnt(0) does wait for)
  const unsigned n = qcount[(size_t)gbin * STRIDE + vzero];
  const unsigned vmax = qmax[(size_t)lrel * STRIDE + vzero];
  if (WAIT) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::"v"(n), "v"(vmax) : "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    qcount[(size_t)gbin * STRIDE] = 0u;
    if (atomicAdd(&qdone[(size_t)lrel * STRIDE], 1u) == (unsigned)BINS - 1u) {
      qmax[(size_t)lrel * STRIDE] = 0u;
      qdone[(size_t)lrel * STRIDE] = 0u;
    }
  }
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0ull;   // (what the real kernel does next)
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    if (n != (unsigned)gbin + 1u) atomicAdd(&stale[0], 1ull);
    if (vmax != 0x3f800000u) atomicAdd(&stale[1], 1ull);
  }
}

__global__ void k_stream(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = a[i];
    v.x += 1.0f;
    b[i] = v;
  }
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 20000;
  unsigned *qcount, *qmax, *qdone;
  unsigned long long* stale;
  const size_t cbytes = (size_t)(LEVELS * BINS + 2 * LEVELS) * STRIDE * sizeof(unsigned);
  (void)hipMalloc(&qcount, cbytes);
  (void)hipMemset(qcount, 0, cbytes);
  qmax = qcount + (size_t)LEVELS * BINS * STRIDE;
  qdone = qmax + (size_t)LEVELS * STRIDE;
  (void)hipMalloc(&stale, 16);
  const size_t nload = (size_t)1 << 24;   // 256 MiB in, 256 MiB out per launch of the background stream (~0.1 ms)
  float4 *la, *lb;
  (void)hipMalloc(&la, nload * sizeof(float4));
  (void)hipMalloc(&lb, nload * sizeof(float4));
  (void)hipMemset(la, 0, nload * sizeof(float4));
  hipStream_t s0, s1;
  (void)hipStreamCreate(&s0);
  (void)hipStreamCreate(&s1);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_read<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_read<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_read<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int wait = 0; wait < 3; ++wait)     // 0: vector loads, no wait (the hazard); 1: vector loads + wait (the fix); 2: scalar loads, no wait
    for (int load = 0; load < 2; ++load) {
      (void)hipMemset(stale, 0, 16);
      (void)hipDeviceSynchronize();
      for (int r = 0; r < rounds; ++r) {
        if (load && (r % 8) == 0) hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, s1, la, lb, nload);
        hipLaunchKernelGGL(k_set, dim3(1), dim3(256), 0, s0, qcount, qmax);
        if (wait == 1) hipLaunchKernelGGL((k_read<true, true>), dim3(LEVELS * BINS), dim3(1024), 65536, s0, qcount, qmax, qdone, stale);
        else if (wait == 0) hipLaunchKernelGGL((k_read<false, true>), dim3(LEVELS * BINS), dim3(1024), 65536, s0, qcount, qmax, qdone, stale);
        else hipLaunchKernelGGL((k_read<false, false>), dim3(LEVELS * BINS), dim3(1024), 65536, s0, qcount, qmax, qdone, stale);
      }
      (void)hipDeviceSynchronize();
      unsigned long long h[2];
      (void)hipMemcpy(h, stale, 16, hipMemcpyDeviceToHost);
      printf("%-34s second stream streaming: %-3s  %d rounds x %d workgroups x 16 waves: stale counts %llu, stale maxima %llu\n",
             wait == 0 ? "vector loads, no wait (hazard):" : wait == 1 ? "vector loads + wait (the fix):" : "scalar loads, no explicit wait:",
             load ? "yes" : "no", rounds, LEVELS * BINS, h[0], h[1]);
      fflush(stdout);
    }
  return 0;
}
