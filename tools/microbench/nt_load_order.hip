// nt_load_order.hip — may a streaming (`nt`) global load and a plain one complete OUT OF ISSUE ORDER on gfx950?
//
// Why it matters (round 5, DESIGN 4 "Round 5"): with `nt` loads of the encode's Jacobian in
// k_field_mlp_bwd_base_coop, training lost its run-to-run bit-reproducibility in 4 of 4 runs.  Those were the only streaming
// loads of the library in flight TOGETHER WITH plain loads and stores and consumed behind PARTIAL waits
// (`s_waitcnt vmcnt(n)`, n > 0).  hipcc's wait counts assume that vector-memory operations of a wave complete in issue
// order (vmcnt is ONE counter on gfx9-family parts: it can only say "all but the youngest n are done").  If an `nt` access —
// which takes another cache policy through the TCP / TCC — could overtake or be overtaken, a partial wait would release a
// register whose load is still in flight: WRONG DATA, not just reordered sums.
//
// The test, per lane and round (everything in ONE inline-asm block, so the compiler inserts no waits of its own):
//     dest registers <- sentinel
//     load A (slow: a line of a multi-GiB buffer, a different one every round -> HBM / TLB miss)
//     [store S (plain) in between, mode *_st]
//     load B (fast: a line the wave has just touched -> L1 / L2 hit)
//     s_waitcnt vmcnt(1 | 0 more)            <- "everything but the youngest is done": A must have arrived
//     out_a <- dest A                        <- the sentinel here = A was overtaken by B and the wait let us through
//     s_waitcnt vmcnt(0) ; out_b <- dest B
// in all four policy combinations (A / B plain or nt), with and without a second stream that streams GiBs through HBM.
// A mismatch of out_a against the buffer's known pattern is counted as "released early".
//
// build: hipcc -O2 --offload-arch=gfx950 tools/microbench/nt_load_order.hip -o /tmp/nt_load_order
// run:   /tmp/nt_load_order [rounds = 400]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr unsigned SENTINEL = 0xdeadbeefu;

__device__ __forceinline__ unsigned pattern(size_t word) { return (unsigned)(word * 2654435761ull) ^ 0x5bd1e995u; }

__global__ void k_fill(unsigned* buf, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = pattern(i);
}

// The policy suffix has to be part of the instruction text: one explicit kernel per combination.
#define ORDER_KERNEL_W(NAME, SFX_A, STORE, SFX_B, WAITN)                                                                       \
  __global__ __launch_bounds__(256) void NAME(const unsigned* __restrict__ big, size_t big_words,                      \
                                              const unsigned* __restrict__ hot, unsigned* __restrict__ sink, int rounds, \
                                              unsigned seed, unsigned long long* __restrict__ bad) {                   \
    const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;                                                  \
    unsigned long long bad_a = 0, bad_b = 0;                                                                           \
    for (int r = 0; r < rounds; ++r) {                                                                                 \
      const size_t line = ((tid * 0x9e3779b97f4a7c15ull) ^ ((size_t)(seed + r) * 0xbf58476d1ce4e5b9ull)) % (big_words / 32); \
      const size_t wa = line * 32 + (tid & 31);                                                                        \
      const size_t wb = (tid * 4 + (r & 3)) & 1023;                                                                    \
      const unsigned* pa = big + wa;                                                                                   \
      const unsigned* pb = hot + wb;                                                                                   \
      unsigned* ps = sink + tid;                                                                                       \
      unsigned ra, rb, oa, ob;                                                                                         \
      asm volatile(                                                                                                    \
          "v_mov_b32 %0, %6\n\t"                                                                                       \
          "v_mov_b32 %1, %6\n\t"                                                                                       \
          "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"                                                                          \
          "global_load_dword %0, %4, off" SFX_A "\n\t" STORE                                                           \
          "global_load_dword %1, %5, off" SFX_B "\n\t"                                                                 \
          "s_waitcnt vmcnt(" WAITN ")\n\t"                                                                              \
          "v_mov_b32 %2, %0\n\t"                                                                                       \
          "s_waitcnt vmcnt(0)\n\t"                                                                                     \
          "v_mov_b32 %3, %1\n\t"                                                                                       \
          : "=&v"(ra), "=&v"(rb), "=&v"(oa), "=&v"(ob)                                                                 \
          : "v"(pa), "v"(pb), "v"(SENTINEL), "v"(ps)                                                                   \
          : "memory");                                                                                                 \
      (void)ra;                                                                                                        \
      (void)rb;                                                                                                        \
      if (oa != pattern(wa)) ++bad_a;                                                                                  \
      if (ob != pattern(wb)) ++bad_b;                                                                                  \
    }                                                                                                                  \
    if (bad_a) atomicAdd(&bad[0], bad_a);                                                                              \
    if (bad_b) atomicAdd(&bad[1], bad_b);                                                                              \
  }

#define ORDER_KERNEL(NAME, SFX_A, STORE, SFX_B) ORDER_KERNEL_W(NAME, SFX_A, STORE, SFX_B, "1")
// In the *_st kernels a plain store sits between the loads: three operations in flight, the partial wait is vmcnt(1) all the
// same ("all but the youngest"), so BOTH A and the store must be complete when it falls through.
ORDER_KERNEL(k_plain_plain, "", "", "")
// positive control of the detector: the same block with `vmcnt(2)` — a wait that does NOT cover A — must see sentinels
ORDER_KERNEL_W(k_control_no_wait, " nt", "", "", "2")
ORDER_KERNEL(k_nt_plain, " nt", "", "")
ORDER_KERNEL(k_plain_nt, "", "", " nt")
ORDER_KERNEL(k_nt_nt, " nt", "", " nt")
ORDER_KERNEL(k_nt_st_plain, " nt", "global_store_dword %7, %6, off\n\t", "")
ORDER_KERNEL(k_plain_st_nt, "", "global_store_dword %7, %6, off\n\t", " nt")
ORDER_KERNEL(k_nt_ntst_plain, " nt", "global_store_dword %7, %6, off nt\n\t", "")

// The failing kernel's own pattern (k_field_mlp_bwd_base_coop with the Jacobian's loads streaming, ISA of round 6): twelve nt
// loads, then — under an exec mask — four PLAIN stores, then `s_waitcnt vmcnt(9)`: "the oldest loads are done whether or not the
// stores were issued".  Here: three slow loads (policy SFX), four stores to lines the lane owns, `vmcnt(4)` = exactly the
// stores may still be pending, all three loads are consumed.
#define LOADS_THEN_STORES_KERNEL(NAME, SFX)                                                                             \
  __global__ __launch_bounds__(256) void NAME(const unsigned* __restrict__ big, size_t big_words,                      \
                                              const unsigned* __restrict__ hot, unsigned* __restrict__ sink, int rounds, \
                                              unsigned seed, unsigned long long* __restrict__ bad) {                   \
    const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;                                                  \
    unsigned long long bad_a = 0;                                                                                      \
    (void)hot;                                                                                                         \
    for (int r = 0; r < rounds; ++r) {                                                                                 \
      size_t w[3];                                                                                                     \
      for (int q = 0; q < 3; ++q) {                                                                                    \
        const size_t line = ((tid * 0x9e3779b97f4a7c15ull) ^ ((size_t)(seed + 3 * r + q) * 0xbf58476d1ce4e5b9ull)) % (big_words / 32); \
        w[q] = line * 32 + (tid & 31);                                                                                 \
      }                                                                                                                \
      const unsigned *p0 = big + w[0], *p1 = big + w[1], *p2 = big + w[2];                                             \
      unsigned* ps = sink + tid;                                                                                       \
      unsigned r0, r1, r2, o0, o1, o2;                                                                                 \
      asm volatile(                                                                                                    \
          "v_mov_b32 %0, %9\n\t"                                                                                       \
          "v_mov_b32 %1, %9\n\t"                                                                                       \
          "v_mov_b32 %2, %9\n\t"                                                                                       \
          "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"                                                                          \
          "global_load_dword %0, %6, off" SFX "\n\t"                                                                   \
          "global_load_dword %1, %7, off" SFX "\n\t"                                                                   \
          "global_load_dword %2, %8, off" SFX "\n\t"                                                                   \
          "global_store_dword %10, %9, off\n\t"                                                                        \
          "global_store_dword %10, %9, off offset:1024\n\t"                                                            \
          "global_store_dword %10, %9, off offset:2048\n\t"                                                            \
          "global_store_dword %10, %9, off offset:3072\n\t"                                                            \
          "s_waitcnt vmcnt(4)\n\t"                                                                                     \
          "v_mov_b32 %3, %0\n\t"                                                                                       \
          "v_mov_b32 %4, %1\n\t"                                                                                       \
          "v_mov_b32 %5, %2\n\t"                                                                                       \
          "s_waitcnt vmcnt(0)\n\t"                                                                                     \
          : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(o0), "=&v"(o1), "=&v"(o2)                                           \
          : "v"(p0), "v"(p1), "v"(p2), "v"(SENTINEL), "v"(ps)                                                          \
          : "memory");                                                                                                 \
      (void)r0;                                                                                                        \
      (void)r1;                                                                                                        \
      (void)r2;                                                                                                        \
      bad_a += (o0 != pattern(w[0])) + (o1 != pattern(w[1])) + (o2 != pattern(w[2]));                                  \
    }                                                                                                                  \
    if (bad_a) atomicAdd(&bad[0], bad_a);                                                                              \
  }
LOADS_THEN_STORES_KERNEL(k_plain3_st4, "")
LOADS_THEN_STORES_KERNEL(k_nt3_st4, " nt")

// The Jacobian's access shape: twelve 8-byte loads per lane in one burst — lane (g = lane / 16, j = lane % 16) reads element
// n of stream 3 (4 m + g) + a, the streams N elements apart — then `s_waitcnt vmcnt(0)` and all 24 registers are consumed.
#define BURST_KERNEL(NAME, SFX)                                                                                         \
  __global__ __launch_bounds__(512) void NAME(const unsigned* __restrict__ big, size_t big_words,                      \
                                              const unsigned* __restrict__ hot, unsigned* __restrict__ sink, int rounds, \
                                              unsigned seed, unsigned long long* __restrict__ bad) {                   \
    (void)hot;                                                                                                         \
    (void)sink;                                                                                                        \
    const size_t N = big_words / 2 / 48;             /* 48 streams of N 8-byte elements */                             \
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;                                                   \
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;                                          \
    unsigned long long bad_a = 0;                                                                                      \
    for (int r = 0; r < rounds; ++r) {                                                                                 \
      const size_t n = (((wave * 0x9e3779b97f4a7c15ull) ^ ((size_t)(seed + r) * 0xbf58476d1ce4e5b9ull)) % (N / 16)) * 16 + j; \
      const unsigned* p[12];                                                                                           \
      for (int m = 0; m < 4; ++m)                                                                                      \
        for (int a = 0; a < 3; ++a) p[3 * m + a] = big + 2 * (((size_t)(4 * m + g) * 3 + a) * N + n);                 \
      unsigned long long v[12];                                                                                        \
      asm volatile(                                                                                                    \
          "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"                                                                          \
          "global_load_dwordx2 %0, %12, off" SFX "\n\t"                                                                \
          "global_load_dwordx2 %1, %13, off" SFX "\n\t"                                                                \
          "global_load_dwordx2 %2, %14, off" SFX "\n\t"                                                                \
          "global_load_dwordx2 %3, %15, off" SFX "\n\t"                                                                \
          "global_load_dwordx2 %4, %16, off" SFX "\n\t"                                                                \
          "global_load_dwordx2 %5, %17, off" SFX "\n\t"                                                                \
          "global_load_dwordx2 %6, %18, off" SFX "\n\t"                                                                \
          "global_load_dwordx2 %7, %19, off" SFX "\n\t"                                                                \
          "global_load_dwordx2 %8, %20, off" SFX "\n\t"                                                                \
          "global_load_dwordx2 %9, %21, off" SFX "\n\t"                                                                \
          "global_load_dwordx2 %10, %22, off" SFX "\n\t"                                                               \
          "global_load_dwordx2 %11, %23, off" SFX "\n\t"                                                               \
          "s_waitcnt vmcnt(0)\n\t"                                                                                     \
          : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]),    \
            "=&v"(v[8]), "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11])                                                       \
          : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]), "v"(p[8]),         \
            "v"(p[9]), "v"(p[10]), "v"(p[11])                                                                          \
          : "memory");                                                                                                 \
      for (int q = 0; q < 12; ++q) {                                                                                   \
        const size_t w = (size_t)(p[q] - big);                                                                         \
        const unsigned long long want = (unsigned long long)pattern(w) | ((unsigned long long)pattern(w + 1) << 32);   \
        bad_a += v[q] != want;                                                                                         \
      }                                                                                                                \
    }                                                                                                                  \
    if (bad_a) atomicAdd(&bad[0], bad_a);                                                                              \
  }
// What hipcc does in the real kernel (ISA of round 6): the DESTINATION of a later load is the ADDRESS register pair of an
// earlier one (`global_load_dwordx2 v[60:61], v[52:53], off nt ; s_nop 0 ; global_load_dwordx2 v[52:53], v[26:27], off nt`).
// Safe if a load has read its address when it issues.  Here: A (slow, miss) is addressed through registers that B (fast, hit)
// then loads INTO; A's data must still be A's.
#define ADDR_REUSE_KERNEL(NAME, SFX)                                                                                    \
  __global__ __launch_bounds__(256) void NAME(const unsigned* __restrict__ big, size_t big_words,                      \
                                              const unsigned* __restrict__ hot, unsigned* __restrict__ sink, int rounds, \
                                              unsigned seed, unsigned long long* __restrict__ bad) {                   \
    (void)sink;                                                                                                        \
    const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;                                                  \
    unsigned long long bad_a = 0, bad_b = 0;                                                                           \
    for (int r = 0; r < rounds; ++r) {                                                                                 \
      const size_t line = ((tid * 0x9e3779b97f4a7c15ull) ^ ((size_t)(seed + r) * 0xbf58476d1ce4e5b9ull)) % (big_words / 32); \
      const size_t wa = line * 32 + 2 * (tid & 15);                                                                    \
      const size_t wb = (tid * 2 + 4 * (r & 3)) & 1022;                                                                \
      unsigned long long a_then_b = (unsigned long long)(big + wa);     /* A's address, then B's data */               \
      const unsigned* pb = hot + wb;                                                                                   \
      unsigned long long da;                                                                                           \
      asm volatile(                                                                                                    \
          "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"                                                                          \
          "global_load_dwordx2 %0, %1, off" SFX "\n\t"                                                                 \
          "s_nop 0\n\t"                                                                                                \
          "global_load_dwordx2 %1, %2, off" SFX "\n\t"                                                                 \
          "s_waitcnt vmcnt(0)\n\t"                                                                                     \
          : "=&v"(da), "+v"(a_then_b)                                                                                  \
          : "v"(pb)                                                                                                    \
          : "memory");                                                                                                 \
      const unsigned long long want_a = (unsigned long long)pattern(wa) | ((unsigned long long)pattern(wa + 1) << 32); \
      const unsigned long long want_b = (unsigned long long)pattern(wb) | ((unsigned long long)pattern(wb + 1) << 32); \
      bad_a += da != want_a;                                                                                           \
      bad_b += a_then_b != want_b;                                                                                     \
    }                                                                                                                  \
    if (bad_a) atomicAdd(&bad[0], bad_a);                                                                              \
    if (bad_b) atomicAdd(&bad[1], bad_b);                                                                              \
  }
ADDR_REUSE_KERNEL(k_addr_reuse_plain, "")
ADDR_REUSE_KERNEL(k_addr_reuse_nt, " nt")

BURST_KERNEL(k_burst_plain, "")
BURST_KERNEL(k_burst_nt, " nt")

__global__ void k_stream(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = a[i];
    v.x += 1.0f;
    b[i] = v;
  }
}

typedef void (*order_fn)(const unsigned*, size_t, const unsigned*, unsigned*, int, unsigned, unsigned long long*);

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 400;
  const size_t big_words = (size_t)1 << 30;   // 4 GiB of pattern words: every A load is a fresh line
  unsigned *big, *hot, *sink;
  unsigned long long* bad;
  if (hipMalloc(&big, big_words * 4) != hipSuccess) return printf("hipMalloc 4 GiB failed\n"), 1;
  (void)hipMalloc(&hot, 4096);
  (void)hipMalloc(&sink, (size_t)2048 * 256 * 4 + 4096);
  (void)hipMalloc(&bad, 16);
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, big, big_words);
  hipLaunchKernelGGL(k_fill, dim3(4), dim3(256), 0, 0, hot, (size_t)1024);
  const size_t nload = (size_t)1 << 24;
  float4 *la, *lb;
  (void)hipMalloc(&la, nload * sizeof(float4));
  (void)hipMalloc(&lb, nload * sizeof(float4));
  (void)hipMemset(la, 0, nload * sizeof(float4));
  hipStream_t s0, s1;
  (void)hipStreamCreate(&s0);
  (void)hipStreamCreate(&s1);
  (void)hipDeviceSynchronize();
  const struct { const char* name; order_fn fn; } modes[] = {
      {"DETECTOR CHECK: vmcnt(2), A not waited", k_control_no_wait}, {"A plain, B plain (control)", k_plain_plain}, {"A nt,    B plain", k_nt_plain}, {"A plain, B nt", k_plain_nt},
      {"A nt,    B nt", k_nt_nt}, {"A nt, store, B plain", k_nt_st_plain}, {"A plain, store, B nt", k_plain_st_nt},
      {"A nt, nt store, B plain", k_nt_ntst_plain},
      {"3 plain loads, 4 stores, vmcnt(4)", k_plain3_st4}, {"3 nt loads, 4 stores, vmcnt(4)", k_nt3_st4},
      {"burst of 12 plain dwordx2, vmcnt(0)", k_burst_plain}, {"burst of 12 nt dwordx2, vmcnt(0)", k_burst_nt},
      {"B loads INTO A's address regs, plain", k_addr_reuse_plain}, {"B loads INTO A's address regs, nt", k_addr_reuse_nt}};
  const int launches = 40;
  for (const auto& m : modes)
    for (int load = 0; load < 2; ++load) {
      (void)hipMemset(bad, 0, 16);
      (void)hipDeviceSynchronize();
      for (int l = 0; l < launches; ++l) {
        if (load) hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, s1, la, lb, nload);
        hipLaunchKernelGGL(m.fn, dim3(2048), dim3(256), 0, s0, big, big_words, hot, sink, rounds, (unsigned)(l * 7919 + 13), bad);
      }
      const hipError_t err = hipDeviceSynchronize();
      if (err != hipSuccess || hipGetLastError() != hipSuccess) return printf("HIP error: %s\n", hipGetErrorString(err)), 1;
      unsigned long long h[2];
      (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
      printf("%-40s second stream streaming: %-3s  %llu partial waits: A released early %llu, B wrong %llu\n", m.name,
             load ? "yes" : "no", (unsigned long long)launches * 2048ull * 256ull * (unsigned long long)rounds, h[0], h[1]);
      fflush(stdout);
    }
  return 0;
}
