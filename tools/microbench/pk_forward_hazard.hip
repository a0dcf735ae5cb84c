// pk_forward_hazard.hip — is ONE ds_bpermute_b32 enough of a wait state between a packed-FP32 VALU result and the VALU
// instruction that reads it, on gfx950?
//
// The instruction sequence round 6's hunt ended at (k_field_mlp_bwd_base_coop as hipcc schedules it once the Jacobian's loads
// are `nt`; profiles/r06_raw/nt_hunt.md):
//     v_pk_add_f32 v[22:23], v[24:25], v[22:23] op_sel:[0,1] op_sel_hi:[1,0]     ; P: packed add, cross-half operand selects
//     ds_bpermute_b32 v25, v35, v19                                              ; one LDS-crossbar instruction
//     v_pk_add_f32 v[20:21], v[20:21], v[22:23]                                  ; C: reads P's result
// ~10 of 12 288 waves per launch ended with a wrong LOW half of C — only waves that reach the sequence while the workgroup's
// other waves are not using the LDS pipe — and any source change that makes hipcc pick another schedule removes it.  hipcc
// puts `s_nop 0` between P and C when they are adjacent; with the DS instruction between them it adds nothing.  This file runs
// exactly that sequence in inline assembly, with 0 / 1 / 2 extra wait states, against the same arithmetic done with scalar
// instructions, while half of the workgroup's waves keep the LDS pipe busy or idle.
//
// build: hipcc -O2 --offload-arch=gfx950 tools/microbench/pk_forward_hazard.hip -o /tmp/pk_forward_hazard
// run:   /tmp/pk_forward_hazard [rounds = 2000]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ float mix(unsigned x) {   // a float in [1, 2) from a hash: sums stay exact enough to compare bitwise
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return __uint_as_float(0x3f800000u | (x & 0x7fffffu));
}

// NOPS: extra wait states after the ds_bpermute (-1: no ds_bpermute at all, P and C adjacent but for an s_nop 0 — what hipcc emits)
template <int NOPS, bool LDS_BUSY>
__global__ __launch_bounds__(512) void k_seq(int rounds, unsigned seed, unsigned long long* __restrict__ bad) {
  __shared__ float s_buf[8 * 64 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long bad_lo = 0, bad_hi = 0;
  float keep = 0.0f;
  for (int r = 0; r < rounds; ++r) {
    const unsigned h = (blockIdx.x * 512u + threadIdx.x) * 2654435761u ^ (seed + r) * 0x9e3779b9u;
    if (wave >= 4) {   // the other half of the workgroup: LDS traffic or nothing (the failing waves were always waves 0..3)
      if (LDS_BUSY) {
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
          s_buf[(wave * 64 + lane) * 4 + (k & 3)] = keep + (float)k;
          keep += s_buf[(wave * 64 + ((lane + k) & 63)) * 4 + ((k + 1) & 3)];
        }
      }
      continue;
    }
    // P = (a0, a1) (+) (b0, b1) with op_sel:[0,1] op_sel_hi:[1,0]:  P.lo = a.lo + b.hi,  P.hi = a.hi + b.lo
    // C = (c0, c1) + P
    const float a0 = mix(h), a1 = mix(h + 1), b0 = mix(h + 2), b1 = mix(h + 3), c0 = mix(h + 4), c1 = mix(h + 5), z = mix(h + 6);
    const float want_lo = c0 + (a0 + b1), want_hi = c1 + (a1 + b0);
    float2 A = make_float2(a0, a1), B = make_float2(b0, b1), Cc = make_float2(c0, c1);
    float zs;
    const int addr = ((lane ^ 16) & 63) << 2;
    if (NOPS < 0) {
      asm volatile(
          "v_pk_add_f32 %1, %3, %1 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
          "s_nop 0\n\t"
          "v_pk_add_f32 %0, %0, %1\n\t"
          "ds_bpermute_b32 %2, %4, %5\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          : "+v"(Cc), "+v"(B), "=&v"(zs)
          : "v"(A), "v"(addr), "v"(z));
    } else if (NOPS == 0) {
      asm volatile(
          "v_pk_add_f32 %1, %3, %1 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
          "ds_bpermute_b32 %2, %4, %5\n\t"
          "v_pk_add_f32 %0, %0, %1\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          : "+v"(Cc), "+v"(B), "=&v"(zs)
          : "v"(A), "v"(addr), "v"(z));
    } else if (NOPS == 1) {
      asm volatile(
          "v_pk_add_f32 %1, %3, %1 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
          "ds_bpermute_b32 %2, %4, %5\n\t"
          "s_nop 0\n\t"
          "v_pk_add_f32 %0, %0, %1\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          : "+v"(Cc), "+v"(B), "=&v"(zs)
          : "v"(A), "v"(addr), "v"(z));
    } else {
      asm volatile(
          "v_pk_add_f32 %1, %3, %1 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
          "ds_bpermute_b32 %2, %4, %5\n\t"
          "s_nop 1\n\t"
          "v_pk_add_f32 %0, %0, %1\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          : "+v"(Cc), "+v"(B), "=&v"(zs)
          : "v"(A), "v"(addr), "v"(z));
    }
    keep += zs;
    bad_lo += Cc.x != want_lo;
    bad_hi += Cc.y != want_hi;
  }
  if (keep == 123.456f) bad_lo += 1;   // (keeps the LDS traffic and the shuffle alive)
  if (bad_lo) atomicAdd(&bad[0], bad_lo);
  if (bad_hi) atomicAdd(&bad[1], bad_hi);
}

// The kernel's sequence with ITS register overlaps (hard-coded registers): the shuffle's destination v25 is a source of P,
// the two shuffles behind C write C's own sources and read C's result, then the round's packed add consumes them.
template <bool LDS_BUSY, int NOP_AFTER_P>
__global__ __launch_bounds__(512) void k_exact(int rounds, unsigned seed, unsigned long long* __restrict__ bad) {
  __shared__ float s_buf[8 * 64 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long bad_lo = 0, bad_hi = 0, bad_z = 0;
  float keep = 0.0f;
  for (int r = 0; r < rounds; ++r) {
    const unsigned h = (blockIdx.x * 512u + threadIdx.x) * 2654435761u ^ (seed + r) * 0x9e3779b9u;
    if (wave >= 4) {
      if (LDS_BUSY) {
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
          s_buf[(wave * 64 + lane) * 4 + (k & 3)] = keep + (float)k;
          keep += s_buf[(wave * 64 + ((lane + k) & 63)) * 4 + ((k + 1) & 3)];
        }
      }
      continue;
    }
    // v[24:25] = a, v[22:23] = b, v[20:21] = c, v19 = z, v35 = byte address of lane ^ 16
    const unsigned hp = (blockIdx.x * 512u + (threadIdx.x ^ 16)) * 2654435761u ^ (seed + r) * 0x9e3779b9u;   // the partner lane's hash
    const float a0 = mix(h), a1 = mix(h + 1), b0 = mix(h + 2), b1 = mix(h + 3), c0 = mix(h + 4), c1 = mix(h + 5), z = mix(h + 6);
    const float pa0 = mix(hp), pa1 = mix(hp + 1), pb0 = mix(hp + 2), pb1 = mix(hp + 3), pc0 = mix(hp + 4), pc1 = mix(hp + 5),
                pz = mix(hp + 6);
    const float C_lo = c0 + (a0 + b1), C_hi = c1 + (a1 + b0), pC_lo = pc0 + (pa0 + pb1), pC_hi = pc1 + (pa1 + pb0);
    const float want_lo = C_lo + pC_lo, want_hi = C_hi + pC_hi, want_z = z + pz;
    const int addr = ((lane ^ 16) & 63) << 2;
    float out_lo, out_hi, out_z;
    asm volatile(
        "v_mov_b32 v24, %3\n\tv_mov_b32 v25, %4\n\tv_mov_b32 v22, %5\n\tv_mov_b32 v23, %6\n\t"
        "v_mov_b32 v20, %7\n\tv_mov_b32 v21, %8\n\tv_mov_b32 v19, %9\n\tv_mov_b32 v35, %10\n\t"
        "s_nop 4\n\t"
        "v_pk_add_f32 v[22:23], v[24:25], v[22:23] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
        "ds_bpermute_b32 v25, v35, v19\n\t"
        "v_pk_add_f32 v[20:21], v[20:21], v[22:23]\n\t"
        "ds_bpermute_b32 v23, v35, v21\n\t"
        "ds_bpermute_b32 v22, v35, v20\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        "v_add_f32_e32 v19, v19, v25\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_pk_add_f32 v[20:21], v[20:21], v[22:23]\n\t"
        "s_nop 4\n\t"
        "v_mov_b32 %0, v20\n\tv_mov_b32 %1, v21\n\tv_mov_b32 %2, v19\n\t"
        : "=&v"(out_lo), "=&v"(out_hi), "=&v"(out_z)
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1), "v"(z), "v"(addr)
        : "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v35");
    bad_lo += out_lo != want_lo;
    bad_hi += out_hi != want_hi;
    bad_z += out_z != want_z;
  }
  if (keep == 123.456f) bad_lo += 1;
  if (bad_lo) atomicAdd(&bad[0], bad_lo);
  if (bad_hi) atomicAdd(&bad[1], bad_hi);
  if (bad_z) atomicAdd(&bad[1], bad_z << 32);
}

typedef void (*seq_fn)(int, unsigned, unsigned long long*);

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
  unsigned long long* bad;
  (void)hipMalloc(&bad, 16);
  const struct { const char* name; seq_fn fn; } modes[] = {
      {"P, s_nop 0, C (adjacent: what hipcc emits)      LDS idle", k_seq<-1, false>},
      {"P, ds_bpermute, C                               LDS idle", k_seq<0, false>},
      {"P, ds_bpermute, C                               LDS busy", k_seq<0, true>},
      {"P, ds_bpermute, s_nop 0, C                      LDS idle", k_seq<1, false>},
      {"P, ds_bpermute, s_nop 1, C                      LDS idle", k_seq<2, false>},
      {"the kernel's own registers and overlaps         LDS idle", k_exact<false, 0>},
      {"the kernel's own registers and overlaps         LDS busy", k_exact<true, 0>}};
  for (const auto& m : modes) {
    (void)hipMemset(bad, 0, 16);
    for (int l = 0; l < 50; ++l) hipLaunchKernelGGL(m.fn, dim3(1024), dim3(512), 0, 0, rounds, (unsigned)(l * 7919 + 1), bad);
    const hipError_t err = hipDeviceSynchronize();
    if (err != hipSuccess) return printf("HIP error: %s\n", hipGetErrorString(err)), 1;
    unsigned long long h[2];
    (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
    printf("%s  %llu sequences: wrong low half %llu, wrong high half %llu\n", m.name,
           50ull * 1024ull * 256ull * (unsigned long long)rounds, h[0], h[1]);
    fflush(stdout);
  }
  return 0;
}
