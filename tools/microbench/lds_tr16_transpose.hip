// lds_tr16_transpose.hip — what ds_read_b64_tr_b16 returns, checked on the layout the per-wave MLP backward uses to turn
// accumulator-layout activations (lane = sample, registers = features) into MFMA operands whose K dimension is the sample
// (dW = G^T X): every lane writes the 4 features it holds of ONE sample as one ds_write_b64 into a [sample][16 features]
// bf16 block; lane t of 16-lane group g then reads the 8-byte chunk (row 4g + t/4, column group t%4) with the transposing
// read and must receive feature t of samples 4g .. 4g+3 — in the row-major chunk order and in the bank-friendly one the kernels use.
//   hipcc -O3 --offload-arch=gfx950 lds_tr16_transpose.hip -o lds_tr16_transpose && ./lds_tr16_transpose
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) short s16x4;

// chunk (s, c) = features 4c .. 4c+3 of sample s, in elements (4 per chunk).  ORDER 0: row-major [s][c] (4-way bank conflicts on
// the writes); 1: the order field_mlp_bwd_pw.hip ships, 16 c + (s ^ 8 (c >> 1)) (conflict-free both ways).  The transposing read
// only needs lane t of a 16-lane group to point at chunk (row t/4, column group t%4): the chunk order is free.
#ifndef ORDER
#define ORDER 1
#endif
__device__ int chunk(int s, int c) { return ORDER ? 4 * (16 * c + (s ^ ((c >> 1) << 3))) : 16 * s + 4 * c; }

__global__ void k(unsigned short* out, unsigned short* out_rowmajor) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[2 * 16 * 16];
  const int l = threadIdx.x, j = l & 15, g = l >> 4;
  // value id of (sample s, feature f) = 256 + 16 s + f; lane (j, g) holds features 4g..4g+3 of sample j (tile a) and of
  // sample 16 + j (tile b, second block)
  for (int tile = 0; tile < 2; ++tile) {
    s16x4 v;
    for (int r = 0; r < 4; ++r) v[r] = (short)(256 + 16 * (16 * tile + j) + 4 * g + r);
    *reinterpret_cast<s16x4*>(lds + tile * 256 + chunk(j, g)) = v;
  }
  __syncthreads();
  for (int tile = 0; tile < 2; ++tile) {
    const unsigned short* p = lds + tile * 256 + chunk(4 * g + (j >> 2), j & 3);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
    for (int e = 0; e < 4; ++e) out[(tile * 64 + l) * 4 + e] = (unsigned short)v[e];
    // the same addresses read without the transpose, for reference
    s16x4 w = *reinterpret_cast<const s16x4*>(p);
    for (int e = 0; e < 4; ++e) out_rowmajor[(tile * 64 + l) * 4 + e] = (unsigned short)w[e];
  }
}

int main() {
  unsigned short *d, *d2;
  hipMalloc(&d, 2 * 64 * 4 * 2);
  hipMalloc(&d2, 2 * 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, d2);
  std::vector<unsigned short> h(512), h2(512);
  hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
  hipMemcpy(h2.data(), d2, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int tile = 0; tile < 2; ++tile)
    for (int l = 0; l < 64; ++l) {
      const int t = l & 15, g = l >> 4;
      for (int e = 0; e < 4; ++e) {
        const int want = 256 + 16 * (16 * tile + 4 * g + e) + t;  // feature t of sample 4g + e
        const int got = h[(tile * 64 + l) * 4 + e];
        if (got != want) {
          if (bad < 16) printf("tile %d lane %2d e %d: got (s %d, f %d) want (s %d, f %d)\n", tile, l, e, (got - 256) >> 4, (got - 256) & 15, (want - 256) >> 4, (want - 256) & 15);
          ++bad;
        }
      }
    }
  printf("lane 5 (g 0, t 5), tile 0: tr  =");
  for (int e = 0; e < 4; ++e) printf(" (s %d, f %d)", (h[5 * 4 + e] - 256) >> 4, (h[5 * 4 + e] - 256) & 15);
  printf("\nlane 5 (g 0, t 5), tile 0: raw =");
  for (int e = 0; e < 4; ++e) printf(" (s %d, f %d)", (h2[5 * 4 + e] - 256) >> 4, (h2[5 * 4 + e] - 256) & 15);
  printf("\nlane 37 (g 2, t 5), tile 1: tr =");
  for (int e = 0; e < 4; ++e) printf(" (s %d, f %d)", (h[(64 + 37) * 4 + e] - 256) >> 4, (h[(64 + 37) * 4 + e] - 256) & 15);
  printf("\n%s: %d mismatches of 512\n", bad ? "FAIL" : "OK", bad);
  return bad != 0;
}
