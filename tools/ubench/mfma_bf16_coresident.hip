// Developer probe (GPU box): is v_mfma_f32_16x16x32_bf16 deterministic when two workgroups share a CU
// (two waves per SIMD)?  Block b runs dependent chains on operands seeded by b % 7: blocks of the same class
// must agree whatever their neighbours on the CU compute.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 2) void k(uint32_t *out, int iters, int lds_touch) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, cls = blockIdx.x % 7;
  u32x4 a[5], b[3];
  for (int i = 0; i < 5; i++) a[i] = (u32x4){0x3f803f80u + tid * 3 + i + 16 * cls, 0x3f003f80u ^ (tid << 4), 0x3e803f00u + i, 0x3f803e80u};
  for (int i = 0; i < 3; i++) b[i] = (u32x4){0x3f803f00u + tid + i + 32 * cls, 0x3e803f80u ^ (tid << 3), 0x3f003f00u + i, 0x3f803f80u};
  f32x4 acc[5][3];
  for (int i = 0; i < 5; i++) for (int j = 0; j < 3; j++) acc[i][j] = (f32x4){0, 0, 0, 0};
  if (lds_touch) { ((uint32_t *)smem)[tid] = tid; __syncthreads(); }
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[j]), __builtin_bit_cast(bf16x8, a[i]), acc[i][j], 0, 0, 0);
      }
  uint32_t sum = 0;
  for (int i = 0; i < 5; i++) for (int j = 0; j < 3; j++) for (int v = 0; v < 4; v++) sum = sum * 31u + __float_as_uint(acc[i][j][v] * 1e-6f);
  out[blockIdx.x * 256 + tid] = sum;
}
int main() {
  const int nb = 2048;
  uint32_t *d;
  hipMalloc(&d, nb * 256 * 4);
  for (int lds = 0; lds < 2; lds++) {
    const int bytes = 78752;
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), bytes, 0, d, 40, lds);
    std::vector<uint32_t> h(nb * 256);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int badblocks = 0, first = -1;
    for (int bI = 7; bI < nb; bI++) {
      bool bad = false;
      for (int t = 0; t < 256; t++) bad |= h[bI * 256 + t] != h[(bI % 7) * 256 + t];
      if (bad) { badblocks++; if (first < 0) first = bI; }
    }
    printf("lds_touch %d: blocks differing from block 0: %d of %d (first %d)\n", lds, badblocks, nb, first);
  }
  return 0;
}
