// Developer probe (GPU box): two workgroups per CU, every wave interleaves ds_read_b128 operand loads with
// v_mfma_f32_16x16x32_bf16 chains (the shape of the expiry CNN's conv2 loop).  Block b works on data of class
// b % 7; blocks of one class must agree bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kBytes = 78752;
__global__ __launch_bounds__(256, 2) void k(uint32_t *out, const u32x4 *bglob, int iters, int width) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, cls = blockIdx.x % 7;
  uint32_t *w = (uint32_t *)smem;
  for (int i = tid; i < 62720 / 4; i += 256) {
    const uint32_t e = (uint32_t)((i * 2654435761u) >> 20) & 0x7fu;  // small bf16-ish magnitudes
    w[i] = (0x3c003c00u + (e << 16) + ((e * 5u) & 0x7fu)) ^ (cls << 2);
  }
  __syncthreads();
  f32x4 acc[5][3];
  for (int i = 0; i < 5; i++) for (int j = 0; j < 3; j++) acc[i][j] = (f32x4){0, 0, 0, 0};
  const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) unsigned char *)smem;
  for (int it = 0; it < iters; it++) {
    u32x4 a[5], b[3];
    for (int j = 0; j < 3; j++) b[j] = bglob[((it % 44) * 3 + j) * 64 + (tid & 63)];
    for (int i = 0; i < 5; i++) {
      const uint32_t addr = base + (uint32_t)(((tid * 7 + i * 1123 + it * 517) % 3900) * 16);
      if (width == 128) a[i] = *(const u32x4 *)(smem + (addr - base));
      else {
        const uint32_t *p = (const uint32_t *)(smem + (addr - base));
        a[i] = (u32x4){p[0], p[1], p[2], p[3]};
      }
    }
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
      for (int j = 0; j < 3; j++)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
  }
  uint32_t sum = 0;
  for (int i = 0; i < 5; i++) for (int j = 0; j < 3; j++) for (int v = 0; v < 4; v++) sum = sum * 31u + __float_as_uint(acc[i][j][v]);
  out[blockIdx.x * 256 + tid] = sum;
}
int main() {
  const int nb = 4096;
  uint32_t *d;
  u32x4 *b;
  hipMalloc(&d, nb * 256 * 4);
  hipMalloc(&b, 44 * 3 * 64 * 16);
  std::vector<uint32_t> hb(44 * 3 * 64 * 4);
  for (size_t i = 0; i < hb.size(); i++) hb[i] = 0x3c003c00u + (uint32_t)((i * 40503u) & 0x7f007fu);
  hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, kBytes);
  for (int width : {128, 32}) {
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), kBytes, 0, d, b, 44, width);
    std::vector<uint32_t> h(nb * 256);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int badblocks = 0, first = -1;
    for (int bI = 7; bI < nb; bI++) {
      bool bad = false;
      for (int t = 0; t < 256; t++) bad |= h[bI * 256 + t] != h[(bI % 7) * 256 + t];
      if (bad) { badblocks++; if (first < 0) first = bI; }
    }
    printf("LDS read width %d: blocks differing from their class reference: %d of %d (first %d)\n", width, badblocks, nb, first);
  }
  return 0;
}
