// Developer probe (GPU box): v_mfma_f32_16x16x32_bf16 chains whose B fragments are PREFETCHED from global memory
// (dwordx4 or dword loads in flight while the matrix instructions of the previous step issue), two workgroups per
// CU.  Block b works on LDS data of class b % 7; blocks of one class must agree bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kBytes = 78752;
template <int MODE>  // 0: x4 prefetch, 1: dword prefetch, 2: x4 no prefetch (load, wait, mfma)
__global__ __launch_bounds__(256, 2) void k(uint32_t *out, const u32x4 *bglob, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, cls = blockIdx.x % 7;
  uint32_t *w = (uint32_t *)smem;
  for (int i = tid; i < 62720 / 4; i += 256) {
    const uint32_t e = (uint32_t)((i * 2654435761u) >> 20) & 0x7fu;
    w[i] = (0x3c003c00u + (e << 16) + ((e * 5u) & 0x7fu)) ^ (cls << 2);
  }
  __syncthreads();
  f32x4 acc[5][3];
  for (int i = 0; i < 5; i++) for (int j = 0; j < 3; j++) acc[i][j] = (f32x4){0, 0, 0, 0};
  auto loadb = [&](int it, u32x4 (&b)[3]) {
    for (int j = 0; j < 3; j++) {
      const u32x4 *p = bglob + ((it % 44) * 3 + j) * 64 + (tid & 63);
      if (MODE == 1) {
        const uint32_t *q = (const uint32_t *)p;
        b[j] = (u32x4){__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1), __builtin_nontemporal_load(q + 2), __builtin_nontemporal_load(q + 3)};
      } else b[j] = *p;
    }
  };
  u32x4 bn[3];
  loadb(0, bn);
  for (int it = 0; it < iters; it++) {
    u32x4 a[5], b[3];
    if (MODE == 2) loadb(it, b);
    else { for (int j = 0; j < 3; j++) b[j] = bn[j]; loadb(it + 1, bn); }
    for (int i = 0; i < 5; i++) a[i] = *(const u32x4 *)(smem + ((tid * 7 + i * 1123 + it * 517) % 3900) * 16);
    if (MODE == 2) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
      for (int j = 0; j < 3; j++)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
    if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
  }
  uint32_t sum = 0;
  for (int i = 0; i < 5; i++) for (int j = 0; j < 3; j++) for (int v = 0; v < 4; v++) sum = sum * 31u + __float_as_uint(acc[i][j][v]);
  out[blockIdx.x * 256 + tid] = sum;
}
template <int MODE>
void run(uint32_t *d, const u32x4 *b, int nb) {
  hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, kBytes);
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), kBytes, 0, d, b, 44);
    std::vector<uint32_t> h(nb * 256);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int badblocks = 0, first = -1;
    for (int bI = 7; bI < nb; bI++) {
      bool bad = false;
      for (int t = 0; t < 256; t++) bad |= h[bI * 256 + t] != h[(bI % 7) * 256 + t];
      if (bad) { badblocks++; if (first < 0) first = bI; }
    }
    printf("mode %d rep %d: blocks differing from their class reference: %d of %d (first %d)\n", MODE, rep, badblocks, nb, first);
  }
}
int main() {
  const int nb = 4096;
  uint32_t *d;
  u32x4 *b;
  hipMalloc(&d, nb * 256 * 4);
  hipMalloc(&b, 45 * 3 * 64 * 16);
  std::vector<uint32_t> hb(45 * 3 * 64 * 4);
  for (size_t i = 0; i < hb.size(); i++) hb[i] = 0x3c003c00u + (uint32_t)((i * 40503u) & 0x7f007fu);
  hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  run<0>(d, b, nb);
  run<1>(d, b, nb);
  run<2>(d, b, nb);
  return 0;
}
