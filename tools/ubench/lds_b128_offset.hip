// Developer probe (GPU box): does ds_read_b128 with a large immediate offset read the right bytes in the
// SECOND workgroup of a CU (LDS base != 0, absolute addresses beyond 128 KB)?
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_b128_offset.hip -o /tmp/lds_probe && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kBytes = 78752;
__global__ __launch_bounds__(256, 2) void k(int *bad, int mode, int spin) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t *w = (uint32_t *)smem;
  const int tid = threadIdx.x;
  for (int i = tid; i < kBytes / 4; i += 256) w[i] = 0x01000000u * (blockIdx.x & 127) + i;
  __syncthreads();
  int nbad = 0;
  for (int rep = 0; rep < spin; rep++) {
    for (int q = tid; q < 31360 / 16; q += 256) {
      const uint32_t a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) unsigned char *)smem + 16 * q;
      u32x4 r;
      if (mode == 0) asm volatile("ds_read_b128 %0, %1 offset:31360\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(a));
      else if (mode == 1) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(a + 31360));
      else { u32x4 r2; asm volatile("ds_read_b128 %0, %2 offset:31360\n\tds_read_b128 %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r), "=&v"(r2) : "v"(a)); }
      const uint32_t want = 0x01000000u * (blockIdx.x & 127) + (31360 + 16 * q) / 4;
      if (r.x != want || r.y != want + 1 || r.z != want + 2 || r.w != want + 3) nbad++;
    }
  }
  if (nbad) atomicAdd(&bad[blockIdx.x], nbad);
}
int main() {
  const int nb = 2048;
  int *d;
  hipMalloc(&d, nb * 4);
  for (int mode = 0; mode < 3; mode++) {
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, kBytes);
    hipMemset(d, 0, nb * 4);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), kBytes, 0, d, mode, 20);
    std::vector<int> h(nb);
    hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost);
    int nbadblocks = 0, first = -1;
    long tot = 0;
    for (int i = 0; i < nb; i++)
      if (h[i]) { nbadblocks++; tot += h[i]; if (first < 0) first = i; }
    printf("mode %d: bad blocks %d of %d (first %d), mismatching reads %ld\n", mode, nbadblocks, nb, first, tot);
  }
  return 0;
}
