// Developer probe: how many 64-thread workgroups fit a CU as a function of their LDS size (allocation granularity)
// build: hipcc --offload-arch=gfx950 -O2 -o lds_granularity lds_granularity.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(64) void k(int *o) {
  extern __shared__ int s[];
  s[threadIdx.x] = threadIdx.x;
  __syncthreads();
  o[threadIdx.x] = s[63 - threadIdx.x];
}
int main() {
  int prev = -1;
  for (int bytes = 12000; bytes <= 21000; bytes += 16) {
    int nb = 0;
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 64, bytes);
    if (nb != prev) printf("lds %d B -> %d workgroups per CU\n", bytes, nb);
    prev = nb;
  }
  return 0;
}
