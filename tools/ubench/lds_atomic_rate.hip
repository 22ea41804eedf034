// Developer micro-benchmark (GPU box): cost of ds_add_u32 (no return) per wave instruction on gfx950 as a function of how many
// lanes meet in one address / one bank.  One workgroup of W waves on one CU; cycles per instruction per wave and per CU.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_atomic_rate.hip -o /tmp/t && /tmp/t
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(long long *out, int mode, int iters) {
  __shared__ unsigned int h[4096];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += blockDim.x) h[i] = 0;
  __syncthreads();
  int idx;
  switch (mode) {
    case 0: idx = lane; break;                  // 64 addresses, 64 banks... (32 banks x 2)
    case 1: idx = 0; break;                     // one address
    case 2: idx = lane & 3; break;              // 4 addresses, 16 lanes each
    case 3: idx = lane & 15; break;             // 16 addresses, 4 lanes each
    case 4: idx = 32 * lane; break;             // 64 addresses, ONE bank
    case 5: idx = (lane * 7) & 63; break;       // permutation of 64 addresses
    default: idx = lane >> 1; break;            // pairs
  }
  idx += 64 * (tid >> 6);  // a region per wave
  const long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) atomicAdd(&h[(idx + 64 * 0) & 4095], 1u);
  }
  __syncthreads();
  const long long t1 = clock64();
  if (tid == 0) out[0] = t1 - t0;
  if (h[tid & 4095] == 0xffffffffu) out[1] = 1;
}
int main() {
  long long *d, h[2];
  hipMalloc(&d, 16);
  const char *names[7] = {"64 distinct addresses", "1 address (64 lanes)", "4 addresses x 16 lanes", "16 addresses x 4 lanes",
                          "64 addresses in one bank", "64 addresses permuted", "32 addresses x 2 lanes"};
  for (int waves = 1; waves <= 16; waves *= 4)
    for (int mode = 0; mode < 7; mode++) {
      const int iters = 256;
      hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, d, mode, iters);
      hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      printf("%2d waves, %-26s: %7.1f cycles per wave instruction, %6.1f per instruction on the CU\n", waves, names[mode],
             (double)h[0] / (iters * 8), (double)h[0] / (iters * 8 * waves));
    }
  return 0;
}
