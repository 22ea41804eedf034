#include <hip/hip_runtime.h>
__global__ void k(int *out) {
  int v = threadIdx.x * 3 + 1;
  int up = __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false);  // wave_shr:1
  int dn = __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false);  // wave_shl:1
  out[threadIdx.x] = up; out[64 + threadIdx.x] = dn;
  out[128 + threadIdx.x] = __shfl_up(v, 1, 64); out[192 + threadIdx.x] = __shfl_down(v, 1, 64);
}
int main() {
  int *d; hipMalloc(&d, 256 * 4); hipLaunchKernelGGL(k, 1, 64, 0, 0, d);
  int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; i++) { if (h[i] != h[128 + i]) bad++; if (h[64 + i] != h[192 + i]) bad++; }
  printf("mismatches %d (up0 %d/%d dn63 %d/%d)\n", bad, h[0], h[128], h[127], h[255]);
  return 0;
}
