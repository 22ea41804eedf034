// Developer micro-test (GPU box): does v_mfma_f32_16x16x32_f16 keep subnormal f16 inputs?
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f16_denorm.hip -o /tmp/mfma_f16_denorm && /tmp/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float *out, float a, float b) {
  f16x8 A, B;
  for (int i = 0; i < 8; i++) A[i] = (_Float16)0.0f, B[i] = (_Float16)0.0f;
  A[0] = (_Float16)a;  // every lane: A[row][k = 8 (lane >> 4)] = a
  B[0] = (_Float16)b;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, acc, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = acc[0];
}
int main() {
  float *d, h;
  hipMalloc(&d, 4);
  const float as[4] = {1.0f, 0x1p-16f, 0x1p-20f, 0x1p-24f};
  for (int i = 0; i < 4; i++) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, as[i], 1024.0f);
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("a = %g (f16 %s) x 1024, four k-groups: got %g expected %g\n", as[i], as[i] < 0x1p-14f ? "subnormal" : "normal", h, 4.0f * as[i] * 1024.0f);
  }
  return 0;
}
