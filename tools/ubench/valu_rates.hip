// Developer micro-benchmark (GPU box): issue cost of f64 / f32 / int VALU instructions on gfx950.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int OP>
__global__ __launch_bounds__(256) void k(double *out, double a, double b, int iters) {
  double x[8];
  float f[8];
  int n[8];
  for (int i = 0; i < 8; i++) x[i] = a + i + threadIdx.x, f[i] = (float)x[i], n[i] = (int)x[i];
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) x[i] = __builtin_fma(x[i], a, b);
      if (OP == 1) x[i] = x[i] * a;
      if (OP == 2) x[i] = x[i] + b;
      if (OP == 3) x[i] = __builtin_amdgcn_rcp(x[i]);
      if (OP == 4) f[i] = __builtin_fmaf(f[i], (float)a, (float)b);
      if (OP == 5) n[i] = n[i] * 3 + (int)b;
      if (OP == 6) n[i] = (int)__builtin_amdgcn_udot4((unsigned)n[i], 0x01020304u, (unsigned)it, false);
      if (OP == 7) x[i] = __builtin_amdgcn_fract(x[i]) + b;
    }
  }
  double s = 0;
  for (int i = 0; i < 8; i++) s += x[i] + f[i] + n[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char *name) {
  double *d;
  hipMalloc(&d, sizeof(double) * 256 * 1024 * 8);
  const int iters = 4096, blocks = 256 * 8;  // 8 workgroups per CU
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 1e-9, 16);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 1e-9, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // wave-instructions per SIMD: blocks * 4 waves * iters * 8 / (256 CUs * 4 SIMDs)
  const double winstr = (double)blocks * 4 * iters * 8 / (256.0 * 4);
  printf("%-22s %.3f ms  -> %.2f cycles per wave64 instruction per SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / winstr);
  hipFree(d);
}

int main() {
  run<0>("v_fma_f64");
  run<1>("v_mul_f64");
  run<2>("v_add_f64");
  run<3>("v_rcp_f64");
  run<4>("v_fma_f32");
  run<5>("v_mad_u32 (mul+add)");
  run<6>("v_dot4_u32_u8");
  run<7>("v_fract_f64 + add");
  return 0;
}
