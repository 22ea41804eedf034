// Developer micro-benchmark (GPU box): issue cost of f64 / f32 / int VALU instructions on gfx950.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int OP>
__global__ __launch_bounds__(256) void k(double *out, double a, double b, int iters) {
  const int m = (int)(a * 1000.0) + (int)threadIdx.x;
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
      if (OP == 8) n[i] = n[i] * m;                                              // v_mul_lo_u32
      if (OP == 9) n[i] = (int)__umul24((unsigned)n[i], (unsigned)m);  // v_mul_u32_u24
      if (OP == 10) n[i] = (int)(__umul24((unsigned)n[i], (unsigned)m) + (unsigned)it);  // v_mad_u32_u24
      if (OP == 11) n[i] = n[i] * m + it;                                        // v_mad_u64_u32 / mul_lo + add
      if (OP == 12) n[i] = (int)__builtin_amdgcn_alignbyte((unsigned)n[i], (unsigned)m, (unsigned)it);
      if (OP == 13) n[i] = (int)__builtin_amdgcn_perm((unsigned)n[i], (unsigned)m, 0x05010400u + (unsigned)it);
      if (OP == 14) n[i] = (int)__builtin_amdgcn_sad_u8((unsigned)n[i], (unsigned)m, (unsigned)it);
      if (OP == 15) n[i] = max(min(n[i], m), it);                                // v_med3_i32
      if (OP == 16) x[i] = (double)(n[i] + it) + x[i];                           // v_cvt_f64_i32 + add
      if (OP == 17) n[i] = __double2loint(x[i] = x[i] + 6755399441055744.0) + n[i];
      if (OP == 18) f[i] = __builtin_amdgcn_exp2f(f[i]);
      if (OP == 19) f[i] = __builtin_amdgcn_rcpf(f[i]);
      if (OP == 20) { typedef float f2 __attribute__((ext_vector_type(2))); f2 v = {f[i], f[(i + 1) & 7]}; const f2 aa = {(float)a, (float)a}, bb = {(float)b, (float)b}; v = __builtin_elementwise_fma(v, aa, bb); f[i] = v.x; f[(i + 1) & 7] = v.y; }
      if (OP == 21) f[i] = __builtin_fmaxf(__builtin_fmaxf(f[i], (float)a), (float)it);
      if (OP == 22) f[i] = __builtin_amdgcn_sqrtf(f[i]);
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
  run<8>("v_mul_lo_u32");
  run<9>("v_mul_u32_u24");
  run<10>("v_mad_u32_u24");
  run<11>("i32 mul + add");
  run<12>("v_alignbyte_b32");
  run<13>("v_perm_b32");
  run<14>("v_sad_u8");
  run<15>("v_med3_i32");
  run<16>("cvt_f64_i32 + add_f64");
  run<17>("add_f64 + add_u32");
  run<18>("v_exp_f32");
  run<19>("v_rcp_f32");
  run<20>("v_pk_fma_f32 (per pk instr, 8 per iter)");
  run<21>("v_max3_f32");
  run<22>("v_sqrt_f32");
  return 0;
}
