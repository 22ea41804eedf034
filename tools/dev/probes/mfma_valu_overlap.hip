// Developer probe (GPU box): do the matrix pipe and the vector ALU of a SIMD work at the same time?  Every SIMD holds W waves and
// per iteration is given W x NM matrix instructions (v_mfma_f32_16x16x32_f16 or 16x16x4_f32, four independent accumulators per wave)
// and W x NV VALU instructions (eight independent registers per wave), handed out in four ways:
//   M  matrix work only (all W waves)            V  VALU work only (all W waves)
//   I  every wave does NM matrix + NV VALU, interleaved in program order (one matrix instruction, then NV / NM VALU)
//   S  half of the SIMD's waves do 2 NM matrix each, the other half 2 NV VALU each
// If the pipes overlap, I and S take max(M, V); if a SIMD does one or the other, M + V.
// (512-thread workgroups: a workgroup's waves go round the four SIMDs, waves 0-3 / 4-7 take the two roles of S.)
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__device__ __forceinline__ void mm(f32x4 &acc, f16x8 a, f16x8 b) {
  if (SHAPE == 0) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  else acc = __builtin_amdgcn_mfma_f32_16x16x4f32((float)a[0], (float)b[0], acc, 0, 0, 0);
}
template <int VOP>
__device__ __forceinline__ void vv(float &x, float c) {
  if (VOP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
  else if (VOP == 1) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
  else asm volatile("v_exp_f32 %0, %0" : "+v"(x));
}

// one iteration: NM matrix and NV VALU instructions in program order M V..V M V..V
template <int SHAPE, int VOP, int NM, int NV>
__device__ __forceinline__ void body(f32x4 (&acc)[4], float (&x)[8], f16x8 a, f16x8 b, float c) {
  constexpr int N = NM > 0 ? NM : 1, PER = NM > 0 ? NV / NM : NV;
#pragma unroll
  for (int j = 0; j < N; j++) {
    if (NM > 0) mm<SHAPE>(acc[j & 3], a, b);
#pragma unroll
    for (int q = 0; q < PER; q++) vv<VOP>(x[(j * PER + q) & 7], c);
  }
}

template <int SHAPE, int VOP, int NM, int NV>
__global__ __launch_bounds__(512) void k(int mode, int iters, float *out) {
  f32x4 acc[4];
  float x[8];
  for (int i = 0; i < 4; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 1e-3f + i;
  f16x8 a, b;
  for (int i = 0; i < 8; i++) a[i] = (_Float16)(threadIdx.x & 3), b[i] = (_Float16)1;
  const float c = 0.999f;
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
  if (mode == 0) for (int it = 0; it < iters; it++) body<SHAPE, VOP, NM, 0>(acc, x, a, b, c);
  else if (mode == 1) for (int it = 0; it < iters; it++) body<SHAPE, VOP, 0, NV>(acc, x, a, b, c);
  else if (mode == 2) for (int it = 0; it < iters; it++) body<SHAPE, VOP, NM, NV>(acc, x, a, b, c);
  else if (role == 0) for (int it = 0; it < iters; it++) body<SHAPE, VOP, 2 * NM, 0>(acc, x, a, b, c);
  else for (int it = 0; it < iters; it++) body<SHAPE, VOP, 0, 2 * NV>(acc, x, a, b, c);
  float s = 0;
  for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; i++) s += x[i];
  if (s == 123.456f) out[0] = s;
}

// the same for v_mfma_f32_32x32x16_f16 (8 passes; two independent 16-register accumulators)
template <int VOP, int NM, int NV>
__device__ __forceinline__ void body32(f32x16 (&acc)[2], float (&x)[8], f16x8 a, f16x8 b, float c) {
  constexpr int N = NM > 0 ? NM : 1, PER = NM > 0 ? NV / NM : NV;
#pragma unroll
  for (int j = 0; j < N; j++) {
    if (NM > 0) acc[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j & 1], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < PER; q++) vv<VOP>(x[(j * PER + q) & 7], c);
  }
}
template <int VOP, int NM, int NV>
__global__ __launch_bounds__(512) void k32(int mode, int iters, float *out) {
  f32x16 acc[2];
  float x[8];
  for (int i = 0; i < 2; i++)
    for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 1e-3f + i;
  f16x8 a, b;
  for (int i = 0; i < 8; i++) a[i] = (_Float16)(threadIdx.x & 3), b[i] = (_Float16)1;
  const float c = 0.999f;
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
  if (mode == 0) for (int it = 0; it < iters; it++) body32<VOP, NM, 0>(acc, x, a, b, c);
  else if (mode == 1) for (int it = 0; it < iters; it++) body32<VOP, 0, NV>(acc, x, a, b, c);
  else if (mode == 2) for (int it = 0; it < iters; it++) body32<VOP, NM, NV>(acc, x, a, b, c);
  else if (role == 0) for (int it = 0; it < iters; it++) body32<VOP, 2 * NM, 0>(acc, x, a, b, c);
  else for (int it = 0; it < iters; it++) body32<VOP, 0, 2 * NV>(acc, x, a, b, c);
  float s = 0;
  for (int i = 0; i < 2; i++)
    for (int e = 0; e < 16; e++) s += acc[i][e];
  for (int i = 0; i < 8; i++) s += x[i];
  if (s == 123.456f) out[0] = s;
}
template <int VOP, int NM, int NV>
static float run32(int mode, int waves_per_simd, float *out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  const int grid = 256 * waves_per_simd / 2, iters = 20000;
  k32<VOP, NM, NV><<<grid, 512>>>(mode, iters, out);
  (void)hipEventRecord(e0);
  k32<VOP, NM, NV><<<grid, 512>>>(mode, iters, out);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / iters;
}
template <int VOP, int NM, int NV>
static void line32(const char *name, float *out) {
  for (int w = 2; w <= 4; w *= 2) {
    const float m = run32<VOP, NM, NV>(0, w, out), v = run32<VOP, NM, NV>(1, w, out);
    const float i = run32<VOP, NM, NV>(2, w, out), s = run32<VOP, NM, NV>(3, w, out);
    printf("%s  %d waves/SIMD, per wave and iteration %2d matrix + %3d VALU:  M %6.0f ns (%.1f ns per instruction and SIMD)  V %6.0f ns (%.2f)  "
           "I %6.0f  S %6.0f   [max %6.0f, sum %6.0f]\n", name, w, NM, NV, m, m / (w * NM), v, v / (w * NV), i, s, m > v ? m : v, m + v);
  }
}

template <int SHAPE, int VOP, int NM, int NV>
static float run(int mode, int waves_per_simd, float *out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  const int grid = 256 * waves_per_simd / 2, iters = 20000;  // 512 threads = two waves per SIMD
  k<SHAPE, VOP, NM, NV><<<grid, 512>>>(mode, iters, out);
  (void)hipEventRecord(e0);
  k<SHAPE, VOP, NM, NV><<<grid, 512>>>(mode, iters, out);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / iters;  // ns per iteration
}

template <int SHAPE, int VOP, int NM, int NV>
static void line(const char *name, float *out) {
  for (int w = 2; w <= 8; w *= 2) {
    const float m = run<SHAPE, VOP, NM, NV>(0, w, out), v = run<SHAPE, VOP, NM, NV>(1, w, out);
    const float i = run<SHAPE, VOP, NM, NV>(2, w, out), s = run<SHAPE, VOP, NM, NV>(3, w, out);
    printf("%s  %d waves/SIMD, per wave and iteration %2d matrix + %3d VALU:  M %6.0f ns (%.1f ns per instruction and SIMD)  V %6.0f ns (%.2f)  "
           "I %6.0f  S %6.0f   [max %6.0f, sum %6.0f]\n", name, w, NM, NV, m, m / (w * NM), v, v / (w * NV), i, s, m > v ? m : v, m + v);
  }
}

int main() {
  float *out;
  (void)hipMalloc(&out, 4);
  line<0, 0, 8, 32>("16x16x32 f16 + v_fma_f32 ", out);
  line<0, 0, 8, 64>("16x16x32 f16 + v_fma_f32 ", out);
  line<0, 0, 8, 128>("16x16x32 f16 + v_fma_f32 ", out);
  line<0, 1, 8, 32>("16x16x32 f16 + v_max3_f32", out);
  line<0, 1, 8, 64>("16x16x32 f16 + v_max3_f32", out);
  line<0, 2, 8, 16>("16x16x32 f16 + v_exp_f32 ", out);
  line<0, 2, 8, 32>("16x16x32 f16 + v_exp_f32 ", out);
  line<1, 0, 8, 64>("16x16x4 f32  + v_fma_f32 ", out);
  line<1, 0, 8, 128>("16x16x4 f32  + v_fma_f32 ", out);
  line<1, 1, 8, 64>("16x16x4 f32  + v_max3_f32", out);
  // the same matrix work in half as many instructions of twice the size
  line32<0, 4, 32>("32x32x16 f16 + v_fma_f32 ", out);
  line32<0, 4, 64>("32x32x16 f16 + v_fma_f32 ", out);
  line32<0, 4, 128>("32x32x16 f16 + v_fma_f32 ", out);
  line32<1, 4, 64>("32x32x16 f16 + v_max3_f32", out);
  return 0;
}
