// Round 5 probe: does a wave's arithmetic stay exact while ANOTHER queue's kernel keeps the SIMDs busy?
// Kernel A (the "victim", one wave per workgroup, 128 workgroups -- the shape of k_homography): every lane runs a long
// dependent chain twice on the same inputs and compares the bits; the chain is one of
//   mode 0: fma only                       mode 1: IEEE divisions (v_div_scale / v_rcp_f32 / v_div_fmas / v_div_fixup)
//   mode 2: square roots (v_sqrt_f32)      mode 3: raw v_rcp_f32 + fma
// Kernel B (the "hog", on a second stream, optional): mode h = 0 none, 1 v_exp_f32 loop (transcendental unit), 2 fma loop,
// 3 MFMA loop, 4 LDS + VMEM traffic.  Reported: lanes whose two evaluations differ, by lane quarter.
// usage: trans_contention [launches]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ float chain(float x, int mode, int iters) {
  float a = x, b = 1.0f + x * 0.25f;
  for (int i = 0; i < iters; i++) {
    if (mode == 0) {
      a = __builtin_fmaf(a, 0.99991f, b);
      b = __builtin_fmaf(b, 0.5f, a * 0.25f);
    } else if (mode == 1) {
      a = (a + 3.0f) / (b + 1.5f);
      b = (b + 2.0f) / (a + 1.25f);
    } else if (mode == 2) {
      a = sqrtf(a * a + b);
      b = sqrtf(b + a) + 0.5f;
    } else {
      a = __builtin_fmaf(__builtin_amdgcn_rcpf(b + 1.5f), a + 3.0f, 0.125f);
      b = __builtin_fmaf(__builtin_amdgcn_rcpf(a + 1.25f), b + 2.0f, 0.25f);
    }
  }
  return a + b;
}

__global__ __launch_bounds__(64) void k_victim(int mode, int iters, const float *in, unsigned *bad /* [4] by quarter */, float *sink) {
  const int lane = threadIdx.x;
  float x = in[blockIdx.x * 64 + lane];
  float x2 = x;
  asm volatile("" : "+v"(x2));
  const float r1 = chain(x, mode, iters);
  asm volatile("" ::: "memory");
  const float r2 = chain(x2, mode, iters);
  if (__float_as_uint(r1) != __float_as_uint(r2)) atomicAdd(&bad[lane >> 4], 1u);
  sink[blockIdx.x * 64 + lane] = r1;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void k_hog(int kind, int iters, float *sink) {
  __shared__ float lds[4096];
  const int t = threadIdx.x;
  float a = 0.001f * t, b = 1.0f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  bf16x8 va = {1, 2, 3, 4, 5, 6, 7, 8}, vb = {2, 3, 4, 5, 6, 7, 8, 9};
  lds[t] = a;
  __syncthreads();
  for (int i = 0; i < iters; i++) {
    if (kind == 1) {
      a = __builtin_amdgcn_exp2f(a * 0.5f - 1.0f);
      b = __builtin_amdgcn_exp2f(b * 0.25f - 0.5f) + a;
    } else if (kind == 2) {
      a = __builtin_fmaf(a, 0.999f, b);
      b = __builtin_fmaf(b, 0.5f, 0.25f);
    } else if (kind == 3) {
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vb, acc, 0, 0, 0);
    } else {
      a += lds[(t * 17 + i) & 4095];
      lds[(t + i) & 4095] = b;
      b += sink[(blockIdx.x * 256 + t + i * 64) & 0xFFFF];
    }
  }
  sink[blockIdx.x * 256 + t] = a + b + acc[0];
}

int main(int argc, char **argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 400;
  const int nb = 128, iters = 400;
  float *in, *sink, *hsink;
  unsigned *bad;
  hipMalloc(&in, nb * 64 * 4); hipMalloc(&sink, nb * 64 * 4); hipMalloc(&hsink, 1 << 22); hipMalloc(&bad, 16);
  float h[nb * 64];
  for (int i = 0; i < nb * 64; i++) h[i] = 0.5f + (float)(i % 977) * 0.01f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  hipMemset(hsink, 0, 1 << 22);
  hipStream_t sa, sb;
  hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  static const char *vm[] = {"fma chain", "IEEE division chain", "sqrt chain", "rcp + fma chain"};
  static const char *hm[] = {"no other kernel", "v_exp_f32 hog", "fma hog", "MFMA hog", "LDS + memory hog"};
  for (int hog = 0; hog < 5; hog++)
    for (int mode = 0; mode < 4; mode++) {
      hipMemset(bad, 0, 16);
      hipDeviceSynchronize();
      for (int l = 0; l < launches; l++) {
        if (hog && (l % 8) == 0) hipLaunchKernelGGL(k_hog, dim3(256 * 16), dim3(256), 0, sb, hog, 6000, hsink);
        hipLaunchKernelGGL(k_victim, dim3(nb), dim3(64), 0, sa, mode, iters, in, bad, sink);
      }
      hipDeviceSynchronize();
      unsigned hb[4];
      hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost);
      printf("%-18s | %-20s | lanes whose two evaluations differ, by quarter: %u %u %u %u  (of %d lane-evaluations)\n", hm[hog], vm[mode],
             hb[0], hb[1], hb[2], hb[3], launches * nb * 64);
    }
  return 0;
}
