// Developer micro-benchmark (GPU box): how long a wave waits for an L2-resident global load while the other waves of its CU
// are busy with (a) nothing, (b) VALU work, (c) LDS reads, (d) matrix instructions, (e) other global loads.  Motivation: in the
// scan kernels a new workgroup waits 5 - 20 k cycles for its first (cache-resident) loads once the kernel is in steady state,
// against ~3 k in the kernel's first round (csrc/digits.hip -DDMZ_DG_LOADWAIT).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/load_latency_under_load.hip -o /tmp/t && /tmp/t
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(1024) void k(const unsigned *__restrict__ src, long long *out, int mode, int iters, int nload) {
  __shared__ unsigned int h[8192];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 8192; i += blockDim.x) h[i] = i;
  __syncthreads();
  if (wave == 0) {
    // the probe wave: nload dword loads per lane (1 KB per instruction), timed; repeated a few times
    long long worst = 0, sum = 0;
    for (int rep = 0; rep < 16; rep++) {
      __builtin_amdgcn_s_sleep(20);
      unsigned v[17];
      const long long t0 = (long long)__builtin_readcyclecounter();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int q = 0; q < 17; q++) v[q] = q < nload ? __builtin_nontemporal_load(src + ((rep * 17 + q) * 256 + lane + 4096 * blockIdx.x) % (1 << 18)) : 0u;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const long long t1 = (long long)__builtin_readcyclecounter();
      unsigned acc = 0;
#pragma unroll
      for (int q = 0; q < 17; q++) acc ^= v[q];
      if (acc == 0x12345u) out[3] = 1;
      sum += t1 - t0;
      worst = t1 - t0 > worst ? t1 - t0 : worst;
    }
    if (lane == 0 && blockIdx.x == gridDim.x / 2) out[0] = sum / 16, out[1] = worst;
    return;
  }
  // the other waves: background work
  float a = lane * 0.5f, b = 1.0001f;
  unsigned x = lane;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  bf16x8 fa, fb;
  for (int i = 0; i < 8; i++) fa[i] = (__bf16)(float)(lane + i), fb[i] = (__bf16)1.0f;
  for (int it = 0; it < iters; it++) {
    if (mode == 1) {
#pragma unroll
      for (int u = 0; u < 32; u++) a = a * b + 0.25f;
    } else if (mode == 2) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const uint4 w = *(const uint4 *)&h[((x + 64 * u) * 4) & 8188];
        x += w.x + w.y + w.z + w.w;
      }
    } else if (mode == 3) {
#pragma unroll
      for (int u = 0; u < 8; u++) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c, 0, 0, 0);
    } else if (mode == 4) {
#pragma unroll
      for (int u = 0; u < 4; u++) x += src[(x * 64 + lane + 256 * u) & ((1 << 18) - 1)];
    } else if (mode == 5) {  // LDS reads + VALU + matrix, like a convolution loop
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint4 w = *(const uint4 *)&h[((x + 64 * u) * 4) & 8188];
        x += w.x & 3;
        a = a * b + (float)w.y;
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c, 0, 0, 0);
      }
    }
  }
  if (a + c[0] + (float)x == 0.123f) out[2] = 1;
}
int main() {
  long long *d, hh[4];
  unsigned *src;
  hipMalloc(&d, 32);
  hipMalloc(&src, 4 << 18);
  hipMemset(src, 1, 4 << 18);
  const char *names[6] = {"idle", "VALU (v_fma_f32)", "LDS reads (ds_read_b128)", "matrix (v_mfma 16x16x32 bf16)", "global loads (L2 hits)",
                          "LDS + VALU + matrix"};
  for (int waves = 4; waves <= 16; waves *= 2)
    for (int mode = 0; mode < 6; mode++)
      for (int nload = 1; nload <= 17; nload += 16) {
        hipMemset(d, 0, 32);
        hipLaunchKernelGGL(k, dim3(256 * 2), dim3(64 * waves), 0, 0, src, d, mode, mode == 0 ? 0 : 4000, nload);
        hipDeviceSynchronize();
        hipMemcpy(hh, d, 32, hipMemcpyDeviceToHost);
        printf("%2d waves per workgroup, others: %-32s %2d loads per lane: mean %6lld cycles, worst %6lld\n", waves, names[mode], nload, hh[0], hh[1]);
      }
  return 0;
}
