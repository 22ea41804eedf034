// Developer micro-test (GPU box): precision of v_mfma_f32_16x16x32_f16 when the A operand holds bytes as f16 SUBNORMALS
// (bits 0x00dd = d x 2^-24) against the same contraction with the bytes as bf16 numbers on v_mfma_f32_16x16x32_bf16,
// and with the bytes' bits as bf16 numbers (0x00dd = d x 2^-133: exponent fields 0 and 1 continue one linear scale) against
// weights x 2^100.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f16_subnormal_dot.hip -o /tmp/t && /tmp/t
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
// A: [16 rows][32 k] bytes; B: [32 k][16 cols] floats.  out[variant][row][col]
__global__ void k(const unsigned char *A, const float *B, float *out) {
  const int lane = threadIdx.x, m = lane & 15, kk = lane >> 4;
  u16x8 ab, af;
  u16x8 bh[3], fh[3], sh[3];
  for (int e = 0; e < 8; e++) {
    const int kidx = 8 * kk + e;
    const unsigned char d = A[m * 32 + kidx];
    af[e] = d;                                                            // f16 subnormal d x 2^-24
    ab[e] = __builtin_bit_cast(unsigned short, (__bf16)(float)d);         // exact bf16
    const float w = B[kidx * 16 + m];                                     // lane's column = m
    // three bf16 parts
    float r = w;
    for (int p = 0; p < 3; p++) {
      const __bf16 h = (__bf16)r;
      bh[p][e] = __builtin_bit_cast(unsigned short, h);
      sh[p][e] = __builtin_bit_cast(unsigned short, (__bf16)((float)h * 0x1p100f));  // the same part, scaled (exact)
      r -= (float)h;
    }
    // three f16 parts of w x 4096
    double rr = (double)w * 4096.0;
    for (int p = 0; p < 3; p++) { const _Float16 h = (_Float16)(float)rr; fh[p][e] = __builtin_bit_cast(unsigned short, h); rr -= (double)(float)h; }
  }
  f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0};
  for (int p = 2; p >= 0; p--) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ab), __builtin_bit_cast(bf16x8, bh[p]), c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af), __builtin_bit_cast(f16x8, fh[p]), c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, sh[p]), c2, 0, 0, 0);
  }
  for (int v = 0; v < 4; v++) {
    out[(4 * kk + v) * 16 + m] = c0[v];
    out[256 + (4 * kk + v) * 16 + m] = c1[v] * 4096.0f;
    out[512 + (4 * kk + v) * 16 + m] = c2[v] * 0x1p33f;
  }
}
int main() {
  unsigned char hA[512];
  float hB[512], hout[768];
  srand(7);
  double worst[3] = {0, 0, 0};
  long differ = 0;
  unsigned char *dA; float *dB, *dout;
  hipMalloc(&dA, 512); hipMalloc(&dB, 2048); hipMalloc(&dout, 3072);
  for (int trial = 0; trial < 200; trial++) {
    for (int i = 0; i < 512; i++) hA[i] = (unsigned char)(rand() & 255), hB[i] = ((float)rand() / RAND_MAX - 0.5f) * 0.02f;
    hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dout);
    hipMemcpy(hout, dout, 3072, hipMemcpyDeviceToHost);
    for (int r = 0; r < 16; r++)
      for (int c = 0; c < 16; c++) {
        double ex = 0, mag = 0;
        for (int kx = 0; kx < 32; kx++) ex += (double)hA[r * 32 + kx] * (double)hB[kx * 16 + c], mag += fabs((double)hA[r * 32 + kx] * (double)hB[kx * 16 + c]);
        if (hout[r * 16 + c] != hout[512 + r * 16 + c]) differ++;
        for (int v = 0; v < 3; v++) { const double e = fabs((double)hout[v * 256 + r * 16 + c] - ex) / mag; if (e > worst[v]) worst[v] = e; }
      }
  }
  printf("worst |error| / sum |terms|:  bf16 bytes x 3 bf16 parts %.3g   f16-subnormal bytes x 3 f16 parts %.3g   bf16-bits bytes x 3 scaled bf16 parts %.3g (%ld of %d outputs differ from the first form)   (2^-24 = %.3g)\n", worst[0], worst[1], worst[2], differ, 200 * 256, ldexp(1.0, -24));
  return 0;
}
