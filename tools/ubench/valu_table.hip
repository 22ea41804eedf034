// Developer micro-benchmark (GPU box): issue cost of single VALU opcodes on gfx950, by inline assembly (the compiler cannot
// re-associate or strength-reduce them).  8 workgroups of 4 waves per CU, 8 independent register chains per lane, 32
// instructions per loop iteration.  Prints "cycles per wave64 instruction per SIMD" at the nominal 2.4 GHz; what matters is
// the RATIO between opcode classes (the sustained clock under an all-VALU load is ~2.1 GHz).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_table.hip -o /tmp/valu_table && /tmp/valu_table
#include <hip/hip_runtime.h>
#include <stdio.h>

// 32-bit chains: %0 = chain register (in/out), %1, %2 = loop-invariant registers
#define OP32(NAME, ASM)                                                                       \
  __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned a, unsigned b, int iters) { \
    unsigned r[8];                                                                            \
    for (int i = 0; i < 8; i++) r[i] = a + i * 977u + threadIdx.x;                            \
    unsigned m = b + threadIdx.x, c = a * 3u + 1u;                                            \
    for (int it = 0; it < iters; it++) {                                                      \
      _Pragma("unroll") for (int u = 0; u < 4; u++) {                                         \
        _Pragma("unroll") for (int i = 0; i < 8; i++)                                         \
          asm volatile(ASM : "+v"(r[i]) : "v"(m), "v"(c));                                    \
      }                                                                                       \
    }                                                                                         \
    unsigned s = 0;                                                                           \
    for (int i = 0; i < 8; i++) s += r[i];                                                    \
    out[blockIdx.x * 256 + threadIdx.x] = s;                                                  \
  }

// 64-bit chains (fp64, packed f32): %0 = 64-bit register pair
#define OP64(NAME, ASM)                                                                       \
  __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned a, unsigned b, int iters) { \
    double r[8];                                                                              \
    for (int i = 0; i < 8; i++) r[i] = 1.0 + 1e-3 * (a + i + threadIdx.x);                    \
    double m = 1.0000001 + 1e-9 * b, c = 1e-9 * (threadIdx.x + 1);                            \
    for (int it = 0; it < iters; it++) {                                                      \
      _Pragma("unroll") for (int u = 0; u < 4; u++) {                                         \
        _Pragma("unroll") for (int i = 0; i < 8; i++)                                         \
          asm volatile(ASM : "+v"(r[i]) : "v"(m), "v"(c));                                    \
      }                                                                                       \
    }                                                                                         \
    double s = 0;                                                                             \
    for (int i = 0; i < 8; i++) s += r[i];                                                    \
    out[blockIdx.x * 256 + threadIdx.x] = (unsigned)s;                                        \
  }

// integer, 32 bit
OP32(k_mov, "v_mov_b32 %0, %1")
OP32(k_add_u32, "v_add_u32 %0, %0, %1")
OP32(k_sub_u32, "v_sub_u32 %0, %0, %1")
OP32(k_and, "v_and_b32 %0, %0, %1")
OP32(k_xor, "v_xor_b32 %0, %0, %1")
OP32(k_lshl, "v_lshlrev_b32 %0, 1, %0")
OP32(k_ashr, "v_ashrrev_i32 %0, 1, %0")
OP32(k_lshr, "v_lshrrev_b32 %0, 1, %0")
OP32(k_lshr_v, "v_lshrrev_b32 %0, %1, %0")
OP32(k_ashr_v, "v_ashrrev_i32 %0, %1, %0")
OP32(k_lshl_v, "v_lshlrev_b32 %0, %1, %0")
OP32(k_or, "v_or_b32 %0, %0, %1")
OP32(k_not, "v_not_b32 %0, %0")
OP32(k_subrev, "v_subrev_u32 %0, %0, %1")
OP32(k_xad, "v_xad_u32 %0, %0, %1, %2")
OP32(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
OP32(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
OP32(k_bfrev, "v_bfrev_b32 %0, %0")
OP32(k_add_lit, "v_add_u32 %0, 0x12345, %0")
OP32(k_and_lit, "v_and_b32 %0, 0xfffff, %0")
OP32(k_mul_u24_vop2, "v_mul_u32_u24 %0, %0, %1")
OP32(k_max_u16, "v_max_u16 %0, %0, %1")
OP32(k_add_u16, "v_add_u16 %0, %0, %1")
OP32(k_min_u16, "v_min_u16 %0, %0, %1")
OP32(k_max_i16, "v_max_i16 %0, %0, %1")
OP32(k_sub_u16, "v_sub_u16 %0, %0, %1")
OP32(k_mul_lo_u16, "v_mul_lo_u16 %0, %0, %1")
OP32(k_lshl_b16, "v_lshlrev_b16 %0, 1, %0")
OP32(k_lshr_b16, "v_lshrrev_b16 %0, 1, %0")
OP32(k_mad_u16, "v_mad_u16 %0, %0, %1, %2")
OP32(k_cmp_u16, "v_cmp_gt_u16 vcc, %0, %1")
OP32(k_cmp_u16_e64, "v_cmp_gt_u16 s[20:21], %0, %1")
OP32(k_cmp_eq_u32, "v_cmp_eq_u32 vcc, %0, %1")
OP32(k_max3_u16, "v_max3_u16 %0, %0, %1, %2")
OP32(k_med3_u16, "v_med3_u16 %0, %0, %1, %2")
OP32(k_add_f16, "v_add_f16 %0, %0, %1")
OP32(k_max_f16, "v_max_f16 %0, %0, %1")
OP32(k_mul_f16, "v_mul_f16 %0, %0, %1")
OP32(k_cvt_f16_u16, "v_cvt_f16_u16 %0, %0")
OP32(k_sub_f32, "v_sub_f32 %0, %0, %1")
OP32(k_min_f32, "v_min_f32 %0, %0, %1")
OP32(k_mul_legacy, "v_mul_legacy_f32 %0, %0, %1")
OP32(k_add_f32_clamp, "v_add_f32_e64 %0, %0, %1 clamp")
OP32(k_add_f32_abs, "v_add_f32_e64 %0, %0, |%1|")
OP32(k_fma_f32_abs, "v_fma_f32 %0, |%0|, %1, %2")
OP32(k_add_f32_e64, "v_add_f32_e64 %0, %0, %1")
OP32(k_mul_f32_lit, "v_mul_f32 %0, 0x3f8ccccd, %0")
OP32(k_add_f32_sgpr, "v_add_f32 %0, s20, %0")
OP32(k_readlane, "v_readlane_b32 s20, %0, 5")
OP32(k_max_i32, "v_max_i32 %0, %0, %1")
OP32(k_min_u32, "v_min_u32 %0, %0, %1")
OP32(k_add3, "v_add3_u32 %0, %0, %1, %2")
OP32(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
OP32(k_lshl_or, "v_lshl_or_b32 %0, %0, 2, %1")
OP32(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
OP32(k_bfe, "v_bfe_u32 %0, %0, 3, 9")
OP32(k_bfi, "v_bfi_b32 %0, %1, %0, %2")
OP32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
OP32(k_cmp_i32, "v_cmp_gt_i32 vcc, %0, %1")
OP32(k_cmp_cnd, "v_cmp_gt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
OP32(k_mul_i24, "v_mul_i32_i24 %0, %0, %1")
OP32(k_mad_i24, "v_mad_i32_i24 %0, %0, %1, %2")
OP32(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
OP32(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
OP32(k_med3_i32, "v_med3_i32 %0, %0, %1, %2")
OP32(k_max3_i32, "v_max3_i32 %0, %0, %1, %2")
OP32(k_sad_u8, "v_sad_u8 %0, %0, %1, %2")
OP32(k_sad_u16, "v_sad_u16 %0, %0, %1, %2")
OP32(k_perm, "v_perm_b32 %0, %0, %1, %2")
OP32(k_alignbyte, "v_alignbyte_b32 %0, %0, %1, 1")
OP32(k_dot4, "v_dot4_u32_u8 %0, %1, %2, %0")
OP32(k_dot2_u16, "v_dot2_u32_u16 %0, %1, %2, %0")
OP32(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
OP32(k_pk_max_u16, "v_pk_max_u16 %0, %0, %1")
OP32(k_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
OP32(k_pk_sub_i16, "v_pk_sub_i16 %0, %0, %1")
OP32(k_cvt_pk_i16, "v_cvt_pk_i16_i32 %0, %0, %1")
OP32(k_mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
OP32(k_add_dpp, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
OP32(k_add_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD")
// fp32
OP32(k_add_f32, "v_add_f32 %0, %0, %1")
OP32(k_mul_f32, "v_mul_f32 %0, %0, %1")
OP32(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
OP32(k_fmac_f32, "v_fmac_f32 %0, %1, %2")
OP32(k_max_f32, "v_max_f32 %0, %0, %1")
OP32(k_med3_f32, "v_med3_f32 %0, %0, %1, %2")
OP32(k_max3_f32, "v_max3_f32 %0, %0, %1, %2")
OP32(k_cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
OP32(k_cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
OP32(k_cvt_f32_ub0, "v_cvt_f32_ubyte0 %0, %0")
OP32(k_cvt_f32_ub2, "v_cvt_f32_ubyte2 %0, %0")
OP32(k_cvt_u32_f32, "v_cvt_u32_f32 %0, %0")
OP32(k_rndne_f32, "v_rndne_f32 %0, %0")
OP32(k_cmp_f32, "v_cmp_gt_f32 vcc, %0, %1")
OP32(k_cmp_cnd_f32, "v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
OP32(k_fma_f16, "v_fma_f16 %0, %0, %1, %2")
OP32(k_pk_fma_f16, "v_pk_fma_f16 %0, %0, %1, %2")
OP32(k_pk_add_f16, "v_pk_add_f16 %0, %0, %1")
OP32(k_exp_f32, "v_exp_f32 %0, %0")
OP32(k_rcp_f32, "v_rcp_f32 %0, %0")
// 64-bit operands
OP64(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
OP64(k_add_f64, "v_add_f64 %0, %0, %1")
OP64(k_mul_f64, "v_mul_f64 %0, %0, %1")
OP64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
OP64(k_pk_add_f32, "v_pk_add_f32 %0, %0, %1")
OP64(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
OP64(k_lshl_b64, "v_lshlrev_b64 %0, 1, %0")

typedef void (*kern_t)(unsigned *, unsigned, unsigned, int);
static double g_ref = 0;
static void run(const char *name, kern_t k, int per_asm) {
  unsigned *d;
  hipMalloc(&d, sizeof(unsigned) * 256 * 2048);
  const int iters = 2048, blocks = 256 * 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u, 16);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double winstr = (double)blocks * 4 * iters * 32 * per_asm / (256.0 * 4);
  const double cyc = ms * 1e-3 * 2.4e9 / winstr;
  if (g_ref == 0) g_ref = cyc;
  printf("%-28s %8.3f ms  %5.2f cycles / wave64 instruction / SIMD   (x %.2f of v_mov_b32)\n", name, ms, cyc, cyc / g_ref);
  hipFree(d);
}
#define RUN(K) run(#K + 2, K, 1)
#define RUN2(K) run(#K + 2, K, 2)

int main() {
  RUN(k_mov); RUN(k_add_u32); RUN(k_sub_u32); RUN(k_and); RUN(k_xor); RUN(k_lshl); RUN(k_ashr); RUN(k_lshr); RUN(k_lshr_v); RUN(k_ashr_v); RUN(k_lshl_v); RUN(k_or); RUN(k_not); RUN(k_subrev); RUN(k_xad); RUN(k_bcnt); RUN(k_mbcnt); RUN(k_bfrev); RUN(k_add_lit); RUN(k_and_lit); RUN(k_mul_u24_vop2); RUN(k_max_u16); RUN(k_add_u16); RUN(k_min_u16); RUN(k_max_i16); RUN(k_sub_u16); RUN(k_mul_lo_u16); RUN(k_lshl_b16); RUN(k_lshr_b16); RUN(k_mad_u16); RUN(k_cmp_u16); RUN(k_cmp_u16_e64); RUN(k_cmp_eq_u32); RUN(k_max3_u16); RUN(k_med3_u16); RUN(k_add_f16); RUN(k_max_f16); RUN(k_mul_f16); RUN(k_cvt_f16_u16); RUN(k_sub_f32); RUN(k_min_f32); RUN(k_mul_legacy); RUN(k_add_f32_clamp); RUN(k_add_f32_abs); RUN(k_fma_f32_abs); RUN(k_add_f32_e64); RUN(k_mul_f32_lit); RUN(k_add_f32_sgpr); RUN(k_readlane); RUN(k_max_i32); RUN(k_min_u32);
  RUN(k_add3); RUN(k_lshl_add); RUN(k_lshl_or); RUN(k_and_or); RUN(k_bfe); RUN(k_bfi); RUN(k_cndmask); RUN(k_cmp_i32);
  RUN2(k_cmp_cnd); RUN(k_mul_i24); RUN(k_mad_i24); RUN(k_mad_u24); RUN(k_mul_lo); RUN(k_med3_i32); RUN(k_max3_i32);
  RUN(k_sad_u8); RUN(k_sad_u16); RUN(k_perm); RUN(k_alignbyte); RUN(k_dot4); RUN(k_dot2_u16); RUN(k_pk_add_u16);
  RUN(k_pk_max_u16); RUN(k_pk_mad_u16); RUN(k_pk_sub_i16); RUN(k_cvt_pk_i16); RUN(k_mov_dpp); RUN(k_add_dpp); RUN(k_add_sdwa);
  RUN(k_add_f32); RUN(k_mul_f32); RUN(k_fma_f32); RUN(k_fmac_f32); RUN(k_max_f32); RUN(k_med3_f32); RUN(k_max3_f32);
  RUN(k_cvt_f32_i32); RUN(k_cvt_i32_f32); RUN(k_cvt_f32_ub0); RUN(k_cvt_f32_ub2); RUN(k_cvt_u32_f32); RUN(k_rndne_f32);
  RUN(k_cmp_f32); RUN2(k_cmp_cnd_f32); RUN(k_fma_f16); RUN(k_pk_fma_f16); RUN(k_pk_add_f16); RUN(k_exp_f32); RUN(k_rcp_f32);
  RUN(k_fma_f64); RUN(k_add_f64); RUN(k_mul_f64); RUN(k_pk_fma_f32); RUN(k_pk_add_f32); RUN(k_pk_mul_f32); RUN(k_lshl_b64);
  return 0;
}
