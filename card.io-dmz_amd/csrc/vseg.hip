// vseg.hip -- number-row search on a batch of rectified 428 x 270 cards.
//
// Replaces best_n_vseg (scan/n_vseg.cpp:94-168) and the frame gates that follow it
// (scan/frame.cpp:38-47): per row ROI (10,y,408,1): 3-tap morphological gradient
// (cv/morph.cpp:108-112), x0.5 linear down-sample (cv/convert.cpp:195-197), min-max
// normalise (cv/convert.cpp:380-383), MLP 204-50-3 (modelm_befe75da.cpp:1770-1786),
// 27-row running box sum (n_vseg.cpp:49-92) -- coarse pass on every 4th row, then the
// fine pass around the best offset, exactly as the reference schedules them.
//
// CDNA4 mapping: one workgroup (4 waves) per card, ~21 KB of LDS so that four to
// five cards are resident per CU and one card's serial phases hide behind the others'.
//   * row features: a wave per row, six rows in flight; lane t loads three aligned
//     dwords (12 bytes) of the row and produces 4 of the 204 down-sampled gradient
//     bytes in registers (v_max3/v_min3); a wave min/max reduction gives the row's
//     normalisation (scale, shift).  LDS keeps the u8 gradients + (scale, shift) per
//     row -- 1/4 of the float features.
//   * hidden layer = the one real contraction of the stage, [rows x 204] x [204 x 50], on
//     v_mfma_f32_16x16x32_f16 with exact data operands (a gradient byte zero-extended to 16 bits is
//     the f16 subnormal d x 2^-24; the weights / 255 x 2^12 go in three f16 parts, the row's scale and
//     shift enter once per output: see vseg_mlp_rows_bf16).  Wave w owns hidden units 16w..16w+15 and streams
//     its 21 weight fragments once per pass; all row tiles of a pass go through one sweep
//     over k.  (The fp32 matrix core with register-resident weights, v_mfma_f32_16x16x4_f32,
//     needed 16 times the matrix-pipe time per k and four VALU instructions per feature:
//     0.47 instead of 0.33 ms per 8192 cards; the model entry point below still uses it.)
//   * tanh, the 50->3 logistic layer (wave-local 16-lane reductions + a 4-wave LDS
//     sum), softmax, and the literal running box sum on one lane (its LDS reads are
//     independent of the float chain and are issued nine steps ahead).
// Scores are float probabilities: contract |delta| <= 1e-4 (the reference's own KAT
// tolerance is 1e-5); the arg-max over window sums is exact except for float near-ties.
#include <float.h>

#include "dmz_hip_internal.h"
#include "dmz_wave.h"

// developer ablation (tools/ablate.sh): extra dynamic LDS per workgroup = fewer workgroups per CU
#ifndef DMZ_LDS_PAD
#define DMZ_LDS_PAD 0
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int max3i(int a, int b, int c) { return imax(a, imax(b, c)); }
__device__ __forceinline__ int min3i(int a, int b, int c) { return imin(a, imin(b, c)); }

// tanh(x) = 1 - 2 / (exp(2x) + 1) on v_exp_f32 / v_rcp_f32: |error| ~ 2e-7 absolute
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);  // 2 * log2(e)
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// sum over the 16 lanes of a DPP row; the total lands in lane 15 of the row
__device__ __forceinline__ float row16_sum(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xf, 0xf, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x112, 0xf, 0xf, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x114, 0xf, 0xf, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x118, 0xf, 0xf, true));
  return x;
}

// developer ablation (tools/ablate.sh): return after phase k
#ifndef DMZ_VSEG_BLOCKS  /* workgroups per CU the register allocation aims at */
#define DMZ_VSEG_BLOCKS 7
#endif
#ifndef VS_SCAN_BATCH
#define VS_SCAN_BATCH 4
#endif
#ifndef DMZ_VSEG_STOP
#define DMZ_VSEG_STOP 99
#endif
#define VS_STOP(k, expr) if (DMZ_VSEG_STOP == (k)) { if (tid == 0) res->vseg_score = (float)(expr); return; }

constexpr int VS_THREADS = 256;
constexpr int VS_WAVES = 4;
constexpr int VS_MAXROWS = 68;   // coarse pass rows; the fine pass needs <= 43
constexpr int VS_PROWS = 68;     // rows of the per-wave partial-sum table
constexpr int VS_GSTRIDE = 224;  // gradient row stride in bytes (7 x 32 k-values, zero tail)
constexpr int VS_KSTEPS = 13;    // fp32 matrix core (model entry point): 13 x 16 = 208 >= 204
constexpr int VS_KS32 = 7;       // bf16 matrix core: 7 x 32 = 224 >= 204
constexpr int VS_RIF = 6;        // rows in flight per wave while loading

struct RowRaw {
  uint32_t w0, w1, w2;
};

// lane t (< 51) owns down-sampled outputs 4t..4t+3 = card columns 9+8t .. 20+8t of the row
// (lanes 51 .. 63 repeat lane 50: their gradients then never change the row's minimum or maximum, and the feature pass needs no
// "lane < 51" selects -- eight v_cndmask per row of a kernel that lives on its instruction count)
__device__ __forceinline__ RowRaw vseg_row_load(const uint8_t *__restrict__ row, int lane) {
  RowRaw r;
  const uint32_t *p = (const uint32_t *)(row + 8 + 8 * (lane < 50 ? lane : 50));
  r.w0 = p[0]; r.w1 = p[1]; r.w2 = p[2];
  return r;
}

// n_vseg.cpp:39-43 for one row: gradient, down-sample, and the min-max normalisation
// constants; writes 4 gradient bytes per lane and (scale, shift) of the row.
// Row i of `grad` is stored with bit 4 of the byte offset flipped when i & 8 (vs_swz): the matrix-core A fragments are
// ds_read_b64 at row * 224 + 8 kk + 32 ks, a half-wave covers 16 rows x 2 kk, and 224 B = 56 banks sends rows r and
// r + 8 to the same banks (a two-way conflict on every fragment read: SQ_LDS_BANK_CONFLICT was 1.44 x the kernel's
// active LDS cycles in round 2); with the flip rows 8..15 of a tile use the other half of each 32-byte group.
__device__ __forceinline__ int vs_swz(int row) { return (row & 8) << 1; }

// Round 5: the eight 3-tap gradients of a lane on PACKED 16-bit pairs.  With b0..b11 the lane's twelve bytes, output m needs
// a, b, c, e = b[1 + 2m .. 4 + 2m]: the four sequences over m = 0..3 are unpacked into two registers of two zero-extended
// bytes each (one v_perm_b32 per register; the replicate fix-ups at the ROI ends -- column 9 -> 10 on lane 0, column 418 ->
// 417 on lanes >= 50 -- are a different byte selector on those lanes, a per-lane constant: VsegSel), max(b, c) / min(b, c)
// are shared between the two gradients of an output, and everything up to the packed bytes is v_pk_*_u16: 38 instructions
// per row where the byte-at-a-time form took ~55.
struct VsegSel {
  uint32_t a01, e23;  // selectors of {b1, b3} (lane 0: {b2, b3}) and {b8, b10} (lanes >= 50: {b8, b9})
};
__device__ __forceinline__ VsegSel vseg_selectors(int lane) {
  VsegSel q;
  q.a01 = lane == 0 ? 0x0c030c02u : 0x0c030c01u;
  q.e23 = lane >= 50 ? 0x0c010c00u : 0x0c020c00u;
  return q;
}
typedef unsigned short vs_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ vs_u16x2 vs_pk(uint32_t v) { return __builtin_bit_cast(vs_u16x2, v); }
__device__ __forceinline__ uint32_t vs_u32(vs_u16x2 v) { return __builtin_bit_cast(uint32_t, v); }

__device__ __forceinline__ void vseg_row_features(const RowRaw &rw, const VsegSel &sel, unsigned char *__restrict__ grow, int swz,
                                                  float *__restrict__ norm /* 2 */, int lane) {
  // v_perm_b32(S0, S1, sel): selector byte 0..3 = byte of S1, 4..7 = byte of S0, 0x0c = zero
  const vs_u16x2 A[2] = {vs_pk(__builtin_amdgcn_perm(rw.w0, rw.w0, sel.a01)),       // {b1, b3}
                         vs_pk(__builtin_amdgcn_perm(rw.w1, rw.w1, 0x0c030c01u))};  // {b5, b7}
  const vs_u16x2 B[2] = {vs_pk(__builtin_amdgcn_perm(rw.w1, rw.w0, 0x0c040c02u)),   // {b2, b4}
                         vs_pk(__builtin_amdgcn_perm(rw.w2, rw.w1, 0x0c040c02u))};  // {b6, b8}
  const vs_u16x2 C[2] = {vs_pk(__builtin_amdgcn_perm(rw.w1, rw.w0, 0x0c050c03u)),   // {b3, b5}
                         vs_pk(__builtin_amdgcn_perm(rw.w2, rw.w1, 0x0c050c03u))};  // {b7, b9}
  const vs_u16x2 E[2] = {vs_pk(__builtin_amdgcn_perm(rw.w1, rw.w1, 0x0c020c00u)),   // {b4, b6}
                         vs_pk(__builtin_amdgcn_perm(rw.w2, rw.w2, sel.e23))};      // {b8, b10}
  const vs_u16x2 one = {1, 1};
  vs_u16x2 d[2];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const vs_u16x2 mx = __builtin_elementwise_max(B[h], C[h]), mn = __builtin_elementwise_min(B[h], C[h]);
    const vs_u16x2 g0 = __builtin_elementwise_max(A[h], mx) - __builtin_elementwise_min(A[h], mn);  // grad[2o]
    const vs_u16x2 g1 = __builtin_elementwise_max(mx, E[h]) - __builtin_elementwise_min(mn, E[h]);  // grad[2o+1]
    d[h] = (g0 + g1 + one) >> 1;
  }
  const vs_u16x2 lo2 = __builtin_elementwise_min(d[0], d[1]), hi2 = __builtin_elementwise_max(d[0], d[1]);
  int vmin = (int)(lo2.x < lo2.y ? lo2.x : lo2.y), vmax = (int)(hi2.x > hi2.y ? hi2.x : hi2.y);
  // (lanes >= 51 hold lane 50's values)
  vmin = 255 - (int)dmzwave::max_u32((unsigned)(255 - vmin));
  vmax = (int)dmzwave::max_u32((unsigned)vmax);
  if (lane < 56)  // lanes 51..55 write the zero k-tail 204..223
    *(uint32_t *)(grow + ((4 * lane) ^ swz)) = lane < 51 ? __builtin_amdgcn_perm(vs_u32(d[1]), vs_u32(d[0]), 0x06040200u) : 0u;
  // the row's (min, max) parked as two integers; vseg_row_norms turns them into (scale, shift) for many rows
  // at once (the fp64 division costs ~30 issue slots whether one lane or 64 need it)
  if (lane == 0) {
    norm[0] = __int_as_float(vmin);
    norm[1] = __int_as_float(vmax);
  }
}

// cvConvertScale(1/255) then cvNormalize(0,1,MINMAX) for the rows wave, wave + 4, ... this wave
// has just prepared: one lane per row
__device__ __forceinline__ void vseg_row_norms(float *__restrict__ norm, int nrows, int wave, int lane) {
  __builtin_amdgcn_wave_barrier();
  for (int i = wave + VS_WAVES * lane; i < nrows; i += VS_WAVES * 64) {
    const int vmin = __float_as_int(norm[2 * i]), vmax = __float_as_int(norm[2 * i + 1]);
    const float s255 = 1.0f / 255.0f;
    const double smin = (double)((float)vmin * s255), smax = (double)((float)vmax * s255);
    const double scale = (smax - smin > DBL_EPSILON) ? 1. / (smax - smin) : 0.;
    const double shift = 0.0 - smin * scale;
    norm[2 * i] = (float)scale;
    norm[2 * i + 1] = (float)shift;
  }
}

// wave `wave` prepares rows wave, wave+4, ... of the list row_y[0..nrows)
__device__ __forceinline__ void vseg_prepare_rows(const uint8_t *__restrict__ card,
                                                  const unsigned short *__restrict__ row_y, int nrows,
                                                  unsigned char *__restrict__ grad,
                                                  float *__restrict__ norm, int wave, int lane) {
  const VsegSel sel = vseg_selectors(lane);
  for (int i0 = wave; i0 < nrows; i0 += VS_WAVES * VS_RIF) {
    RowRaw raw[VS_RIF];
#pragma unroll
    for (int k = 0; k < VS_RIF; k++) {
      const int i = i0 + k * VS_WAVES;
      if (i < nrows) raw[k] = vseg_row_load(card + (size_t)row_y[i] * DMZ_CARD_WIDTH, lane);
    }
#pragma unroll
    for (int k = 0; k < VS_RIF; k++) {
      const int i = i0 + k * VS_WAVES;
      if (i < nrows) vseg_row_features(raw[k], sel, grad + i * VS_GSTRIDE, vs_swz(i), norm + 2 * i, lane);
    }
  }
  vseg_row_norms(norm, nrows, wave, lane);
}

// Hidden + logistic layers for up to 16 NT rows on v_mfma_f32_16x16x32_bf16 with EXACT operand splits.
// The feature of gradient byte d in a row with normalisation (s, t) is ((d / 255) s + t) (three float
// operations in the reference), so  sum_k W[j][k] f_k = s sum_k (W[j][k] / 255) d_k + t sum_k W[j][k].
// A byte zero-extended to 16 bits IS a bf16 number: d x 2^-133 (exponent fields 0 and 1 continue one linear scale, and
// the matrix core keeps subnormal inputs), so the A operand is one v_perm_b32 per two features -- no conversion at all
// (round 2 converted every byte to float and packed the upper halves: 1.5 instructions per feature against 0.5).
// W / 255 x 2^100 goes in three bf16 parts (24 bits: the parts of W / 255, their exponents moved), the 2^33 that is
// left comes back with the row's scale: powers of two, nothing rounds differently -- tools/ubench/mfma_f16_subnormal_dot.hip
// measures the same worst error as with the bytes as bf16 integers (6.2e-8 of the sum of |terms| per k-step; 1.3 % of the
// outputs differ in the last bit).  The same trick on the f16 matrix instruction (bytes = f16 subnormals d x 2^-24, weights
// x 2^12 in two or three f16 parts) was measured too: that instruction's accumulation is a little less exact (1.0e-7),
// and because a weight-side error is the same for every row it adds up linearly in the 27-row window sums -- max
// |vseg.score - oracle| 2.7e-5 instead of 7.6e-6; not used.  The products are exact in fp32 (8 x 8 bits): three matrix
// instructions per 32 k reproduce the fp32 product of the integer sums to ~2^-24 per term; the two row constants enter once
// per output.  (Against the reference this regroups three float roundings per feature: differences of the order of the 1e-6
// that any reordering of the 204-term sums makes; the 1e-4 contract on the scores and the proven-near-tie rule for y_offset
// are unchanged.)
// All NT row tiles go through one sweep over k, so a wave streams its 21 weight fragments (fragment order,
// 1-KB coalesced loads, prefetched one k-step ahead) once per pass.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 vs_h4 __attribute__((ext_vector_type(4)));
typedef _Float16 vs_h2 __attribute__((ext_vector_type(2)));

// What a lane needs of the layers behind the hidden product.  The hidden product is evaluated TRANSPOSED (weights as the A
// operand, gradient rows as B: the same register contents, the operands swapped), so a lane's four accumulators are four
// hidden units u = 16 wave + 4 (lane >> 4) + v of ONE card row (lane & 15) -- which is exactly the B operand of a
// v_mfma_f32_16x16x16_f16 whose contraction runs over the hidden units.  The logistic layer 50 -> 3 is then a matrix product
// too (round 5): out^T[c][row] = sum_u w2[c][u] h[u][row], with both operands split into an f16 rounding and the f16
// rounding of the remainder and the three products that carry 2^-22 (as the expiry CNN's convolutions).  Rounds 2 - 4 summed
// w2[c][u] h over the sixteen lanes of a DPP row: twelve v_add_f32_dpp and three multiplies per accumulator.
struct VsegTail {
  const float *unit_tab;  // LDS: rowsum[64] then b1[64] of the hidden units (zero beyond unit 49); the lane reads its four per tile
                          // (eight more resident registers spilled seven dwords at the kernel's 72)
  vs_h4 w2hi, w2lo;       // A operand of the logistic layer: row lane & 15 = class (three used), k = the lane's four units
};
__device__ __forceinline__ VsegTail vseg_tail_load(const float *__restrict__ wts, const float *unit_tab, int wave, int lane) {
  VsegTail q;
  q.unit_tab = unit_tab + 16 * wave + 4 * (lane >> 4);
  const int c = lane & 15, u0 = 16 * wave + 4 * (lane >> 4);
#pragma unroll
  for (int v = 0; v < 4; v++) {
    const int u = u0 + v;
    const bool unit = u < 50;
    const float w = (unit && c < 3) ? wts[dmzw::VSEG_W2 + c * 50 + u] : 0.0f;
    const _Float16 hi = (_Float16)w;
    q.w2hi[v] = hi;
    q.w2lo[v] = (_Float16)(w - (float)hi);
  }
  return q;
}

template <int NT>
__device__ __forceinline__ void vseg_mlp_rows_bf16(const bf16x8 *__restrict__ wb /* this wave + lane: [ks 7][part 3] x 64 */,
                                                  const VsegTail &tl, const unsigned char *__restrict__ grad,
                                                  const float *__restrict__ norm, int nrows,
                                                  float *__restrict__ part /* [4][VS_PROWS][3] */, int wave,
                                                  int lane) {
  int lane_here = lane;
  asm volatile("" : "+v"(lane_here));  // (round 6: 8 kk is recomputed here from the lane -- kept alive across the feature
                                       // phase it was the kernel's one spilled register)
  const int ii = lane_here & 15, kk = lane_here >> 4;
  const unsigned char *ap[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const int row = imin(t * 16 + ii, nrows - 1);
    ap[t] = grad + row * VS_GSTRIDE + ((8 * kk) ^ vs_swz(row));
  }
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 wn[3];
#pragma unroll
  for (int p = 0; p < 3; p++) wn[p] = wb[p * 64];
#pragma unroll 1
  for (int ks = 0; ks < VS_KS32; ks++) {
    bf16x8 w[3];
#pragma unroll
    for (int p = 0; p < 3; p++) {
      w[p] = wn[p];
      wn[p] = wb[(imin(ks + 1, VS_KS32 - 1) * 3 + p) * 64];
    }
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const uint2 by = *(const uint2 *)(ap[t] + 32 * ks);  // eight gradient bytes k = 32 ks + 8 kk ..
      u32x4 a;  // bytes zero-extended to 16 bits: the bf16 numbers d x 2^-133
      a.x = __builtin_amdgcn_perm(0u, by.x, 0x0c010c00u);
      a.y = __builtin_amdgcn_perm(0u, by.x, 0x0c030c02u);
      a.z = __builtin_amdgcn_perm(0u, by.y, 0x0c010c00u);
      a.w = __builtin_amdgcn_perm(0u, by.y, 0x0c030c02u);
      const bf16x8 av = __builtin_bit_cast(bf16x8, a);
      // (weights as A, rows as B: the transposed product)
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[2], av, acc[t], 0, 0, 0);  // small terms first
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], av, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], av, acc[t], 0, 0, 0);
    }
  }
  // D layout: row (hidden unit of this wave) = 4 * (lane >> 4) + v, column (card row of the tile) = lane & 15
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const int row = t * 16 + ii, rc = imin(row, nrows - 1);
    const float sc = norm[2 * rc] * 0x1p33f, sh = norm[2 * rc + 1];  // (A carries 2^100, B 2^-133)
    const f32x4 rs4 = *(const f32x4 *)tl.unit_tab, b14 = *(const f32x4 *)(tl.unit_tab + 64);
    float hv[4];
#pragma unroll
    for (int v = 0; v < 4; v++) hv[v] = fast_tanh(fmaf(sc, acc[t][v], fmaf(sh, rs4[v], b14[v])));  // units >= 50: tanh(0) = 0
    // h = hi + lo to 2^-22, both rounded to nearest (the packed conversion truncates: its errors all have one sign and add
    // up over the fifty units -- measured: 1.5 x the score error and a few more flipped near-ties on the fuzz frames)
    vs_h4 bhi, blo;
#pragma unroll
    for (int v = 0; v < 4; v++) {
      bhi[v] = (_Float16)hv[v];
      blo[v] = (_Float16)(hv[v] - (float)bhi[v]);
    }
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    o = __builtin_amdgcn_mfma_f32_16x16x16f16(tl.w2lo, bhi, o, 0, 0, 0);  // small terms first
    o = __builtin_amdgcn_mfma_f32_16x16x16f16(tl.w2hi, blo, o, 0, 0, 0);
    o = __builtin_amdgcn_mfma_f32_16x16x16f16(tl.w2hi, bhi, o, 0, 0, 0);
    // D layout again: row (class) = 4 * (lane >> 4) + v, column = card row: lanes 0..15 hold the three sums of their row
    if (kk == 0 && row < nrows) {
      float *p = part + (wave * VS_PROWS + row) * 3;
      p[0] = o[0]; p[1] = o[1]; p[2] = o[2];
    }
  }
}

// n_vseg.cpp:49-92, called by one wave.  The reference's ring buffer entry read at step y is the
// score of row y - 26, so the window is fed from the score arrays directly.  The two running
// sums (visa / amex) are literal float add / subtract chains -- one lane each (lane 0 the visa chain,
// lane 1 the amex chain) -- and the reference's
// scan "first strict maximum of v(26), a(26), v(27), a(27), ..." is an arg-max over the 488
// window sums with ties to the smallest scan position: exact, and parallel over the wave.
// Scores live in arrays padded to 288; `wsum` is 2 x 256 floats of scratch.
__device__ __forceinline__ void vseg_best_segmentation(const float *__restrict__ vis,
                                                       const float *__restrict__ amx, float *wsum,
                                                       int lane, float *score, int *y_off, int *pattern) {
  const int p = lane & 1;
  const float *src = p ? amx : vis;
  float *w = wsum + p * 256;
  float sum = 0.0f;
  // (lanes 0 and 1 only: 32 lanes storing the same value to one address are serialised by the LDS -- the two scans of a
  // card were 90 % of the kernel's SQ_LDS_BANK_CONFLICT cycles in round 2)
  if (lane < 2) {
  // rows 0..25: the window is not full yet
  for (int y0 = 0; y0 < 26; y0 += 13) {
    float v[13];
#pragma unroll
    for (int k = 0; k < 13; k++) v[k] = src[y0 + k];
#pragma unroll
    for (int k = 0; k < 13; k++) sum = sum + v[k];
  }
  // rows 26..269, VS_SCAN_BATCH at a time with the loads up front
  for (int y0 = 26; y0 < 270; y0 += VS_SCAN_BATCH) {
    float v[VS_SCAN_BATCH], ov[VS_SCAN_BATCH];
#pragma unroll
    for (int k = 0; k < VS_SCAN_BATCH; k++) {
      v[k] = src[imin(y0 + k, 287)];
      ov[k] = src[y0 + k - 26];
    }
#pragma unroll
    for (int k = 0; k < VS_SCAN_BATCH; k++) {
      sum = sum + v[k];
      if (y0 + k < 270) w[y0 + k - 26] = sum;
      sum = sum - ov[k];
    }
  }
  }
  __builtin_amdgcn_wave_barrier();
  float best = 0.0f;
  int bo = 0x7fffffff;  // scan position 2 * (y - 26) + (0 visa, 1 amex)
  for (int o = lane; o < 2 * 244; o += 64) {
    const float x = wsum[(o & 1) * 256 + (o >> 1)];
    if (x > best) { best = x; bo = o; }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const float ob = __shfl_down(best, d, 64);
    const int oo = __shfl_down(bo, d, 64);
    if (ob > best || (ob == best && oo < bo)) { best = ob; bo = oo; }
  }
  best = __shfl(best, 0, 64);
  bo = __shfl(bo, 0, 64);
  const bool any = bo != 0x7fffffff;
  *score = any ? best : 0.0f;
  *y_off = any ? bo >> 1 : 0;
  *pattern = any ? (bo & 1) + 1 : 0;
}

// logistic bias + softmax of the rows just evaluated (modelm_befe75da.cpp:1781-1784)
__device__ __forceinline__ void vseg_finish_rows(const float *__restrict__ wts,
                                                 const float *__restrict__ part,
                                                 const unsigned short *__restrict__ row_y, int nrows,
                                                 float *__restrict__ vis, float *__restrict__ amx,
                                                 int tid) {
  if (tid < nrows) {
    float o[3];
#pragma unroll
    for (int c = 0; c < 3; c++)
      o[c] = ((part[(0 * VS_PROWS + tid) * 3 + c] + part[(1 * VS_PROWS + tid) * 3 + c]) +
              (part[(2 * VS_PROWS + tid) * 3 + c] + part[(3 * VS_PROWS + tid) * 3 + c])) + wts[dmzw::VSEG_B2 + c];
    const float e0 = expf(o[0]), e1 = expf(o[1]), e2 = expf(o[2]);
    const float sum = e0 + (e1 + e2);  // Eigen 3-element redux tree
    const int y = row_y[tid];
    vis[y] = e1 / sum;
    amx[y] = e2 / sum;
  }
}

struct VsegWeights {
  f32x4 bw[VS_KSTEPS];
  float b1, w20, w21, w22;
};

__device__ __forceinline__ void vseg_load_weights(const float *__restrict__ wts, int wave, int lane,
                                                  VsegWeights &w) {
  const int j = 16 * wave + (lane & 15), kk = lane >> 4;
  const bool unit = j < 50;
#pragma unroll
  for (int u = 0; u < VS_KSTEPS; u++) {
    const int k0 = 16 * u + 4 * kk;
    if (unit && k0 < 204) w.bw[u] = *(const f32x4 *)(wts + dmzw::VSEG_W1 + j * 204 + k0);
    else w.bw[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  w.b1 = unit ? wts[dmzw::VSEG_B1 + j] : 0.0f;
  w.w20 = unit ? wts[dmzw::VSEG_W2 + 0 * 50 + j] : 0.0f;
  w.w21 = unit ? wts[dmzw::VSEG_W2 + 1 * 50 + j] : 0.0f;
  w.w22 = unit ? wts[dmzw::VSEG_W2 + 2 * 50 + j] : 0.0f;
}

__global__ __launch_bounds__(VS_THREADS, DMZ_VSEG_BLOCKS) void k_vseg(const float *__restrict__ wts,
                                                      const float *__restrict__ wfrag /* dmzv layout */,
                                                      const uint8_t *__restrict__ cards,
                                                      size_t card_stride, int n, int mode,
                                                      dmz_hip_frame_result *__restrict__ results) {
  __shared__ __attribute__((aligned(16))) unsigned char grad[VS_MAXROWS * VS_GSTRIDE];  // 14,144 B
  __shared__ float norm[2 * VS_MAXROWS];
  __shared__ float part[4 * VS_PROWS * 3];  // 3,264 B
  __shared__ float vis[288], amx[288];
  __shared__ unsigned short row_y[VS_MAXROWS];
  __shared__ int s_int[4];
  __shared__ __attribute__((aligned(16))) float unit_tab[128];

  const int f = blockIdx.x;
  if (f >= n) return;
  dmz_hip_frame_result *res = results + f;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int in_flags = res->flags;
  if ((mode & DMZ_HIP_SCAN_ONLY_WARPED) && !(in_flags & DMZ_HIP_FLAG_WARPED)) {
    // not rectified: no gate can have passed, whatever the caller's record held
    if (tid == 0) {
      res->flags = 0;
      res->vseg_score = 0.0f;
      res->vseg_y_offset = 0;
      res->pattern_type = 0;
      res->n_offsets = 0;
      res->hseg_score = 0.0f;
      res->number_width = 0.0f;
      res->pattern_offset = 0;
      res->number_score = 0.0f;
    }
    for (int i = tid; i < 160; i += VS_THREADS) (&res->scores[0][0])[i] = 0.0f;
    if (tid < 16) { res->digits[tid] = 0; res->offsets[tid] = 0; }
    return;
  }
  const uint8_t *card = cards + (size_t)f * card_stride;

  for (int i = tid; i < 288; i += VS_THREADS) { vis[i] = 0.0f; amx[i] = 0.0f; }
  // coarse pass: rows 0, 4, ..., 268 (n_vseg.cpp:116-125)
  for (int i = tid; i < VS_MAXROWS; i += VS_THREADS) row_y[i] = (unsigned short)(4 * i);
  __syncthreads();
  vseg_prepare_rows(card, row_y, VS_MAXROWS, grad, norm, wave, lane);
  // this wave's hidden units 16 wave .. + 15: weight fragments, row sum, bias and the logistic weights
  const bf16x8 *wb = (const bf16x8 *)(wfrag + dmzv::WB3) + wave * VS_KS32 * 3 * 64 + lane;
  if (tid < 128) {  // rowsum[64] | b1[64] of the hidden units
    const int u = tid & 63;
    unit_tab[tid] = tid < 64 ? wfrag[dmzv::ROWSUM + u] : (u < 50 ? wts[dmzw::VSEG_B1 + u] : 0.0f);
  }
  const VsegTail tl = vseg_tail_load(wts, unit_tab, wave, lane);
  __syncthreads();
  VS_STOP(1, grad[0] + norm[5])
  vseg_mlp_rows_bf16<5>(wb, tl, grad, norm, VS_MAXROWS, part, wave, lane);
  __syncthreads();
  VS_STOP(2, part[0] + part[100])
  vseg_finish_rows(wts, part, row_y, VS_MAXROWS, vis, amx, tid);
  __syncthreads();
  if (tid < 64) {
    float score;
    int y_off, pattern;
    vseg_best_segmentation(vis, amx, part, lane, &score, &y_off, &pattern);  // part is dead here
    // fine pass rows (n_vseg.cpp:140-152): at most 27 + 16 rows, one lane each
    int ymin = y_off < 8 ? 0 : y_off - 8;
    ymin = imin(270, ymin);
    const int ymax = imin(270, y_off + 27 + 8);
    const int y = ymin + lane;
    const bool fresh = y < ymax && vis[imin(y, 287)] == 0.0f && amx[imin(y, 287)] == 0.0f;
    const unsigned long long m = __ballot(fresh);
    if (fresh) row_y[__popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)y;
    if (lane == 0) s_int[0] = __popcll(m);
  }
  __syncthreads();
  VS_STOP(3, s_int[0] + vis[100])
  const int nfine = s_int[0];
  if (nfine > 0) {
    vseg_prepare_rows(card, row_y, nfine, grad, norm, wave, lane);
    __syncthreads();
    // (43 rows around the coarse winner minus the 10 or 11 the coarse pass scored: 32 fresh rows in three cards of four -- two tiles)
    if (nfine <= 32) vseg_mlp_rows_bf16<2>(wb, tl, grad, norm, nfine, part, wave, lane);
    else vseg_mlp_rows_bf16<3>(wb, tl, grad, norm, nfine, part, wave, lane);
    __syncthreads();
    vseg_finish_rows(wts, part, row_y, nfine, vis, amx, tid);
    __syncthreads();
  }
  VS_STOP(4, vis[150] + amx[3])
  if (tid < 64) {
    float score;
    int y_off, pattern;
    vseg_best_segmentation(vis, amx, part, lane, &score, &y_off, &pattern);
  if (tid == 0) {
    int flags = in_flags & DMZ_HIP_FLAG_WARPED;
    if (y_off < (DMZ_CARD_HEIGHT - 27) / 2) flags |= DMZ_HIP_FLAG_UPSIDE_DOWN;  // frame.cpp:38
    else if (score > 15.0f)                                                      // frame.cpp:43-49
      flags |= DMZ_HIP_FLAG_VSEG_OK | ((mode & DMZ_HIP_SCAN_SKIP_NUMBER) ? DMZ_HIP_FLAG_USABLE : 0);
    res->vseg_score = score;
    res->vseg_y_offset = y_off;
    res->pattern_type = pattern;
    res->flags = flags;
    // defaults of the later stages (frames that stop here)
    res->n_offsets = 0;
    res->hseg_score = 0.0f;
    res->number_width = 0.0f;
    res->pattern_offset = 0;
    res->number_score = 0.0f;
  }
  }
  // clear the per-digit outputs (NumberScores::Zero(), n_categorize.cpp:93)
  for (int i = tid; i < 160; i += VS_THREADS) (&res->scores[0][0])[i] = 0.0f;
  if (tid < 16) { res->digits[tid] = 0; res->offsets[tid] = 0; }
}

// Stand-alone model entry point (KAT): up to 16 input vectors per workgroup, same MFMA
// path fed with float features.
__global__ __launch_bounds__(VS_THREADS) void k_vseg_model(const float *__restrict__ wts,
                                                            const float *__restrict__ x, int n,
                                                            float *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) float feat[16 * 208];
  __shared__ float part[4 * 80 * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int base = blockIdx.x * 16;
  const int rows = imin(16, n - base);
  if (rows <= 0) return;
  VsegWeights w;
  vseg_load_weights(wts, wave, lane, w);
  for (int i = tid; i < 16 * 208; i += VS_THREADS) {
    const int r = i / 208, k = i - r * 208;
    feat[i] = (r < rows && k < 204) ? x[(size_t)(base + r) * 204 + k] : 0.0f;
  }
  __syncthreads();
  {
    const int ii = lane & 15, kk = lane >> 4;
    const float *ap = feat + imin(ii, rows - 1) * 208 + 4 * kk;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < VS_KSTEPS; u++) {
      const f32x4 a = *(const f32x4 *)(ap + 16 * u);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w.bw[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w.bw[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w.bw[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w.bw[u].w, acc, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const float hv = fast_tanh(acc[v] + w.b1);
      const float o0 = row16_sum(w.w20 * hv), o1 = row16_sum(w.w21 * hv), o2 = row16_sum(w.w22 * hv);
      const int row = 4 * kk + v;
      if (ii == 15 && row < rows) {
        float *p = part + (wave * 80 + row) * 4;
        p[0] = o0; p[1] = o1; p[2] = o2;
      }
    }
  }
  __syncthreads();
  if (tid < rows) {
    float o[3];
    for (int c = 0; c < 3; c++)
      o[c] = ((part[(0 * 80 + tid) * 4 + c] + part[(1 * 80 + tid) * 4 + c]) +
              (part[(2 * 80 + tid) * 4 + c] + part[(3 * 80 + tid) * 4 + c])) + wts[dmzw::VSEG_B2 + c];
    const float e0 = expf(o[0]), e1 = expf(o[1]), e2 = expf(o[2]);
    const float sum = e0 + (e1 + e2);
    out[(size_t)(base + tid) * 3 + 0] = e0 / sum;
    out[(size_t)(base + tid) * 3 + 1] = e1 / sum;
    out[(size_t)(base + tid) * 3 + 2] = e2 / sum;
  }
}

}  // namespace

void dmz_launch_vseg(hipStream_t s, const float *weights, const float *wfrag, const uint8_t *cards, size_t card_stride,
                     int n, int mode, dmz_hip_frame_result *results) {
  DMZ_REPEAT(vseg)
  hipLaunchKernelGGL(k_vseg, dim3(n), dim3(VS_THREADS), DMZ_LDS_PAD, s, weights, wfrag, cards, card_stride, n,
                     mode, results);
}

void dmz_launch_vseg_model(hipStream_t s, const float *weights, const float *x, int n, float *out) {
  hipLaunchKernelGGL(k_vseg_model, dim3((n + 15) / 16), dim3(VS_THREADS), 0, s, weights, x, n, out);
}

int dmz_configure_vseg(void) { return 0; }
