// hseg.hip -- digit x-offset search on the 428 x 27 number strip of each card.
//
// Replaces best_n_hseg (scan/n_hseg.cpp:88-151) and best_n_hseg_constrained
// (n_hseg.cpp:39-84): 5-tap cross morphological gradient of the strip clamped at the
// strip edge (cv/morph.cpp:190-220), column sums (cvReduce), min-max normalisation
// (cvNormalize), then four passes over (digit width, pattern offset)
// candidates enumerated with the reference's float loop increments; each candidate's
// score is the L1 distance between the column profile and the 19-tap digit template
// pasted at every digit position, summed sequentially over the 428 columns (the scalar
// Eigen order), and the strict-< winner in iteration order is kept.  hseg.score,
// number_width and the offsets are bit-exact.
//
// CDNA4 mapping: one wave per card, one lane per candidate.
//   * column sums straight from HBM/L2: a lane owns 4 adjacent columns (one dword per
//     row, plus its two neighbour dwords), walks the 27 rows with three rows in
//     registers; no strip in LDS (LDS holds only the 428 normalised floats), so up to 32
//     cards are resident per CU and their serial sums hide each other's latency.
//   * the sequential sum is organised by digit: for digit k every lane adds the columns
//     [c_k, c_{k+1}) in order (the same column order as 0..427), so the template tap index
//     j is wave-uniform (template values are instruction literals) and only the LDS
//     address differs per lane; lanes whose segment is shorter add +0.0f.
#include <float.h>

#include "dmz_hip_internal.h"
#include "dmz_wave.h"

namespace {

__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int max5(int a, int b, int c, int d, int e) { return imax(imax(a, b), imax(imax(c, d), e)); }
__device__ __forceinline__ int min5(int a, int b, int c, int d, int e) { return imin(imin(a, b), imin(imin(c, d), e)); }

// n_vseg.cpp:26-30 tables: number length, pattern length, digit-slot bit mask (bit pi set
// when number_pattern[pi] == 1)
__device__ __forceinline__ int number_len(int pt) { return pt == 1 ? 16 : (pt == 2 ? 15 : 0); }
__device__ __forceinline__ int pattern_len(int pt) { return pt == 1 ? 19 : (pt == 2 ? 17 : 0); }
__device__ __forceinline__ unsigned pattern_mask(int pt) {
  // visa-like  1111 0 1111 0 1111 0 1111       amex-like 1111 0 111111 0 11111 00
  return pt == 1 ? 0x7BDEFu : (pt == 2 ? 0x1F7EFu : 0u);
}

// n_hseg.cpp:15-20 (data): the 19-tap digit gradient template
#define HSEG_T(j)                                                                                   \
  ((j) == 0 ? 0.26228655f : (j) == 1 ? 0.30289554f : (j) == 2 ? 0.34632607f : (j) == 3 ? 0.38725636f \
   : (j) == 4 ? 0.42745813f : (j) == 5 ? 0.45875135f : (j) == 6 ? 0.46498017f : (j) == 7 ? 0.45258447f \
   : (j) == 8 ? 0.43045216f : (j) == 9 ? 0.42430462f : (j) == 10 ? 0.44796554f : (j) == 11 ? 0.47726529f \
   : (j) == 12 ? 0.48471646f : (j) == 13 ? 0.46457738f : (j) == 14 ? 0.42799847f : (j) == 15 ? 0.38851183f \
   : (j) == 16 ? 0.33966308f : (j) == 17 ? 0.28802608f : 0.25377602f)

__device__ __forceinline__ int wave_max(int v) { return (int)dmzwave::max_u32((unsigned)v); }  // v >= 0

// the pattern-offset range and count of one width iteration (n_hseg.cpp:46-53)
__device__ __forceinline__ int offsets_for_width(int plen, float width, int omin, int omax, int ostep) {
  const float pw = (float)plen * width;
  unsigned short pom = (unsigned short)omax;
  const unsigned short maxo = (unsigned short)(428 - __float2int_rn(pw));
  if (pom == 0xFFFF || pom > maxo) pom = maxo;
  return ((int)pom > omin) ? ((int)pom - omin + ostep - 1) / ostep : 0;
}

// centre (left column) of the digit in pattern slot pi (n_hseg.cpp:60)
__device__ __forceinline__ int slot_center(int off, int pi, float width) {
  return (int)(unsigned short)(off + __float2int_rn((float)pi * width));
}

// L1 score of one candidate, sequential in column order.  g has 64 zero floats of padding
// behind column 427 so that inactive lanes may read past their segment.
// PT (the number pattern) is a template parameter: the slots of the pattern are then compile-time constants, the digit
// loop unrolls and every digit's left column is computed ONCE (the rolled form evaluated slot_center three times per slot:
// in the bounds test, as a segment's start and as its predecessor's end).
template <int PT>
__device__ float hseg_score_t(const float *__restrict__ g, float width, int off, bool has) {
  constexpr int plen = PT == 1 ? 19 : 17;
  constexpr unsigned mask = PT == 1 ? 0x7BDEFu : 0x1F7EFu;
  constexpr int nd = PT == 1 ? 16 : 15;
  int c[nd];
  {
    int k = 0;
#pragma unroll
    for (int pi = 0; pi < plen; pi++)
      if ((mask >> pi) & 1u) c[k++] = slot_center(off, pi, width);
  }
  // in-bounds test (n_hseg.cpp:61-66)
  bool in_bounds = has;
#pragma unroll
  for (int k = 0; k < nd; k++)
    if (!(c[k] + 19 < 428)) in_bounds = false;
  const bool live = in_bounds;
  const int first = c[0];
  float s = 0.0f;
  // A term that lies outside the lane's segment must leave the sum as it is.  The reference's "j < len ? a : 0" costs a compare
  // and a select per term (both half-rate instructions, profiles/r4_valu_table_gfx950.txt); here the term is multiplied by
  // m = clamp(len - j) = 1 inside / 0 outside INSIDE the addition: fma(|a|, m, s) rounds a + s once, i.e. it IS s + a
  // for m = 1 and s for m = 0 (a is finite: g is padded with zeros) -- the same bits from full-rate instructions only.
  // (v_add_f32 with the clamp modifier, spelled out: the compiler turns min(max(x, 0), 1) into v_max_f32 ... clamp, a half-rate opcode)
  auto clamp01 = [](float x) {
    float m;
    asm("v_add_f32_e64 %0, %1, 0 clamp" : "=v"(m) : "v"(x));
    return m;
  };
  // leading gap: columns [0, first) against pattern value 0
  {
    const int len = live ? first : 0;
    const int mx = wave_max(len);
    float rem = (float)len;  // len - j
    for (int j = 0; j < mx; j += 8) {  // (eight reads in flight; g is padded, and past the segment the mask is 0)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = g[j + u];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        s = __builtin_fmaf(fabsf(v[u]), clamp01(rem), s);
        rem -= 1.0f;
      }
    }
  }
  // digit segments in slot order; segment k covers [c_k, c_next) (c_next = 428 for the last)
#pragma unroll
  for (int k = 0; k < nd; k++) {
    const int cn = k + 1 < nd ? c[k + 1] : 428;
    const int len = live ? cn - c[k] : 0;
    const float lenf = (float)len;
    const float *gp = g + (live ? c[k] : 0);
    // the first 16 taps are inside every live lane's segment (digit spacing >= 16); the sum of a
    // lane that is not live is discarded below, so it needs no masking here
#pragma unroll
    for (int j = 0; j < 16; j++) s = s + fabsf(gp[j] - HSEG_T(j));
    // clamp(len - 16), clamp(len - 17), clamp(len - 18): one v_add_f32 each (-1.0 and -2.0 are inline constants); written out
    // because the compiler rewrites (float)len - (float)j as an integer subtraction and a half-rate conversion per term
    float x16, m16, m17, m18;
    asm("v_add_f32_e32 %0, 0xc1800000, %1" : "=v"(x16) : "v"(lenf));  // len - 16
    asm("v_add_f32_e64 %0, %1, 0 clamp" : "=v"(m16) : "v"(x16));
    asm("v_add_f32_e64 %0, %1, -1.0 clamp" : "=v"(m17) : "v"(x16));
    asm("v_add_f32_e64 %0, %1, -2.0 clamp" : "=v"(m18) : "v"(x16));
    s = __builtin_fmaf(fabsf(gp[16] - HSEG_T(16)), m16, s);
    s = __builtin_fmaf(fabsf(gp[17] - HSEG_T(17)), m17, s);
    s = __builtin_fmaf(fabsf(gp[18] - HSEG_T(18)), m18, s);
    const int mx = wave_max(len);
    float rem;
    asm("v_add_f32_e32 %0, 0xc0400000, %1" : "=v"(rem) : "v"(x16));  // len - 19
    for (int j = 19; j < mx; j += 4) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = gp[j + u];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        s = __builtin_fmaf(fabsf(v[u]), clamp01(rem), s);
        rem -= 1.0f;
      }
    }
  }
  return live ? s : FLT_MAX;
}
__device__ __forceinline__ float hseg_score(const float *__restrict__ g, int pt, float width, int off, bool has) {
  // (pattern type 0 never reaches the search: VSEG_OK implies a pattern)
  return pt == 2 ? hseg_score_t<2>(g, width, off, has) : hseg_score_t<1>(g, width, off, has);
}

struct HsegBest {
  float score;
  float width;
  int offset;
};

// one pass of best_n_hseg_constrained; every lane carries an identical copy of `best`
__device__ void hseg_pass(const float *__restrict__ g, int pt, float wmin, float wmax, float wstep,
                          int omin, int omax, int ostep, HsegBest &best, int lane) {
  const int plen = pattern_len(pt);
  int total = 0;
  for (float width = wmin; width < wmax; width += wstep)
    total += offsets_for_width(plen, width, omin, omax, ostep);
  for (int base = 0; base < total; base += 64) {
    const int my = base + lane;
    // locate candidate `my` in the reference's iteration order (width outer, offset inner)
    float my_w = 0.0f;
    int my_off = 0, idx = 0;
    bool has = false;
    for (float width = wmin; width < wmax; width += wstep) {
      const int cnt = offsets_for_width(plen, width, omin, omax, ostep);
      if (!has && my >= idx && my < idx + cnt) {
        has = true;
        my_w = width;
        my_off = omin + (my - idx) * ostep;
      }
      idx += cnt;
    }
    const float score = hseg_score(g, pt, my_w, my_off, has);
    // wave arg-min; ties -> earliest candidate (strict < in iteration order)
    // (scores are non-negative floats: their bit patterns order like the values)
    const unsigned long long key = dmzwave::min_u64(__float_as_uint(score), (unsigned int)my);
    const float smin = __uint_as_float((unsigned int)(key >> 32));
    const int winner = (int)(key & 0xffffffffu);
    if (smin < best.score) {
      const int wl = winner - base;  // lane that holds the winner
      best.score = smin;
      best.width = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), wl));
      best.offset = __builtin_amdgcn_readlane(my_off, wl);
    }
  }
}

__global__ __launch_bounds__(64) void k_hseg(const uint8_t *__restrict__ cards, size_t card_stride,
                                              int n, dmz_hip_frame_result *__restrict__ results) {
  __shared__ float g[428 + 64];
  __shared__ int colsum[428];

  const int f = blockIdx.x;
  if (f >= n) return;
  dmz_hip_frame_result *res = results + f;
  if (!(res->flags & DMZ_HIP_FLAG_VSEG_OK)) return;
  const int lane = threadIdx.x;
  const int y_off = res->vseg_y_offset;
  const int pt = res->pattern_type;
  const uint32_t *strip = (const uint32_t *)(cards + (size_t)f * card_stride + (size_t)y_off * DMZ_CARD_WIDTH);

  // ---- cross gradient clamped at the strip (ROI) edge + column sums (n_hseg.cpp:90-95) ----
  int lmin = 1 << 30, lmax = -1;
  for (int q = lane; q < 107; q += 64) {  // dword q = columns 4q .. 4q+3
    int sum0 = 0, sum1 = 0, sum2 = 0, sum3 = 0;
    uint32_t up, mid, dn, lw, rw;
    mid = strip[q];
    up = mid;  // row -1 replicates row 0
    for (int r = 0; r < 27; r++) {
      const uint32_t *row = strip + r * 107;
      dn = r < 26 ? row[107 + q] : mid;  // row 27 replicates row 26
      lw = q > 0 ? row[q - 1] : 0u;
      rw = q < 106 ? row[q + 1] : 0u;
      const int c0 = mid & 255, c1 = (mid >> 8) & 255, c2 = (mid >> 16) & 255, c3 = mid >> 24;
      const int wl = q > 0 ? (int)(lw >> 24) : c0;          // column -1 replicates column 0
      const int er = q < 106 ? (int)(rw & 255) : c3;        // column 428 replicates column 427
      const int n0 = up & 255, n1 = (up >> 8) & 255, n2 = (up >> 16) & 255, n3 = up >> 24;
      const int s0 = dn & 255, s1 = (dn >> 8) & 255, s2 = (dn >> 16) & 255, s3 = dn >> 24;
      sum0 += max5(n0, wl, c0, c1, s0) - min5(n0, wl, c0, c1, s0);
      sum1 += max5(n1, c0, c1, c2, s1) - min5(n1, c0, c1, c2, s1);
      sum2 += max5(n2, c1, c2, c3, s2) - min5(n2, c1, c2, c3, s2);
      sum3 += max5(n3, c2, c3, er, s3) - min5(n3, c2, c3, er, s3);
      up = mid;
      mid = dn;
    }
    colsum[4 * q + 0] = sum0; colsum[4 * q + 1] = sum1; colsum[4 * q + 2] = sum2; colsum[4 * q + 3] = sum3;
    lmin = imin(lmin, imin(imin(sum0, sum1), imin(sum2, sum3)));
    lmax = imax(lmax, imax(imax(sum0, sum1), imax(sum2, sum3)));
  }
  lmin = (int)dmzwave::min_u32((unsigned)lmin);  // column sums are >= 0; idle lanes hold 2^30 / -1 -> 0
  lmax = (int)dmzwave::max_u32((unsigned)imax(lmax, 0));
  __syncthreads();
  {
    // cvNormalize(0,1,MINMAX) on the float sums
    const double smin = (double)(float)lmin, smax = (double)(float)lmax;
    const double scale = (smax - smin > DBL_EPSILON) ? 1. / (smax - smin) : 0.;
    const double shift = 0.0 - smin * scale;
    const float fs = (float)scale, fb = (float)shift;
    for (int c = lane; c < 428 + 64; c += 64) g[c] = c < 428 ? (float)colsum[c] * fs + fb : 0.0f;
  }
  __syncthreads();

  HsegBest best;
  best.score = 428.0f;
  best.width = 0.0f;
  best.offset = 0;
  hseg_pass(g, pt, 17.1f, 19.7f, 0.5f, 0, 0xFFFF, 10, best, lane);
  {
    const int po = best.offset;
    hseg_pass(g, pt, best.width - 0.5f, best.width + 0.5f, 0.2f, po < 10 ? 0 : po - 10, po + 10, 1, best, lane);
  }
  {
    const int po = best.offset;
    hseg_pass(g, pt, best.width - 0.2f, best.width + 0.2f, 0.1f, po < 3 ? 0 : po - 3, po + 3, 1, best, lane);
  }
  {
    const int po = best.offset;
    hseg_pass(g, pt, best.width - 0.1f, best.width + 0.1f, 0.05f, po < 3 ? 0 : po - 3, po + 3, 1, best, lane);
  }
  if (lane == 0) {
    res->n_offsets = number_len(pt);
    res->hseg_score = best.score;
    res->number_width = best.width;
    res->pattern_offset = best.offset;
  }
  if (lane < 16) {
    // offsets of the winner (n_hseg.cpp:60,68,75); unused slots are 0.  A search that never
    // improved on the initial 428 keeps the zero offsets of n_hseg.cpp:104.
    int value = 0;
    if (best.score < 428.0f) {
      const unsigned mask = pattern_mask(pt);
      int k = 0;
      for (int pi = 0; pi < pattern_len(pt); pi++) {
        if (!((mask >> pi) & 1u)) continue;
        if (k == lane) value = slot_center(best.offset, pi, best.width);
        k++;
      }
    }
    res->offsets[lane] = (unsigned short)value;
  }
}

}  // namespace

void dmz_launch_hseg(hipStream_t s, const uint8_t *cards, size_t card_stride, int n,
                     dmz_hip_frame_result *results) {
  hipLaunchKernelGGL(k_hseg, dim3(n), dim3(64), 0, s, cards, card_stride, n, results);
}
