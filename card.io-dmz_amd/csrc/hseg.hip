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
//   * column sums straight from HBM/L2: a lane owns 4 adjacent columns (one dword per row), all 27 row dwords of a 64-lane
//     pass are requested together, the neighbour columns come from the neighbour lanes (DPP wave shifts); no strip in LDS
//     (LDS holds the 428 normalised floats and the filter's table: 8.8 KB, 18 cards per CU).
//   * the ordered form of a pass (lane = candidate): the sequential sum is organised by digit -- for digit k every lane adds
//     the columns [c_k, c_{k+1}) in order (the same column order as 0..427), so the template tap index j is wave-uniform
//     (template values are instruction literals) and only the LDS address differs per lane; lanes whose segment is shorter
//     add +0.0f.  Since round 4 only the passes the filter cannot decide run in this form.
//   * round 4: the search is FILTERED.  A candidate's score is sum_c |g_c| + sum_digits W_L[c_k] with
//     W_L[c] = sum_{j<L} (|g_{c+j} - T_j| - |g_{c+j}|) (L = the digit's segment length, 16..19): one table of 428 x 4 floats
//     per card turns a candidate into sixteen LDS reads and additions instead of 428 ordered terms.  That value is within a
//     proven distance of the reference's ordered float sum ("the filtered search" below), so whenever the best candidate leads every
//     candidate with OTHER digit positions by more than that distance the pass is decided without any ordered sum; otherwise
//     (measured: 1.1 % of the passes) the pass runs in its ordered form.  The winner's ordered sum is evaluated once at the end:
//     hseg.score keeps the reference's bits.
#include <float.h>

#include "dmz_hip_internal.h"
#include "dmz_wave.h"

namespace {

constexpr int kHsegG = 428 + 164;  // the normalised gradient sums + the zero padding the masked tail reads reach (k_hseg)

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

// developer counters (-DDMZ_HSEG_DBG; tools/dev/hseg_dbg.py): passes, passes that ran in the ordered form
#ifdef DMZ_HSEG_DBG
__device__ unsigned long long g_hs_dbg[4];
#endif

// developer probe (-DDMZ_HSEG_TIMING): the phase timeline of one wave
#ifdef DMZ_HSEG_TIMING
#define HS_T(i) if (blockIdx.x == gridDim.x / 2) hs_t[i] = (long long)__builtin_readcyclecounter();
#else
#define HS_T(i)
#endif

struct HsegBest {
  float score;     // the reference's ordered float sum of the incumbent -- valid when `exact`
  float width;
  int offset;
  // filtered search: the incumbent's table score and digit-position signature
  float approx;
  int sig_off;     // -1 = the initial incumbent (score 428, no positions)
  unsigned sig_a, sig_b;
  bool exact, improved;
  bool fit;        // the incumbent has a table score (false: every later pass runs in the ordered form)
};

// one pass of best_n_hseg_constrained in its ordered form; every lane carries an identical copy of `best`
__device__ __forceinline__ HsegBest hseg_pass(const float *__restrict__ g, int pt, float wmin, float wmax, float wstep,
                                           int omin, int omax, int ostep, HsegBest best, int lane) {
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
      best.improved = true;
    }
  }
  return best;
}

// ---- the filtered search ---------------------------------------------------------------------------------------------------
// Notation: a_c = |g_c|, b_{c,j} = fl(|g_{c+j} - T_j|) (the reference's float term), S* = the REAL sum of a candidate's 428 float
// terms, S_ref = the reference's ordered float sum of them, S^ = the table score below.  u = 2^-24.
//   |S_ref - S*| <= gamma S*, gamma = 429 u (427 additions of non-negative terms: (1 + u)^427 - 1 < 428.01 u).
//   |S^ - S*|   <= delta = u (4500 + 30 G):  a table entry is the float sum of <= 19 differences fl(b - a), each within u |b - a|
//                  and |b - a| <= T_j, so an entry is within 20 u sum(T) < 151 u of its real value; G = sum a_c by seven additions per
//                  lane and a six-level tree is within 13 u G; the sixteen additions G + W + W + ... have partial sums below
//                  G + 16 sum(T) < G + 121: 16 u (G + 121).  Together u (16 x 151 + 13 G + 16 G + 1936) < u (4500 + 30 G).
// Candidate i cannot beat candidate j in the reference (S_ref,i > S_ref,j) when S*_i (1 - gamma) > S*_j (1 + gamma), which
// holds when  S^_i > S^_j + 2 delta + 2.1 gamma (S^_j + delta).  With j = the smallest table score of the pass (the incumbent
// included) every candidate above that threshold is out; if the ones below it all have the same digit positions they have the
// same terms, hence the same S_ref, and the reference keeps the earliest of them (the incumbent first: strict <).  Otherwise the
// pass is repeated in the ordered form.
struct HsegFilter {
  float G;       // sum of |g_c|
  float e1, e2;  // threshold = m + e1 + e2 m
};

// The table score of candidate (width, off) and its digit-position signature (the offset apart).  For widths in [16.05, 23.5) --
// the caller checks the pass' range -- neighbouring digits are 16 .. 24 columns apart and digits across a gap 32 .. 47, so the
// positions grow with the slot (the last digit decides the bounds test of n_hseg.cpp:61), a spacing less its smallest value fits
// four bits, and a digit's segment has 16 .. 19 template taps (19 across a gap and for the last digit).
template <int PT>
__device__ __forceinline__ float hseg_table_score_t(const float *__restrict__ W, float G, float width, int off, bool has,
                                                    unsigned &sig_a, unsigned &sig_b) {
  constexpr int plen = PT == 1 ? 19 : 17;
  constexpr unsigned mask = PT == 1 ? 0x7BDEFu : 0x1F7EFu;
  constexpr int nd = PT == 1 ? 16 : 15;
  int r[nd];    // digit k's column less the offset (n_hseg.cpp:60)
  int dpi[nd];  // pattern slots between digit k and digit k + 1 (compile-time)
  {
    int k = 0, last = 0;
#pragma unroll
    for (int pi = 0; pi < plen; pi++)
      if ((mask >> pi) & 1u) {
        r[k] = pi == 0 ? 0 : __float2int_rn((float)pi * width);
        if (k > 0) dpi[k - 1] = pi - last;
        last = pi;
        k++;
      }
    dpi[nd - 1] = 2;
  }
  float s = G;
  unsigned sa = 0u, sb = 0u, base_a = 0u, base_b = 0u;
#pragma unroll
  for (int k = 0; k < nd; k++) {
    int L = 19;
    if (k + 1 < nd) {
      const int sp = r[k + 1] - r[k];
      if (dpi[k] == 1) L = imin(sp, 19);
      // sum of (spacing << 4 k); less the same sum of the smallest spacings it is the packed four-bit differences
      if (k < 8) sa += (unsigned)sp << (4 * k), base_a += (unsigned)(16 * dpi[k]) << (4 * k);
      else sb += (unsigned)sp << (4 * (k - 8)), base_b += (unsigned)(16 * dpi[k]) << (4 * (k - 8));
    }
    s = s + W[(off + r[k]) * 4 + (L - 16)];
  }
  sig_a = sa - base_a;
  sig_b = sb - base_b;
  return (has && off + r[nd - 1] + 19 < 428) ? s : __builtin_inff();
}
__device__ __forceinline__ float hseg_table_score(const float *__restrict__ W, float G, int pt, float width, int off, bool has,
                                                  unsigned &sig_a, unsigned &sig_b) {
  return pt == 2 ? hseg_table_score_t<2>(W, G, width, off, has, sig_a, sig_b)
                 : hseg_table_score_t<1>(W, G, width, off, has, sig_a, sig_b);
}

// The winner's ordered sum at the end of the search, when the table is no longer needed: the 428 terms in parallel (the pattern is
// pasted into LDS digit by digit in slot order, a later digit over an earlier one's tail: the memcpy of n_hseg.cpp:62-63), then
// one chain of 428 additions over them -- four terms per LDS read, every lane the same chain (~600 instructions where the
// lane-per-candidate form spends ~1 400 on its one live lane).
__device__ const float k_hseg_template[19] = {0.26228655f, 0.30289554f, 0.34632607f, 0.38725636f, 0.42745813f, 0.45875135f,
                                              0.46498017f, 0.45258447f, 0.43045216f, 0.42430462f, 0.44796554f, 0.47726529f,
                                              0.48471646f, 0.46457738f, 0.42799847f, 0.38851183f, 0.33966308f, 0.28802608f,
                                              0.25377602f};
template <int PT>
__device__ __forceinline__ float hseg_ordered_chain_t(const float *__restrict__ g, float *__restrict__ pat /* 448 floats */,
                                                      float width, int off, int lane) {
  constexpr int plen = PT == 1 ? 19 : 17;
  constexpr unsigned mask = PT == 1 ? 0x7BDEFu : 0x1F7EFu;
  const float tl = k_hseg_template[lane < 19 ? lane : 18];
#pragma unroll
  for (int i = 0; i < 7; i++) pat[lane + 64 * i] = 0.0f;
#pragma unroll
  for (int pi = 0; pi < plen; pi++)
    if ((mask >> pi) & 1u) {
      const int c = slot_center(off, pi, width);  // (in bounds: the winner passed the test of n_hseg.cpp:61)
      if (lane < 19) pat[c + lane] = tl;
    }
  __syncthreads();  // (one wave, but lanes read what other lanes wrote: without it the compiler forwards a lane's own zero)
#pragma unroll
  for (int i = 0; i < 7; i++) {
    const int c = lane + 64 * i;  // (g is padded with zeros behind column 427)
    pat[c] = g[c] - pat[c];
  }
  __syncthreads();
  float s = 0.0f;  // (0 + |t_0| = |t_0|: the reference starts from the first term)
  const float4 *p4 = (const float4 *)pat;
  for (int q = 0; q < 104; q += 4) {
    float4 t[4];
#pragma unroll
    for (int u = 0; u < 4; u++) t[u] = p4[q + u];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      s = s + fabsf(t[u].x);
      s = s + fabsf(t[u].y);
      s = s + fabsf(t[u].z);
      s = s + fabsf(t[u].w);
    }
  }
#pragma unroll
  for (int q = 104; q < 107; q++) {
    const float4 t = p4[q];
    s = s + fabsf(t.x);
    s = s + fabsf(t.y);
    s = s + fabsf(t.z);
    s = s + fabsf(t.w);
  }
  return s;
}

// the incumbent's ordered sum, where a pass needs it (one lane's worth of work)
__device__ __noinline__ float hseg_ordered_score(const float *__restrict__ g, int pt, float width, int offset, int lane) {
  const float sc = hseg_score(g, pt, width, offset, lane == 0);
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sc)));
}
__device__ __forceinline__ void hseg_make_exact(const float *__restrict__ g, int pt, HsegBest &best, int lane) {
  if (best.exact) return;
  best.score = hseg_ordered_score(g, pt, best.width, best.offset, lane);
  best.exact = true;
}

// one pass of best_n_hseg_constrained, filtered
__device__ __forceinline__ void hseg_pass_filtered(const float *__restrict__ g, const float *__restrict__ W, const HsegFilter &flt, int pt,
                                   float wmin, float wmax, float wstep, int omin, int omax, int ostep, HsegBest &best,
                                   int lane) {
  const int plen = pattern_len(pt);
  // the widths of the pass and their offset counts, once (wave-uniform; the float loop increments of n_hseg.cpp:45).  The
  // four passes have at most six widths; a seventh or more sends the pass to the ordered form.
  constexpr int kMaxW = 7;
  float wv[kMaxW];
  int cn[kMaxW];
  int total = 0;
  float width = wmin;
#pragma unroll
  for (int i = 0; i < kMaxW; i++) {
    const bool in = width < wmax;
    wv[i] = width;
    int cnt = 0;
    if (in) {
      // offsets_for_width with the division spelled out for the two steps the search uses
      const float pw = (float)plen * width;
      unsigned short pom = (unsigned short)omax;
      const unsigned short maxo = (unsigned short)(428 - __float2int_rn(pw));
      if (pom == 0xFFFF || pom > maxo) pom = maxo;
      const int d = (int)pom - omin;
      cnt = d > 0 ? (ostep == 1 ? d : (ostep == 10 ? (d + 9) / 10 : (d + ostep - 1) / ostep)) : 0;
      width += wstep;
    }
    cn[i] = cnt;
    total += cnt;
  }
  const bool more_widths = width < wmax;
  if (total == 0 && !more_widths) return;
  bool decided = false;
  // (the table's precondition on the widths, and offsets that keep every table index inside the table: an offset is below
  // 428 - plen x width, so the last digit starts before column 428 - 16)
  if (total <= 128 && !more_widths && wmin >= 16.05f && wmax <= 23.5f && omin >= 0 && best.fit) {
    float sc[2], cw[2];
    int co[2];
    unsigned sa[2], sb[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int my = 64 * j + lane;
      float my_w = wv[0];  // (lanes without a candidate score the first one: any valid table index)
      int my_off = omin, idx = 0;
      bool has = false;
      if (64 * j < total) {
#pragma unroll
        for (int i = 0; i < kMaxW; i++) {
          const int rel = my - idx;
          if ((unsigned)rel < (unsigned)cn[i]) {  // (counts are disjoint ranges: at most one hit)
            has = true;
            my_w = wv[i];
            my_off = omin + rel * ostep;
          }
          idx += cn[i];
        }
      }
      cw[j] = my_w;
      co[j] = my_off;
      sa[j] = sb[j] = 0u;
      sc[j] = __builtin_inff();
      if (64 * j < total) sc[j] = hseg_table_score(W, flt.G, pt, my_w, my_off, has, sa[j], sb[j]);
    }
    // (non-negative floats and +inf order like their bit patterns)
    const unsigned mine = __float_as_uint(sc[0] < sc[1] ? sc[0] : sc[1]);
    unsigned mu = dmzwave::min_u32(mine);
    const unsigned inc = __float_as_uint(best.approx);
    if (inc < mu) mu = inc;
    const float m = __uint_as_float(mu);
    const float thr = (m + (flt.e1 + flt.e2 * m)) * 1.000001f;
    if (m < 3.0e38f) {
      const bool n0 = sc[0] <= thr, n1 = sc[1] <= thr, ninc = best.approx <= thr;
      const unsigned long long b0 = __builtin_amdgcn_ballot_w64(n0), b1 = __builtin_amdgcn_ballot_w64(n1);
      // the earliest of them: the incumbent, else the first candidate in iteration order
      int r_off = best.sig_off;
      unsigned r_a = best.sig_a, r_b = best.sig_b;
      float r_w = best.width, r_s = best.approx;
      if (!ninc) {
        const int j = b0 ? 0 : 1;
        const int wl = b0 ? (int)__builtin_ctzll(b0) : (int)__builtin_ctzll(b1);
        r_off = __builtin_amdgcn_readlane(j ? co[1] : co[0], wl);
        r_a = (unsigned)__builtin_amdgcn_readlane((int)(j ? sa[1] : sa[0]), wl);
        r_b = (unsigned)__builtin_amdgcn_readlane((int)(j ? sb[1] : sb[0]), wl);
        r_w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(j ? cw[1] : cw[0]), wl));
        r_s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(j ? sc[1] : sc[0]), wl));
      }
      const bool other = (n0 && (co[0] != r_off || sa[0] != r_a || sb[0] != r_b)) ||
                         (n1 && (co[1] != r_off || sa[1] != r_a || sb[1] != r_b));
      if (__builtin_amdgcn_ballot_w64(other) == 0ull) {
        decided = true;
        if (!ninc) {
          best.width = r_w;
          best.offset = r_off;
          best.approx = r_s;
          best.sig_off = r_off;
          best.sig_a = r_a;
          best.sig_b = r_b;
          best.exact = false;
          best.improved = true;
        }
      }
    } else {
      decided = true;  // no candidate in bounds and the incumbent is the initial one: nothing changes
    }
  }
#ifdef DMZ_HSEG_DBG
  if (lane == 0) atomicAdd(&g_hs_dbg[0], 1ull), atomicAdd(&g_hs_dbg[1], decided ? 0ull : 1ull);
#endif
  if (decided) return;
  // the ordered form of the pass
  hseg_make_exact(g, pt, best, lane);
  const float before_w = best.width;
  const int before_o = best.offset;
  const float before_s = best.score;
  best = hseg_pass(g, pt, wmin, wmax, wstep, omin, omax, ostep, best, lane);
  if (best.score != before_s || best.width != before_w || best.offset != before_o) {
    // (an in-bounds winner of a width the table takes: its digits start before column 409)
    best.fit = best.width >= 16.05f && best.width < 23.5f && best.offset >= 0;
    if (best.fit) {
      unsigned a, b;
      best.approx = hseg_table_score(W, flt.G, pt, best.width, best.offset, true, a, b);
      best.sig_off = best.offset;
      best.sig_a = a;
      best.sig_b = b;
    }
  }
}

#ifndef DMZ_HSEG_WAVES  /* waves per SIMD the register allocation aims at (3 .. 6: 0.470 .. 0.481 ms, 4 = no scratch; LDS allows 18 workgroups per CU) */
#define DMZ_HSEG_WAVES 4
#endif
__global__ __launch_bounds__(64, DMZ_HSEG_WAVES) void k_hseg(const uint8_t *__restrict__ cards, size_t card_stride,
                                              int n, dmz_hip_frame_result *__restrict__ results) {
  // g[0 .. 427] + zero padding.  hseg_score_t's tail loops read up to g[c_k + wave_max(len) + 2]: the segment length is
  // the WAVE's maximum, so a lane whose last digit starts at column <= 408 reads up to 408 + (428 - 256) + 2 = 582 when
  // another lane's last digit starts at 256 (the leftmost a 15-digit pattern admits).  Those terms are multiplied by a zero
  // mask INSIDE an fma -- exact only for finite operands -- so every word such a read can reach is allocated and zeroed here
  // (reads past the allocation would see a previous workgroup's leftovers: NaN bit patterns poison the sum).
  __shared__ float g[kHsegG];
  // the filter's table W[c][L - 16]; its first 428 words hold the integer column sums until g is built
  __shared__ __attribute__((aligned(16))) float W[428 * 4];
  int *colsum = (int *)W;

  const int f = blockIdx.x;
  if (f >= n) return;
  dmz_hip_frame_result *res = results + f;
  if (!(res->flags & DMZ_HIP_FLAG_VSEG_OK)) return;
  const int lane = threadIdx.x;
  const int y_off = res->vseg_y_offset;
  const int pt = res->pattern_type;
  const uint32_t *strip = (const uint32_t *)(cards + (size_t)f * card_stride + (size_t)y_off * DMZ_CARD_WIDTH);

#ifdef DMZ_HSEG_TIMING
  long long hs_t[12];
#endif
  HS_T(0)
  // ---- cross gradient clamped at the strip (ROI) edge + column sums (n_hseg.cpp:90-95) ----
  // A lane owns one dword (four columns) of every row; the neighbour columns come from the neighbour lanes (DPP wave shifts).
  // Two passes of 64 lanes with a halo lane at each end: pass 0 = dwords -1 .. 62 (results for 0 .. 61), pass 1 = dwords 61 .. 107
  // (results for 62 .. 106); the "dwords" -1 and 107 are dwords 0 and 106 with the edge column copied into the byte the neighbour
  // reads (the replicated border of cv/morph.cpp:190-220), so the row loop has no edge cases.  All 27 loads of a pass are issued
  // before the first row is used: one memory latency per pass (the row-by-row form waited ~1.7 k cycles for each of its 2 x 27
  // rows: 90 k of a wave's 140 k cycles, -DDMZ_HSEG_TIMING).
  // (Measured and rejected: the five-tap max / min on packed 16-bit pairs -- even columns in one register, odd ones in another,
  // v_pk_max_u16 / v_pk_min_u16: 805 instead of 1 230 instructions per pass and 12 % SLOWER, 0.56 vs 0.50 ms.)
  int lmin = 1 << 30, lmax = -1;
#pragma unroll 1
  for (int pass = 0; pass < 2; pass++) {
    const int di = pass == 0 ? imax(lane - 1, 0) : imin(61 + lane, 106);
    const int q = pass == 0 ? lane - 1 : 61 + lane;
    const bool on = pass == 0 ? (lane >= 1 && lane <= 62) : (lane >= 1 && lane <= 45);
    // byte 3 := byte 0 in the lane of "dword -1", byte 0 := byte 3 in the lane of "dword 107"
    const unsigned fix = (pass == 0 && lane == 0) ? 0x00020100u : ((pass == 1 && lane == 46) ? 0x03020103u : 0x03020100u);
    uint32_t col[27];
#pragma unroll
    for (int r = 0; r < 27; r++) col[r] = strip[r * 107 + di];
#pragma unroll
    for (int r = 0; r < 27; r++) col[r] = __builtin_amdgcn_perm(col[r], col[r], fix);
    int sum0 = 0, sum1 = 0, sum2 = 0, sum3 = 0;
    uint32_t mid = col[0];
    int c0 = mid & 255, c1 = (mid >> 8) & 255, c2 = (mid >> 16) & 255, c3 = mid >> 24;
    int n0 = c0, n1 = c1, n2 = c2, n3 = c3;  // row -1 = row 0
#pragma unroll
    for (int r = 0; r < 27; r++) {
      uint32_t dn = col[r < 26 ? r + 1 : 26];  // row 27 = row 26
      const int s0 = dn & 255, s1 = (dn >> 8) & 255, s2 = (dn >> 16) & 255, s3 = dn >> 24;
      // lane - 1's dword and lane + 1's dword (wave_shr:1 / wave_shl:1; the wave's end lanes are halo lanes)
      const uint32_t lw = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mid, 0x138, 0xf, 0xf, true);
      const uint32_t rw = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mid, 0x130, 0xf, 0xf, true);
      const int wl = (int)(lw >> 24), er = (int)(rw & 255);
      sum0 += max5(n0, wl, c0, c1, s0) - min5(n0, wl, c0, c1, s0);
      sum1 += max5(n1, c0, c1, c2, s1) - min5(n1, c0, c1, c2, s1);
      sum2 += max5(n2, c1, c2, c3, s2) - min5(n2, c1, c2, c3, s2);
      sum3 += max5(n3, c2, c3, er, s3) - min5(n3, c2, c3, er, s3);
      // row by row: left alone the compiler first moves the neighbour dwords of all 27 rows (199 registers)
      asm volatile("" : "+v"(sum0), "+v"(sum1), "+v"(sum2), "+v"(sum3), "+v"(dn));
      n0 = c0; n1 = c1; n2 = c2; n3 = c3;
      c0 = s0; c1 = s1; c2 = s2; c3 = s3;
      mid = dn;
    }
    if (on) {
      *(int4 *)(colsum + 4 * q) = make_int4(sum0, sum1, sum2, sum3);
      lmin = imin(lmin, imin(imin(sum0, sum1), imin(sum2, sum3)));
      lmax = imax(lmax, imax(imax(sum0, sum1), imax(sum2, sum3)));
    }
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
    for (int c = lane; c < kHsegG; c += 64) g[c] = c < 428 ? (float)colsum[c] * fs + fb : 0.0f;
  }
  __syncthreads();

  HS_T(1)
  // ---- the filter's table and constants ----
  HsegFilter flt;
  {
    float ga = 0.0f;
    for (int c = lane; c < 428; c += 64) ga = ga + fabsf(g[c]);
    // wave sum (any order: inside delta)
    ga += __int_as_float(DMZ_DPP_SHR0(__float_as_int(ga), 1));
    ga += __int_as_float(DMZ_DPP_SHR0(__float_as_int(ga), 2));
    ga += __int_as_float(DMZ_DPP_SHR0(__float_as_int(ga), 4));
    ga += __int_as_float(DMZ_DPP_SHR0(__float_as_int(ga), 8));
    flt.G = (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(ga), 15)) +
             __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ga), 31))) +
            (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(ga), 47)) +
             __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ga), 63)));
    const float u = 5.9604645e-8f;  // 2^-24
    const float delta = u * (4500.0f + 30.0f * flt.G) * 1.001f;
    const float gam2 = 2.1f * 429.0f * u;
    flt.e1 = 2.0f * delta + gam2 * delta;
    flt.e2 = gam2;
  }
  __syncthreads();  // (colsum is dead: W takes its place)
  for (int c = lane; c < 428; c += 64) {
    float acc = 0.0f, w16 = 0.0f, w17 = 0.0f, w18 = 0.0f;
#pragma unroll
    for (int j = 0; j < 19; j++) {
      const float gv = g[c + j];
      acc = acc + (fabsf(gv - HSEG_T(j)) - fabsf(gv));
      if (j == 15) w16 = acc;
      if (j == 16) w17 = acc;
      if (j == 17) w18 = acc;
    }
    *(float4 *)(W + 4 * c) = make_float4(w16, w17, w18, acc);
  }
  __syncthreads();

  HS_T(2)
  HsegBest best;
  best.score = 428.0f;
  best.width = 0.0f;
  best.offset = 0;
  best.approx = 428.0f;
  best.sig_off = -1;
  best.sig_a = best.sig_b = 0u;
  best.exact = true;
  best.improved = false;
  best.fit = true;
  for (int pass = 0; pass < 4; pass++) {  // n_hseg.cpp:106-139
    const int po = best.offset;
    const float hw = pass == 1 ? 0.5f : (pass == 2 ? 0.2f : 0.1f);
    const int r = pass == 1 ? 10 : 3;
    const float wmin = pass == 0 ? 17.1f : best.width - hw, wmax = pass == 0 ? 19.7f : best.width + hw;
    const float wstep = pass == 0 ? 0.5f : (pass == 1 ? 0.2f : (pass == 2 ? 0.1f : 0.05f));
    const int omin = pass == 0 ? 0 : (po < r ? 0 : po - r), omax = pass == 0 ? 0xFFFF : po + r;
    hseg_pass_filtered(g, W, flt, pt, wmin, wmax, wstep, omin, omax, pass == 0 ? 10 : 1, best, lane);
    HS_T(3 + pass)
  }
  if (!best.exact) {  // (W is dead: its LDS holds the terms)
    best.score = pt == 2 ? hseg_ordered_chain_t<2>(g, W, best.width, best.offset, lane)
                         : hseg_ordered_chain_t<1>(g, W, best.width, best.offset, lane);
    best.exact = true;
  }
  HS_T(7)
#ifdef DMZ_HSEG_TIMING
  if (lane == 0 && blockIdx.x == gridDim.x / 2)
    printf("hseg wave: gradient %lld  table %lld  passes %lld %lld %lld %lld  ordered score %lld\n", hs_t[1] - hs_t[0],
           hs_t[2] - hs_t[1], hs_t[3] - hs_t[2], hs_t[4] - hs_t[3], hs_t[5] - hs_t[4], hs_t[6] - hs_t[5], hs_t[7] - hs_t[6]);
#endif
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
    if (best.improved) {
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

#ifdef DMZ_HSEG_DBG
extern "C" void dmz_dbg_hseg(unsigned long long *out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hs_dbg), sizeof(unsigned long long) * 4);
  if (reset) {
    unsigned long long z[4] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_hs_dbg), z, sizeof(z));
  }
}
#endif

void dmz_launch_hseg(hipStream_t s, const uint8_t *cards, size_t card_stride, int n,
                     dmz_hip_frame_result *results) {
  DMZ_REPEAT(hseg)
  hipLaunchKernelGGL(k_hseg, dim3(n), dim3(64), 0, s, cards, card_stride, n, results);
}
