// scan.hip -- digit segmentation and digit categorisation on a batch of rectified
// 428 x 270 cards whose number row was found by k_vseg (vseg.hip).
//
// Replaces, for a whole batch, scan_card_image's number path (scan/frame.cpp:24-81):
//   k_hseg   = best_n_hseg (scan/n_hseg.cpp:88-151): 5-tap cross gradient of the
//              428x27 strip (cv/morph.cpp:190-220), column sums, min-max, 4-pass L1
//              template search with the reference's float loop increments.
//   k_digits = number_scores (scan/n_categorize.cpp:75-107): per digit cross
//              gradient, llcv_equalize_hist (cv/stats.cpp:116-159), /255, three
//              CNNs (modelc_*.cpp:1893-1937), vote (n_categorize.cpp:45-71), and the
//              usable gate frame.cpp:63-64.
//
// Integer / index work is bit-exact with the reference semantics; the hseg L1
// scores and the 160-float score sum keep the reference's sequential (scalar
// Eigen) summation order so that their arg-min / gate are exact.  The MLP/CNN
// arithmetic uses fused multiply-adds and the device tanhf/expf: contract is
// |delta| <= 1e-4 on probabilities (the reference's own KAT tolerance is 1e-5).
// Compiled with -ffp-contract=off: fmaf() is used explicitly where fusing is
// allowed.
#include <float.h>

#include "dmz_hip_internal.h"

namespace {

__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }

// n_vseg.cpp:26-30 tables
__constant__ unsigned char c_pattern_len[3] = {0, 19, 17};
__constant__ unsigned char c_number_len[3] = {0, 16, 15};
__constant__ unsigned char c_patterns[3][19] = {
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {1, 1, 1, 1, 0, 1, 1, 1, 1, 0, 1, 1, 1, 1, 0, 1, 1, 1, 1},
    {1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 0},
};
// n_hseg.cpp:15-20 (data)
__constant__ float c_grad_sum_pattern[19] = {
    0.26228655f, 0.30289554f, 0.34632607f, 0.38725636f, 0.42745813f, 0.45875135f, 0.46498017f,
    0.45258447f, 0.43045216f, 0.42430462f, 0.44796554f, 0.47726529f, 0.48471646f, 0.46457738f,
    0.42799847f, 0.38851183f, 0.33966308f, 0.28802608f, 0.25377602f,
};

// ===========================================================================
// hseg
// ===========================================================================
constexpr int HS_THREADS = 128;

struct HsegBest {
  float score;
  float width;
  int offset;
};

// One pass of best_n_hseg_constrained (n_hseg.cpp:39-84): candidates are numbered
// in the reference's iteration order (width outer, offset inner); thread t scores
// candidate base+t with the reference's sequential 428-term sum.
__device__ void hseg_pass(const float *__restrict__ g, int pattern_type, float wmin, float wmax,
                          float wstep, int omin, int omax, int ostep, HsegBest *best,
                          unsigned short *best_offsets /* LDS 16 */, unsigned short *cent /* LDS 16 x HS_THREADS */,
                          unsigned long long *red /* LDS HS_THREADS/64 */, int tid) {
  const int plen = c_pattern_len[pattern_type];
  // total number of candidates
  int total = 0;
  for (float width = wmin; width < wmax; width += wstep) {
    const float pw = (float)plen * width;
    unsigned short pom = (unsigned short)omax;
    const unsigned short maxo = (unsigned short)(428 - __float2int_rn(pw));
    if (pom == 0xFFFF || pom > maxo) pom = maxo;
    if ((int)pom > omin) total += ((int)pom - omin + ostep - 1) / ostep;
  }
  for (int base = 0; base < total; base += HS_THREADS) {
    const int my = base + tid;
    // locate candidate `my`
    float my_w = 0.0f;
    int my_off = 0;
    bool has = false;
    int idx = 0;
    for (float width = wmin; width < wmax; width += wstep) {
      const float pw = (float)plen * width;
      unsigned short pom = (unsigned short)omax;
      const unsigned short maxo = (unsigned short)(428 - __float2int_rn(pw));
      if (pom == 0xFFFF || pom > maxo) pom = maxo;
      const int cnt = ((int)pom > omin) ? ((int)pom - omin + ostep - 1) / ostep : 0;
      if (!has && my >= idx && my < idx + cnt) {
        has = true;
        my_w = width;
        my_off = omin + (my - idx) * ostep;
      }
      idx += cnt;
    }
    float score = FLT_MAX;
    int nd = 0;
    if (has) {
      bool in_bounds = true;
      for (int pi = 0; pi < plen; pi++) {
        if (c_patterns[pattern_type][pi]) {
          const unsigned short center = (unsigned short)(my_off + __float2int_rn((float)pi * my_w));
          if (!((int)center + 19 < 428)) in_bounds = false;
          cent[nd * HS_THREADS + tid] = center;
          nd++;
        }
      }
      if (in_bounds) {
        int k = -1;
        int next_c = cent[tid];
        int cur_c = -1000;
        float s = 0.0f;
        for (int i = 0; i < 428; i++) {
          while (k + 1 < nd && next_c <= i) {
            k++;
            cur_c = next_c;
            next_c = (k + 1 < nd) ? (int)cent[(k + 1) * HS_THREADS + tid] : 100000;
          }
          const int rel = i - cur_c;
          const float pv = (rel >= 0 && rel < 19) ? c_grad_sum_pattern[rel] : 0.0f;
          const float a = fabsf(g[i] - pv);
          s = (i == 0) ? a : s + a;
        }
        score = s;
      }
    }
    // block arg-min, ties -> earliest candidate (strict < in iteration order)
    unsigned long long key = ((unsigned long long)__float_as_uint(score) << 32) | (unsigned int)my;
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_xor(key, o, 64);
      key = other < key ? other : key;
    }
    if ((tid & 63) == 0) red[tid >> 6] = key;
    __syncthreads();
    unsigned long long kmin = red[0];
    for (int i = 1; i < HS_THREADS / 64; i++) kmin = red[i] < kmin ? red[i] : kmin;
    const float smin = __uint_as_float((unsigned int)(kmin >> 32));
    const int winner = (int)(kmin & 0xffffffffu);
    const bool better = smin < best->score;
    if (better && my == winner) {
      for (int d = 0; d < 16; d++) best_offsets[d] = d < nd ? cent[d * HS_THREADS + tid] : 0;
    }
    __syncthreads();
    if (better) {
      // every thread updates its private copy of `best` identically
      float ww = 0.0f;
      int wo = 0;
      int idx2 = 0;
      bool got = false;
      for (float width = wmin; width < wmax; width += wstep) {
        const float pw = (float)plen * width;
        unsigned short pom = (unsigned short)omax;
        const unsigned short maxo = (unsigned short)(428 - __float2int_rn(pw));
        if (pom == 0xFFFF || pom > maxo) pom = maxo;
        const int cnt = ((int)pom > omin) ? ((int)pom - omin + ostep - 1) / ostep : 0;
        if (!got && winner >= idx2 && winner < idx2 + cnt) {
          got = true;
          ww = width;
          wo = omin + (winner - idx2) * ostep;
        }
        idx2 += cnt;
      }
      best->score = smin;
      best->width = ww;
      best->offset = wo;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(HS_THREADS) void k_hseg(const uint8_t *__restrict__ cards,
                                                      size_t card_stride, int n,
                                                      dmz_hip_frame_result *__restrict__ results) {
  __shared__ __attribute__((aligned(16))) unsigned char strip[27 * 428];
  __shared__ float g[428];
  __shared__ int s_minmax[2];
  __shared__ unsigned short cent[16 * HS_THREADS];
  __shared__ unsigned short best_offsets[16];
  __shared__ unsigned long long red[HS_THREADS / 64];

  const int f = blockIdx.x;
  if (f >= n) return;
  dmz_hip_frame_result *res = results + f;
  if (!(res->flags & DMZ_HIP_FLAG_VSEG_OK)) return;
  const int tid = threadIdx.x;
  const int y_off = res->vseg_y_offset;
  const int pattern_type = res->pattern_type;
  const uint8_t *src = cards + (size_t)f * card_stride + (size_t)y_off * DMZ_CARD_WIDTH;
  // 27 x 428 bytes = 2889 aligned words (card rows start 4-byte aligned)
  for (int i = tid; i < 27 * 107; i += HS_THREADS) ((uint32_t *)strip)[i] = ((const uint32_t *)src)[i];
  if (tid == 0) { s_minmax[0] = 1 << 30; s_minmax[1] = -1; }
  if (tid < 16) best_offsets[tid] = 0;
  __syncthreads();
  // cross gradient clamped at the strip (ROI) edge + column sums (n_hseg.cpp:90-95)
  int lmin = 1 << 30, lmax = -1;
  int colsum[4];
  int nc = 0;
  for (int c = tid; c < 428; c += HS_THREADS, nc++) {
    const int cl = c > 0 ? c - 1 : c, cr = c < 427 ? c + 1 : c;
    int s = 0;
    for (int r = 0; r < 27; r++) {
      const int ru = r > 0 ? r - 1 : r, rd = r < 26 ? r + 1 : r;
      const int nn = strip[ru * 428 + c], ww = strip[r * 428 + cl], cc = strip[r * 428 + c],
                ee = strip[r * 428 + cr], ss = strip[rd * 428 + c];
      s += imax(nn, imax(ww, imax(cc, imax(ee, ss)))) - imin(nn, imin(ww, imin(cc, imin(ee, ss))));
    }
    colsum[nc] = s;
    lmin = imin(lmin, s);
    lmax = imax(lmax, s);
  }
  atomicMin(&s_minmax[0], lmin);
  atomicMax(&s_minmax[1], lmax);
  __syncthreads();
  {
    // cvNormalize(0,1,MINMAX) on the float sums
    const double smin = (double)(float)s_minmax[0], smax = (double)(float)s_minmax[1];
    const double scale = (smax - smin > DBL_EPSILON) ? 1. / (smax - smin) : 0.;
    const double shift = 0.0 - smin * scale;
    const float fs = (float)scale, fb = (float)shift;
    nc = 0;
    for (int c = tid; c < 428; c += HS_THREADS, nc++) g[c] = (float)colsum[nc] * fs + fb;
  }
  __syncthreads();

  HsegBest best;
  best.score = 428.0f;
  best.width = 0.0f;
  best.offset = 0;
  hseg_pass(g, pattern_type, 17.1f, 19.7f, 0.5f, 0, 0xFFFF, 10, &best, best_offsets, cent, red, tid);
  {
    const int po = best.offset;
    hseg_pass(g, pattern_type, best.width - 0.5f, best.width + 0.5f, 0.2f, po < 10 ? 0 : po - 10,
              po + 10, 1, &best, best_offsets, cent, red, tid);
  }
  {
    const int po = best.offset;
    hseg_pass(g, pattern_type, best.width - 0.2f, best.width + 0.2f, 0.1f, po < 3 ? 0 : po - 3,
              po + 3, 1, &best, best_offsets, cent, red, tid);
  }
  {
    const int po = best.offset;
    hseg_pass(g, pattern_type, best.width - 0.1f, best.width + 0.1f, 0.05f, po < 3 ? 0 : po - 3,
              po + 3, 1, &best, best_offsets, cent, red, tid);
  }
  if (tid == 0) {
    res->n_offsets = c_number_len[pattern_type];
    res->hseg_score = best.score;
    res->number_width = best.width;
    res->pattern_offset = best.offset;
  }
  if (tid < 16) res->offsets[tid] = best_offsets[tid];
}

// ===========================================================================
// digits
// ===========================================================================
constexpr int DG_THREADS = 256;
constexpr int DG_XSTRIDE = 516;  // floats per digit patch (513 used)

// conv 3x3 valid (24x15 outputs) -> 3x3 max pool (8x5) -> + bias -> tanh for one
// (digit, pooled position); the 8 kernels share the 5x5 input patch.
__device__ __forceinline__ void digit_conv_pool(const float *__restrict__ xp /* digit patch */,
                                                const float *__restrict__ mw /* model weights */,
                                                int pos, float *__restrict__ pooled /* 320 */) {
  const int pr = pos / 5, pc = pos - pr * 5;
  float in[5][5];
#pragma unroll
  for (int i = 0; i < 5; i++)
#pragma unroll
    for (int j = 0; j < 5; j++) in[i][j] = xp[(pr * 3 + i) * 19 + pc * 3 + j];
#pragma unroll 1
  for (int k = 0; k < 8; k++) {
    float w[9];
#pragma unroll
    for (int q = 0; q < 9; q++) w[q] = mw[dmzw::D_CONV_W + k * 9 + q];
    float m = -FLT_MAX;
#pragma unroll
    for (int oy = 0; oy < 3; oy++)
#pragma unroll
      for (int ox = 0; ox < 3; ox++) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) s = fmaf(w[i * 3 + j], in[oy + i][ox + j], s);
        m = s > m ? s : m;
      }
    pooled[k * 40 + pos] = tanhf(m + mw[dmzw::D_CONV_B + k]);
  }
}

__global__ __launch_bounds__(DG_THREADS) void k_digits(const float *__restrict__ wts,
                                                        const float *__restrict__ hidwt /* 3 x [320][32] */,
                                                        const uint8_t *__restrict__ cards,
                                                        size_t card_stride, int n,
                                                        dmz_hip_frame_result *__restrict__ results) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float *x = (float *)lds;                               // 16 x DG_XSTRIDE floats   (33,024 B)
  float *pooled = x + 16 * DG_XSTRIDE;                   // 16 x 320 floats           (20,480 B)
  int *hist = (int *)pooled;                             // 16 x 256 ints, dead before pooled is written
  unsigned char *img = (unsigned char *)(pooled + 16 * 320);  // 16 x 528 bytes     (8,448 B)
  float *hid = (float *)(img + 16 * 528);                // 16 x 32
  float *prob = hid + 16 * 32;                           // 3 x 16 x 10
  float *s_misc = prob + 480;                            // 4

  const int f = blockIdx.x;
  if (f >= n) return;
  dmz_hip_frame_result *res = results + f;
  if (!(res->flags & DMZ_HIP_FLAG_VSEG_OK)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nd = res->n_offsets;
  const int y_off = res->vseg_y_offset;
  const uint8_t *strip = cards + (size_t)f * card_stride + (size_t)y_off * DMZ_CARD_WIDTH;

  // ---- (a) per digit: cross gradient clamped at the 19x27 ROI edge, histogram ----
  for (int i = tid; i < 16 * 256; i += DG_THREADS) hist[i] = 0;
  __syncthreads();
  for (int i = tid; i < nd * 513; i += DG_THREADS) {
    const int d = i / 513, p = i - d * 513;
    const int r = p / 19, c = p - r * 19;
    const uint8_t *roi = strip + res->offsets[d];
    const int ru = r > 0 ? r - 1 : r, rd = r < 26 ? r + 1 : r;
    const int cl = c > 0 ? c - 1 : c, cr = c < 18 ? c + 1 : c;
    const int nn = roi[ru * DMZ_CARD_WIDTH + c], ww = roi[r * DMZ_CARD_WIDTH + cl],
              cc = roi[r * DMZ_CARD_WIDTH + c], ee = roi[r * DMZ_CARD_WIDTH + cr],
              ss = roi[rd * DMZ_CARD_WIDTH + c];
    const int gv = imax(nn, imax(ww, imax(cc, imax(ee, ss)))) - imin(nn, imin(ww, imin(cc, imin(ee, ss))));
    img[d * 528 + p] = (unsigned char)gv;
    atomicAdd(&hist[d * 256 + gv], 1);
  }
  __syncthreads();
  // ---- equalisation LUT (stats.cpp:135-151): one wave per digit, 4 bins per lane ----
  for (int d = wave; d < nd; d += DG_THREADS / 64) {
    int *h = hist + d * 256;
    const int h0 = h[lane * 4 + 0], h1 = h[lane * 4 + 1], h2 = h[lane * 4 + 2], h3 = h[lane * 4 + 3];
    const int tot = h0 + h1 + h2 + h3;
    int incl = tot;
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    const int excl = incl - tot;
    const float scale = 255.f / (19 * 27);
    int c0 = excl + h0, c1 = c0 + h1, c2 = c1 + h2, c3 = c2 + h3;
    int l0 = __float2int_rn((float)c0 * scale), l1 = __float2int_rn((float)c1 * scale),
        l2 = __float2int_rn((float)c2 * scale), l3 = __float2int_rn((float)c3 * scale);
    l0 = imin(255, imax(0, l0)); l1 = imin(255, imax(0, l1));
    l2 = imin(255, imax(0, l2)); l3 = imin(255, imax(0, l3));
    if (lane == 0) l0 = 0;  // lut[0] = 0 (stats.cpp:151)
    h[lane * 4 + 0] = l0; h[lane * 4 + 1] = l1; h[lane * 4 + 2] = l2; h[lane * 4 + 3] = l3;
  }
  __syncthreads();
  // ---- x = lut[img] * (1/255)  (n_categorize.cpp:98-99) ----
  {
    const float s255 = 1.0f / 255.0f;
    for (int i = tid; i < nd * 513; i += DG_THREADS) {
      const int d = i / 513, p = i - d * 513;
      x[d * DG_XSTRIDE + p] = (float)hist[d * 256 + img[d * 528 + p]] * s255;
    }
  }
  __syncthreads();

  // ---- (b)-(e) three CNNs ----
  for (int m = 0; m < 3; m++) {
    const float *mw = wts + dmzw::DIGIT0 + m * dmzw::DIGIT_STRIDE;
    const float *hwt = hidwt + (size_t)m * 320 * 32;
    for (int i = tid; i < nd * 40; i += DG_THREADS) {
      const int d = i / 40, pos = i - d * 40;
      digit_conv_pool(x + d * DG_XSTRIDE, mw, pos, pooled + d * 320);
    }
    __syncthreads();
    // hidden 320 -> 32: thread = (digit pair, unit j)
    {
      const int j = tid & 31, dg = tid >> 5;  // 8 digit groups x 2 digits
      const int d0 = dg * 2, d1 = dg * 2 + 1;
      float a0 = 0.0f, a1 = 0.0f;
      if (d0 < nd) {
        const float *p0 = pooled + d0 * 320, *p1 = pooled + (d1 < nd ? d1 : d0) * 320;
        for (int i = 0; i < 320; i++) {
          const float wv = hwt[i * 32 + j];
          a0 = fmaf(wv, p0[i], a0);
          a1 = fmaf(wv, p1[i], a1);
        }
        const float bj = mw[dmzw::D_HID_B + j];
        hid[d0 * 32 + j] = tanhf(a0 + bj);
        if (d1 < nd) hid[d1 * 32 + j] = tanhf(a1 + bj);
      }
    }
    __syncthreads();
    // logistic 32 -> 10 + exp
    if (tid < nd * 10) {
      const int d = tid / 10, c = tid - d * 10;
      float a = 0.0f;
      for (int j = 0; j < 32; j++) a = fmaf(mw[dmzw::D_LOG_W + c * 32 + j], hid[d * 32 + j], a);
      prob[(m * 16 + d) * 10 + c] = expf(a + mw[dmzw::D_LOG_B + c]);
    }
    __syncthreads();
    if (tid < nd) {
      float *pp = prob + (m * 16 + tid) * 10;
      // Eigen 10-element redux tree: ((0+1)+(2+(3+4))) + ((5+6)+(7+(8+9)))
      const float sum = ((pp[0] + pp[1]) + (pp[2] + (pp[3] + pp[4]))) +
                        ((pp[5] + pp[6]) + (pp[7] + (pp[8] + pp[9])));
      for (int c = 0; c < 10; c++) pp[c] = pp[c] / sum;
    }
    __syncthreads();
  }
  // ---- (f) vote (n_categorize.cpp:69-70), arg-max, usable gate (frame.cpp:63-64) ----
  float *fin = x;  // reuse: 160 floats
  if (tid < 160) {
    const int d = tid / 10, c = tid - d * 10;
    float v = 0.0f;
    if (d < nd) {
      const float r0 = prob[(0 * 16 + d) * 10 + c], r1 = prob[(1 * 16 + d) * 10 + c],
                  r2 = prob[(2 * 16 + d) * 10 + c];
      float mx = r0 > r1 ? r0 : r1;
      mx = mx > r2 ? mx : r2;
      v = (((r0 + r1) + r2) - mx) / 2.0f;
    }
    fin[tid] = v;
    (&res->scores[0][0])[tid] = v;
  }
  __syncthreads();
  if (tid < 16) {
    int best = 0;
    for (int c = 1; c < 10; c++)
      if (fin[tid * 10 + c] > fin[tid * 10 + best]) best = c;
    res->digits[tid] = (uint8_t)best;
  }
  if (tid == 0) {
    float sum = fin[0];
    for (int i = 1; i < 160; i++) sum = sum + fin[i];  // sequential, Redux.h:168-184
    const float number_score = (float)nd - sum;
    res->number_score = number_score;
    if (number_score < 3.0f) res->flags = res->flags | DMZ_HIP_FLAG_USABLE;
  }
  (void)s_misc;
  (void)lane;
}

// Stand-alone digit model entry point (KAT): one block of 64 per input patch.
__global__ __launch_bounds__(64) void k_digit_model(const float *__restrict__ wts,
                                                     const float *__restrict__ hidwt, int model,
                                                     const float *__restrict__ xin, int n,
                                                     float *__restrict__ out) {
  __shared__ float x[DG_XSTRIDE];
  __shared__ float pooled[320];
  __shared__ float hid[32];
  __shared__ float prob[10];
  const int i = blockIdx.x, tid = threadIdx.x;
  if (i >= n) return;
  const float *mw = wts + dmzw::DIGIT0 + model * dmzw::DIGIT_STRIDE;
  const float *hwt = hidwt + (size_t)model * 320 * 32;
  for (int k = tid; k < 513; k += 64) x[k] = xin[(size_t)i * 513 + k];
  __syncthreads();
  if (tid < 40) digit_conv_pool(x, mw, tid, pooled);
  __syncthreads();
  if (tid < 32) {
    float a = 0.0f;
    for (int k = 0; k < 320; k++) a = fmaf(hwt[k * 32 + tid], pooled[k], a);
    hid[tid] = tanhf(a + mw[dmzw::D_HID_B + tid]);
  }
  __syncthreads();
  if (tid < 10) {
    float a = 0.0f;
    for (int j = 0; j < 32; j++) a = fmaf(mw[dmzw::D_LOG_W + tid * 32 + j], hid[j], a);
    prob[tid] = expf(a + mw[dmzw::D_LOG_B + tid]);
  }
  __syncthreads();
  if (tid < 10) {
    const float sum = ((prob[0] + prob[1]) + (prob[2] + (prob[3] + prob[4]))) +
                      ((prob[5] + prob[6]) + (prob[7] + (prob[8] + prob[9])));
    out[i * 10 + tid] = prob[tid] / sum;
  }
}

constexpr int kDigitsLds = (16 * DG_XSTRIDE + 16 * 320) * 4 + 16 * 528 + (16 * 32 + 480 + 4) * 4;

}  // namespace

void dmz_launch_hseg(hipStream_t s, const uint8_t *cards, size_t card_stride, int n,
                     dmz_hip_frame_result *results) {
  hipLaunchKernelGGL(k_hseg, dim3(n), dim3(HS_THREADS), 0, s, cards, card_stride, n, results);
}

void dmz_launch_digits(hipStream_t s, const float *weights, const float *hidwt, const uint8_t *cards,
                       size_t card_stride, int n, dmz_hip_frame_result *results) {
  hipLaunchKernelGGL(k_digits, dim3(n), dim3(DG_THREADS), kDigitsLds, s, weights, hidwt, cards,
                     card_stride, n, results);
}

void dmz_launch_digit_model(hipStream_t s, const float *weights, const float *hidwt, int model,
                            const float *x, int n, float *out) {
  hipLaunchKernelGGL(k_digit_model, dim3(n), dim3(64), 0, s, weights, hidwt, model, x, n, out);
}

int dmz_configure_scan(void) {
  int e = dmz_configure_vseg();
  if (e) return e;
  return (int)hipFuncSetAttribute((const void *)k_digits,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kDigitsLds);
}
