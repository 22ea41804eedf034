/*
 * orc_scan.c -- ORACLE (test infrastructure only, see dmz_oracle.h): CPU
 * restatement of the card-number half of the hot path: number-row search
 * (scan/n_vseg.cpp), digit segmentation (scan/n_hseg.cpp), digit categorisation
 * (scan/n_categorize.cpp + models/generated/*), the frame gates
 * (scan/frame.cpp:20-81) and the little image kernels they use
 * (cv/morph.cpp, cv/convert.cpp, cv/stats.cpp).
 *
 * x86 flavour of the reference: dmz_has_neon_runtime() is false, so every
 * llcv_* call takes its OpenCV twin.
 * Reductions follow Eigen 3.2.4's scalar (EIGEN_DONT_VECTORIZE) order.
 */
#include "dmz_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static const float *g_w = NULL;
void orc_set_weights(const float *blob) { g_w = blob; }

static int u8max(int a, int b) { return a > b ? a : b; }
static int u8min(int a, int b) { return a < b ? a : b; }

/* morph.cpp:108-112 on a 1-row ROI == max3 - min3 with replicated ends (A5) */
void orc_morph_grad3_1d(const uint8_t *src, int n, uint8_t *dst) {
  for (int i = 0; i < n; i++) {
    int l = src[i > 0 ? i - 1 : 0], c = src[i], r = src[i < n - 1 ? i + 1 : n - 1];
    dst[i] = (uint8_t)(u8max(l, u8max(c, r)) - u8min(l, u8min(c, r)));
  }
}

/* morph.cpp:190-220: 5-tap cross max-min, clamped at the ROI edge */
void orc_morph_grad3_2d_cross(const uint8_t *src, int stride, int w, int h, uint8_t *dst,
                              int dstride) {
  for (int r = 0; r < h; r++) {
    const uint8_t *r1 = src + (size_t)(r == 0 ? r : r - 1) * stride;
    const uint8_t *r2 = src + (size_t)r * stride;
    const uint8_t *r3 = src + (size_t)(r == h - 1 ? r : r + 1) * stride;
    for (int c = 0; c < w; c++) {
      int c1 = c == 0 ? c : c - 1, c3 = c == w - 1 ? c : c + 1;
      int n = r1[c], wv = r2[c1], cv = r2[c], e = r2[c3], s = r3[c];
      int mx = u8max(n, u8max(wv, u8max(cv, u8max(e, s))));
      int mn = u8min(n, u8min(wv, u8min(cv, u8min(e, s))));
      dst[(size_t)r * dstride + c] = (uint8_t)(mx - mn);
    }
  }
}

/* convert.cpp:195-197: cvResize INTER_LINEAR x0.5 in 1-D == (a+b+1)>>1 (A6) */
void orc_lineardown2_1d(const uint8_t *src, int n_out, uint8_t *dst) {
  for (int i = 0; i < n_out; i++) dst[i] = (uint8_t)((src[2 * i] + src[2 * i + 1] + 1) >> 1);
}

/* cvNormalize(src32F, dst32F, 0, 1, CV_MINMAX) in place (A8) */
static void orc_normalize_minmax_f32(float *v, int n) {
  double smin = v[0], smax = v[0];
  for (int i = 1; i < n; i++) {
    if (v[i] < smin) smin = v[i];
    if (v[i] > smax) smax = v[i];
  }
  double scale = (1.0 - 0.0) * (smax - smin > DBL_EPSILON ? 1. / (smax - smin) : 0);
  double shift = 0.0 - smin * scale;
  const float fs = (float)scale, fb = (float)shift;
  for (int i = 0; i < n; i++) v[i] = v[i] * fs + fb;
}

/* convert.cpp:380-383: cvConvertScale(1/255) then cvNormalize MINMAX (A7, A8) */
void orc_norm_convert_1d(const uint8_t *src, int n, float *dst) {
  const float s = 1.0f / 255.0f;
  for (int i = 0; i < n; i++) dst[i] = (float)src[i] * s;
  orc_normalize_minmax_f32(dst, n);
}

/* stats.cpp:116-159 on a non-continuous w x h image */
void orc_equalize_hist(uint8_t *img, int stride, int w, int h) {
  int hist[256];
  memset(hist, 0, sizeof(hist));
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) hist[img[(size_t)y * stride + x]]++;
  float scale = 255.f / (w * h);
  int sum = 0;
  uint8_t lut[256];
  for (int i = 0; i < 256; i++) {
    sum += hist[i];
    int val = (int)lrint((double)(sum * scale));
    lut[i] = (uint8_t)(val < 0 ? 0 : (val > 255 ? 255 : val));
  }
  lut[0] = 0;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) img[(size_t)y * stride + x] = lut[img[(size_t)y * stride + x]];
}

/* ---- models ---------------------------------------------------------------- */
static float dotf(const float *a, const float *b, int n) {
  float s = 0.0f;
  for (int i = 0; i < n; i++) s += a[i] * b[i];
  return s;
}

/* Eigen fixed-size .sum() in scalar mode: balanced binary split (Redux.h:77-90) */
static float tree_sum(const float *v, int start, int len) {
  if (len == 1) return v[start];
  int half = len / 2;
  return tree_sum(v, start, half) + tree_sum(v, start + half, len - half);
}

static void mlp_forward(const float *w1, const float *b1, const float *w2, const float *b2,
                        int nin, int nhid, int nout, const float *x, float *out) {
  float hid[128];
  for (int j = 0; j < nhid; j++) hid[j] = tanhf(dotf(w1 + (size_t)j * nin, x, nin) + b1[j]);
  for (int k = 0; k < nout; k++) out[k] = expf(dotf(w2 + (size_t)k * nhid, hid, nhid) + b2[k]);
  float s = tree_sum(out, 0, nout);
  for (int k = 0; k < nout; k++) out[k] /= s;
}

/* modelm_befe75da.cpp:1770-1786 */
void orc_applym_vseg(const float x[204], float out[3]) {
  mlp_forward(g_w + ORC_W_VSEG_W1, g_w + ORC_W_VSEG_B1, g_w + ORC_W_VSEG_W2, g_w + ORC_W_VSEG_B2,
              204, 50, 3, x, out);
}

/* modelm_730c4cbd.cpp:2431-2449 */
void orc_applym_slash(const float x[176], float out[2]) {
  const float *w = g_w + ORC_W_SLASH;
  mlp_forward(w, w + 80 * 176, w + 80 * 176 + 80, w + 80 * 176 + 80 + 160, 176, 80, 2, x, out);
}

/* modelc_{5c241121,01266c1b,b00bf70c}.cpp:1844-1937 (scalar conv branch 1873-1878) */
void orc_applyc_digit(int model, const float x[27 * 19], float out[10]) {
  const float *w = g_w + ORC_W_DIGIT0 + (size_t)model * ORC_DIGIT_STRIDE;
  const float *conv_w = w + ORC_DIGIT_CONV_W, *conv_b = w + ORC_DIGIT_CONV_B;
  float acc[320], hid[32];
  for (int k = 0; k < 8; k++) {
    const float *kw = conv_w + k * 9;
    float conv[24 * 15];
    for (int r = 0; r < 24; r++)
      for (int c = 0; c < 15; c++) {
        float e[9];
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++) e[i * 3 + j] = kw[i * 3 + j] * x[(r + i) * 19 + (c + j)];
        conv[r * 15 + c] = tree_sum(e, 0, 9);
      }
    for (int pr = 0; pr < 8; pr++)
      for (int pc = 0; pc < 5; pc++) {
        float m = conv[(pr * 3) * 15 + pc * 3];
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++) {
            float v = conv[(pr * 3 + i) * 15 + pc * 3 + j];
            if (v > m) m = v;
          }
        acc[k * 40 + pr * 5 + pc] = m + conv_b[k];
      }
  }
  for (int i = 0; i < 320; i++) acc[i] = tanhf(acc[i]);
  for (int j = 0; j < 32; j++)
    hid[j] = tanhf(dotf(w + ORC_DIGIT_HID_W + (size_t)j * 320, acc, 320) + w[ORC_DIGIT_HID_B + j]);
  for (int k = 0; k < 10; k++)
    out[k] = expf(dotf(w + ORC_DIGIT_LOG_W + k * 32, hid, 32) + w[ORC_DIGIT_LOG_B + k]);
  float s = tree_sum(out, 0, 10);
  for (int k = 0; k < 10; k++) out[k] /= s;
}

/* modelc_bf4dd6c8.cpp:12500-12635 (conv1 "full", 16x11 -> computes 20x14, pool 2x2 ->
 * 10x7, +b, ReLU), 12688-12724 (conv2 valid over 50 maps -> 6x3, pool 2x3 -> 3x1, +b,
 * ReLU), 13457-13505 (FC 120->176 ReLU, FC 176->10 softmax). */
void orc_applyc_expiry(const float xin[16 * 11], float out[10], float *l1_out, float *l2_out,
                       float *l3_out) {
  const float *w = g_w + ORC_W_EXPIRY;
  /* modelc_bf4dd6c8.cpp:13459: input minus its mean (sequential 176-float sum) */
  float x[16 * 11], msum = xin[0];
  for (int i = 1; i < 176; i++) msum = msum + xin[i];
  const float mean = msum / 176.0f;
  for (int i = 0; i < 176; i++) x[i] = xin[i] - mean;
  const float *c1w = w, *c1b = c1w + 1250, *c2w = c1b + 50, *c2b = c2w + 50000;
  const float *hw = c2b + 40, *hb = hw + 21120, *lw = hb + 176, *lb = lw + 1760;
  static __thread float l1[50 * 70];
  float l2[120], l3[176];
  /* layer 1: correlation with zero padding 4 on every side, outputs 20 x 14 of the
   * 20 x 15 "full" result are pooled 2x2 -> 10 x 7 */
  for (int k = 0; k < 50; k++) {
    const float *kw = c1w + k * 25;
    float conv[20 * 14];
    for (int r = 0; r < 20; r++)
      for (int c = 0; c < 14; c++) {
        float s = 0.0f;
        for (int i = 0; i < 5; i++)
          for (int j = 0; j < 5; j++) {
            int yy = r + i - 4, xx = c + j - 4;
            if (yy >= 0 && yy < 16 && xx >= 0 && xx < 11) s += kw[i * 5 + j] * x[yy * 11 + xx];
          }
        conv[r * 14 + c] = s;
      }
    for (int pr = 0; pr < 10; pr++)
      for (int pc = 0; pc < 7; pc++) {
        float m = conv[(pr * 2) * 14 + pc * 2];
        for (int i = 0; i < 2; i++)
          for (int j = 0; j < 2; j++) {
            float v = conv[(pr * 2 + i) * 14 + pc * 2 + j];
            if (v > m) m = v;
          }
        float v = m + c1b[k];
        l1[k * 70 + pr * 7 + pc] = v > 0.0f ? v : 0.0f;
      }
  }
  /* layer 2: valid correlation summed over the 50 maps: 10x7 -> 6x3, pool 2x3 -> 3x1 */
  for (int k = 0; k < 40; k++) {
    float conv[6 * 3];
    for (int r = 0; r < 6; r++)
      for (int c = 0; c < 3; c++) {
        float s = 0.0f;
        for (int m = 0; m < 50; m++) {
          const float *kw = c2w + ((size_t)k * 50 + m) * 25;
          for (int i = 0; i < 5; i++)
            for (int j = 0; j < 5; j++) s += kw[i * 5 + j] * l1[m * 70 + (r + i) * 7 + (c + j)];
        }
        conv[r * 3 + c] = s;
      }
    for (int pr = 0; pr < 3; pr++) {
      float m = conv[(pr * 2) * 3];
      for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++) {
          float v = conv[(pr * 2 + i) * 3 + j];
          if (v > m) m = v;
        }
      float v = m + c2b[k];
      l2[k * 3 + pr] = v > 0.0f ? v : 0.0f;
    }
  }
  for (int j = 0; j < 176; j++) {
    float v = dotf(hw + (size_t)j * 120, l2, 120) + hb[j];
    l3[j] = v > 0.0f ? v : 0.0f;
  }
  for (int k = 0; k < 10; k++) out[k] = expf(dotf(lw + (size_t)k * 176, l3, 176) + lb[k]);
  float s = tree_sum(out, 0, 10);
  for (int k = 0; k < 10; k++) out[k] /= s;
  if (l1_out) memcpy(l1_out, l1, sizeof(l1));
  if (l2_out) memcpy(l2_out, l2, sizeof(l2));
  if (l3_out) memcpy(l3_out, l3, sizeof(l3));
}

/* ---- n_vseg.cpp ------------------------------------------------------------ */
/* n_vseg.cpp:39-43: ROI (10, y, 408, 1) -> grad -> 1/2 -> norm */
void orc_vseg_row_features(const uint8_t *row408, float feat[204]) {
  uint8_t grad[408], down[204];
  orc_morph_grad3_1d(row408, 408, grad);
  orc_lineardown2_1d(grad, 204, down);
  orc_norm_convert_1d(down, 204, feat);
}

/* n_vseg.cpp:49-92, literally (running float sum with add/subtract drift) */
void orc_best_segmentation_for_vseg_scores(const float *visa, const float *amex, float *score,
                                           int *y_off, int *pattern) {
  float vsum = 0.0f, asum = 0.0f;
  float vring[27], aring[27];
  *score = 0.0f;
  *pattern = 0;
  *y_off = 0;
  for (int y = 0; y < 270; y++) {
    float v = visa[y], a = amex[y];
    vsum += v;
    asum += a;
    int bi = y % 27;
    vring[bi] = v;
    aring[bi] = a;
    if (y >= 26) {
      if (vsum > *score) { *score = vsum; *pattern = 1; *y_off = y - 27 + 1; }
      if (asum > *score) { *score = asum; *pattern = 2; *y_off = y - 27 + 1; }
      int nbi = (y + 1) % 27;
      vsum -= vring[nbi];
      asum -= aring[nbi];
    }
  }
}

/* n_vseg.cpp:94-168 */
void orc_best_n_vseg(const uint8_t *card, int stride, float *score, int *y_offset, int *pattern,
                     float *visa_out, float *amex_out) {
  float visa[270], amex[270], feat[204], p[3];
  memset(visa, 0, sizeof(visa));
  memset(amex, 0, sizeof(amex));
  for (int y = 0; y < 270; y += 4) {
    orc_vseg_row_features(card + (size_t)y * stride + 10, feat);
    orc_applym_vseg(feat, p);
    visa[y] = p[1];
    amex[y] = p[2];
  }
  orc_best_segmentation_for_vseg_scores(visa, amex, score, y_offset, pattern);
  int best_y = *y_offset;
  int ymin = best_y < 8 ? 0 : best_y - 8;
  if (ymin > 270) ymin = 270;
  int ymax = best_y + 27 + 8;
  if (ymax > 270) ymax = 270;
  for (int y = ymin; y < ymax; y++) {
    if (visa[y] == 0 && amex[y] == 0) {
      orc_vseg_row_features(card + (size_t)y * stride + 10, feat);
      orc_applym_vseg(feat, p);
      visa[y] = p[1];
      amex[y] = p[2];
    }
  }
  orc_best_segmentation_for_vseg_scores(visa, amex, score, y_offset, pattern);
  if (visa_out) memcpy(visa_out, visa, sizeof(visa));
  if (amex_out) memcpy(amex_out, amex, sizeof(amex));
}

/* ---- n_hseg.cpp ------------------------------------------------------------ */
static const float k_number_grad_sum_pattern[19] = { /* n_hseg.cpp:15-20 (data) */
    0.26228655f, 0.30289554f, 0.34632607f, 0.38725636f, 0.42745813f, 0.45875135f, 0.46498017f,
    0.45258447f, 0.43045216f, 0.42430462f, 0.44796554f, 0.47726529f, 0.48471646f, 0.46457738f,
    0.42799847f, 0.38851183f, 0.33966308f, 0.28802608f, 0.25377602f,
};
static const uint8_t k_pattern_len[3] = {0, 19, 17};   /* n_vseg.cpp:27 */
static const uint8_t k_number_len[3] = {0, 16, 15};    /* n_vseg.cpp:26 */
static const uint8_t k_patterns[3][19] = {             /* n_vseg.cpp:28-30 */
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {1, 1, 1, 1, 0, 1, 1, 1, 1, 0, 1, 1, 1, 1, 0, 1, 1, 1, 1},
    {1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 0},
};

/* n_hseg.cpp:90-96: cross gradient of the 428x27 strip, column sums, min-max */
void orc_hseg_grad_sums(const uint8_t *strip, int stride, float sums[428]) {
  uint8_t grad[428 * 27];
  orc_morph_grad3_2d_cross(strip, stride, 428, 27, grad, 428);
  for (int c = 0; c < 428; c++) {
    int s = 0;
    for (int r = 0; r < 27; r++) s += grad[r * 428 + c];
    sums[c] = (float)s;
  }
  orc_normalize_minmax_f32(sums, 428);
}

/* n_hseg.cpp:39-84.  `score`, `number_width`, `pattern_offset`, `offsets` carry the
 * incoming best and are updated in place. omax == 0xFFFF means "no limit". */
void orc_best_n_hseg_constrained(const float *grad_sums, int pattern_type, float wmin, float wmax,
                                 float wstep, int omin, int omax, int ostep, uint16_t offsets[16],
                                 float *score, float *number_width, int *pattern_offset) {
  const int plen = k_pattern_len[pattern_type];
  const uint8_t *npat = k_patterns[pattern_type];
  float pattern[428];
  /* the reference leaves temp_offsets[15] uninitialised for 15-digit patterns;
   * zero here so that the (unused) 16th offset is deterministic */
  uint16_t temp_offsets[16] = {0};
  for (float width = wmin; width < wmax; width += wstep) {
    float pattern_width = plen * width;
    uint16_t pattern_offset_max = (uint16_t)omax;
    uint16_t maximum_pattern_offset_max = (uint16_t)(428 - lrintf(pattern_width));
    if (pattern_offset_max == 0xFFFF || pattern_offset_max > maximum_pattern_offset_max)
      pattern_offset_max = maximum_pattern_offset_max;
    for (uint16_t offset = (uint16_t)omin; offset < pattern_offset_max;
         offset = (uint16_t)(offset + ostep)) {
      memset(pattern, 0, sizeof(pattern));
      int offset_index = 0;
      int in_bounds = 1;
      for (int pi = 0; pi < plen; pi++) {
        if (npat[pi]) {
          uint16_t center = (uint16_t)(offset + lrintf(pi * width));
          if (center + 19 < 428)
            memcpy(pattern + center, k_number_grad_sum_pattern, sizeof(k_number_grad_sum_pattern));
          else
            in_bounds = 0;
          temp_offsets[offset_index++] = center;
        }
      }
      if (in_bounds) {
        float s = fabsf(grad_sums[0] - pattern[0]);
        for (int i = 1; i < 428; i++) s = s + fabsf(grad_sums[i] - pattern[i]);
        if (s < *score) {
          memcpy(offsets, temp_offsets, sizeof(temp_offsets));
          *score = s;
          *number_width = width;
          *pattern_offset = offset;
        }
      }
    }
  }
}

/* n_hseg.cpp:88-151 */
void orc_best_n_hseg(const uint8_t *strip, int stride, int pattern_type, orc_frame_result *res) {
  float sums[428];
  orc_hseg_grad_sums(strip, stride, sums);
  uint16_t offsets[16];
  memset(offsets, 0, sizeof(offsets));
  float score = 428.0f, nw = 0.0f;
  int po = 0; /* the reference leaves pattern_offset uninitialised; 0 here */
  orc_best_n_hseg_constrained(sums, pattern_type, 17.1f, 19.7f, 0.5f, 0, 0xFFFF, 10, offsets,
                              &score, &nw, &po);
  orc_best_n_hseg_constrained(sums, pattern_type, nw - 0.5f, nw + 0.5f, 0.2f,
                              po < 10 ? 0 : po - 10, po + 10, 1, offsets, &score, &nw, &po);
  orc_best_n_hseg_constrained(sums, pattern_type, nw - 0.2f, nw + 0.2f, 0.1f,
                              po < 3 ? 0 : po - 3, po + 3, 1, offsets, &score, &nw, &po);
  orc_best_n_hseg_constrained(sums, pattern_type, nw - 0.1f, nw + 0.1f, 0.05f,
                              po < 3 ? 0 : po - 3, po + 3, 1, offsets, &score, &nw, &po);
  res->n_offsets = k_number_len[pattern_type];
  memcpy(res->offsets, offsets, sizeof(offsets));
  res->hseg_score = score;
  res->number_width = nw;
  res->pattern_offset = po;
}

/* ---- n_categorize.cpp ------------------------------------------------------ */
/* n_categorize.cpp:75-107 + 45-71.  strip = 27 rows starting at y_offset. */
void orc_number_scores(const uint8_t *strip, int stride, const uint16_t *offsets, int n,
                       float scores[160]) {
  memset(scores, 0, sizeof(float) * 160);
  for (int d = 0; d < n; d++) {
    uint8_t img[27 * 20];
    float x[27 * 19], r0[10], r1[10], r2[10];
    orc_morph_grad3_2d_cross(strip + offsets[d], stride, 19, 27, img, 20);
    orc_equalize_hist(img, 20, 19, 27);
    const float s = 1.0f / 255.0f;
    for (int r = 0; r < 27; r++)
      for (int c = 0; c < 19; c++) x[r * 19 + c] = (float)img[r * 20 + c] * s;
    orc_applyc_digit(0, x, r0);
    orc_applyc_digit(1, x, r1);
    orc_applyc_digit(2, x, r2);
    for (int k = 0; k < 10; k++) {
      float mx = r0[k] > r1[k] ? r0[k] : r1[k]; /* cwiseMax chain: max(max(r0,r1),r2) */
      mx = mx > r2[k] ? mx : r2[k];
      scores[d * 10 + k] = (r0[k] + r1[k] + r2[k] - mx) / 2.0f;
    }
  }
}

/* frame.cpp:49-81 once the vseg gates have passed: res->vseg_y_offset / pattern_type are set */
static void scan_number_from_vseg(const uint8_t *card, int stride, int collect_card_number, orc_frame_result *res) {
  const int y_off = res->vseg_y_offset, pattern = res->pattern_type;
  res->flags |= ORC_FLAG_VSEG_OK;
  if (!collect_card_number) {
    res->flags |= ORC_FLAG_USABLE;
    return;
  }
  const uint8_t *strip = card + (size_t)y_off * stride;
  orc_best_n_hseg(strip, stride, pattern, res);
  orc_number_scores(strip, stride, res->offsets, res->n_offsets, &res->scores[0][0]);
  /* scores.sum(): 160 floats, sequential row-major in scalar mode (Redux.h:168-184) */
  const float *sc = &res->scores[0][0];
  float sum = sc[0];
  for (int i = 1; i < 160; i++) sum = sum + sc[i];
  res->number_score = res->n_offsets - sum;
  if (res->number_score < 3) res->flags |= ORC_FLAG_USABLE; /* kMaxNumberScoreDelta */
  for (int d = 0; d < 16; d++) {
    int best = 0;
    for (int k = 1; k < 10; k++)
      if (res->scores[d][k] > res->scores[d][best]) best = k;
    res->digits[d] = (uint8_t)best;
  }
}

/* frame.cpp:24-81, number path (expiry handled elsewhere).  collect_card_number = 0 is what
 * scanner_add_frame_with_expiry passes once the session's number is accepted (scan.cpp:43-48): a frame
 * that passes the vseg gates is usable as it is (frame.cpp:43-49), hseg and the digit CNNs do not run. */
void orc_scan_card_image_ex(const uint8_t *card, int stride, int collect_card_number, orc_frame_result *res) {
  float score;
  int y_off, pattern;
  res->flags &= ORC_FLAG_WARPED;
  res->n_offsets = 0;
  memset(res->offsets, 0, sizeof(res->offsets));
  res->hseg_score = 0;
  res->number_width = 0;
  res->pattern_offset = 0;
  res->number_score = 0;
  memset(res->digits, 0, sizeof(res->digits));
  memset(res->scores, 0, sizeof(res->scores));
  orc_best_n_vseg(card, stride, &score, &y_off, &pattern, NULL, NULL);
  res->vseg_score = score;
  res->vseg_y_offset = y_off;
  res->pattern_type = pattern;
  if (y_off < (ORC_CARD_H - ORC_NUM_H) / 2) { /* kFlipVSegYOffsetCutoff */
    res->flags |= ORC_FLAG_UPSIDE_DOWN;
    return;
  }
  if (!(score > 15)) return; /* kMinVSegScore */
  scan_number_from_vseg(card, stride, collect_card_number, res);
}

/* Test infrastructure only (no counterpart in the reference): frame.cpp:38-81 downstream of a GIVEN vertical
 * segmentation.  Where the device's y_offset differs from the oracle's by a proven float near-tie of two window sums, the
 * parity tests re-run the oracle's later stages at the device's choice, so that a tie cannot hide a second difference. */
void orc_scan_card_image_at(const uint8_t *card, int stride, int y_off, int pattern, float score,
                            int collect_card_number, orc_frame_result *res) {
  res->flags &= ORC_FLAG_WARPED;
  res->n_offsets = 0;
  memset(res->offsets, 0, sizeof(res->offsets));
  res->hseg_score = 0;
  res->number_width = 0;
  res->pattern_offset = 0;
  res->number_score = 0;
  memset(res->digits, 0, sizeof(res->digits));
  memset(res->scores, 0, sizeof(res->scores));
  res->vseg_score = score;
  res->vseg_y_offset = y_off;
  res->pattern_type = pattern;
  if (y_off < (ORC_CARD_H - ORC_NUM_H) / 2) {
    res->flags |= ORC_FLAG_UPSIDE_DOWN;
    return;
  }
  if (!(score > 15)) return;
  scan_number_from_vseg(card, stride, collect_card_number, res);
}

void orc_scan_card_image(const uint8_t *card, int stride, orc_frame_result *res) {
  orc_scan_card_image_ex(card, stride, 1, res);
}

/* detect -> transform (Y plane) -> scan : the sequence of cython_dmz/dmz.pyx:379-483 */
void orc_scan_frame(const uint8_t *y, int stride, int w, int h, int orientation,
                    int truncate_corners, uint8_t *card_out, orc_frame_result *res) {
  memset(res, 0, sizeof(*res));
  uint8_t *card = card_out ? card_out : (uint8_t *)malloc(ORC_CARD_W * ORC_CARD_H);
  if (orc_detect_edges(y, stride, w, h, NULL, NULL, 0, orientation, res)) {
    orc_transform_card(y, stride, w, h, res->corners, orientation, truncate_corners, card);
    res->flags |= ORC_FLAG_WARPED;
    orc_scan_card_image(card, ORC_CARD_W, res);
  } else if (card_out) {
    memset(card_out, 0, ORC_CARD_W * ORC_CARD_H);
  }
  if (!card_out) free(card);
}
