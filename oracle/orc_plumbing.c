/*
 * orc_plumbing.c -- CPU ORACLE (test infrastructure, see dmz_oracle.h) for the camera-side plumbing
 * around the scan path: plane de-interleaving and the YCbCr -> RGB conversion
 * of the rectified card.  Integer arithmetic only; bit-exact on every target.
 *
 * Pinning: orc_deinterleave_rgba_to_r against the reference's compiled scalar branch
 * (tests/test_oracle_vs_ref.py).  llcv_split_u8 (= cvSplit) and llcv_YCbCr2RGB_u8_c need
 * cvGetSize / cvSplit to run: restated from the in-tree source, "parity unpinned".
 */
#include "dmz_oracle.h"

/* cv/convert.cpp:105-107 (x86: cvSplit): channel1 = first byte of every pair, channel2 = second */
void orc_split_u8(const uint8_t *interleaved, int stride, int w, int h, uint8_t *c1, uint8_t *c2) {
  for (int r = 0; r < h; r++)
    for (int c = 0; c < w; c++) {
      c1[(size_t)r * w + c] = interleaved[(size_t)r * stride + 2 * c];
      c2[(size_t)r * w + c] = interleaved[(size_t)r * stride + 2 * c + 1];
    }
}

/* dmz.cpp:62-105, scalar branch (size is a multiple of 4) */
void orc_deinterleave_rgba_to_r(const uint8_t *source, uint8_t *dest, int size) {
  for (int i = 0; i < size; i++) dest[i] = source[4 * (size_t)i];
}

/* cv/convert.cpp:448-490: fixed-point BT.601 with 14 fractional bits, channels = 3 (RGB) or 4 (RGBA, A = 255) */
void orc_ycbcr_to_rgb(const uint8_t *y, const uint8_t *cb, const uint8_t *cr, int w, int h, int channels,
                      uint8_t *rgb) {
  for (int r = 0; r < h; r++)
    for (int c = 0; c < w; c++) {
      const size_t i = (size_t)r * w + c;
      const int pix_y = y[i];
      const int8_t sCb = (int8_t)(cb[i] - 128), sCr = (int8_t)(cr[i] - 128);
      int32_t b = pix_y + ((sCb * 29049 + (1 << 13)) >> 14);
      int32_t g = pix_y + ((sCb * -5636 + sCr * -11698 + (1 << 13)) >> 14);
      int32_t rr = pix_y + ((sCr * 22987 + (1 << 13)) >> 14);
      uint8_t *o = rgb + i * (size_t)channels;
      o[0] = (uint8_t)(rr < 0 ? 0 : (rr > 255 ? 255 : rr));
      o[1] = (uint8_t)(g < 0 ? 0 : (g > 255 ? 255 : g));
      o[2] = (uint8_t)(b < 0 ? 0 : (b > 255 ? 255 : b));
      if (channels == 4) o[3] = 0xff;
    }
}

/* ---- quality scores: dmz_focus_score / dmz_brightness_score ---------------- */
#include <math.h>

/* dmz.cpp:138-165 */
void orc_card_rect_for_screen(int card_w, int card_h, int std_w, int std_h, int act_w, int act_h, int rect[4]) {
  rect[0] = rect[1] = rect[2] = rect[3] = 0;
  if (card_w == 0 || card_h == 0 || std_w == 0 || std_h == 0 || act_w == 0 || act_h == 0) return;
  int rw, rh;
  if (act_w == std_w && act_h == std_h) {
    rw = card_w;
    rh = card_h;
  } else {
    float wr = ((float)act_w) / ((float)std_w), hr = ((float)act_h) / ((float)std_h);
    float ratio = wr < hr ? wr : hr;
    rw = (int)(card_w * ratio);
    rh = (int)(card_h * ratio);
  }
  rect[0] = (act_w - rw) / 2;
  rect[1] = (act_h - rh) / 2;
  rect[2] = rw;
  rect[3] = rh;
}

/* dmz.cpp:167-185: the centre 1/9th of the guide frame unless use_full_image */
void orc_scoring_roi(int img_w, int img_h, int use_full_image, int rect[4]) {
  const int fw = use_full_image ? ORC_CARD_W : ORC_CARD_W / 3, fh = use_full_image ? ORC_CARD_H : ORC_CARD_H / 3;
  orc_card_rect_for_screen(fw, fh, 640, 480, img_w, img_h, rect);
}

/* dmz.cpp:114-126 + 187-192: llcv_sobel3_dx_dy (sobel.cpp:556-607, indices clamped at the ROI edge),
 * cvAbs, cvAvgSdv = cv::meanStdDev: exact integer sums, mean = s * (1/N), sqrt(max(sq/N - mean^2, 0)) in double */
float orc_focus_score(const uint8_t *img, int stride, int w, int h, int use_full_image) {
  int rc[4];
  orc_scoring_roi(w, h, use_full_image, rc);
  const uint8_t *roi = img + (size_t)rc[1] * stride + rc[0];
  const int rw = rc[2], rh = rc[3];
  if (rw <= 0 || rh <= 0) return 0.0f;
  double s = 0.0, sq = 0.0;
  for (int r = 0; r < rh; r++) {
    const uint8_t *r1 = roi + (size_t)(r == 0 ? 0 : r - 1) * stride;
    const uint8_t *r2 = roi + (size_t)(r == rh - 1 ? rh - 1 : r + 1) * stride;
    for (int c = 0; c < rw; c++) {
      const int cl = c == 0 ? 0 : c - 1, cr = c == rw - 1 ? rw - 1 : c + 1;
      int d = r1[cl] - r1[cr] - r2[cl] + r2[cr];
      if (d < 0) d = -d;
      s += d;
      sq += (double)d * d;
    }
  }
  const double scale = 1. / ((double)rw * rh);
  const double mean = s * scale;
  double var = sq * scale - mean * mean;
  if (!(var > 0.)) var = 0.;
  return (float)sqrt(var);
}

/* dmz.cpp:128-135 + 194-199: cvAvg over the same ROI */
float orc_brightness_score(const uint8_t *img, int stride, int w, int h, int use_full_image) {
  int rc[4];
  orc_scoring_roi(w, h, use_full_image, rc);
  if (rc[2] <= 0 || rc[3] <= 0) return 0.0f;
  double s = 0.0;
  for (int r = 0; r < rc[3]; r++)
    for (int c = 0; c < rc[2]; c++) s += img[(size_t)(rc[1] + r) * stride + rc[0] + c];
  return (float)(s * (1. / ((double)rc[2] * rc[3])));
}

/* ---- dmz_blur_card (dmz.cpp:499-515): cv::medianBlur(25) in place on the boxes of the leading digits of
 * the RGB result image.  cv::medianBlur on a Mat made from the IplImage ROI works on a copy of the ROI padded
 * by BORDER_REPLICATE (the ROI edge, not the surrounding image) and writes the exact per-channel median of
 * the 25 x 25 window; the boxes are processed in digit order, each on the image the previous ones left. ---- */
#include <stdlib.h>
#include <string.h>

static void median_blur_roi(uint8_t *img, int stride, int ch, int x, int y, int w, int h, int ksize) {
  const int r = ksize / 2, t = ksize * ksize / 2;
  uint8_t *copy = (uint8_t *)malloc((size_t)w * h * ch);
  for (int yy = 0; yy < h; yy++) memcpy(copy + (size_t)yy * w * ch, img + (size_t)(y + yy) * stride + (size_t)x * ch, (size_t)w * ch);
  for (int yy = 0; yy < h; yy++)
    for (int xx = 0; xx < w; xx++)
      for (int c = 0; c < ch; c++) {
        int hist[256];
        memset(hist, 0, sizeof(hist));
        for (int dy = -r; dy <= r; dy++) {
          int sy = yy + dy;
          sy = sy < 0 ? 0 : (sy > h - 1 ? h - 1 : sy);
          for (int dx = -r; dx <= r; dx++) {
            int sx = xx + dx;
            sx = sx < 0 ? 0 : (sx > w - 1 ? w - 1 : sx);
            hist[copy[((size_t)sy * w + sx) * ch + c]]++;
          }
        }
        int acc = 0, v = 0;
        for (; v < 256; v++) {
          acc += hist[v];
          if (acc > t) break;
        }
        img[(size_t)(y + yy) * stride + (size_t)(x + xx) * ch + c] = (uint8_t)v;
      }
  free(copy);
}

void orc_blur_card(uint8_t *rgb, int width, int height, int channels, const uint16_t *offsets, int n_offsets,
                   float number_width, int y_offset, int unblur_digits) {
  if (unblur_digits < 0) return;
  const int blur_count = n_offsets - unblur_digits;
  for (int i = 0; i < n_offsets && i < blur_count; i++) {
    int x = offsets[i] - 1, y = y_offset - 1;
    int w = (int)(number_width + 2), h = ORC_NUM_H + 2;  /* int num_w = hseg.number_width + 2 (float sum, truncated) */
    if (i < 4) h *= 2;
    /* cvSetImageROI clips the rectangle to the image */
    int x1 = x + w, y1 = y + h;
    if (x < 0) x = 0;
    if (y < 0) y = 0;
    if (x1 > width) x1 = width;
    if (y1 > height) y1 = height;
    if (x1 <= x || y1 <= y) continue;
    median_blur_roi(rgb, width * channels, channels, x, y, x1 - x, y1 - y, 25);
  }
}
