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
