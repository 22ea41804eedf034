// dmz_host.cpp -- host side of the HIP_DMZ flavour: the reference's per-frame entry
// points (dmz.cpp:23-32,371-497; cv/warp.cpp:34-169; scan/scan.cpp:22-200;
// dmz_olm.cpp:12-130; mz.cpp) as batch-of-1 wrappers over the C-ABI of
// include/dmz_hip.h, plus the session aggregator, which is sequential per session and
// O(160) flops per frame and therefore stays on the host.
#include "dmz.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <time.h>

#include "../../include/dmz_hip.h"

namespace {

// For the entry points that take no dmz_context (dmz_detect_edges, the plumbing and score functions):
// one lazily created context per host thread -- a HIP context is driven by one thread at a time, like
// the reference's single-threaded state -- destroyed when the thread ends.
struct DefaultContext {
  dmz_hip_context *ctx = nullptr;
  bool tried = false;
  ~DefaultContext() {
    if (ctx) dmz_hip_context_destroy(ctx);
  }
};
thread_local DefaultContext g_default;

dmz_hip_context *hip_of(dmz_context *dmz) {
  if (dmz && dmz->mz) return (dmz_hip_context *)dmz->mz;
  if (!g_default.tried) {
    g_default.tried = true;
    if (dmz_hip_context_create(0, &g_default.ctx) != DMZ_HIP_OK) {
      g_default.ctx = nullptr;
      fprintf(stderr, "dmz (HIP): no usable MI355X context; there is no CPU fallback\n");
    }
  }
  return g_default.ctx;
}

const uint8_t *image_origin(const IplImage *im, int *w, int *h) {
  const uint8_t *p = (const uint8_t *)im->imageData;
  *w = im->width;
  *h = im->height;
  if (im->roi) {  // cv/image_util.cpp:35-41
    p += (size_t)im->roi->yOffset * im->widthStep + im->roi->xOffset;
    *w = im->roi->width;
    *h = im->roi->height;
  }
  return p;
}

// n_vseg.cpp:26-30 tables
const uint8_t kNumberLength[3] = {0, 16, 15};
const uint8_t kPatternLength[3] = {0, 19, 17};
const uint8_t kPatterns[3][19] = {
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {1, 1, 1, 1, 0, 1, 1, 1, 1, 0, 1, 1, 1, 1, 0, 1, 1, 1, 1},
    {1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 0},
};

void fill_frame_result(const dmz_hip_frame_result &r, FrameScanResult *out) {
  out->usable = (r.flags & DMZ_HIP_FLAG_USABLE) != 0;
  out->upside_down = (r.flags & DMZ_HIP_FLAG_UPSIDE_DOWN) != 0;
  out->vseg.score = r.vseg_score;
  out->vseg.y_offset = (uint16_t)r.vseg_y_offset;
  out->vseg.pattern_type = (NumberPatternType)r.pattern_type;
  memcpy(out->vseg.number_pattern, kPatterns[r.pattern_type], 19);
  out->vseg.number_pattern_length = kPatternLength[r.pattern_type];
  out->vseg.number_length = kNumberLength[r.pattern_type];
  out->hseg.n_offsets = (uint8_t)r.n_offsets;
  memcpy(out->hseg.offsets, r.offsets, sizeof(r.offsets));
  out->hseg.score = r.hseg_score;
  out->hseg.number_width = r.number_width;
  out->hseg.pattern_offset = (uint16_t)r.pattern_offset;
  memcpy(out->scores.v, r.scores, sizeof(r.scores));
}

}  // namespace

// ---- life cycle (dmz.cpp:23-39, mz.h:19-25) ---------------------------------------
void *mz_create(void) {
  dmz_hip_context *c = nullptr;
  return dmz_hip_context_create(0, &c) == DMZ_HIP_OK ? (void *)c : nullptr;
}
void mz_destroy(void *mz) { dmz_hip_context_destroy((dmz_hip_context *)mz); }
void mz_prepare_for_backgrounding(void *mz) {
  if (mz) dmz_hip_synchronize((dmz_hip_context *)mz);
}
int dmz_has_hip_runtime(void) { return dmz_hip_device_count() > 0; }

// dmz.cpp:41-47: the reference probes cvCreateImage; this flavour probes its own image allocator
int dmz_has_opencv(void) {
  IplImage *probe = dmz_create_image_8u(kCreditCardTargetWidth, kCreditCardTargetHeight, 1);
  const int rv = probe != NULL;
  dmz_release_image(&probe);
  return rv;
}

// processor_support.cpp:87-118.  No NEON / VFP code paths exist here; the GLES-warp switch of the
// Android flavour (:95-102) is kept as the "accelerator rectifies" switch of the HIP warp.
static int g_hip_warp_allowed = 1;
int dmz_has_neon_runtime(void) { return 0; }
int dmz_use_vfp3_16(void) { return 0; }
void dmz_set_gles_warp(int newstate) { g_hip_warp_allowed = 1 & newstate; }
int dmz_use_gles_warp(void) { return g_hip_warp_allowed && dmz_has_hip_runtime(); }

dmz_context *dmz_context_create(void) {
  dmz_context *dmz = (dmz_context *)calloc(1, sizeof(dmz_context));
  if (!dmz) return NULL;
  dmz->mz = mz_create();
  if (!dmz->mz) {  // no GPU: fail loudly instead of falling back to a CPU path
    free(dmz);
    return NULL;
  }
  return dmz;
}
void dmz_context_destroy(dmz_context *dmz) {
  if (!dmz) return;
  mz_destroy(dmz->mz);
  free(dmz);
}
void dmz_prepare_for_backgrounding(dmz_context *dmz) {
  if (dmz) mz_prepare_for_backgrounding(dmz->mz);
}

// ---- dmz_olm.cpp:12-49 -------------------------------------------------------------
dmz_point dmz_create_point(float x, float y) { dmz_point p; p.x = x; p.y = y; return p; }
dmz_rect dmz_create_rect(float x, float y, float w, float h) {
  dmz_rect r; r.x = x; r.y = y; r.w = w; r.h = h; return r;
}
void dmz_rect_get_points(dmz_rect rect, dmz_point points[4]) {
  points[0] = dmz_create_point(rect.x, rect.y);
  points[1] = dmz_create_point(rect.x + rect.w, rect.y);
  points[2] = dmz_create_point(rect.x, rect.y + rect.h);
  points[3] = dmz_create_point(rect.x + rect.w, rect.y + rect.h);
}
// dmz_olm.cpp:20-23
dmz_point dmz_scale_point(const dmz_point src_p, const dmz_rect src_f, const dmz_rect dst_f) {
  return dmz_create_point(dst_f.x + (src_p.x - src_f.x) * dst_f.w / src_f.w,
                          dst_f.y + (src_p.y - src_f.y) * dst_f.h / src_f.h);
}
// dmz_olm.cpp:134-164 with dmz_constants.h:10-27: the guide frame is the card-sized hole of the 480 x 640
// (portrait) / 640 x 480 (landscape) sample, as fractions of the preview
dmz_rect dmz_guide_frame(FrameOrientation orientation, float preview_width, float preview_height) {
  const float portrait_h = (float)((480 - kCreditCardTargetWidth) / 2) / (float)480;    // kPortraitHorizontalPercentInset
  const float portrait_v = (float)((640 - kCreditCardTargetHeight) / 2) / (float)640;   // kPortraitVerticalPercentInset
  const float landscape_v = (float)((480 - kCreditCardTargetHeight) / 2) / (float)480;  // kLandscapeVerticalPercentInset
  const float landscape_h = (float)((640 - kCreditCardTargetWidth) / 2) / (float)640;   // kLandscapeHorizontalPercentInset
  float inset_w = 0.0f, inset_h = 0.0f;
  if (orientation == FrameOrientationPortrait || orientation == FrameOrientationPortraitUpsideDown) {
    inset_w = portrait_h * preview_width;
    inset_h = portrait_v * preview_height;
  } else if (orientation == FrameOrientationLandscapeLeft || orientation == FrameOrientationLandscapeRight) {
    inset_w = landscape_v * preview_width;   // sic: the reference pairs the vertical percentage with the width
    inset_h = landscape_h * preview_height;
  }
  return dmz_create_rect(inset_w, inset_h, preview_width - 2.0f * inset_w, preview_height - 2.0f * inset_h);
}
// dmz_olm.cpp:166-179
FrameOrientation dmz_opposite_orientation(FrameOrientation orientation) {
  // pairs 1 <-> 2 and 3 <-> 4; anything else maps to portrait
  static const FrameOrientation opposite[5] = {FrameOrientationPortrait, FrameOrientationPortraitUpsideDown,
                                               FrameOrientationPortrait, FrameOrientationLandscapeLeft,
                                               FrameOrientationLandscapeRight};
  return orientation <= 4 ? opposite[orientation] : (FrameOrientation)FrameOrientationPortrait;
}
bool dmz_passes_luhn_checksum(uint8_t *number_array, uint8_t number_length) {
  int sum = 0;
  bool doubled = false;  // the rightmost digit is not doubled
  for (int i = number_length - 1; i >= 0; i--) {
    const int addend = number_array[i] * (doubled ? 2 : 1);
    sum += addend % 10 + addend / 10;
    doubled = !doubled;
  }
  return sum % 10 == 0;
}

// dmz_olm.cpp:51-130: issuer prefix ranges (data) and the single-match rule
dmz_card_info dmz_card_info_for_prefix_and_length(uint8_t *number_array, uint8_t number_length,
                                                  bool allow_incomplete_number) {
  static const dmz_card_info table[] = {
      {CardTypeMastercard, 16, 4, 2221, 2720}, {CardTypeDiscover, 14, 3, 300, 305},
      {CardTypeDiscover, 14, 3, 309, 309},     {CardTypeAmex, 15, 2, 34, 34},
      {CardTypeJCB, 16, 4, 3528, 3589},        {CardTypeDiscover, 14, 2, 36, 36},
      {CardTypeDiscover, 14, 2, 38, 39},       {CardTypeAmex, 15, 2, 37, 37},
      {CardTypeVisa, 16, 1, 4, 4},             {CardTypeMaestro, 16, 2, 50, 50},
      {CardTypeMastercard, 16, 2, 51, 55},     {CardTypeMaestro, 16, 2, 56, 59},
      {CardTypeDiscover, 16, 4, 6011, 6011},   {CardTypeMaestro, 16, 2, 61, 61},
      {CardTypeDiscover, 16, 2, 62, 62},       {CardTypeMaestro, 16, 2, 63, 63},
      {CardTypeDiscover, 16, 3, 644, 649},     {CardTypeDiscover, 16, 2, 65, 65},
      {CardTypeMaestro, 16, 2, 66, 69},        {CardTypeDiscover, 16, 2, 88, 88},
  };
  const dmz_card_info unrecognized = {CardTypeUnrecognized, -1, 1, 9, 9};
  const dmz_card_info ambiguous = {CardTypeAmbiguous, -1, 1, 9, 9};
  if (number_length == 0) return unrecognized;
  dmz_card_info match = unrecognized;
  int matches = 0;
  for (size_t t = 0; t < sizeof(table) / sizeof(table[0]); t++) {
    const dmz_card_info &info = table[t];
    if (allow_incomplete_number ? number_length > info.number_length
                                : number_length != info.number_length)
      continue;
    int plen = info.prefix_length;
    long factor = 1;
    for (; plen > number_length; plen--) factor *= 10;  // compare only the digits we have
    long prefix = 0;
    for (int j = 0; j < plen; j++) prefix = prefix * 10 + number_array[j];
    if (prefix >= info.min_prefix / factor && prefix <= info.max_prefix / factor) {
      matches++;
      match = info;
    }
  }
  if (matches == 1) return match;
  return matches > 1 ? ambiguous : unrecognized;
}

// ---- images --------------------------------------------------------------------------
IplImage *dmz_create_image_8u(int width, int height, int channels) {
  IplImage *im = (IplImage *)calloc(1, sizeof(IplImage));
  if (!im) return NULL;
  im->nSize = (int)sizeof(IplImage);
  im->nChannels = channels;
  im->depth = IPL_DEPTH_8U;
  im->align = 4;
  im->width = width;
  im->height = height;
  im->widthStep = (width * channels + 3) & ~3;  // cvCreateImage pads rows to 4 bytes
  im->imageSize = im->widthStep * height;
  im->imageData = im->imageDataOrigin = (char *)calloc(1, (size_t)im->imageSize);
  if (!im->imageData) { free(im); return NULL; }
  return im;
}
void dmz_release_image(IplImage **image) {
  if (image && *image) {
    free((*image)->imageDataOrigin);
    free(*image);
    *image = NULL;
  }
}

// ---- detection (dmz.cpp:273-439) -----------------------------------------------------
bool dmz_found_all_edges(dmz_edges e) { return e.top.found && e.bottom.found && e.left.found && e.right.found; }

bool dmz_detect_edges(IplImage *y_sample, IplImage *cb_sample, IplImage *cr_sample,
                      FrameOrientation orientation, dmz_edges *found_edges,
                      dmz_corner_points *corner_points) {
  if (!y_sample || !found_edges || !corner_points) return false;  // the reference asserts
  memset(found_edges, 0, sizeof(*found_edges));
  dmz_hip_context *ctx = hip_of(NULL);
  if (!ctx) return false;
  int w, h, cw = 0, ch = 0;
  const uint8_t *y = image_origin(y_sample, &w, &h);
  const uint8_t *cb = NULL, *cr = NULL;
  size_t cstride = 0;
  int crow = 0;
  if (cb_sample && cr_sample) {
    cb = image_origin(cb_sample, &cw, &ch);
    cr = image_origin(cr_sample, &cw, &ch);
    if (cw != w / 2 || ch != h / 2 || cb_sample->widthStep != cr_sample->widthStep) cb = cr = NULL;
    else { crow = cb_sample->widthStep; cstride = (size_t)crow * ch; }
  }
  dmz_hip_frame_result r;
  memset(&r, 0, sizeof(r));
  if (dmz_hip_detect_batch(ctx, y, (size_t)y_sample->widthStep * h, y_sample->widthStep, w, h, cb, cr,
                           cstride, crow, 1, orientation, &r) != DMZ_HIP_OK) {
    fprintf(stderr, "dmz (HIP): detect failed: %s\n", dmz_hip_last_error(ctx));
    return false;
  }
  dmz_found_edge *e[4] = {&found_edges->top, &found_edges->left, &found_edges->bottom, &found_edges->right};
  for (int i = 0; i < 4; i++) {
    e[i]->found = r.found[i];
    e[i]->location.rho = r.rho[i];
    e[i]->location.theta = r.theta[i];
  }
  if (r.found_all) {
    corner_points->top_left = dmz_create_point(r.corners[0], r.corners[1]);
    corner_points->bottom_left = dmz_create_point(r.corners[2], r.corners[3]);
    corner_points->top_right = dmz_create_point(r.corners[4], r.corners[5]);
    corner_points->bottom_right = dmz_create_point(r.corners[6], r.corners[7]);
  }
  return r.found_all != 0;
}

// ---- transformation (dmz.cpp:443-497, cv/warp.cpp:26-169) ----------------------------
bool llcv_warp_auto_upsamples(void) { return false; }  // warp.cpp:26-32, non-iOS

void llcv_calc_persp_transform(float *matrixData, int matrixDataSize, bool rowMajor,
                               const dmz_point sourcePoints[], const dmz_point destPoints[]) {
  for (int i = 0; i < matrixDataSize; i++) matrixData[i] = 0.0f;
  dmz_hip_context *ctx = hip_of(NULL);
  if (!ctx) return;
  float s[8], d[8], m[9];
  for (int i = 0; i < 4; i++) {
    s[2 * i] = sourcePoints[i].x; s[2 * i + 1] = sourcePoints[i].y;
    d[2 * i] = destPoints[i].x; d[2 * i + 1] = destPoints[i].y;
  }
  if (dmz_hip_calc_persp_transform(ctx, s, d, m) != DMZ_HIP_OK) return;
  // warp.cpp:84-121: 3x3, or the 4x4 layout with the projective row/column moved out
  const int size = matrixDataSize >= 16 ? 4 : 3;
  float p[4][4];
  memset(p, 0, sizeof(p));
  const int o = size - 3;
  p[0][0] = m[0]; p[0][1] = m[1]; p[1][0] = m[3]; p[1][1] = m[4]; p[2][2] = 1.0f;
  p[0][2 + o] = m[2]; p[1][2 + o] = m[5]; p[2 + o][0] = m[6]; p[2 + o][1] = m[7]; p[2 + o][2 + o] = 1.0f;
  for (int c = 0; c < size; c++)
    for (int r = 0; r < size; r++) {
      const int index = rowMajor ? (c + r * size) : (r + c * size);
      if (index < matrixDataSize) matrixData[index] = p[r][c];
    }
}

void llcv_unwarp(dmz_context *dmz, IplImage *input, const dmz_point source_points[4],
                 const dmz_rect to_rect, IplImage *output) {
  dmz_hip_context *ctx = hip_of(dmz);
  if (!ctx || !input || !output || !output->imageData) return;
  if (!g_hip_warp_allowed) {  // dmz_set_gles_warp(0): the reference would fall back to its CPU warp; there is none here
    fprintf(stderr, "dmz (HIP): llcv_unwarp: the accelerator warp was switched off (dmz_set_gles_warp(0)) and "
                    "there is no CPU fallback\n");
    return;
  }
  if (input->nChannels != 1 || output->nChannels != 1 || output->width != kCreditCardTargetWidth ||
      output->height != kCreditCardTargetHeight || output->widthStep != kCreditCardTargetWidth) {
    fprintf(stderr, "dmz (HIP): llcv_unwarp handles 1-channel 428x270 outputs only\n");
    return;
  }
  dmz_point dest[4];
  dmz_rect_get_points(to_rect, dest);
  float m[9];
  llcv_calc_persp_transform(m, 9, true, source_points, dest);
  int w, h;
  const uint8_t *p = image_origin(input, &w, &h);
  if (dmz_hip_warp_perspective_batch(ctx, p, (size_t)input->widthStep * h, input->widthStep, w, h, 1, m,
                                     (uint8_t *)output->imageData,
                                     (size_t)kCreditCardTargetWidth * kCreditCardTargetHeight) != DMZ_HIP_OK)
    fprintf(stderr, "dmz (HIP): warp failed: %s\n", dmz_hip_last_error(ctx));
}

void dmz_transform_card(dmz_context *dmz, IplImage *sample, dmz_corner_points corner_points,
                        FrameOrientation orientation, bool upsample, IplImage **transformed) {
  dmz_point src[4];
  switch (orientation) {  // dmz.cpp:446-471
    case FrameOrientationPortrait:
      src[0] = corner_points.bottom_left; src[1] = corner_points.top_left;
      src[2] = corner_points.bottom_right; src[3] = corner_points.top_right;
      break;
    case FrameOrientationLandscapeLeft:
      src[0] = corner_points.bottom_right; src[1] = corner_points.bottom_left;
      src[2] = corner_points.top_right; src[3] = corner_points.top_left;
      break;
    case FrameOrientationPortraitUpsideDown:
      src[0] = corner_points.top_right; src[1] = corner_points.bottom_right;
      src[2] = corner_points.top_left; src[3] = corner_points.bottom_left;
      break;
    default:  // FrameOrientationLandscapeRight, "the canonical one"
      src[0] = corner_points.top_left; src[1] = corner_points.top_right;
      src[2] = corner_points.bottom_left; src[3] = corner_points.bottom_right;
      break;
  }
  if (upsample && !llcv_warp_auto_upsamples())
    for (int i = 0; i < 4; i++) { src[i].x /= 2.0f; src[i].y /= 2.0f; }
  const dmz_rect dst = dmz_create_rect(0, 0, kCreditCardTargetWidth - 1, kCreditCardTargetHeight - 1);
  if (*transformed == NULL)
    *transformed = dmz_create_image_8u(kCreditCardTargetWidth, kCreditCardTargetHeight, sample->nChannels);
  llcv_unwarp(dmz, sample, src, dst, *transformed);
}

// ---- camera-side plumbing (dmz.cpp:49-105) --------------------------------------------
// The reference's versions take no dmz_context; like dmz_detect_edges they run on the process
// default context.  Rows are repacked when widthStep carries padding.
static uint8_t *packed_rows(const IplImage *im, int bytes_per_px, bool *owned) {
  const int row = im->width * bytes_per_px;
  if (im->widthStep == row) {
    *owned = false;
    return (uint8_t *)im->imageData;
  }
  uint8_t *p = (uint8_t *)malloc((size_t)row * im->height);
  for (int r = 0; r < im->height; r++) memcpy(p + (size_t)r * row, im->imageData + (size_t)r * im->widthStep, row);
  *owned = true;
  return p;
}
static void unpack_rows(IplImage *im, int bytes_per_px, const uint8_t *p) {
  const int row = im->width * bytes_per_px;
  for (int r = 0; r < im->height; r++) memcpy(im->imageData + (size_t)r * im->widthStep, p + (size_t)r * row, row);
}

void dmz_deinterleave_uint8_c2(IplImage *interleaved, IplImage **channel1, IplImage **channel2) {
  dmz_hip_context *ctx = hip_of(NULL);
  if (!ctx || !interleaved || interleaved->nChannels != 2 || !channel1 || !channel2) return;
  if (!*channel1) *channel1 = dmz_create_image_8u(interleaved->width, interleaved->height, 1);
  if (!*channel2) *channel2 = dmz_create_image_8u(interleaved->width, interleaved->height, 1);
  const size_t n = (size_t)interleaved->width * interleaved->height;
  bool owned;
  uint8_t *src = packed_rows(interleaved, 2, &owned);
  uint8_t *c1 = (uint8_t *)malloc(n), *c2 = (uint8_t *)malloc(n);
  if (dmz_hip_deinterleave_c2(ctx, src, n, c1, c2) != DMZ_HIP_OK)
    fprintf(stderr, "dmz (HIP): deinterleave failed: %s\n", dmz_hip_last_error(ctx));
  unpack_rows(*channel1, 1, c1);
  unpack_rows(*channel2, 1, c2);
  free(c1);
  free(c2);
  if (owned) free(src);
}

void dmz_deinterleave_RGBA_to_R(uint8_t *source, uint8_t *dest, int size) {
  dmz_hip_context *ctx = hip_of(NULL);
  if (!ctx || !source || !dest || size <= 0) return;
  if (dmz_hip_deinterleave_rgba_to_r(ctx, source, dest, (size_t)size) != DMZ_HIP_OK)
    fprintf(stderr, "dmz (HIP): deinterleave failed: %s\n", dmz_hip_last_error(ctx));
}

void dmz_YCbCr_to_RGB(IplImage *y, IplImage *cb, IplImage *cr, IplImage **rgb) {
  dmz_hip_context *ctx = hip_of(NULL);
  if (!ctx || !y || !cb || !cr || !rgb) return;
  if (*rgb == NULL) *rgb = dmz_create_image_8u(y->width, y->height, 3);
  const int ch = (*rgb)->nChannels;
  const size_t n = (size_t)y->width * y->height;
  bool oy, ob, orr;
  uint8_t *py = packed_rows(y, 1, &oy), *pb = packed_rows(cb, 1, &ob), *pr = packed_rows(cr, 1, &orr);
  uint8_t *out = (uint8_t *)malloc(n * ch);
  if (dmz_hip_ycbcr_to_rgb(ctx, py, pb, pr, n, ch, out) != DMZ_HIP_OK)
    fprintf(stderr, "dmz (HIP): colour conversion failed: %s\n", dmz_hip_last_error(ctx));
  unpack_rows(*rgb, ch, out);
  free(out);
  if (oy) free(py);
  if (ob) free(pb);
  if (orr) free(pr);
}

// ---- quality scores (dmz.cpp:114-199) ---------------------------------------------------
static float score_of(IplImage *image, bool use_full_image, bool want_focus) {
  dmz_hip_context *ctx = hip_of(NULL);
  if (!ctx || !image || image->nChannels != 1) return 0.0f;
  float v = 0.0f;
  const int rc = dmz_hip_scores_batch(ctx, (const uint8_t *)image->imageData, (size_t)image->widthStep * image->height,
                                      image->widthStep, image->width, image->height, 1, use_full_image,
                                      want_focus ? &v : NULL, want_focus ? NULL : &v);
  if (rc != DMZ_HIP_OK) fprintf(stderr, "dmz (HIP): score failed: %s\n", dmz_hip_last_error(ctx));
  return v;
}
float dmz_focus_score(IplImage *image, bool use_full_image) { return score_of(image, use_full_image, true); }
float dmz_brightness_score(IplImage *image, bool use_full_image) { return score_of(image, use_full_image, false); }

// ---- session aggregator (scan/scan.cpp:22-200) ---------------------------------------
#define kDecayFactor 0.8f
#define kMinStability 0.7f
#define EXTRA_TIME_FOR_EXPIRY_IN_MICROSECONDS 1000

void scanner_initialize(ScannerState *state) {
  state->dmz = NULL;
  scanner_reset(state);
}

void scanner_reset(ScannerState *state) {
  state->count15 = state->count16 = 0;
  state->session_analytics.num_frames_scanned = 0;  // scan_analytics_init, scan_analytics.cpp:27-30
  state->session_analytics.frames_ring_start = 0;
  memset(&state->aggregated15, 0, sizeof(NumberScores));
  memset(&state->aggregated16, 0, sizeof(NumberScores));
  state->timeOfCardNumberCompletionInMilliseconds = 0;
  state->scan_expiry = false;
  state->expiry_month = state->expiry_year = 0;
  state->expiry_groups.clear();
  state->name_groups.clear();
}

void scanner_add_frame(ScannerState *state, IplImage *y, FrameScanResult *result) {
  scanner_add_frame_with_expiry(state, y, false, result);
}

// ---- expiry_categorize.cpp:162-330, the cross-frame (session) half of expiry_extract ----
#define GROUPED_RECTS_VERTICAL_ALLOWANCE (kTrimmedCharacterImageHeight / 2)
#define GROUPED_RECTS_HORIZONTAL_ALLOWANCE (kTrimmedCharacterImageWidth / 2)
#define kExpiryDecayFactor 0.7f
#define kExpiryMinStability 0.7f

static bool g_allow_past_expiry = false;
void dmz_hip_host_allow_past_expiry(bool allow) { g_allow_past_expiry = allow; }

// ---- cross-frame aggregation of the expiry groups: the BEHAVIOUR of expiry_categorize.cpp:256-445, in this file's own
// structure (the device's batched form of the same policy is session.hip) ----
namespace {

// two groups are "the same place on the card" when their origins are within half a character box and they hold the
// same number of characters
inline bool same_place(int top, int left, size_t n_chars, const GroupedRects &g) {
  return abs(g.top - top) <= GROUPED_RECTS_VERTICAL_ALLOWANCE && abs(g.left - left) <= GROUPED_RECTS_HORIZONTAL_ALLOWANCE &&
         g.character_rects.size() == n_chars;
}

// dst = (dst * wa + src * wb) / div, element by element in float, in that association (div == 1: no division)
inline void blend_scores(ExpiryGroupScores &dst, float wa, const ExpiryGroupScores &src, float wb, float div) {
  float *d = &dst.v[0][0];
  const float *q = &src.v[0][0];
  for (int i = 0; i < kExpiryMaxValidLength * 10; i++) {
    const float t = d[i] * wa + q[i] * wb;
    d[i] = div == 1.0f ? t : t / div;
  }
}

template <class T>
void drop_marked(std::vector<T> &v, const std::vector<char> &gone) {
  size_t w = 0;
  for (size_t r = 0; r < v.size(); r++)
    if (!gone[r]) {
      if (w != r) v[w] = v[r];
      w++;
    }
  v.resize(w);
}

}  // namespace

void expiry_aggregate_grouped_rects(GroupedRectsList &aggregated_groups, GroupedRectsList &new_groups) {
  const size_t n_new = new_groups.size();
  std::vector<char> merged(n_new, 0);  // a frame's group that went into another group
  // 1. within the frame: the LATER groups at an earlier group's place fold into it, last one first, as a running mean
  for (size_t keep = 0; keep < n_new; keep++) {
    if (merged[keep]) continue;
    GroupedRects &k = new_groups[keep];
    const int top = k.top, left = k.left;
    const size_t n_chars = k.character_rects.size();
    float members = 1.0f;
    for (size_t j = n_new; j-- > keep + 1;) {
      if (merged[j] || !same_place(top, left, n_chars, new_groups[j])) continue;
      blend_scores(k.scores, members, new_groups[j].scores, 1.0f, members + 1.0f);
      members += 1.0f;
      merged[j] = 1;
    }
  }
  // 2. into the session: every frame group at a session group's place (the place the session group had BEFORE this
  // frame) refreshes it -- exponential decay of the scores, the origin follows the newest sighting
  for (size_t o = 0; o < aggregated_groups.size(); o++) {
    GroupedRects &old = aggregated_groups[o];
    const int top = old.top, left = old.left;
    const size_t n_chars = old.character_rects.size();
    for (size_t j = n_new; j-- > 0;) {
      if (merged[j] || !same_place(top, left, n_chars, new_groups[j])) continue;
      old.recently_seen_count++;
      old.total_seen_count++;
      blend_scores(old.scores, kExpiryDecayFactor, new_groups[j].scores, 1 - kExpiryDecayFactor, 1.0f);
      old.top = new_groups[j].top;
      old.left = new_groups[j].left;
      merged[j] = 1;
    }
  }
  // 3. every session group ages by one frame; the ones not seen lately are forgotten
  {
    std::vector<char> stale(aggregated_groups.size(), 0);
    for (size_t o = 0; o < aggregated_groups.size(); o++) stale[o] = --aggregated_groups[o].recently_seen_count <= 0;
    drop_marked(aggregated_groups, stale);
  }
  // 4. what is left of the frame's groups is new to the session: three frames of grace
  drop_marked(new_groups, merged);
  for (size_t j = 0; j < new_groups.size(); j++) {
    aggregated_groups.push_back(new_groups[j]);
    aggregated_groups.back().recently_seen_count = 3;
    aggregated_groups.back().total_seen_count = 1;
  }
}

namespace {

// Is (month, two-digit year) a date to report, and better than the one already held?  Returns the four-digit year or 0.
int acceptable_expiry_year(int month, int year2, int held_month, int held_year) {
  if (month <= 0 || month > 12) return 0;
  int year4 = year2 + 2000;
  if (!(year4 > held_year || (year4 == held_year && month > held_month))) return 0;  // no later than the date held
  const time_t now = time(NULL);
  const struct tm *t = localtime(&now);
  const int this_year = t->tm_year + 1900, this_month = t->tm_mon + 1;
  const bool in_window = year4 < this_year + 5;
  if (in_window && (year4 > this_year || (year4 == this_year && month >= this_month))) return year4;
  if (!g_allow_past_expiry) return 0;
  // test builds of the reference (DMZ_DEBUG || CYTHON_DMZ) also take dates in the past, two-digit years above 60 as 19YY
  if (year2 > 60) year4 = year2 + 1900;
  return year4 < this_year + 5 ? year4 : 0;
}

}  // namespace

void get_stable_expiry_month_and_year(GroupedRects &group, int *expiry_month, int *expiry_year) {
  // a character counts when its best class holds at least kExpiryMinStability of the row's score mass
  int digit[kExpiryMaxValidLength];
  for (int i = 0; i < kExpiryMaxValidLength; i++) digit[i] = -1;
  const size_t n = group.character_rects.size() < (size_t)kExpiryMaxValidLength ? group.character_rects.size()
                                                                                  : (size_t)kExpiryMaxValidLength;
  for (size_t i = 0; i < n; i++) {
    const float *p = group.scores.v[i];
    int best = 0;  // (first maximum, as Eigen's maxCoeff visits them)
    for (int k = 1; k < 10; k++)
      if (p[k] > p[best]) best = k;
    // the row sum in the order of Eigen's ten-element reduction tree
    const float mass = ((p[0] + p[1]) + (p[2] + (p[3] + p[4]))) + ((p[5] + p[6]) + (p[7] + (p[8] + p[9])));
    if (!(p[best] / mass < kExpiryMinStability)) digit[i] = best;
  }
  if (group.pattern != ExpiryPatternMMsYY) return;  // the only pattern the reference reads (MM/YY; position 2 is the slash)
  if (digit[0] < 0 || digit[1] < 0 || digit[3] < 0 || digit[4] < 0) return;
  int month = digit[0] * 10 + digit[1], year2 = digit[3] * 10 + digit[4];
  if (month > 12 && year2 > 0 && year2 <= 12) std::swap(month, year2);  // YY/MM cards
  const int year4 = acceptable_expiry_year(month, year2, *expiry_month, *expiry_year);
  if (year4 > 0) {
    *expiry_month = month;
    *expiry_year = year4;
  }
}

// FrameScanResult.expiry_groups from the device record (expiry_seg.cpp:651-668 field values)
static void fill_expiry_groups(const dmz_hip_expiry_result &x, GroupedRectsList *out) {
  out->clear();
  for (int g = 0; g < x.n_groups; g++) {
    const dmz_hip_expiry_group &d = x.groups[g];
    GroupedRects gr;
    gr.top = d.top;
    gr.left = d.left;
    gr.width = d.width;
    gr.height = d.height;
    gr.grouped_yet = false;
    gr.sum = 0;
    gr.character_width = kTrimmedCharacterImageWidth;
    gr.pattern = ExpiryPatternMMsYY;
    gr.recently_seen_count = 0;
    gr.total_seen_count = 0;
    memset(&gr.scores, 0, sizeof(gr.scores));
    for (int i = 0; i < 5; i++) gr.character_rects.push_back(CharacterRect(d.char_top[i], d.char_left[i], 0));
    for (int row = 0; row < 4; row++) memcpy(gr.scores.v[row < 2 ? row : row + 1], d.scores[row], sizeof(float) * 10);
    out->push_back(gr);
  }
}

// scan_card_image (scan/frame.cpp:24-81) on the device: the number path, and the expiry path of frame.cpp:71-73 when asked.
// Returns false where the reference asserts (frame.cpp:25-29) or when the device call fails.
static bool scan_card_image_on(dmz_hip_context *ctx, IplImage *y, bool collect_card_number, bool scan_expiry,
                               FrameScanResult *result) {
  result->usable = false;
  result->upside_down = false;
  result->expiry_groups.clear();
  result->name_groups.clear();
  if (!ctx || !y || y->roi || y->width != kCreditCardTargetWidth || y->height != kCreditCardTargetHeight ||
      y->nChannels != 1)
    return false;  // frame.cpp:25-29 asserts these
  // rows must be 428 bytes apart
  const uint8_t *cards = (const uint8_t *)y->imageData;
  uint8_t *packed = NULL;
  if (y->widthStep != kCreditCardTargetWidth) {
    packed = (uint8_t *)malloc((size_t)kCreditCardTargetWidth * kCreditCardTargetHeight);
    for (int r = 0; r < kCreditCardTargetHeight; r++)
      memcpy(packed + (size_t)r * kCreditCardTargetWidth, y->imageData + (size_t)r * y->widthStep, kCreditCardTargetWidth);
    cards = packed;
  }
  const size_t card_stride = (size_t)kCreditCardTargetWidth * kCreditCardTargetHeight;
  dmz_hip_frame_result r;
  memset(&r, 0, sizeof(r));
  int rc = dmz_hip_scan_cards_batch(ctx, cards, card_stride, 1, collect_card_number ? 0 : DMZ_HIP_SCAN_SKIP_NUMBER, &r);
  dmz_hip_expiry_result x;
  memset(&x, 0, sizeof(x));
  if (rc == DMZ_HIP_OK && scan_expiry) rc = dmz_hip_scan_expiry_batch(ctx, cards, card_stride, 1, &r, &x);  // frame.cpp:71-73
  free(packed);
  if (rc != DMZ_HIP_OK) {
    fprintf(stderr, "dmz (HIP): scan failed: %s\n", dmz_hip_last_error(ctx));
    return false;
  }
  fill_frame_result(r, result);
  if (scan_expiry) fill_expiry_groups(x, &result->expiry_groups);
  return true;
}

void scan_card_image(IplImage *y, bool collect_card_number, bool scan_expiry, FrameScanResult *result) {
  (void)scan_card_image_on(hip_of(NULL), y, collect_card_number, scan_expiry, result);
}

// scan/frame.cpp:84-98 (the Cython flavour's entry point; the sequence BASELINE configs[0] ends in)
void cython_scan_card_image(IplImage *y, CythonFrameScanResult *result) {
  FrameScanResult frameScanResult;
  frameScanResult.focus_score = 666;
  frameScanResult.brightness_score = 150;
  frameScanResult.iso_speed = 400;
  frameScanResult.shutter_speed = 5;
  frameScanResult.torch_is_on = 0;
  frameScanResult.flipped = 0;
  memset(&frameScanResult.hseg, 0, sizeof(frameScanResult.hseg));
  memset(&frameScanResult.vseg, 0, sizeof(frameScanResult.vseg));
  // (the reference also segments the expiry here and drops it: only usable / hseg / vseg leave the call)
  scan_card_image(y, true, false, &frameScanResult);
  result->usable = frameScanResult.usable;
  result->hseg = frameScanResult.hseg;
  result->vseg = frameScanResult.vseg;
}

void scanner_add_frame_with_expiry(ScannerState *state, IplImage *y, bool scan_expiry,
                                   FrameScanResult *result) {
  const bool need_number = state->timeOfCardNumberCompletionInMilliseconds == 0;
  const bool need_expiry = scan_expiry && (state->expiry_month == 0 || state->expiry_year == 0);  // scan.cpp:44
  // scan_card_image(y, still_need_to_collect_card_number, ...), scan.cpp:48
  if (!scan_card_image_on(hip_of(state->dmz), y, need_number, need_expiry, result)) return;
  if (result->upside_down) return;                     // scan.cpp:49-51
  {                                                    // scan_analytics_record_frame (scan.cpp:53): frame counter + ring
    ScanSessionAnalytics *sa = &state->session_analytics;
    if (sa->num_frames_scanned > kScanSessionNumFramesStored)
      sa->frames_ring_start = (uint8_t)((sa->num_frames_scanned + 1) % kScanSessionNumFramesStored);
    sa->frames_ring[sa->num_frames_scanned % kScanSessionNumFramesStored].frame_index = sa->num_frames_scanned;
    sa->num_frames_scanned += 1;
  }
  if (!result->usable) return;                         // scan.cpp:57-59
  if (need_expiry) {                                   // scan.cpp:61-67 + expiry_extract (:332-376)
    state->scan_expiry = true;
    if (!result->expiry_groups.empty()) {
      // the device categorised the digits of every group of this (usable) frame already
      expiry_aggregate_grouped_rects(state->expiry_groups, result->expiry_groups);
      for (GroupedRectsList::iterator group = state->expiry_groups.begin(); group != state->expiry_groups.end(); ++group) {
        if (group->total_seen_count < 3) continue;  // not trusted yet
        get_stable_expiry_month_and_year(*group, &state->expiry_month, &state->expiry_year);
      }
    }
    state->name_groups = result->name_groups;
  }
  if (need_number) {                                   // scan.cpp:69-85
    state->mostRecentUsableHSeg = result->hseg;
    state->mostRecentUsableVSeg = result->vseg;
    NumberScores *agg = result->hseg.n_offsets == 15 ? &state->aggregated15
                      : result->hseg.n_offsets == 16 ? &state->aggregated16 : NULL;
    if (agg) {
      for (int i = 0; i < 16; i++)
        for (int k = 0; k < 10; k++) {
          agg->v[i][k] = agg->v[i][k] * kDecayFactor;
          agg->v[i][k] = agg->v[i][k] + result->scores.v[i][k] * (1 - kDecayFactor);
        }
      if (result->hseg.n_offsets == 15) state->count15++;
      else state->count16++;
    }
  }
}

void scanner_result(ScannerState *state, ScannerResult *result) {
  result->complete = false;
  if (state->timeOfCardNumberCompletionInMilliseconds > 0) {
    *result = state->successfulCardNumberResult;
  } else {
    const uint16_t max_count = state->count15 > state->count16 ? state->count15 : state->count16;
    const uint16_t min_count = state->count15 > state->count16 ? state->count16 : state->count15;
    if (max_count - min_count < 3) return;   // at least a three frame lead (scan.cpp:103-105)
    if (min_count * 2 > max_count) return;   // a significant 15-vs-16 opinion (scan.cpp:108-110)
    result->hseg = state->mostRecentUsableHSeg;
    result->vseg = state->mostRecentUsableVSeg;
    const NumberScores *agg;
    if (state->count15 > state->count16) { result->n_numbers = 15; agg = &state->aggregated15; }
    else { result->n_numbers = 16; agg = &state->aggregated16; }
    uint8_t digits[16];
    for (int i = 0; i < result->n_numbers; i++) {
      int best = 0;  // Eigen maxCoeff: first maximum
      for (int k = 1; k < 10; k++)
        if (agg->v[i][k] > agg->v[i][best]) best = k;
      // Eigen 10-element redux tree
      const float *p = agg->v[i];
      const float sum = ((p[0] + p[1]) + (p[2] + (p[3] + p[4]))) + ((p[5] + p[6]) + (p[7] + (p[8] + p[9])));
      result->predictions.v[i] = best;
      digits[i] = (uint8_t)best;
      if (agg->v[i][best] / sum < kMinStability) return;  // scan.cpp:143-146
    }
    const CardType type = dmz_card_info_for_prefix_and_length(digits, result->n_numbers, false).card_type;
    if (type != CardTypeAmbiguous && type != CardTypeUnrecognized &&
        dmz_passes_luhn_checksum(digits, result->n_numbers)) {
      struct timeval tv;
      gettimeofday(&tv, NULL);
      state->timeOfCardNumberCompletionInMilliseconds = (long)((tv.tv_sec * 1000) + (tv.tv_usec / 1000));
      state->successfulCardNumberResult = *result;
    }
  }
  if (state->timeOfCardNumberCompletionInMilliseconds > 0) {
    if (state->scan_expiry) {  // only set once the expiry path exists (scan.cpp:61-67)
      struct timeval tv;
      gettimeofday(&tv, NULL);
      const long now = (long)((tv.tv_sec * 1000) + (tv.tv_usec / 1000));
      if ((state->expiry_month > 0 && state->expiry_year > 0) ||
          now - (long)state->timeOfCardNumberCompletionInMilliseconds > EXTRA_TIME_FOR_EXPIRY_IN_MICROSECONDS) {
        result->expiry_month = state->expiry_month;
        result->expiry_year = state->expiry_year;
        result->complete = true;
      }
    } else {
      result->expiry_month = 0;
      result->expiry_year = 0;
      result->complete = true;
    }
  }
}

void scanner_destroy(ScannerState *state) { (void)state; }

// ---- dmz_blur_card (dmz.cpp:499-515) -----------------------------------------------------
void dmz_blur_card(IplImage *cardImageRGB, ScannerState *state, int unblurDigits) {
  dmz_hip_context *ctx = hip_of(state ? state->dmz : NULL);
  if (!ctx || !cardImageRGB || !state || unblurDigits < 0) return;
  if (cardImageRGB->width != kCreditCardTargetWidth || cardImageRGB->height != kCreditCardTargetHeight ||
      (cardImageRGB->nChannels != 3 && cardImageRGB->nChannels != 4))
    return;
  const int ch = cardImageRGB->nChannels;
  dmz_hip_session_result s;
  memset(&s, 0, sizeof(s));
  s.n_offsets = state->mostRecentUsableHSeg.n_offsets;
  memcpy(s.offsets, state->mostRecentUsableHSeg.offsets, sizeof(s.offsets));
  s.number_width = state->mostRecentUsableHSeg.number_width;
  s.vseg_y_offset = state->mostRecentUsableVSeg.y_offset;
  bool owned;
  uint8_t *p = packed_rows(cardImageRGB, ch, &owned);
  if (!owned) {
    if (dmz_hip_blur_cards_batch(ctx, p, (size_t)kCreditCardTargetWidth * kCreditCardTargetHeight * ch, ch, 1, &s, unblurDigits) != DMZ_HIP_OK)
      fprintf(stderr, "dmz (HIP): blur failed: %s\n", dmz_hip_last_error(ctx));
  } else {
    if (dmz_hip_blur_cards_batch(ctx, p, (size_t)kCreditCardTargetWidth * kCreditCardTargetHeight * ch, ch, 1, &s, unblurDigits) == DMZ_HIP_OK)
      unpack_rows(cardImageRGB, ch, p);
    free(p);
  }
}

// ---- the Cython flavour's entry points (dmz.cpp:517-674, mz.cpp: py_mz_*) ----------------------------------
static void image_extent(const IplImage *im, int *x, int *y, int *w, int *h) {
  *x = im->roi ? im->roi->xOffset : 0, *y = im->roi ? im->roi->yOffset : 0;
  *w = im->roi ? im->roi->width : im->width, *h = im->roi ? im->roi->height : im->height;
}

void dmz_scharr3_dx_abs(IplImage *src, IplImage *dst) {  // llcv_scharr3_dx_abs (cv/sobel.cpp:706-804)
  dmz_hip_context *ctx = hip_of(NULL);
  if (!ctx || !src || !dst || src->nChannels != 1 || dst->nChannels != 1 || src->depth != IPL_DEPTH_8U) return;
  // the reference asserts a 16-bit signed destination (sobel.cpp:712-716); an 8-bit one of the same size would be overrun
  if (dst->depth != IPL_DEPTH_16S || dst->widthStep < 2 * dst->width) {
    fprintf(stderr, "dmz (HIP): dmz_scharr3_dx_abs needs an IPL_DEPTH_16S destination\n");
    return;
  }
  int sx, sy, sw, sh, dx, dy, dw, dh;
  image_extent(src, &sx, &sy, &sw, &sh);
  image_extent(dst, &dx, &dy, &dw, &dh);
  if (sw != dw || sh != dh) return;  // (the reference asserts)
  const uint8_t *sp = (const uint8_t *)src->imageData + (size_t)sy * src->widthStep + sx;
  int16_t *dp = (int16_t *)(dst->imageData + (size_t)dy * dst->widthStep) + dx;
  if (dmz_hip_scharr3_dx_abs(ctx, sp, src->widthStep, sw, sh, dp, dst->widthStep / 2) != DMZ_HIP_OK)
    fprintf(stderr, "dmz (HIP): scharr failed: %s\n", dmz_hip_last_error(ctx));
}

static CythonGroupedRects to_cython_group(const GroupedRects &g) {  // dmz.cpp:546-577
  CythonGroupedRects c;
  memset(&c, 0, sizeof(c));
  c.top = g.top, c.left = g.left, c.width = g.width, c.height = g.height;
  c.character_width = g.character_width;
  c.pattern = (uint8_t)g.pattern;
  for (int i = 0; i < kExpiryMaxValidLength; i++)
    for (int d = 0; d < 10; d++) c.scores[i][d] = g.scores(i, d);
  c.recently_seen_count = g.recently_seen_count;
  c.total_seen_count = g.total_seen_count;
  c.number_of_character_rects = (int)g.character_rects.size();
  c.character_rects = (CythonCharacterRect *)malloc(sizeof(CythonCharacterRect) * (g.character_rects.size() + 1));
  for (size_t i = 0; i < g.character_rects.size(); i++)
    c.character_rects[i].top = g.character_rects[i].top, c.character_rects[i].left = g.character_rects[i].left;
  return c;
}
static GroupedRects from_cython_group(const CythonGroupedRects *c) {  // dmz.cpp:580-601
  GroupedRects g;
  g.top = c->top, g.left = c->left, g.width = c->width, g.height = c->height;
  g.grouped_yet = false;
  g.sum = 0;
  g.character_width = c->character_width;
  g.pattern = (ExpiryPattern)c->pattern;
  for (int i = 0; i < kExpiryMaxValidLength; i++)
    for (int d = 0; d < 10; d++) g.scores(i, d) = c->scores[i][d];
  g.recently_seen_count = c->recently_seen_count;
  g.total_seen_count = c->total_seen_count;
  for (int i = 0; i < c->number_of_character_rects; i++)
    g.character_rects.push_back(CharacterRect(c->character_rects[i].top, c->character_rects[i].left, 0));
  return g;
}

// the card as the packed 428 x 270 bytes the device entries take (NULL: not a card image)
static uint8_t *card_bytes(IplImage *card_y, bool *owned) {
  if (!card_y || card_y->roi || card_y->width != kCreditCardTargetWidth || card_y->height != kCreditCardTargetHeight ||
      card_y->nChannels != 1 || card_y->depth != IPL_DEPTH_8U)
    return NULL;
  return packed_rows(card_y, 1, owned);
}

void dmz_best_expiry_seg(IplImage *card_y, uint16_t starting_y_offset, CythonGroupedRects **expiry_groups,
                         uint16_t *number_of_groups) {
  if (expiry_groups) *expiry_groups = NULL;
  if (number_of_groups) *number_of_groups = 0;
  dmz_hip_context *ctx = hip_of(NULL);
  bool owned = false;
  uint8_t *p = ctx && expiry_groups && number_of_groups ? card_bytes(card_y, &owned) : NULL;
  if (!p) return;
  // best_expiry_seg (expiry_seg.cpp:707-902) runs whenever the number row is known (frame.cpp:71-73); no digit is categorised
  dmz_hip_frame_result r;
  dmz_hip_expiry_result x;
  memset(&r, 0, sizeof(r));
  memset(&x, 0, sizeof(x));
  r.flags = DMZ_HIP_FLAG_VSEG_OK;
  r.vseg_y_offset = starting_y_offset;
  if (dmz_hip_scan_expiry_batch(ctx, p, (size_t)kCreditCardTargetWidth * kCreditCardTargetHeight, 1, &r, &x) == DMZ_HIP_OK) {
    GroupedRectsList groups;
    fill_expiry_groups(x, &groups);
    *expiry_groups = (CythonGroupedRects *)malloc(sizeof(CythonGroupedRects) * (groups.size() + 1));
    for (size_t i = 0; i < groups.size(); i++) (*expiry_groups)[i] = to_cython_group(groups[i]);
    *number_of_groups = (uint16_t)groups.size();
  } else {
    fprintf(stderr, "dmz (HIP): expiry segmentation failed: %s\n", dmz_hip_last_error(ctx));
  }
  if (owned) free(p);
}

// categorize_expiry_digits (expiry_categorize.cpp:138-160) for caller-supplied groups: characters 0, 1, 3, 4 of every group
static bool categorize_groups(IplImage *card_y, GroupedRectsList &groups) {
  dmz_hip_context *ctx = hip_of(NULL);
  bool owned = false;
  uint8_t *p = ctx ? card_bytes(card_y, &owned) : NULL;
  if (!p) return false;
  bool ok = true;
  for (size_t base = 0; base < groups.size() && ok; base += DMZ_HIP_EXPIRY_MAX_GROUPS) {
    dmz_hip_expiry_result x;
    memset(&x, 0, sizeof(x));
    const size_t m = groups.size() - base < (size_t)DMZ_HIP_EXPIRY_MAX_GROUPS ? groups.size() - base : DMZ_HIP_EXPIRY_MAX_GROUPS;
    for (size_t i = 0; i < m && ok; i++) {
      const GroupedRects &g = groups[base + i];
      ok = g.character_rects.size() == 5;  // ExpiryPatternMMsYY: the only pattern the segmentation produces
      for (int c = 0; c < 5 && ok; c++)
        x.groups[i].char_top[c] = (int16_t)g.character_rects[c].top, x.groups[i].char_left[c] = (int16_t)g.character_rects[c].left;
    }
    x.n_groups = (int)m;
    ok = ok && dmz_hip_categorize_expiry_groups_batch(ctx, p, (size_t)kCreditCardTargetWidth * kCreditCardTargetHeight, 1, &x) == DMZ_HIP_OK;
    for (size_t i = 0; i < m && ok; i++) {
      memset(&groups[base + i].scores, 0, sizeof(ExpiryGroupScores));
      for (int row = 0; row < 4; row++)
        memcpy(groups[base + i].scores.v[row < 2 ? row : row + 1], x.groups[i].scores[row], sizeof(float) * 10);
    }
  }
  if (owned) free(p);
  return ok;
}

void dmz_expiry_extract(IplImage *card_y, uint16_t *number_of_expiry_groups, CythonGroupedRects **cython_expiry_groups,
                        uint16_t *number_of_new_groups, CythonGroupedRects **cython_new_groups, int *expiry_month,
                        int *expiry_year) {
  if (!number_of_expiry_groups || !cython_expiry_groups || !number_of_new_groups || !cython_new_groups || !expiry_month || !expiry_year)
    return;
  GroupedRectsList expiry_groups, new_groups;
  for (int i = 0; i < *number_of_expiry_groups; i++) expiry_groups.push_back(from_cython_group(*cython_expiry_groups + i));
  for (int i = 0; i < *number_of_new_groups; i++) new_groups.push_back(from_cython_group(*cython_new_groups + i));
  // expiry_extract (expiry_categorize.cpp:448-501)
  if (!new_groups.empty()) {
    if (categorize_groups(card_y, new_groups)) {
      expiry_aggregate_grouped_rects(expiry_groups, new_groups);
      for (GroupedRectsList::iterator group = expiry_groups.begin(); group != expiry_groups.end(); ++group) {
        if (group->total_seen_count < 3) continue;
        get_stable_expiry_month_and_year(*group, expiry_month, expiry_year);
      }
    } else {
      fprintf(stderr, "dmz (HIP): dmz_expiry_extract: the new groups could not be categorised (not a 428 x 270 card image, a "
                      "rect outside it, or a group without five rects): the session's groups are left as they were\n");
    }
  }
  // As dmz.cpp:643-655: ONLY the session's array is re-allocated to the new size and rewritten; the new-groups array and
  // its count stay the caller's, untouched, and the character-rect arrays of the replaced entries are not freed (the
  // reference has that free commented out: a caller may still hold those pointers).
  *number_of_expiry_groups = (uint16_t)expiry_groups.size();
  *cython_expiry_groups = (CythonGroupedRects *)realloc(*cython_expiry_groups, sizeof(CythonGroupedRects) * (expiry_groups.size() + 1));
  for (size_t i = 0; i < expiry_groups.size(); i++) (*cython_expiry_groups)[i] = to_cython_group(expiry_groups[i]);
}

void dmz_expiry_extract_group(IplImage *card_y, CythonGroupedRects &cython_group, CythonGroupScores cython_scores,
                              int *expiry_month, int *expiry_year) {
  // expiry_extract_group (expiry_categorize.cpp:504-522): this frame's scores blended into the group's old ones
  GroupedRectsList one(1, from_cython_group(&cython_group));
  const ExpiryGroupScores old_scores = one[0].scores;
  if (!categorize_groups(card_y, one)) return;
  GroupedRects &group = one[0];
  for (int i = 0; i < kExpiryMaxValidLength; i++)
    for (int d = 0; d < 10; d++) group.scores(i, d) = (old_scores(i, d) * kExpiryDecayFactor) + (group.scores(i, d) * (1 - kExpiryDecayFactor));
  get_stable_expiry_month_and_year(group, expiry_month, expiry_year);
  for (int i = 0; i < kExpiryMaxValidLength; i++)
    for (int d = 0; d < 10; d++) cython_scores[i][d] = group.scores(i, d);
}

// mz.h:37-52 (cython_dmz/mz.cpp): an IplImage header over the caller's pixels
IplImage *py_mz_create_from_cv_image_data(char *image_data, int image_size, int width, int height, int64_t depth,
                                          int n_channels, int roi_x_offset, int roi_y_offset, int roi_width, int roi_height) {
  IplImage *im = (IplImage *)calloc(1, sizeof(IplImage));
  if (!im) return NULL;
  im->nSize = (int)sizeof(IplImage);
  im->nChannels = n_channels;
  im->depth = (int)depth;
  im->align = 4;
  im->width = width;
  im->height = height;
  im->widthStep = height > 0 ? image_size / height : 0;
  im->imageSize = image_size;
  im->imageData = im->imageDataOrigin = image_data;
  if (roi_width > 0 && roi_height > 0 && (roi_x_offset || roi_y_offset || roi_width != width || roi_height != height))
    py_mz_cvSetImageROI(im, roi_x_offset, roi_y_offset, roi_width, roi_height);
  return im;
}
void py_mz_release_ipl_image(IplImage *image) {  // the header only: the pixels are the caller's
  if (!image) return;
  free(image->roi);
  free(image);
}
void py_mz_get_cv_image_data(IplImage *source, char **image_data, int *image_size, int *width, int *height, int64_t *depth,
                             int *n_channels, int *roi_x_offset, int *roi_y_offset, int *roi_width, int *roi_height) {
  *image_data = source->imageData;
  *image_size = source->imageSize;
  *width = source->width, *height = source->height;
  *depth = source->depth;
  *n_channels = source->nChannels;
  int x, y, w, h;
  image_extent(source, &x, &y, &w, &h);
  *roi_x_offset = x, *roi_y_offset = y, *roi_width = w, *roi_height = h;
}
void py_mz_cvSetImageROI(IplImage *image, int left, int top, int width, int height) {
  if (!image) return;
  if (!image->roi) image->roi = (IplROI *)calloc(1, sizeof(IplROI));
  image->roi->coi = 0;
  image->roi->xOffset = left, image->roi->yOffset = top, image->roi->width = width, image->roi->height = height;
}
void py_mz_cvResetImageROI(IplImage *image) {
  if (image && image->roi) {
    free(image->roi);
    image->roi = NULL;
  }
}

// ---- flat-array hook for the host-logic tests (tests/test_host_logic.py): replays a session's
// expiry_extract calls (expiry_categorize.cpp:332-376) on caller-supplied per-frame groups ----
extern "C" int dmz_hip_host_expiry_session_replay(int n_frames, const int *groups_per_frame, const int16_t *tops,
                                                  const int16_t *lefts, const float *scores /* [g][4][10] */,
                                                  int *months_out, int *years_out, int *n_aggregated_out) {
  GroupedRectsList aggregated;
  int month = 0, year = 0, g = 0;
  for (int f = 0; f < n_frames; f++) {
    dmz_hip_expiry_result x;
    memset(&x, 0, sizeof(x));
    x.n_groups = groups_per_frame[f] < DMZ_HIP_EXPIRY_MAX_GROUPS ? groups_per_frame[f] : DMZ_HIP_EXPIRY_MAX_GROUPS;
    for (int i = 0; i < groups_per_frame[f]; i++, g++) {
      if (i >= DMZ_HIP_EXPIRY_MAX_GROUPS) continue;
      dmz_hip_expiry_group &d = x.groups[i];
      d.top = tops[g];
      d.left = lefts[g];
      d.width = 5 * 13;
      d.height = kSmallCharacterHeight;
      for (int c = 0; c < 5; c++) d.char_top[c] = tops[g], d.char_left[c] = (int16_t)(lefts[g] + 13 * c);
      memcpy(d.scores, scores + (size_t)g * 40, sizeof(float) * 40);
    }
    GroupedRectsList new_groups;
    fill_expiry_groups(x, &new_groups);
    if (!new_groups.empty()) {
      expiry_aggregate_grouped_rects(aggregated, new_groups);
      for (GroupedRectsList::iterator group = aggregated.begin(); group != aggregated.end(); ++group) {
        if (group->total_seen_count < 3) continue;
        get_stable_expiry_month_and_year(*group, &month, &year);
      }
    }
    months_out[f] = month;
    years_out[f] = year;
    n_aggregated_out[f] = (int)aggregated.size();
  }
  return 0;
}
