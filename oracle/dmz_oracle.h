/*
 * dmz_oracle.h -- CPU ORACLE for the card.io-dmz per-frame scan hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  It is a plain-C restatement of
 * the reference algorithm (each function cites the /root/reference file:line it
 * follows).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it, and only as the checker / reported baseline.  The shipped
 * path (card.io-dmz_amd/csrc) never links or calls anything in oracle/.
 *
 * Parity pinning (see DESIGN.md "Oracle"):
 *   pinned   : the six model forward passes against the reference's embedded
 *              known-answer vectors (tests/golden/model_kats.npz, 1e-5), and --
 *              when oracle/_ref was built in the build container -- geometry,
 *              homography (Eigen HouseholderQR), vseg box-sum, hseg search and
 *              Luhn against the reference's own compiled code (bit-exact).
 *   UNPINNED : every stage whose arithmetic lives in un-vendored OpenCV 2.4
 *              (cvSobel, cvWarpPerspective, cvMorphologyEx, cvResize,
 *              cvNormalize, cvConvertScale, cvReduce) or needs an OpenCV
 *              library symbol to run (Canny, Hough, equalize-hist: cvGetMat /
 *              cvGetSize).  Those are restated from the in-tree source and the
 *              published OpenCV 2.4 semantics; no
 *              reference fixture exists for them => "parity unpinned".
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off; no FMA, IEEE semantics)
 */
#ifndef DMZ_ORACLE_H
#define DMZ_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_CARD_W 428
#define ORC_CARD_H 270
#define ORC_NUM_W 19
#define ORC_NUM_H 27

/* ---- weights blob (card.io-dmz_amd/weights/dmz_models.bin, after the 16-byte
 * header): float offsets, same order as tools/extract_models.py writes ------- */
enum {
  ORC_W_VSEG_W1 = 0,                         /* 50x204 */
  ORC_W_VSEG_B1 = ORC_W_VSEG_W1 + 50 * 204,  /* 50 */
  ORC_W_VSEG_W2 = ORC_W_VSEG_B1 + 50,        /* 3x50 */
  ORC_W_VSEG_B2 = ORC_W_VSEG_W2 + 150,       /* 3 */
  ORC_W_DIGIT0 = ORC_W_VSEG_B2 + 3,          /* 3 models x 10682 */
  ORC_DIGIT_CONV_W = 0,                      /* 8x3x3 */
  ORC_DIGIT_CONV_B = 72,                     /* 8 */
  ORC_DIGIT_HID_W = 80,                      /* 32x320 */
  ORC_DIGIT_HID_B = 80 + 10240,              /* 32 */
  ORC_DIGIT_LOG_W = 80 + 10240 + 32,         /* 10x32 */
  ORC_DIGIT_LOG_B = 80 + 10240 + 32 + 320,   /* 10 */
  ORC_DIGIT_STRIDE = 80 + 10240 + 32 + 320 + 10, /* 10682 */
  ORC_W_SLASH = ORC_W_DIGIT0 + 3 * ORC_DIGIT_STRIDE, /* 80x176,80,2x80,2 */
  ORC_W_EXPIRY = ORC_W_SLASH + 80 * 176 + 80 + 160 + 2,
  ORC_W_TOTAL = ORC_W_EXPIRY + 1250 + 50 + 50000 + 40 + 21120 + 176 + 1760 + 10
};

/* Install the weight blob (pointer must stay valid). */
void orc_set_weights(const float *blob);

/* ---- per-frame result record: field-for-field the layout of
 * dmz_hip_frame_result (include/dmz_hip.h); 1024 bytes ----------------------- */
typedef struct {
  int32_t found[4];    /* top, left, bottom, right  (dmz.h:33-38 order) */
  float rho[4];
  float theta[4];
  float corners[8];    /* top_left, bottom_left, top_right, bottom_right (x,y) */
  int32_t found_all;   /* return value of dmz_detect_edges */
  int32_t flags;       /* ORC_FLAG_* */
  float vseg_score;
  int32_t vseg_y_offset;
  int32_t pattern_type; /* 0 unknown, 1 visa-like, 2 amex-like */
  int32_t n_offsets;
  uint16_t offsets[16];
  float hseg_score;
  float number_width;
  int32_t pattern_offset;
  float number_score;  /* n_offsets - sum(scores), frame.cpp:63 */
  uint8_t digits[16];  /* argmax of each scores row, first max wins */
  float scores[16][10];
  int32_t expiry_month;
  int32_t expiry_year;
  uint8_t reserved[1024 - 816];
} orc_frame_result;

#define ORC_FLAG_USABLE 1       /* FrameScanResult.usable (frame.cpp:43,64) */
#define ORC_FLAG_UPSIDE_DOWN 2  /* frame.cpp:38-41 */
#define ORC_FLAG_VSEG_OK 4      /* vseg.score > 15 and not upside down */
#define ORC_FLAG_WARPED 8       /* card was rectified (all four edges found) */

/* ---- cv/ ------------------------------------------------------------------- */
/* dmz.cpp:279-341; boxes[4][4] = top,bottom,left,right x (x,y,w,h) */
void orc_detection_boxes(int width, int height, int orientation, int boxes[4][4]);
/* sobel.cpp:476-478 = cvSobel(src,dst,dx,dy,7) on an isolated ROI */
void orc_sobel7(const uint8_t *src, int stride, int w, int h, int want_dx, int16_t *dst);
/* canny.cpp:568-580 + 58-336; out 0/255, also returns low/high */
void orc_adaptive_canny7(const int16_t *dx, const int16_t *dy, int w, int h,
                         uint8_t *out, int *low_out, int *high_out);
/* hough.cpp:52-195 with dmz.cpp:246-249 parameters; returns 1 if a line was found */
int orc_hough(const uint8_t *edges, const int16_t *dx, const int16_t *dy, int w, int h,
              int vertical, float *rho, float *theta, int *n_out, int *r_out, int *max_out);
/* dmz.cpp:224-271 on ROI (x,y,w,h) of an 8U plane; returns found, line in ROI coords */
int orc_best_line(const uint8_t *plane, int stride, int x, int y, int w, int h,
                  int vertical, float *rho, float *theta);
/* geometry.cpp:34-43 */
void orc_line_by_shifting_origin(float rho, float theta, int xoff, int yoff,
                                 float *rho_out, float *theta_out);
/* geometry.cpp:14-32; returns 1 if intersect */
int orc_parametric_intersect(float rho1, float theta1, float rho2, float theta2,
                             float *x, float *y);
/* dmz.cpp:371-439.  cb/cr may be NULL (then only the Y plane is searched). */
int orc_detect_edges(const uint8_t *y, int y_stride, int w, int h,
                     const uint8_t *cb, const uint8_t *cr, int c_stride,
                     int orientation, orc_frame_result *res);
/* warp.cpp:34-125 (row-major 3x3) */
void orc_calc_persp_transform(const float src_pts[8], const float dst_pts[8], float m[9]);
/* ... in the summation order of a stock x86-64 (SSE2 packet) build of the reference's Eigen (orc_cv.c) */
void orc_calc_persp_transform_sse(const float src_pts[8], const float dst_pts[8], float m[9]);
/* 0 (default): orc_transform_card / orc_scan_frame restate the scalar-Eigen build; 1: the stock x86-64 (SSE2) one */
void orc_set_reference_flavour(int flavour);
/* cvWarpPerspective(INTER_LINEAR|FILL_OUTLIERS, 0), SURVEY Appendix A10 */
void orc_warp_perspective(const uint8_t *src, int stride, int sw, int sh,
                          const float m[9], uint8_t *dst, int dstride, int dw, int dh);
/* dmz.cpp:443-497 for a 1-channel plane.  options: bit 0 = cast the corner points to int
 * (cython_dmz/dmz.pyx:267-270), bit 1 = `upsample` (half-size chroma plane: points / 2, dmz.cpp:473-481) */
void orc_transform_card(const uint8_t *plane, int stride, int w, int h,
                        const float corners[8], int orientation, int options,
                        uint8_t *card /* 428x270, stride 428 */);

/* ---- scan/ ----------------------------------------------------------------- */
void orc_morph_grad3_1d(const uint8_t *src, int n, uint8_t *dst);           /* morph.cpp:108-112 */
void orc_morph_grad3_2d_cross(const uint8_t *src, int stride, int w, int h,
                              uint8_t *dst, int dstride);                   /* morph.cpp:190-220 */
void orc_lineardown2_1d(const uint8_t *src, int n_out, uint8_t *dst);       /* convert.cpp:195-197 */
void orc_norm_convert_1d(const uint8_t *src, int n, float *dst);            /* convert.cpp:380-383 */
void orc_equalize_hist(uint8_t *img, int stride, int w, int h);             /* stats.cpp:116-159 */
void orc_applym_vseg(const float x[204], float out[3]);                     /* modelm_befe75da.cpp:1770-1786 */
void orc_applyc_digit(int model, const float x[27 * 19], float out[10]);    /* modelc_*.cpp:1893-1937 */
void orc_applym_slash(const float x[176], float out[2]);                    /* modelm_730c4cbd.cpp:2431-2449 */
void orc_applyc_expiry(const float x[16 * 11], float out[10],
                       float *l1 /*50*70 or NULL*/, float *l2 /*120*/, float *l3 /*176*/); /* modelc_bf4dd6c8.cpp:13457-13505 */
void orc_vseg_row_features(const uint8_t *row408, float feat[204]);         /* n_vseg.cpp:39-43 */
void orc_best_segmentation_for_vseg_scores(const float *visa, const float *amex, float *score,
                                           int *y_offset, int *pattern);    /* n_vseg.cpp:49-92 */
void orc_best_n_vseg(const uint8_t *card, int stride, float *score, int *y_offset, int *pattern,
                     float *visa_scores /*270 or NULL*/, float *amex_scores); /* n_vseg.cpp:94-168 */
void orc_hseg_grad_sums(const uint8_t *strip, int stride, float sums[428]); /* n_hseg.cpp:90-96 */
void orc_best_n_hseg_constrained(const float *grad_sums, int pattern_type,
                                 float wmin, float wmax, float wstep,
                                 int omin, int omax, int ostep,
                                 uint16_t offsets[16], float *score, float *number_width,
                                 int *pattern_offset);
                      /* n_hseg.cpp:39-84 */
void orc_best_n_hseg(const uint8_t *strip, int stride, int pattern_type, orc_frame_result *res); /* n_hseg.cpp:88-151 */
void orc_number_scores(const uint8_t *strip, int stride, const uint16_t *offsets, int n,
                       float scores[160]);                                  /* n_categorize.cpp:75-107 */
void orc_scan_card_image(const uint8_t *card, int stride, orc_frame_result *res); /* frame.cpp:24-81 (number path) */
void orc_scan_card_image_ex(const uint8_t *card, int stride, int collect_card_number, orc_frame_result *res);
/* test infrastructure: the stages after vseg at a GIVEN (y_offset, pattern, score) -- see orc_scan.c */
void orc_scan_card_image_at(const uint8_t *card, int stride, int y_off, int pattern, float score,
                            int collect_card_number, orc_frame_result *res);

/* ---- expiry path: scan/expiry_seg.cpp, scan/expiry_categorize.cpp.
 * Per-frame part only: segmentation into MM/YY groups and the four digit score rows of each
 * group; the cross-frame aggregation (expiry_categorize.cpp:251-330) is session logic.
 * Layout == dmz_hip_expiry_group / dmz_hip_expiry_result (include/dmz_hip.h). ------------- */
#define ORC_EXPIRY_MAX_GROUPS 8
typedef struct {
  int16_t top, left, width, height;  /* GroupedRects of the 5 characters (expiry_seg.cpp:651-668) */
  int16_t char_top[5], char_left[5]; /* the 11x16 trimmed character rects */
  int16_t stripe_base_row;
  int16_t reserved;
  float scores[4][10];               /* characters 0,1,3,4 (M M / Y Y); zero when not categorised */
} orc_expiry_group;                  /* 192 bytes */
typedef struct {
  int32_t n_groups;                  /* min(n_found, ORC_EXPIRY_MAX_GROUPS) */
  int32_t n_found;                   /* groups the segmentation produced */
  int32_t n_stripes;
  int32_t stripe_base_row[3];
  int64_t stripe_sum[3];
  int32_t categorised;               /* 1 when scores were filled (frame usable) */
  int32_t reserved;
  orc_expiry_group groups[ORC_EXPIRY_MAX_GROUPS];
} orc_expiry_result;                 /* 56 + 8*192 = 1592 bytes */

/* sobel.cpp:706-804 (x86 scalar branch) on rows [y0, y0+h) of an 8U image, dst int16 same geometry */
void orc_scharr3_dx_abs(const uint8_t *src, int stride, int w, int h, int16_t *dst, int dstride);
/* expiry_seg.cpp:707-902 + 437-704: fills everything but the scores */
void orc_best_expiry_seg(const uint8_t *card, int stride, int starting_y_offset, orc_expiry_result *out);
/* expiry_categorize.cpp:35-70: 11x16 ROI -> gradient, equalise, bilateral, /255 */
void orc_prepare_image_for_cat(const uint8_t *card, int stride, int left, int top, float x[176]);
/* expiry_categorize.cpp:138-160 for every group of `out` (fills scores, sets categorised) */
void orc_categorize_expiry_groups(const uint8_t *card, int stride, orc_expiry_result *out);
/* frame.cpp:71-73 gate + scan.cpp:62-64 gate, given the number-path result of the same card */
void orc_scan_card_expiry(const uint8_t *card, int stride, const orc_frame_result *res,
                          orc_expiry_result *out);
/* libstdc++'s std::sort with a "key descending" comparator, as a permutation: order[k] = index of the k-th
 * element after the sort (expiry_seg.cpp:456, 842 with the comparators at 75-87) */
void orc_sort_order_desc(int n, const long *key, int *order);
/* test hook: capture the key lists orc_best_expiry_seg sorts, as [n, keys...] records */
void orc_expiry_capture_sort_lists(int64_t *buf, int len);
int orc_expiry_captured_len(void);
int orc_sort_heap_sorts(void); /* test hook: depth-limit heap sorts so far on this thread */
/* flat-array hooks onto gather_into_groups (expiry_seg.cpp:131-167) and regrid_group (169-229) */
int orc_expiry_gather_into_groups(int n_items, const int *lefts, const int64_t *sums, int top, int height,
                                  int *group_n, int *group_left, int *group_width, int *rect_left,
                                  int64_t *rect_sum);
void orc_expiry_regrid_group(const int16_t *sobel, int top, int height, int *left, int *width,
                             int *character_width, int *n, int *rect_left, int64_t *rect_sum);

/* ---- camera-side plumbing ---- */
void orc_split_u8(const uint8_t *interleaved, int stride, int w, int h, uint8_t *c1, uint8_t *c2); /* convert.cpp:105-107 */
void orc_deinterleave_rgba_to_r(const uint8_t *source, uint8_t *dest, int size);                  /* dmz.cpp:62-105 */
void orc_ycbcr_to_rgb(const uint8_t *y, const uint8_t *cb, const uint8_t *cr, int w, int h, int channels,
                      uint8_t *rgb);                                                               /* convert.cpp:448-490 */

/* ---- quality scores ---- */
void orc_card_rect_for_screen(int card_w, int card_h, int std_w, int std_h, int act_w, int act_h, int rect[4]); /* dmz.cpp:138-165 */
void orc_scoring_roi(int img_w, int img_h, int use_full_image, int rect[4]);                                     /* dmz.cpp:167-185 */
float orc_focus_score(const uint8_t *img, int stride, int w, int h, int use_full_image);       /* dmz.cpp:114-126,187-192 */
float orc_brightness_score(const uint8_t *img, int stride, int w, int h, int use_full_image);  /* dmz.cpp:128-135,194-199 */

/* dmz.cpp:499-515: median-blur (25 x 25, per channel) the boxes of the first n_offsets - unblur_digits digits, in place */
void orc_blur_card(uint8_t *rgb, int width, int height, int channels, const uint16_t *offsets, int n_offsets,
                   float number_width, int y_offset, int unblur_digits);

/* ---- per-session policy (scan/scan.cpp:41-194 + expiry_categorize.cpp:162-376) replayed over the
 * per-frame records of one session; layout == dmz_hip_session_result (include/dmz_hip.h) ---- */
typedef struct {
  int32_t complete;        /* ScannerResult.complete when the replay stopped */
  int32_t complete_frame;  /* frame after which scanner_result first reported complete, -1 if never */
  int32_t number_frame;    /* frame after which the card number was accepted, -1 if never */
  int32_t n_numbers;
  uint8_t predictions[16];
  int32_t card_type;       /* dmz_olm.h CardType of the accepted number */
  int32_t expiry_month, expiry_year;
  int32_t count15, count16;
  int32_t usable_frames;
  int32_t n_expiry_groups; /* aggregated expiry groups alive when the replay stopped */
  int32_t vseg_y_offset, n_offsets;
  uint16_t offsets[16];
  float number_width;      /* NHorizontalSegmentation.number_width of the accepted number */
  int32_t reserved[6];
} orc_session_result;      /* 128 bytes */
void orc_scan_session(const orc_frame_result *frames, const orc_expiry_result *expiry /* or NULL */,
                      int n_frames, int scan_expiry, int frame_interval_ms, int now_year, int now_month,
                      int allow_past_expiry, orc_session_result *out);
int orc_card_type(const uint8_t *digits, int n, int allow_incomplete, int *number_length); /* dmz_olm.cpp:51-130 */

/* ---- full per-frame pipeline: detect -> transform(Y) -> scan ---------------- */
void orc_scan_frame(const uint8_t *y, int stride, int w, int h, int orientation,
                    int truncate_corners, uint8_t *card_out /* 428*270 or NULL */,
                    orc_frame_result *res);

/* dmz_olm.cpp:40-49 */
int orc_passes_luhn(const uint8_t *digits, int n);

/* ---- synthetic frames (bench / test inputs; integer + IEEE-exact math so the
 * HIP generator in card.io-dmz_amd/csrc/synth.hip produces identical bytes) -- */
void orc_synth_frame(uint64_t seed, uint64_t frame_index, uint8_t *y /* 640x480 */, uint8_t digits_out[16]);
void orc_synth_card(uint64_t seed, uint64_t frame_index, uint8_t *card /* 428x270 */, uint8_t digits_out[16]);

#ifdef __cplusplus
}
#endif
#endif
