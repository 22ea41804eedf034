/*
 * dmz_hip.h -- C-ABI of the MI355X (gfx950) implementation of the card.io-dmz
 * per-frame scan hot path.  Plain pointers and sizes only; no C++/torch types.
 *
 * This is the drop-in boundary: each entry point is the batched form of one of
 * the reference's per-frame entry points, and the reference-side binding a
 * maintainer would add is shown in INTEGRATION.md.  The single-frame C++
 * mirrors of the reference API (dmz_detect_edges, dmz_transform_card,
 * scanner_add_frame_with_expiry, ... in card.io-dmz_amd/host/dmz.h) are
 * batch-of-1 wrappers over these functions, with `dmz_context.mz`
 * (reference dmz.h:17-20, mz.h:19-25) holding the dmz_hip_context exactly where
 * the Android flavour keeps its GLES warp context (mz_android.cpp:233-240).
 *
 * Pointer convention: every image/result pointer may be a DEVICE pointer
 * (HBM-resident batches: the fast path, nothing is copied) or a HOST pointer
 * (staged through an internal device buffer; PCIe-bound).  The kind is detected
 * with hipPointerGetAttributes.
 *
 * Error convention: the reference has no error codes (bool/found flags, asserts,
 * silent CPU fallback on accelerator failure: mz_android.cpp:8-24).  Here every
 * function returns DMZ_HIP_OK (0) or a negative DMZ_HIP_E* code and
 * dmz_hip_last_error() gives the text.  There is NO CPU fallback: a missing GPU
 * or kernel image is an error.
 */
#ifndef DMZ_HIP_H
#define DMZ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMZ_HIP_OK 0
#define DMZ_HIP_EINVAL (-1)     /* bad argument (assert in the reference: dmz.cpp:373-377, frame.cpp:25-29) */
#define DMZ_HIP_ENODEVICE (-2)  /* no usable gfx950 device / HIP runtime error at create */
#define DMZ_HIP_ERUNTIME (-3)   /* HIP error during a batch call */
#define DMZ_HIP_EUNSUPPORTED (-4) /* geometry outside what the LDS-resident kernels hold */

/* FrameOrientation, reference dmz_olm.h:17-23 */
#define DMZ_ORIENTATION_PORTRAIT 1
#define DMZ_ORIENTATION_PORTRAIT_UPSIDE_DOWN 2
#define DMZ_ORIENTATION_LANDSCAPE_RIGHT 3
#define DMZ_ORIENTATION_LANDSCAPE_LEFT 4

#define DMZ_CARD_WIDTH 428  /* kCreditCardTargetWidth,  dmz_constants.h:8 */
#define DMZ_CARD_HEIGHT 270 /* kCreditCardTargetHeight, dmz_constants.h:9 */

/* Fixed-size POD result record, one per frame (1024 bytes).  It flattens what
 * the reference returns per frame: dmz_edges + dmz_corner_points (dmz.h:22-37,
 * dmz_olm.h:37-42) and the number-path fields of FrameScanResult (frame.h:14-28:
 * NVerticalSegmentation n_vseg.h:14-21, NHorizontalSegmentation n_hseg.h:13-19,
 * NumberScores n_categorize.h:14). */
typedef struct dmz_hip_frame_result {
  int32_t found[4];     /* dmz_edges order: top, left, bottom, right */
  float rho[4];         /* ParametricLine.rho   (frame coordinates) */
  float theta[4];       /* ParametricLine.theta */
  float corners[8];     /* top_left, bottom_left, top_right, bottom_right (x,y) */
  int32_t found_all;    /* return value of dmz_detect_edges (dmz.cpp:418-438) */
  int32_t flags;        /* DMZ_HIP_FLAG_* */
  float vseg_score;     /* NVerticalSegmentation.score */
  int32_t vseg_y_offset;
  int32_t pattern_type; /* 0 unknown, 1 visa-like (16), 2 amex-like (15); n_vseg.cpp:20-24 */
  int32_t n_offsets;    /* NHorizontalSegmentation.n_offsets */
  uint16_t offsets[16];
  float hseg_score;
  float number_width;
  int32_t pattern_offset;
  float number_score;   /* n_offsets - scores.sum()  (frame.cpp:63) */
  uint8_t digits[16];   /* argmax of each scores row (first maximum) */
  float scores[16][10]; /* NumberScores, row-major */
  int32_t expiry_month; /* 0 = not scanned */
  int32_t expiry_year;
  uint8_t reserved[1024 - 816];
} dmz_hip_frame_result;

#define DMZ_HIP_FLAG_USABLE 1      /* FrameScanResult.usable */
#define DMZ_HIP_FLAG_UPSIDE_DOWN 2 /* FrameScanResult.upside_down */
#define DMZ_HIP_FLAG_VSEG_OK 4     /* passed the vseg gates (frame.cpp:38-47) */
#define DMZ_HIP_FLAG_WARPED 8      /* card image was rectified */
#define DMZ_HIP_FLAG_FAULT 16      /* a device self-check did not settle for this frame (the homography evaluated six times without
                                      two consecutive results alike, geometry.hip): the frame is NOT rectified (no
                                      DMZ_HIP_FLAG_WARPED, zero card, scan stages skip it).  Never seen in a sweep; a caller
                                      re-submits the frame */

/* ---- expiry path: scan/expiry_seg.h best_expiry_seg + the per-frame half of
 * scan/expiry_categorize.cpp (categorize_expiry_digits, :138-160).  One record per frame; the
 * groups are GroupedRects of pattern ExpiryPatternMMsYY (expiry_types.h:37-45,65-79) in the
 * order the reference appends them to FrameScanResult.expiry_groups (frame.h:19). --------- */
#define DMZ_HIP_EXPIRY_MAX_GROUPS 8
typedef struct dmz_hip_expiry_group {
  int16_t top, left, width, height;  /* GroupedRects.top/left/width/height (expiry_seg.cpp:651-668) */
  int16_t char_top[5], char_left[5]; /* CharacterRect.top/left of the 5 characters (11 x 16 px each) */
  int16_t stripe_base_row;           /* the stripe the group was found in */
  int16_t reserved;
  float scores[4][10];               /* ExpiryGroupScores rows 0,1,3,4 (M M / Y Y); 0 unless categorised */
} dmz_hip_expiry_group;              /* 192 bytes */
typedef struct dmz_hip_expiry_result {
  int32_t n_groups;                  /* min(n_found, DMZ_HIP_EXPIRY_MAX_GROUPS) */
  int32_t n_found;                   /* groups best_expiry_seg produced for this frame */
  int32_t n_stripes;                 /* probable stripes tried (<= 3), expiry_seg.cpp:838-858 */
  int32_t stripe_base_row[3];
  int64_t stripe_sum[3];
  int32_t categorised;               /* 1: frame was usable, scores are filled (scan.cpp:57-64) */
  int32_t reserved;
  dmz_hip_expiry_group groups[DMZ_HIP_EXPIRY_MAX_GROUPS];
} dmz_hip_expiry_result;             /* 1592 bytes */

/* dmz_hip_transform_batch / pipeline option bits */
#define DMZ_HIP_OPT_TRUNCATE_CORNERS 1 /* cast corner points to int like cython_dmz/dmz.pyx:267-270 */
#define DMZ_HIP_OPT_UPSAMPLE 2         /* dmz_transform_card(upsample = true), dmz.cpp:473-481: the plane is a
                                          half-size Cb/Cr plane, the corner points are halved */
#define DMZ_HIP_OPT_EIGEN_SSE2 4       /* llcv_calc_persp_transform's float Householder QR (warp.cpp:34-125) in the summation
                                          order of a STOCK x86-64 build of the reference (Eigen 3.2.4's SSE2 packet paths; eigen.h
                                          defines no EIGEN_DONT_VECTORIZE).  Default (bit clear): Eigen's scalar order, what the
                                          reference's non-NEON ARM builds and -DEIGEN_DONT_VECTORIZE compute.  The two differ in
                                          the last bits of the homography on three frames of four, hence in ~15 bytes of such a
                                          card.  HOMOGRAPHY ONLY: hseg_score's last bits and the bilateral
                                          filter stay in the scalar order.  Also a context default:
                                          dmz_hip_set_reference_flavour. */
#define DMZ_HIP_OPT_EIGEN_SCALAR 8     /* this call in Eigen's scalar order even when the context's default flavour is SSE2
                                          (the per-call override of dmz_hip_set_reference_flavour; wins over DMZ_HIP_OPT_EIGEN_SSE2) */

typedef struct dmz_hip_context dmz_hip_context;

/* Number of HIP devices (does not initialise a device). */
int dmz_hip_device_count(void);

/* Replaces mz_create()/dmz_context_create() (mz.h:19, dmz.cpp:23-27): binds a
 * device, uploads the model weights and the Hough/geometry tables. */
int dmz_hip_context_create(int device_ordinal, dmz_hip_context **out);
/* Replaces mz_destroy()/dmz_context_destroy() (mz.h:22, dmz.cpp:29-32). */
void dmz_hip_context_destroy(dmz_hip_context *ctx);
/* Replaces mz_prepare_for_backgrounding() (mz.h:25): drains the stream. */
int dmz_hip_synchronize(dmz_hip_context *ctx);
/* Use an externally owned hipStream_t (e.g. torch's current stream); NULL = the context's own. */
int dmz_hip_set_stream(dmz_hip_context *ctx, void *hip_stream);
const char *dmz_hip_last_error(const dmz_hip_context *ctx);

/* Batched dmz_detect_edges (dmz.h:82-83, dmz.cpp:371-439) on n luma planes.
 * y: n planes of height x width bytes, plane i at y + i*frame_stride, rows
 * row_stride bytes apart.  cb/cr (half-size planes, chroma fallback of
 * dmz.cpp:346-369) may be NULL.  Resets each record to zero, then writes found/rho/theta/corners/found_all. */
int dmz_hip_detect_batch(dmz_hip_context *ctx, const uint8_t *y, size_t frame_stride,
                         int row_stride, int width, int height,
                         const uint8_t *cb, const uint8_t *cr, size_t chroma_frame_stride,
                         int chroma_row_stride, int n, int orientation,
                         dmz_hip_frame_result *results);

/* Batched dmz_transform_card (dmz.h:92-96, dmz.cpp:443-497; llcv_unwarp
 * warp.cpp:130-169) of a 1-channel plane: for every frame with
 * results[i].found_all, rectifies to cards + i*card_stride (428x270, row stride
 * 428) and sets DMZ_HIP_FLAG_WARPED; other cards are zero-filled. */
int dmz_hip_transform_batch(dmz_hip_context *ctx, const uint8_t *plane, size_t frame_stride,
                            int row_stride, int width, int height, int n, int orientation,
                            int options, dmz_hip_frame_result *results, uint8_t *cards,
                            size_t card_stride);

/* Batched scan_card_image number path (frame.h:34, frame.cpp:24-81 as driven by
 * scanner_add_frame_with_expiry scan.cpp:41-50) on n 428x270 cards.  `mode` is a bit mask:
 * DMZ_HIP_SCAN_ONLY_WARPED: skip cards whose result lacks DMZ_HIP_FLAG_WARPED (their scan fields and
 *   gate flags are cleared);
 * DMZ_HIP_SCAN_SKIP_NUMBER: scan_card_image(collect_card_number = false), what the reference runs once a
 *   session's number has been accepted (scan.cpp:43-48): only the vseg search and its gates run, and
 *   a frame that passes them is usable (frame.cpp:43-49); hseg / digit fields stay zero. */
#define DMZ_HIP_SCAN_ONLY_WARPED 1
#define DMZ_HIP_SCAN_SKIP_NUMBER 2
int dmz_hip_scan_cards_batch(dmz_hip_context *ctx, const uint8_t *cards, size_t card_stride,
                             int n, int mode, dmz_hip_frame_result *results);

/* best_n_hseg (scan/n_hseg.h, n_hseg.cpp:88-151) alone: the digit x-offset search on the 27-row strip at
 * results[i].vseg_y_offset of card i for results[i].pattern_type (1 = sixteen digits, 2 = fifteen), for every record
 * that carries DMZ_HIP_FLAG_VSEG_OK and 0 <= vseg_y_offset <= 243; fills n_offsets, offsets, hseg_score, number_width and
 * pattern_offset and leaves the rest of the record alone (the stage entry dmz_hip_scan_cards_batch runs between its
 * vseg and digit stages). */
int dmz_hip_best_n_hseg_batch(dmz_hip_context *ctx, const uint8_t *cards, size_t card_stride, int n,
                              dmz_hip_frame_result *results);

/* detect -> transform(Y) -> scan for n frames (the cython_dmz/dmz.pyx:379-483
 * call sequence).  cards may be NULL (an internal buffer is used). */
int dmz_hip_pipeline_batch(dmz_hip_context *ctx, const uint8_t *y, size_t frame_stride,
                           int row_stride, int width, int height, int n, int orientation,
                           int options, uint8_t *cards, size_t card_stride,
                           dmz_hip_frame_result *results);

/* Batched expiry scan of n 428x270 cards whose number path has already run (results[i] holds
 * flags and vseg_y_offset): best_expiry_seg (expiry_seg.h:14, expiry_seg.cpp:707-902) for every
 * frame that passed the vseg gates with y_offset < 240 (frame.cpp:71-73), then
 * categorize_expiry_digits (expiry_categorize.cpp:138-160) for the usable ones (scan.cpp:57-64).
 * cards and card_stride must be 4-byte aligned. */
int dmz_hip_scan_expiry_batch(dmz_hip_context *ctx, const uint8_t *cards, size_t card_stride, int n,
                              const dmz_hip_frame_result *results, dmz_hip_expiry_result *expiry);

/* categorize_expiry_digits (scan/expiry_categorize.cpp:138-160) on CALLER-SUPPLIED groups: for every frame the records'
 * n_groups and the five character rectangles of each group (char_top / char_left, 11 x 16 px, inside the card) are read,
 * the scores[4][10] of each group are written (characters 0, 1, 3, 4); nothing else of the record changes.  This is
 * the half of expiry_extract (expiry_categorize.cpp:448-501; Cython flavour dmz_expiry_extract*, dmz.h:110-119) that needs the
 * device when the groups come from the caller instead of from dmz_hip_scan_expiry_batch.  cards: device or host;
 * expiry: HOST records (in / out). */
int dmz_hip_categorize_expiry_groups_batch(dmz_hip_context *ctx, const uint8_t *cards, size_t card_stride, int n,
                                           dmz_hip_expiry_result *expiry);

/* llcv_scharr3_dx_abs (cv/sobel.cpp:706-804; Cython flavour dmz_scharr3_dx_abs, dmz.h:105): |right - left| with clamped
 * columns, then 3 / 10 / 3 down the column with clamped rows, 8U -> 16S.  Device or host pointers; strides in elements
 * of their type. */
int dmz_hip_scharr3_dx_abs(dmz_hip_context *ctx, const uint8_t *src, int src_stride, int width, int height, int16_t *dst,
                           int dst_stride);

/* Arithmetic of the expiry CNN's convolutions (applyc_bf4dd6c8, models/expiry/modelc_bf4dd6c8.cpp:
 * 12688-12724: conv2 is 72 x 1250 x 40 per group, the largest contraction of the path), BASELINE configs[3]
 * "bf16 conv with fp32 parity check":
 *   F16X3   (default) v_mfma_f32_16x16x32_f16 on operands split into an f16 rounding and the f16 rounding of the
 *           remainder (22 bits), a.b ~ al.bh + ah.bl + ah.bh with fp32 accumulation, both convolutions: the
 *           scores agree with the fp32 variant and the CPU oracle to ~2e-6, inside the reference's own
 *           known-answer tolerance 1e-5 (tests/test_gpu_expiry.py, bench.py).  Range: the layer-1 activations are
 *           bounded by 16.4 max|x| for this model (sum of |conv1 weights| + |bias| per map), far inside f16 for the
 *           pipeline's inputs (|x| < 1); dmz_hip_apply_expiry_model checks HOST inputs and runs a call whose |x| exceeds
 *           2048 (or is not finite) in the F32 variant; device inputs beyond ~4000 need F32 / BF16X3 set by the caller;
 *   F32     v_mfma_f32_16x16x4_f32 / packed FMAs, the reference's k-ordered fp32 accumulation;
 *   BF16X3  the same three products on bf16 parts (16 bits): ~2^-16 per product, scores within the 1e-4 contract
 *           (the default up to round 2);
 *   BF16    one bf16 pass (~2^-8 per product): reported beside the others, not a parity mode.
 * Everything else on the expiry path is unaffected. */
#define DMZ_HIP_EXPIRY_CONV_F32 0
#define DMZ_HIP_EXPIRY_CONV_BF16X3 1
#define DMZ_HIP_EXPIRY_CONV_BF16 2
#define DMZ_HIP_EXPIRY_CONV_F16X3 3
int dmz_hip_set_expiry_conv(dmz_hip_context *ctx, int mode);

/* dmz_hip_pipeline_expiry_batch schedules the expiry segmentation (which depends on the number row only) on a
 * second device queue beside the digit segmentation / categorisation, and joins the two before the expiry CNN.  Results
 * are identical either way; enable = 0 keeps everything on the context's stream (the default is 1; profiling
 * (dmz_hip_set_profiling) always runs on one queue so that the per-stage times add up). */
int dmz_hip_set_two_queues(dmz_hip_context *ctx, int enable);

/* dmz_hip_pipeline_batch followed by dmz_hip_scan_expiry_batch (scanner_add_frame_with_expiry
 * with scan_expiry = true, scan.cpp:41-86 / BASELINE config 4). */
int dmz_hip_pipeline_expiry_batch(dmz_hip_context *ctx, const uint8_t *y, size_t frame_stride,
                                  int row_stride, int width, int height, int n, int orientation,
                                  int options, uint8_t *cards, size_t card_stride,
                                  dmz_hip_frame_result *results, dmz_hip_expiry_result *expiry);

/* ---- camera-side plumbing; all buffers tightly packed, device or host ----
 * dmz_deinterleave_uint8_c2 (dmz.h:64, dmz.cpp:49-56): n_pairs interleaved 2-channel pixels ->
 * two planes (channel1 = first byte of each pair, as cvSplit). */
int dmz_hip_deinterleave_c2(dmz_hip_context *ctx, const uint8_t *interleaved, size_t n_pairs,
                            uint8_t *channel1, uint8_t *channel2);
/* dmz_deinterleave_RGBA_to_R (dmz.h:67, dmz.cpp:62-105): dest[i] = source[4 i]; size % 4 == 0. */
int dmz_hip_deinterleave_rgba_to_r(dmz_hip_context *ctx, const uint8_t *source, uint8_t *dest, size_t size);
/* dmz_YCbCr_to_RGB (dmz.h:72, dmz.cpp:58-60, cv/convert.cpp:448-490) on n_pixels pixels of three
 * equally sized planes (e.g. a batch of rectified Y / Cb / Cr cards); channels = 3 (RGB) or 4 (RGBA). */
int dmz_hip_ycbcr_to_rgb(dmz_hip_context *ctx, const uint8_t *y, const uint8_t *cb, const uint8_t *cr,
                         size_t n_pixels, int channels, uint8_t *rgb);

/* Quality scores: dmz_focus_score / dmz_brightness_score (dmz.h:77-79,
 * dmz.cpp:114-199) of n luma planes: the scoring ROI is the centre ninth of the guide frame, or the
 * whole guide frame when use_full_image != 0 (dmz.cpp:167-185).  focus / brightness: n floats each,
 * either may be NULL. */
int dmz_hip_scores_batch(dmz_hip_context *ctx, const uint8_t *y, size_t frame_stride, int row_stride,
                         int width, int height, int n, int use_full_image, float *focus, float *brightness);

/* ---- per-session policy, batched.  One record per session: what
 * scanner_result (scan/scan.h:67, scan.cpp:88-194) reports after the last frame fed, plus where in
 * the session it happened. ------------------------------------------------------------------- */
typedef struct dmz_hip_session_result {
  int32_t complete;        /* ScannerResult.complete when the replay stopped */
  int32_t complete_frame;  /* frame after which scanner_result first reported complete, -1 if never */
  int32_t number_frame;    /* frame after which the card number was accepted, -1 if never */
  int32_t n_numbers;       /* ScannerResult.n_numbers */
  uint8_t predictions[16]; /* ScannerResult.predictions */
  int32_t card_type;       /* dmz_olm.h CardType of the accepted number */
  int32_t expiry_month, expiry_year;
  int32_t count15, count16; /* ScannerState.count15/16 */
  int32_t usable_frames;
  int32_t n_expiry_groups; /* ScannerState.expiry_groups.size() */
  int32_t vseg_y_offset, n_offsets; /* ScannerResult.vseg / hseg of the accepted number */
  uint16_t offsets[16];
  float number_width;      /* NHorizontalSegmentation.number_width of the accepted number */
  int32_t reserved[6];
} dmz_hip_session_result;  /* 128 bytes */

/* Replays, for n_sessions sessions of frames_per_session consecutive per-frame records each
 * (session s owns records [s*F, (s+1)*F)), the SDK loop
 *     scanner_add_frame_with_expiry(state, card, scan_expiry, &frame); scanner_result(state, &result);
 * (scan/scan.h:51-72, scan.cpp:41-194, with expiry_extract's cross-frame half
 * expiry_categorize.cpp:162-376 and dmz_olm.cpp:40-130) until `complete`.  The wall clock is
 * replaced by a frame clock (frame f is handled at 1 + f*frame_interval_ms milliseconds) and
 * the date by (now_year, now_month).  expiry may be NULL (then scan_expiry finds nothing).
 * allow_past_expiry != 0 selects the DMZ_DEBUG / CYTHON_DMZ flavour of expiry_categorize.cpp:236-248.
 * Batched-mode convention: records are produced with the number path always on; once a
 * session's number is accepted a frame counts as usable when DMZ_HIP_FLAG_VSEG_OK is set
 * (frame.cpp:43 with collect_card_number = false), and uncategorised expiry records add nothing. */
int dmz_hip_scan_sessions_batch(dmz_hip_context *ctx, const dmz_hip_frame_result *results,
                                const dmz_hip_expiry_result *expiry, int n_sessions,
                                int frames_per_session, int scan_expiry, int frame_interval_ms,
                                int now_year, int now_month, int allow_past_expiry,
                                dmz_hip_session_result *out);

/* dmz_blur_card (dmz.h:101, dmz.cpp:499-515) on n result images (428 x 270, `channels` = 3 or 4 interleaved
 * bytes per pixel, card i at rgb + i*card_stride, rows tightly packed), in place: the boxes of the first
 * n_offsets - unblur_digits digits of sessions[i] (offsets, number_width, vseg_y_offset as
 * dmz_hip_scan_sessions_batch reports them, = ScannerState.mostRecentUsableHSeg/VSeg) are median-blurred
 * with a 25 x 25 window.  unblur_digits < 0: nothing happens, as in the reference.  Boxes wider than 64
 * pixels (number_width > 62; real segmentations are <= 24) are cut to 64. */
int dmz_hip_blur_cards_batch(dmz_hip_context *ctx, uint8_t *rgb, size_t card_stride, int channels, int n,
                             const dmz_hip_session_result *sessions, int unblur_digits);

/* Single homography, llcv_calc_persp_transform (cv/warp.h:25, warp.cpp:34-125),
 * computed on the device with the same kernel code the batch path uses.
 * src_pts/dst_pts: 4 (x,y) pairs; m: 9 floats row-major (host pointers). */
int dmz_hip_calc_persp_transform(dmz_hip_context *ctx, const float *src_pts,
                                 const float *dst_pts, float *m);
/* The summation order of the HOMOGRAPHY (llcv_calc_persp_transform, and only that: see DMZ_HIP_OPT_EIGEN_SSE2) for the calls
 * of this context that do not say: 0 = Eigen's scalar paths (default), 1 = a stock x86-64 build (SSE2 packets).  It is the
 * default of every later transform / pipeline call -- a call with DMZ_HIP_OPT_EIGEN_SCALAR or DMZ_HIP_OPT_EIGEN_SSE2 in its
 * options chooses for itself -- and applies to dmz_hip_calc_persp_transform. */
int dmz_hip_set_reference_flavour(dmz_hip_context *ctx, int flavour);
/* Batched cvWarpPerspective as used by llcv_unwarp (warp.cpp:153-166) with
 * caller-supplied 3x3 float matrices (n x 9, row-major). */
int dmz_hip_warp_perspective_batch(dmz_hip_context *ctx, const uint8_t *plane, size_t frame_stride,
                                   int row_stride, int width, int height, int n,
                                   const float *matrices, uint8_t *cards, size_t card_stride);

/* Model forward passes on n inputs (device or host pointers): the generated
 * applym_befe75da / applyc_{5c241121,01266c1b,b00bf70c} entry points
 * (models/generated/modelm_befe75da.hpp, modelc_5c241121.hpp, ...), used by the
 * known-answer tests. */
int dmz_hip_apply_vseg_model(dmz_hip_context *ctx, const float *x /* n x 204 */, int n,
                             float *out /* n x 3 */);
int dmz_hip_apply_digit_model(dmz_hip_context *ctx, int model /* 0..2 */,
                              const float *x /* n x 27 x 19 */, int n, float *out /* n x 10 */);
/* applym_730c4cbd (models/expiry/modelm_730c4cbd.hpp) and applyc_bf4dd6c8
 * (models/expiry/modelc_bf4dd6c8.hpp): the slash MLP and the expiry digit CNN. */
int dmz_hip_apply_slash_model(dmz_hip_context *ctx, const float *x /* n x 176 */, int n,
                              float *out /* n x 2 */);
int dmz_hip_apply_expiry_model(dmz_hip_context *ctx, const float *x /* n x 16 x 11 */, int n,
                               float *out /* n x 10 */);

/* The candidate order of the expiry segmentation on caller-supplied lists (host pointers): the reference
 * sorts its window sums and stripe sums with std::sort and a "sum >" comparator (scan/expiry_seg.cpp:75-87, 456,
 * 842), and which of two EQUAL sums comes first is libstdc++'s introsort permutation.  pos[list * stride + i] =
 * position of element i of that list:
 *   kind 0  the form k_expiry_seg uses: a wave follows the introsort's partition phase through every range that still
 *           holds two MARKED elements (marks[list * stride + i] != 0; marks == NULL marks all): any two marked elements
 *           with equal keys are then in the library's order by their pos (the closing insertion sort is stable, so
 *           with all marked the library's whole order is "key descending, then pos").  keys < 2^20, lens <= 420;
 *   kind 1 / 2  the complete sort on one lane (k_expiry_stripes' form, and kind 0's fall-back): pos is the final
 *           position.  keys < 2^20 / 2^25, lens <= 420 / 128.
 * flags[list] = 1 when the wave form met the introsort's depth limit and the one-lane form took over.
 * Known-answer entry like the model passes above (tests/test_gpu_expiry.py: against the reference's own
 * instantiation of std::sort). */
int dmz_hip_expiry_sort_positions(dmz_hip_context *ctx, const int32_t *keys /* n_lists x stride */,
                                  const int32_t *marks /* n_lists x stride, or NULL */,
                                  const int32_t *lens /* n_lists */, int n_lists, int stride, int kind,
                                  int32_t *pos /* n_lists x stride */, int32_t *flags /* n_lists */);

/* Synthetic inputs resident in HBM (bench/test generator; byte-identical to
 * oracle/orc_synth.c).  Frames are 640x480, cards 428x270, tightly packed. */
int dmz_hip_synth_frames(dmz_hip_context *ctx, uint64_t seed, uint64_t first_index, int n,
                         uint8_t *y);
int dmz_hip_synth_cards(dmz_hip_context *ctx, uint64_t seed, uint64_t first_index, int n,
                        uint8_t *cards);

/* Per-stage device timing with hipEvents on the context's stream. */
#define DMZ_HIP_STAGE_DETECT 0
#define DMZ_HIP_STAGE_GEOMETRY 1
#define DMZ_HIP_STAGE_WARP 2
#define DMZ_HIP_STAGE_VSEG 3
#define DMZ_HIP_STAGE_HSEG 4
#define DMZ_HIP_STAGE_DIGITS 5
#define DMZ_HIP_STAGE_EXPIRY_SEG 6
#define DMZ_HIP_STAGE_EXPIRY_CAT 7
#define DMZ_HIP_STAGE_COUNT 8
int dmz_hip_set_profiling(dmz_hip_context *ctx, int enabled);
/* ms[i] = accumulated milliseconds, launches[i] = launch count since the last reset. */
int dmz_hip_get_stage_times(dmz_hip_context *ctx, float ms[DMZ_HIP_STAGE_COUNT],
                            int launches[DMZ_HIP_STAGE_COUNT], int reset);

/* Raw device memory helpers so that non-torch hosts (tests, C++ callers) can
 * keep batches resident. */
int dmz_hip_malloc(dmz_hip_context *ctx, size_t bytes, void **dptr);
int dmz_hip_free(dmz_hip_context *ctx, void *dptr);
int dmz_hip_memcpy_h2d(dmz_hip_context *ctx, void *dst, const void *src, size_t bytes);
int dmz_hip_memcpy_d2h(dmz_hip_context *ctx, void *dst, const void *src, size_t bytes);

/* ---- Frames sharded over the GPUs of a node (north_star: "frames shard embarrassingly across the 8 GPUs
 * of one node with an RCCL gather of ScannerResult over xGMI").  One process -- or one thread -- per GPU, each with its
 * own dmz_hip_context; nothing is exchanged while a batch is scanned, the only collective is the gather of the fixed-size
 * per-frame records on the root.  The reference has no counterpart (it scans one frame per call on one CPU thread): these
 * are the calls a C++ host adds around its dmz_hip_*_batch loop; bench.py and the tests use the same ones.
 *
 * dmz_hip_shard_range: rank r of `world` owns the contiguous frame range [first, first + count); the ranges tile
 * [0, n_total) in rank order (the same split as card.io-dmz_amd/sharding.py).  Pure arithmetic, no device needed.
 *
 * dmz_hip_comm_unique_id / dmz_hip_comm_init: the RCCL communicator of the context.  Rank 0 creates the 128-byte id and the
 * host distributes it (MPI, a socket, a file: the host's business, as with ncclGetUniqueId); every rank then calls
 * dmz_hip_comm_init with it.  librccl is loaded at run time (dlopen): a single-GPU host never needs it, and world = 1 works
 * without it.
 *
 * dmz_hip_gather_records: every rank passes its device-resident records (`count` x record_bytes, its shard of a batch of
 * n_total) -- rank `root` also passes the destination for all n_total records (device memory, frame order); the other
 * ranks pass NULL.  The exchange is a group of point-to-point transfers into the root's xGMI links (ncclSend / ncclRecv:
 * 1 / world of an all-gather's traffic), enqueued on the context's communication queue BEHIND the work already on the
 * context's stream, and returns at once: the next batch's kernels overlap it.  `slot` (0 .. DMZ_HIP_GATHER_SLOTS - 1) names
 * the gather for the wait: the caller alternates between two record buffers / two destinations (one slot per buffer and
 * record type) and calls dmz_hip_gather_wait(slot) before it reuses that buffer or reads that destination -- the gather of
 * batch k-1 stays in flight while batch k is scanned.  dmz_hip_gather_wait makes the context's stream wait for the slot's
 * last gather (and blocks the host if `host_sync`); slot < 0 = every slot. ---- */
#define DMZ_HIP_GATHER_SLOTS 8
void dmz_hip_shard_range(int64_t n_total, int world, int rank, int64_t *first, int64_t *count);
int dmz_hip_comm_unique_id(void *id128);
int dmz_hip_comm_init(dmz_hip_context *ctx, const void *id128, int world, int rank);
int dmz_hip_comm_destroy(dmz_hip_context *ctx);
int dmz_hip_gather_records(dmz_hip_context *ctx, const void *local, size_t record_bytes, int64_t n_total, int root,
                           void *root_dst, int slot);
int dmz_hip_gather_wait(dmz_hip_context *ctx, int slot, int host_sync);

#ifdef __cplusplus
}
#endif
#endif /* DMZ_HIP_H */
