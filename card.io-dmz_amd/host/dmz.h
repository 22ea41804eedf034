// dmz.h -- HIP_DMZ flavour of the reference's public headers for the per-frame scan
// path: dmz.h:13-101, dmz_olm.h:17-104, scan/scan.h:17-72, scan/frame.h:14-28,
// scan/n_vseg.h:14-21, scan/n_hseg.h:13-19, cv/warp.h:20-25, mz.h:19-25.
//
// Same names, argument meaning, ownership and (lack of) error codes as the reference, so
// that an SDK call site RECOMPILES against this header and links libdmz_host.so (source
// compatibility: the structs below keep the reference's field order, but Eigen and OpenCV
// types are replaced, so object files built against the reference headers do not relink);
// every function is a batch-of-1 wrapper over the C-ABI of include/dmz_hip.h.  `dmz_context.mz` holds the
// dmz_hip_context exactly where the Android flavour keeps its GLES warp context
// (mz_android.cpp:233-240).  Differences, all forced by dropping Eigen/OpenCV types:
//   * NumberScores / NumberPredictions are plain arrays (reference: Eigen matrices,
//     n_categorize.h:14, scan.h:17) with the accessors call sites use: scores(i, k),
//     predictions(i, 0), rows(), cols(), sum(), setZero();
//   * ExpiryGroupScores is a plain 11 x 10 float array with the same accessors (reference:
//     Eigen, expiry_types.h:47);
//     CharacterRect.sum of the groups a frame reports is 0 (the device does not return it and
//     nothing downstream reads it);
//   * IplImage is declared here with OpenCV 2.4's field layout (types_c.h:462-507) unless
//     OpenCV's own header was included first.
// There is NO CPU fallback: without a GPU dmz_context_create() returns NULL and the
// per-frame calls report "not found"/"not usable" after logging once to stderr.
#ifndef DMZ_HIP_HOST_DMZ_H
#define DMZ_HIP_HOST_DMZ_H

#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#ifndef __OPENCV_CORE_TYPES_H__
#define IPL_DEPTH_8U 8
typedef struct _IplROI {
  int coi, xOffset, yOffset, width, height;
} IplROI;
typedef struct _IplImage {
  int nSize, ID, nChannels, alphaChannel, depth;
  char colorModel[4], channelSeq[4];
  int dataOrder, origin, align, width, height;
  struct _IplROI *roi;
  struct _IplImage *maskROI;
  void *imageId;
  struct _IplTileInfo *tileInfo;
  int imageSize;
  char *imageData;
  int widthStep;
  int BorderMode[4], BorderConst[4];
  char *imageDataOrigin;
} IplImage;
#endif

#define kCreditCardTargetWidth 428
#define kCreditCardTargetHeight 270
#define kNumberWidth 19
#define kNumberHeight 27

// ---- dmz_olm.h:17-62 ----
typedef uint8_t FrameOrientation;
enum {
  FrameOrientationPortrait = 1,
  FrameOrientationPortraitUpsideDown = 2,
  FrameOrientationLandscapeRight = 3,
  FrameOrientationLandscapeLeft = 4
};
typedef struct { float x, y; } dmz_point;
typedef struct { float x, y, w, h; } dmz_rect;
typedef struct { dmz_point top_left, bottom_left, top_right, bottom_right; } dmz_corner_points;
typedef uint8_t CardType;
enum {
  CardTypeUnrecognized = 0, CardTypeAmbiguous, CardTypeAmex, CardTypeJCB, CardTypeVisa,
  CardTypeMastercard, CardTypeDiscover, CardTypeMaestro
};
typedef struct {
  CardType card_type;
  int number_length, prefix_length;
  long min_prefix, max_prefix;
} dmz_card_info;

// ---- dmz.h:17-37 ----
typedef struct { void *mz; } dmz_context;
typedef struct { float rho, theta; } ParametricLine;
typedef struct { int found; ParametricLine location; } dmz_found_edge;
typedef struct { dmz_found_edge top, left, bottom, right; } dmz_edges;

// ---- scan/n_vseg.h:14-21, scan/n_hseg.h:13-19 ----
typedef uint8_t NumberPatternType;
typedef struct {
  float score;
  uint16_t y_offset;
  NumberPatternType pattern_type;
  uint8_t number_pattern[19];
  uint8_t number_pattern_length;
  uint8_t number_length;
} NVerticalSegmentation;
typedef struct {
  uint8_t n_offsets;
  uint16_t offsets[16];
  float score;
  float number_width;
  uint16_t pattern_offset;
} NHorizontalSegmentation;
// Eigen::Matrix<float, R, C, RowMajor> as a plain array with the Eigen accessors the SDK call sites use
template <int R, int C>
struct DmzScoreMatrix {
  float v[R][C];
  float &operator()(int r, int c) { return v[r][c]; }
  const float &operator()(int r, int c) const { return v[r][c]; }
  static int rows() { return R; }
  static int cols() { return C; }
  void setZero() { for (int r = 0; r < R; r++) for (int c = 0; c < C; c++) v[r][c] = 0.0f; }
  float sum() const {  // plain row-major order (Eigen's redux order differs in the last ulp)
    float s = 0.0f;
    for (int r = 0; r < R; r++) for (int c = 0; c < C; c++) s += v[r][c];
    return s;
  }
};
typedef DmzScoreMatrix<16, 10> NumberScores;             // row-major, n_categorize.h:14
struct NumberPredictions {                               // Eigen::Matrix<Index, 16, 1>, scan.h:17
  long v[16];
  long &operator()(int r, int = 0) { return v[r]; }
  const long &operator()(int r, int = 0) const { return v[r]; }
  static int rows() { return 16; }
  static int cols() { return 1; }
};

// ---- scan/expiry_types.h:16-79 ----
#define kSmallCharacterWidth 9
#define kSmallCharacterHeight 15
#define kTrimmedCharacterImageWidth 11
#define kTrimmedCharacterImageHeight 16
#define kExpiryMaxValidLength 11
enum ExpiryPattern {
  ExpiryPatternMMsYY,
  ExpiryPatternMMs20YY,
  ExpiryPatternXXsXXsYY,
  ExpiryPatternXXsXXs20YY,
  ExpiryPatternMMdMMsYY,
  ExpiryPatternMMdMMs20YY,
  ExpiryPatternMMsYYdMMsYY,
};
typedef DmzScoreMatrix<kExpiryMaxValidLength, 10> ExpiryGroupScores;
struct CharacterRect {
  int top;
  int left;
  long sum;
  CharacterRect() : top(0), left(0), sum(0) {}
  CharacterRect(const int top, const int left, const long sum) : top(top), left(left), sum(sum) {}
};
typedef std::vector<CharacterRect> CharacterRectList;
struct GroupedRects {
  int top;
  int left;
  int width;
  int height;
  bool grouped_yet;
  long sum;
  int character_width;
  CharacterRectList character_rects;
  ExpiryPattern pattern;
  ExpiryGroupScores scores;
  int recently_seen_count;  // used when aggregating groups across frames
  int total_seen_count;     // used when aggregating groups across frames
};
typedef std::vector<GroupedRects> GroupedRectsList;

// ---- scan/frame.h:14-28 ----
typedef struct {
  float focus_score;
  NumberScores scores;
  NHorizontalSegmentation hseg;
  NVerticalSegmentation vseg;
  GroupedRectsList expiry_groups;
  GroupedRectsList name_groups;
  bool usable;
  bool upside_down;
  bool flipped;
  float brightness_score;
  uint16_t iso_speed;
  float shutter_speed;
  bool torch_is_on;
} FrameScanResult;

// ---- scan/scan_analytics.h:12-33 (the analytics hook is a counter + ring index in the reference too:
// scan_analytics.cpp:17-20 records no field) ----
#define kScanSessionNumFramesStored 20
typedef struct {
  uint32_t frame_index;
  std::map<std::string, std::string> frame_values;
} ScanFrameAnalytics;
typedef struct {
  uint32_t num_frames_scanned;
  uint8_t frames_ring_start;
  ScanFrameAnalytics frames_ring[kScanSessionNumFramesStored];
} ScanSessionAnalytics;

// ---- scan/scan.h:19-48 ----
typedef struct {
  bool complete;
  NumberPredictions predictions;
  NHorizontalSegmentation hseg;
  NVerticalSegmentation vseg;
  uint8_t n_numbers;
  int expiry_month, expiry_year;
} ScannerResult;
typedef struct ScannerState {
  uint16_t count15, count16;
  NumberScores aggregated15, aggregated16;
  ScanSessionAnalytics session_analytics;  // same position as scan/scan.h:38
  ScannerResult successfulCardNumberResult;
  NHorizontalSegmentation mostRecentUsableHSeg;
  NVerticalSegmentation mostRecentUsableVSeg;
  unsigned long timeOfCardNumberCompletionInMilliseconds;
  bool scan_expiry;
  int expiry_month, expiry_year;
  GroupedRectsList expiry_groups;
  GroupedRectsList name_groups;
  // -- end of the reference's fields (scan/scan.h:33-48); appended by the HIP flavour: --
  dmz_context *dmz;  // the context the frames are scanned on (NULL = the calling thread's default context)
} ScannerState;

// life cycle (dmz.h:48-57, mz.h:19-25, processor_support.h pattern)
dmz_context *dmz_context_create(void);
void dmz_context_destroy(dmz_context *dmz);
void dmz_prepare_for_backgrounding(dmz_context *dmz);
void *mz_create(void);
void mz_destroy(void *mz);
void mz_prepare_for_backgrounding(void *mz);
int dmz_has_hip_runtime(void);
// dmz.h:60: "can images be allocated" probe of the reference (cvCreateImage); here of dmz_create_image_8u
int dmz_has_opencv(void);
// processor_support.h:59-66.  No NEON / VFP code exists in this flavour (both 0).  The GLES-warp switch
// of the Android flavour maps onto the HIP warp: dmz_use_gles_warp() reports whether the accelerator
// rectifies (a HIP device is present and the switch is on); dmz_set_gles_warp(0) is recorded and reported
// back, but it cannot select a CPU path -- there is none: llcv_unwarp then fails loudly (stderr, output
// untouched) instead of falling back as mz_android.cpp:8-24 does.
int dmz_has_neon_runtime(void);
int dmz_use_vfp3_16(void);
int dmz_use_gles_warp(void);
void dmz_set_gles_warp(int newstate);

// detection / transformation (dmz.h:78-96, cv/warp.h:20-25)
bool dmz_found_all_edges(dmz_edges found_edges);
bool dmz_detect_edges(IplImage *y_sample, IplImage *cb_sample, IplImage *cr_sample,
                      FrameOrientation orientation, dmz_edges *found_edges,
                      dmz_corner_points *corner_points);
void dmz_transform_card(dmz_context *dmz, IplImage *sample, dmz_corner_points corner_points,
                        FrameOrientation orientation, bool upsample, IplImage **transformed);
void llcv_calc_persp_transform(float *matrixData, int matrixDataSize, bool rowMajor,
                               const dmz_point sourcePoints[], const dmz_point destPoints[]);
void llcv_unwarp(dmz_context *dmz, IplImage *input, const dmz_point source_points[4],
                 const dmz_rect to_rect, IplImage *output);
bool llcv_warp_auto_upsamples(void);

// camera-side plumbing (dmz.h:64-72); *channel1 / *channel2 / *rgb are allocated when NULL (caller frees)
void dmz_deinterleave_uint8_c2(IplImage *interleaved, IplImage **channel1, IplImage **channel2);
void dmz_deinterleave_RGBA_to_R(uint8_t *source, uint8_t *dest, int size);
void dmz_YCbCr_to_RGB(IplImage *y, IplImage *cb, IplImage *cr, IplImage **rgb);

// quality scores (dmz.h:77-79, dmz.cpp:114-199)
float dmz_focus_score(IplImage *image, bool use_full_image);
float dmz_brightness_score(IplImage *image, bool use_full_image);

// scan/frame.h:30-46: one rectified card through the number (and expiry) path on the calling thread's default context;
// the Cython flavour's wrapper keeps usable / hseg / vseg only
void scan_card_image(IplImage *y, bool collect_card_number, bool scan_expiry, FrameScanResult *result);
typedef struct {
  NHorizontalSegmentation hseg;
  NVerticalSegmentation vseg;
  bool usable;
} CythonFrameScanResult;
void cython_scan_card_image(IplImage *y, CythonFrameScanResult *result);

// scanning (scan/scan.h:51-72)
void scanner_initialize(ScannerState *state);
void scanner_reset(ScannerState *state);
void scanner_add_frame(ScannerState *state, IplImage *y, FrameScanResult *result);
void scanner_add_frame_with_expiry(ScannerState *state, IplImage *y, bool scan_expiry,
                                   FrameScanResult *result);
void scanner_result(ScannerState *state, ScannerResult *result);
void scanner_destroy(ScannerState *state);

// dmz.h:101, dmz.cpp:499-515 (8-bit, 3 or 4 channels, 428 x 270)
void dmz_blur_card(IplImage *cardImageRGB, ScannerState *state, int unblurDigits);

// the cross-frame half of scan/expiry_categorize.cpp (:162-330), exported for the host-logic tests
void expiry_aggregate_grouped_rects(GroupedRectsList &aggregated_groups, GroupedRectsList &new_groups);
void get_stable_expiry_month_and_year(GroupedRects &group, int *expiry_month, int *expiry_year);
// expiry_categorize.cpp:236-248 accepts dates in the past only in the DMZ_DEBUG / CYTHON_DMZ
// flavours; the production behaviour is the default here, this switches to the other one
void dmz_hip_host_allow_past_expiry(bool allow);

// dmz_olm.h:70-104
dmz_point dmz_create_point(float x, float y);
dmz_rect dmz_create_rect(float x, float y, float w, float h);
void dmz_rect_get_points(dmz_rect rect, dmz_point points[4]);
dmz_point dmz_scale_point(const dmz_point src_p, const dmz_rect src_f, const dmz_rect dst_f);
dmz_rect dmz_guide_frame(FrameOrientation orientation, float preview_width, float preview_height);
FrameOrientation dmz_opposite_orientation(FrameOrientation orientation);
bool dmz_passes_luhn_checksum(uint8_t *number_array, uint8_t number_length);
dmz_card_info dmz_card_info_for_prefix_and_length(uint8_t *number_array, uint8_t number_length,
                                                  bool allow_incomplete_number);

// ---- the Cython flavour's extra entry points (dmz.h:103-119 under CYTHON_DMZ, scan/expiry_types.h:94-117, mz.h:37-52) ----
#define IPL_DEPTH_16S ((int)0x80000000 | 16)
typedef struct {
  int top;
  int left;
} CythonCharacterRect;
typedef float CythonGroupScores[kExpiryMaxValidLength][10];
typedef struct {
  int top;
  int left;
  int width;
  int height;
  int character_width;
  uint8_t pattern;
  CythonGroupScores scores;
  int recently_seen_count;
  int total_seen_count;
  int number_of_character_rects;
  CythonCharacterRect *character_rects;
} CythonGroupedRects;
// dmz.h:105 -- src 8U one channel, dst 16S one channel of the same size (ROIs honoured as image extents)
void dmz_scharr3_dx_abs(IplImage *src, IplImage *dst);
// dmz.h:110 -- *expiry_groups is malloc'ed (and each group's character_rects), as in the reference; the caller frees
void dmz_best_expiry_seg(IplImage *card_y, uint16_t starting_y_offset, CythonGroupedRects **expiry_groups, uint16_t *number_of_groups);
// dmz.h:111-114 -- categorises the digits of the new groups, aggregates them into the session's groups and picks the stable
// month / year.  As in dmz.cpp:625-655 only the SESSION's array is re-allocated and rewritten (the categorised scores come
// back there); the new-groups array and its count are left untouched, and no character_rects array is freed
void dmz_expiry_extract(IplImage *card_y, uint16_t *number_of_expiry_groups, CythonGroupedRects **cython_expiry_groups,
                        uint16_t *number_of_new_groups, CythonGroupedRects **cython_new_groups, int *expiry_month,
                        int *expiry_year);
// dmz.h:115-119
void dmz_expiry_extract_group(IplImage *card_y, CythonGroupedRects &cython_group, CythonGroupScores cython_scores,
                              int *expiry_month, int *expiry_year);
// mz.h:37-52: image headers over caller data (no Python object is involved in these signatures)
IplImage *py_mz_create_from_cv_image_data(char *image_data, int image_size, int width, int height, int64_t depth,
                                          int n_channels, int roi_x_offset, int roi_y_offset, int roi_width, int roi_height);
void py_mz_release_ipl_image(IplImage *image);
void py_mz_get_cv_image_data(IplImage *source, char **image_data, int *image_size, int *width, int *height, int64_t *depth,
                             int *n_channels, int *roi_x_offset, int *roi_y_offset, int *roi_width, int *roi_height);
void py_mz_cvSetImageROI(IplImage *image, int left, int top, int width, int height);
void py_mz_cvResetImageROI(IplImage *image);

// image helpers standing in for cvCreateImage / cvReleaseImage on 8-bit images
// (dmz_transform_card allocates *transformed when it is NULL; the caller frees it)
IplImage *dmz_create_image_8u(int width, int height, int channels);
void dmz_release_image(IplImage **image);

#endif  // DMZ_HIP_HOST_DMZ_H
