// dmz_hip_internal.h -- shared between the HIP kernels and the C-ABI host code.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dmz_hip.h"

// ---------------------------------------------------------------------------
// Model weights: float offsets into the blob written by tools/extract_models.py
// (card.io-dmz_amd/weights/dmz_models.bin, after its 16-byte header).
// ---------------------------------------------------------------------------
namespace dmzw {
constexpr int VSEG_W1 = 0;                    // 50 x 204   modelm_befe75da.cpp:16
constexpr int VSEG_B1 = VSEG_W1 + 50 * 204;   // 50
constexpr int VSEG_W2 = VSEG_B1 + 50;         // 3 x 50
constexpr int VSEG_B2 = VSEG_W2 + 150;        // 3
constexpr int DIGIT0 = VSEG_B2 + 3;           // 3 x DIGIT_STRIDE  modelc_*.cpp:22-1818
constexpr int D_CONV_W = 0;                   // 8 x 3 x 3
constexpr int D_CONV_B = 72;                  // 8
constexpr int D_HID_W = 80;                   // 32 x 320
constexpr int D_HID_B = 80 + 10240;           // 32
constexpr int D_LOG_W = D_HID_B + 32;         // 10 x 32
constexpr int D_LOG_B = D_LOG_W + 320;        // 10
constexpr int DIGIT_STRIDE = D_LOG_B + 10;    // 10682
constexpr int SLASH = DIGIT0 + 3 * DIGIT_STRIDE;
constexpr int EXPIRY = SLASH + 80 * 176 + 80 + 160 + 2;
constexpr int TOTAL = EXPIRY + 1250 + 50 + 50000 + 40 + 21120 + 176 + 1760 + 10;
}  // namespace dmzw

// ---------------------------------------------------------------------------
// Detection tables.  Everything that the reference evaluates with libm on the
// host for a (plane size, orientation, box) is evaluated on the host here too
// (glibc, same as the x86 reference) and handed to the kernels, so that the
// device never calls a transcendental on the bit-exact detect->warp chain:
//   hough.cpp:112-124 tabSin/tabCos/slope bounds, hough.cpp:190-191 angles,
//   geometry.cpp:37-41 origin shift, geometry.cpp:22 cosf/sinf of the angle.
// ---------------------------------------------------------------------------
constexpr int kNumAngle = 10;  // cvRound(2 * 5deg / 1deg), hough.cpp:98

// Vote counters are kept in this many copies, a lane voting into copy (lane & (copies - 1)): the voters of one instruction are
// neighbours on an edge and name the same few counters, and equal addresses serialise an LDS atomic (detect.hip, phase E).
constexpr int kDetectVoteCopies = 2;
struct DmzBoxParams {
  int x, y, w, h;        // ROI in the plane (dmz.cpp:279-341)
  int vertical;          // LineOrientationVertical (left/right boxes)
  int numrho;            // hough.cpp:99
  int rho_lo, rho_cnt;   // the rho bins a pixel of this box can vote for: [rho_lo, rho_lo + rho_cnt) (fill_box_params)
  int acc_copy_bytes;    // one copy of the vote counters (16-byte multiple); the kernel keeps kDetectVoteCopies of them
  int threshold;         // max(w,h)/6, dmz.cpp:246
  int tab_sin[kNumAngle];
  int tab_cos[kNumAngle];
  float slope_a, slope_b;        // hough.cpp:117-124
  double slope_ta, slope_tb;     // (float)dy / (float)dx >= slope_a  <=>  dy / dx >= slope_ta (midpoint below
                                 // slope_a); <= slope_b  <=>  dy / dx <= slope_tb (midpoint above slope_b)
  float theta_n[kNumAngle];      // n*theta + theta_min (float)
  double delta_rho[kNumAngle];   // geometry.cpp:37-40 for this box origin and angle n
  float cos_t[kNumAngle];        // cosf(theta_n), geometry.cpp:22
  float sin_t[kNumAngle];        // sinf(theta_n)
  float rho_multiplier;          // dmz.cpp:383
  uint32_t inv_w;                // ceil(2^32 / lanes): flat walk-space index -> (step, lane)
  // LDS layout of k_detect_walk for this box (bytes from the dynamic LDS base)
  int lanes, steps;              // across / along extents: (w, h) for horizontal-line boxes, (h, w) otherwise
  int tile_off, tile_stride;     // LDS column of ROI pixel 0, row stride of the source tile
  int lds_map, lds_acc, lds_red; // offsets of the edge map, the accumulator/candidate list, the scratch
  int lds_total;
  int list_cap;                  // candidate list capacity (u16 entries)
  int nthreads;                  // 64 * ceil(lanes / 62)
};

struct DmzDetectParams {
  DmzBoxParams box[4];  // result order: top, left, bottom, right
};

// Per-frame detection scratch written by k_detect_box and consumed by k_geometry.
struct DmzBoxHit {
  int found;  // line found in this box (hough maxVal > threshold)
  int n;      // angle index
  int r;      // rho index
  int max_val;
};

// Inverse homography (dst -> src) of one frame, double, as cv::invert leaves it.
// win[]: per destination strip of k_warp, the source window k_warp_windows derived from the
// strip's four corner pixels (wdw: dword columns; > 0 staged with aligned dword loads, < 0 staged
// with the border-checked loop, 0 = generic path)
// ... and the strip's uniform fp64 constants, evaluated ONCE per strip by k_warp_windows and read by k_warp through scalar
// loads (round 5: k_warp's waves derived them on the vector unit and moved them to scalar registers, ~20 v_readfirstlane and
// ~10 fp64 operations per wave of 30 rows): the rounding constants with the window origin folded in, exact path (mx, my) and
// filtered path (kx, ky: + alpha where the strip takes the affine form; aqx, aqy: that alpha as the adder rounded it).
struct DmzWarpWin {
  int wx0, wy0, wdw, wrows;
  double mx, my, kx, ky, aqx, aqy;
  double rx0, ry0, rw0s;  // the row terms M0 x + M1 y + M2, M3 x + M4 y + M5, (M6 x + M7 y + M8) / 32 at the strip's first row
                          // (the filtered path's chains start from them; the exact sequence evaluates its own)
};
constexpr int DMZ_WARP_STRIPS = 21;
struct DmzWarpMat {
  double m[9];
  int valid;
  int pad_;
  double sw, dw2;  // M7 / 32 = Wd(row + 1) - Wd(row), and twice that (k_warp_windows)
  DmzWarpWin win[DMZ_WARP_STRIPS];
};
// what k_homography / k_mats_from_float write (the rest belongs to k_warp_windows)
__device__ __forceinline__ void dmz_store_mat_head(DmzWarpMat *dst, const DmzWarpMat &wm) {
  for (int i = 0; i < 9; i++) dst->m[i] = wm.m[i];
  dst->valid = wm.valid;
  dst->pad_ = 0;
}

// Expiry path.  Per (frame, stripe) staging written by k_expiry_seg and merged, in stripe order,
// by k_expiry_cat; group headers only (the first 32 bytes of dmz_hip_expiry_group).
struct DmzExpiryStage {
  int n;  // groups this stripe produced (may exceed the 8 kept)
  int pad_[3];
  short hdr[DMZ_HIP_EXPIRY_MAX_GROUPS][16];
};
// cv::bilateralFilter tables (expiry_categorize.cpp:52-57): evaluated with glibc on the host.
struct DmzExpiryTables {
  float color_weight[256];
  float space_weight[8];  // 5 used: (-1,0) (0,-1) (0,0) (0,1) (1,0)
};
// Re-laid-out copies of the expiry models (float offsets into one device buffer)
namespace dmzx {
constexpr int SLASH_W1T = 0;                      // [176][80]  (input-major)
constexpr int CONV2_P = SLASH_W1T + 176 * 80;     // [1252][48]: tap-major, zero-padded to the MFMA tile grid
// conv2 for the bf16 matrix-core variants (v_mfma_f32_16x16x32_bf16): K ordered tap-major with the 50
// maps of a tap padded to 56 (seven runs of eight), 44 k-steps of 32; B fragments stored exactly as
// the lanes load them, [k-step 44][n-tile 3][lane 64][8 bf16], once as the bf16 rounding of the
// weights (HI) and once as the bf16 rounding of the remainder (LO)
constexpr int C2_KSTEPS = 44, C2_MAPS_PAD = 56;
constexpr int CONV2_BH = CONV2_P + 1252 * 48;                // C2_KSTEPS * 3 * 64 * 8 bf16 = 33,792 floats
constexpr int CONV2_BL = CONV2_BH + C2_KSTEPS * 3 * 64 * 4;
// slash MLP hidden layer for v_mfma_f32_16x16x32_bf16: W1 / 255 split into three bf16 parts (hi, mid, lo),
// fragments [part 3][k-step 6][n-tile 5][lane 64][8 bf16]; K = 176 padded to 192
constexpr int SLASH_KSTEPS = 6;
constexpr int SLASH_B3 = CONV2_BL + C2_KSTEPS * 3 * 64 * 4;
// expiry CNN conv1 for v_mfma_f32_16x16x32_bf16: B[k = tap (25, padded to 32)][n = map (50, padded to 64)] split into three
// bf16 parts, fragments [part 3][n-tile 4][lane 64][8 bf16]
constexpr int CONV1_B3 = SLASH_B3 + 3 * SLASH_KSTEPS * 5 * 64 * 4;
// the same two convolutions for v_mfma_f32_16x16x32_f16 (F16X3): operands in two f16 parts (hi = f16 rounding, lo = f16
// rounding of the remainder), fragments as above: conv2 [k-step 44][n-tile 3][lane 64][8 f16] twice, conv1 [part 2][n-tile 4][lane 64][8 f16]
constexpr int CONV2_FH = CONV1_B3 + 3 * 4 * 64 * 4;
constexpr int CONV2_FL = CONV2_FH + C2_KSTEPS * 3 * 64 * 4;
constexpr int CONV1_F2 = CONV2_FL + C2_KSTEPS * 3 * 64 * 4;
// the two dense layers of the expiry CNN as B fragments of v_mfma_f32_16x16x4_f32, four k-steps of a lane per 16 bytes:
// FC1 [n-tile 11][k-group 8][lane 64][4] (k-steps 30, 31 are zero), FC2 [k-group 11][lane 64][4] (units 10 .. 15 are zero)
constexpr int FC1_F = CONV1_F2 + 2 * 4 * 64 * 4;
constexpr int FC2_F = FC1_F + 11 * 8 * 64 * 4;
// slash MLP hidden layer with the vertical 3 / 10 / 3 pass of the Scharr operator folded into the weights: W' over the 18 x 11
// `inter` bytes under a 16 x 11 window, / 255, three bf16 parts x 2^100 (the A operand is the bytes' bits: d x 2^-133),
// fragments [part 3][k-step 7][n-tile 5][lane 64][8 bf16]; k-steps 0..5: inter row 8 (kk & 1) + e, column 2 ks + (kk >> 1);
// k-step 6: inter row 16 + (kk >> 1), column 8 (kk & 1) + e
constexpr int SLASH_FSTEPS = 7;
constexpr int SLASH_F3 = FC2_F + 11 * 64 * 4;
constexpr int TOTAL = SLASH_F3 + 3 * SLASH_FSTEPS * 5 * 64 * 4;
}  // namespace dmzx
// offsets inside the expiry CNN block of the weight blob (modelc_bf4dd6c8.cpp)
namespace dmzw {
constexpr int X_C1W = 0, X_C1B = 1250, X_C2W = 1300, X_C2B = 51300, X_HW = 51340, X_HB = 72460,
              X_LW = 72636, X_LB = 74396;
constexpr int S_W1 = 0, S_B1 = 80 * 176, S_W2 = S_B1 + 80, S_B2 = S_W2 + 160;
}  // namespace dmzw

// Limits of the detect kernel (one (frame, box) per workgroup, box resident in LDS).
constexpr int kDetectMaxLds = 160 * 1024;
constexpr int kDetectMaxThreads = 1024;

// ---- launchers (defined in the .hip files) --------------------------------
int dmz_launch_detect(hipStream_t s, const uint8_t *planes, size_t frame_stride, int row_stride,
                      int n, const DmzDetectParams &p, DmzBoxHit *hits /* n x 4 */,
                      const int *skip_mask /* n x 4 or null: nonzero = already found */);
void dmz_launch_geometry(hipStream_t s, int n, const DmzDetectParams *params_by_plane /* 3, device */,
                         const DmzBoxHit *hits /* 3 planes x n x 4 */, int nplanes,
                         dmz_hip_frame_result *results);
void dmz_launch_homography(hipStream_t s, int n, int orientation, int options,
                           dmz_hip_frame_result *results, DmzWarpMat *mats);
void dmz_launch_persp(hipStream_t s, int n, const float *src_pts, const float *dst_pts, float *m9, int options);
void dmz_launch_mats_from_float(hipStream_t s, int n, const float *m9, DmzWarpMat *mats);
void dmz_launch_warp(hipStream_t s, const uint8_t *planes, size_t frame_stride, int row_stride,
                     int width, int height, int n, DmzWarpMat *mats, uint8_t *cards,
                     size_t card_stride);
// Digit models' hidden matrices in the fragment order of k_digits' chunked FC1 (digits.hip), 3 x 320 x 32 floats' worth, first in
// the buffer.  Round 6 (DMZ_DG_FC1_F16 = 1): B fragments of v_mfma_f32_16x16x32_f16 in two f16 parts (hi = the f16 rounding of the
// weight, lo = of the remainder: 22 bits) -- [model 3][pooled column 5][k-step 2][n-tile 2][part 2][lane 64][8 f16]: lane (unit = 16 nt
// + (lane & 15), run = lane >> 4), element e holds W[unit][map * 40 + row * 5 + column] with map * 8 + row = 32 ks + 8 run + e.
// DMZ_DG_FC1_F16 = 0 (rounds 3 - 5): f32x4 fragments of v_mfma_f32_16x16x4_f32, [model 3][pooled column 5][K-quarter 4][n-tile 2]
// [lane 64]: lane (unit, kq = lane >> 4), element e: map * 8 + row = 16 quarter + 4 kq + e.
#ifndef DMZ_DG_FC1_F16
#define DMZ_DG_FC1_F16 1
#endif
// vseg hidden-layer weights in fragment order, appended to it (float offsets)
namespace dmzv {
constexpr int WFRAG = 3 * 320 * 32;                 // offset of this block in the buffer
// (offsets below are relative to WFRAG)
constexpr int WB3 = 0;                              // W1 / 255 in three bf16 parts x 2^100, fragments of v_mfma_f32_16x16x32_bf16:
                                                    // [wave 4][k-step 7][part 3][lane 64][8 bf16]
constexpr int ROWSUM = 4 * 7 * 3 * 64 * 4;          // sum_k W1[j][k], 64 floats (zero beyond unit 49)
// the digit models' 3x3 conv weights x 1/255 (the input scaling of n_categorize.cpp:99 folded in) as B fragments of
// v_mfma_f32_16x16x32_bf16, [column parity 2][n-tile 2][lane 64][8 bf16]: column n = 16 nt + (lane & 15) = model * 8 + map
// (zero beyond 23), k = 8 (lane >> 4) + slot; the slot -> (tap, bf16 part) tables are in capi.cpp next to digits.hip's
// description of the two K layouts
constexpr int DCONV_B = ROWSUM + 64;                // 2 * 2 * 64 * 4 floats
constexpr int DCONV_BIAS = DCONV_B + 2 * 2 * 64 * 4;  // conv biases by column n (32, zero beyond 23)
// what the tail of the digit CNNs reads, contiguous (k_digits stages it into LDS): hidden biases [model 3][32]; the logistic
// layer as the B operand of a 16 x 32 x 16 product, [model 3][class 16 (10 used, rest zero)][DT_PITCH floats, 32 used]; its
// biases [model 3][16]
constexpr int DT_PITCH = 36;
constexpr int DT_HB = 0, DT_LW = 96, DT_LB = DT_LW + 3 * 16 * DT_PITCH, DT_FLOATS = DT_LB + 48;  // 1872 floats
constexpr int DTAIL = DCONV_BIAS + 32;
constexpr int WFRAG_FLOATS = DTAIL + DT_FLOATS;
}  // namespace dmzv
void dmz_launch_vseg(hipStream_t s, const float *weights, const float *wfrag /* dmzv layout */, const uint8_t *cards, size_t card_stride,
                     int n, int mode /* DMZ_HIP_SCAN_* */, dmz_hip_frame_result *results);
void dmz_launch_hseg(hipStream_t s, const uint8_t *cards, size_t card_stride, int n,
                     dmz_hip_frame_result *results);
void dmz_launch_digits(hipStream_t s, const float *weights, const float *hidw /* fragment-ordered hidden matrices + dmzv block */,
                       const uint8_t *cards, size_t card_stride, int n,
                       dmz_hip_frame_result *results, void *patches /* n x dmz_digit_patch_bytes() of device scratch, zeroed once */);
size_t dmz_digit_patch_bytes(void);
void dmz_launch_vseg_model(hipStream_t s, const float *weights, const float *x, int n, float *out);
void dmz_launch_digit_model(hipStream_t s, const float *weights, const float *hidwt, int model,
                            const float *x, int n, float *out);
size_t dmz_synth_params_bytes(int n);
int dmz_synth_upload_params(hipStream_t s, uint64_t seed, uint64_t first, int n, void *scratch);
void dmz_launch_synth_frames(hipStream_t s, const void *params, int n, uint8_t *y);
void dmz_launch_synth_cards(hipStream_t s, const void *params, int n, uint8_t *cards);
int dmz_launch_fill_lds(hipStream_t s, int device, uint32_t word);  // include/dmz_hip_test.h; nonzero: device attributes unavailable
// Developer probe (tools/dev/marginal_cost.sh): -DDMZ_DUP=<kernel tag> launches that kernel TWICE (every kernel of the pipeline
// is idempotent: same inputs, same outputs): the difference in the step time is what the kernel costs INSIDE the pipeline,
// beside whatever runs on the other queues -- as opposed to its run time alone.
#define DMZ_TAG_detect_h 1
#define DMZ_TAG_detect_v 2
#define DMZ_TAG_warp 3
#define DMZ_TAG_vseg 4
#define DMZ_TAG_hseg 5
#define DMZ_TAG_patches 6
#define DMZ_TAG_digits 7
#define DMZ_TAG_stripes 8
#define DMZ_TAG_xseg 9
#define DMZ_TAG_xcat 10
#define DMZ_TAG_CAT_(x) DMZ_TAG_##x
#define DMZ_TAG_CAT(x) DMZ_TAG_CAT_(x)
#ifdef DMZ_DUP
#define DMZ_REPEAT(tag) for (int dup__ = 0; dup__ < (DMZ_TAG_CAT(DMZ_DUP) == DMZ_TAG_##tag ? 2 : 1); dup__++)
#else
#define DMZ_REPEAT(tag)
#endif
void dmz_launch_expiry(hipStream_t s, const float *weights, const float *xw /* dmzx layout */,
                       const DmzExpiryTables *tables, const uint8_t *cards, size_t card_stride, int n,
                       const dmz_hip_frame_result *results, DmzExpiryStage *stage /* n x 3 */,
                       dmz_hip_expiry_result *out, hipEvent_t mid /* recorded between seg and cat, or null */,
                       int conv_mode /* DMZ_HIP_EXPIRY_CONV_* */, int phases = 3 /* 1: stripes + seg, 2: cat */);
void dmz_launch_sort_order(hipStream_t s, const int *keys, const int *marks /* or null */, const int *lens, int n_lists,
                           int stride, int kind, int *pos, int *flags);
void dmz_launch_slash_model(hipStream_t s, const float *weights, const float *xw, const float *x, int n, float *out);
void dmz_launch_expiry_model(hipStream_t s, const float *weights, const float *xw, const float *x, int n, float *out,
                             int conv_mode);
void dmz_launch_sessions(hipStream_t s, const dmz_hip_frame_result *frames, const dmz_hip_expiry_result *expiry,
                         int n_sessions, int frames_per_session, int scan_expiry, int frame_interval_ms,
                         int now_year, int now_month, int allow_past, dmz_hip_session_result *out);
void dmz_launch_split_c2(hipStream_t s, const uint8_t *src, size_t n_pairs, uint8_t *c1, uint8_t *c2);
void dmz_launch_rgba_to_r(hipStream_t s, const uint8_t *src, size_t n_px, uint8_t *dst);
void dmz_launch_ycbcr_to_rgb(hipStream_t s, const uint8_t *y, const uint8_t *cb, const uint8_t *cr, size_t n_px,
                             int channels, uint8_t *rgb);
void dmz_launch_scores(hipStream_t s, const uint8_t *y, size_t frame_stride, int row_stride, int n, int rx, int ry,
                       int rw, int rh, float *focus, float *brightness);
void dmz_launch_scharr3_dx_abs(hipStream_t s, const uint8_t *src, int src_stride, int w, int h, int16_t *dst, int dst_stride);
void dmz_launch_blur_cards(hipStream_t s, uint8_t *rgb, size_t card_stride, int channels, int n,
                           const dmz_hip_session_result *sessions, int unblur_digits);
int dmz_configure_expiry(void);
int dmz_configure_detect(void);  // one-time hipFuncSetAttribute calls; return hipError_t
int dmz_configure_scan(void);
int dmz_configure_vseg(void);
