/*
 * orc_cv.c -- ORACLE (test infrastructure only, see dmz_oracle.h): CPU
 * restatement of the card-detection and rectification half of the hot path.
 *
 * Follows, in /root/reference: dmz.cpp:199-497, cv/sobel.cpp:476-478,
 * cv/canny.cpp:58-336,555-580, cv/hough.cpp:52-195, geometry.cpp:10-43,
 * cv/warp.cpp:34-169, Eigen/src/QR/HouseholderQR.h:219-250,306-334,
 * Eigen/src/Householder/Householder.h:65-130, Eigen/src/LU/Inverse.h:70-89,
 * and the OpenCV 2.4 semantics of cvSobel / cvWarpPerspective restated in
 * SURVEY.md Appendix A (A2, A10) -- the OpenCV sources are NOT in the reference
 * tree, so those two are "parity unpinned".
 *
 * Compile with -ffp-contract=off: every float/double expression below is meant
 * to be evaluated exactly as written (one IEEE rounding per operator).
 */
#include "dmz_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_PI 3.1415926535897932384626433832795 /* CV_PI, opencv2/core/types_c.h */

static int orc_round(double v) { return (int)lrint(v); }  /* cvRound: half-to-even */
static int orc_floor(double v) { return (int)floor(v); }  /* cvFloor */

/* ------------------------------------------------------------------------- */
/* dmz.cpp:279-341 (+ dmz_constants.h:16-27, geometry.h:10-15)               */
void orc_detection_boxes(int width_in, int height, int orientation, int boxes[4][4]) {
  int inset_v = 0, slop_v = 0, inset_h = 0, slop_h = 0;
  int width = (height * 4) / 3;
  int left_margin = (width_in - width) / 2;
  const float slop_pct = 0.03f;
  if (orientation == 1 || orientation == 2) { /* portrait */
    float pv = (float)((480 - 428) / 2) / (float)480; /* kPortraitHorizontalPercentInset */
    float ph = (float)((640 - 270) / 2) / (float)640; /* kPortraitVerticalPercentInset */
    inset_v = (int)roundf(pv * height);
    slop_v = (int)roundf(slop_pct * height);
    inset_h = (int)roundf(ph * width);
    slop_h = (int)roundf(slop_pct * width);
  } else if (orientation == 3 || orientation == 4) { /* landscape */
    float pv = (float)((480 - 270) / 2) / (float)480; /* kLandscapeVerticalPercentInset */
    float ph = (float)((640 - 428) / 2) / (float)640; /* kLandscapeHorizontalPercentInset */
    inset_v = (int)roundf(pv * height);
    slop_v = (int)roundf(slop_pct * height);
    inset_h = (int)roundf(ph * width);
    slop_h = (int)roundf(slop_pct * width);
  }
  int ix = left_margin, iy = 0, iw = width - 1, ih = height - 1;
  int ox = ix + (inset_h - slop_h), oy = iy + (inset_v - slop_v);
  /* outer width/height are not used by the boxes */
  int nx = ix + (inset_h + slop_h), ny = iy + (inset_v + slop_v);
  int nw = iw - 2 * (inset_h + slop_h), nh = ih - 2 * (inset_v + slop_v);
  int b[4][4] = {
      {nx, oy, nw, 2 * slop_v},      /* top */
      {nx, ny + nh, nw, 2 * slop_v}, /* bottom */
      {ox, ny, 2 * slop_h, nh},      /* left */
      {nx + nw, ny, 2 * slop_h, nh}, /* right */
  };
  memcpy(boxes, b, sizeof(b));
}

/* ------------------------------------------------------------------------- */
/* cvSobel(src, dst, dx, dy, 7) on an isolated ROI (sobel.cpp:476-478;
 * semantics SURVEY A1/A2): separable integer correlation, BORDER_REPLICATE at
 * the ROI edge, exact int32 accumulation, saturate to int16. */
static const int k_deriv7[7] = {-1, -4, -5, 0, 5, 4, 1};
static const int k_smooth7[7] = {1, 6, 15, 20, 15, 6, 1};

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

void orc_sobel7(const uint8_t *src, int stride, int w, int h, int want_dx, int16_t *dst) {
  const int *kx = want_dx ? k_deriv7 : k_smooth7;
  const int *ky = want_dx ? k_smooth7 : k_deriv7;
  int *tmp = (int *)malloc(sizeof(int) * (size_t)w * (size_t)h);
  for (int r = 0; r < h; r++) {
    const uint8_t *row = src + (size_t)r * stride;
    for (int c = 0; c < w; c++) {
      int acc = 0;
      for (int j = 0; j < 7; j++) acc += kx[j] * (int)row[clampi(c + j - 3, 0, w - 1)];
      tmp[r * w + c] = acc;
    }
  }
  for (int r = 0; r < h; r++) {
    for (int c = 0; c < w; c++) {
      int acc = 0;
      for (int i = 0; i < 7; i++) acc += ky[i] * tmp[clampi(r + i - 3, 0, h - 1) * w + c];
      dst[r * w + c] = (int16_t)clampi(acc, -32768, 32767);
    }
  }
  free(tmp);
}

/* ------------------------------------------------------------------------- */
/* canny.cpp:355-361 (cvAbs saturates: |-32768| -> 32767), 568-580, 58-336 */
static double orc_sum_abs_magnitude(const int16_t *img, int n) {
  long long s = 0;
  for (int i = 0; i < n; i++) {
    int v = img[i];
    v = v < 0 ? -v : v;
    if (v > 32767) v = 32767;
    s += v;
  }
  return (double)s;
}

void orc_adaptive_canny7(const int16_t *dx, const int16_t *dy, int w, int h, uint8_t *out,
                         int *low_out, int *high_out) {
  double mean = (orc_sum_abs_magnitude(dx, w * h) + orc_sum_abs_magnitude(dy, w * h)) / (w * h);
  double low_thresh = mean;
  double high_thresh = 3.0f * low_thresh;
  if (low_thresh > high_thresh) { double t = low_thresh; low_thresh = high_thresh; high_thresh = t; }
  const int low = orc_floor(low_thresh);
  const int high = orc_floor(high_thresh);
  if (low_out) *low_out = low;
  if (high_out) *high_out = high;

  const int mapstep = w + 2;
  uint8_t *map = (uint8_t *)malloc((size_t)mapstep * (h + 2));
  int *magbuf = (int *)calloc((size_t)(w + 2) * 3, sizeof(int));
  int *mag_buf[3] = {magbuf, magbuf + (w + 2), magbuf + 2 * (w + 2)};
  size_t stack_cap = (size_t)w * h + 16;
  uint8_t **stack = (uint8_t **)malloc(sizeof(uint8_t *) * stack_cap);
  size_t sp = 0;

  memset(map, 1, (size_t)mapstep);
  memset(map + (size_t)mapstep * (h + 1), 1, (size_t)mapstep);

  const int TG22 = (int)(0.4142135623730950488016887242097 * (1 << 15) + 0.5);

  for (int i = 0; i <= h; i++) {
    int *_mag = mag_buf[(i > 0) + 1] + 1;
    if (i < h) {
      _mag[-1] = _mag[w] = 0;
      for (int j = 0; j < w; j++) _mag[j] = abs((int)dx[i * w + j]) + abs((int)dy[i * w + j]);
    } else {
      memset(_mag - 1, 0, (size_t)(w + 2) * sizeof(int));
    }
    if (i == 0) continue;

    uint8_t *_map = map + (size_t)mapstep * i + 1;
    _map[-1] = _map[w] = 1;
    _mag = mag_buf[1] + 1;
    const int16_t *_dx = dx + (size_t)(i - 1) * w;
    const int16_t *_dy = dy + (size_t)(i - 1) * w;
    const ptrdiff_t magstep1 = mag_buf[2] - mag_buf[1];
    const ptrdiff_t magstep2 = mag_buf[0] - mag_buf[1];
    int prev_flag = 0;

    for (int j = 0; j < w; j++) {
      int64_t x = _dx[j], y = _dy[j];
      const int s = (x ^ y) < 0 ? -1 : 1;
      const int m = _mag[j];
      x = llabs(x);
      y = llabs(y);
      if (m > low) {
        const int64_t tg22x = x * TG22;
        const int64_t tg67x = tg22x + ((x + x) << 15);
        int is_max;
        y <<= 15;
        if (y < tg22x)
          is_max = m > _mag[j - 1] && m >= _mag[j + 1];
        else if (y > tg67x)
          is_max = m > _mag[j + magstep2] && m >= _mag[j + magstep1];
        else
          is_max = m > _mag[j + magstep2 - s] && m > _mag[j + magstep1 + s];
        if (is_max) {
          if (m > high && !prev_flag && _map[j - mapstep] != 2) {
            _map[j] = 2;
            stack[sp++] = _map + j;
            prev_flag = 1;
          } else {
            _map[j] = 0;
          }
          continue;
        }
      }
      prev_flag = 0;
      _map[j] = 1;
    }
    int *t = mag_buf[0];
    mag_buf[0] = mag_buf[1];
    mag_buf[1] = mag_buf[2];
    mag_buf[2] = t;
  }

  static const int nb_dx[8] = {-1, 1, -1, 0, 1, -1, 0, 1};
  static const int nb_dy[8] = {0, 0, -1, -1, -1, 1, 1, 1};
  while (sp > 0) {
    uint8_t *m = stack[--sp];
    for (int k = 0; k < 8; k++) {
      uint8_t *q = m + nb_dy[k] * mapstep + nb_dx[k];
      if (!*q) {
        *q = 2;
        stack[sp++] = q;
      }
    }
  }
  for (int i = 0; i < h; i++) {
    const uint8_t *_map = map + (size_t)mapstep * (i + 1) + 1;
    for (int j = 0; j < w; j++) out[i * w + j] = (uint8_t)-(_map[j] >> 1);
  }
  free(stack);
  free(magbuf);
  free(map);
}

/* ------------------------------------------------------------------------- */
/* hough.cpp:52-195 called with dmz.cpp:246-258's parameters:
 * rho=1, theta=pi/180, threshold=max(w,h)/6, +-5 deg around 90/180 deg,
 * gradient angle threshold 10 deg. */
int orc_hough(const uint8_t *edges, const int16_t *dxm, const int16_t *dym, int width, int height,
              int vertical, float *rho_out, float *theta_out, int *n_out, int *r_out,
              int *max_out) {
  const float rho = 1.0f;
  const float theta = (float)ORC_PI / 180.0f;
  const int threshold = (width > height ? width : height) / 6;
  const float base_angle = vertical ? (float)ORC_PI : (float)(ORC_PI / 2.0f);
  const float max_dev = (float)(5.0f * (ORC_PI / 180.0f));
  const float theta_min = base_angle - max_dev;
  const float theta_max = base_angle + max_dev;
  const float gat = 10;

  const float irho = 1 / rho;
  const int numangle = orc_round((theta_max - theta_min) / theta);
  const int numrho = orc_round(((width + height) * 2 + 1) / rho);
  int *accum = (int *)calloc((size_t)(numangle + 2) * (numrho + 2), sizeof(int));
  int *tab_sin = (int *)malloc(sizeof(int) * numangle);
  int *tab_cos = (int *)malloc(sizeof(int) * numangle);
  float ang = theta_min;
  for (int n = 0; n < numangle; ang += theta, n++) {
    tab_sin[n] = (int)floorf(1024 * sinf(ang) * irho);
    tab_cos[n] = (int)floorf(1024 * cosf(ang) * irho);
  }
  float slope_a, slope_b;
  if (vertical) {
    slope_a = tanf((float)(ORC_PI * (180 - gat) / 180.0f));
    slope_b = tanf((float)(ORC_PI * (180 + gat) / 180.0f));
  } else {
    slope_a = tanf((float)(ORC_PI * (90 - gat) / 180.0f));
    slope_b = tanf((float)(ORC_PI * (90 + gat) / 180.0f));
  }
  for (int i = 0; i < height; i++) {
    for (int j = 0; j < width; j++) {
      if (edges[i * width + j] == 0) continue;
      const int16_t del_x = dxm[i * width + j], del_y = dym[i * width + j];
      int use = 0;
      if (del_x != 0) {
        float slope = (float)del_y / (float)del_x;
        if (vertical)
          use = slope >= slope_a && slope <= slope_b;
        else
          use = slope >= slope_a || slope <= slope_b;
      } else {
        use = !vertical;
      }
      if (!use) continue;
      for (int n = 0; n < numangle; n++) {
        int r = (j * tab_cos[n] + i * tab_sin[n]) >> 10;
        r += (numrho - 1) / 2;
        accum[(n + 1) * (numrho + 2) + r + 1]++;
      }
    }
  }
  int max_val = 0, max_base = 0;
  for (int r = 0; r < numrho; r++)
    for (int n = 0; n < numangle; n++) {
      int base = (n + 1) * (numrho + 2) + r + 1;
      if (accum[base] > max_val) {
        max_val = accum[base];
        max_base = base;
      }
    }
  int found = 0;
  *rho_out = 0.0f;
  *theta_out = 0.0f;
  if (n_out) *n_out = -1;
  if (r_out) *r_out = -1;
  if (max_out) *max_out = max_val;
  if (max_val > threshold) {
    float scale = 1.0f / (numrho + 2);
    int n = orc_floor(max_base * scale) - 1;
    int r = max_base - (n + 1) * (numrho + 2) - 1;
    *rho_out = (r - (numrho - 1) * 0.5f) * rho;
    *theta_out = n * theta + theta_min;
    if (n_out) *n_out = n;
    if (r_out) *r_out = r;
    found = 1;
  }
  free(tab_cos);
  free(tab_sin);
  free(accum);
  return found;
}

/* dmz.cpp:224-271 */
int orc_best_line(const uint8_t *plane, int stride, int x, int y, int w, int h, int vertical,
                  float *rho, float *theta) {
  const uint8_t *roi = plane + (size_t)y * stride + x;
  int16_t *dx = (int16_t *)malloc(sizeof(int16_t) * (size_t)w * h);
  int16_t *dy = (int16_t *)malloc(sizeof(int16_t) * (size_t)w * h);
  uint8_t *canny = (uint8_t *)malloc((size_t)w * h);
  orc_sobel7(roi, stride, w, h, 1, dx);
  orc_sobel7(roi, stride, w, h, 0, dy);
  orc_adaptive_canny7(dx, dy, w, h, canny, NULL, NULL);
  int found = orc_hough(canny, dx, dy, w, h, vertical, rho, theta, NULL, NULL, NULL);
  if (!found) {
    *rho = FLT_MAX; /* ParametricLineNone(), geometry.h:24-29 */
    *theta = FLT_MAX;
  }
  free(canny);
  free(dy);
  free(dx);
  return found;
}

/* geometry.cpp:34-43.  atan() on a float argument resolves to the float
 * overload in C++ (atanf); the rest is double. */
void orc_line_by_shifting_origin(float rho, float theta, int xoff, int yoff, float *rho_out,
                                 float *theta_out) {
  double offset_angle = xoff == 0 ? ORC_PI / 2.0f : (double)atanf((float)yoff / (float)xoff);
  double delta_angle = theta - offset_angle + ORC_PI / 2.0f;
  double offset_magnitude = sqrt((double)(xoff * xoff + yoff * yoff));
  double delta_rho = offset_magnitude * cos(ORC_PI / 2 - delta_angle);
  *theta_out = theta;
  *rho_out = (float)(rho + delta_rho);
}

/* geometry.cpp:14-32 with Eigen's fixed 2x2 determinant / inverse / product
 * (Eigen/src/LU/Determinant.h, Inverse.h:70-89) written out. */
int orc_parametric_intersect(float rho1, float theta1, float rho2, float theta2, float *x,
                             float *y) {
  if (theta1 == FLT_MAX || theta2 == FLT_MAX) return 0;
  const float t00 = cosf(theta1), t01 = sinf(theta1), t10 = cosf(theta2), t11 = sinf(theta2);
  const float det = t00 * t11 - t10 * t01;
  if (det < 1e-10) return 0;
  const float invdet = 1.0f / det;
  const float i00 = t11 * invdet, i10 = -t10 * invdet, i01 = -t01 * invdet, i11 = t00 * invdet;
  *x = i00 * rho1 + i01 * rho2;
  *y = i10 * rho1 + i11 * rho2;
  return 1;
}

/* dmz.cpp:346-439 */
static void orc_find_line(const uint8_t *const planes[3], const int strides[3], int nplanes,
                          int boxes[3][4], int vertical, int32_t *found, float *rho,
                          float *theta) {
  static const float rho_multiplier[3] = {1.0f, 2.0f, 2.0f};
  for (int i = 0; i < nplanes && !*found; i++) {
    float lr, lt, sr, st;
    orc_best_line(planes[i], strides[i], boxes[i][0], boxes[i][1], boxes[i][2], boxes[i][3],
                  vertical, &lr, &lt);
    orc_line_by_shifting_origin(lr, lt, boxes[i][0], boxes[i][1], &sr, &st);
    sr *= rho_multiplier[i];
    *rho = sr;
    *theta = st;
    *found = !(st == FLT_MAX);
  }
}

int orc_detect_edges(const uint8_t *y, int y_stride, int w, int h, const uint8_t *cb,
                     const uint8_t *cr, int c_stride, int orientation, orc_frame_result *res) {
  const uint8_t *planes[3] = {y, cb, cr};
  const int strides[3] = {y_stride, c_stride, c_stride};
  const int nplanes = (cb && cr) ? 3 : 1;
  int boxes[3][4][4];
  orc_detection_boxes(w, h, orientation, boxes[0]);
  if (nplanes == 3) {
    orc_detection_boxes(w / 2, h / 2, orientation, boxes[1]);
    orc_detection_boxes(w / 2, h / 2, orientation, boxes[2]);
  }
  /* result order top,left,bottom,right; boxes order top,bottom,left,right */
  static const int box_of_edge[4] = {0, 2, 1, 3};
  static const int vertical_of_edge[4] = {0, 1, 0, 1};
  /* the reference searches top, bottom, left, right; the searches are independent */
  for (int e = 0; e < 4; e++) {
    int b3[3][4];
    for (int p = 0; p < 3; p++) memcpy(b3[p], boxes[p][box_of_edge[e]], sizeof(int) * 4);
    res->found[e] = 0;
    res->rho[e] = 0;
    res->theta[e] = 0;
    orc_find_line(planes, strides, nplanes, b3, vertical_of_edge[e], &res->found[e],
                  &res->rho[e], &res->theta[e]);
  }
  int all = res->found[0] && res->found[1] && res->found[2] && res->found[3];
  memset(res->corners, 0, sizeof(res->corners));
  if (all) {
    /* top=0,left=1,bottom=2,right=3 ; corners tl, bl, tr, br */
    int a = orc_parametric_intersect(res->rho[0], res->theta[0], res->rho[1], res->theta[1],
                                     &res->corners[0], &res->corners[1]);
    int b = orc_parametric_intersect(res->rho[2], res->theta[2], res->rho[1], res->theta[1],
                                     &res->corners[2], &res->corners[3]);
    int c = orc_parametric_intersect(res->rho[0], res->theta[0], res->rho[3], res->theta[3],
                                     &res->corners[4], &res->corners[5]);
    int d = orc_parametric_intersect(res->rho[2], res->theta[2], res->rho[3], res->theta[3],
                                     &res->corners[6], &res->corners[7]);
    all = a && b && c && d;
  }
  res->found_all = all;
  return all;
}

/* ------------------------------------------------------------------------- */
/* warp.cpp:34-125: A x = b by Eigen 3.2.4 HouseholderQR<Matrix8f> (compute():
 * one 8-wide block => householder_qr_inplace_unblocked, HouseholderQR.h:219-250;
 * solve(): apply H_0..H_7 to b then back-substitute, HouseholderQR.h:306-334).
 * Storage is column-major.  Two evaluation orders (`sse`):
 *   0  scalar (EIGEN_DONT_VECTORIZE): every reduction is a sequential sum -- what the reference's ARM builds without NEON
 *      compute and what the device reproduces by default;
 *   1  the order of a stock x86-64 build (eigen.h defines no EIGEN_DONT_VECTORIZE: SSE2 packets).  Only the REDUCTIONS
 *      differ, everything element-wise gives the same bits in packets.  There are three, all evaluated as
 *      <expression>.sum() = a linear vectorised redux (Redux.h:202-245): tail.squaredNorm() in makeHouseholder
 *      (Householder.h:74); every coefficient of essential^T * bottom in applyHouseholderOnTheLeft (Householder.h:125) --
 *      in compute() a coefficient-based product (product_type_selector<1, Small, Small>: a block of an 8 x 8 matrix has
 *      MaxSize 8) whose dynamic-size coefficient is (lhs.row(i).transpose().cwiseProduct(rhs.col(j))).sum()
 *      (CoeffBasedProduct.h), in solve() GeneralProduct<InnerProduct> on the right-hand side vector
 *      (GeneralProduct.h:159-166).  None of the summed expressions has direct access, so first_aligned() is 0 and the
 *      packets start at element 0 (unaligned loads): for n >= 4 terms (n <= 7 here) the first four are ONE Packet4f,
 *      predux = (x0 + x2) + (x1 + x3) (SSE2, PacketMath.h: movehl + add, shuffle + add_ss), the terms 4 .. n-1 follow
 *      one by one; n < 4 terms are summed sequentially.  The triangular solve is one 8-wide panel of element-wise updates.
 *      Pinned on oracle/_ref/libdmzref_vec.so stage by stage (matrixQR, hCoeffs, Q^T b, x) and end to end,
 *      tests/test_oracle_vs_ref.py. */
#define QA(r, c) a[(c) * 8 + (r)]

/* sum of x[0 .. n-1] in the order of the flavour */
static float orc_eigen_redux(const float *x, int n, int sse) {
  float r;
  int i;
  if (sse && n >= 4) {
    r = (x[0] + x[2]) + (x[1] + x[3]);
    i = 4;
  } else {
    r = x[0];
    i = 1;
  }
  for (; i < n; i++) r = r + x[i];
  return r;
}

static void orc_householder_qr_solve8(float *a /* col-major 8x8, destroyed */, float *b, int sse) {
  float hcoef[8];
  for (int k = 0; k < 8; k++) {
    const int rem = 8 - k; /* remainingRows */
    /* makeHouseholderInPlace on a(k..7, k)  (Householder.h:65-93) */
    float tail_sq = 0.0f, terms[8];
    for (int i = 1; i < rem; i++) {
      float v = QA(k + i, k);
      terms[i - 1] = v * v;
    }
    if (rem > 1) tail_sq = orc_eigen_redux(terms, rem - 1, sse);
    float c0 = QA(k, k);
    float tau, beta;
    if (rem == 1 || tail_sq == 0.0f) {
      tau = 0.0f;
      beta = c0;
      for (int i = 1; i < rem; i++) QA(k + i, k) = 0.0f;
    } else {
      beta = sqrtf(c0 * c0 + tail_sq);
      if (c0 >= 0.0f) beta = -beta;
      float denom = c0 - beta;
      for (int i = 1; i < rem; i++) QA(k + i, k) = QA(k + i, k) / denom;
      tau = (beta - c0) / beta;
    }
    hcoef[k] = tau;
    QA(k, k) = beta;
    /* applyHouseholderOnTheLeft to a(k..7, k+1..7)  (Householder.h:112-130) */
    const int rcols = 8 - k - 1;
    if (rcols > 0) {
      if (rem == 1) {
        for (int c = 0; c < rcols; c++) QA(k, k + 1 + c) *= (1.0f - tau);
      } else {
        for (int c = 0; c < rcols; c++) {
          const int col = k + 1 + c;
          float prod[8];
          for (int i = 1; i < rem; i++) prod[i - 1] = QA(k + i, k) * QA(k + i, col);
          float tmp = orc_eigen_redux(prod, rem - 1, sse);
          tmp += QA(k, col);
          QA(k, col) -= tau * tmp;
          for (int i = 1; i < rem; i++) QA(k + i, col) -= (tau * QA(k + i, k)) * tmp;
        }
      }
    }
  }
  /* c = Q^T b : apply H_0 .. H_7 in order (HouseholderSequence transpose) */
  for (int k = 0; k < 8; k++) {
    const int rem = 8 - k;
    const float tau = hcoef[k];
    if (rem == 1) {
      b[k] *= (1.0f - tau);
    } else {
      float terms[8];
      for (int i = 1; i < rem; i++) terms[i - 1] = QA(k + i, k) * b[k + i];
      float tmp = orc_eigen_redux(terms, rem - 1, sse);
      tmp += b[k];
      b[k] -= tau * tmp;
      for (int i = 1; i < rem; i++) b[k + i] -= (tau * QA(k + i, k)) * tmp;
    }
  }
  /* upper-triangular back substitution (TriangularSolverVector, col-major, Upper) */
  for (int i = 7; i >= 0; i--) {
    b[i] = b[i] / QA(i, i);
    for (int r = 0; r < i; r++) b[r] -= b[i] * QA(r, i);
  }
}

static void orc_calc_persp_transform_order(const float sp[8], const float dp[8], float m[9], int sse);
/* which build of the reference orc_transform_card (and with it orc_scan_frame) restates: 0 = Eigen's scalar order (default),
 * 1 = a stock x86-64 build's SSE2 order.  Process-wide: set once, before the oracle's threads start (tests only). */
static int orc_reference_flavour = 0;
void orc_set_reference_flavour(int flavour) { orc_reference_flavour = flavour != 0; }
void orc_calc_persp_transform(const float sp[8], const float dp[8], float m[9]) { orc_calc_persp_transform_order(sp, dp, m, 0); }
/* the same in the summation order of a stock x86-64 (SSE2) build of the reference */
void orc_calc_persp_transform_sse(const float sp[8], const float dp[8], float m[9]) { orc_calc_persp_transform_order(sp, dp, m, 1); }

static void orc_calc_persp_transform_order(const float sp[8], const float dp[8], float m[9], int sse) {
  float a[64], b[8];
  for (int i = 0; i < 4; i++) {
    const float sx = sp[2 * i], sy = sp[2 * i + 1], dx = dp[2 * i], dy = dp[2 * i + 1];
    QA(i, 0) = sx; QA(i, 1) = sy; QA(i, 2) = 1; QA(i, 3) = 0; QA(i, 4) = 0; QA(i, 5) = 0;
    QA(i, 6) = -sx * dx; QA(i, 7) = -sy * dx;
    QA(i + 4, 0) = 0; QA(i + 4, 1) = 0; QA(i + 4, 2) = 0;
    QA(i + 4, 3) = sx; QA(i + 4, 4) = sy; QA(i + 4, 5) = 1;
    QA(i + 4, 6) = -sx * dy; QA(i + 4, 7) = -sy * dy;
    b[i] = dx;
    b[i + 4] = dy;
  }
  orc_householder_qr_solve8(a, b, sse);
  m[0] = b[0]; m[1] = b[1]; m[2] = b[2];
  m[3] = b[3]; m[4] = b[4]; m[5] = b[5];
  m[6] = b[6]; m[7] = b[7]; m[8] = 1.0f;
}

/* ------------------------------------------------------------------------- */
/* cvWarpPerspective(src, dst, M32F, CV_INTER_LINEAR + CV_WARP_FILL_OUTLIERS, 0)
 * (warp.cpp:165), OpenCV 2.4 semantics per SURVEY Appendix A10. */
static int orc_sat_int(double v) { /* saturate_cast<int>(double) after the INT clamp */
  return (int)lrint(v);
}
static int orc_sat16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

void orc_warp_perspective(const uint8_t *src, int stride, int sw, int sh, const float mf[9],
                          uint8_t *dst, int dstride, int dw, int dh) {
  double s[9], M[9];
  for (int i = 0; i < 9; i++) s[i] = (double)mf[i];
  /* cv::invert 3x3 double (adjugate * 1/det) */
  double det = s[0] * (s[4] * s[8] - s[5] * s[7]) - s[1] * (s[3] * s[8] - s[5] * s[6]) +
               s[2] * (s[3] * s[7] - s[4] * s[6]);
  if (det != 0.) {
    double d = 1. / det;
    M[0] = (s[4] * s[8] - s[5] * s[7]) * d;
    M[1] = (s[2] * s[7] - s[1] * s[8]) * d;
    M[2] = (s[1] * s[5] - s[2] * s[4]) * d;
    M[3] = (s[5] * s[6] - s[3] * s[8]) * d;
    M[4] = (s[0] * s[8] - s[2] * s[6]) * d;
    M[5] = (s[2] * s[3] - s[0] * s[5]) * d;
    M[6] = (s[3] * s[7] - s[4] * s[6]) * d;
    M[7] = (s[1] * s[6] - s[0] * s[7]) * d;
    M[8] = (s[0] * s[4] - s[1] * s[3]) * d;
  } else {
    for (int i = 0; i < 9; i++) M[i] = 0.; /* cv::invert zero-fills a singular input */
  }
  const int BLOCK_SZ = 32;
  int bh0 = BLOCK_SZ / 2 < dh ? BLOCK_SZ / 2 : dh;
  int bw0 = BLOCK_SZ * BLOCK_SZ / bh0 < dw ? BLOCK_SZ * BLOCK_SZ / bh0 : dw;
  for (int y = 0; y < dh; y++) {
    for (int x = 0; x < dw; x += bw0) {
      const int bw = bw0 < dw - x ? bw0 : dw - x;
      const double X0 = M[0] * x + M[1] * y + M[2];
      const double Y0 = M[3] * x + M[4] * y + M[5];
      const double W0 = M[6] * x + M[7] * y + M[8];
      for (int x1 = 0; x1 < bw; x1++) {
        double W = W0 + M[6] * x1;
        W = W ? 32. / W : 0;
        double fX = (X0 + M[0] * x1) * W;
        double fY = (Y0 + M[3] * x1) * W;
        fX = fX < (double)INT_MAX ? fX : (double)INT_MAX; /* std::min(INT_MAX, .) */
        fX = fX > (double)INT_MIN ? fX : (double)INT_MIN; /* std::max(INT_MIN, .) */
        fY = fY < (double)INT_MAX ? fY : (double)INT_MAX;
        fY = fY > (double)INT_MIN ? fY : (double)INT_MIN;
        const int X = orc_sat_int(fX), Y = orc_sat_int(fY);
        const int sx = orc_sat16(X >> 5), sy = orc_sat16(Y >> 5);
        const int ax = X & 31, ay = Y & 31;
        /* BilinearTab_i: w = 32768*(1-fx)(1-fy) .. ; the alpha==0 entry
         * {32767,0,0,1} gives the same 8-bit result as {32768,0,0,0} */
        const int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32;
        const int w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
        int v;
        if ((unsigned)sx < (unsigned)(sw - 1) && (unsigned)sy < (unsigned)(sh - 1)) {
          const uint8_t *p = src + (size_t)sy * stride + sx;
          v = (p[0] * w00 + p[1] * w01 + p[stride] * w10 + p[stride + 1] * w11 + (1 << 14)) >> 15;
        } else if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {
          v = 0;
        } else {
          int v0 = (sx >= 0 && sy >= 0 && sx < sw && sy < sh) ? src[(size_t)sy * stride + sx] : 0;
          int v1 = (sx + 1 >= 0 && sy >= 0 && sx + 1 < sw && sy < sh) ? src[(size_t)sy * stride + sx + 1] : 0;
          int v2 = (sx >= 0 && sy + 1 >= 0 && sx < sw && sy + 1 < sh) ? src[(size_t)(sy + 1) * stride + sx] : 0;
          int v3 = (sx + 1 >= 0 && sy + 1 >= 0 && sx + 1 < sw && sy + 1 < sh) ? src[(size_t)(sy + 1) * stride + sx + 1] : 0;
          v = (v0 * w00 + v1 * w01 + v2 * w10 + v3 * w11 + (1 << 14)) >> 15;
        }
        dst[(size_t)y * dstride + x + x1] = (uint8_t)(v > 255 ? 255 : v);
      }
    }
  }
  (void)bh0;
}

/* dmz.cpp:443-497 (1-channel plane, upsample=false) + warp.cpp:153-166 */
void orc_transform_card(const uint8_t *plane, int stride, int w, int h, const float c[8],
                        int orientation, int options, uint8_t *card) {
  const int truncate_corners = options & 1, upsample = options & 2;
  /* corners in: tl(0,1) bl(2,3) tr(4,5) br(6,7) */
  static const int order[5][4] = {
      {0, 2, 1, 3}, /* unused */
      {1, 0, 3, 2}, /* portrait:            bl, tl, br, tr */
      {2, 3, 0, 1}, /* portrait upside down: tr, br, tl, bl */
      {0, 2, 1, 3}, /* landscape right:      tl, tr, bl, br */
      {3, 1, 2, 0}, /* landscape left:       br, bl, tr, tl */
  };
  float sp[8], dp[8];
  const int *o = order[(orientation >= 1 && orientation <= 4) ? orientation : 3];
  for (int i = 0; i < 4; i++) {
    float px = c[2 * o[i]], py = c[2 * o[i] + 1];
    if (truncate_corners) { /* cython_dmz/dmz.pyx:267-270 casts corner points to int */
      px = (float)(int)px;
      py = (float)(int)py;
    }
    if (upsample) { /* dmz.cpp:473-481: Cb/Cr planes are half size */
      px /= 2.0f;
      py /= 2.0f;
    }
    sp[2 * i] = px;
    sp[2 * i + 1] = py;
  }
  /* dmz_rect_get_points(dmz_create_rect(0,0,427,269)) dmz_olm.cpp:31-36 */
  const float rx = 0, ry = 0, rw = ORC_CARD_W - 1, rh = ORC_CARD_H - 1;
  dp[0] = rx; dp[1] = ry; dp[2] = rx + rw; dp[3] = ry;
  dp[4] = rx; dp[5] = ry + rh; dp[6] = rx + rw; dp[7] = ry + rh;
  float m[9];
  orc_calc_persp_transform_order(sp, dp, m, orc_reference_flavour);
  orc_warp_perspective(plane, stride, w, h, m, card, ORC_CARD_W, ORC_CARD_W, ORC_CARD_H);
}

/* dmz_olm.cpp:40-49 */
int orc_passes_luhn(const uint8_t *digits, int n) {
  int even = 0, sum = 0;
  for (int i = n - 1; i >= 0; i--) {
    int addend = digits[i] * (1 << (even++ & 1));
    sum += addend % 10 + addend / 10;
  }
  return sum % 10 == 0;
}
